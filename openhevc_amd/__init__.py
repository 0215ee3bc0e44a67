"""openhevc_amd — MI355X-native HEVC block-reconstruction engine (drop-in under openHEVC's CTU loop).

Python here is plumbing only (ctypes over the C ABI in include/); the product is
libohevc_hip.so (HIP kernels + engine) and libohevc_host.so (recorder + synthetic streams).
"""
