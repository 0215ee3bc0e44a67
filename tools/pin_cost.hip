// cost of page-locking host memory for the output planes of a picture: hipHostMalloc vs hipHostRegister of malloc'ed memory, and the
// device-to-host rate into each (gfx950 box; build: hipcc -O2 -o /tmp/pin_cost tools/pin_cost.hip)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t sizes[2] = { (size_t)66 << 20, (size_t)17 << 20 };
    void *dev = nullptr;
    if (hipMalloc(&dev, sizes[0]) != hipSuccess) return 1;
    hipStream_t st; hipStreamCreate(&st);
    for (size_t n : sizes) {
        for (int rep = 0; rep < 3; rep++) {
            double t0 = now(); void *p = nullptr; hipHostMalloc(&p, n, hipHostMallocDefault); double t1 = now();
            hipMemcpyAsync(p, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); double t2 = now();
            hipMemcpyAsync(p, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); double t3 = now();
            hipHostFree(p); double t4 = now();
            printf("%3zu MiB hipHostMalloc %.2f ms, first D2H %.2f ms, second D2H %.2f ms (%.1f GB/s), hipHostFree %.2f ms\n", n >> 20, t1 - t0, t2 - t1, t3 - t2, n / (t3 - t2) / 1e6, t4 - t3);
            t0 = now(); void *m = malloc(n); double ta = now(); hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault); t1 = now();
            hipMemcpyAsync(m, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); t2 = now();
            hipMemcpyAsync(m, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); t3 = now();
            hipHostUnregister(m); t4 = now(); free(m);
            printf("%3zu MiB malloc %.3f ms + hipHostRegister %.2f ms (%s), first D2H %.2f ms, second D2H %.2f ms (%.1f GB/s), unregister %.2f ms\n", n >> 20, ta - t0, t1 - ta, hipGetErrorString(e), t2 - t1, t3 - t2, n / (t3 - t2) / 1e6, t4 - t3);
            t0 = now(); void *q = malloc(n); memset(q, 1, n); t1 = now(); memset(q, 2, n); t2 = now();
            hipMemcpyAsync(q, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); t3 = now(); free(q);
            printf("%3zu MiB malloc + first touch %.2f ms, second memset %.2f ms, D2H into pageable memory %.2f ms (%.1f GB/s)\n", n >> 20, t1 - t0, t2 - t1, t3 - t2, n / (t3 - t2) / 1e6);
        }
    }
    return 0;
}
