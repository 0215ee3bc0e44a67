// how fast a kernel pulls work lists out of page-locked host memory over PCIe (against the copy engines' 55-57 GB/s):
// grid-stride copy of 16-byte units from hipHostMalloc memory into HBM, for several grid sizes and segment counts
// build: hipcc -O2 --offload-arch=gfx950 -o /tmp/pull_rate tools/pull_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
struct Seg { const u4 *src; u4 *dst; unsigned n; unsigned pad; };
__global__ __launch_bounds__(256) void pull(const Seg *segs, int nseg)
{
    for (int s = blockIdx.y; s < nseg; s += gridDim.y) {
        const Seg g = segs[s];
        for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < g.n; i += gridDim.x * 256)
            g.dst[i] = __builtin_nontemporal_load(g.src + i);
    }
}
int main()
{
    const size_t total = (size_t)132 << 20;                  // a batch of 32 work lists of 4.1 MB
    void *h = nullptr, *d = nullptr; Seg *st = nullptr;
    CK(hipHostMalloc(&h, total, hipHostMallocDefault)); memset(h, 1, total);
    CK(hipMalloc(&d, total));
    CK(hipHostMalloc((void **)&st, 4096 * sizeof(Seg), hipHostMallocDefault));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nseg : { 32, 480 })                              // one segment per list, or fifteen
        for (int gx : { 8, 32, 128, 512 }) {
            const size_t per = total / nseg / 16 * 16;
            for (int i = 0; i < nseg; i++) st[i] = Seg{ (const u4 *)((char *)h + i * per), (u4 *)((char *)d + i * per), (unsigned)(per / 16), 0 };
            const int gy = nseg < 64 ? nseg : 64;
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(pull, dim3(gx, gy), dim3(256), 0, s, st, nseg);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("%3d segments, grid %3d x %2d workgroups: %.2f ms = %.1f GB/s\n", nseg, gx, gy, best, per * nseg / best / 1e6);
        }
    float ms;
    CK(hipEventRecord(e0, s)); CK(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("copy engine, one 132 MiB copy: %.2f ms = %.1f GB/s\n", ms, total / ms / 1e6);
    return 0;
}
