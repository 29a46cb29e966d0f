// streaming-read ceiling on this box: every lane loads 16 B, coalesced, sum kept alive; buffer 24 GB (>> Infinity Cache)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ p, size_t n, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v;
        if (NT) { u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + i)); v = make_uint4(t.x, t.y, t.z, t.w); }
        else v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ p, uint4* __restrict__ q, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) q[i] = p[i];
}
int main()
{
    const size_t bytes = (size_t)24 << 30, n = bytes / 16;
    uint4 *a, *b; unsigned* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes / 2); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes / 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : { 2048, 8192, 32768 }) {
        for (int nt = 0; nt < 2; nt++) {
            float best = 1e9f;
            for (int r = 0; r < 4; r++) {
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, a, n, o); else hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, a, n, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("read  grid %6d nt=%d: %.3f ms -> %.2f TB/s\n", grid, nt, best, bytes / (best * 1e-3) / 1e12);
        }
    }
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, a, b, n / 2); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("copy 12+12 GB: %.3f ms -> %.2f TB/s (read+write)\n", best, bytes / (best * 1e-3) / 1e12);
    return 0;
}
