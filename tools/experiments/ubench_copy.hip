// Stream-copy variants (blocks per CU x 16-byte loads in flight per thread) to pick the form gg_ubench_stream_copy uses.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int U>
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ s, u32x4 *__restrict__ d, long long n)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(s + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], d + i + u * stride);
    }
    for (; i < n; i += stride) d[i] = s[i];
}
template <int U> void run(const u32x4 *s, u32x4 *d, long long n, int bpc)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        k<U><<<256 * bpc, 256>>>(s, d, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    printf("unroll %d, %3d blocks per CU: %.1f GB/s (read + write)\n", U, bpc, 2.0 * n * 16 / best / 1e6);
}
int main()
{
    const long long bytes = 1LL << 30, n = bytes / 16;
    u32x4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
    for (int bpc : {4, 8, 16, 32, 64}) { run<1>(s, d, n, bpc); run<2>(s, d, n, bpc); run<4>(s, d, n, bpc); run<8>(s, d, n, bpc); }
    return 0;
}
