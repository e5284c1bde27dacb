// What does the SHAPE of a small dependent kernel cost per launch (captured graph of 200 dependent launches)?  Workgroup size, dynamic LDS,
// grid size, a 400-byte by-value parameter struct, 128 / 256 VGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
struct Big { const float4 *in; float4 *out; int n; int pad[93]; };
template <int BS>
__global__ __launch_bounds__(BS) void step(const float4 *__restrict__ in, float4 *__restrict__ out, int n)
{
    extern __shared__ float4 sm[];
    const int i = (blockIdx.x * BS + threadIdx.x) % n;
    float4 v = in[i];
    if (n < 0) sm[threadIdx.x] = v;      // (keeps the dynamic LDS referenced)
    v.x += 1.f;
    out[i] = v;
}
template <int BS>
__global__ __launch_bounds__(BS) void step_big(const Big p)
{
    const int i = (blockIdx.x * BS + threadIdx.x) % p.n;
    float4 v = p.in[i];
    v.x += 1.f + (float)p.pad[92];
    p.out[i] = v;
}
template <int BS, int BIG> int run(const char *name, int grid, size_t lds)
{
    const int n = 240 * 512, L = 200;
    float4 *a, *b;
    CK(hipMalloc(&a, n * sizeof(float4))); CK(hipMalloc(&b, n * sizeof(float4)));
    CK(hipMemset(a, 0, n * sizeof(float4))); CK(hipMemset(b, 0, n * sizeof(float4)));
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void *)step<BS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) {
        if (BIG) { Big p = {}; p.in = (l & 1) ? b : a; p.out = (l & 1) ? a : b; p.n = n; hipLaunchKernelGGL((step_big<BS>), dim3(grid), dim3(BS), 0, st, p); }
        else hipLaunchKernelGGL((step<BS>), dim3(grid), dim3(BS), lds, st, (l & 1) ? b : a, (l & 1) ? a : b, n);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-56s %7.3f us per launch\n", name, ms * 1e3 / (50.0 * L));
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}
int main()
{
    return run<256, 0>("240 x 256 threads", 240, 0) | run<512, 0>("240 x 512 threads", 240, 0) | run<512, 0>("240 x 512 threads, 64 KB LDS", 240, 64 * 1024)
         | run<512, 0>("240 x 512 threads, 128 KB LDS", 240, 128 * 1024) | run<512, 0>("50 x 512 threads, 128 KB LDS", 50, 128 * 1024)
         | run<256, 0>("32 x 256 threads", 32, 0) | run<256, 0>("960 x 256 threads", 960, 0) | run<512, 0>("480 x 512 threads, 64 KB LDS", 480, 64 * 1024)
         | run<512, 1>("240 x 512 threads, 400-byte parameter struct", 240, 0) | run<1024, 0>("240 x 1024 threads", 240, 0);
}
