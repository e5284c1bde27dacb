// Per-launch cost of a small dependent kernel versus the size of its by-value parameter struct (captured graph of 200 dependent
// launches, 240 x 256 threads): the kernel reads the first fields and ONE int at the end of the struct (LAST 1) or only the first (LAST 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int PAD> struct Args { const float4 *in; float4 *out; int n; int pad[PAD]; };
template <int PAD, int LAST>
__global__ __launch_bounds__(256) void step(const Args<PAD> p)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v = p.in[i];
    v.x += 1.f;
    if (LAST) v.y += (float)p.pad[PAD - 1];
    p.out[i] = v;
}
template <int PAD, int LAST> int run()
{
    const int n = 240 * 256, L = 200;
    float4 *a, *b;
    CK(hipMalloc(&a, n * sizeof(float4))); CK(hipMalloc(&b, n * sizeof(float4)));
    CK(hipMemset(a, 0, n * sizeof(float4))); CK(hipMemset(b, 0, n * sizeof(float4)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) { Args<PAD> p = {}; p.in = (l & 1) ? b : a; p.out = (l & 1) ? a : b; p.n = n; hipLaunchKernelGGL((step<PAD, LAST>), dim3(240), dim3(256), 0, st, p); }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
    const double cpu_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("parameter struct %4zu bytes, last field %s  %7.3f us per launch (host time inside hipGraphLaunch: %.3f us per node)\n", sizeof(Args<PAD>), LAST ? "read  " : "unread", ms * 1e3 / (50.0 * L), cpu_us / (50.0 * L));
    // every object of this case is released again (the ten cases used to leak their graph, exec, stream and events)
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(st));
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}
int main()
{
    return run<1, 1>() | run<11, 1>() | run<27, 1>() | run<43, 1>() | run<59, 1>() | run<93, 1>() | run<93, 0>() | run<123, 1>() | run<123, 0>() | run<251, 1>();
}
