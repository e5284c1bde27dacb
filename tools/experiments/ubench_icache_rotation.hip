// Is the instruction cache warm when a captured chain ROTATES through different kernels (as a network forward does)?  200 dependent
// launches of kernels that each execute 2048 straight-line 4-byte VALU instructions (8 KB of code), rotating through NK distinct copies
// (NK x 8 KB of code: 1 copy = the same kernel every time, 32 copies = 256 KB, far beyond a 64 KB instruction cache); us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int ID, int N>
__global__ __launch_bounds__(256) void step(const float4 *__restrict__ in, float4 *__restrict__ out, int s)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float4 v = in[i];
    int x = __float_as_int(v.x) + ID;
    if (N == 2048) asm volatile(".rept 2048\n\tv_add_u32 %0, %0, %1\n\t.endr" : "+v"(x) : "v"(s));
    if (N == 256) asm volatile(".rept 256\n\tv_add_u32 %0, %0, %1\n\t.endr" : "+v"(x) : "v"(s));
    v.x = __int_as_float(x);
    out[i] = v;
}
typedef void (*kfn)(const float4 *, float4 *, int);
template <int N> struct Tab {
    static kfn get(int id) {
        switch (id) {
#define K(I) case I: return step<I, N>;
            K(0) K(1) K(2) K(3) K(4) K(5) K(6) K(7) K(8) K(9) K(10) K(11) K(12) K(13) K(14) K(15)
            K(16) K(17) K(18) K(19) K(20) K(21) K(22) K(23) K(24) K(25) K(26) K(27) K(28) K(29) K(30) K(31)
#undef K
        }
        return nullptr;
    }
};
template <int N> int run(int NK)
{
    const int n = 240 * 256, L = 192;
    float4 *a, *b;
    CK(hipMalloc(&a, n * sizeof(float4))); CK(hipMalloc(&b, n * sizeof(float4)));
    CK(hipMemset(a, 0, n * sizeof(float4))); CK(hipMemset(b, 0, n * sizeof(float4)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) hipLaunchKernelGGL(Tab<N>::get(l % NK), dim3(240), dim3(256), 0, st, (l & 1) ? b : a, (l & 1) ? a : b, 1);
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
    printf("%4d straight-line instructions per kernel, %2d distinct kernels in rotation: %7.3f us per launch\n", N, NK, ms * 1e3 / (50.0 * L));
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}
int main() { return run<2048>(1) | run<2048>(2) | run<2048>(4) | run<2048>(8) | run<2048>(16) | run<2048>(32) | run<256>(1) | run<256>(8) | run<256>(32) | run<2048>(1); }
