// Cold instruction fetch of small dependent kernels: does the instruction cache survive a kernel boundary, and what does a cold line cost?
// A captured graph of 200 dependent launches (240 workgroups x BS threads), us per launch, for kernels that execute N dependent
// v_add_u32 (4 bytes each) either straight line (4 N bytes of code, every line cold) or as a 16-instruction loop body (one line).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define STR2(x) #x
#define STR(x) STR2(x)
template <int MODE, int N, int BS>
__global__ __launch_bounds__(BS) void step(const float4 *__restrict__ in, float4 *__restrict__ out, int n, int s)
{
    const int i = blockIdx.x * BS + threadIdx.x;
    float4 v = in[i];
    int x = __float_as_int(v.x);
    if (MODE == 1) {
#pragma unroll 1
        for (int k = 0; k < N / 16; ++k) asm volatile(".rept 16\n\tv_add_u32 %0, %0, %1\n\t.endr" : "+v"(x) : "v"(s));
    } else if (MODE == 2) {
        if (N == 256) asm volatile(".rept 256\n\tv_add_u32 %0, %0, %1\n\t.endr" : "+v"(x) : "v"(s));
        if (N == 1024) asm volatile(".rept 1024\n\tv_add_u32 %0, %0, %1\n\t.endr" : "+v"(x) : "v"(s));
        if (N == 4096) asm volatile(".rept 4096\n\tv_add_u32 %0, %0, %1\n\t.endr" : "+v"(x) : "v"(s));
    }
    v.x = __int_as_float(x);
    out[i] = v;
}
template <int MODE, int N, int BS> int run(const char *name)
{
    const int n = 240 * BS, L = 200;
    float4 *a, *b;
    CK(hipMalloc(&a, n * sizeof(float4))); CK(hipMalloc(&b, n * sizeof(float4)));
    CK(hipMemset(a, 0, n * sizeof(float4)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) { hipLaunchKernelGGL((step<MODE, N, BS>), dim3(240), dim3(BS), 0, st, (l & 1) ? b : a, (l & 1) ? a : b, n, 1); }
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
    printf("%-44s %7.3f us per launch\n", name, ms * 1e3 / (50.0 * L));
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}
int main()
{
    return run<0, 0, 512>("nothing extra, 512 threads")
         | run<1, 256, 512>("256 adds, loop") | run<2, 256, 512>("256 adds, straight line (1 KB)")
         | run<1, 1024, 512>("1024 adds, loop") | run<2, 1024, 512>("1024 adds, straight line (4 KB)")
         | run<1, 4096, 512>("4096 adds, loop") | run<2, 4096, 512>("4096 adds, straight line (16 KB)")
         | run<1, 4096, 64>("4096 adds, loop, 64 threads") | run<2, 4096, 64>("4096 adds, straight line, 64 threads");
}
