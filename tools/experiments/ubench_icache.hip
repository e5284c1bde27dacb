// Cold instruction fetch of small dependent kernels (the instruction cache does not survive a kernel boundary).
// A captured graph of 200 dependent launches (240 workgroups x BS threads), us per launch, for kernels that execute
//   (0) nothing extra, (1) N dependent v_fma in a 16-instruction loop, (2) N dependent v_fma straight line (8 N bytes of code),
//   (3) N s_add straight line (4 N bytes), (4) N s_add in a 64-instruction loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE, int N, int BS>
__global__ __launch_bounds__(BS) void step(const float4 *__restrict__ in, float4 *__restrict__ out, int n, float s)
{
    const int i = blockIdx.x * BS + threadIdx.x;
    if (i < n) {
        float4 v = in[i];
        float x = v.x;
        int cnt = 0;
        if (MODE == 1) {
#pragma unroll 1
            for (int k = 0; k < N / 16; ++k) {
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(s));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < N; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(s));
        } else if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < N; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(cnt) : : "scc");
        } else if (MODE == 4) {
#pragma unroll 1
            for (int k = 0; k < N / 64; ++k) {
#pragma unroll
                for (int j = 0; j < 64; ++j) asm volatile("s_add_u32 %0, %0, 1" : "+s"(cnt) : : "scc");
            }
        }
        v.x = x * s + 1.f + (float)(cnt & 1);
        out[i] = v;
    }
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
    for (int l = 0; l < L; ++l) { hipLaunchKernelGGL((step<MODE, N, BS>), dim3(240), dim3(BS), 0, st, (l & 1) ? b : a, (l & 1) ? a : b, n, 0.5f); }
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
    return run<0, 0, 256>("nothing extra, 256 threads") | run<0, 0, 512>("nothing extra, 512 threads")
         | run<1, 256, 512>("256 fma loop") | run<2, 256, 512>("256 fma straight (2 KB)")
         | run<1, 512, 512>("512 fma loop") | run<2, 512, 512>("512 fma straight (4 KB)")
         | run<1, 1024, 512>("1024 fma loop") | run<2, 1024, 512>("1024 fma straight (8 KB)")
         | run<1, 2048, 512>("2048 fma loop") | run<2, 2048, 512>("2048 fma straight (16 KB)")
         | run<1, 2048, 64>("2048 fma loop, 64 threads") | run<2, 2048, 64>("2048 fma straight (16 KB), 64 threads");
}
