// Does it matter WHICH workgroup wrote the data a dependent small kernel reads?  A captured graph of 200 dependent launches
// (240 workgroups x 256 threads, 16 bytes per thread); workgroup b reads what workgroup (b + SHIFT) % 240 of the previous launch wrote
// (SHIFT 0: same XCD -- consecutive workgroup ids round-robin over the 8 XCDs; SHIFT 1: the neighbouring XCD; SHIFT 8: same XCD, other CU).
// Also: a second, independent read of a 4 KB constant table (as scale/shift rows), and 2 / 8 dependent loads (pointer chase through the data).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int SHIFT, int CHASE>
__global__ __launch_bounds__(256) void step(const float4 *__restrict__ in, float4 *__restrict__ out, int n)
{
    const int b = (blockIdx.x + SHIFT) % 240;
    int i = b * 256 + threadIdx.x;
    float4 v = in[i];
#pragma unroll
    for (int c = 0; c < CHASE; ++c) { i = (i + 256 * 8 + (__float_as_int(v.y) & 1)) % n; v = in[i]; }     // dependent loads (v.y is 0)
    v.x += 1.f;
    out[blockIdx.x * 256 + threadIdx.x] = v;
}
template <int SHIFT, int CHASE> int run(const char *name)
{
    const int n = 240 * 256, L = 200;
    float4 *a, *b;
    CK(hipMalloc(&a, n * sizeof(float4))); CK(hipMalloc(&b, n * sizeof(float4)));
    CK(hipMemset(a, 0, n * sizeof(float4))); CK(hipMemset(b, 0, n * sizeof(float4)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) { hipLaunchKernelGGL((step<SHIFT, CHASE>), dim3(240), dim3(256), 0, st, (l & 1) ? b : a, (l & 1) ? a : b, n); }
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
    printf("%-52s %7.3f us per launch\n", name, ms * 1e3 / (50.0 * L));
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}
int main()
{
    return run<0, 0>("reads its own workgroup's data") | run<1, 0>("reads the next workgroup's (next XCD)") | run<8, 0>("reads workgroup + 8 (same XCD)")
         | run<3, 0>("reads workgroup + 3") | run<0, 1>("own data + 1 dependent load") | run<0, 2>("own data + 2 dependent loads") | run<0, 4>("own data + 4 dependent loads")
         | run<1, 4>("next XCD + 4 dependent loads");
}
