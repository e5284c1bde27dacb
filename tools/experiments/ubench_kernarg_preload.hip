// Does preloading kernel arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=14) shorten a chain of small dependent kernels?
// Build twice (with / without the flag), run both on the same box:  hipcc --offload-arch=gfx950 -O3 [-mllvm ...] -o ubench ubench_kernarg_preload.hip
// A captured graph of 200 dependent launches (240 workgroups x 256 threads, 16 bytes per thread in and out), time per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void step(const float4 *__restrict__ in, float4 *__restrict__ out, int n, float s)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { float4 v = in[i]; v.x = v.x * s + 1.f; v.y *= s; v.z *= s; v.w *= s; out[i] = v; }
}
int main()
{
    const int n = 240 * 256, L = 200;
    float4 *a, *b;
    CK(hipMalloc(&a, n * sizeof(float4))); CK(hipMalloc(&b, n * sizeof(float4)));
    CK(hipMemset(a, 0, n * sizeof(float4)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) { hipLaunchKernelGGL(step, dim3(240), dim3(256), 0, st, (l & 1) ? b : a, (l & 1) ? a : b, n, 0.5f); }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%.3f us per launch\n", ms * 1e3 / (50.0 * L));
    }
    return 0;
}
