// Does the store policy of a small dependent kernel change what the kernel boundary costs (the release at kernel end writes dirty L2 lines
// back)?  Captured graph of 200 dependent launches, 320 workgroups x 256 threads, each thread 2 x 16 bytes (2.6 MB in, 2.6 MB out per
// launch, cross-XCD: workgroup b reads what b + 3 wrote); plain stores vs __builtin_nontemporal_store vs sc0 sc1 (write-through) stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) float f4;
template <int MODE, int LD>
__global__ __launch_bounds__(256) void step(const f4 *__restrict__ in, f4 *__restrict__ out, int n)
{
    const int b = (blockIdx.x + 3) % gridDim.x;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = (b * 256 + threadIdx.x) * 2 + k, o = (blockIdx.x * 256 + threadIdx.x) * 2 + k;
        f4 v;
        if (LD == 1) v = __builtin_nontemporal_load(in + i); else v = in[i];
        v.x += 1.f;
        if (MODE == 0) out[o] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, out + o);
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(out + o), "v"(v) : "memory");
    }
}
template <int MODE, int LD> int run(const char *name)
{
    const int G = 320, n = G * 256 * 2, L = 200;
    f4 *a, *b;
    CK(hipMalloc(&a, n * sizeof(f4))); CK(hipMalloc(&b, n * sizeof(f4)));
    CK(hipMemset(a, 0, n * sizeof(f4))); CK(hipMemset(b, 0, n * sizeof(f4)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < L; ++l) hipLaunchKernelGGL((step<MODE, LD>), dim3(G), dim3(256), 0, st, (l & 1) ? b : a, (l & 1) ? a : b, n);
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
    printf("%-50s %7.3f us per launch\n", name, ms * 1e3 / (50.0 * L));
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}
int main()
{
    return run<0, 0>("plain stores") | run<1, 0>("nontemporal stores") | run<2, 0>("sc0 sc1 stores") | run<0, 1>("plain stores, nontemporal loads")
         | run<1, 1>("nontemporal stores and loads") | run<0, 0>("plain stores (again)");
}
