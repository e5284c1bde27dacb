// On-box peak measurements for bench.py's roofline ("the harness must also measure an MFMA microbench and a
// stream-copy on the box and report against both").  Measurement infrastructure, not on the sampling path.
//   gg_ubench_mfma_bf16:  register-resident bf16 MFMA loop on every SIMD of every CU (random non-zero operands, independent
//                         accumulators, no memory traffic inside the loop) -> what the matrix pipes deliver at the clock the chip
//                         holds under that load (MI355X_MICROARCH.md, DVFS give-back: well below 2.4 GHz on random data)
//   gg_ubench_stream_copy: 16 bytes per lane grid-stride copy -> achievable HBM bandwidth (read + write bytes)
#include "gg_common.h"

// hash -> bf16 in +-[0.5, 2): every operand register differs per lane and per register (no zeros, no repeated fragments)
__device__ __forceinline__ bf16_t ub_val(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float m = 0.5f + (float)(x & 0xFFFF) * (1.5f / 65536.0f);
    return (bf16_t)((x & 0x10000u) ? -m : m);
}

template <int SHAPE>   // 0: v_mfma_f32_16x16x32_bf16 (8 independent accumulators), 1: v_mfma_f32_32x32x16_bf16 (4)
__global__ __launch_bounds__(256) void ubench_mfma_kernel(const int iters, float *sink)
{
    bf16x8 a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[i][e] = ub_val((threadIdx.x * 4 + i) * 8 + e + blockIdx.x * 8192);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) b[i][e] = ub_val(0x9E3779B9u + (threadIdx.x * 2 + i) * 8 + e + blockIdx.x * 8192);
    float total = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)                 // 16 MFMAs per iteration
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) ^ r], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        typedef __attribute__((ext_vector_type(16))) float f32x16;
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)                 // 8 MFMAs per iteration (same FLOPs as 16 of the 16x16x32 form)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[r], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) total += acc[i][e];
    }
    if (total == 12345.678f) sink[0] = total;          // keeps the loop alive; practically never taken
}

__global__ __launch_bounds__(256) void ubench_copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, const long long n16)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    // non-temporal loads and stores, 4 workgroups per CU: the fastest of the forms tried (tools/experiments/ubench_copy.hip: 6.3 TB/s;
    // 8-64 workgroups per CU or 2-8 loads in flight per thread: 4.7-5.6)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

extern "C" int gg_ubench_mfma_bf16(int32_t shape, int32_t iters, int32_t waves_per_simd, float *sink, double *flops_out, void *stream)
{
    if ((shape != 0 && shape != 1) || iters < 1 || waves_per_simd < 1 || waves_per_simd > 8 || !sink || !flops_out)
        GG_FAIL(GG_ERR_BAD_SHAPE, "gg_ubench_mfma_bf16: shape in {0,1}, iters >= 1, waves_per_simd in 1..8");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        GG_FAIL(GG_ERR_HIP, "gg_ubench_mfma_bf16: device query failed");
    const int blocks = cus * waves_per_simd;            // 256 threads = one wave per SIMD of a CU
    if (shape == 0) hipLaunchKernelGGL(ubench_mfma_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
    else hipLaunchKernelGGL(ubench_mfma_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
    GG_CHECK_LAUNCH();
    // 16 MFMAs of 16x16x32 (or 8 of 32x32x16) per iteration and wave: 16 * 2*16*16*32 = 8 * 2*32*32*16 = 262144 FLOP
    *flops_out = (double)blocks * 4.0 * (double)iters * 262144.0;
    return GG_OK;
}

extern "C" int gg_ubench_stream_copy(const void *src, void *dst, int64_t bytes, void *stream)
{
    if (!src || !dst || bytes < 16 || (bytes & 15)) GG_FAIL(GG_ERR_BAD_SHAPE, "gg_ubench_stream_copy: bytes must be a positive multiple of 16");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        GG_FAIL(GG_ERR_HIP, "gg_ubench_stream_copy: device query failed");
    hipLaunchKernelGGL(ubench_copy_kernel, dim3(cus * 4), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)src, (u32x4 *)dst, (long long)(bytes >> 4));
    GG_CHECK_LAUNCH();
    return GG_OK;
}
