// Shared device/host helpers for libguidegen_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/guidegen_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define GG_WAVE 64

// thread-local error message (error convention)
void gg_set_error(const char *fmt, ...);

#define GG_FAIL(code, ...)            \
    do {                              \
        gg_set_error(__VA_ARGS__);    \
        return (code);                \
    } while (0)

#define GG_CHECK_LAUNCH()                                                              \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) GG_FAIL(GG_ERR_HIP, "HIP launch: %s", hipGetErrorString(e_)); \
    } while (0)

// Kernel arguments reach a wave through scalar loads from the kernarg segment, and the compiler sinks each load to its first use: a
// kernel that touches its arguments in four places pays four SERIAL scalar round trips before its first memory instruction.
// gg_pin launders a wave-uniform value through an SGPR: the pinned arguments are fetched in one batch where the pins stand and
// cannot be re-materialised from the kernarg segment later.  (Pin only what the first memory instructions need: pinning all ~50
// dwords of a conv descriptor raised the SGPR pressure and was slower.)
template <class T>
__device__ __forceinline__ T gg_pin(T v)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "gg_pin: 32- or 64-bit scalars");
    if constexpr (sizeof(T) == 4) {
        unsigned u = __builtin_bit_cast(unsigned, v);
        asm volatile("" : "+s"(u));
        return __builtin_bit_cast(T, u);
    } else {
        unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        asm volatile("" : "+s"(u));
        return __builtin_bit_cast(T, u);
    }
}

// Weights of the latent UNet's batch-1 convs are read ONCE per forward (535 MB per forward against 8 x 4 MB of L2 and 256 MB of MALL):
// loaded with the default policy they evict the activations, accumulators and parameters the NEXT kernels are about to read.
// GG_STREAM_LOAD marks them non-temporal (measured: tools/experiments/README.md, "non-temporal weight loads").
#ifndef GG_STREAM_WEIGHTS
#define GG_STREAM_WEIGHTS 1
#endif
#if GG_STREAM_WEIGHTS
#define GG_STREAM_LOAD(P) __builtin_nontemporal_load(P)
#else
#define GG_STREAM_LOAD(P) (*(P))
#endif

// Integer division is a ~30 (32-bit) to ~150 (64-bit) instruction sequence on this ISA, and a wave issues its instructions one by
// one: in the launch-bound kernels of the latent UNet an index division IS microseconds.
// gg_fastdiv: n / d through one multiply-high; exact for 0 <= n, n * d < 2^32 (gg_magic_u32 returns 0 when that does not hold or
// d == 1: plain division then).  gg_div_small: n / d for 0 <= n < 2^21 through the fp32 reciprocal (rcp = v_rcp_f32 of d, 1 ulp).
__device__ __forceinline__ long long gg_fastdiv(long long n, int d, unsigned magic)
{
    return magic ? (long long)__umulhi((unsigned)n, magic) : n / d;
}
__device__ __forceinline__ int gg_div_small(int n, float rcp) { return (int)(((float)n + 0.5f) * rcp); }
static inline unsigned gg_magic_u32(long long nmax, int d)
{
    return (d > 1 && nmax * (long long)d < (1LL << 32)) ? (unsigned)(((1ULL << 32) + (unsigned)d - 1) / (unsigned)d) : 0u;
}

// SiLU with the hardware reciprocal (v_rcp_f32, 1 ulp), as the conv prologues compute it: an IEEE division is ~10 VALU instructions per
// element, 16 elements per thread in the launch-bound GroupNorm kernels
__device__ __forceinline__ float gg_silu(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }

__device__ __forceinline__ f32x8 gg_bf16x8_to_f32(bf16x8 v)
{
    f32x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (float)v[i];
    return r;
}
__device__ __forceinline__ bf16x8 gg_f32_to_bf16x8(f32x8 v)
{
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[i];
    return r;
}

static inline int gg_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
