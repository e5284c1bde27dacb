// Issue cost of the vector instructions the fused GroupNorm*SiLU prologue is made of, wave64 on gfx950: cycles (s_memtime) per instruction
// per SIMD with 1 and 2 waves per SIMD, 16 independent chains of each instruction.   hipcc --offload-arch=gfx950 -O3 ubench_valu.hip -o ub_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(INS) \
    for (int it = 0; it < iters; ++it) { asm volatile(REP16(INS "\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1) : "v"(c0), "v"(c1)); }
template <int K>
__global__ __launch_bounds__(1024) void k(long long *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 0.5f, a2 = a0 * 0.25f, a3 = a0 + 3.f, c0 = 0.999f, c1 = 0.001f;
    double b0d = a0, b1d = a1;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 b0 = {a0, a1}, b1 = {a2, a3};
    long long t0 = __builtin_amdgcn_s_memtime();
    if (K == 0) BODY("v_fma_f32 %0, %0, %6, %7")
    if (K == 1) BODY("v_exp_f32 %0, %0")
    if (K == 2) BODY("v_rcp_f32 %0, %0")
    if (K == 3) BODY("v_exp_f16 %0, %0")
    if (K == 4) BODY("v_rcp_f16 %0, %0")
    if (K == 5) BODY("v_pk_fma_f32 %4, %4, %5, %5")
    if (K == 6) BODY("v_pk_mul_f32 %4, %4, %5")
    if (K == 7) BODY("v_pk_fma_f16 %0, %0, %6, %7")
    if (K == 8) BODY("v_cvt_pk_bf16_f32 %0, %0, %1")
    if (K == 9) BODY("v_lshlrev_b32 %0, 16, %0")
    if (K == 10) BODY("v_cndmask_b32 %0, %0, %1, vcc")
    if (K == 11) BODY("v_mul_f32 %0, %0, %6")
    if (K == 12) BODY("v_add_f32 %0, 1.0, %0")
    if (K == 13) BODY("v_exp_f32 %0, %0\nv_fma_f32 %1, %1, %6, %7\nv_fma_f32 %2, %2, %6, %7")
    if (K == 14) BODY("v_exp_f32 %0, %1\nv_exp_f32 %2, %3")
    if (K == 15) BODY("v_cndmask_b32_e64 %0, %0, %1, s[4:5]")
    if (K == 16) BODY("v_cndmask_b32_e64 %0, 0, %1, s[4:5]\nv_cndmask_b32_e64 %2, 0, %3, s[4:5]")
    if (K == 17) BODY("v_and_b32 %0, %0, %1")
    if (K == 18) BODY("v_cmp_gt_u32 vcc, %0, %1")
    if (K == 19) BODY("v_fma_f32 %0, %1, %6, %7\nv_fma_f32 %2, %3, %6, %7")
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a0 + a1 + a2 + a3 + b0[0] + b1[1] + (float)b0d + (float)b1d == 1234.5f) out[1] = 1;
}
const char *names[] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_exp_f16", "v_rcp_f16", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_fma_f16", "v_cvt_pk_bf16_f32",
                       "v_lshlrev_b32", "v_cndmask_b32", "v_mul_f32", "v_add_f32", "exp+2fma (3 instr)", "2 independent v_exp (2 instr)", "v_cndmask_e64 sgpr cond (dep)", "2 indep v_cndmask_e64", "v_and_b32", "v_cmp_gt_u32 vcc", "2 independent v_fma"};
template <int K> void run(long long *d, int threads)
{
    const int iters = 2000;
    k<K><<<256, threads>>>(d, iters, 1.0f);
    k<K><<<256, threads>>>(d, iters, 1.0f);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const int per = (K == 13) ? 3 : (K == 14 || K == 16 || K == 19) ? 2 : 1;
    printf("%-32s %d waves/SIMD: %.2f cycles per instruction and wave, %.2f per SIMD\n", names[K], threads / 256, (double)h[0] / (iters * 16.0 * per),
           (double)h[0] / (iters * 16.0 * per) / (threads / 256));
}
int main()
{
    long long *d;
    hipMalloc(&d, 64);
    for (int threads : {256, 512, 1024}) {
        run<0>(d, threads); run<11>(d, threads); run<12>(d, threads); run<1>(d, threads); run<2>(d, threads); run<3>(d, threads); run<4>(d, threads); run<5>(d, threads);
        run<6>(d, threads); run<7>(d, threads); run<8>(d, threads); run<9>(d, threads); run<10>(d, threads); run<13>(d, threads); run<14>(d, threads); run<15>(d, threads); run<16>(d, threads); run<17>(d, threads); run<18>(d, threads); run<19>(d, threads);
    }
    return 0;
}
