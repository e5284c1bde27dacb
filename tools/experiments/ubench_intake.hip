// What can ONE CU take in, and how does it depend on the row length of a gather?  (tools/experiments/README.md)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_intake tools/experiments/ubench_intake.hip && /tmp/ubench_intake
// Every workgroup (8 waves) pulls KB kilobytes (L2-resident after the warm-up launch) as 16-byte-per-lane loads whose 64 lanes
// cover rows of ROWB contiguous bytes spaced STRIDE bytes apart, either through registers or by LDS-DMA; wave 0 stamps
// s_memrealtime (100 MHz) before the first load is issued and after the last one has landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int DMA>
__global__ __launch_bounds__(512) void intake(const char *src, long long wg_stride, int rowb, int stride, int per_wave, unsigned long long *stamps, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lanes_per_row = rowb == 32 ? 4 : rowb / 16, rows_per_instr = 64 / lanes_per_row;
    const char *base = src + (long long)blockIdx.x * wg_stride;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    for (int i = 0; i < per_wave; ++i) {
        const int instr = wave * per_wave + i;
        long long row = (long long)instr * rows_per_instr + lane / lanes_per_row;
        int col = (lane % lanes_per_row) * 16;
        if (rowb == 32) {   // "64 x 2": consecutive instructions of a wave read the two 64-byte halves of the SAME 16 rows (128-byte lines)
            row = (long long)(instr >> 1) * 16 + lane / 4;
            col = (instr & 1) * 64 + (lane % 4) * 16;
        }
        const char *p = base + row * stride + col;
        if (DMA) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                             (__attribute__((address_space(3))) void *)(lds + (instr % 128) * 1024), 16, 0, 0);
        } else {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(p);
            acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    if (!DMA) asm volatile("" :: "v"(acc));
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = t1; }
    if (acc == 0x12345678u) sink[0] = acc + lds[lane];
}

int main()
{
    const size_t bytes = 1024ull << 20;
    char *src; unsigned long long *st; unsigned *sink;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    hipMalloc(&st, 4096 * 16); hipMalloc(&sink, 64);
    hipFuncSetAttribute((const void *)intake<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void *)intake<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    std::vector<unsigned long long> h(2048);
    printf("%-8s %5s %6s %7s %5s %5s | %8s %10s %12s\n", "path", "WGs", "rowB", "stride", "KB/WG", "share", "us/WG", "GB/s/CU", "aggregate TB/s");
    for (int wgs : {240, 48})
        for (int dma : {1, 0})
            for (int shared : {0, 1})                  // 1: all workgroups read the SAME bytes (hot lines), 0: disjoint regions
                for (int rowb : {64, 32, 128, 1024}) {
                    const int per_wave = 16;               // 16 KiB per wave, 128 KiB per workgroup
                    const int stride = rowb == 1024 ? 1024 : 1280;
                    const long long rows = rowb == 32 ? 8LL * per_wave * 8 : 8LL * per_wave * (1024 / rowb);
                    const long long wg_stride = shared ? 0 : ((rows * stride + 4095) / 4096) * 4096;
                    if ((unsigned long long)(wgs - 1) * wg_stride + (unsigned long long)rows * stride + 4096 > bytes) { printf("skip (out of range)\n"); return 1; }
                    double best = 1e9;
                    for (int rep = 0; rep < 4; ++rep) {
                        if (dma) hipLaunchKernelGGL(intake<1>, dim3(wgs), dim3(512), 128 * 1024, 0, src, wg_stride, rowb, stride, per_wave, st, sink);
                        else hipLaunchKernelGGL(intake<0>, dim3(wgs), dim3(512), 128 * 1024, 0, src, wg_stride, rowb, stride, per_wave, st, sink);
                        hipDeviceSynchronize();
                        hipMemcpy(h.data(), st, wgs * 16, hipMemcpyDeviceToHost);
                        double sum = 0;
                        for (int b = 0; b < wgs; ++b) sum += (double)(h[2 * b + 1] - h[2 * b]) / 100.0;
                        if (rep) best = std::min(best, sum / wgs);
                    }
                    const double kb = 8.0 * per_wave;
                    printf("%-8s %5d %6d %7d %5.0f %5s | %8.2f %10.1f %12.2f\n", dma ? "lds-dma" : "vgpr", wgs, rowb, stride, kb, shared ? "same" : "own", best,
                           kb * 1024 / best / 1e3, kb * 1024 * wgs / best / 1e6);
                }
    return 0;
}
