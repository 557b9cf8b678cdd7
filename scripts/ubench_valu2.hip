// Finer issue-cost table: one instruction type per kernel, 8 independent destinations,
// 8 waves per SIMD (throughput) and 1 wave per SIMD (single-wave issue interval).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/u2 scripts/ubench_valu2.hip && /tmp/u2
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define I8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)

#define KERNEL(NAME, BODY)                                                                         \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters)                             \
    {                                                                                              \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = out[threadIdx.x & 63], c = out[(threadIdx.x + 1) & 63];                          \
        for (int i = 0; i < iters; ++i) {                                                          \
            REP16(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(0x7fffffff) : "vcc");) \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;              \
    }

#define R8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
KERNEL(k_add_e32, "v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7\n")
KERNEL(k_add_e64, "v_add_f32_e64 %0, %8, %0\n v_add_f32_e64 %1, %8, %1\n v_add_f32_e64 %2, %8, %2\n v_add_f32_e64 %3, %8, %3\n v_add_f32_e64 %4, %8, %4\n v_add_f32_e64 %5, %8, %5\n v_add_f32_e64 %6, %8, %6\n v_add_f32_e64 %7, %8, %7\n")
KERNEL(k_mul_e32, "v_mul_f32_e32 %0, %8, %0\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n v_mul_f32_e32 %4, %8, %4\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n")
KERNEL(k_fmac_e32, "v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n")
KERNEL(k_fma_vop3, "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")
KERNEL(k_min_e32, "v_min_f32_e32 %0, %8, %0\n v_min_f32_e32 %1, %8, %1\n v_min_f32_e32 %2, %8, %2\n v_min_f32_e32 %3, %8, %3\n v_min_f32_e32 %4, %8, %4\n v_min_f32_e32 %5, %8, %5\n v_min_f32_e32 %6, %8, %6\n v_min_f32_e32 %7, %8, %7\n")
KERNEL(k_xor_e32, "v_xor_b32_e32 %0, %8, %0\n v_xor_b32_e32 %1, %8, %1\n v_xor_b32_e32 %2, %8, %2\n v_xor_b32_e32 %3, %8, %3\n v_xor_b32_e32 %4, %8, %4\n v_xor_b32_e32 %5, %8, %5\n v_xor_b32_e32 %6, %8, %6\n v_xor_b32_e32 %7, %8, %7\n")
KERNEL(k_and_sgpr, "v_and_b32_e32 %0, %10, %0\n v_and_b32_e32 %1, %10, %1\n v_and_b32_e32 %2, %10, %2\n v_and_b32_e32 %3, %10, %3\n v_and_b32_e32 %4, %10, %4\n v_and_b32_e32 %5, %10, %5\n v_and_b32_e32 %6, %10, %6\n v_and_b32_e32 %7, %10, %7\n")
KERNEL(k_mov_e32, "v_mov_b32_e32 %0, %8\n v_mov_b32_e32 %1, %8\n v_mov_b32_e32 %2, %8\n v_mov_b32_e32 %3, %8\n v_mov_b32_e32 %4, %8\n v_mov_b32_e32 %5, %8\n v_mov_b32_e32 %6, %8\n v_mov_b32_e32 %7, %8\n")
KERNEL(k_addu_e32, "v_add_u32_e32 %0, %8, %0\n v_add_u32_e32 %1, %8, %1\n v_add_u32_e32 %2, %8, %2\n v_add_u32_e32 %3, %8, %3\n v_add_u32_e32 %4, %8, %4\n v_add_u32_e32 %5, %8, %5\n v_add_u32_e32 %6, %8, %6\n v_add_u32_e32 %7, %8, %7\n")
KERNEL(k_cmp_cnd_vcc, "v_cmp_gt_f32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %8, %9, vcc\n v_cmp_gt_f32_e32 vcc, %2, %8\n v_cndmask_b32_e32 %3, %8, %9, vcc\n v_cmp_gt_f32_e32 vcc, %4, %8\n v_cndmask_b32_e32 %5, %8, %9, vcc\n v_cmp_gt_f32_e32 vcc, %6, %8\n v_cndmask_b32_e32 %7, %8, %9, vcc\n")
KERNEL(k_cmp_only, "v_cmp_gt_f32_e32 vcc, %0, %8\n v_cmp_gt_f32_e32 vcc, %1, %8\n v_cmp_gt_f32_e32 vcc, %2, %8\n v_cmp_gt_f32_e32 vcc, %3, %8\n v_cmp_gt_f32_e32 vcc, %4, %8\n v_cmp_gt_f32_e32 vcc, %5, %8\n v_cmp_gt_f32_e32 vcc, %6, %8\n v_cmp_gt_f32_e32 vcc, %7, %8\n")
KERNEL(k_cnd_only, "v_cndmask_b32_e32 %0, %8, %9, vcc\n v_cndmask_b32_e32 %1, %8, %9, vcc\n v_cndmask_b32_e32 %2, %8, %9, vcc\n v_cndmask_b32_e32 %3, %8, %9, vcc\n v_cndmask_b32_e32 %4, %8, %9, vcc\n v_cndmask_b32_e32 %5, %8, %9, vcc\n v_cndmask_b32_e32 %6, %8, %9, vcc\n v_cndmask_b32_e32 %7, %8, %9, vcc\n")
KERNEL(k_bfi, "v_bfi_b32 %0, %8, %9, %0\n v_bfi_b32 %1, %8, %9, %1\n v_bfi_b32 %2, %8, %9, %2\n v_bfi_b32 %3, %8, %9, %3\n v_bfi_b32 %4, %8, %9, %4\n v_bfi_b32 %5, %8, %9, %5\n v_bfi_b32 %6, %8, %9, %6\n v_bfi_b32 %7, %8, %9, %7\n")
KERNEL(k_min3, "v_min3_f32 %0, %8, %9, %0\n v_min3_f32 %1, %8, %9, %1\n v_min3_f32 %2, %8, %9, %2\n v_min3_f32 %3, %8, %9, %3\n v_min3_f32 %4, %8, %9, %4\n v_min3_f32 %5, %8, %9, %5\n v_min3_f32 %6, %8, %9, %6\n v_min3_f32 %7, %8, %9, %7\n")
KERNEL(k_add_dpp, "v_add_f32_dpp %0, %4, %0 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %5, %1 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %6, %2 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %7, %3 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %0, %4 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %1, %5 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %2, %6 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %3, %7 row_ror:3 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_add_sdwa, "v_add_f32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n v_add_f32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n")

template <typename K>
void run(const char *name, K kern, float *d, double per_instr_results = 1.0)
{
    const int iters = 200;
    for (int wps : {1, 2, 8}) {
        int blocks = 256 * wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double instr_per_simd = (double)iters * 16 * 8 * wps;
        printf("%-22s waves/SIMD %d  %.4f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.3GHz, launch overhead included)\n", name, wps,
               best, best * 1e6 / instr_per_simd, best * 1e6 / instr_per_simd * 2.3);
    }
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 256 * 2048 * sizeof(float) + 4096);
    (void)hipMemset(d, 0, 256 * 2048 * sizeof(float));
    run("v_add_f32_e32", k_add_e32, d);
    run("v_add_f32_e64", k_add_e64, d);
    run("v_mul_f32_e32", k_mul_e32, d);
    run("v_fmac_f32_e32", k_fmac_e32, d);
    run("v_fma_f32 (VOP3)", k_fma_vop3, d);
    run("v_min_f32_e32", k_min_e32, d);
    run("v_min3_f32 (VOP3)", k_min3, d);
    run("v_xor_b32_e32", k_xor_e32, d);
    run("v_and_b32_e32 sgpr", k_and_sgpr, d);
    run("v_mov_b32_e32", k_mov_e32, d);
    run("v_add_u32_e32", k_addu_e32, d);
    run("cmp_e32+cndmask_e32", k_cmp_cnd_vcc, d);
    run("v_cmp_gt_f32_e32", k_cmp_only, d);
    run("v_cndmask_b32_e32", k_cnd_only, d);
    run("v_bfi_b32 (VOP3)", k_bfi, d);
    run("v_add_f32_dpp", k_add_dpp, d);
    run("v_add_f32_sdwa", k_add_sdwa, d);
    return 0;
}
