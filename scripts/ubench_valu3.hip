// Issue cost of the cross-lane / 64-bit / scalar-operand instructions the GF(2) elimination step uses.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/u3 scripts/ubench_valu3.hip && /tmp/u3
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x

#define KERNEL(NAME, BODY)                                                                         \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters)                          \
    {                                                                                              \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        unsigned long long w0 = a0, w1 = a1, w2 = a2, w3 = a3;                                     \
        unsigned b = out[threadIdx.x & 63] & 31;                                                   \
        unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0;                                                   \
        int sl = iters & 63;                                                                       \
        for (int i = 0; i < iters; ++i) {                                                          \
            REP16(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                               "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)        \
                               : "v"(b), "s"(sl) : "vcc", "m0", "scc");)                           \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(w0 + w1 + w2 + w3) + s0 + s1 + s2 + s3; \
    }
// operands: %0-%7 a0..a7 (v32), %8-%11 w0..w3 (v64), %12-%15 s0..s3, %16 b (v), %17 sl (s)
KERNEL(k_readlane, "v_readlane_b32 %12, %0, %17\n v_readlane_b32 %13, %1, %17\n v_readlane_b32 %14, %2, %17\n v_readlane_b32 %15, %3, %17\n v_readlane_b32 %12, %4, %17\n v_readlane_b32 %13, %5, %17\n v_readlane_b32 %14, %6, %17\n v_readlane_b32 %15, %7, %17\n")
KERNEL(k_writelane, "s_mov_b32 m0, %17\n v_writelane_b32 %0, %12, m0\n v_writelane_b32 %1, %12, m0\n v_writelane_b32 %2, %12, m0\n v_writelane_b32 %3, %12, m0\n v_writelane_b32 %4, %12, m0\n v_writelane_b32 %5, %12, m0\n v_writelane_b32 %6, %12, m0\n v_writelane_b32 %7, %12, m0\n")
KERNEL(k_lshl64, "v_lshlrev_b64 %8, %16, %8\n v_lshlrev_b64 %9, %16, %9\n v_lshlrev_b64 %10, %16, %10\n v_lshlrev_b64 %11, %16, %11\n v_lshlrev_b64 %8, %16, %8\n v_lshlrev_b64 %9, %16, %9\n v_lshlrev_b64 %10, %16, %10\n v_lshlrev_b64 %11, %16, %11\n")
KERNEL(k_lshl32, "v_lshlrev_b32 %0, %16, %0\n v_lshlrev_b32 %1, %16, %1\n v_lshlrev_b32 %2, %16, %2\n v_lshlrev_b32 %3, %16, %3\n v_lshlrev_b32 %4, %16, %4\n v_lshlrev_b32 %5, %16, %5\n v_lshlrev_b32 %6, %16, %6\n v_lshlrev_b32 %7, %16, %7\n")
KERNEL(k_lshl32_sgpr, "v_lshlrev_b32 %0, %16, %12\n v_lshlrev_b32 %1, %16, %12\n v_lshlrev_b32 %2, %16, %12\n v_lshlrev_b32 %3, %16, %12\n v_lshlrev_b32 %4, %16, %12\n v_lshlrev_b32 %5, %16, %12\n v_lshlrev_b32 %6, %16, %12\n v_lshlrev_b32 %7, %16, %12\n")
KERNEL(k_bfe_i32_sgpr, "v_bfe_i32 %0, %0, %17, 1\n v_bfe_i32 %1, %1, %17, 1\n v_bfe_i32 %2, %2, %17, 1\n v_bfe_i32 %3, %3, %17, 1\n v_bfe_i32 %4, %4, %17, 1\n v_bfe_i32 %5, %5, %17, 1\n v_bfe_i32 %6, %6, %17, 1\n v_bfe_i32 %7, %7, %17, 1\n")
KERNEL(k_bfe_u32_v, "v_bfe_u32 %0, %0, %16, 8\n v_bfe_u32 %1, %1, %16, 8\n v_bfe_u32 %2, %2, %16, 8\n v_bfe_u32 %3, %3, %16, 8\n v_bfe_u32 %4, %4, %16, 8\n v_bfe_u32 %5, %5, %16, 8\n v_bfe_u32 %6, %6, %16, 8\n v_bfe_u32 %7, %7, %16, 8\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %16\n v_lshl_add_u32 %1, %1, 2, %16\n v_lshl_add_u32 %2, %2, 2, %16\n v_lshl_add_u32 %3, %3, 2, %16\n v_lshl_add_u32 %4, %4, 2, %16\n v_lshl_add_u32 %5, %5, 2, %16\n v_lshl_add_u32 %6, %6, 2, %16\n v_lshl_add_u32 %7, %7, 2, %16\n")
KERNEL(k_bitop3_sgpr, "v_bitop3_b32 %0, %0, %16, %12 bitop3:0x78\n v_bitop3_b32 %1, %1, %16, %12 bitop3:0x78\n v_bitop3_b32 %2, %2, %16, %12 bitop3:0x78\n v_bitop3_b32 %3, %3, %16, %12 bitop3:0x78\n v_bitop3_b32 %4, %4, %16, %12 bitop3:0x78\n v_bitop3_b32 %5, %5, %16, %12 bitop3:0x78\n v_bitop3_b32 %6, %6, %16, %12 bitop3:0x78\n v_bitop3_b32 %7, %7, %16, %12 bitop3:0x78\n")
KERNEL(k_bitop3_v, "v_bitop3_b32 %0, %0, %16, %1 bitop3:0x78\n v_bitop3_b32 %1, %1, %16, %2 bitop3:0x78\n v_bitop3_b32 %2, %2, %16, %3 bitop3:0x78\n v_bitop3_b32 %3, %3, %16, %4 bitop3:0x78\n v_bitop3_b32 %4, %4, %16, %5 bitop3:0x78\n v_bitop3_b32 %5, %5, %16, %6 bitop3:0x78\n v_bitop3_b32 %6, %6, %16, %7 bitop3:0x78\n v_bitop3_b32 %7, %7, %16, %0 bitop3:0x78\n")
KERNEL(k_cmp_i32_e64, "v_cmp_gt_i32_e64 vcc, 0, %0\n v_cmp_gt_i32_e64 vcc, 0, %1\n v_cmp_gt_i32_e64 vcc, 0, %2\n v_cmp_gt_i32_e64 vcc, 0, %3\n v_cmp_gt_i32_e64 vcc, 0, %4\n v_cmp_gt_i32_e64 vcc, 0, %5\n v_cmp_gt_i32_e64 vcc, 0, %6\n v_cmp_gt_i32_e64 vcc, 0, %7\n")
KERNEL(k_min_u32, "v_min_u32_e32 %0, %16, %0\n v_min_u32_e32 %1, %16, %1\n v_min_u32_e32 %2, %16, %2\n v_min_u32_e32 %3, %16, %3\n v_min_u32_e32 %4, %16, %4\n v_min_u32_e32 %5, %16, %5\n v_min_u32_e32 %6, %16, %6\n v_min_u32_e32 %7, %16, %7\n")
KERNEL(k_min3_u32, "v_min3_u32 %0, %0, %16, %1\n v_min3_u32 %1, %1, %16, %2\n v_min3_u32 %2, %2, %16, %3\n v_min3_u32 %3, %3, %16, %4\n v_min3_u32 %4, %4, %16, %5\n v_min3_u32 %5, %5, %16, %6\n v_min3_u32 %6, %6, %16, %7\n v_min3_u32 %7, %7, %16, %0\n")
KERNEL(k_min3_i32, "v_min3_i32 %0, %0, %16, %1\n v_min3_i32 %1, %1, %16, %2\n v_min3_i32 %2, %2, %16, %3\n v_min3_i32 %3, %3, %16, %4\n v_min3_i32 %4, %4, %16, %5\n v_min3_i32 %5, %5, %16, %6\n v_min3_i32 %6, %6, %16, %7\n v_min3_i32 %7, %7, %16, %0\n")
KERNEL(k_min_i16pk, "v_pk_min_u16 %0, %0, %16\n v_pk_min_u16 %1, %1, %16\n v_pk_min_u16 %2, %2, %16\n v_pk_min_u16 %3, %3, %16\n v_pk_min_u16 %4, %4, %16\n v_pk_min_u16 %5, %5, %16\n v_pk_min_u16 %6, %6, %16\n v_pk_min_u16 %7, %7, %16\n")
KERNEL(k_sub_co_addc, "v_sub_co_u32_e32 %0, vcc, %16, %0\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_sub_co_u32_e32 %2, vcc, %16, %2\n v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n v_sub_co_u32_e32 %4, vcc, %16, %4\n v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n v_sub_co_u32_e32 %6, vcc, %16, %6\n v_addc_co_u32_e32 %7, vcc, 0, %7, vcc\n")
KERNEL(k_cmp_addc, "v_cmp_gt_i32_e32 vcc, %16, %0\n v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n v_cmp_gt_i32_e32 vcc, %16, %2\n v_addc_co_u32_e32 %3, vcc, 0, %3, vcc\n v_cmp_gt_i32_e32 vcc, %16, %4\n v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n v_cmp_gt_i32_e32 vcc, %16, %6\n v_addc_co_u32_e32 %7, vcc, 0, %7, vcc\n")
KERNEL(k_salu, "s_add_u32 %12, %12, %17\n s_add_u32 %13, %13, %17\n s_add_u32 %14, %14, %17\n s_add_u32 %15, %15, %17\n s_lshl_b32 %12, %12, 1\n s_lshl_b32 %13, %13, 1\n s_lshl_b32 %14, %14, 1\n s_lshl_b32 %15, %15, 1\n")
KERNEL(k_salu_dep, "s_add_u32 %12, %12, %17\n s_lshl_b32 %12, %12, 1\n s_add_u32 %12, %12, %17\n s_lshl_b32 %12, %12, 1\n s_add_u32 %12, %12, %17\n s_lshl_b32 %12, %12, 1\n s_add_u32 %12, %12, %17\n s_lshl_b32 %12, %12, 1\n")
KERNEL(k_readlane_dep, "v_readlane_b32 %12, %0, %17\n s_and_b32 %12, %12, 63\n v_readlane_b32 %13, %1, %12\n s_and_b32 %13, %13, 63\n v_readlane_b32 %12, %2, %13\n s_and_b32 %12, %12, 63\n v_readlane_b32 %13, %3, %12\n s_and_b32 %13, %13, 63\n")
KERNEL(k_branch, "s_cmp_eq_u32 %17, 77\n s_cbranch_scc1 1f\n s_add_u32 %12, %12, 1\n1:\n s_cmp_eq_u32 %17, 78\n s_cbranch_scc1 2f\n s_add_u32 %13, %13, 1\n2:\n s_cmp_lg_u32 %17, 79\n s_cbranch_scc1 3f\n s_add_u32 %14, %14, 1\n3:\n s_cmp_lg_u32 %17, 80\n s_cbranch_scc1 4f\n s_add_u32 %15, %15, 1\n4:\n")

template <typename K>
void run(const char *name, K kern, unsigned *d, int per_body)
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
        double instr_per_simd = (double)iters * 16 * per_body * wps;
        printf("%-26s waves/SIMD %d  %.4f ms  -> %.2f cycles @2.3GHz per wave-instr per SIMD (launch overhead included)\n", name, wps,
               best, best * 1e6 / instr_per_simd * 2.3);
    }
}

int main()
{
    unsigned *d;
    (void)hipMalloc(&d, 256 * 2048 * sizeof(unsigned) + 4096);
    (void)hipMemset(d, 0, 256 * 2048 * sizeof(unsigned));
    run("v_readlane_b32", k_readlane, d, 8);
    run("v_writelane_b32 (m0)", k_writelane, d, 8);
    run("v_lshlrev_b64", k_lshl64, d, 8);
    run("v_lshlrev_b32", k_lshl32, d, 8);
    run("v_lshlrev_b32 sgpr src", k_lshl32_sgpr, d, 8);
    run("v_bfe_i32 sgpr offset", k_bfe_i32_sgpr, d, 8);
    run("v_bfe_u32 vgpr", k_bfe_u32_v, d, 8);
    run("v_lshl_add_u32", k_lshl_add, d, 8);
    run("v_bitop3_b32 sgpr", k_bitop3_sgpr, d, 8);
    run("v_bitop3_b32 vgpr", k_bitop3_v, d, 8);
    run("v_cmp_gt_i32_e64", k_cmp_i32_e64, d, 8);
    run("v_min_u32_e32", k_min_u32, d, 8);
    run("v_min3_u32", k_min3_u32, d, 8);
    run("v_min3_i32", k_min3_i32, d, 8);
    run("v_pk_min_u16", k_min_i16pk, d, 8);
    run("v_sub_co + v_addc", k_sub_co_addc, d, 8);
    run("v_cmp_gt_i32 + v_addc", k_cmp_addc, d, 8);
    run("salu independent", k_salu, d, 8);
    run("salu dependent chain", k_salu_dep, d, 8);
    run("readlane->salu->readlane", k_readlane_dep, d, 8);
    run("s_cmp + not-taken/taken br", k_branch, d, 12);
    return 0;
}
