// Micro-benchmark: issue cost of the VALU instructions the NMS kernel is made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_valu scripts/ubench_valu.hip && /tmp/ubench_valu
// Prints cycles per wave-instruction per SIMD at 1/2/4/8 waves per SIMD (shader clock from
// s_memtime vs s_memrealtime).  Used to price the kernel in DESIGN.md; not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP16(x) x x x x x x x x x x x x x x x x

#ifndef NUM_VGPR
#define NUM_VGPR 24
#endif
template <int OP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(NUM_VGPR))) void k(float *out, int iters, unsigned long long *clk)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = out[threadIdx.x & 63];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0) {  // v_add_f32 VOP2
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 1) {  // v_med3_f32 VOP3
            REP16(asm volatile("v_med3_f32 %0, %0, %8, %1\n v_med3_f32 %1, %1, %8, %2\n v_med3_f32 %2, %2, %8, %3\n v_med3_f32 %3, %3, %8, %4\n"
                               "v_med3_f32 %4, %4, %8, %5\n v_med3_f32 %5, %5, %8, %6\n v_med3_f32 %6, %6, %8, %7\n v_med3_f32 %7, %7, %8, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 2) {  // v_add_f32 DPP row_ror (src from another accumulator 4 back)
            REP16(asm volatile("v_add_f32_dpp %0, %4, %0 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %5, %1 row_ror:3 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %2, %6, %2 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %7, %3 row_ror:3 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %4, %0, %4 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %1, %5 row_ror:3 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %6, %2, %6 row_ror:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %3, %7 row_ror:3 row_mask:0xf bank_mask:0xf\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 3) {  // v_cmp_gt_f32 e64 -> SGPR pair, v_cndmask e64 reading it
            REP16(asm volatile("v_cmp_gt_f32 s[20:21], |%0|, %8\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cmp_gt_f32 s[22:23], |%2|, %8\n v_cndmask_b32 %3, %3, %8, s[22:23]\n"
                               "v_cmp_gt_f32 s[24:25], |%4|, %8\n v_cndmask_b32 %5, %5, %8, s[24:25]\n v_cmp_gt_f32 s[26:27], |%6|, %8\n v_cndmask_b32 %7, %7, %8, s[26:27]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
                               : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if constexpr (OP == 4) {  // v_pk_add_f32 (2 results per instruction)
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(double *)&b));)
        } else if constexpr (OP == 5) {  // v_bfi_b32 + v_xor_b32
            REP16(asm volatile("v_xor_b32 %0, %0, %8\n v_bfi_b32 %1, %8, %1, %0\n v_xor_b32 %2, %2, %8\n v_bfi_b32 %3, %8, %3, %2\n"
                               "v_xor_b32 %4, %4, %8\n v_bfi_b32 %5, %8, %5, %4\n v_xor_b32 %6, %6, %8\n v_bfi_b32 %7, %8, %7, %6\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 6) {  // v_min_f32 e64 with abs modifiers
            REP16(asm volatile("v_min_f32 %0, |%0|, |%8|\n v_min_f32 %1, |%1|, |%8|\n v_min_f32 %2, |%2|, |%8|\n v_min_f32 %3, |%3|, |%8|\n"
                               "v_min_f32 %4, |%4|, |%8|\n v_min_f32 %5, |%5|, |%8|\n v_min_f32 %6, |%6|, |%8|\n v_min_f32 %7, |%7|, |%8|\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 7) {  // dependent chain v_add_f32 (latency)
            REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                               "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                               : "+v"(a0) : "v"(b));)
        } else if constexpr (OP == 8) {  // v_mov_b32 DPP (unfused rotation)
            REP16(asm volatile("v_mov_b32_dpp %0, %4 row_ror:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_ror:3 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %6 row_ror:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 row_ror:3 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %4, %0 row_ror:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %1 row_ror:3 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %6, %2 row_ror:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %3 row_ror:3 row_mask:0xf bank_mask:0xf\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 10) {  // NMS-like mix: sub_dpp, med3, min|.|, xor, cmp+cndmask, xor, bfi, add_dpp
            REP16(asm volatile("v_sub_f32_dpp %0, %4, %0 row_ror:3 row_mask:0xf bank_mask:0xf\n v_med3_f32 %1, %1, %8, |%0|\n v_min_f32 %2, %2, |%0|\n v_xor_b32 %3, %3, %0\n"
                               "v_cmp_gt_f32 s[20:21], |%0|, %2\n v_cndmask_b32 %5, %1, %2, s[20:21]\n v_xor_b32 %6, %3, %0\n v_bfi_b32 %7, %8, %5, %6\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21");)
        } else if constexpr (OP == 9) {  // ds_bpermute_b32 (LDS crossbar shuffle)
            REP16(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                               "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
void run(const char *name, float *d, unsigned long long *dclk)
{
    const int iters = 200;
    for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
        int blocks = 256 * wps;  // 4 waves per block -> wps waves per SIMD on 256 CUs
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, dclk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, dclk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2]; hipMemcpy(c, dclk, sizeof(c), hipMemcpyDeviceToHost);
        double ghz = (double)c[0] / ((double)c[1] / 100e6) / 1e9;  // memrealtime ticks at 100 MHz
        double instr = (double)iters * 16 * 8;                     // per wave
        double cyc_per_instr_wave = (double)c[0] / instr;          // as seen by one wave
        double cyc_per_instr_simd = cyc_per_instr_wave / wps;      // SIMD throughput view
        printf("%-28s waves/SIMD %d  kernel %.3f ms  clk %.2f GHz  cycles/instr: per-wave %.2f  per-SIMD %.2f\n", name, wps, ms, ghz,
               cyc_per_instr_wave, cyc_per_instr_simd);
    }
}

int main()
{
    float *d; unsigned long long *dclk;
    hipMalloc(&d, 256 * 2048 * sizeof(float) + 4096);
    hipMemset(d, 0, 256 * 2048 * sizeof(float));
    hipMalloc(&dclk, 16);
    run<0>("v_add_f32", d, dclk);
    run<1>("v_med3_f32", d, dclk);
    run<2>("v_add_f32_dpp row_ror", d, dclk);
    run<3>("v_cmp_gt_e64+v_cndmask_e64", d, dclk);
    run<4>("v_pk_add_f32", d, dclk);
    run<5>("v_xor_b32+v_bfi_b32", d, dclk);
    run<6>("v_min_f32 |a|,|b| (e64)", d, dclk);
    run<7>("v_add_f32 dependent chain", d, dclk);
    run<8>("v_mov_b32_dpp row_ror", d, dclk);
    run<9>("ds_bpermute_b32", d, dclk);
    run<10>("NMS-like mix (dependent)", d, dclk);
    return 0;
}
