// Per-frame set-up and metric evaluation shared by the OSD search kernels (conventional, FS, PB):
// primed-order values and P' rows in LDS, byte LUTs of partial |y'| sums, the canonical metric order.
//   FS_OSD/convention_osd.py:49-76 (convention_osd_main), PB_OSD/pb_testing.py:100-149, FS_OSD/fs_testing.py:51-64
#pragma once

#include "ldpc_wave.h"

namespace ldpc {

struct __attribute__((aligned(16))) SearchLds {
    float lut[8][256];   // lut[b][v] = sum of |y'[64+8b+t]| over the set bits t of v, ascending t
    u64 P[64];           // rows of P'
    float w[128];        // |y'|
    u64 cw[2];           // codeword being assembled in original bit order
    unsigned char perm[128];
};
struct __attribute__((aligned(16))) SearchLdsLean {   // the same without the byte LUTs (1.2 KiB instead of 9.4)
    u64 P[64];
    float w[128];
    u64 cw[2];
    unsigned char perm[128];
};

// The metric of tep_cost() without the LUTs: the canonical order written out (each parity byte's set positions ascending from
// 0.0f, the byte sums added in order) -- bit-identical, ~130 instructions instead of 8 LUT reads: for kernels that evaluate
// one or two candidates per frame.
__device__ __forceinline__ float tep_cost_direct(const float *w, float mrb, u64 D)
{
    float acc = mrb;
#pragma unroll 1
    for (int b = 0; b < 8; ++b) {
        const unsigned v = (unsigned)(D >> (8 * b)) & 255u;
        float bs = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; ++t) bs = ((v >> t) & 1u) ? bs + w[64 + 8 * b + t] : bs;
        acc = acc + bs;
    }
    return acc;
}

template <int B>
__device__ __forceinline__ float lut_term(const SearchLds &L, u64 D) { return lut_byte<B>(L.lut, D); }

__device__ __forceinline__ float tep_cost(const SearchLds &L, float mrb, u64 D)
{
    float acc = mrb;
    acc = acc + lut_term<0>(L, D); acc = acc + lut_term<1>(L, D); acc = acc + lut_term<2>(L, D); acc = acc + lut_term<3>(L, D);
    acc = acc + lut_term<4>(L, D); acc = acc + lut_term<5>(L, D); acc = acc + lut_term<6>(L, D); acc = acc + lut_term<7>(L, D);
    return acc;
}

// The same sum with the eight LUT reads issued together (their LDS round trips overlap; the compiler, left to the form
// above under register pressure, issues read - wait - add eight times over).  The empty asm pins the reads before the adds.
__device__ __forceinline__ float tep_cost_wide(const SearchLds &L, float mrb, u64 D)
{
    float t0 = lut_term<0>(L, D), t1 = lut_term<1>(L, D), t2 = lut_term<2>(L, D), t3 = lut_term<3>(L, D);
    float t4 = lut_term<4>(L, D), t5 = lut_term<5>(L, D), t6 = lut_term<6>(L, D), t7 = lut_term<7>(L, D);
    asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7));
    float acc = mrb + t0;
    acc = acc + t1; acc = acc + t2; acc = acc + t3; acc = acc + t4; acc = acc + t5; acc = acc + t6; acc = acc + t7;
    return acc;
}

// The same sum with an exact early exit: every term is >= 0, so once the prefix (MRB weights + the two
// most reliable parity bytes) exceeds an upper bound of the final minimum the candidate can neither win
// nor tie, and its six remaining LUT reads are skipped (the scan is LDS-bound: random LUT reads, 63 % of
// the LDS cycles were bank conflicts).  At 2.5 dB ~93 % of the order-2 TEPs leave after two bytes.
// Returns false for a pruned candidate; otherwise `cost` is bit-identical to tep_cost().
__device__ __forceinline__ bool tep_cost_bounded(const SearchLds &L, float mrb, u64 D, float bound, float &cost)
{
    float acc = mrb + lut_term<0>(L, D);
    acc = acc + lut_term<1>(L, D);
    if (acc > bound) return false;
    acc = acc + lut_term<2>(L, D); acc = acc + lut_term<3>(L, D); acc = acc + lut_term<4>(L, D);
    acc = acc + lut_term<5>(L, D); acc = acc + lut_term<6>(L, D); acc = acc + lut_term<7>(L, D);
    cost = acc;
    return true;
}

// tep_cost_direct for a candidate that is the SAME in every lane (the order-0 candidate): lane b sums byte b, the byte sums are
// added in order from lanes 0..7 -- the same operations in the same order, a quarter of the instructions.
__device__ __forceinline__ float tep_cost_direct_uniform(const float *w, float mrb, u64 D, int lane)
{
    const int b = lane & 7;
    const unsigned v = (unsigned)(D >> (8 * b)) & 255u;
    float bs = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) bs = ((v >> t) & 1u) ? bs + w[64 + 8 * b + t] : bs;
    float acc = mrb;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bs), k));
    return acc;
}

// per-frame set-up shared by every search: primed-order values into LDS, hard decisions, byte
// LUTs, and the parity discrepancy d0 of the order-0 candidate
struct SearchFrame {
    u64 hm, hp, d0;   // hard decisions of the MRB / parity part (y' > 0 ? 0 : 1), order-0 discrepancy
    int o1, o2;       // original bit index of primed positions lane and 64 + lane
};

// (LUTS = false: the caller builds the byte LUTs itself, e.g. spread over the wavefronts of a workgroup)
// (LDS: SearchLds, or a struct with the same P / w / cw members and no LUTs -- SearchLdsLean -- for a kernel that evaluates a
//  handful of candidates per frame and is better served by the occupancy the 8 KiB buy)
// (y1 / y2: the channel values of primed positions lane / 64 + lane, y'[p] = y[perm[p]])
template <bool LUTS = true, class LDS = SearchLds>
__device__ __forceinline__ SearchFrame search_prepare_vals(LDS &L, float y1, float y2, int o1, int o2, u64 Prow, int lane)
{
    SearchFrame S;
    S.o1 = o1;
    S.o2 = o2;
    L.perm[lane] = (unsigned char)S.o1;
    L.perm[lane + 64] = (unsigned char)S.o2;
    L.w[lane] = __builtin_fabsf(y1);
    L.w[lane + 64] = __builtin_fabsf(y2);
    L.P[lane] = Prow;
    if (lane < 2) L.cw[lane] = 0;
    S.hm = __ballot(!(y1 > 0.0f));
    S.hp = __ballot(!(y2 > 0.0f));
    wave_fence();
    if constexpr (LUTS) build_byte_luts<8>(L.lut, &L.w[64], lane);
    // d0 = (u0 . P') ^ h_parity : XOR-reduce the rows selected by the MRB hard decisions
    S.d0 = wave_xor64(((S.hm >> lane) & 1) ? Prow : 0ull) ^ S.hp;
    wave_fence();
    return S;
}

template <bool LUTS = true, class LDS = SearchLds>
__device__ __forceinline__ SearchFrame search_prepare_regs(LDS &L, const float *__restrict__ y, long long src,
                                                           int o1, int o2, u64 Prow, int lane)
{
    return search_prepare_vals<LUTS>(L, y[src * 128 + o1], y[src * 128 + o2], o1, o2, Prow, lane);
}

template <bool LUTS = true, class LDS = SearchLds>
__device__ __forceinline__ SearchFrame search_prepare(LDS &L, const float *__restrict__ y, long long src,
                                                      const unsigned char *__restrict__ perm_in,
                                                      const u64 *__restrict__ parity_in, long long f, int lane)
{
    return search_prepare_regs<LUTS>(L, y, src, perm_in[f * 128 + lane], perm_in[f * 128 + 64 + lane], parity_in[f * 64 + lane], lane);
}

// candidate (E = flipped MRB positions, D = parity discrepancy) -> codeword in ORIGINAL bit order
template <class LDS>
__device__ __forceinline__ void search_finish(LDS &L, const SearchFrame &S, u64 E, u64 D, long long f, int lane,
                                              u64 *__restrict__ cw_out)
{
    const u64 mrb_bits = S.hm ^ E, par_bits = D ^ S.hp;
    if ((mrb_bits >> lane) & 1) atomicOr(&L.cw[S.o1 >> 6], 1ull << (S.o1 & 63));
    if ((par_bits >> lane) & 1) atomicOr(&L.cw[S.o2 >> 6], 1ull << (S.o2 & 63));
    wave_fence();
    if (lane < 2) cw_out[f * 2 + lane] = L.cw[lane];
    wave_fence();
}

// one TEP (ascending support s.x < s.y < s.z, weight s.w) -> parity discrepancy, flip mask, MRB weight sum
__device__ __forceinline__ void tep_apply(const SearchLds &L, uchar4 s, u64 d0, u64 &D, u64 &E, float &mrb)
{
    D = d0; E = 0; mrb = 0.0f;
    if (s.w > 0) { D ^= L.P[s.x]; E |= 1ull << s.x; mrb = L.w[s.x]; }
    if (s.w > 1) { D ^= L.P[s.y]; E |= 1ull << s.y; mrb = mrb + L.w[s.y]; }
    if (s.w > 2) { D ^= L.P[s.z]; E |= 1ull << s.z; mrb = mrb + L.w[s.z]; }
}

// wave arg-min on (cost, index): every lane returns the winner
__device__ __forceinline__ void wave_argmin(float &best, int &bestt, u64 &bestD, u64 &bestE, int lane)
{
    const int w = wave_argmin_lane(best, bestt);
    best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best), w));
    bestt = __builtin_amdgcn_readlane(bestt, w);
    bestD = readlane64(bestD, w);
    bestE = readlane64(bestE, w);
}

}  // namespace ldpc
