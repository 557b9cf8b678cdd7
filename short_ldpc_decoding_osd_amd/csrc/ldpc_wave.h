// Wave-level device helpers shared by the OSD translation units (gfx950, wave64): lane reads,
// DPP reductions, 64x64 bit transpose, the register-resident GF(2) elimination, rank-sort pieces.
#pragma once

#include "ldpc_internal.h"

namespace ldpc {

typedef unsigned long long u64;

constexpr int kOsdN = 128, kOsdK = 64;

// ---------------------------------------------------------------------------------------
// wave-level helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ u64 readlane64(u64 v, int lane)
{
    unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, lane);
    unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 shfl64(u64 v, int src)
{
    unsigned lo = __shfl((int)(unsigned)v, src, 64);
    unsigned hi = __shfl((int)(unsigned)(v >> 32), src, 64);
    return ((u64)hi << 32) | lo;
}

// ---- wave reductions on DPP (row/bank permutes inside the VALU, ~4 cycles per step) instead of
// ds_bpermute shuffles (~30 cycles per wave-instruction on gfx950, profiles/r01/ubench_valu_issue.txt).
// Pattern: xor-1, xor-2 quad permutes, row_half_mirror, row_mirror -> every lane holds its row's result;
// row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3 -> lane 63 holds the wave's result.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i(int old, int x)
{
    return __builtin_amdgcn_update_dpp(old, x, CTRL, ROWMASK, 0xF, false);
}
#define LDPC_WAVE_REDUCE(v, OP)                                   \
    v = OP(v, dpp_i<0xB1, 0xF>(v, v));  /* quad_perm [1,0,3,2] */ \
    v = OP(v, dpp_i<0x4E, 0xF>(v, v));  /* quad_perm [2,3,0,1] */ \
    v = OP(v, dpp_i<0x141, 0xF>(v, v)); /* row_half_mirror */     \
    v = OP(v, dpp_i<0x140, 0xF>(v, v)); /* row_mirror */          \
    v = OP(v, dpp_i<0x142, 0xA>(ID, v)); /* row_bcast:15 */       \
    v = OP(v, dpp_i<0x143, 0xC>(ID, v)); /* row_bcast:31 */

__device__ __forceinline__ int op_xor(int a, int b) { return a ^ b; }
__device__ __forceinline__ int op_min_i(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int op_min_f(int a, int b) { return __float_as_int(__builtin_fminf(__int_as_float(a), __int_as_float(b))); }

// XOR of a 64-bit value over the wave, result in every lane
__device__ __forceinline__ u64 wave_xor64(u64 x)
{
    int lo = (int)(unsigned)x, hi = (int)(unsigned)(x >> 32);
    { const int ID = 0; LDPC_WAVE_REDUCE(lo, op_xor) LDPC_WAVE_REDUCE(hi, op_xor) }
    const unsigned rl = __builtin_amdgcn_readlane(lo, 63), rh = __builtin_amdgcn_readlane(hi, 63);
    return ((u64)rh << 32) | rl;
}
// minimum of non-negative-or-inf floats over the wave, result in every lane
__device__ __forceinline__ float wave_min_f32(float x)
{
    int v = __float_as_int(x);
    { const int ID = 0x7F800000; LDPC_WAVE_REDUCE(v, op_min_f) }
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
__device__ __forceinline__ int wave_min_i32(int x)
{
    int v = x;
    { const int ID = 0x7FFFFFFF; LDPC_WAVE_REDUCE(v, op_min_i) }
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int op_max_i(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int wave_max_i32(int x)
{
    int v = x;
    { const int ID = (int)0x80000000; LDPC_WAVE_REDUCE(v, op_max_i) }
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int op_max_f(int a, int b) { return __float_as_int(__builtin_fmaxf(__int_as_float(a), __int_as_float(b))); }
__device__ __forceinline__ int op_add_i(int a, int b) { return a + b; }
// maximum of floats >= -1 over the wave / sum of ints over the wave, result in every lane
__device__ __forceinline__ float wave_max_f32(float x)
{
    int v = __float_as_int(x);
    { const int ID = (int)0xBF800000; LDPC_WAVE_REDUCE(v, op_max_f) }
    return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
__device__ __forceinline__ int wave_add_i32(int x)
{
    int v = x;
    { const int ID = 0; LDPC_WAVE_REDUCE(v, op_add_i) }
    return __builtin_amdgcn_readlane(v, 63);
}
// inclusive prefix sum over the lanes (lane l gets x_0 + ... + x_l): row_shr DPP steps inside a row of 16, then the
// row totals by row_bcast:15 / row_bcast:31
__device__ __forceinline__ int wave_incl_add_dpp(int x)
{
    int v = x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
// the same for floats (the additions run in the ladder's order, not ascending: for bounds and statistics only)
__device__ __forceinline__ float wave_incl_addf_dpp(float x)
{
    float v = x;
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return v;
}
// inclusive prefix minimum over the lanes, same DPP ladder (a lane without a source keeps its own value)
__device__ __forceinline__ float wave_incl_min_dpp(float x)
{
    int v = __float_as_int(x);
    const auto step = [](int v, int got) { return __float_as_int(__builtin_fminf(__int_as_float(v), __int_as_float(got))); };
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xF, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));
    v = step(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false));
    return __int_as_float(v);
}
// lane that holds the smallest (value, index) pair; ties on value go to the lower index
__device__ __forceinline__ int wave_argmin_lane(float s, int idx)
{
    const float m = wave_min_f32(s);
    const int mi = wave_min_i32(s == m ? idx : 0x7FFFFFFF);
    return __builtin_ctzll(__ballot(s == m && idx == mi));
}

// 64x64 bit transpose across the wavefront: in: lane a holds bits b; out: lane b holds bits a
template <int S>
__device__ __forceinline__ u64 transpose_stage(u64 x, int lane)
{
    constexpr u64 m = S == 32 ? 0x00000000FFFFFFFFull : S == 16 ? 0x0000FFFF0000FFFFull : S == 8 ? 0x00FF00FF00FF00FFull
                    : S == 4 ? 0x0F0F0F0F0F0F0F0Full : S == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
    const u64 p = shfl64(x, lane ^ S);
    return (lane & S) ? (((p >> S) & m) | (x & ~m)) : ((x & m) | ((p & m) << S));
}

__device__ __forceinline__ u64 transpose64(u64 x, int lane)
{
    x = transpose_stage<32>(x, lane);
    x = transpose_stage<16>(x, lane);
    x = transpose_stage<8>(x, lane);
    x = transpose_stage<4>(x, lane);
    x = transpose_stage<2>(x, lane);
    x = transpose_stage<1>(x, lane);
    return x;
}

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------
// Gauss-Jordan over GF(2), column-major in registers (full_gf2elim, pb_testing.py:231-266).
//   C1/C2 : columns lane / lane+64, bit = PHYSICAL row
//   rho   : lane l = logical row l -> physical row (row exchanges only permute this map)
//   idx1/2: the reference's index_order entry travelling with each column (:276-281)
// For i = j = 0..63:  first logical row >= i with a 1 in column i becomes the pivot (:238-245);
// if there is none, column i is exchanged with the first column >= i in which logical row i has
// a 1 and the pair is recorded (:251-256); then column i is cleared in every other row (:258-262).
// Returns the number of column exchanges, or -1 for a rank-deficient matrix.
// ---------------------------------------------------------------------------------------
// (pieces pinned with inline asm: left to itself the compiler moves the pivot column into VGPRs and
//  spends ~30 VALU instructions per step, most of them in the 4-cycle class; this form needs ~12)
__device__ __forceinline__ u64 ge_clear_bit(u64 cj, int pr)
{
    u64 e;   // cj & ~(1 << pr) on the scalar unit
    asm("s_lshl_b64 %0, 1, %2\n\ts_andn2_b64 %0, %1, %0" : "=&s"(e) : "s"(cj), "s"(pr) : "scc");
    return e;
}

__device__ __forceinline__ u64 ge_sign_ballot(unsigned hi)
{
    u64 b;   // lanes whose word has bit 31 set (the compiler would widen this to a 64-bit compare)
    asm("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(b) : "v"(hi));
    return b;
}

__device__ __forceinline__ int ge_sign_mask(u64 c, int nshift)
{
    // -1 where bit (63 - nshift) of c is set: a 64-bit shift and an arithmetic shift, no scalar work
    // (selecting the 32-bit half with a wave-uniform branch costs ~7 scalar instructions, and the
    //  scalar unit issues only one instruction per 4 cycles per SIMD, like the vector unit)
    int m;
    const unsigned hi = (unsigned)((c << nshift) >> 32);
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(hi));
    return m;
}

// m1 / m2 = -1 in the lanes whose column (C1 / C2, as two 32-bit halves) has a 1 in row pr -- wave-uniform, 0..63.  The half that
// holds the row is picked by ONE scalar branch, then a signed one-bit field extract per column (v_bfe_i32 uses the low five bits
// of its offset, so pr serves for either half): 2 vector + 3 scalar instructions per step, where the shift form (ge_sign_mask:
// a 64-bit shift and an arithmetic shift per column) took 4 vector instructions.  Round 4: the front end's elimination loop
// 12 -> 10 vector instructions per step.  (Written as one asm block: left to the compiler the two-sided branch cost nine
// scalar instructions.)
__device__ __forceinline__ void ge_row_masks(unsigned c1l, unsigned c1h, unsigned c2l, unsigned c2h, int pr, int &m1, int &m2)
{
    asm("s_cmp_lt_u32 %6, 32\n\t"
        "s_cbranch_scc0 1f\n\t"
        "v_bfe_i32 %0, %2, %6, 1\n\t"
        "v_bfe_i32 %1, %4, %6, 1\n\t"
        "s_branch 2f\n"
        "1:\n\t"
        "v_bfe_i32 %0, %3, %6, 1\n\t"
        "v_bfe_i32 %1, %5, %6, 1\n"
        "2:"
        : "=&v"(m1), "=&v"(m2) : "v"(c1l), "v"(c1h), "v"(c2l), "v"(c2h), "s"(pr) : "scc");
}

__device__ __forceinline__ unsigned ge_xor_masked(unsigned c, int m, unsigned e)
{
    asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(c) : "v"(m), "s"(e));   // c ^ (m & e)
    return c;
}

__device__ __forceinline__ int ge_columns(u64 &C1, u64 &C2, int &rho, int &idx1, int &idx2, int lane,
                                          unsigned char *swaps /* LDS or global [64][2], may be null */)
{
    int nsw = 0;
    bool deficient = false;   // (no early exit from the loop: it would make the compiler guard every step)
    int nrho = 63 - rho;   // the loop carries 63 - rho: "bit rho of cj" is then the sign of cj << nrho
    u64 ge_i = ~0ull;      // lanes (logical rows) >= i
#pragma unroll
    for (int i = 0; i < kOsdK; ++i, ge_i <<= 1) {
        u64 cj = readlane64(C1, i);
        u64 bal = ge_sign_ballot((unsigned)((cj << nrho) >> 32)) & ge_i;
        if (bal == 0) {
            const int pri = 63 - __builtin_amdgcn_readlane(nrho, i);
            const u64 b1 = __ballot((C1 >> pri) & 1) & ge_i;
            unsigned col;
            if (b1) col = __builtin_ctzll(b1);
            else {
                const u64 b2 = __ballot((C2 >> pri) & 1);
                deficient |= b2 == 0;   // all-zero row: carry on with garbage, report at the end
                col = 64 + (b2 ? __builtin_ctzll(b2) : 0);
            }
            // (select form on purpose: branching on col < 64 makes the compiler spill C1/C2 to scratch)
            const unsigned cl = col & 63;
            const bool lo = col < 64;
            const u64 cc1 = readlane64(C1, cl), cc2 = readlane64(C2, cl);
            const int ic1 = __builtin_amdgcn_readlane(idx1, cl), ic2 = __builtin_amdgcn_readlane(idx2, cl);
            const u64 cc = lo ? cc1 : cc2;
            const int ic = lo ? ic1 : ic2;
            const int ii = __builtin_amdgcn_readlane(idx1, i);
            const bool hit = (unsigned)lane == cl;
            C1 = (hit && lo) ? cj : C1;
            idx1 = (hit && lo) ? ii : idx1;
            C2 = (hit && !lo) ? cj : C2;
            idx2 = (hit && !lo) ? ii : idx2;
            C1 = (lane == i) ? cc : C1;
            idx1 = (lane == i) ? ic : idx1;
            if (swaps && lane == 0) { swaps[2 * nsw] = (unsigned char)i; swaps[2 * nsw + 1] = (unsigned char)col; }
            ++nsw;
            cj = readlane64(C1, i);   // the exchanged-in column has its 1 in logical row i
            bal = ge_i & (0ull - ge_i);   // = 1 << i (keeps the loop counter 32 bits wide)
        }
        const int r = __builtin_ctzll(bal);
        const int npr = __builtin_amdgcn_readlane(nrho, r);
        if (r != i) {   // exchange logical rows i and r: two lanes of the row map, no data moves
            const int npi = __builtin_amdgcn_readlane(nrho, i);
            // (lane select through M0: a VALU instruction may read only one SGPR on gfx9)
            asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(nrho) : "s"(npr), "s"(i) : "m0");
            asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(nrho) : "s"(npi), "s"(r) : "m0");
        }
        const int pr = 63 - npr;
        const u64 e = ge_clear_bit(cj, pr);
        if (e != 0) {   // nothing to clear when the pivot column is already a unit vector (common: G = [P | I])
            const unsigned el = (unsigned)e, eh = (unsigned)(e >> 32);
            unsigned c1l = (unsigned)C1, c1h = (unsigned)(C1 >> 32), c2l = (unsigned)C2, c2h = (unsigned)(C2 >> 32);
            int m1, m2;      // -1 in the lanes whose column has a 1 in the pivot row
            ge_row_masks(c1l, c1h, c2l, c2h, pr, m1, m2);
            c1l = ge_xor_masked(c1l, m1, el); c1h = ge_xor_masked(c1h, m1, eh);
            c2l = ge_xor_masked(c2l, m2, el); c2h = ge_xor_masked(c2h, m2, eh);
            C1 = ((u64)c1h << 32) | c1l;
            C2 = ((u64)c2h << 32) | c2l;
        }
    }
    rho = 63 - nrho;
    return deficient ? -1 : nsw;
}

// ---------------------------------------------------------------------------------------
// Rank sort of 128 distinct 64-bit keys (two per lane) in DESCENDING order, one wavefront.
// The keys are grouped into 64 value buckets (any bucket function that is monotone in the key), the buckets laid
// out in descending order, and every key counts the keys of ITS OWN bucket that sort before it: ~4 mates (a dozen
// in the fullest bucket) instead of all 128 -- the all-pairs form spent 512 vector instructions on 256 compare +
// add-carry pairs per lane.  Entries past a bucket's end belong to lower buckets, i.e. smaller keys, so the count
// may run on to the wave's largest bucket size without a mask; 128 zero entries pad the array (a bucket may start at
// 127 and the wave's fullest bucket may hold all 128 keys -- clipped or saturated inputs -- so reads reach entry 254).
// ---------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) RankLds {
    u64 bk[256];             // the keys grouped by bucket (descending), then 128 zero entries (never "before me")
    int hist[64], cur[64];   // bucket counts / placement cursors
    int base[64];            // number of keys in higher buckets
};

// bucket of a magnitude (given as its bit pattern) on a linear scale up to the frame's largest magnitude; monotone in
// the bits for every input (a NaN lands in the top bucket, as its bit pattern demands)
__device__ __forceinline__ float bucket_scale(unsigned a1, unsigned a2)
{
    const int m = wave_max_i32((int)(a1 > a2 ? a1 : a2));
    return 63.5f / __int_as_float(m);
}
__device__ __forceinline__ int bucket_of(unsigned a, float scale) { return (int)__builtin_fminf(__uint_as_float(a) * scale, 63.0f); }

__device__ __forceinline__ void bucket_ranks(RankLds &R, u64 k1, u64 k2, int b1, int b2, int lane, int &r1, int &r2)
{
    R.hist[lane] = 0; R.cur[lane] = 0; R.bk[128 + lane] = 0ull; R.bk[192 + lane] = 0ull;
    wave_fence();
    atomicAdd(&R.hist[b1], 1); atomicAdd(&R.hist[b2], 1);
    wave_fence();
    const int h = R.hist[lane];
    R.base[lane] = 128 - wave_incl_add_dpp(h);                 // keys in the buckets above mine
    const int nmax = wave_max_i32(h);
    wave_fence();
    const int s1 = R.base[b1], s2 = R.base[b2];
    R.bk[s1 + atomicAdd(&R.cur[b1], 1)] = k1;
    R.bk[s2 + atomicAdd(&R.cur[b2], 1)] = k2;
    wave_fence();
    r1 = s1; r2 = s2;
    for (int jj = 0; jj < nmax; ++jj) { r1 += R.bk[s1 + jj] > k1; r2 += R.bk[s2 + jj] > k2; }
}

// r += (k > a) as v_cmp_gt_i32 + v_addc through VCC (both 4-byte encodings)
__device__ __forceinline__ void rank_gt(int &r, int a, int k)
{
    asm("v_cmp_gt_i32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(r) : "v"(a), "v"(k) : "vcc");
}

__device__ __forceinline__ int below_mask(const unsigned (&m)[4], int x)
{
    // number of set bits of the 128-bit mask strictly below position x
    int c = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int t = x - 32 * d;
        const unsigned bm = t <= 0 ? 0u : (t >= 32 ? 0xFFFFFFFFu : ((1u << t) - 1u));
        c += __popc(m[d] & bm);
    }
    return c;
}

// Byte LUTs of partial weight sums: lut[b][v] = sum over the set bits t of v (ascending t) of w[8 b + t].
// Lane l owns the entries whose low five index bits equal l & 31 and whose top index bit equals
// l >> 5 (4 x 8 = 32 entries): within each 32-lane LDS group the stores of one instruction then
// hit 32 distinct banks.  (Letting a lane fill 32 CONSECUTIVE entries puts all 64 lanes on one
// bank per store: 32-way conflicts, which cost as much LDS time as all the lookups of a scan.)
template <int NB>
__device__ __forceinline__ void build_byte_luts(float (*lut)[256], const float *w, int lane)
{
    const int lo5 = lane & 31, top = lane >> 5;
    // the lane's index bits as 0.0f / 1.0f factors: "add w if the bit is set" becomes one fma (exact: the
    // product by 0 or 1 is exact and the fma rounds once, like the add) instead of a select plus an add --
    // v_cndmask issues at half the rate of v_fma on gfx950.  (Weights are finite magnitudes.)
    const float b0 = (lo5 & 1) ? 1.0f : 0.0f, b1 = (lo5 & 2) ? 1.0f : 0.0f, b2 = (lo5 & 4) ? 1.0f : 0.0f;
    const float b3 = (lo5 & 8) ? 1.0f : 0.0f, b4 = (lo5 & 16) ? 1.0f : 0.0f, b7 = top ? 1.0f : 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float4 wa = *reinterpret_cast<const float4 *>(&w[8 * b]);       // broadcast reads
        const float4 wb = *reinterpret_cast<const float4 *>(&w[8 * b + 4]);
        float base = wa.x * b0;
        base = __builtin_fmaf(wa.y, b1, base);
        base = __builtin_fmaf(wa.z, b2, base);
        base = __builtin_fmaf(wa.w, b3, base);
        base = __builtin_fmaf(wb.x, b4, base);
        const float h7 = wb.w * b7;
        // index bits 5, 6 = h, bit 7 = top; sums stay in ascending position order
        lut[b][(top * 4 + 0) * 32 + lo5] = base + h7;
        lut[b][(top * 4 + 1) * 32 + lo5] = (base + wb.y) + h7;
        lut[b][(top * 4 + 2) * 32 + lo5] = (base + wb.z) + h7;
        lut[b][(top * 4 + 3) * 32 + lo5] = ((base + wb.y) + wb.z) + h7;
    }
}

// lut[B][byte B of D]: the byte is extracted and scaled to a byte offset by ONE SDWA shift (instead of
// v_bfe + v_lshl_add); with a compile-time LDS base the table offset folds into the ds_read immediate
template <int B>
__device__ __forceinline__ float lut_byte(const float (*lut)[256], u64 D)
{
    const unsigned word = B < 4 ? (unsigned)D : (unsigned)(D >> 32);
    unsigned off;
    if constexpr ((B & 3) == 0) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(word));
    else if constexpr ((B & 3) == 1) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "v"(word));
    else if constexpr ((B & 3) == 2) asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "v"(word));
    else asm("v_lshlrev_b32_sdwa %0, 2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "v"(word));
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(lut[B]) + off);
}

}  // namespace ldpc
