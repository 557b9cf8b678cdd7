// Ordered-statistics decoding kernels for (128,64) codes on gfx950 (MI355X).
//
// Reference (paths relative to LDPC_128/ of the reference):
//   swapped_info / identify_mrb / full_gf2elim   PB_OSD/pb_testing.py:231-320 (== FS_OSD/fs_testing.py:233-322)
//   generate_teps / convention_osd_main           FS_OSD/convention_osd.py:13-76
//
// One frame per wavefront, no MFMA (bit and compare work); the long kernels run one wavefront per
// workgroup (compile-time LDS addresses, frames balanced by the hardware dispatcher).  Kernels:
//
//   osd_front_kernel    reliability sort (rank sort of |y|, ties -> lower index), gather of the
//                       G columns in sorted order, GF(2) Gauss-Jordan with the reference's pivot
//                       rule (ge_columns, ldpc_wave.h), MRB/LRB bookkeeping.  The matrix lives
//                       COLUMN-major in registers: lane p holds columns p and p+64 as two 64-bit
//                       words (bit = row), so the column gather is one load per lane, a pivot step
//                       is a handful of wave-uniform scalars (v_readlane / compare / s_ff1) plus
//                       ~12 VALU ops, row exchanges only touch a lane-resident row map and column
//                       exchanges move two lanes.  Output: perm (original bit at each primed
//                       position) and the rows of P' (G' = [I | P']) after a 64x64 bit transpose.
//   osd_search2_kernel  conventional order 2, register-resident, two-stage scan with an exact
//                       prefix early exit and survivor compaction (the headline configuration).
//   osd_search_kernel   conventional order-p search over the reference's TEP table: per frame a
//                       byte-indexed LUT of partial |y'| sums in LDS (8 x 256 floats), each lane
//                       evaluates one TEP per round: parity word = d0 ^ P'[i] ^ P'[j] ..., metric
//                       = flipped-MRB weights + 8 LUT terms in a FIXED order (the canonical order
//                       the oracle uses, see oracle/np_oracle.py weighted_distance), first minimum.
//   osd_fs_kernel       FS-OSD (FS_OSD/fs_testing.py:22-64,129-161), 64 TEPs per round.
//   osd_pb_kernel       PB-OSD (PB_OSD/pb_testing.py:35-41,100-149,366-500), best-first frontier.
//   osd_ge_kernel       full_gf2elim on caller-supplied matrices; osd_counts_kernel: success counters.
#include <math.h>
#include <stdlib.h>

#include "ldpc_internal.h"
#include "ldpc_wave.h"

namespace ldpc {


// ---------------------------------------------------------------------------------------
// ldpc_osd_ge: elimination of caller-supplied matrices (row-major in, row-major out)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void osd_ge_kernel(const u64 *__restrict__ rows_in, long long F,
                                                     u64 *__restrict__ rows_out, unsigned char *__restrict__ swaps,
                                                     int *__restrict__ nswaps)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (long long f = wave; f < F; f += (long long)gridDim.x * 4) {
        const u64 *src = rows_in + f * 128;
        u64 C1 = transpose64(src[lane * 2], lane);      // row r, columns 0..63  -> column lane, bit r
        u64 C2 = transpose64(src[lane * 2 + 1], lane);
        int rho = lane, idx1 = lane, idx2 = lane + 64;
        const int ns = ge_columns(C1, C2, rho, idx1, idx2, lane, swaps ? swaps + f * 128 : nullptr);
        // back to row-major, logical row order: lane = physical row after the transpose
        const u64 R1 = transpose64(C1, lane), R2 = transpose64(C2, lane);
        rows_out[f * 128 + lane * 2] = shfl64(R1, rho);
        rows_out[f * 128 + lane * 2 + 1] = shfl64(R2, rho);
        if (nswaps && lane == 0) nswaps[f] = ns;
    }
}

// ---------------------------------------------------------------------------------------
// OSD front end
// ---------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) FrontLds {
    int abits[128];          // |y| as integer keys
    u64 colbuf[64];          // parity columns in primed order
    unsigned mask[4];        // 128-bit membership mask of the MRB indices
    unsigned char pi1[128];  // sorted position -> original bit
    unsigned char rowsrc[64];
    unsigned char perm[128]; // primed position -> original bit
};

struct FrontResult {
    int o1, o2;   // original bit index of primed positions lane and 64 + lane
    u64 Prow;     // row `lane` of P'
    int ns;       // recorded column exchanges (-1: rank-deficient)
};


// sort + column gather + elimination + bookkeeping of one frame (one wavefront); results in registers
__device__ __forceinline__ FrontResult front_device(FrontLds &L, const float *__restrict__ y, long long src,
                                                    const u64 *__restrict__ Gcols, int lane)
{
    // ---- reliability sort: rank of each |y| in descending order, ties -> lower index ------
    const int a1 = __float_as_int(y[src * 128 + lane]) & 0x7FFFFFFF;
    const int a2 = __float_as_int(y[src * 128 + 64 + lane]) & 0x7FFFFFFF;
    L.abits[lane] = a1;
    L.abits[lane + 64] = a2;
    wave_fence();
    // rank = number of keys that sort before mine.  Fast path: strict integer compares only
    // (v_cmp + v_addc through VCC, 2 instructions per key and element); equal keys then collide
    // on a rank, which the read-back below detects, and only such frames (exact float ties,
    // ~5e-4 of random frames) redo the count with the full "lower index first" rule.
    int r1 = 0, r2 = 0;
#pragma unroll 8
    for (int u4 = 0; u4 < 32; ++u4) {
        const int4 kq = *reinterpret_cast<const int4 *>(&L.abits[u4 * 4]);
        rank_gt(r1, a1, kq.x); rank_gt(r2, a2, kq.x);
        rank_gt(r1, a1, kq.y); rank_gt(r2, a2, kq.y);
        rank_gt(r1, a1, kq.z); rank_gt(r2, a2, kq.z);
        rank_gt(r1, a1, kq.w); rank_gt(r2, a2, kq.w);
    }
    L.pi1[r1] = (unsigned char)lane;
    L.pi1[r2] = (unsigned char)(lane + 64);
    wave_fence();
    if (__ballot(L.pi1[r1] != lane || L.pi1[r2] != lane + 64)) {
        wave_fence();
        r1 = 0; r2 = 0;
        // "u before v"  <=>  a_u > a_v  or  (a_u == a_v and u < v)
        for (int u = 0; u < 128; ++u) {
            const int ku = L.abits[u];
            r1 += (ku > a1) || (ku == a1 && u < lane);
            r2 += (ku > a2) || (ku == a2 && u < lane + 64);
        }
    }
    wave_fence();
    L.pi1[r1] = (unsigned char)lane;
    L.pi1[r2] = (unsigned char)(lane + 64);
    if (lane < 4) L.mask[lane] = 0;
    wave_fence();
    // ---- G with columns in sorted order, column-major ------------------------------------
    u64 C1 = Gcols[L.pi1[lane]];
    u64 C2 = Gcols[L.pi1[lane + 64]];
    int rho = lane, idx1 = lane, idx2 = lane + 64;
    const int ns = ge_columns(C1, C2, rho, idx1, idx2, lane, nullptr);
    // ---- identify_mrb bookkeeping (pb_testing.py:276-304) --------------------------------
    atomicOr(&L.mask[idx1 >> 5], 1u << (idx1 & 31));
    wave_fence();
    const unsigned m[4] = {L.mask[0], L.mask[1], L.mask[2], L.mask[3]};
    const int rankM = below_mask(m, idx1);         // new MRB position of slot `lane`
    const int rankL = idx2 - below_mask(m, idx2);  // new parity column of slot `lane`
    L.perm[rankM] = L.pi1[idx1];
    L.perm[64 + rankL] = L.pi1[idx2];
    L.colbuf[rankL] = C2;
    L.rowsrc[rankM] = (unsigned char)rho;          // pivot of MRB slot `lane` is physical row rho
    wave_fence();
    const u64 R = transpose64(L.colbuf[lane], lane);   // lane = physical row, bit = parity column
    FrontResult res;
    res.Prow = shfl64(R, L.rowsrc[lane]);
    res.o1 = L.perm[lane];
    res.o2 = L.perm[64 + lane];
    res.ns = ns;
    wave_fence();
    return res;
}

__global__ __launch_bounds__(64) void osd_front_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                        const int *__restrict__ count, long long F,
                                                        const u64 *__restrict__ Gcols,
                                                        unsigned char *__restrict__ perm_out,
                                                        u64 *__restrict__ parity_out, int *__restrict__ nswaps)
{
    __shared__ FrontLds L;   // one wavefront per workgroup
    const int lane = threadIdx.x;
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }

    for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const long long src = index ? index[f] : f;
        const FrontResult res = front_device(L, y, src, Gcols, lane);
        perm_out[f * 128 + lane] = (unsigned char)res.o1;
        perm_out[f * 128 + 64 + lane] = (unsigned char)res.o2;
        parity_out[f * 64 + lane] = res.Prow;
        if (nswaps && lane == 0) nswaps[f] = res.ns;
    }
}

// ---------------------------------------------------------------------------------------
// conventional order-p search (convention_osd_main, convention_osd.py:49-76)
// ---------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) SearchLds {
    float lut[8][256];   // lut[b][v] = sum of |y'[64+8b+t]| over the set bits t of v, ascending t
    u64 P[64];           // rows of P'
    float w[128];        // |y'|
    u64 cw[2];           // codeword being assembled in original bit order
    unsigned char perm[128];
};

template <int B>
__device__ __forceinline__ float lut_term(const SearchLds &L, u64 D) { return lut_byte<B>(L.lut, D); }

__device__ __forceinline__ float tep_cost(const SearchLds &L, float mrb, u64 D)
{
    float acc = mrb;
    acc = acc + lut_term<0>(L, D); acc = acc + lut_term<1>(L, D); acc = acc + lut_term<2>(L, D); acc = acc + lut_term<3>(L, D);
    acc = acc + lut_term<4>(L, D); acc = acc + lut_term<5>(L, D); acc = acc + lut_term<6>(L, D); acc = acc + lut_term<7>(L, D);
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

// per-frame set-up shared by every search: primed-order values into LDS, hard decisions, byte
// LUTs, and the parity discrepancy d0 of the order-0 candidate
struct SearchFrame {
    u64 hm, hp, d0;   // hard decisions of the MRB / parity part (y' > 0 ? 0 : 1), order-0 discrepancy
    int o1, o2;       // original bit index of primed positions lane and 64 + lane
};

__device__ __forceinline__ SearchFrame search_prepare_regs(SearchLds &L, const float *__restrict__ y, long long src,
                                                           int o1, int o2, u64 Prow, int lane)
{
    SearchFrame S;
    S.o1 = o1;
    S.o2 = o2;
    const float y1 = y[src * 128 + S.o1], y2 = y[src * 128 + S.o2];   // y'[p] = y[perm[p]]
    L.perm[lane] = (unsigned char)S.o1;
    L.perm[lane + 64] = (unsigned char)S.o2;
    L.w[lane] = __builtin_fabsf(y1);
    L.w[lane + 64] = __builtin_fabsf(y2);
    L.P[lane] = Prow;
    if (lane < 2) L.cw[lane] = 0;
    S.hm = __ballot(!(y1 > 0.0f));
    S.hp = __ballot(!(y2 > 0.0f));
    wave_fence();
    build_byte_luts<8>(L.lut, &L.w[64], lane);
    // d0 = (u0 . P') ^ h_parity : XOR-reduce the rows selected by the MRB hard decisions
    S.d0 = wave_xor64(((S.hm >> lane) & 1) ? Prow : 0ull) ^ S.hp;
    wave_fence();
    return S;
}

__device__ __forceinline__ SearchFrame search_prepare(SearchLds &L, const float *__restrict__ y, long long src,
                                                      const unsigned char *__restrict__ perm_in,
                                                      const u64 *__restrict__ parity_in, long long f, int lane)
{
    return search_prepare_regs(L, y, src, perm_in[f * 128 + lane], perm_in[f * 128 + 64 + lane], parity_in[f * 64 + lane], lane);
}

// candidate (E = flipped MRB positions, D = parity discrepancy) -> codeword in ORIGINAL bit order
__device__ __forceinline__ void search_finish(SearchLds &L, const SearchFrame &S, u64 E, u64 D, long long f, int lane,
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

// WAVES = 1 for the long scans (one wavefront per workgroup: compile-time LDS base for the LUT reads, frames
// balanced by the dispatcher), 4 for orders 0 and 1, where a frame is too little work to pay for a workgroup
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void osd_search_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                                const int *__restrict__ count, long long F,
                                                                const unsigned char *__restrict__ perm_in,
                                                                const u64 *__restrict__ parity_in,
                                                                const uchar4 *__restrict__ teps, int ntep,
                                                                u64 *__restrict__ cw_out, float *__restrict__ metric_out,
                                                                int *__restrict__ best_out, int *__restrict__ ntep_out)
{
    __shared__ SearchLds lds[WAVES];
    const int lane = threadIdx.x & 63;
    SearchLds &L = lds[WAVES == 1 ? 0 : threadIdx.x >> 6];
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }
    const long long wave = (long long)blockIdx.x * WAVES + (threadIdx.x >> 6);

    for (long long f = wave; f < nframes; f += (long long)gridDim.x * WAVES) {
        const long long src = index ? index[f] : f;
        const SearchFrame S = search_prepare(L, y, src, perm_in, parity_in, f, lane);
        // scan the TEP table, one TEP per lane per round; strict '<' keeps the first minimum
        float best = __builtin_inff();
        int bestt = 0x7FFFFFFF;
        u64 bestD = 0, bestE = 0;
        // (exact early exit on the metric prefix, see tep_cost_bounded; `bound` = the wave's best so far)
        float bound = __builtin_inff();
        int trip = 0;
        for (int t0 = 0; t0 < ntep; t0 += 64, ++trip) {
            const int t = t0 + lane;
            if (t < ntep) {
                u64 D, E;
                float mrb, c;
                tep_apply(L, teps[t], S.d0, D, E, mrb);
                if (tep_cost_bounded(L, mrb, D, bound, c) && c < best) { best = c; bestt = t; bestD = D; bestE = E; }
            }
            if ((trip & 7) == 0) bound = wave_min_f32(best);
        }
        wave_argmin(best, bestt, bestD, bestE, lane);
        search_finish(L, S, bestE, bestD, f, lane, cw_out);
        if (lane == 0) {
            if (metric_out) metric_out[f] = best;
            if (best_out) best_out[f] = bestt;
            if (ntep_out) ntep_out[f] = ntep;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Order-2 conventional search, register-resident: same result as osd_search_kernel with the
// 2081-entry table, without per-TEP reads of P' / |y'| / the table.  Lane l keeps P'[l], P'[63-l],
// |y'_l|, |y'_{63-l}|; order 1 = one TEP per lane; order 2 = 32 rounds over a triangular pairing
//   lanes l > r : pair (r, l)          lanes l <= r : pair (62 - r, 63 - l)      (r = 0..31)
// (63 - r) + (r + 1) = 64 pairs per round, 2016 in all; the pivot rows of a round arrive by
// v_readlane.  "First minimum in table order" is kept by ranking the TEPs with the closed form of
// the reference's ordering (weight class, then descending index sum, then ascending first index;
// convention_osd.py:19-24): rank({}) = 0, rank({p}) = 64 - p, rank({i<j}) = 65 + base[i+j] +
// i - max(0, i+j-63), base[s] = number of pairs with a larger sum (uploaded table).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int tep2_rank(int bi, int bj, const int *__restrict__ base2)
{
    if (bj < 0) return 0;
    if (bi < 0) return 64 - bj;
    const int s = bi + bj;
    return 65 + base2[s] + bi - (s > 63 ? s - 63 : 0);
}

// The order-0/1/2 scan of one frame per wavefront (one wavefront per workgroup: the LDS base is then a
// compile-time constant and every LUT read is "SDWA shift + ds_read with an immediate offset").
//
// Order 2 runs in two stages.  Stage 1 (every round, all lanes): candidate D, MRB weight and the first two
// parity bytes; a candidate whose prefix already exceeds `bound` (the smallest complete metric seen by any
// lane) can neither win nor tie -- every further term is >= 0 -- and is dropped: ~70-95 % of the TEPs.
// The survivors are appended to a 128-entry LDS ring (ballot + mbcnt compaction) and stage 2 finishes them
// 64 at a time, so the six remaining LUT reads and the arg-min bookkeeping run on full wavefronts only.
// The order of evaluation changes, the result does not: the arg-min is on (metric, table rank).
// (Measured alternatives: four wavefronts per frame sharing one LUT set -- 139 us against 123 us, the
//  barriers and the single-wave prologue cost more than the occupancy gains; pivot rows through the scalar
//  cache instead of v_readlane -- no difference.)
struct __attribute__((aligned(16))) Search2Lds {
    SearchLds s;
    uint4 q[128];   // survivors: D.lo, D.hi, prefix metric bits, r * 64 + lane
};

__device__ __forceinline__ void search2_finish_batch(const SearchLds &L, uint4 e, bool valid, const int *__restrict__ base2,
                                                     float &best, int &bi, int &bj, u64 &bestD)
{
    if (!valid) return;
    const u64 D = ((u64)e.y << 32) | e.x;
    float acc = __uint_as_float(e.z);
    acc = acc + lut_term<2>(L, D); acc = acc + lut_term<3>(L, D); acc = acc + lut_term<4>(L, D);
    acc = acc + lut_term<5>(L, D); acc = acc + lut_term<6>(L, D); acc = acc + lut_term<7>(L, D);
    if (!(acc <= best)) return;
    const int r = (int)(e.w >> 6), l = (int)(e.w & 63);
    const bool up = l > r;
    const int ci = up ? r : 62 - r, cj = up ? l : 63 - l;
    // equal metrics are ordered by table rank (practically never taken)
    if (acc < best || tep2_rank(ci, cj, base2) < tep2_rank(bi, bj, base2)) { best = acc; bi = ci; bj = cj; bestD = D; }
}

__device__ __forceinline__ void search2_device(Search2Lds &LL, const SearchFrame &S, const int *__restrict__ base2, int lane,
                                               float &best_out, int &rank_out, u64 &D_out, u64 &E_out)
{
    SearchLds &L = LL.s;
    const u64 Pl = L.P[lane], Pm = L.P[63 - lane];
    const float wl = L.w[lane], wm = L.w[63 - lane];
    // order 0 (rank 0, identical in every lane), then order 1: lane l owns TEP {l}
    float best = tep_cost(L, 0.0f, S.d0);
    int bi = -1, bj = -1;
    u64 bestD = S.d0;
    {
        const u64 D = S.d0 ^ Pl;
        const float c = tep_cost(L, wl, D);
        if (c < best) { best = c; bj = lane; bestD = D; }      // a tie keeps the lower rank (order 0)
    }
    float bound = wave_min_f32(best);
    int qhead = 0, qn = 0;   // ring state (wave-uniform)
    for (int r = 0; r < 32; ++r) {
        const u64 Pr = readlane64(Pl, r), Pq = readlane64(Pl, 62 - r);
        const float wr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), r));
        const float wq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), 62 - r));
        const bool up = lane > r;
        const bool active = up || r < 31;        // at r = 31 the lower half would repeat i = 31
        const float M = up ? (wr + wl) : (wq + wm);            // |y'_i| + |y'_j|, i < j
        const u64 D = S.d0 ^ (up ? (Pr ^ Pl) : (Pq ^ Pm));
        float acc = M + lut_term<0>(L, D);
        acc = acc + lut_term<1>(L, D);
        const bool keep = active && !(acc > bound);
        const u64 km = __ballot(keep);
        if (km) {
            if (keep) {
                const int slot = (qhead + qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0))) & 127;
                LL.q[slot] = make_uint4((unsigned)D, (unsigned)(D >> 32), __float_as_uint(acc), (unsigned)(r * 64 + lane));
            }
            qn += __popcll(km);
            if (qn >= 64) {
                wave_fence();
                search2_finish_batch(L, LL.q[(qhead + lane) & 127], true, base2, best, bi, bj, bestD);
                qhead = (qhead + 64) & 127;
                qn -= 64;
                bound = wave_min_f32(best);
                wave_fence();
            }
        }
    }
    wave_fence();
    search2_finish_batch(L, LL.q[(qhead + lane) & 127], lane < qn, base2, best, bi, bj, bestD);
    wave_fence();
    int bestt = tep2_rank(bi, bj, base2);
    u64 bestE = (bi >= 0 ? 1ull << bi : 0ull) | (bj >= 0 ? 1ull << bj : 0ull);
    wave_argmin(best, bestt, bestD, bestE, lane);
    best_out = best; rank_out = bestt; D_out = bestD; E_out = bestE;
}

__global__ __launch_bounds__(64) void osd_search2_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                         const int *__restrict__ count, long long F,
                                                         const unsigned char *__restrict__ perm_in,
                                                         const u64 *__restrict__ parity_in,
                                                         const int *__restrict__ base2,
                                                         u64 *__restrict__ cw_out, float *__restrict__ metric_out,
                                                         int *__restrict__ best_out, int *__restrict__ ntep_out)
{
    __shared__ Search2Lds LL;
    SearchLds &L = LL.s;
    const int lane = threadIdx.x;
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }

    for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const long long src = index ? index[f] : f;
        const SearchFrame S = search_prepare(L, y, src, perm_in, parity_in, f, lane);
        float best; int bestt; u64 bestD, bestE;
        search2_device(LL, S, base2, lane, best, bestt, bestD, bestE);
        search_finish(L, S, bestE, bestD, f, lane, cw_out);
        if (lane == 0) {
            if (metric_out) metric_out[f] = best;
            if (best_out) best_out[f] = bestt;
            if (ntep_out) ntep_out[f] = 2081;
        }
    }
}

// ---------------------------------------------------------------------------------------
// FS-OSD (fs_osd, FS_OSD/fs_testing.py:129-161): order-by-order scan in the order of
// generate_sequential_teps (:32-49) with two Hamming-distance rules (one_tep_compare :51-64):
//   HD < tau_e            -> stop everything (the candidate is appended to optimal_list, :143-146)
//   HD < tau_psc and a smaller weighted distance -> new best (:147-152)
// and a lower bound per order: scan weight w only if (sum of the w least reliable MRB |y'|) +
// beta (n-k) < best so far (:137-139, acquire_pnc_boundary :22-30).  64 TEPs are evaluated per
// round; the sequential semantics are recovered with a ballot (first tau_e hit) and an arg-min
// over the lanes before it.  quirk = 1 returns what the reference keeps in `optimal_codeword`
// (the best BEFORE a tau_e hit), quirk = 0 the tau_e candidate itself.
// ---------------------------------------------------------------------------------------
struct FsParams {
    int order, quirk;
    float beta_term, tau_e, tau_psc;
    int cls_off[4], cls_cnt[4];   // weight class w: offset / count inside the FS-ordered table
};

// (one wavefront per workgroup, as the order-2 scan: compile-time LDS base for the LUT reads, and the
//  dispatcher balances the very uneven per-frame TEP counts)
__global__ __launch_bounds__(64) void osd_fs_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                    const int *__restrict__ count, long long F,
                                                    const unsigned char *__restrict__ perm_in,
                                                    const u64 *__restrict__ parity_in,
                                                    const uchar4 *__restrict__ teps_fs, FsParams P,
                                                    u64 *__restrict__ cw_out, float *__restrict__ metric_out,
                                                    int *__restrict__ best_out, int *__restrict__ ntep_out)
{
    __shared__ SearchLds L;
    const int lane = threadIdx.x;
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }

    for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const long long src = index ? index[f] : f;
        const SearchFrame S = search_prepare(L, y, src, perm_in, parity_in, f, lane);
        float best = tep_cost(L, 0.0f, S.d0);      // all-zero TEP (:131)
        u64 bestD = S.d0, bestE = 0, hitD = 0, hitE = 0;
        float hitc = 0.0f;
        int bestidx = 0, ntep = 1, visited = 1, hitidx = 0;
        bool hit = false;
        if (!((float)__popcll(S.d0) < P.tau_e)) {
            for (int w = 1; w <= P.order && !hit; ++w) {
                float bsum = 0.0f;                  // w least reliable MRB values, ascending position
                for (int t = 64 - w; t < 64; ++t) bsum = bsum + L.w[t];
                if (!(bsum + P.beta_term < best)) break;
                const int cnt = P.cls_cnt[w];
                const uchar4 *tab = teps_fs + P.cls_off[w];
                for (int t0 = 0; t0 < cnt && !hit; t0 += 64) {
                    const int t = t0 + lane;
                    const bool valid = t < cnt;
                    u64 D = 0, E = 0;
                    float mrb = 0.0f;
                    if (valid) tep_apply(L, tab[t], S.d0, D, E, mrb);
                    const float hd = (float)(w + __popcll(D));
                    const u64 stop = __ballot(valid && hd < P.tau_e);
                    const int lim = stop ? __builtin_ctzll(stop) : 64;
                    const int nvalid = (cnt - t0) < 64 ? (cnt - t0) : 64;
                    ntep += stop ? lim + 1 : nvalid;
                    // best among the TEPs visited before the stop that pass the tau_psc rule: the metric is only
                    // needed for those, and only if it can beat `best` (exact prefix early exit, tep_cost_bounded)
                    float cc = __builtin_inff();
                    if (valid && lane < lim && hd < P.tau_psc) {
                        float c;
                        if (tep_cost_bounded(L, mrb, D, best, c)) cc = c;
                    }
                    if (__ballot(cc < best)) {
                        int ci = lane;
                        u64 cD = D, cE = E;
                        wave_argmin(cc, ci, cD, cE, lane);
                        best = cc; bestD = cD; bestE = cE; bestidx = visited + t0 + ci;
                    }
                    if (stop) {
                        hit = true;
                        hitD = readlane64(D, lim); hitE = readlane64(E, lim);
                        hitc = tep_cost(L, __shfl(mrb, lim, 64), hitD);   // the stopping candidate's own metric
                        hitidx = visited + t0 + lim;
                    }
                }
                visited += cnt;
            }
        }
        const bool use_hit = hit && !P.quirk;
        search_finish(L, S, use_hit ? hitE : bestE, use_hit ? hitD : bestD, f, lane, cw_out);
        if (lane == 0) {
            if (metric_out) metric_out[f] = use_hit ? hitc : best;
            if (best_out) best_out[f] = use_hit ? hitidx : bestidx;
            if (ntep_out) ntep_out[f] = ntep;
        }
    }
}

// ---------------------------------------------------------------------------------------
// PB-OSD (pb_osd, PB_OSD/pb_testing.py:100-149): best-first TEP generation from a frontier
// (optimal_tep_sequence :366-397) with two probabilistic stopping rules
// (acquire_prob_promising :448-461, acquire_p_e_suc :423-436, thresholds :485-500).
// The search is sequential per frame; a wavefront owns a frame, the lanes share the frontier
// scan (arg-min on (reliability sum, insertion number) == "first minimum in list order") and
// the per-position set-up, everything else is wave-uniform.  All probabilities follow the float
// conventions of the oracle (oracle/ldpc_oracle.c orc_pb_osd): float32 with det_expf (IEEE
// + - * / only, so host and device agree bit for bit), float64 binomial-CDF recurrences,
// threshold comparisons in float64.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float det_expf(float x)
{
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) return 0.0f;
    const float kf = __builtin_floorf(x * 1.44269504f + 0.5f);
    const float r = (x - kf * 0.693359375f) - kf * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    const float e = (p * (r * r) + r) + 1.0f;
    return e * __int_as_float(((int)kf + 127) << 23);
}

// Frontier = the reference's growing TEP list (optimal_tep_sequence :366-397) kept in INSERTION order:
// a popped entry is tombstoned in place (sum = +inf), children are appended, so "first minimum in list
// order" is the arg-min on (sum, slot).  A search that never stops visits all N_max TEPs with a list of
// tens of thousands of entries, so the arg-min is kept hierarchical: cmin[c] = best (sum, slot) of the 64
// slots of chunk c, smin[s] = best of the 64 chunks of super-chunk s.  A pop reads the <= 32 super-minima,
// then re-reduces one chunk and one super-chunk: ~3 wave reductions per TEP whatever the list length.
// Slots < kPbLdsSlots and chunk minima < kPbLdsChunks live in LDS, the rest in a per-wave global area.
struct PbEntry {
    float sum;          // reliability sum of the flipped MRB positions (ascending, sequential); +inf = removed
    unsigned pos;       // slots: pos0 | pos1 << 8 | pos2 << 16 | weight << 24;  minima: slot index
};
constexpr int kPbLdsSlots = 512, kPbLdsChunks = 64, kPbSuper = 32;   // 32 super-chunks x 4096 slots >= 2 N_max (order 3)

struct __attribute__((aligned(16))) PbLds {
    double cdfA[65];             // P[Bin(64, p1) <= b]
    double cdfH[65];             // P[Bin(64, 1/2) <= b] (copied once per wavefront: a global read per TEP sat on the critical path)
    float q[128];                // sigmoid(c4 |y'_p|)
    PbEntry fr[kPbLdsSlots];     // head of the list
    PbEntry cmin[kPbLdsChunks];  // chunk minima of the first 4096 slots
    PbEntry smin[kPbSuper];      // super-chunk minima
};

struct PbParams {
    int order, nmax;
    float c4;
    long long cmin_off;          // offset of the spilled chunk minima inside a wave's global area
};

struct PbList {
    PbLds *B;
    PbEntry *spill;              // slots >= kPbLdsSlots, then chunk minima >= kPbLdsChunks at cmin_off
    long long cmin_off;
    __device__ __forceinline__ PbEntry slot(int i) const { return i < kPbLdsSlots ? B->fr[i] : spill[i - kPbLdsSlots]; }
    __device__ __forceinline__ void set_slot(int i, PbEntry e) const { if (i < kPbLdsSlots) B->fr[i] = e; else spill[i - kPbLdsSlots] = e; }
    __device__ __forceinline__ PbEntry cmin(int c) const { return c < kPbLdsChunks ? B->cmin[c] : spill[cmin_off + c - kPbLdsChunks]; }
    __device__ __forceinline__ void set_cmin(int c, PbEntry e) const { if (c < kPbLdsChunks) B->cmin[c] = e; else spill[cmin_off + c - kPbLdsChunks] = e; }
};

// wave arg-min on (sum, index): lower index wins ties; result in every lane
__device__ __forceinline__ void argmin_si(float &s, int &idx, int lane)
{
    const float m = wave_min_f32(s);
    idx = wave_min_i32(s == m ? idx : 0x7FFFFFFF);
    s = m;
}

__global__ __launch_bounds__(256) void osd_pb_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                     const int *__restrict__ count, long long F,
                                                     const unsigned char *__restrict__ perm_in,
                                                     const u64 *__restrict__ parity_in, PbParams P,
                                                     const double *__restrict__ cdf_half /*[65]*/,
                                                     const double *__restrict__ coef /*[64] (64-i)/(i+1)*/,
                                                     PbEntry *__restrict__ spill_all, long long spill_stride,
                                                     int *__restrict__ queue /* zeroed per launch */,
                                                     u64 *__restrict__ cw_out, float *__restrict__ metric_out,
                                                     int *__restrict__ best_out, int *__restrict__ ntep_out,
                                                     int *__restrict__ aux_out /*[F][4] or null*/)
{
    __shared__ SearchLds lds[4];
    __shared__ PbLds pbl[4];
    const int lane = threadIdx.x & 63;
    SearchLds &L = lds[threadIdx.x >> 6];
    PbLds &B = pbl[threadIdx.x >> 6];
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    PbEntry *spill = spill_all + wave * spill_stride;
    B.cdfH[lane] = cdf_half[lane];
    if (lane == 0) B.cdfH[64] = cdf_half[64];
    wave_fence();

    // frames are handed out through a device counter: PB-OSD run times differ by orders of magnitude between
    // frames (a frame on which no rule fires visits all N_max TEPs), a static assignment would wait for the
    // unluckiest wave
    for (;;) {
        int fq = 0;
        if (lane == 0) fq = atomicAdd(queue, 1);
        const long long f = __builtin_amdgcn_readfirstlane(fq);
        if (f >= nframes) break;
        const long long src = index ? index[f] : f;
        const SearchFrame S = search_prepare(L, y, src, perm_in, parity_in, f, lane);
        B.q[lane] = 1.0f / (1.0f + det_expf(-(P.c4 * L.w[lane])));
        B.q[lane + 64] = 1.0f / (1.0f + det_expf(-(P.c4 * L.w[lane + 64])));
        wave_fence();
        // sequential (ascending position) means / product, as the oracle defines them
        float a1 = 0.0f, aw = 0.0f, at = 0.0f, spl = 1.0f;
#pragma unroll 4
        for (int p = 0; p < 64; ++p) {
            a1 = a1 + B.q[64 + p];
            aw = aw + L.w[64 + p];
            at = at + B.q[p];
            spl = spl * (1.0f - B.q[p]);
        }
        const float p1 = a1 / 64.0f, lrb_mean = aw / 64.0f, pt = at / 64.0f;
        // binomial CDF tables by the pmf recurrence (float64): full table for p1, up to `order` for pt
        double niu;
        {
            double q = 1.0 - (double)p1, t = q;
            for (int s = 0; s < 6; ++s) t = t * t;
            const double ratio = (double)p1 / q;
            double acc = t;
            if (lane == 0) B.cdfA[0] = acc;
#pragma unroll 2
            for (int i = 0; i < 64; ++i) {
                t = t * coef[i] * ratio;
                acc = acc + t;
                if (lane == 0) B.cdfA[i + 1] = acc;
            }
            q = 1.0 - (double)pt; t = q;
            for (int s = 0; s < 6; ++s) t = t * t;
            const double ratio2 = (double)pt / q;
            acc = t;
            for (int i = 0; i < P.order; ++i) { t = t * coef[i] * ratio2; acc = acc + t; }
            niu = acc;
        }
        const double p_t_suc = 0.99 * niu, p_t_pro = 0.002 * __builtin_sqrt((1.0 - niu) / (double)P.nmax);
        if (lane == 0) {   // starting point: the single TEP {k-1} (pb_testing.py:109-110)
            PbEntry e0; e0.sum = L.w[63]; e0.pos = 63u | (1u << 24); B.fr[0] = e0;
            PbEntry m0; m0.sum = e0.sum; m0.pos = 0; B.cmin[0] = m0; B.smin[0] = m0;
        }
        wave_fence();
        int nused = 1, nlive = 1, ntep = P.nmax, bestidx = 0, stop = 0, cmp = 0, suc1 = 0, suc2 = 0;
        int tail_ck = 0, tail_ci = 0, tail_sk = 0, tail_si = 0;   // last chunk / super-chunk of the list and their minima
        float tail_cs = L.w[63], tail_ss = L.w[63];
        float best = tep_cost(L, 0.0f, S.d0);
        u64 bestD = S.d0, bestE = 0;
        const PbList FL{&B, spill, P.cmin_off};
        for (int j = 0; j < P.nmax - 1 && nlive > 0; ++j) {
            // first minimum of the list = arg-min on (sum, slot), read off the super-chunk minima
            const int nsuper = (nused + 4095) >> 12;
            float ms = __builtin_inff();
            int mi = 0x7FFFFFFF;
            if (lane < nsuper) { const PbEntry t = B.smin[lane]; ms = t.sum; mi = (int)t.pos; }
            argmin_si(ms, mi, lane);
            cmp += nlive == 1 ? 1 : 2;
            // Both levels of the list that this pop touches are loaded NOW, side by side: the 64 slots of the
            // popped slot's chunk (lane mi & 63 of it is the popped entry itself) and the 64 chunk minima of its
            // super-chunk.  Everything that changes below (the tombstone, children that land in the same chunk,
            // the new chunk minimum) is patched into these registers, so one round trip to the spilled part of the
            // list (global memory) is on the critical path of a TEP instead of three dependent ones.
            const int ck0 = mi >> 6, sk0 = ck0 >> 6;
            PbEntry mys, myc;
            mys.sum = myc.sum = __builtin_inff(); mys.pos = 0; myc.pos = 0x7FFFFFFFu;
            if (ck0 * 64 + lane < nused) mys = FL.slot(ck0 * 64 + lane);
            if ((sk0 * 64 + lane) * 64 < nused) myc = FL.cmin(sk0 * 64 + lane);
            PbEntry e;
            e.sum = ms;
            e.pos = (unsigned)__builtin_amdgcn_readlane((int)mys.pos, mi & 63);
            const int ew = (int)(e.pos >> 24);
            const int p0 = e.pos & 0xFF, pA = (e.pos >> 8) & 0xFF, pB = (e.pos >> 16) & 0xFF;
            const int last = ew == 1 ? p0 : (ew == 2 ? pA : pB);
            const int prev = ew == 2 ? p0 : pA;     // second largest (ew > 1)
            // children (wave-uniform): extended e U {63}, adjacent = largest index moved down by one
            PbEntry c1, c2;
            c1.sum = c2.sum = __builtin_inff(); c1.pos = c2.pos = 0;
            bool has1 = false, has2 = false;
            if (last < 63 && ew < P.order) {
                c1.pos = (e.pos & 0x00FFFFFFu) | (63u << (8 * ew)) | ((unsigned)(ew + 1) << 24);
                c1.sum = e.sum + L.w[63];
                has1 = true;
            }
            if (ew > 1) {
                if (last - prev > 1) {
                    c2.pos = (e.pos & ~(0xFFu << (8 * (ew - 1)))) | ((unsigned)(last - 1) << (8 * (ew - 1)));
                    const int q0 = c2.pos & 0xFF, q1 = (c2.pos >> 8) & 0xFF, q2 = (c2.pos >> 16) & 0xFF;
                    float sacc = L.w[q0] + L.w[q1];
                    if (ew > 2) sacc = sacc + L.w[q2];
                    c2.sum = sacc;
                    has2 = true;
                }
            } else if (last - 1 > -1) {
                c2.pos = (unsigned)(last - 1) | (1u << 24);
                c2.sum = L.w[last - 1];
                has2 = true;
            }
            if (has2 && !has1) { c1 = c2; has1 = true; has2 = false; }      // children in list order: c1 then c2
            const int s1 = nused, s2 = nused + 1;
            if (lane == 0) {
                PbEntry dead;
                dead.sum = __builtin_inff(); dead.pos = 0;
                FL.set_slot(mi, dead);
                if (has1) FL.set_slot(s1, c1);
                if (has2) FL.set_slot(s2, c2);
            }
            nused += (has1 ? 1 : 0) + (has2 ? 1 : 0);
            nlive += (has1 ? 1 : 0) + (has2 ? 1 : 0) - 1;
            // ---- chunk level: the popped slot's chunk from the patched registers; the tail chunk incrementally
            if (lane == (mi & 63)) mys.sum = __builtin_inff();
            if (has1 && (s1 >> 6) == ck0 && lane == (s1 & 63)) mys = c1;
            if (has2 && (s2 >> 6) == ck0 && lane == (s2 & 63)) mys = c2;
            float cs0 = mys.sum;
            int ci0 = ck0 * 64 + lane;
            argmin_si(cs0, ci0, lane);
            if (lane == 0) { PbEntry m; m.sum = cs0; m.pos = (unsigned)ci0; FL.set_cmin(ck0, m); }
            if (ck0 == tail_ck) { tail_cs = cs0; tail_ci = ci0; }
            // ---- super-chunk level, same scheme on the chunk minima (patched as the chunk level changes them)
            if (lane == (ck0 & 63)) { myc.sum = cs0; myc.pos = (unsigned)ci0; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool has = u == 0 ? has1 : has2;
                const int sl = u == 0 ? s1 : s2;
                const float csum = u == 0 ? c1.sum : c2.sum;
                if (!has) continue;
                const int ck = sl >> 6;
                if (ck != tail_ck) { tail_ck = ck; tail_cs = __builtin_inff(); tail_ci = 0x7FFFFFFF; }   // a new chunk starts
                if (ck == ck0) continue;                                   // covered by the reduction above
                if (csum < tail_cs) { tail_cs = csum; tail_ci = sl; }      // (a tie keeps the older, lower slot)
                if (lane == 0) { PbEntry m; m.sum = tail_cs; m.pos = (unsigned)tail_ci; FL.set_cmin(ck, m); }
                if ((ck >> 6) == sk0 && lane == (ck & 63)) { myc.sum = tail_cs; myc.pos = (unsigned)tail_ci; }
            }
            float ss0 = myc.sum;
            int si0 = (int)myc.pos;
            argmin_si(ss0, si0, lane);
            if (lane == 0) { PbEntry m; m.sum = ss0; m.pos = (unsigned)si0; B.smin[sk0] = m; }
            if (sk0 == tail_sk) { tail_ss = ss0; tail_si = si0; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool has = u == 0 ? has1 : has2;
                const int sl = u == 0 ? s1 : s2;
                const float csum = u == 0 ? c1.sum : c2.sum;
                if (!has) continue;
                const int sk = sl >> 12;
                if (sk != tail_sk) { tail_sk = sk; tail_ss = __builtin_inff(); tail_si = 0x7FFFFFFF; }
                if (sk == sk0) continue;
                if (csum < tail_ss) { tail_ss = csum; tail_si = sl; }
                if (lane == 0) { PbEntry m; m.sum = tail_ss; m.pos = (unsigned)tail_si; B.smin[sk] = m; }
            }
            wave_fence();
            // promising-probability rule
            const float rs = e.sum;
            const float w1 = det_expf(P.c4 * rs) * spl, w2 = 1.0f - w1;
            const float bt = __builtin_floorf((best - rs) / lrb_mean);
            const int beta = bt > 0.0f ? (bt < 64.0f ? (int)bt : 64) : 0;
            float bs = 0.0f;
            bs = bs + w1 * (float)B.cdfA[beta];
            bs = bs + w2 * (float)B.cdfH[beta];
            if ((double)bs < p_t_pro) { stop = 1; ntep = j + 1; break; }
            u64 D = S.d0 ^ L.P[p0], E = 1ull << p0;
            if (ew > 1) { D ^= L.P[pA]; E |= 1ull << pA; }
            if (ew > 2) { D ^= L.P[pB]; E |= 1ull << pB; }
            const float cost = tep_cost(L, rs, D);
            ++suc1;
            if (cost < best) {
                best = cost; bestD = D; bestE = E; bestidx = j + 1;
                const float ratio = (1.0f - w1) / w1;
                float prod = 1.0f;
#pragma unroll 4
                for (int p = 0; p < 64; ++p) {
                    const float qp = B.q[64 + p];
                    prod = prod * (((D >> p) & 1) ? 2.0f * qp : 2.0f * (1.0f - qp));
                }
                const float p_suc = 1.0f / (1.0f + ratio / prod);
                ++suc2;
                if ((double)p_suc > p_t_suc) { stop = 2; ntep = j + 1; break; }
            }
        }
        search_finish(L, S, bestE, bestD, f, lane, cw_out);
        if (lane == 0) {
            if (metric_out) metric_out[f] = best;
            if (best_out) best_out[f] = bestidx;
            if (ntep_out) ntep_out[f] = ntep;
            if (aux_out) { aux_out[f * 4] = cmp; aux_out[f * 4 + 1] = suc1; aux_out[f * 4 + 2] = suc2; aux_out[f * 4 + 3] = stop; }
        }
        wave_fence();
    }
}

__global__ __launch_bounds__(256) void osd_counts_kernel(const u64 *__restrict__ cw, const u64 *__restrict__ label,
                                                         const int *__restrict__ index, const int *__restrict__ count,
                                                         const int *__restrict__ ntep, long long F,
                                                         u64 *__restrict__ counts)
{
    __shared__ u64 part[4][3];
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }
    u64 n = 0, wrong = 0, teps = 0;
    for (long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x; f < nframes; f += (long long)gridDim.x * blockDim.x) {
        const long long src = index ? index[f] : f;
        n += 1;
        wrong += (cw[f * 2] != label[src * 2]) || (cw[f * 2 + 1] != label[src * 2 + 1]);
        teps += ntep ? (u64)ntep[f] : 0ull;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_down(n, off, 64); wrong += __shfl_down(wrong, off, 64); teps += __shfl_down(teps, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[wave][0] = n; part[wave][1] = wrong; part[wave][2] = teps; }
    __syncthreads();
    if (threadIdx.x < 3)
        atomicAdd(&counts[threadIdx.x], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------
// context pieces: G columns, TEP table (order <= 3), front-end workspace
// ---------------------------------------------------------------------------------------
struct OsdState {
    int64_t ntep[4] = {0, 0, 0, 0};
    uchar4 *d_tep_fs = nullptr;       // FS visit order, weight classes 1..3 back to back
    int *d_base2 = nullptr;           // order-2 ranks: number of index pairs with a larger sum
    double *d_cdf_half = nullptr;     // PB-OSD: P[Bin(64, 1/2) <= b], b = 0..64
    double *d_coef = nullptr;         // PB-OSD: (64-i)/(i+1)
    void *d_pb_spill = nullptr;       // PB-OSD frontier overflow [waves][stride]
    int *d_pb_queue = nullptr;        // PB-OSD frame hand-out counter
    int64_t pb_spill_stride = 0;
    int fs_off[4] = {0, 0, 0, 0}, fs_cnt[4] = {0, 0, 0, 0};
    unsigned char *d_perm = nullptr;  // workspace [cap][128]
    u64 *d_parity = nullptr;          // workspace [cap][64]
    int64_t cap = 0;
};

static OsdState *state(ldpc_ctx *ctx) { return reinterpret_cast<OsdState *>(ctx->osd_state); }

int osd_ctx_init(ldpc_ctx *ctx)
{
    const ldpc_code &c = ctx->code;
    ctx->osd_ok = false;
    if (c.n != kOsdN || c.k != kOsdK) return LDPC_OK;  // OSD entry points will report UNSUPPORTED
    std::vector<u64> cols(kOsdN, 0);
    for (int r = 0; r < kOsdK; ++r)
        for (int v = 0; v < kOsdN; ++v)
            if (c.G[(size_t)r * kOsdN + v]) cols[v] |= 1ull << r;
    LDPC_HIP(hipMalloc((void **)&ctx->d_Gcols, sizeof(u64) * kOsdN));
    LDPC_HIP(hipMemcpy(ctx->d_Gcols, cols.data(), sizeof(u64) * kOsdN, hipMemcpyHostToDevice));
    // one table for order 3; orders 0..2 are its prefixes (weight classes are concatenated)
    OsdState *st = new OsdState();
    ctx->osd_state = st;
    int64_t bounds[4];
    const int64_t total = tep_table(kOsdK, 3, nullptr, bounds);
    std::vector<uint8_t> sup((size_t)total * 3), packed((size_t)total * 4);
    tep_table(kOsdK, 3, sup.data(), nullptr);
    for (int64_t t = 0; t < total; ++t) {
        int w = 0;
        for (int q = 0; q < 3; ++q) { packed[4 * t + q] = sup[3 * t + q] == 0xFF ? 0 : sup[3 * t + q]; w += sup[3 * t + q] != 0xFF; }
        packed[4 * t + 3] = (uint8_t)w;
    }
    for (int o = 0; o < 4; ++o) st->ntep[o] = bounds[o];
    LDPC_HIP(hipMalloc((void **)&ctx->d_tep, packed.size()));
    LDPC_HIP(hipMemcpy(ctx->d_tep, packed.data(), packed.size(), hipMemcpyHostToDevice));
    // FS-OSD visit order (generate_sequential_teps, fs_testing.py:32-49), supports stored ascending
    std::vector<uint8_t> fs;
    int off = 0;
    for (int w = 1; w <= 3; ++w) {
        const int64_t cnt = tep_table_fs(kOsdK, w, nullptr);
        std::vector<uint8_t> sup3((size_t)cnt * 3);
        tep_table_fs(kOsdK, w, sup3.data());
        st->fs_off[w] = off; st->fs_cnt[w] = (int)cnt;
        for (int64_t t = 0; t < cnt; ++t) {
            for (int q = 0; q < 3; ++q) fs.push_back(sup3[3 * t + q] == 0xFF ? 0 : sup3[3 * t + q]);
            fs.push_back((uint8_t)w);
        }
        off += (int)cnt;
    }
    LDPC_HIP(hipMalloc((void **)&st->d_tep_fs, fs.size()));
    LDPC_HIP(hipMemcpy(st->d_tep_fs, fs.data(), fs.size(), hipMemcpyHostToDevice));
    {
        int base2[127];
        auto npairs = [](int t) { return (t - 1) / 2 - (t > 63 ? t - 63 : 0) + 1; };
        for (int sidx = 0; sidx < 127; ++sidx) {
            int acc = 0;
            for (int t = sidx + 1; t <= 125; ++t) acc += npairs(t);
            base2[sidx] = acc;
        }
        LDPC_HIP(hipMalloc((void **)&st->d_base2, sizeof(base2)));
        LDPC_HIP(hipMemcpy(st->d_base2, base2, sizeof(base2), hipMemcpyHostToDevice));
    }
    // PB-OSD constants, same float64 recurrence as the kernel / oracle
    {
        double coef[64], cdf[65], t = 0.5;
        for (int i = 0; i < 64; ++i) coef[i] = (double)(64 - i) / (double)(i + 1);
        for (int q = 0; q < 6; ++q) t = t * t;
        double acc = t;
        cdf[0] = acc;
        for (int i = 0; i < 64; ++i) { t = t * coef[i] * (0.5 / 0.5); acc = acc + t; cdf[i + 1] = acc; }
        LDPC_HIP(hipMalloc((void **)&st->d_cdf_half, sizeof(cdf)));
        LDPC_HIP(hipMemcpy(st->d_cdf_half, cdf, sizeof(cdf), hipMemcpyHostToDevice));
        LDPC_HIP(hipMalloc((void **)&st->d_coef, sizeof(coef)));
        LDPC_HIP(hipMalloc((void **)&st->d_pb_queue, sizeof(int)));
        LDPC_HIP(hipMemcpy(st->d_coef, coef, sizeof(coef), hipMemcpyHostToDevice));
    }
    ctx->osd_ok = true;
    return LDPC_OK;
}

void osd_ctx_release(ldpc_ctx *ctx)
{
    (void)hipFree(ctx->d_Gcols);
    (void)hipFree(ctx->d_tep);
    if (OsdState *st = state(ctx)) {
        (void)hipFree(st->d_perm);
        (void)hipFree(st->d_parity);
        (void)hipFree(st->d_tep_fs);
        (void)hipFree(st->d_base2);
        (void)hipFree(st->d_cdf_half);
        (void)hipFree(st->d_coef);
        (void)hipFree(st->d_pb_spill);
        (void)hipFree(st->d_pb_queue);
        delete st;
    }
    ctx->osd_state = nullptr;
}

static int reserve(ldpc_ctx *ctx, int64_t frames)
{
    OsdState *st = state(ctx);
    if (frames <= st->cap) return LDPC_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)cs;
    (void)hipFree(st->d_perm); (void)hipFree(st->d_parity);
    st->d_perm = nullptr; st->d_parity = nullptr; st->cap = 0;
    if (hipMalloc((void **)&st->d_perm, (size_t)frames * 128) != hipSuccess ||
        hipMalloc((void **)&st->d_parity, (size_t)frames * 64 * sizeof(u64)) != hipSuccess)
        return fail(LDPC_E_NOMEM, "OSD workspace for %lld frames could not be allocated", (long long)frames);
    st->cap = frames;
    return LDPC_OK;
}

static unsigned osd_grid(int64_t F)
{
    // persistent grid: enough blocks to fill 256 CUs a few times over, frames are strided over waves
    int64_t want = (F + 3) / 4;
    return (unsigned)(want < 1 ? 1 : (want < 4096 ? want : 4096));
}


}  // namespace ldpc

using namespace ldpc;

extern "C" {

int ldpc_osd_reserve(ldpc_ctx *ctx, int64_t max_frames)
{
    if (!ctx || max_frames < 0) return fail(LDPC_E_ARG, "ldpc_osd_reserve: bad arguments");
    if (!ctx->osd_ok) return fail(LDPC_E_UNSUPPORTED, "OSD kernels need an (n=128, k=64) code; this one is (%d,%d)", ctx->code.n, ctx->code.k);
    return reserve(ctx, max_frames);
}

int ldpc_osd_ge(ldpc_ctx *ctx, const uint64_t *d_rows_in, int64_t F, uint64_t *d_rows_out, uint8_t *d_swaps,
                int32_t *d_nswaps, void *stream)
{
    if (!ctx || F < 0 || (F > 0 && (!d_rows_in || !d_rows_out))) return fail(LDPC_E_ARG, "ldpc_osd_ge: bad arguments");
    if (!ctx->osd_ok) return fail(LDPC_E_UNSUPPORTED, "OSD kernels need an (n=128, k=64) code; this one is (%d,%d)", ctx->code.n, ctx->code.k);
    if (F == 0) return LDPC_OK;
    hipLaunchKernelGGL(osd_ge_kernel, dim3(osd_grid(F)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const u64 *>(d_rows_in), (long long)F, reinterpret_cast<u64 *>(d_rows_out), d_swaps,
                       d_nswaps);
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

int ldpc_osd_front(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                   uint8_t *d_perm, uint64_t *d_parity, int32_t *d_nswaps, void *stream)
{
    if (!ctx || F < 0 || (F > 0 && (!d_y || !d_perm || !d_parity))) return fail(LDPC_E_ARG, "ldpc_osd_front: bad arguments");
    if (!ctx->osd_ok) return fail(LDPC_E_UNSUPPORTED, "OSD kernels need an (n=128, k=64) code; this one is (%d,%d)", ctx->code.n, ctx->code.k);
    if (F == 0) return LDPC_OK;
    hipLaunchKernelGGL(osd_front_kernel, dim3((unsigned)(F < 65536 ? F : 65536)), dim3(64), 0, (hipStream_t)stream, d_y, d_index, d_count,
                       (long long)F, reinterpret_cast<const u64 *>(ctx->d_Gcols), d_perm, reinterpret_cast<u64 *>(d_parity),
                       d_nswaps);
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

static int check_params(ldpc_ctx *ctx, const ldpc_osd_params *p, const char *who)
{
    if (!ctx->osd_ok) return fail(LDPC_E_UNSUPPORTED, "OSD kernels need an (n=128, k=64) code; this one is (%d,%d)", ctx->code.n, ctx->code.k);
    if (p->order < 0 || p->order > 3) return fail(LDPC_E_ARG, "%s: order %d outside 0..3", who, p->order);
    if (p->algo != LDPC_OSD_CONVENTIONAL && p->algo != LDPC_OSD_FS && p->algo != LDPC_OSD_PB)
        return fail(LDPC_E_ARG, "%s: unknown search algorithm %d", who, p->algo);
    if (p->algo == LDPC_OSD_PB && p->order < 1) return fail(LDPC_E_ARG, "%s: PB-OSD needs order >= 1", who);
    return LDPC_OK;
}

// launches the search kernel selected by p->algo on front-end results (d_perm, d_parity)
static int launch_search(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                         const unsigned char *d_perm, const u64 *d_parity, const ldpc_osd_params *p, uint64_t *d_cw,
                         float *d_metric, int32_t *d_best, int32_t *d_ntep, hipStream_t s)
{
    OsdState *st = state(ctx);
    if (p->algo == LDPC_OSD_PB) {
        const int64_t nmax = st->ntep[p->order];
        const unsigned blocks = osd_grid(F) < 512 ? osd_grid(F) : 512;       // bounded: each wave owns a spill area
        // the list is append-only: at most 1 + 2 (N_max - 1) slots; spilled slots first, spilled chunk minima after
        const int64_t slots = 2 * nmax + 2;
        if (slots > (int64_t)kPbSuper * 4096) return fail(LDPC_E_UNSUPPORTED, "ldpc_osd_decode: PB-OSD list of %lld slots exceeds the kernel's limit", (long long)slots);
        const int64_t spill_slots = slots > kPbLdsSlots ? slots - kPbLdsSlots : 0;
        const int64_t stride = spill_slots + (slots / 64 + 2) + 2;
        if (stride > st->pb_spill_stride) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
                return fail(LDPC_E_NOMEM, "ldpc_osd_decode: PB-OSD frontier workspace must be sized before capturing (run one call first)");
            (void)hipFree(st->d_pb_spill);
            st->d_pb_spill = nullptr; st->pb_spill_stride = 0;
            if (hipMalloc(&st->d_pb_spill, sizeof(PbEntry) * (size_t)stride * 512 * 4) != hipSuccess)
                return fail(LDPC_E_NOMEM, "ldpc_osd_decode: PB-OSD frontier workspace (%lld entries per wave) could not be allocated", (long long)stride);
            st->pb_spill_stride = stride;
        }
        PbParams pp;
        pp.order = p->order; pp.nmax = (int)nmax; pp.cmin_off = spill_slots;
        pp.c4 = (float)(-4.0 * (1.0 / pow(10.0, (double)p->snr_db / 10.0)));    // -4 * noise_variance, pb_testing.py:50-52
        LDPC_HIP(hipMemsetAsync(st->d_pb_queue, 0, sizeof(int), s));
        hipLaunchKernelGGL(osd_pb_kernel, dim3(blocks), dim3(256), 0, s, d_y, d_index, d_count, (long long)F, d_perm,
                           d_parity, pp, st->d_cdf_half, st->d_coef, reinterpret_cast<PbEntry *>(st->d_pb_spill),
                           (long long)st->pb_spill_stride, st->d_pb_queue, reinterpret_cast<u64 *>(d_cw), d_metric, d_best, d_ntep,
                           reinterpret_cast<int *>(p->d_aux));
    } else if (p->algo == LDPC_OSD_FS) {
        FsParams fp;
        fp.order = p->order; fp.quirk = p->fs_reference_quirk != 0;
        fp.beta_term = (float)((double)p->fs_beta * (double)(kOsdN - kOsdK));   // fs_testing.py:138
        fp.tau_e = p->fs_tau_e; fp.tau_psc = p->fs_tau_psc;
        for (int w = 0; w < 4; ++w) { fp.cls_off[w] = st->fs_off[w]; fp.cls_cnt[w] = st->fs_cnt[w]; }
        hipLaunchKernelGGL(osd_fs_kernel, dim3((unsigned)(F < 65536 ? F : 65536)), dim3(64), 0, s, d_y, d_index, d_count, (long long)F,
                           d_perm, d_parity, st->d_tep_fs, fp, reinterpret_cast<u64 *>(d_cw), d_metric, d_best,
                           d_ntep);
    } else if (p->order == 2 && !p->reserved) {
        hipLaunchKernelGGL(osd_search2_kernel, dim3((unsigned)(F < 65536 ? F : 65536)), dim3(64), 0, s, d_y, d_index, d_count, (long long)F,
                           d_perm, d_parity, st->d_base2, reinterpret_cast<u64 *>(d_cw), d_metric, d_best, d_ntep);
    } else {   // table-driven scan: any order (and order 2 when params->reserved = 1, the cross-check path)
        if (p->order >= 2)
            hipLaunchKernelGGL(osd_search_kernel<1>, dim3((unsigned)(F < 65536 ? F : 65536)), dim3(64), 0, s, d_y, d_index, d_count, (long long)F,
                               d_perm, d_parity, reinterpret_cast<const uchar4 *>(ctx->d_tep), (int)st->ntep[p->order],
                               reinterpret_cast<u64 *>(d_cw), d_metric, d_best, d_ntep);
        else
            hipLaunchKernelGGL(osd_search_kernel<4>, dim3(osd_grid(F)), dim3(256), 0, s, d_y, d_index, d_count, (long long)F,
                               d_perm, d_parity, reinterpret_cast<const uchar4 *>(ctx->d_tep), (int)st->ntep[p->order],
                               reinterpret_cast<u64 *>(d_cw), d_metric, d_best, d_ntep);
    }
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

int ldpc_osd_search(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                    const uint8_t *d_perm, const uint64_t *d_parity, const ldpc_osd_params *p, uint64_t *d_cw,
                    float *d_metric, int32_t *d_best, int32_t *d_ntep, void *stream)
{
    if (!ctx || !p || F < 0 || (F > 0 && (!d_y || !d_cw || !d_perm || !d_parity)))
        return fail(LDPC_E_ARG, "ldpc_osd_search: bad arguments");
    int rc = check_params(ctx, p, "ldpc_osd_search");
    if (rc) return rc;
    if (F == 0) return LDPC_OK;
    return launch_search(ctx, d_y, d_index, d_count, F, d_perm, reinterpret_cast<const u64 *>(d_parity), p, d_cw, d_metric,
                         d_best, d_ntep, (hipStream_t)stream);
}

int ldpc_osd_decode(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                    const ldpc_osd_params *p, uint64_t *d_cw, float *d_metric, int32_t *d_best, int32_t *d_ntep,
                    void *stream)
{
    if (!ctx || !p || F < 0 || (F > 0 && (!d_y || !d_cw))) return fail(LDPC_E_ARG, "ldpc_osd_decode: bad arguments");
    int rc = check_params(ctx, p, "ldpc_osd_decode");
    if (rc) return rc;
    if (F == 0) return LDPC_OK;
    OsdState *st = state(ctx);
    if (F > st->cap) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(LDPC_E_NOMEM, "ldpc_osd_decode: workspace holds %lld frames, %lld needed; call ldpc_osd_reserve before capturing", (long long)st->cap, (long long)F);
        rc = reserve(ctx, F);
        if (rc) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(osd_front_kernel, dim3((unsigned)(F < 65536 ? F : 65536)), dim3(64), 0, s, d_y, d_index, d_count, (long long)F,
                       reinterpret_cast<const u64 *>(ctx->d_Gcols), st->d_perm, st->d_parity, (int *)nullptr);
    return launch_search(ctx, d_y, d_index, d_count, F, st->d_perm, st->d_parity, p, d_cw, d_metric, d_best, d_ntep, s);
}

int ldpc_osd_counts(ldpc_ctx *ctx, const uint64_t *d_cw, const uint64_t *d_label_bits, const int32_t *d_index,
                    const int32_t *d_count, const int32_t *d_ntep, int64_t F, int64_t *d_counts, void *stream)
{
    if (!ctx || F < 0 || !d_counts || (F > 0 && (!d_cw || !d_label_bits))) return fail(LDPC_E_ARG, "ldpc_osd_counts: bad arguments");
    if (F == 0) return LDPC_OK;
    int64_t g = (F + 1023) / 1024;
    hipLaunchKernelGGL(osd_counts_kernel, dim3((unsigned)(g < 128 ? g : 128)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const u64 *>(d_cw), reinterpret_cast<const u64 *>(d_label_bits), d_index, d_count,
                       d_ntep, (long long)F, reinterpret_cast<u64 *>(d_counts));
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

}  // extern "C"
