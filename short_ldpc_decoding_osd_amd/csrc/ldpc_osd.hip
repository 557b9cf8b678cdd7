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
//   osd_search2r_kernel conventional order 2 (the headline configuration): register-resident, rotation pairing by
//                       wave_rol DPP, two-stage scan with an exact prefix early exit and survivor compaction,
//                       software-pipelined LUT reads, prefetched inputs, success counters on request.
//   osd_search2_kernel  the round-1 form of the same scan (v_readlane pairing): fallback and cross-check route.
//   osd_search_kernel   conventional order-p search over the reference's TEP table: per frame a
//                       byte-indexed LUT of partial |y'| sums in LDS (8 x 256 floats), each lane
//                       evaluates one TEP per round: parity word = d0 ^ P'[i] ^ P'[j] ..., metric
//                       = flipped-MRB weights + 8 LUT terms in a FIXED order (the canonical order
//                       the oracle uses, see oracle/np_oracle.py weighted_distance), first minimum.
//   osd_fs_kernel       FS-OSD (FS_OSD/fs_testing.py:22-64,129-161), 64 TEPs per round.
//   osd_ge_kernel       full_gf2elim on caller-supplied matrices; osd_counts_kernel: success counters.
//   (PB-OSD: ldpc_osd_pb.hip; shared per-frame set-up: ldpc_search.h; host state: ldpc_osd_state.h)
#include <math.h>
#include <stdlib.h>

#include "ldpc_internal.h"
#include "ldpc_wave.h"
#include "ldpc_search.h"
#include "ldpc_front.h"
#include "ldpc_osd_state.h"

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
// OSD front end (the per-frame device code: ldpc_front.h)
// ---------------------------------------------------------------------------------------
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
// Order-2 scan, second form (the one the launcher uses when the wave_rol probe succeeded):
//  * pairing by ROTATION: in round r = 1..32 lane l meets lane (l -+ r) mod 64, whose row of P' and weight
//    arrive by three `wave_rol:1` DPP moves of a rotating copy -- no v_readlane (8 issue cycles each, six per
//    round in the triangular pairing above) and no selects; rounds 1..31 cover every unordered pair once per
//    lane, round 32 pairs l with l +- 32 and only lanes < 32 count;
//  * persistent workgroups (one wavefront each) stride over the frames, and the global inputs of frame
//    f + 2 grid (perm, P' row) and f + grid (channel values, addressed through its perm) are in flight while
//    frame f is scanned, so the ~2 us of dependent global latency of the prologue is hidden;
//  * everything else (two-stage scan with the exact prefix bound, survivor ring, rank-ordered ties) as above.
// ---------------------------------------------------------------------------------------
// LDS of the rotation scan: the byte LUT and the survivor ring, exactly 10 KiB -> 16 workgroups = 4 wavefronts per SIMD
// (the scan is occupancy-sensitive: 1.75x the time at half the residency).  The parity weights are only needed while the
// LUT is built and the codeword words only after the last survivor batch, so both borrow the ring's memory.
struct __attribute__((aligned(16))) Search2rLds {
    float lut[8][256];   // lut[b][v] = sum of |y'[64+8b+t]| over the set bits t of v, ascending t
    uint4 q[128];        // survivors: D.lo, D.hi, prefix metric bits, r * 64 + lane
    __device__ __forceinline__ float *wpar() { return reinterpret_cast<float *>(q); }           // [64], before the scan
    __device__ __forceinline__ u64 *cw() { return reinterpret_cast<u64 *>(q) + 32; }            // [2], after the scan
};
static_assert(sizeof(Search2rLds) == 10240, "LDS budget of the rotation scan");

__device__ __forceinline__ float cost2r(const Search2rLds &L, float mrb, u64 D)
{
    float acc = mrb;
    acc = acc + lut_byte<0>(L.lut, D); acc = acc + lut_byte<1>(L.lut, D); acc = acc + lut_byte<2>(L.lut, D); acc = acc + lut_byte<3>(L.lut, D);
    acc = acc + lut_byte<4>(L.lut, D); acc = acc + lut_byte<5>(L.lut, D); acc = acc + lut_byte<6>(L.lut, D); acc = acc + lut_byte<7>(L.lut, D);
    return acc;
}

__device__ __forceinline__ int wave_rot1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x134, 0xF, 0xF, true); }

__device__ __forceinline__ void search2r_finish_batch(const Search2rLds &L, uint4 e, bool valid, int dir, const int *__restrict__ base2,
                                                      float &best, int &bi, int &bj, u64 &bestD)
{
    if (!valid) return;
    const u64 D = ((u64)e.y << 32) | e.x;
    float acc = __uint_as_float(e.z);
    acc = acc + lut_byte<2>(L.lut, D); acc = acc + lut_byte<3>(L.lut, D); acc = acc + lut_byte<4>(L.lut, D);
    acc = acc + lut_byte<5>(L.lut, D); acc = acc + lut_byte<6>(L.lut, D); acc = acc + lut_byte<7>(L.lut, D);
    if (!(acc <= best)) return;
    const int r = (int)(e.w >> 6), l = (int)(e.w & 63), m = (l - dir * r) & 63;
    const int ci = l < m ? l : m, cj = l < m ? m : l;
    // equal metrics are ordered by table rank (practically never taken)
    if (acc < best || tep2_rank(ci, cj, base2) < tep2_rank(bi, bj, base2)) { best = acc; bi = ci; bj = cj; bestD = D; }
}

__device__ __forceinline__ void search2r_device(Search2rLds &LL, const SearchFrame &S, u64 Pl, float wl, int dir,
                                                const int *__restrict__ base2, int lane, float &best_out, int &rank_out,
                                                u64 &D_out, u64 &E_out)
{
    Search2rLds &L = LL;
    // order 0 (rank 0, identical in every lane), then order 1: lane l owns TEP {l}
    float best = cost2r(L, 0.0f, S.d0);
    int bi = -1, bj = -1;
    u64 bestD = S.d0;
    const u64 dP = S.d0 ^ Pl;
    {
        const float c = cost2r(L, wl, dP);
        if (c < best) { best = c; bj = lane; bestD = dP; }      // a tie keeps the lower rank (order 0)
    }
    float bound = wave_min_f32(best);
    int qhead = 0, qn = 0;   // ring state (wave-uniform)
    int plo = (int)(unsigned)Pl, phi = (int)(unsigned)(Pl >> 32), wr = __float_as_int(wl);
    // software pipeline: the two LUT reads of round r + 1 are issued before round r is finished, so their LDS latency
    // (bank conflicts included) overlaps the survivor bookkeeping -- at 3.5 wavefronts per SIMD nothing else hides it
    plo = wave_rot1(plo); phi = wave_rot1(phi); wr = wave_rot1(wr);
    u64 D = dP ^ (((u64)(unsigned)phi << 32) | (unsigned)plo);
    float m = wl + __int_as_float(wr), t0 = lut_byte<0>(L.lut, D), t1 = lut_byte<1>(L.lut, D);
    for (int r = 1; r <= 32; ++r) {
        plo = wave_rot1(plo); phi = wave_rot1(phi); wr = wave_rot1(wr);          // (round 33 is computed and never used)
        const u64 Dn = dP ^ (((u64)(unsigned)phi << 32) | (unsigned)plo);
        const float mn = wl + __int_as_float(wr), u0 = lut_byte<0>(L.lut, Dn), u1 = lut_byte<1>(L.lut, Dn);
        float acc = m + t0;                                                       // |y'_i| + |y'_j| (commutative), then byte 0
        acc = acc + t1;
        const bool keep = (r < 32 || lane < 32) && !(acc > bound);
        const u64 km = __ballot(keep);
        if (km) {
            if (keep) {
                const int slot = (qhead + qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0))) & 127;
                LL.q[slot] = make_uint4((unsigned)D, (unsigned)(D >> 32), __float_as_uint(acc), (unsigned)(r * 64 + lane));
            }
            qn += __popcll(km);
            if (qn >= 64) {
                wave_fence();
                search2r_finish_batch(L, LL.q[(qhead + lane) & 127], true, dir, base2, best, bi, bj, bestD);
                qhead = (qhead + 64) & 127;
                qn -= 64;
                bound = wave_min_f32(best);
                wave_fence();
            }
        }
        D = Dn; m = mn; t0 = u0; t1 = u1;
    }
    wave_fence();
    search2r_finish_batch(L, LL.q[(qhead + lane) & 127], lane < qn, dir, base2, best, bi, bj, bestD);
    wave_fence();
    int bestt = tep2_rank(bi, bj, base2);
    u64 bestE = (bi >= 0 ? 1ull << bi : 0ull) | (bj >= 0 ? 1ull << bj : 0ull);
    wave_argmin(best, bestt, bestD, bestE, lane);
    best_out = best; rank_out = bestt; D_out = bestD; E_out = bestE;
}

// Frame assignment is static (frame = block + k grid).  Dynamic hand-out was measured and dropped: a device-scope
// ticket word saturates at ~88 fetch-adds per us (MI355X_MICROARCH.md, "dequeue") and a returning atomic takes
// microseconds under load -- every frame through ONE ticket word: 551 us; through 16 words on their own 128-byte
// lines, result awaited at once: 213 us; the last 40 % of the frames through 16 words, drawn a whole scan before
// they are looked at: 129 us; static: 102 us (the wavefronts are then alive for ~63 % of the launch: the scan time
// varies with the number of survivors) -- so the balance comes from the hardware dispatcher instead: the grid is 6x
// the resident wavefronts (see the launcher).
__global__ __launch_bounds__(64) void osd_search2r_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                          const int *__restrict__ count, long long F,
                                                          const unsigned char *__restrict__ perm_in,
                                                          const u64 *__restrict__ parity_in, int dir,
                                                          const int *__restrict__ base2,
                                                          u64 *__restrict__ cw_out, float *__restrict__ metric_out,
                                                          int *__restrict__ best_out, int *__restrict__ ntep_out,
                                                          const u64 *__restrict__ label, u64 *__restrict__ counts)
{
    __shared__ Search2rLds LL;
    const int lane = threadIdx.x;
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }
    // the OSD success counters ride along when the caller wants them (ldpc_pipeline_run): {frames, wrong, TEPs};
    // frames and TEPs are known up front, a wrong codeword costs one fire-and-forget atomic (~6 % of the frames)
    if (counts && blockIdx.x == 0 && lane == 0) { atomicAdd(&counts[0], (u64)nframes); atomicAdd(&counts[2], (u64)nframes * 2081ull); }
    // software pipeline over the frames of this workgroup: (o, P) two frames ahead, y one frame ahead
    const long long G = gridDim.x;
    long long f0 = blockIdx.x, f1 = f0 + G, f2 = f1 + G;
    int o1a = 0, o2a = 0, o1b = 0, o2b = 0;
    u64 Pa = 0, Pb = 0;
    long long srca = 0, srcb = 0;
    float y1a = 0.0f, y2a = 0.0f;
    if (f0 < nframes) {
        o1a = perm_in[f0 * 128 + lane]; o2a = perm_in[f0 * 128 + 64 + lane]; Pa = parity_in[f0 * 64 + lane];
        srca = index ? index[f0] : f0;
    }
    if (f1 < nframes) {
        o1b = perm_in[f1 * 128 + lane]; o2b = perm_in[f1 * 128 + 64 + lane]; Pb = parity_in[f1 * 64 + lane];
        srcb = index ? index[f1] : f1;
    }
    u64 laba = 0;
    if (f0 < nframes) { y1a = y[srca * 128 + o1a]; y2a = y[srca * 128 + o2a]; if (label && lane < 2) laba = label[srca * 2 + lane]; }
    while (f0 < nframes) {
        // issue the loads of the frames ahead (they are consumed one / two trips later)
        float y1b = 0.0f, y2b = 0.0f;
        u64 labb = 0;
        if (f1 < nframes) { y1b = y[srcb * 128 + o1b]; y2b = y[srcb * 128 + o2b]; if (label && lane < 2) labb = label[srcb * 2 + lane]; }
        int o1c = 0, o2c = 0;
        u64 Pc = 0;
        long long srcc = 0;
        if (f2 < nframes) {
            o1c = perm_in[f2 * 128 + lane]; o2c = perm_in[f2 * 128 + 64 + lane]; Pc = parity_in[f2 * 64 + lane];
            srcc = index ? index[f2] : f2;
        }
        // ---- frame f0
        SearchFrame S;
        S.o1 = o1a; S.o2 = o2a;
        const float w1 = __builtin_fabsf(y1a), w2 = __builtin_fabsf(y2a);
        LL.wpar()[lane] = w2;
        S.hm = __ballot(!(y1a > 0.0f));
        S.hp = __ballot(!(y2a > 0.0f));
        wave_fence();
        build_byte_luts<8>(LL.lut, LL.wpar(), lane);
        S.d0 = wave_xor64(((S.hm >> lane) & 1) ? Pa : 0ull) ^ S.hp;
        wave_fence();
        float best; int bestt; u64 bestD, bestE;
        search2r_device(LL, S, Pa, w1, dir, base2, lane, best, bestt, bestD, bestE);
        {   // search_finish, with the codeword words still in hand for the success test (convention_osd.py:65-66)
            const u64 mrb_bits = S.hm ^ bestE, par_bits = bestD ^ S.hp;
            u64 *const cw = LL.cw();
            if (lane < 2) cw[lane] = 0;
            wave_fence();
            if ((mrb_bits >> lane) & 1) atomicOr(&cw[S.o1 >> 6], 1ull << (S.o1 & 63));
            if ((par_bits >> lane) & 1) atomicOr(&cw[S.o2 >> 6], 1ull << (S.o2 & 63));
            wave_fence();
            const u64 word = lane < 2 ? cw[lane] : 0ull;
            if (lane < 2) cw_out[f0 * 2 + lane] = word;
            if (counts && __ballot(lane < 2 && word != laba) && lane == 0) atomicAdd(&counts[1], 1ull);
            wave_fence();
        }
        if (lane == 0) {
            if (metric_out) metric_out[f0] = best;
            if (best_out) best_out[f0] = bestt;
            if (ntep_out) ntep_out[f0] = 2081;
        }
        // ---- rotate the pipeline
        f0 = f1; f1 = f2; f2 += G;
        o1a = o1b; o2a = o2b; Pa = Pb; srca = srcb; y1a = y1b; y2a = y2b; laba = labb;
        o1b = o1c; o2b = o2c; Pb = Pc; srcb = srcc;
    }
}

// ---------------------------------------------------------------------------------------
// ldpc_osd_decode / ldpc_pipeline_run for the conventional order-2 OSD when the caller does not ask for the front-end
// results: front end AND scan of a frame in ONE wavefront, back to back -- the permutation, the rows of P' and the primed
// channel values pass from one to the other in registers and LDS and never touch memory.  Two launches moved 640 B of
// workspace per frame out and in again and read y twice (PMC, round 3: 41 + 49 MB per 33.5 k frames against 18 MB
// algorithmic: 5.1x); this form reads 512 B and writes 24 B per frame.  The front end's 3.6 KiB of LDS lie inside the
// scan's LUT area (built afterwards), the frame's y row in its survivor ring (filled afterwards): 10 KiB, 16 wavefronts per CU.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void osd_fused2r_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                         const int *__restrict__ count, long long F,
                                                         const u64 *__restrict__ Gcols, int dir, const int *__restrict__ base2,
                                                         u64 *__restrict__ cw_out, float *__restrict__ metric_out,
                                                         int *__restrict__ best_out, int *__restrict__ ntep_out,
                                                         const u64 *__restrict__ label, u64 *__restrict__ counts)
{
    __shared__ Search2rLds LL;
    static_assert(sizeof(FrontLds) <= sizeof(LL.lut), "the front end works inside the LUT area");
    FrontLds &LF = *reinterpret_cast<FrontLds *>(LL.lut);
    float *const yrow = reinterpret_cast<float *>(LL.q) + 256;      // bytes 1024 .. 1535 of the ring (wpar: 0 .. 255, cw: 512 .. 527)
    const int lane = threadIdx.x;
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }
    if (counts && blockIdx.x == 0 && lane == 0) { atomicAdd(&counts[0], (u64)nframes); atomicAdd(&counts[2], (u64)nframes * 2081ull); }
    // software pipeline over the frames of this workgroup: the frame number two frames ahead, the y row one frame ahead
    const long long G = gridDim.x;
    long long f0 = blockIdx.x, f1 = f0 + G, f2 = f1 + G;
    long long srca = 0, srcb = 0;
    float ya1 = 0.0f, ya2 = 0.0f;
    u64 laba = 0;
    if (f0 < nframes) srca = index ? index[f0] : f0;
    if (f1 < nframes) srcb = index ? index[f1] : f1;
    if (f0 < nframes) { ya1 = y[srca * 128 + lane]; ya2 = y[srca * 128 + 64 + lane]; if (label && lane < 2) laba = label[srca * 2 + lane]; }
    while (f0 < nframes) {
        float yb1 = 0.0f, yb2 = 0.0f;
        u64 labb = 0;
        if (f1 < nframes) { yb1 = y[srcb * 128 + lane]; yb2 = y[srcb * 128 + 64 + lane]; if (label && lane < 2) labb = label[srcb * 2 + lane]; }
        long long srcc = 0;
        if (f2 < nframes) srcc = index ? index[f2] : f2;
        // ---- frame f0: front end
        yrow[lane] = ya1; yrow[64 + lane] = ya2;
        const FrontResult fr = front_device_vals(LF, __float_as_uint(ya1) & 0x7FFFFFFFu, __float_as_uint(ya2) & 0x7FFFFFFFu, Gcols, lane);
        const float y1 = yrow[fr.o1], y2 = yrow[fr.o2];            // y'[p] = y[perm[p]]
        wave_fence();
        // ---- scan (the frame body of osd_search2r_kernel)
        SearchFrame S;
        S.o1 = fr.o1; S.o2 = fr.o2;
        const u64 Pa = fr.Prow;
        const float w1 = __builtin_fabsf(y1), w2 = __builtin_fabsf(y2);
        LL.wpar()[lane] = w2;
        S.hm = __ballot(!(y1 > 0.0f));
        S.hp = __ballot(!(y2 > 0.0f));
        wave_fence();
        build_byte_luts<8>(LL.lut, LL.wpar(), lane);
        S.d0 = wave_xor64(((S.hm >> lane) & 1) ? Pa : 0ull) ^ S.hp;
        wave_fence();
        float best; int bestt; u64 bestD, bestE;
        search2r_device(LL, S, Pa, w1, dir, base2, lane, best, bestt, bestD, bestE);
        {
            const u64 mrb_bits = S.hm ^ bestE, par_bits = bestD ^ S.hp;
            u64 *const cw = LL.cw();
            if (lane < 2) cw[lane] = 0;
            wave_fence();
            if ((mrb_bits >> lane) & 1) atomicOr(&cw[S.o1 >> 6], 1ull << (S.o1 & 63));
            if ((par_bits >> lane) & 1) atomicOr(&cw[S.o2 >> 6], 1ull << (S.o2 & 63));
            wave_fence();
            const u64 word = lane < 2 ? cw[lane] : 0ull;
            if (lane < 2) cw_out[f0 * 2 + lane] = word;
            if (counts && __ballot(lane < 2 && word != laba) && lane == 0) atomicAdd(&counts[1], 1ull);
            wave_fence();
        }
        if (lane == 0) {
            if (metric_out) metric_out[f0] = best;
            if (best_out) best_out[f0] = bestt;
            if (ntep_out) ntep_out[f0] = 2081;
        }
        f0 = f1; f1 = f2; f2 += G;
        srca = srcb; srcb = srcc; ya1 = yb1; ya2 = yb2; laba = labb;
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
// One given TEP per frame (one_tep_compare, FS_OSD/fs_testing.py:51-64): re-encode the MRB hard decisions with the
// positions of `mask` flipped, Hamming distance and weighted distance of the candidate -- the SAME LUT evaluation and
// float order as the searches (the Python helper of that name used to restate the order in NumPy).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void osd_tep_eval_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                           const int *__restrict__ count, long long F,
                                                           const unsigned char *__restrict__ perm_in,
                                                           const u64 *__restrict__ parity_in, const u64 *__restrict__ mask,
                                                           u64 *__restrict__ cw_out, float *__restrict__ metric_out,
                                                           int *__restrict__ hd_out)
{
    __shared__ SearchLds lds[4];
    const int lane = threadIdx.x & 63;
    SearchLds &L = lds[threadIdx.x >> 6];
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }
    for (long long f = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); f < nframes; f += (long long)gridDim.x * 4) {
        const long long src = index ? index[f] : f;
        const SearchFrame S = search_prepare(L, y, src, perm_in, parity_in, f, lane);
        const u64 E = mask[f];
        const u64 D = S.d0 ^ wave_xor64(((E >> lane) & 1) ? L.P[lane] : 0ull);
        float mrb = 0.0f;                                  // flipped MRB weights, ascending position, sequential
        for (u64 m = E; m; m &= m - 1) mrb = mrb + L.w[__builtin_ctzll(m)];
        const float cost = tep_cost(L, mrb, D);
        search_finish(L, S, E, D, f, lane, cw_out);
        if (lane == 0) {
            if (metric_out) metric_out[f] = cost;
            if (hd_out) hd_out[f] = __popcll(E) + __popcll(D);
        }
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
    }
    LDPC_HIP(hipMalloc((void **)&st->d_index_errors, sizeof(unsigned long long)));
    LDPC_HIP(hipMemset(st->d_index_errors, 0, sizeof(unsigned long long)));
    if (int rc = pb_ctx_init(ctx)) return rc;
    ctx->osd_ok = true;
    return LDPC_OK;
}

static void free_stream_ws(StreamWs &w)
{
    (void)hipFree(w.d_perm); (void)hipFree(w.d_parity); (void)hipFree(w.d_index_safe);
    (void)hipFree(w.d_pb_ctl); (void)hipFree(w.d_pb_list); (void)hipFree(w.d_pb_spill); (void)hipFree(w.d_pb_prep);
    (void)hipFree(w.d_pb_carry);
    w = StreamWs();
}

void osd_ctx_release(ldpc_ctx *ctx)
{
    (void)hipFree(ctx->d_Gcols);
    (void)hipFree(ctx->d_tep);
    if (OsdState *st = state(ctx)) {
        for (auto &kv : st->ws) free_stream_ws(kv.second);
        (void)hipFree(st->d_tep_fs);
        (void)hipFree(st->d_base2);
        (void)hipFree(st->d_cdf_half);
        (void)hipFree(st->d_index_errors);
        delete st;
    }
    ctx->osd_state = nullptr;
}

bool stream_capturing(hipStream_t s)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}

// the workspace of `s` with room for the front-end results of `frames` frames (0: just look it up);
// allocation happens on a stream's first call or when a call outgrows it -- never while `s` is capturing, and never
// again once a graph was captured on `s` (the graph's nodes hold the buffer addresses: ADVICE r02)
static int stream_ws(ldpc_ctx *ctx, hipStream_t s, int64_t frames, StreamWs **out)
{
    OsdState *st = state(ctx);
    std::lock_guard<std::mutex> lock(st->mu);
    StreamWs &w = st->ws[s];
    if (frames > 0 && frames < st->reserve_frames) frames = st->reserve_frames;
    const bool capturing = stream_capturing(s);
    if (frames > w.cap) {
        if (capturing)
            return fail(LDPC_E_NOMEM, "OSD workspace of this stream holds %lld frames, %lld needed: run one call (or "
                        "ldpc_osd_reserve_stream) on the stream before capturing", (long long)w.cap, (long long)frames);
        if (w.captured)
            return fail(LDPC_E_NOMEM, "OSD workspace of this stream (%lld frames) is referenced by a captured graph and cannot grow to "
                        "%lld: destroy the graph and call ldpc_osd_release_stream, or reserve the larger size before capturing",
                        (long long)w.cap, (long long)frames);
        (void)hipFree(w.d_perm); (void)hipFree(w.d_parity);
        w.d_perm = nullptr; w.d_parity = nullptr; w.cap = 0;
        if (hipMalloc((void **)&w.d_perm, (size_t)frames * 128) != hipSuccess ||
            hipMalloc((void **)&w.d_parity, (size_t)frames * 64 * sizeof(u64)) != hipSuccess)
            return fail(LDPC_E_NOMEM, "OSD workspace for %lld frames could not be allocated", (long long)frames);
        w.cap = frames;
    }
    if (capturing) w.captured = true;
    *out = &w;
    return LDPC_OK;
}

// ldpc_osd_params.y_frames (debug aid): the frame list the kernels will follow, with every entry outside [0, y_frames)
// replaced by 0 and counted
__global__ __launch_bounds__(256) void index_guard_kernel(const int *__restrict__ index, const int *__restrict__ count, long long F,
                                                          long long y_frames, int *__restrict__ safe, unsigned long long *__restrict__ errors)
{
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }
    for (long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x; f < nframes; f += (long long)gridDim.x * blockDim.x) {
        const int v = index[f];
        const bool bad = v < 0 || v >= y_frames;
        safe[f] = bad ? 0 : v;
        if (bad) atomicAdd(errors, 1ull);
    }
}

// returns the list the kernels should use: d_index itself, or its sanitised copy in the stream's workspace
static int guarded_index(ldpc_ctx *ctx, const ldpc_osd_params *p, const int32_t *d_index, const int32_t *d_count, int64_t F, hipStream_t s,
                         const int32_t **out)
{
    *out = d_index;
    if (!d_index || p->y_frames <= 0 || F <= 0) return LDPC_OK;
    OsdState *st = state(ctx);
    int *safe;
    {
        std::lock_guard<std::mutex> lock(st->mu);
        StreamWs &w = st->ws[s];
        if (F > w.index_cap) {
            if (stream_capturing(s) || w.captured)
                return fail(LDPC_E_NOMEM, "ldpc_osd_params.y_frames: the checked frame list of this stream holds %lld entries, %lld needed; "
                            "it cannot grow during or after a capture", (long long)w.index_cap, (long long)F);
            (void)hipFree(w.d_index_safe); w.d_index_safe = nullptr; w.index_cap = 0;
            if (hipMalloc((void **)&w.d_index_safe, sizeof(int) * (size_t)F) != hipSuccess)
                return fail(LDPC_E_NOMEM, "checked frame list for %lld frames could not be allocated", (long long)F);
            w.index_cap = F;
        }
        if (stream_capturing(s)) w.captured = true;
        safe = w.d_index_safe;
    }
    const int64_t g = (F + 255) / 256;
    hipLaunchKernelGGL(index_guard_kernel, dim3((unsigned)(g < 1024 ? g : 1024)), dim3(256), 0, s, d_index, d_count, (long long)F,
                       (long long)p->y_frames, safe, st->d_index_errors);
    LDPC_HIP(hipGetLastError());
    *out = safe;
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

static int check_params(ldpc_ctx *ctx, const ldpc_osd_params *p, const char *who)
{
    if (!ctx->osd_ok) return fail(LDPC_E_UNSUPPORTED, "OSD kernels need an (n=128, k=64) code; this one is (%d,%d)", ctx->code.n, ctx->code.k);
    if (p->order < 0 || p->order > 3) return fail(LDPC_E_ARG, "%s: order %d outside 0..3", who, p->order);
    if (p->algo != LDPC_OSD_CONVENTIONAL && p->algo != LDPC_OSD_FS && p->algo != LDPC_OSD_PB)
        return fail(LDPC_E_ARG, "%s: unknown search algorithm %d", who, p->algo);
    if (p->algo == LDPC_OSD_PB && p->order < 1) return fail(LDPC_E_ARG, "%s: PB-OSD needs order >= 1", who);
    return LDPC_OK;
}

int ldpc_osd_reserve(ldpc_ctx *ctx, int64_t max_frames)
{
    if (!ctx || max_frames < 0) return fail(LDPC_E_ARG, "ldpc_osd_reserve: bad arguments");
    if (!ctx->osd_ok) return fail(LDPC_E_UNSUPPORTED, "OSD kernels need an (n=128, k=64) code; this one is (%d,%d)", ctx->code.n, ctx->code.k);
    {
        OsdState *st = state(ctx);
        std::lock_guard<std::mutex> lock(st->mu);
        if (max_frames > st->reserve_frames) st->reserve_frames = max_frames;
    }
    StreamWs *w;
    return stream_ws(ctx, nullptr, max_frames, &w);   // the NULL stream's workspace now; other streams on their first call
}

int ldpc_osd_reserve_stream(ldpc_ctx *ctx, int64_t max_frames, const ldpc_osd_params *params, void *stream)
{
    if (!ctx || max_frames < 0) return fail(LDPC_E_ARG, "ldpc_osd_reserve_stream: bad arguments");
    if (!ctx->osd_ok) return fail(LDPC_E_UNSUPPORTED, "OSD kernels need an (n=128, k=64) code; this one is (%d,%d)", ctx->code.n, ctx->code.k);
    StreamWs *w;
    int rc = stream_ws(ctx, (hipStream_t)stream, max_frames, &w);
    if (rc || !params) return rc;
    if ((rc = check_params(ctx, params, "ldpc_osd_reserve_stream"))) return rc;
    if (params->algo == LDPC_OSD_PB && max_frames > 0) return pb_reserve(ctx, (hipStream_t)stream, max_frames, params->order);
    return LDPC_OK;
}

int ldpc_osd_release_stream(ldpc_ctx *ctx, void *stream)
{
    if (!ctx) return fail(LDPC_E_ARG, "ldpc_osd_release_stream: null ctx");
    OsdState *st = state(ctx);
    if (!st) return LDPC_OK;
    if (stream_capturing((hipStream_t)stream)) return fail(LDPC_E_ARG, "ldpc_osd_release_stream: the stream is capturing");
    std::lock_guard<std::mutex> lock(st->mu);
    auto it = st->ws.find((hipStream_t)stream);
    if (it != st->ws.end()) { free_stream_ws(it->second); st->ws.erase(it); }
    return LDPC_OK;
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

// launches the search kernel selected by p->algo on front-end results (d_perm, d_parity)
// label / counts / fused: the success counters of ldpc_osd_counts accumulated by the search kernel itself where it can
// (*fused is set then, and the caller skips the separate counting launch)
static int launch_search(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                         const unsigned char *d_perm, const u64 *d_parity, const ldpc_osd_params *p, uint64_t *d_cw,
                         float *d_metric, int32_t *d_best, int32_t *d_ntep, hipStream_t s, const uint64_t *d_label = nullptr,
                         int64_t *d_counts = nullptr, bool *fused = nullptr)
{
    if (fused) *fused = false;
    OsdState *st = state(ctx);
    if (p->algo == LDPC_OSD_PB) {
        return launch_pb(ctx, d_y, d_index, d_count, F, d_perm, d_parity, p, d_cw, d_metric, d_best, d_ntep, s);
    } else if (p->algo == LDPC_OSD_FS) {
        FsParams fp;
        fp.order = p->order; fp.quirk = p->fs_reference_quirk != 0;
        fp.beta_term = (float)((double)p->fs_beta * (double)(kOsdN - kOsdK));   // fs_testing.py:138
        fp.tau_e = p->fs_tau_e; fp.tau_psc = p->fs_tau_psc;
        for (int w = 0; w < 4; ++w) { fp.cls_off[w] = st->fs_off[w]; fp.cls_cnt[w] = st->fs_cnt[w]; }
        hipLaunchKernelGGL(osd_fs_kernel, dim3((unsigned)(F < 65536 ? F : 65536)), dim3(64), 0, s, d_y, d_index, d_count, (long long)F,
                           d_perm, d_parity, st->d_tep_fs, fp, reinterpret_cast<u64 *>(d_cw), d_metric, d_best,
                           d_ntep);
    } else if (p->order == 2 && !(p->reserved & 1) && ctx->dpp_wave_rol_dir != 0 && !(p->reserved & 8)) {
        // 4 wavefronts per SIMD are resident (10 KiB of LDS each); the grid is 6x that, ~1.5 frames per workgroup at the
        // headline size: the dispatcher then evens out the different scan times, and the prefetch still covers the second
        // frame (measured, 33 487 frames: 1x 119 us, 2x 117, 3x 108, 4x 107, 6x 100, 9x 102 per call incl. events)
        const long long grid = (long long)ctx->cu_count * 16 * 6;
        hipLaunchKernelGGL(osd_search2r_kernel, dim3((unsigned)(F < grid ? F : grid)), dim3(64), 0, s, d_y, d_index, d_count, (long long)F,
                           d_perm, d_parity, ctx->dpp_wave_rol_dir, st->d_base2, reinterpret_cast<u64 *>(d_cw), d_metric, d_best, d_ntep,
                           reinterpret_cast<const u64 *>(d_label), d_label ? reinterpret_cast<u64 *>(d_counts) : nullptr);
        if (fused && d_label && d_counts) *fused = true;
    } else if (p->order == 2 && !(p->reserved & 1)) {
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

}  // extern "C"

namespace ldpc {
// ldpc_osd_decode (+ ldpc_osd_counts where the kernel can count itself: *counted_by_search).  The conventional order-2 OSD runs
// front end and scan in ONE kernel with no workspace (osd_fused2r_kernel); everything else the front end into the stream's
// workspace, then the search.  (params->reserved bits 0 / 3, the cross-check scans, keep the two-kernel route.)
int osd_decode_counted(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                       const ldpc_osd_params *p, uint64_t *d_cw, float *d_metric, int32_t *d_best, int32_t *d_ntep,
                       const uint64_t *d_label, int64_t *d_counts, hipStream_t s, bool *counted_by_search)
{
    if (counted_by_search) *counted_by_search = false;
    int rc;
    if ((rc = guarded_index(ctx, p, d_index, d_count, F, s, &d_index))) return rc;
    if (p->algo == LDPC_OSD_CONVENTIONAL && p->order == 2 && !(p->reserved & 9) && ctx->dpp_wave_rol_dir != 0) {
        const long long grid = (long long)ctx->cu_count * 16 * 6;      // (as the scan alone: 6x the resident wavefronts)
        const bool cnt = d_label && d_counts;
        hipLaunchKernelGGL(osd_fused2r_kernel, dim3((unsigned)(F < grid ? F : grid)), dim3(64), 0, s, d_y, d_index, d_count, (long long)F,
                           reinterpret_cast<const u64 *>(ctx->d_Gcols), ctx->dpp_wave_rol_dir, state(ctx)->d_base2,
                           reinterpret_cast<u64 *>(d_cw), d_metric, d_best, d_ntep, cnt ? reinterpret_cast<const u64 *>(d_label) : nullptr,
                           cnt ? reinterpret_cast<u64 *>(d_counts) : nullptr);
        LDPC_HIP(hipGetLastError());
        if (counted_by_search) *counted_by_search = cnt;
        return LDPC_OK;
    }
    // PB-OSD with reserved bit 0: the front end runs INSIDE the first PB kernel and nothing goes through a workspace (round 4;
    // not with bit 2, the list-replay route, which writes no records to set a frame up from).  Measured against the two kernels
    // (osd_front + pb_singles, per 131 072-frame step): 170 against 164 us at 2.5 dB, 507 against 481 us at 1.0 dB, 58 against 102 MB
    // and 236 against 400 MB of HBM traffic -- the path is instruction-bound, the traffic was never its limiter: an option, not
    // the default.
    if (p->algo == LDPC_OSD_PB && (p->reserved & 5) == 1)
        return launch_pb(ctx, d_y, d_index, d_count, F, nullptr, nullptr, p, d_cw, d_metric, d_best, d_ntep, s);
    StreamWs *w;
    if ((rc = stream_ws(ctx, s, F, &w))) return rc;
    hipLaunchKernelGGL(osd_front_kernel, dim3((unsigned)(F < 65536 ? F : 65536)), dim3(64), 0, s, d_y, d_index, d_count, (long long)F,
                       reinterpret_cast<const u64 *>(ctx->d_Gcols), w->d_perm, w->d_parity, (int *)nullptr);
    return launch_search(ctx, d_y, d_index, d_count, F, w->d_perm, w->d_parity, p, d_cw, d_metric, d_best, d_ntep, s);
}

// ldpc_osd_search + ldpc_osd_counts for ldpc_pipeline_run: one launch where the search kernel can count itself
int osd_search_counted(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                       const uint8_t *d_perm, const uint64_t *d_parity, const ldpc_osd_params *p, uint64_t *d_cw, float *d_metric,
                       int32_t *d_best, int32_t *d_ntep, const uint64_t *d_label, int64_t *d_counts, hipStream_t s, bool *counted_by_search)
{
    if (counted_by_search) *counted_by_search = false;
    if (!ctx || !p || F < 0 || (F > 0 && (!d_y || !d_cw || !d_perm || !d_parity)))
        return fail(LDPC_E_ARG, "ldpc_osd_search: bad arguments");
    int rc = check_params(ctx, p, "ldpc_osd_search");
    if (rc) return rc;
    if (F == 0) return LDPC_OK;
    if ((rc = guarded_index(ctx, p, d_index, d_count, F, s, &d_index))) return rc;
    bool fused = false;
    rc = launch_search(ctx, d_y, d_index, d_count, F, d_perm, reinterpret_cast<const u64 *>(d_parity), p, d_cw, d_metric, d_best, d_ntep,
                       s, d_label, d_counts, &fused);
    if (counted_by_search) { *counted_by_search = fused; return rc; }     // (the caller places the counting launch itself)
    if (rc || fused || !d_label || !d_counts) return rc;
    return ldpc_osd_counts(ctx, d_cw, d_label, d_index, d_count, d_ntep, F, d_counts, s);
}
}  // namespace ldpc

extern "C" {

int ldpc_osd_search(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                    const uint8_t *d_perm, const uint64_t *d_parity, const ldpc_osd_params *p, uint64_t *d_cw,
                    float *d_metric, int32_t *d_best, int32_t *d_ntep, void *stream)
{
    if (!ctx || !p || F < 0 || (F > 0 && (!d_y || !d_cw || !d_perm || !d_parity)))
        return fail(LDPC_E_ARG, "ldpc_osd_search: bad arguments");
    int rc = check_params(ctx, p, "ldpc_osd_search");
    if (rc) return rc;
    if (F == 0) return LDPC_OK;
    if ((rc = guarded_index(ctx, p, d_index, d_count, F, (hipStream_t)stream, &d_index))) return rc;
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
    return osd_decode_counted(ctx, d_y, d_index, d_count, F, p, d_cw, d_metric, d_best, d_ntep, nullptr, nullptr, (hipStream_t)stream, nullptr);
}

int ldpc_osd_index_errors(ldpc_ctx *ctx, int64_t *count)
{
    if (!ctx || !count) return fail(LDPC_E_ARG, "ldpc_osd_index_errors: null argument");
    *count = 0;
    OsdState *st = state(ctx);
    if (!st || !st->d_index_errors) return LDPC_OK;
    unsigned long long v = 0;
    LDPC_HIP(hipMemcpy(&v, st->d_index_errors, sizeof(v), hipMemcpyDeviceToHost));   // (synchronises the device)
    *count = (int64_t)v;
    return LDPC_OK;
}

int ldpc_osd_tep_eval(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                      const uint8_t *d_perm, const uint64_t *d_parity, const uint64_t *d_mask, uint64_t *d_cw, float *d_metric,
                      int32_t *d_hd, void *stream)
{
    if (!ctx || F < 0 || (F > 0 && (!d_y || !d_perm || !d_parity || !d_mask || !d_cw)))
        return fail(LDPC_E_ARG, "ldpc_osd_tep_eval: bad arguments");
    if (!ctx->osd_ok) return fail(LDPC_E_UNSUPPORTED, "OSD kernels need an (n=128, k=64) code; this one is (%d,%d)", ctx->code.n, ctx->code.k);
    if (F == 0) return LDPC_OK;
    hipLaunchKernelGGL(osd_tep_eval_kernel, dim3(osd_grid(F)), dim3(256), 0, (hipStream_t)stream, d_y, d_index, d_count, (long long)F,
                       d_perm, reinterpret_cast<const u64 *>(d_parity), reinterpret_cast<const u64 *>(d_mask),
                       reinterpret_cast<u64 *>(d_cw), d_metric, d_hd);
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
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
