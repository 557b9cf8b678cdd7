// PB-OSD kernels for (128,64) codes on gfx950 (MI355X).
//
// Reference (paths relative to LDPC_128/): pb_osd, PB_OSD/pb_testing.py:100-149 -- best-first TEP generation
// from a growing frontier list (optimal_tep_sequence :366-397: "first minimum of the reliability sums in list
// order", pop, append <= 2 children) with two probabilistic stopping rules (acquire_prob_promising :448-461,
// acquire_p_e_suc :423-436, thresholds :485-500).  All probabilities follow the float conventions of
// oracle/ldpc_oracle.c orc_pb_osd: float32 with det_expf (IEEE + - * / only, host and device agree bit for
// bit), float64 binomial-CDF recurrences, threshold comparisons in float64.
//
// The list is NOT replayed TEP by TEP (the round-1 kernel did that: one wavefront, ~1.2 us per TEP, 51 ms for
// the rare frame on which no rule fires).  Three facts make the search batch-parallel and still exact
// (tests/pb_chunk_model.py states the algorithm in NumPy and checks it against the literal oracle):
//   1. every TEP has exactly one parent (extended child: e U {63}; adjacent child: largest index - 1), so the
//      list never holds duplicates and the pop sequence visits each TEP of weight 1..order exactly once;
//   2. a child's float32 sum is >= its parent's (monotone rounding), hence the pop sequence is the TEPs sorted
//      by (sum, list slot), and slot(t) < slot(u) <=> parent(t) is popped before parent(u), or they share the
//      parent and t is the extended child -- a comparator that only recurses when sums tie exactly;
//   3. the stopping rules see the visit order only through "best so far", a prefix minimum.
// So: a CHUNK of the visit order = all TEPs with sum in (lo, hi], sorted; its costs are evaluated in parallel and
// the sequential rules are recovered with prefix scans and a "first stop" reduction.
//
//   pb_singles_kernel  one frame per wavefront: the pop sequence starts with the weight-1 TEPs {63}, {62}, ...
//                      while |y'_p| < |y'_62| + |y'_63| (the smallest weight-2 sum); one TEP per lane.  About
//                      half of the frames stop here at 2.5 dB; the others are appended to list A.
//   pb_block_kernel    one frame of list A per 256-thread workgroup, restarted from its first TEP: two chunks of ~768
//                      TEPs (capacity 1024), each = the members of a sum range generated directly from the sorted
//                      reliabilities (PbItems: no table pass), its upper bound chosen by pb_pick_bound; bucket rank sort
//                      in LDS; exact tie repair; parallel evaluation; sequential rules by prefix scans.  96 % of the
//                      frames stop here; the rest go to list C with their search state, massive ties to list B.
//   pb_heavy_kernel    one frame of list C per 1024-thread workgroup: chunks of ~3072 TEPs (capacity 4096) until a
//                      rule fires or all N_max TEPs are visited.
//   pb_seq_kernel      the literal list replay (round-1 kernel) for list B.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ldpc_internal.h"
#include "ldpc_wave.h"
#include "ldpc_search.h"
#include "ldpc_osd_state.h"

namespace ldpc {

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
    int t1, t2, t3;              // chunk targets: stage A first / second chunk, stage B
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


// (64 - i) / (i + 1): the ratio of consecutive binomial coefficients C(64, i+1) / C(64, i), correctly rounded
// float64 -- the same values the host computes for the oracle's recurrence
struct PbCoef {
    double v[64];
    constexpr PbCoef() : v() { for (int i = 0; i < 64; ++i) v[i] = (double)(64 - i) / (double)(i + 1); }
};
__constant__ PbCoef kPbCoef;

// per-frame PB quantities (wave-uniform), float conventions of the oracle
struct PbFrame {
    float spl, lrb_mean;       // prod (1 - q_p) over the MRB (com_mrb_prob :35-41), mean |y'| over the LRB (:401)
    double p_t_suc, p_t_pro;   // calculate_two_thresholds :485-500
};

// One wavefront: q[p] = sigmoid(c4 |y'_p|), the binomial CDF table of the mean LRB error probability and
// the two thresholds.  L.w must be in place; q / cdfA are per-frame LDS arrays.
__device__ __forceinline__ PbFrame pb_frame_setup(const SearchLds &L, float *q, double *cdfA, float c4, int order, int nmax, int lane,
                                                  float best0 = __builtin_inff())
{
    q[lane] = 1.0f / (1.0f + det_expf(-(c4 * L.w[lane])));
    q[lane + 64] = 1.0f / (1.0f + det_expf(-(c4 * L.w[lane + 64])));
    wave_fence();
    // sequential (ascending position) means / product, as the oracle defines them
    float a1 = 0.0f, aw = 0.0f, at = 0.0f, spl = 1.0f;
#pragma unroll 4
    for (int p = 0; p < 64; ++p) {
        a1 = a1 + q[64 + p];
        aw = aw + L.w[64 + p];
        at = at + q[p];
        spl = spl * (1.0f - q[p]);
    }
    const float p1 = a1 / 64.0f, lrb_mean = aw / 64.0f, pt = at / 64.0f;
    // The rules read cdfA[beta] with beta = clamp(floor((best - sum) / lrb_mean)), best <= best0 (the order-0 metric) and
    // sum >= 0, so entries above floor(best0 / lrb_mean) are never read (division and floor are monotone): the table's
    // dependent float64 recurrence stops there -- typically after ~10 of its 64 steps.
    const float bq = __builtin_floorf(best0 / lrb_mean);
    const int ncdf = bq > 0.0f ? (bq < 64.0f ? (int)bq : 64) : 0;
    // binomial CDF tables by the pmf recurrence (float64): full table for p1, up to `order` for pt.  Lane i holds the
    // i-th coefficient; the dependent chain takes it by v_readlane (a scalar load per step sat on the critical path).
    double niu;
    const double coef_l = kPbCoef.v[lane];
    const auto coef = [coef_l](int i) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(coef_l);
        const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, i), hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), i);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    {
        double qq = 1.0 - (double)p1, t = qq;
        for (int s = 0; s < 6; ++s) t = t * t;
        const double ratio = (double)p1 / qq;
        double acc = t;
        if (lane == 0) cdfA[0] = acc;
#pragma unroll 2
        for (int i = 0; i < ncdf; ++i) {
            t = t * coef(i) * ratio;
            acc = acc + t;
            if (lane == 0) cdfA[i + 1] = acc;
        }
        qq = 1.0 - (double)pt; t = qq;
        for (int s = 0; s < 6; ++s) t = t * t;
        const double ratio2 = (double)pt / qq;
        acc = t;
        for (int i = 0; i < order; ++i) { t = t * coef(i) * ratio2; acc = acc + t; }
        niu = acc;
    }
    PbFrame F;
    F.spl = spl; F.lrb_mean = lrb_mean;
    F.p_t_suc = 0.99 * niu;
    F.p_t_pro = 0.002 * __builtin_sqrt((1.0 - niu) / (double)nmax);
    wave_fence();
    return F;
}

// promising-probability rule (acquire_prob_promising :448-461): true = stop
__device__ __forceinline__ bool pb_not_promising(float rs, float best, const PbFrame &F, float c4, const double *cdfA,
                                                 const double *cdfH, float &w1_out)
{
    const float w1 = det_expf(c4 * rs) * F.spl, w2 = 1.0f - w1;
    const float bt = __builtin_floorf((best - rs) / F.lrb_mean);
    const int beta = bt > 0.0f ? (bt < 64.0f ? (int)bt : 64) : 0;
    float bs = 0.0f;
    bs = bs + w1 * (float)cdfA[beta];
    bs = bs + w2 * (float)cdfH[beta];
    w1_out = w1;
    return (double)bs < F.p_t_pro;
}

// success rule (acquire_p_e_suc :423-436) for a candidate that became the best: true = stop.
// tq[p] = {2 (1 - q_p), 2 q_p} of parity position p (pb_success_terms): the factor of the sequential product is picked
// by the discrepancy bit -- a broadcast LDS read and a select per position instead of recomputing both terms.
__device__ __forceinline__ void pb_success_terms(const float *q, float2 *tq, int lane)
{
    const float qp = q[64 + lane];
    tq[lane] = make_float2(2.0f * (1.0f - qp), 2.0f * qp);
}

__device__ __forceinline__ bool pb_success(u64 D, float w1, const float2 *tq, const PbFrame &F)
{
    const float ratio = (1.0f - w1) / w1;
    float prod = 1.0f;
#pragma unroll 8
    for (int p = 0; p < 64; ++p) {
        const float2 t = tq[p];
        prod = prod * (((D >> p) & 1) ? t.y : t.x);
    }
    const float p_suc = 1.0f / (1.0f + ratio / prod);
    return (double)p_suc > F.p_t_suc;
}

// ---------------------------------------------------------------------------------------
// TEP table of the chunk kernels: ids 0..63 = {63 - id}; ids 64..2079 = pairs, 2080..43743 = triples, each
// class by DESCENDING smallest position, so "all positions >= a" is a prefix of every class.
// ---------------------------------------------------------------------------------------
constexpr int kPbPairs0 = 64, kPbTriples0 = 64 + 2016, kPbTabSize = 64 + 2016 + 41664;

struct PbTep {
    int p0, p1, p2, wt;
};
__device__ __forceinline__ PbTep pb_tep(const uchar4 *__restrict__ tab, int id)
{
    const uchar4 t = tab[id];
    return PbTep{t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ float pb_sum(const float *w, const PbTep &t)
{
    float s = w[t.p0];
    if (t.wt > 1) s = s + w[t.p1];
    if (t.wt > 2) s = s + w[t.p2];
    return s;
}
__device__ __forceinline__ int pb_last(const PbTep &t) { return t.wt == 1 ? t.p0 : (t.wt == 2 ? t.p1 : t.p2); }
// children pushed by a pop minus the popped entry itself (optimal_tep_sequence :381-396)
__device__ __forceinline__ int pb_delta(const PbTep &t, int order)
{
    const int last = pb_last(t), prev = t.wt == 2 ? t.p0 : t.p1;
    const int has1 = last < 63 && t.wt < order;
    const int has2 = t.wt > 1 ? (last - prev > 1) : (last - 1 > -1);
    return has1 + has2 - 1;
}
// t := parent(t); returns 0 = t was the extended child, 1 = the adjacent child, -1 = t is the root {63}
__device__ __forceinline__ int pb_to_parent(PbTep &t)
{
    const int last = pb_last(t);
    if (last == 63) {
        if (t.wt == 1) return -1;
        --t.wt;
        return 0;
    }
    if (t.wt == 1) t.p0 = last + 1; else if (t.wt == 2) t.p1 = last + 1; else t.p2 = last + 1;
    return 1;
}
__device__ __forceinline__ bool pb_same(const PbTep &a, const PbTep &b)
{
    return a.wt == b.wt && a.p0 == b.p0 && (a.wt < 2 || a.p1 == b.p1) && (a.wt < 3 || a.p2 == b.p2);
}
// t is popped before u (t != u): (sum, list slot) order, the slot order through the parents
__device__ bool pb_visit_less(const float *w, PbTep t, PbTep u)
{
    for (;;) {
        const float st = pb_sum(w, t), su = pb_sum(w, u);
        if (st != su) return st < su;
        const int kt = pb_to_parent(t), ku = pb_to_parent(u);
        if (kt < 0) return true;
        if (ku < 0) return false;
        if (pb_same(t, u)) return kt < ku;
    }
}
__device__ __forceinline__ void pb_apply(const SearchLds &L, const PbTep &t, u64 d0, u64 &D, u64 &E)
{
    D = d0 ^ L.P[t.p0]; E = 1ull << t.p0;
    if (t.wt > 1) { D ^= L.P[t.p1]; E |= 1ull << t.p1; }
    if (t.wt > 2) { D ^= L.P[t.p2]; E |= 1ull << t.p2; }
}

struct PbOut {
    u64 *cw; float *metric; int *best, *ntep, *aux;
};

// per-frame quantities of pb_frame_setup, written by the stage that computed them first (pb_singles_kernel) for the
// frames it hands on, so that the workgroup kernels load ~1 KiB instead of repeating two 64-step sequential loops
struct PbPrep {
    double cdfA[65];
    float q[128];
    PbFrame fr;
};
__device__ __forceinline__ void pb_write(SearchLds &L, const SearchFrame &S, const PbOut &O, long long f, int lane, u64 bestE,
                                         u64 bestD, float best, int bestidx, int ntep, int cmp, int suc1, int suc2, int stop)
{
    search_finish(L, S, bestE, bestD, f, lane, O.cw);
    if (lane == 0) {
        if (O.metric) O.metric[f] = best;
        if (O.best) O.best[f] = bestidx;
        if (O.ntep) O.ntep[f] = ntep;
        if (O.aux) { O.aux[f * 4] = cmp; O.aux[f * 4 + 1] = suc1; O.aux[f * 4 + 2] = suc2; O.aux[f * 4 + 3] = stop; }
    }
    wave_fence();
}

// inclusive wave scans (lane order)
__device__ __forceinline__ float wave_incl_min(float v, int) { return wave_incl_min_dpp(v); }
__device__ __forceinline__ int wave_incl_add(int v, int) { return wave_incl_add_dpp(v); }

// ---------------------------------------------------------------------------------------
// stage 1: the weight-1 head of the pop sequence, one frame per wavefront, one TEP per lane
//   mode 0: normal; 1: every frame straight to list A (block kernel); 2: every frame to list B (list replay)
// ---------------------------------------------------------------------------------------
struct PbSinglesLds {
    SearchLds s;
    double cdfA[65], cdfH[65];
    float q[128];
    float2 tq[64];
};

// (one wavefront per workgroup: 11.2 KiB of LDS each, 14 resident per CU -- four per workgroup left 12)
template <bool PROF>
__global__ __launch_bounds__(64) void pb_singles_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                         const int *__restrict__ count, long long F,
                                                         const unsigned char *__restrict__ perm_in,
                                                         const u64 *__restrict__ parity_in, PbParams P, int mode,
                                                         const double *__restrict__ cdf_half,
                                                         int *__restrict__ ctl, int *__restrict__ listA, int *__restrict__ listB, int sub_cap,
                                                         PbPrep *__restrict__ prep, PbOut O, unsigned long long *__restrict__ prof_out)
{
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, plast = 0;
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();
#define PBS_STAMP(k) do { if constexpr (PROF) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); pt[k] += now__ - plast; plast = now__; } } while (0)
    __shared__ PbSinglesLds W;
    const int lane = threadIdx.x;
    SearchLds &L = W.s;
    long long nframes = F;
    if (count) { const long long c = *count; nframes = c < F ? c : F; }
    const long long wave = blockIdx.x;
    if (mode != 0) {   // hand every frame on, in frame order
        for (long long f = wave * 64 + lane; f < nframes; f += (long long)gridDim.x * 64) {
            if (mode == 1) listA[(f & (kPbSub - 1)) * sub_cap + (f >> 4)] = (int)f; else listB[f] = (int)f;
        }
        if (wave == 0 && mode == 1 && lane < kPbSub) ctl[kPbCtlLenA + kPbCtlLine * lane] = (int)((nframes - lane + kPbSub - 1) >> 4);
        if (wave == 0 && mode == 2 && lane == 0) ctl[kPbCtlLenB] = (int)nframes;
        return;
    }
    W.cdfH[lane] = cdf_half[lane];
    if (lane == 0) W.cdfH[64] = cdf_half[64];
    wave_fence();
    for (long long f = wave; f < nframes; f += gridDim.x) {
        const long long src = index ? index[f] : f;
        SearchFrame S;
        if constexpr (PROF) {
            int o1 = perm_in[f * 128 + lane], o2 = perm_in[f * 128 + 64 + lane];
            u64 Pr = parity_in[f * 64 + lane];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(o1), "+v"(o2), "+v"(Pr));
            PBS_STAMP(5);
            S = search_prepare_regs(L, y, src, o1, o2, Pr, lane);
        } else {
            S = search_prepare(L, y, src, perm_in, parity_in, f, lane);
        }
        PBS_STAMP(0);
        const float best0 = tep_cost(L, 0.0f, S.d0);
        const PbFrame Fr = pb_frame_setup(L, W.q, W.cdfA, P.c4, P.order, P.nmax, lane, best0);
        PBS_STAMP(1);
        pb_success_terms(W.q, W.tq, lane);
        wave_fence();
        // lane l <-> TEP {63 - l}, visit index l; valid while its weight is below the smallest weight-2 sum
        const int p = 63 - lane;
        const float rs = L.w[p];
        const float s2min = L.w[62] + L.w[63];
        const u64 vmask = __ballot(P.order == 1 || rs < s2min);
        const int nhead = (~vmask) ? __builtin_ctzll(~vmask) : 64;
        const bool valid = lane < nhead;
        const u64 D = S.d0 ^ L.P[p];
        const float cost = valid ? tep_cost(L, rs, D) : __builtin_inff();
        const float incl = wave_incl_min(cost, lane);
        float before = __shfl_up(incl, 1, 64);
        before = lane == 0 ? best0 : __builtin_fminf(before, best0);
        float w1;
        const bool stop1 = valid && pb_not_promising(rs, before, Fr, P.c4, W.cdfA, W.cdfH, w1);
        const bool newbest = valid && cost < before;
        bool stop2 = false;
        if (newbest) stop2 = pb_success(D, w1, W.tq, Fr);
        const u64 sm = __ballot(stop1 || stop2);
        PBS_STAMP(2);
        if (sm == 0 && P.order > 1) {   // no rule fired on the head: the block kernel takes the frame (and what was computed for it)
            PbPrep &R = prep[f];
            R.q[lane] = W.q[lane]; R.q[lane + 64] = W.q[lane + 64];
            R.cdfA[lane] = W.cdfA[lane];
            if (lane == 0) { R.cdfA[64] = W.cdfA[64]; R.fr = Fr; const int sl = (int)f & (kPbSub - 1); listA[sl * sub_cap + atomicAdd(&ctl[kPbCtlLenA + kPbCtlLine * sl], 1)] = (int)f; }
            PBS_STAMP(3);
            continue;
        }
        const int ls = sm ? __builtin_ctzll(sm) : 63;                 // (order 1 without a stop: all 64 TEPs visited)
        const int reason = sm ? (((__ballot(stop1) >> ls) & 1) ? 1 : 2) : 0;
        const int npop = ls + 1;
        const int nev = reason == 1 ? ls : ls + 1;
        const u64 nbm = __ballot(newbest) & (nev >= 64 ? ~0ull : ((1ull << nev) - 1));
        float best = best0;
        u64 bestD = S.d0, bestE = 0;
        int bestidx = 0;
        if (nbm) {
            const int lb = 63 - __builtin_clzll(nbm);
            best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cost), lb));
            bestD = readlane64(D, lb);
            bestE = 1ull << (63 - lb);
            bestidx = lb + 1;
        }
        // frontier sizes before the pops: 1, 1, 2, 3, ... (order > 1) or always 1 (order 1)
        const int ones = P.order > 1 ? (npop < 2 ? npop : 2) : npop;
        pb_write(L, S, O, f, lane, bestE, bestD, best, bestidx, sm ? npop : P.nmax, 2 * npop - ones, nev, __popcll(nbm), reason);
        PBS_STAMP(4);
    }
    if constexpr (PROF) { if (lane == 0) for (int k = 0; k < 6; ++k) atomicAdd(&prof_out[k], pt[k]); }
#undef PBS_STAMP
}

// ---------------------------------------------------------------------------------------
// stage 2: one frame per workgroup, sorted chunks
// ---------------------------------------------------------------------------------------
constexpr int kPbMaxTie = 16;

template <int NT, int CAP>
struct __attribute__((aligned(16))) PbBlockLds {
    SearchLds s;
    double cdfA[65], cdfH[65];
    float q[128];
    float2 tq[64];
    u64 gath[CAP];          // the chunk as gathered: (sum bits << 32) | table id; after the sort: the costs (float[CAP])
    u64 keys[CAP];          // the chunk in visit order
    int bucket[CAP];        // bucket sort: counts, then cursors
    float red_f[2][NT / 64];
    int red_i[2][NT / 64];
    PbFrame fr;
    // uniform search state
    float lo, hi_cur, best;
    int j, nlive, cmp, suc1, suc2, bestidx;
    u64 bestD, bestE, d0;
    // per-chunk scratch
    int nkeys, bstar, degenerate, fallback, gstop, reason, ones, nev, nnb, lnb, ticket;
    int sub, subtried, sublen[kPbSub];   // stage A: the sub-list being drained
    unsigned long long prof[24], prof_last;   // diagnostic build only (LDPC_PB_PROFILE)
};

// In-kernel stamps of the diagnostic instantiation (PROF = true, launched only when LDPC_PB_PROFILE is set): thread 0
// adds the shader cycles since the previous stamp to slot k.  The product instantiation contains none of this.
enum { kProfSetup = 0, kProfPassA, kProfHist, kProfGather, kProfSort, kProfTie, kProfEval1, kProfEval2, kProfCombine, kProfFill,
       kProfScatter, kProfFinish, kProfFrames, kProfChunks };
#define PB_STAMP(k) do { if constexpr (PROF) { if (tid == 0) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); B.prof[k] += now__ - B.prof_last; B.prof_last = now__; } } } while (0)

// ---------------------------------------------------------------------------------------
// Direct enumeration of a sum range.  The positions are sorted by reliability (w[0] >= w[1] >= ...) and float addition is
// monotone, so with the other positions fixed the sum of a TEP is non-increasing in its LAST position m.  The TEPs are
// therefore 2080 "items" -- all singles {m}; the pairs {i, m} of one i; the triples {i, j, m} of one (i, j) -- inside each
// of which the members with lo < sum <= T are a contiguous run [a, e) of m, found by a 7-step binary search in LDS.
// A thread owns 3 (1024 threads) or 9 (256 threads) items; counting a range is one search per item and a block scan, a
// chunk is written from the runs.  The cost follows the items and the chunk, not the 43 744-entry table, nothing is read
// from global memory, and the upper bound of a chunk can be ANY value -- the search is exact for every choice -- which
// pb_pick_bound uses to size the chunks.
//   item 0: singles;  items 1..63: pairs of i = item - 1;  items 64..2079: triples of the pair tab[item]
// (a wavefront-wide visit per item with lane = m was measured too: 30 instructions per item at ~6 % useful lanes for the
//  early chunks, 4-10x the time of the searches; a 1024-bin histogram of all sums costs as much as the table pass it
//  replaced)
// ---------------------------------------------------------------------------------------
template <int NT>
struct PbItems {
    static constexpr int IPT = (kPbTriples0 + NT - 1) / NT;
    // frame-independent (set once per workgroup), packed: bits 0-6 base + 1 (the last position runs over (base, 63];
    // singles: base = -1, an empty item: 63), bits 7-13 first fixed position + 1 (0: none), bit 14 triple item,
    // bits 16-31 idb (table id of member m = idb + m; singles: 63 - m)
    unsigned st[IPT];
    // per frame
    float sb[IPT];    // sum of the fixed positions
    unsigned ea[IPT]; // e | a << 8: members already visited (sum <= lo) [e, 64); members of the chunk being sized [a, e)
    __device__ __forceinline__ int base(int q) const { return (int)(st[q] & 127u) - 1; }
    __device__ __forceinline__ int e(int q) const { return (int)(ea[q] & 255u); }
    __device__ __forceinline__ int a(int q) const { return (int)(ea[q] >> 8); }
};

// first m in [lo, hi) with sb + w[m] <= T (hi if none), for three items of the thread at once: the seven LDS reads of an
// item depend on each other, those of different items do not
template <int NT, int G>
__device__ __forceinline__ void pb_first_le3(const PbItems<NT> &I, const float *w, float T, int (&lo)[PbItems<NT>::IPT], int (&hi)[PbItems<NT>::IPT])
{
    constexpr int Q0 = 3 * G, Q1 = Q0 + 3 < PbItems<NT>::IPT ? Q0 + 3 : PbItems<NT>::IPT;
#pragma unroll
    for (int it = 0; it < 7; ++it)
#pragma unroll
        for (int q = Q0; q < Q1; ++q) {
            const int mid = (lo[q] + hi[q]) >> 1;
            const bool act = lo[q] < hi[q], ok = I.sb[q] + w[mid < 63 ? mid : 63] <= T;
            hi[q] = (act && ok) ? mid : hi[q];
            lo[q] = (act && !ok) ? mid + 1 : lo[q];
        }
}

// The items are laid out by DESCENDING first position, so those that can hold a member with sum <= T -- their smallest
// sum, (w_i + w_62) + w_63 for the triples of i, is monotone in i -- are a prefix of the item list; a thread's item
// slots q >= the returned count hold no such item for any thread (uniform), and are skipped as a whole.
template <int NT>
__device__ __forceinline__ int pb_items_slots(const float *w, float T, int order, int lane)
{
    const u64 ok = __ballot((w[lane] + w[62]) + w[63] <= T);          // first positions i whose triples can reach below T
    const int imin = ok ? (int)__builtin_ctzll(ok) : 64;
    const int r = 64 - imin;                                          // positions >= imin
    const int items = order > 2 ? kPbPairs0 + r * (r - 1) / 2 : kPbPairs0;
    const int slots = (items + NT - 1) / NT;
    return slots < PbItems<NT>::IPT ? slots : PbItems<NT>::IPT;
}

template <int NT>
__device__ __forceinline__ void pb_first_le(const PbItems<NT> &I, const float *w, float T, int slots, int (&lo)[PbItems<NT>::IPT], int (&hi)[PbItems<NT>::IPT])
{
    pb_first_le3<NT, 0>(I, w, T, lo, hi);
    if constexpr (PbItems<NT>::IPT > 3) { if (slots > 3) pb_first_le3<NT, 1>(I, w, T, lo, hi); }
    if constexpr (PbItems<NT>::IPT > 6) { if (slots > 6) pb_first_le3<NT, 2>(I, w, T, lo, hi); }
    static_assert(PbItems<NT>::IPT <= 9, "item groups");
}

template <int NT>
__device__ __forceinline__ void pb_items_static(PbItems<NT> &I, const uchar4 *__restrict__ tab, int order, int tid)
{
    const int nitems = order > 2 ? kPbTriples0 : (order > 1 ? kPbPairs0 : 1);
#pragma unroll
    for (int q = 0; q < PbItems<NT>::IPT; ++q) {
        const int it = tid + q * NT;
        I.st[q] = 64u;                                              // (an empty item: base = 63)
        if (it == 0) I.st[q] = 0u;
        else if (it < kPbPairs0 && it < nitems) {
            const int i = it - 1, m = 63 - i;
            I.st[q] = (unsigned)(i + 1) | ((unsigned)(i + 1) << 7) | ((unsigned)(kPbPairs0 + m * (m - 1) / 2 - (i + 1)) << 16);
        } else if (it < nitems) {
            const uchar4 t = tab[it];
            const int i = t.x, j = t.y, m = 63 - i, r = 64 - j;
            I.st[q] = (unsigned)(j + 1) | ((unsigned)(i + 1) << 7) | (1u << 14) |
                      ((unsigned)(kPbTriples0 + m * (m - 1) * (m - 2) / 6 + m * (m - 1) / 2 - r * (r - 1) / 2 - (j + 1)) << 16);
        }
    }
}

// per-frame part: fixed sums, and the members with sum <= lo (already visited) cut off
template <int NT>
__device__ __forceinline__ void pb_items_frame(PbItems<NT> &I, const float *w, float lo)
{
    constexpr int IPT = PbItems<NT>::IPT;
    int l[IPT], h[IPT];
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
        const int i1 = (int)((I.st[q] >> 7) & 127u);     // first fixed position + 1
        float sb = i1 ? w[i1 - 1] : 0.0f;
        if (I.st[q] & (1u << 14)) sb = sb + w[I.base(q)];
        I.sb[q] = sb;
        l[q] = I.base(q) + 1; h[q] = 64;
    }
    if (!(lo < 0.0f)) pb_first_le(I, w, lo, IPT, l, h);
#pragma unroll
    for (int q = 0; q < IPT; ++q) { const int e = lo < 0.0f ? 64 : l[q]; I.ea[q] = (unsigned)e | ((unsigned)e << 8); }
}

// a := start of the members with sum <= T (never past e); returns this thread's number of members in [a, e)
template <int NT>
__device__ __forceinline__ int pb_items_bound(PbItems<NT> &I, const float *w, float T, int order, int lane)
{
    constexpr int IPT = PbItems<NT>::IPT;
    int l[IPT], h[IPT];
    const int slots = pb_items_slots<NT>(w, T, order, lane);
#pragma unroll
    for (int q = 0; q < IPT; ++q) { h[q] = I.e(q); l[q] = q < slots ? I.base(q) + 1 : h[q]; l[q] = l[q] < h[q] ? l[q] : h[q]; }
    pb_first_le(I, w, T, slots, l, h);
    int local = 0;
#pragma unroll
    for (int q = 0; q < IPT; ++q) { I.ea[q] = (I.ea[q] & 255u) | ((unsigned)l[q] << 8); local += I.e(q) - l[q]; }
    return local;
}

// exclusive prefix of `local` over the threads of the workgroup and the total (one barrier; `slot` alternates between
// successive calls so that a slow reader of the previous call is never overwritten)
template <int NT, int CAP>
__device__ __forceinline__ int pb_block_scan(PbBlockLds<NT, CAP> &B, int local, int lane, int wave, int slot, int &total)
{
    const int incl = wave_incl_add(local, lane);
    if (lane == 63) B.red_i[slot][wave] = incl;
    __syncthreads();
    int run = incl - local;
    total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { const int v = B.red_i[slot][w]; total += v; run += w < wave ? v : 0; }
    return run;
}

// the members [a, e) of my items as chunk keys from slot `run` on; then the range is consumed (e := a)
template <int NT, int CAP>
__device__ __forceinline__ void pb_items_write(PbBlockLds<NT, CAP> &B, PbItems<NT> &I, const float *w, int run)
{
#pragma unroll
    for (int q = 0; q < PbItems<NT>::IPT; ++q) {
        const int a = I.a(q), e = I.e(q), idb = (int)(I.st[q] >> 16);
        const bool singles = (I.st[q] & 127u) == 0u;
        for (int m = a; m < e; ++m)
            B.gath[run++] = ((u64)__float_as_uint(I.sb[q] + w[m]) << 32) | (unsigned)(singles ? 63 - m : idb + m);
        I.ea[q] = (unsigned)a | ((unsigned)a << 8);
    }
}

// Typical bound of the N smallest sums in units of the smallest triple sum m3 = w61 + w62 + w63 (medians over decoding
// failures at 2.5 dB; the ratio is scale-free and tight: +-6 % between the 10th and 90th percentile, where the count
// changes like the ~6th power of the bound).  Only a first guess: pb_pick_bound corrects it with exact counts.
__device__ __forceinline__ float pb_bound_guess(float n)
{
    const float l = __builtin_amdgcn_logf(n < 64.0f ? 64.0f : n);
    const float x[8] = {8.0f, 9.0f, 10.0f, 11.0f, 12.0f, 13.0f, 14.2877f, 15.4168f};     // log2 of 256 ... 20000, 43744
    const float g[8] = {0.80f, 0.89f, 1.02f, 1.14f, 1.23f, 1.33f, 1.52f, 2.2f};
    if (l <= x[0]) return g[0] * __builtin_amdgcn_exp2f((l - x[0]) / 6.0f);
    float r = g[7];
#pragma unroll
    for (int k = 6; k >= 0; --k) if (l <= x[k + 1]) r = g[k] + (g[k + 1] - g[k]) * (l - x[k]) / (x[k + 1] - x[k]);
    return r;
}

// Upper bound T of the next chunk: 0 < #(lo < sum <= T) <= CAP, aimed at `target` members.  First guess from
// pb_bound_guess (or, past the first chunk, from the growth exponent between the last two bounds), then corrected with the
// exact counts it meets (secant step on log N over log T), bisected when that stops helping.  Every wavefront runs the
// same arithmetic on the same values.  Returns the count in n (-1: the range cannot be split -- massively equal sums --
// and the frame goes to the list replay) and the write offset of this wavefront's share in wbase.
template <int NT, int CAP>
__device__ __forceinline__ float pb_pick_bound(PbBlockLds<NT, CAP> &B, PbItems<NT> &I, const float *w, int order, float lo, int done, int nall,
                                               int target, int lane, int wave, int &n, int &run)
{
    const float inf = __builtin_inff();
    const float m3 = (w[61] + w[62]) + w[63];
    const float want = (float)(done + target);
    float Tl = lo, Th = inf;
    float T = nall - done <= CAP ? inf : m3 * pb_bound_guess(want);
    if (!(T > lo)) T = lo > 0.0f ? lo * 1.05f : w[0];
    float tp = lo, np_ = (float)done;        // last point with a known count (lo > 0 and done > 0: usable for the exponent)
    for (int it = 0;; ++it) {
        int c;
        run = pb_block_scan(B, pb_items_bound(I, w, T, order, lane), lane, wave, it & 1, c);
        if (c > 0 && c <= CAP && (!(T < inf) || 5 * c >= 2 * target || it >= 2)) { n = c; return T; }
        if (it >= 40 || !(T < inf)) break;
        float Tn;
        if (c == 0) { Tl = T; Tn = T * 1.1f; }
        else {
            if (c > CAP) Th = T;
            const float tot = (float)(done + c);
            float p = 6.0f;
            if (tp > 0.0f && np_ > 0.0f && tot != np_ && T != tp) {
                const float pe = (__builtin_amdgcn_logf(tot) - __builtin_amdgcn_logf(np_)) / (__builtin_amdgcn_logf(T) - __builtin_amdgcn_logf(tp));
                if (pe > 1.5f && pe < 20.0f) p = pe;
            }
            tp = T; np_ = tot;
            Tn = T * __builtin_amdgcn_exp2f((__builtin_amdgcn_logf(want) - __builtin_amdgcn_logf(tot)) / p);
        }
        if (it >= 6 || !(Tn > Tl) || !(Tn < Th)) Tn = Th < inf ? Tl + (Th - Tl) * 0.5f : T * 1.2f;
        if (!(Tn > Tl) || !(Tn < Th)) break;
        T = Tn;
    }
    n = -1;
    return lo;
}

// The n gathered keys of one chunk (all TEPs of a sum range): sort into visit order, evaluate in parallel, apply
// the sequential rules.  Returns 0 = no rule fired (B.j / B.nlive advanced), 1 = stopped (stop / ntep set),
// 2 = a run of more than kPbMaxTie equal sums (frame goes to the list replay).
template <int NT, int CAP, bool PROF>
__device__ int pb_process_chunk(PbBlockLds<NT, CAP> &B, const uchar4 *__restrict__ tab, const PbParams &P, const PbFrame &Fr,
                                u64 d0, int n, float mn, float mx, int tid, int &stop, int &ntep)
{
    constexpr int W = NT / 64;
    SearchLds &L = B.s;
    const int lane = tid & 63, wave = tid >> 6;
    u64 *const K = B.gath;                              // the chunk in visit order ends up where it was gathered
    // ---- bucket sort.  CAP buckets over the chunk's sum range (mn, mx] -- its bounds, known to the caller -- hold ~1 key each (<= ~14 near the dense upper
    // end): counts -> offsets (scan) -> scatter into B.keys (grouped by bucket) -> every key counts the keys of its own
    // bucket that sort before it and lands at bucket start + rank in B.gath.  No dependent chains: a bitonic sort of the
    // same 1024 keys took 55 barrier-separated LDS steps (39 % of the stage-A kernel), an insertion sort inside the
    // buckets left one thread with ~50 dependent LDS round trips; this is ~8 barriers and independent reads.
    {
        for (int b = tid; b < CAP; b += NT) B.bucket[b] = 0;
        if (tid == 0) B.fallback = 0;
        __syncthreads();
        PB_STAMP(16);
        const float scale = mx > mn ? (float)CAP / (mx - mn) : 0.0f;
        const bool flat = !(scale < 3.0e38f);           // denormally close sums: one bucket
        for (int i = tid; i < n; i += NT) {
            const float sv = __uint_as_float((unsigned)(B.gath[i] >> 32));
            atomicAdd(&B.bucket[flat ? 0 : (int)__builtin_fminf((sv - mn) * scale, (float)(CAP - 1))], 1);
        }
        __syncthreads();
        PB_STAMP(17);
        constexpr int PERB = CAP / NT;
        int cnts[PERB], local = 0;
#pragma unroll
        for (int q = 0; q < PERB; ++q) { cnts[q] = B.bucket[tid * PERB + q]; local += cnts[q]; if (cnts[q] > 64) B.fallback = 1; }
        const int incl = wave_incl_add(local, lane);
        if (lane == 63) B.red_i[0][wave] = incl;
        __syncthreads();
        int run = incl - local;
        for (int w = 0; w < wave; ++w) run += B.red_i[0][w];
#pragma unroll
        for (int q = 0; q < PERB; ++q) { B.bucket[tid * PERB + q] = run; run += cnts[q]; }    // start offsets = cursors
        __syncthreads();
        PB_STAMP(18);
        if (!B.fallback) {
            for (int i = tid; i < n; i += NT) {
                const u64 key = B.gath[i];
                const float sv = __uint_as_float((unsigned)(key >> 32));
                B.keys[atomicAdd(&B.bucket[flat ? 0 : (int)__builtin_fminf((sv - mn) * scale, (float)(CAP - 1))], 1)] = key;
            }
            __syncthreads();   // cursors are now the END offsets of the buckets
            PB_STAMP(19);
            for (int i = tid; i < n; i += NT) {
                const u64 key = B.keys[i];
                const float sv = __uint_as_float((unsigned)(key >> 32));
                const int b = flat ? 0 : (int)__builtin_fminf((sv - mn) * scale, (float)(CAP - 1));
                const int end = B.bucket[b], start = b ? B.bucket[b - 1] : 0;
                int r = 0;
                for (int jj = start; jj < end; ++jj) r += B.keys[jj] < key;
                K[start + r] = key;
            }
            __syncthreads();
            PB_STAMP(20);
        } else {   // a crowded bucket (massively clustered sums): bitonic sort of the whole chunk in place
            int npow = 2;
            while (npow < n) npow <<= 1;
            for (int i = n + tid; i < npow; i += NT) K[i] = ~0ull;
            __syncthreads();
            for (int k = 2; k <= npow; k <<= 1)
                for (int jj = k >> 1; jj > 0; jj >>= 1) {
                    for (int i = tid; i < (npow >> 1); i += NT) {
                        const int a = ((i & ~(jj - 1)) << 1) | (i & (jj - 1)), b = a | jj;
                        const u64 x = K[a], yv = K[b];
                        if ((x > yv) == ((a & k) == 0)) { K[a] = yv; K[b] = x; }
                    }
                    __syncthreads();
                }
        }
        if constexpr (PROF) { if (tid == 0) { B.prof[14] += B.fallback; B.prof[15] += n; } }
    }
    PB_STAMP(kProfSort);
    // ---- equal sums: list order (pb_visit_less); one thread per run of equal sums
    for (int i = tid; i + 1 < n; i += NT) {
        const unsigned si = (unsigned)(K[i] >> 32);
        if ((i == 0 || (unsigned)(K[i - 1] >> 32) != si) && (unsigned)(K[i + 1] >> 32) == si) {
            int g = 2;
            while (i + g < n && g <= kPbMaxTie && (unsigned)(K[i + g] >> 32) == si) ++g;
            if (g > kPbMaxTie) { B.degenerate = 1; continue; }
            for (int a = 1; a < g; ++a) {
                const u64 ka = K[i + a];
                const PbTep ta = pb_tep(tab, (int)(unsigned)ka);
                int b = a;
                while (b > 0 && pb_visit_less(L.w, ta, pb_tep(tab, (int)(unsigned)K[i + b - 1]))) { K[i + b] = K[i + b - 1]; --b; }
                K[i + b] = ka;
            }
        }
    }
    if (tid == 0) { B.gstop = 0x7FFFFFFF; B.reason = 0; B.ones = 0; B.nev = 0; B.nnb = 0; B.lnb = -1; }
    __syncthreads();
    PB_STAMP(kProfTie);
    if (B.degenerate) return 2;
    // ---- evaluate: thread t owns the entries [t per, (t+1) per) of the sorted chunk and keeps them in registers
    // (table entry, discrepancy, cost, frontier growth) from the evaluation to the sequential rules
    constexpr int PER = CAP / NT;
    const int per = (n + NT - 1) / NT;
    const int i0 = tid * per, i1 = (i0 + per) < n ? (i0 + per) : n;
    u64 kq[PER], Dq[PER];
    uchar4 tq4[PER];
    float cq[PER];
    int dq[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) kq[q] = (q < per && i0 + q < n) ? K[i0 + q] : 0ull;
#pragma unroll
    for (int q = 0; q < PER; ++q) tq4[q] = tab[(unsigned)kq[q] & 0xFFFFu];        // independent loads, issued together
    float tmin = __builtin_inff();
    int tdel = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const bool valid = q < per && i0 + q < n;
        const PbTep t{tq4[q].x, tq4[q].y, tq4[q].z, tq4[q].w};
        u64 E;
        pb_apply(L, t, d0, Dq[q], E);
        cq[q] = valid ? tep_cost_wide(L, __uint_as_float((unsigned)(kq[q] >> 32)), Dq[q]) : __builtin_inff();
        dq[q] = valid ? pb_delta(t, P.order) : 0;
        tmin = __builtin_fminf(tmin, cq[q]);
        tdel += dq[q];
    }
    // exclusive scans over the threads: min of the costs / sum of the frontier growth before my entries
    const float imin = wave_incl_min(tmin, lane);
    const int iadd = wave_incl_add(tdel, lane);
    if (lane == 63) { B.red_f[0][wave] = imin; B.red_i[0][wave] = iadd; }
    __syncthreads();
    float before = __shfl_up(imin, 1, 64);
    if (lane == 0) before = __builtin_inff();
    int nlb = iadd - tdel, tot_del = 0;
    for (int w = 0; w < W; ++w) {
        if (w < wave) { before = __builtin_fminf(before, B.red_f[0][w]); nlb += B.red_i[0][w]; }
        tot_del += B.red_i[0][w];
    }
    before = __builtin_fminf(before, B.best);
    nlb += B.nlive;
    PB_STAMP(kProfEval1);
    // ---- the sequential rules on my entries, assuming no earlier stop
    int ones = 0, nev = 0, nnb = 0, lnb = -1, lstop = 0x7FFFFFFF, lreason = 0;
    float lbest = 0.0f;
    u64 lD = 0;
    uchar4 lt = make_uchar4(0, 0, 0, 0);
    // rule 1 of all my entries first -- independent of each other once "the best before entry q" (a running minimum of the
    // costs: up to the first stop every entry is evaluated) is known, so their exp / divide / CDF reads overlap -- then
    // the sequential pass
    bool npq[PER];
    float w1q[PER];
    {
        float bef = before;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            npq[q] = pb_not_promising(__uint_as_float((unsigned)(kq[q] >> 32)), bef, Fr, P.c4, B.cdfA, B.cdfH, w1q[q]);
            bef = __builtin_fminf(bef, cq[q]);       // (an invalid entry has cost +inf)
        }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (q < per && i0 + q < n && lstop == 0x7FFFFFFF) {
            ones += nlb == 1;
            nlb += dq[q];
            if (npq[q]) { lstop = i0 + q; lreason = 1; }
            else {
                ++nev;
                if (cq[q] < before) {
                    before = cq[q]; lnb = i0 + q; ++nnb; lbest = cq[q]; lD = Dq[q]; lt = tq4[q];
                    if (pb_success(Dq[q], w1q[q], B.tq, Fr)) { lstop = i0 + q; lreason = 2; }
                }
            }
        }
    }
    if (lstop != 0x7FFFFFFF) atomicMin(&B.gstop, lstop);
    __syncthreads();
    PB_STAMP(kProfEval2);
    const int gstop = B.gstop;
    {   // my entries count if they lie before (or contain) the first stop; wave sums first, one atomic set per wavefront
        const bool mine = i0 < i1 && i0 <= gstop;
        int o = mine ? ones : 0, e = mine ? nev : 0, b = mine ? nnb : 0, l = mine ? lnb : -1;
        o = wave_add_i32(o); e = wave_add_i32(e); b = wave_add_i32(b); l = wave_max_i32(l);
        if (lane == 0) { atomicAdd(&B.ones, o); atomicAdd(&B.nev, e); atomicAdd(&B.nnb, b); atomicMax(&B.lnb, l); }
        if (mine && lstop == gstop) B.reason = lreason;
    }
    __syncthreads();
    if (lnb >= 0 && lnb == B.lnb && i0 <= gstop) {   // the last improvement before the stop is mine
        u64 E = 1ull << lt.x;
        if (lt.w > 1) E |= 1ull << lt.y;
        if (lt.w > 2) E |= 1ull << lt.z;
        B.best = lbest; B.bestD = lD; B.bestE = E; B.bestidx = B.j + lnb + 1;
    }
    __syncthreads();
    if (tid == 0) {
        const int npop = gstop != 0x7FFFFFFF ? gstop + 1 : n;
        B.cmp += 2 * npop - B.ones; B.suc1 += B.nev; B.suc2 += B.nnb;
    }
    if constexpr (PROF) { if (tid == 0) B.prof[kProfChunks] += 1; }
    if (gstop != 0x7FFFFFFF) { stop = B.reason; ntep = B.j + gstop + 1; __syncthreads(); PB_STAMP(kProfCombine); return 1; }
    __syncthreads();
    if (tid == 0) { B.j += n; B.nlive += tot_del; }
    __syncthreads();
    PB_STAMP(kProfCombine);
    return 0;
}

// search state handed from the stage-A kernel to the stage-B kernel
struct PbCarry {
    float lo, best;
    int j, nlive, cmp, suc1, suc2, bestidx;
    u64 bestD, bestE;
};

// per-frame set-up of the workgroup kernels: wavefront 0 prepares the frame, all wavefronts build the byte LUTs
template <int NT, int CAP>
__device__ __forceinline__ SearchFrame pb_block_setup(PbBlockLds<NT, CAP> &B, const float *__restrict__ y, long long src, long long f,
                                                      const unsigned char *__restrict__ perm_in, const u64 *__restrict__ parity_in,
                                                      const PbParams &P, const PbPrep *__restrict__ prep,
                                                      int tid)
{
    SearchLds &L = B.s;
    const int lane = tid & 63, wave = tid >> 6;
    SearchFrame S{};
    if (wave == 0) {
        S = search_prepare_regs<false>(L, y, src, perm_in[f * 128 + lane], perm_in[f * 128 + 64 + lane], parity_in[f * 64 + lane], lane);
        if (lane == 0) { B.d0 = S.d0; B.degenerate = 0; }
    } else if (prep && wave == 1) {   // meanwhile: what pb_singles_kernel already computed for this frame
        const PbPrep &R = prep[f];
        B.q[lane] = R.q[lane]; B.q[lane + 64] = R.q[lane + 64];
        B.cdfA[lane] = R.cdfA[lane];
        if (lane == 0) { B.cdfA[64] = R.cdfA[64]; B.fr = R.fr; }
    }
    __syncthreads();
    for (int b = wave; b < 8; b += NT / 64) build_byte_luts<1>(L.lut + b, &L.w[64 + 8 * b], lane);
    if (wave == 0) {
        if (!prep) {
            const PbFrame Fr = pb_frame_setup(L, B.q, B.cdfA, P.c4, P.order, P.nmax, lane);
            if (lane == 0) B.fr = Fr;
        }
        pb_success_terms(B.q, B.tq, lane);
    }
    __syncthreads();
    return S;
}

// Stage A of a frame of list A: the first two chunks of the visit order.  A frame on which no rule fires here goes on
// to list C with its search state (pb_heavy_kernel), massive ties to list B (list replay).
template <int NT, int CAP, bool PROF, int MINW = 1>
__global__ __launch_bounds__(NT, MINW) void pb_block_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                      const unsigned char *__restrict__ perm_in,
                                                      const u64 *__restrict__ parity_in, PbParams P,
                                                      const double *__restrict__ cdf_half,
                                                      const uchar4 *__restrict__ tab, int *__restrict__ ctl,
                                                      const int *__restrict__ listA, int *__restrict__ listB,
                                                      int *__restrict__ listC, int sub_cap, PbCarry *__restrict__ carry,
                                                      const PbPrep *__restrict__ prep, PbOut O,
                                                      unsigned long long *__restrict__ prof_out)
{
    __shared__ PbBlockLds<NT, CAP> B;
    SearchLds &L = B.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // frames are drawn by ticket from the sub-lists of list A, starting with sub-list (workgroup mod 16) and moving on when
    // one is exhausted; every sub-list has its own ticket word
    if (tid < kPbSub) B.sublen[tid] = ctl[kPbCtlLenA + kPbCtlLine * tid];
    if (tid == 0) { B.sub = blockIdx.x & (kPbSub - 1); B.subtried = 0; }
    const int nall = P.order > 2 ? kPbTabSize : (P.order > 1 ? kPbTriples0 : kPbPairs0);
    if (tid < 65) B.cdfH[tid] = cdf_half[tid];
    PbItems<NT> I;
    pb_items_static(I, tab, P.order, tid);
    if constexpr (PROF) { if (tid < 24) B.prof[tid] = 0; if (tid == 0) B.prof_last = __builtin_amdgcn_s_memtime(); }

    for (;;) {
        __syncthreads();
        if (tid == 0) {
            int tk = -1;
            while (B.subtried < kPbSub) {
                tk = atomicAdd(&ctl[kPbCtlTicketA + kPbCtlLine * B.sub], 1);
                if (tk < B.sublen[B.sub]) break;
                tk = -1; B.sub = (B.sub + 1) & (kPbSub - 1); ++B.subtried;
            }
            B.ticket = tk;
        }
        __syncthreads();
        const int tk = B.ticket;
        if (tk < 0) break;
        const long long f = listA[B.sub * sub_cap + tk];
        const long long src = index ? index[f] : f;
        const SearchFrame S = pb_block_setup(B, y, src, f, perm_in, parity_in, P, prep, tid);
        if (tid == 0) {
            B.lo = -1.0f; B.hi_cur = L.w[0];
            B.j = 0; B.nlive = 1; B.cmp = 0; B.suc1 = 0; B.suc2 = 0; B.bestidx = 0;
            B.bestD = B.d0; B.bestE = 0;
            B.best = tep_cost(L, 0.0f, B.d0);
        }
        __syncthreads();
        PB_STAMP(kProfSetup);
        unsigned long long fstart = 0, fchunks = 0;
        if constexpr (PROF) { fstart = __builtin_amdgcn_s_memtime(); fchunks = B.prof[kProfChunks]; }
        const PbFrame Fr = B.fr;
        const u64 d0 = B.d0;
        int stop = 0, ntep = P.nmax, state = 0;   // state: 0 = searching, 1 = a rule fired, 2 = hand the frame to the list replay

        // Two chunks of increasing sums here (aimed at ~512, then ~1536 TEPs: 88 % / 96 % of the frames that reach this
        // kernel stop inside them); a frame that is still searching then goes to stage B with its state.
        pb_items_frame(I, L.w, -1.0f);
        const float smax = P.order > 2 ? (L.w[0] + L.w[1]) + L.w[2] : (P.order > 1 ? L.w[0] + L.w[1] : L.w[0]);
        float lo = -1.0f;
        int done = 0;
        for (int chunk = 0; chunk < 2 && state == 0 && done < nall; ++chunk) {
            int n, run;
            const float T = pb_pick_bound(B, I, L.w, P.order, lo, done, nall, chunk == 0 ? P.t1 : P.t2, lane, wave, n, run);
            PB_STAMP(kProfHist);
            if (n < 0) { state = 2; break; }
            pb_items_write(B, I, L.w, run);
            __syncthreads();
            PB_STAMP(kProfPassA);
            state = pb_process_chunk<NT, CAP, PROF>(B, tab, P, Fr, d0, n, lo < 0.0f ? L.w[63] : lo, T < __builtin_inff() ? T : smax, tid, stop, ntep);
            lo = T;
            done += n;
        }
        if (state == 0 && tid == 0) B.lo = lo;
        __syncthreads();
        if constexpr (PROF) {
            if (tid == 0) {
                const unsigned long long dt = __builtin_amdgcn_s_memtime() - fstart, dc = B.prof[kProfChunks] - fchunks;
                B.prof[21] = dt > B.prof[21] ? dt : B.prof[21];
                B.prof[22] = dc > B.prof[22] ? dc : B.prof[22];
            }
        }
        if (state == 2) {   // massive ties: the literal list replay decodes this frame
            if (tid == 0) listB[atomicAdd(&ctl[kPbCtlLenB], 1)] = (int)f;
            continue;
        }
        if (state == 0 && done < nall) {   // no rule fired so far: the search goes on over the remaining TEPs
            if (tid == 0) {
                PbCarry c;
                c.lo = B.lo; c.best = B.best; c.j = B.j; c.nlive = B.nlive; c.cmp = B.cmp; c.suc1 = B.suc1; c.suc2 = B.suc2;
                c.bestidx = B.bestidx; c.bestD = B.bestD; c.bestE = B.bestE;
                carry[f] = c;
                const int sc = (int)f & (kPbSub - 1);
                listC[sc * sub_cap + atomicAdd(&ctl[kPbCtlLenC + kPbCtlLine * sc], 1)] = (int)f;
            }
            continue;
        }
        if (wave == 0)
            pb_write(L, S, O, f, lane, B.bestE, B.bestD, B.best, B.bestidx, ntep, B.cmp, B.suc1, B.suc2, stop);
        if constexpr (PROF) { if (tid == 0) B.prof[kProfFrames] += 1; }
        PB_STAMP(kProfFinish);
    }
    if constexpr (PROF) { __syncthreads(); if (tid < 21) atomicAdd(&prof_out[tid], B.prof[tid]); else if (tid < 24) atomicMax(&prof_out[tid], B.prof[tid]); }
}

// Stage B of a frame of list C: the visit order continues above the bound stage A reached, in chunks of up to 4096
// TEPs, until a rule fires or the table is exhausted (a full scan of 43 744 TEPs is ~15 chunks).
template <int NT, int CAP, bool PROF, int MINW = 1>
__global__ __launch_bounds__(NT, MINW) void pb_heavy_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                      const unsigned char *__restrict__ perm_in,
                                                      const u64 *__restrict__ parity_in, PbParams P,
                                                      const double *__restrict__ cdf_half,
                                                      const uchar4 *__restrict__ tab, int *__restrict__ ctl,
                                                      const int *__restrict__ listC, int *__restrict__ listB, int sub_cap,
                                                      const PbCarry *__restrict__ carry, const PbPrep *__restrict__ prep, PbOut O,
                                                      unsigned long long *__restrict__ prof_out)
{
    constexpr int W = NT / 64;
    // (dynamic LDS: a hipGraph kernel node with more than 64 KiB of STATIC LDS aborts at replay on ROCm 7.2; the size is
    //  registered once in pb_ctx_init)
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_heavy_lds[];
    PbBlockLds<NT, CAP> &B = *reinterpret_cast<PbBlockLds<NT, CAP> *>(pb_heavy_lds);
    SearchLds &L = B.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // tickets run over the concatenation of the 16 sub-lists of list C (a couple of thousand draws per call)
    int cbase[kPbSub + 1];
    cbase[0] = 0;
#pragma unroll
    for (int q = 0; q < kPbSub; ++q) cbase[q + 1] = cbase[q] + ctl[kPbCtlLenC + kPbCtlLine * q];
    const int nlist = cbase[kPbSub];
    const int nall = P.order == 2 ? kPbTriples0 : kPbTabSize;
    if (tid < 65) B.cdfH[tid] = cdf_half[tid];
    PbItems<NT> I;
    pb_items_static(I, tab, P.order, tid);
    if constexpr (PROF) { if (tid < 24) B.prof[tid] = 0; if (tid == 0) B.prof_last = __builtin_amdgcn_s_memtime(); }

    for (;;) {
        __syncthreads();
        if (tid == 0) B.ticket = atomicAdd(&ctl[kPbCtlTicketC], 1);
        __syncthreads();
        const int tk = B.ticket;
        if (tk >= nlist) break;
        int sc = 0;
#pragma unroll
        for (int q = 1; q < kPbSub; ++q) sc += tk >= cbase[q];
        int cb = 0;
#pragma unroll
        for (int q = 0; q < kPbSub; ++q) cb = q == sc ? cbase[q] : cb;
        const long long f = listC[sc * sub_cap + (tk - cb)];
        const long long src = index ? index[f] : f;
        const SearchFrame S = pb_block_setup(B, y, src, f, perm_in, parity_in, P, prep, tid);
        if (tid == 0) {
            const PbCarry c = carry[f];
            B.lo = c.lo; B.best = c.best; B.j = c.j; B.nlive = c.nlive; B.cmp = c.cmp; B.suc1 = c.suc1; B.suc2 = c.suc2;
            B.bestidx = c.bestidx; B.bestD = c.bestD; B.bestE = c.bestE;
        }
        __syncthreads();
        PB_STAMP(kProfSetup);
        const PbFrame Fr = B.fr;
        const u64 d0 = B.d0;
        int stop = 0, ntep = P.nmax, state = 0;
        // ---- chunks of increasing sums (aimed at 3/4 of the capacity) until a rule fires or the table is exhausted
        float lo = B.lo;
        int done = B.j;
        pb_items_frame(I, L.w, lo);
        const float smax = P.order > 2 ? (L.w[0] + L.w[1]) + L.w[2] : L.w[0] + L.w[1];
        while (state == 0 && done < nall) {
            int n, run;
            const float T = pb_pick_bound(B, I, L.w, P.order, lo, done, nall, P.t3, lane, wave, n, run);
            PB_STAMP(kProfHist);
            if (n < 0) { state = 2; break; }
            pb_items_write(B, I, L.w, run);
            __syncthreads();
            PB_STAMP(kProfGather);
            state = pb_process_chunk<NT, CAP, PROF>(B, tab, P, Fr, d0, n, lo < 0.0f ? L.w[63] : lo, T < __builtin_inff() ? T : smax, tid, stop, ntep);
            lo = T;
            done += n;
        }
        __syncthreads();
        if (state == 2) {
            if (tid == 0) listB[atomicAdd(&ctl[kPbCtlLenB], 1)] = (int)f;
            continue;
        }
        if (wave == 0)
            pb_write(L, S, O, f, lane, B.bestE, B.bestD, B.best, B.bestidx, ntep, B.cmp, B.suc1, B.suc2, stop);
        if constexpr (PROF) { if (tid == 0) B.prof[kProfFrames] += 1; }
        PB_STAMP(kProfFinish);
    }
    if constexpr (PROF) { __syncthreads(); if (tid < 21) atomicAdd(&prof_out[tid], B.prof[tid]); else if (tid < 24) atomicMax(&prof_out[tid], B.prof[tid]); }
}

// ---------------------------------------------------------------------------------------
// stage 3: literal replay of the frontier list, one frame of list B per wavefront (frames whose sums tie
// massively, or every frame when the caller asks for this path as a cross-check)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pb_seq_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                     const unsigned char *__restrict__ perm_in,
                                                     const u64 *__restrict__ parity_in, PbParams P,
                                                     const double *__restrict__ cdf_half,
                                                     PbEntry *__restrict__ spill_all, long long spill_stride,
                                                     int *__restrict__ ctl, const int *__restrict__ listB, PbOut O)
{
    __shared__ SearchLds lds[4];
    __shared__ PbLds pbl[4];
    const int lane = threadIdx.x & 63;
    SearchLds &L = lds[threadIdx.x >> 6];
    PbLds &B = pbl[threadIdx.x >> 6];
    const int nlist = ctl[kPbCtlLenB];
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    PbEntry *spill = spill_all + wave * spill_stride;
    B.cdfH[lane] = cdf_half[lane];
    if (lane == 0) B.cdfH[64] = cdf_half[64];
    wave_fence();

    // frames are handed out through a device counter: run times differ by orders of magnitude between frames
    for (;;) {
        int fq = 0;
        if (lane == 0) fq = atomicAdd(&ctl[kPbCtlTicketB], 1);
        const int tk = __builtin_amdgcn_readfirstlane(fq);
        if (tk >= nlist) break;
        const long long f = listB[tk];
        const long long src = index ? index[f] : f;
        const SearchFrame S = search_prepare(L, y, src, perm_in, parity_in, f, lane);
        const PbFrame Fr = pb_frame_setup(L, B.q, B.cdfA, P.c4, P.order, P.nmax, lane);
        const float spl = Fr.spl, lrb_mean = Fr.lrb_mean;
        const double p_t_suc = Fr.p_t_suc, p_t_pro = Fr.p_t_pro;
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
        pb_write(L, S, O, f, lane, bestE, bestD, best, bestidx, ntep, cmp, suc1, suc2, stop);
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
// the control words of a call start at zero (a one-wavefront kernel, not hipMemsetAsync: captured into a hipGraph the
// memset node did not clear the words on the second and later replays -- ROCm 7.2 -- and the tickets ran on)
__global__ __launch_bounds__(64) void pb_ctl_clear_kernel(int *__restrict__ ctl)
{
    for (int i = threadIdx.x; i < kPbCtlInts; i += 64) ctl[i] = 0;
}

int pb_ctx_init(ldpc_ctx *ctx)
{
    OsdState *st = state(ctx);
    LDPC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_heavy_kernel<1024, 4096, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(PbBlockLds<1024, 4096>)));
    LDPC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_heavy_kernel<1024, 4096, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(PbBlockLds<1024, 4096>)));
    LDPC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_heavy_kernel<512, 2048, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(PbBlockLds<512, 2048>)));
    std::vector<uchar4> tab;
    tab.reserve(kPbTabSize);
    for (int p = 63; p >= 0; --p) tab.push_back(make_uchar4((unsigned char)p, 0, 0, 1));
    for (int p0 = 62; p0 >= 0; --p0)
        for (int p1 = p0 + 1; p1 < 64; ++p1) tab.push_back(make_uchar4((unsigned char)p0, (unsigned char)p1, 0, 2));
    for (int p0 = 61; p0 >= 0; --p0)
        for (int p1 = p0 + 1; p1 < 63; ++p1)
            for (int p2 = p1 + 1; p2 < 64; ++p2) tab.push_back(make_uchar4((unsigned char)p0, (unsigned char)p1, (unsigned char)p2, 3));
    if ((int)tab.size() != kPbTabSize) return fail(LDPC_E_CODE, "PB-OSD table has %zu entries", tab.size());
    LDPC_HIP(hipMalloc((void **)&st->d_pb_tab, sizeof(uchar4) * tab.size()));
    LDPC_HIP(hipMemcpy(st->d_pb_tab, tab.data(), sizeof(uchar4) * tab.size(), hipMemcpyHostToDevice));
    return LDPC_OK;
}

// PB-OSD part of a stream's workspace: control words, the two frame lists, the list replay's spill areas
static int stream_ws_pb(ldpc_ctx *ctx, hipStream_t s, int64_t frames, int64_t spill_stride, StreamWs **out)
{
    OsdState *st = state(ctx);
    std::lock_guard<std::mutex> lock(st->mu);
    StreamWs &w = st->ws[s];
    const bool grow_list = frames > w.pb_cap || !w.d_pb_ctl, grow_spill = spill_stride > w.pb_spill_stride;
    if ((grow_list || grow_spill) && stream_capturing(s))
        return fail(LDPC_E_NOMEM, "PB-OSD workspace of this stream must be sized before capturing (run one call on the stream first)");
    if (grow_list) {
        (void)hipFree(w.d_pb_list); w.d_pb_list = nullptr; w.pb_cap = 0;
        if (!w.d_pb_ctl && hipMalloc((void **)&w.d_pb_ctl, sizeof(int) * kPbCtlInts) != hipSuccess)
            return fail(LDPC_E_NOMEM, "PB-OSD control words could not be allocated");
        (void)hipFree(w.d_pb_carry); w.d_pb_carry = nullptr;
        (void)hipFree(w.d_pb_prep); w.d_pb_prep = nullptr;
        const int64_t sub_cap = (frames + kPbSub - 1) / kPbSub;
        if (hipMalloc((void **)&w.d_pb_list, sizeof(int) * 3 * (size_t)kPbSub * (size_t)sub_cap) != hipSuccess ||
            hipMalloc(&w.d_pb_carry, sizeof(PbCarry) * (size_t)frames) != hipSuccess ||
            hipMalloc(&w.d_pb_prep, sizeof(PbPrep) * (size_t)frames) != hipSuccess)
            return fail(LDPC_E_NOMEM, "PB-OSD frame lists for %lld frames could not be allocated", (long long)frames);
        w.pb_cap = frames; w.pb_sub_cap = sub_cap;
    }
    if (grow_spill) {
        (void)hipFree(w.d_pb_spill); w.d_pb_spill = nullptr; w.pb_spill_stride = 0;
        if (hipMalloc(&w.d_pb_spill, sizeof(PbEntry) * (size_t)spill_stride * kPbSeqBlocks * 4) != hipSuccess)
            return fail(LDPC_E_NOMEM, "PB-OSD frontier workspace (%lld entries per wave) could not be allocated", (long long)spill_stride);
        w.pb_spill_stride = spill_stride;
    }
    *out = &w;
    return LDPC_OK;
}

int launch_pb(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
              const unsigned char *d_perm, const u64 *d_parity, const ldpc_osd_params *p, uint64_t *d_cw, float *d_metric,
              int32_t *d_best, int32_t *d_ntep, hipStream_t s)
{
    OsdState *st = state(ctx);
    const int64_t nmax = st->ntep[p->order];
    // list replay: the list is append-only, at most 1 + 2 (N_max - 1) slots; spilled slots first, spilled chunk minima after
    const int64_t slots = 2 * nmax + 2;
    if (slots > (int64_t)kPbSuper * 4096) return fail(LDPC_E_UNSUPPORTED, "ldpc_osd_decode: PB-OSD list of %lld slots exceeds the kernel's limit", (long long)slots);
    const int64_t spill_slots = slots > kPbLdsSlots ? slots - kPbLdsSlots : 0;
    const int64_t stride = spill_slots + (slots / 64 + 2) + 2;
    StreamWs *w;
    int rc = stream_ws_pb(ctx, s, F, stride, &w);
    if (rc) return rc;
    PbParams pp;
    pp.order = p->order; pp.nmax = (int)nmax; pp.cmin_off = spill_slots;
    pp.t1 = 768; pp.t2 = 768; pp.t3 = 3072;   // 3/4 of the chunk capacities (measured flat between 512 and 1536 for stage A)
    pp.c4 = (float)(-4.0 * (1.0 / pow(10.0, (double)p->snr_db / 10.0)));    // -4 * noise_variance, pb_testing.py:50-52
    const int mode = (p->reserved & 4) ? 2 : ((p->reserved & 2) ? 1 : 0);
    PbOut O{reinterpret_cast<u64 *>(d_cw), d_metric, d_best, d_ntep, reinterpret_cast<int *>(p->d_aux)};
    const int64_t list_len = (int64_t)kPbSub * w->pb_sub_cap;   // >= pb_cap
    int *listA = w->d_pb_list, *listB = w->d_pb_list + list_len, *listC = w->d_pb_list + 2 * list_len;
    const int sub_cap = (int)w->pb_sub_cap;
    PbPrep *prep_w = reinterpret_cast<PbPrep *>(w->d_pb_prep);
    hipLaunchKernelGGL(pb_ctl_clear_kernel, dim3(1), dim3(64), 0, s, w->d_pb_ctl);
    const int64_t want = (F + 3) / 4;
    const unsigned g1 = (unsigned)(F < 1 ? 1 : (F < 32768 ? F : 32768));
    static const bool profile_s = getenv("LDPC_PB_PROFILE") != nullptr;
    if (!profile_s) {
        hipLaunchKernelGGL(pb_singles_kernel<false>, dim3(g1), dim3(64), 0, s, d_y, d_index, d_count, (long long)F, d_perm, d_parity, pp, mode,
                           st->d_cdf_half, w->d_pb_ctl, listA, listB, sub_cap, prep_w, O, (unsigned long long *)nullptr);
    } else {
        static unsigned long long *d_ps = nullptr;
        if (!d_ps) LDPC_HIP(hipMalloc((void **)&d_ps, sizeof(unsigned long long) * 8));
        LDPC_HIP(hipMemsetAsync(d_ps, 0, sizeof(unsigned long long) * 8, s));
        hipLaunchKernelGGL(pb_singles_kernel<true>, dim3(g1), dim3(64), 0, s, d_y, d_index, d_count, (long long)F, d_perm, d_parity, pp, mode,
                           st->d_cdf_half, w->d_pb_ctl, listA, listB, sub_cap, prep_w, O, d_ps);
        unsigned long long h[8];
        LDPC_HIP(hipMemcpyAsync(h, d_ps, sizeof(h), hipMemcpyDeviceToHost, s));
        LDPC_HIP(hipStreamSynchronize(s));
        fprintf(stderr, "[LDPC_PB_PROFILE] singles kernel, cycles of lane 0 summed over wavefronts: start+perm/P' loads=%llu y loads+LUT=%llu frame_setup=%llu rules=%llu hand_on=%llu write=%llu\n",
                h[5], h[0], h[1], h[2], h[3], h[4]);
    }
    // stage A: a multiple of 16 workgroups (one sixteenth of them per sub-list), at most the 1024 that are resident
    const unsigned g2 = (unsigned)(F < 1024 ? ((F + kPbSub - 1) / kPbSub) * kPbSub : 1024), g2b = (unsigned)(F < kPbHeavyGrid ? F : kPbHeavyGrid);
    PbCarry *carry = reinterpret_cast<PbCarry *>(w->d_pb_carry);
    const PbPrep *prep = mode == 0 ? prep_w : nullptr;    // (cross-check routes skip the stage that fills it)
    static const bool profile = getenv("LDPC_PB_PROFILE") != nullptr;   // diagnostic build of the two workgroup kernels
    if (!profile) {
        hipLaunchKernelGGL((pb_block_kernel<256, 1024, false, 4>), dim3(g2), dim3(256), 0, s, d_y, d_index, d_perm, d_parity, pp,
                           st->d_cdf_half, st->d_pb_tab, w->d_pb_ctl, listA, listB, listC, sub_cap, carry, prep, O, (unsigned long long *)nullptr);
        // Two shapes of the stage-B kernel.  Few, long searches (the usual case from ~2.25 dB up): the launch lasts as long
        // as its longest frames, and one 1024-thread workgroup per CU with 4096-TEP chunks gets a frame through fastest.
        // Many searches (low SNR: at 1.0 dB 83 % of the frames reach the OSD and most of them this stage): throughput counts,
        // and two 512-thread workgroups per CU with 2048-TEP chunks interleave their barrier-separated phases -- +29 % at
        // 1.0 dB, +21 % at 1.5 dB, -9 % at 3.0 dB.  The SNR the caller decodes for tells the two regimes apart.
        if (p->snr_db < 2.25f) {
            pp.t3 = 1792;
            hipLaunchKernelGGL((pb_heavy_kernel<512, 2048, false, 4>), dim3(g2b), dim3(512), sizeof(PbBlockLds<512, 2048>), s, d_y, d_index, d_perm,
                               d_parity, pp, st->d_cdf_half, st->d_pb_tab, w->d_pb_ctl, listC, listB, sub_cap, carry, prep, O, (unsigned long long *)nullptr);
        } else {
            hipLaunchKernelGGL((pb_heavy_kernel<1024, 4096, false>), dim3(g2b), dim3(1024), sizeof(PbBlockLds<1024, 4096>), s, d_y, d_index, d_perm,
                               d_parity, pp, st->d_cdf_half, st->d_pb_tab, w->d_pb_ctl, listC, listB, sub_cap, carry, prep, O, (unsigned long long *)nullptr);
        }
    } else {
        static unsigned long long *d_prof = nullptr;
        if (!d_prof) LDPC_HIP(hipMalloc((void **)&d_prof, sizeof(unsigned long long) * 48));
        LDPC_HIP(hipMemsetAsync(d_prof, 0, sizeof(unsigned long long) * 48, s));
        hipLaunchKernelGGL((pb_block_kernel<256, 1024, true, 4>), dim3(g2), dim3(256), 0, s, d_y, d_index, d_perm, d_parity, pp,
                           st->d_cdf_half, st->d_pb_tab, w->d_pb_ctl, listA, listB, listC, sub_cap, carry, prep, O, d_prof);
        hipLaunchKernelGGL((pb_heavy_kernel<1024, 4096, true>), dim3(g2b), dim3(1024), sizeof(PbBlockLds<1024, 4096>), s, d_y, d_index, d_perm, d_parity, pp,
                           st->d_cdf_half, st->d_pb_tab, w->d_pb_ctl, listC, listB, sub_cap, carry, prep, O, d_prof + 24);
        unsigned long long h[48];
        LDPC_HIP(hipMemcpyAsync(h, d_prof, sizeof(h), hipMemcpyDeviceToHost, s));
        LDPC_HIP(hipStreamSynchronize(s));
        static const char *names[24] = {"setup", "write(A)", "bound", "write(B)", "sort", "tie", "eval1", "eval2", "combine", "items", "scatter",
                                        "finish", "FRAMES", "CHUNKS", "FALLBACKS", "KEYS", "s.minmax", "s.count", "s.scan", "s.scatter",
                                        "s.fix", "MAXFRAMECYC", "MAXFRAMECHUNKS", "-"};
        for (int k = 0; k < 2; ++k) {
            fprintf(stderr, "[LDPC_PB_PROFILE] %s kernel, cycles of thread 0 summed over workgroups:", k ? "stage-B" : "stage-A");
            for (int q = 0; q < 23; ++q) fprintf(stderr, " %s=%llu", names[q], h[24 * k + q]);
            fprintf(stderr, "\n");
        }
    }
    const unsigned g3 = (unsigned)(want < kPbSeqBlocks ? (want < 1 ? 1 : want) : kPbSeqBlocks);
    hipLaunchKernelGGL(pb_seq_kernel, dim3(g3), dim3(256), 0, s, d_y, d_index, d_perm, d_parity, pp, st->d_cdf_half,
                       reinterpret_cast<PbEntry *>(w->d_pb_spill), (long long)w->pb_spill_stride, w->d_pb_ctl, listB, O);
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

}  // namespace ldpc
