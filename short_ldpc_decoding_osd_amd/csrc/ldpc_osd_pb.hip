// PB-OSD kernels for (128,64) codes on gfx950 (MI355X).
//
// Reference (paths relative to LDPC_128/): pb_osd, PB_OSD/pb_testing.py:100-149 -- best-first TEP generation
// from a growing frontier list (optimal_tep_sequence :366-397: "first minimum of the reliability sums in list
// order", pop, append <= 2 children) with two probabilistic stopping rules (acquire_prob_promising :448-461,
// acquire_p_e_suc :423-436, thresholds :485-500).  All probabilities follow the float conventions of
// oracle/ldpc_oracle.c orc_pb_osd: float32 with det_expf (IEEE + - * / only, host and device agree bit for
// bit), float64 binomial-CDF recurrences; the promising rule compares in float64 (both sides are float64 tensors in the
// reference, :129), the success rule in float32 (:145: a float32 tensor against a NumPy double, which TensorFlow casts down).
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
//                      while |y'_p| < |y'_62| + |y'_63| (the smallest weight-2 sum); one TEP per lane.  Two thirds
//                      of the frames stop here at 2.5 dB; the others are appended to list A, each with ONE 1536-byte record
//                      that holds everything its search needs (round 4).
//   pb_wave_kernel     one frame of list A per WAVEFRONT (round 3; rounds 1-2 used 256- and 1024-thread workgroups),
//                      continued from the head: chunks of <= 832 TEPs (4-byte keys, round 4), each = the members of a sum
//                      range walked directly from the sorted reliabilities, judged by a sort-free pass (pbw_scan4); what that
//                      cannot settle is redone over the same sum range with 8-byte keys, chunks of <= 384, and sorted if need
//                      be (pbw_redo_range, a real function).  No workgroup barrier anywhere.  Massive ties go to list B; a
//                      search that passes a budget of TEPs (chosen on the device from the number of searching frames) leaves
//                      with its state.  120 VGPRs, 10 KiB of LDS: four wavefronts per SIMD.
//   pb_coop_kernel     those long searches, one frame per 16-wavefront WORKGROUP: chunks of <= 4096 TEPs counted by
//                      bisection and generated once, every wavefront judging the keys it generated; one workgroup
//                      barrier per count and one per ordinary chunk.
//   pb_seq_kernel      the literal list replay (round-1 kernel) for list B.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ldpc_internal.h"
#include "ldpc_wave.h"
#include "ldpc_search.h"
#include "ldpc_front.h"
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
    int t1, t2;                  // chunk targets: first chunk / later chunks
    int t3, budget;              // chunk target of the workgroup kernel; TEPs after which a frame may be handed to it
    int budget_s, budget_m;      // ... when its sub-list is short (< 128 frames) / of medium length (< 448)
    int budget_l, budget_xl;     // ... long (1400 .. 3000) / very long
    int late_min, late_maxlen, late_pct, late_div;   // once late_pct % of a sub-list's frames have STARTED (lists of more than late_min frames in all, sub-lists shorter than late_maxlen) a search leaves after budget / late_div
    int handoff_maxlen;          // ... if its sub-list of list A holds fewer frames than this (many searches: throughput counts, none leaves)
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
// the two thresholds.  w = |y'| (LDS) must be in place; q / cdfA are per-frame LDS arrays.
// `pairs` (optional, [4][64] float2 of LDS): the four chains' (multiplier, addend) per step, written side by side here so that a
// step of the chains is ONE instruction (see below); without it the operands are formed per step.
__device__ __forceinline__ PbFrame pb_frame_setup(const float *w, float *q, double *cdfA, float c4, int order, int nmax, int lane,
                                                  float best0 = __builtin_inff(), float2 *pairs = nullptr)
{
    {
        const float q0 = 1.0f / (1.0f + det_expf(-(c4 * w[lane]))), q1 = 1.0f / (1.0f + det_expf(-(c4 * w[lane + 64])));
        q[lane] = q0;
        q[lane + 64] = q1;
        if (pairs) {
            pairs[lane] = make_float2(1.0f, q1);                  // chain 0: sum of q over the parity part
            pairs[64 + lane] = make_float2(1.0f, w[lane + 64]);    // chain 1: sum of |y'| over the parity part
            pairs[128 + lane] = make_float2(1.0f, q0);            // chain 2: sum of q over the MRB
            pairs[192 + lane] = make_float2(1.0f - q0, 0.0f);     // chain 3: product of 1 - q over the MRB
        }
    }
    wave_fence();
    // sequential (ascending position) means / product, as the oracle defines them: four dependent chains of 64 steps.  Lanes 0..3
    // run one chain each with ONE fused multiply-add per step -- acc * 1 + x is the sum, acc * x + 0 the product, both rounded
    // once like the plain operations (-ffp-contract=off does not touch an explicit fma) -- instead of every lane running all four.
    float a1, aw, at, spl;
    if (pairs) {
        const float2 *const src = pairs + 64 * (lane & 3);
        float acc = (lane & 3) == 3 ? 1.0f : 0.0f;
#pragma unroll 8
        for (int p = 0; p < 64; ++p) {
            const float2 t = src[p];
            acc = __builtin_fmaf(acc, t.x, t.y);
        }
        a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
        aw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 1));
        at = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 2));
        spl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 3));
    } else {
        const int ch = lane & 3;
        const float *src = ch == 0 ? q + 64 : (ch == 1 ? w + 64 : q);
        const bool prod = ch == 3;
        float acc = prod ? 1.0f : 0.0f;
#pragma unroll 8
        for (int p = 0; p < 64; ++p) {
            const float x = src[p];
            const float b = prod ? 1.0f - x : 1.0f, c = prod ? 0.0f : x;
            acc = __builtin_fmaf(acc, b, c);
        }
        a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
        aw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 1));
        at = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 2));
        spl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 3));
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
// (cdfA / cdfH: float64 tables, or the same tables already rounded to float32 -- they are only read through the cast)
template <typename TA>
__device__ __forceinline__ float pb_promising_bs(float rs, float best, const PbFrame &F, float c4, const TA *cdfA, const TA *cdfH, float &w1_out)
{
    const float w1 = det_expf(c4 * rs) * F.spl, w2 = 1.0f - w1;
    const float bt = __builtin_floorf((best - rs) / F.lrb_mean);
    const int beta = bt > 0.0f ? (bt < 64.0f ? (int)bt : 64) : 0;
    float bs = 0.0f;
    bs = bs + w1 * (float)cdfA[beta];
    bs = bs + w2 * (float)cdfH[beta];
    w1_out = w1;
    return bs;
}
template <typename TA>
__device__ __forceinline__ bool pb_not_promising(float rs, float best, const PbFrame &F, float c4, const TA *cdfA,
                                                 const TA *cdfH, float &w1_out)
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

// the same from the table of q_p alone (the chunk kernels: half the LDS; the factor is formed per position, as pb_seq_kernel does)
__device__ __forceinline__ bool pb_success_q(u64 D, float w1, const float *qpar, const PbFrame &F)
{
    const float ratio = (1.0f - w1) / w1;
    float prod = 1.0f;
#pragma unroll 8
    for (int p = 0; p < 64; ++p) {
        const float qp = qpar[p];
        prod = prod * (((D >> p) & 1) ? 2.0f * qp : 2.0f * (1.0f - qp));
    }
    const float p_suc = 1.0f / (1.0f + ratio / prod);
    return p_suc > (float)F.p_t_suc;      // (pb_testing.py:145: TensorFlow compares the float32 tensor with the double cast TO float32)
}

// (D differs from lane to lane here: the factor is picked bitwise -- bit -> 0 / -1 by a signed field extract of the word's
//  half, then (y & m) | (x & ~m): three 32-bit instructions a position where (D >> p) & 1 compiled to a 64-bit shift, a 64-bit
//  compare and a select, five)
__device__ __forceinline__ bool pb_success(u64 D, float w1, const float2 *tq, const PbFrame &F)
{
    const float ratio = (1.0f - w1) / w1;
    float prod = 1.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int dw = (int)(unsigned)(h ? D >> 32 : D);
#pragma unroll 8
        for (int u = 0; u < 32; ++u) {
            const float2 t = tq[32 * h + u];
            const int m = __builtin_amdgcn_sbfe(dw, u, 1);
            prod = prod * __int_as_float((__float_as_int(t.y) & m) | (__float_as_int(t.x) & ~m));
        }
    }
    const float p_suc = 1.0f / (1.0f + ratio / prod);
    return p_suc > (float)F.p_t_suc;      // (pb_testing.py:145: TensorFlow compares the float32 tensor with the double cast TO float32)
}

// number of TEPs of weight 1, 1..2, 1..3 over 64 positions
constexpr int kPbPairs0 = 64, kPbTriples0 = 64 + 2016, kPbTabSize = 64 + 2016 + 41664;

struct PbTep {
    int p0, p1, p2, wt;
};
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
struct PbOut {
    u64 *cw; float *metric; int *best, *ntep, *aux;
};

// What pb_singles_kernel hands to the chunk kernel: ONE contiguous record per frame (1536 B) with everything the search
// needs, so that the receiving wavefront starts after a single wide load instead of the chain frame number -> source
// index -> permutation -> y (round 4; the record replaces the separate 1096-byte per-frame table of rounds 2-3):
//   words [0, 128)     w[128]     |y'|                                       } the head of PbWaveLds (behind its four pad
//   words [128, 256)   P[64]      rows of P'                                 } words): the chunk kernel copies these 360 words
//   words [256, 258)   0          the "row" an unused position of a key reads } into LDS as they are
//   words [258, 326)   cdfA[68]   P[Bin(64, p1) <= b] rounded to float32     }
//   words [326, 358)   perm[128]  original bit index of primed position p, one byte each (+ 2 pad words)
//   words [360, ...)   PbHead     the frame's scalars and the search state after the weight-1 head
// Derived on arrival (a few dozen instructions): the cost-bound table (sorted parity weights) and the success-rule factors.
struct PbHead {
    PbFrame fr;
    u64 d0, hm, hp;            // order-0 parity discrepancy, hard decisions of the MRB / parity part
    u64 hbestD, hbestE;        // the search state after the weight-1 head of the pop sequence (no rule fired on it)
    float hbest;
    int nhead, hsuc2, hbestidx;
};
constexpr int kPbR1Zero = 256, kPbR1Cdf = 258, kPbR1Perm = 326, kPbR1Head = 360, kPbR1Words = 384;     // (words [0, kPbR1Head) are copied into LDS)
static_assert(kPbR1Head * 4 % 8 == 0 && kPbR1Head * 4 + sizeof(PbHead) <= kPbR1Words * 4, "record layout");

template <class LDS>
__device__ __forceinline__ void pb_write(LDS &L, const SearchFrame &S, const PbOut &O, long long f, int lane, u64 bestE,
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

// ---------------------------------------------------------------------------------------
// stage 1: the weight-1 head of the pop sequence, one frame per wavefront, one TEP per lane
//   mode 0: normal; 1: every frame straight to list A (block kernel); 2: every frame to list B (list replay)
// ---------------------------------------------------------------------------------------
template <bool FUSED>
struct PbSinglesLdsT {
    SearchLdsLean s;     // no byte LUTs: the kernel evaluates two candidates per frame and lane, and it answers to occupancy
    double cdfA[65], cdfH[65];
    float q[128];
    union {
        float2 pairs[4][64];      // pb_frame_setup's chain operands ...
        float2 tq[64];            // ... then the success rule's factors
    };
};
// FUSED: the OSD front end of the frame runs in this kernel first (ldpc_osd_decode's route: nothing goes through a workspace);
// its scratch lies under the tables that are filled afterwards, the frame's channel row beside it.
template <>
struct PbSinglesLdsT<true> {
    SearchLdsLean s;
    double cdfH[65];
    float yrow[128];
    union {
        FrontLds front;
        struct {
            double cdfA[65];
            float q[128];
            union {
                float2 pairs[4][64];
                float2 tq[64];
            };
        };
    };
};

// (seven wavefronts per SIMD asked of the register allocator: 72 VGPRs, no scratch -- with the loads of a frame's start issued
//  together the kernel took 81 VGPRs and five per SIMD, 89 us instead of 84; at seven 80 us; at eight, 64 VGPRs + 9 spilled, 81 us)
// (one wavefront per workgroup, 4.8 KiB of LDS each: the register count decides how many are resident.  With the searches' 8 KiB
//  of LUTs it was 11.2 KiB and 14 per CU; padded to 10 per CU the kernel took 130 instead of 98 us per 33 k frames.)
template <bool PROF, bool FUSED = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7))) void pb_singles_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                         const int *__restrict__ count, long long F,
                                                         const unsigned char *__restrict__ perm_in,
                                                         const u64 *__restrict__ parity_in, const u64 *__restrict__ Gcols, PbParams P, int mode,
                                                         const double *__restrict__ cdf_half,
                                                         int *__restrict__ ctl, int *__restrict__ listA, int *__restrict__ listB, int sub_cap,
                                                         unsigned *__restrict__ recs, PbOut O, unsigned long long *__restrict__ prof_out)
{
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, plast = 0;
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();
#define PBS_STAMP(k) do { if constexpr (PROF) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); pt[k] += now__ - plast; plast = now__; } } while (0)
    __shared__ PbSinglesLdsT<FUSED> W;
    const int lane = threadIdx.x;
    SearchLdsLean &L = W.s;
    long long nframes = F;
    const long long wave = blockIdx.x;
    if (mode == 2) {   // every frame to the list replay, in frame order
        if (count) { const long long c = *count; nframes = c < F ? c : F; }
        for (long long f = wave * 64 + lane; f < nframes; f += (long long)gridDim.x * 64) listB[f] = (int)f;
        if (wave == 0 && lane == 0) ctl[kPbCtlLenB] = (int)nframes;
        return;
    }
    // A wavefront serves one frame (two for a few) and half of its life used to be the chain count -> frame number -> permutation
    // and P' -> y (in-kernel stamps, round 4): every load that does not depend on another is now issued before the first
    // is waited for -- the frame's operands are read for f < F (the buffers hold F frames) and dropped if the count says less.
    // (The wave-uniform words -- frame number, count -- go through VECTOR loads, an opaque zero in the address: a scalar load is
    //  waited for where it is issued once its result meets a branch, and three of them in a row were three round trips.)
    int vz = 0;
    asm volatile("" : "+v"(vz));
    for (long long f = wave; f < nframes; f += gridDim.x) {      // (nframes = F until the first frame's loads are out)
        int o1 = 0, o2 = 0;
        u64 Pr = 0;
        if constexpr (!FUSED) { o1 = perm_in[f * 128 + lane]; o2 = perm_in[f * 128 + 64 + lane]; Pr = parity_in[f * 64 + lane]; }
        int srcv = (int)f;
        if (index) srcv = index[f + vz];
        if (f == wave) {
            const double ch = cdf_half[lane], ch64 = cdf_half[64 + vz];
            int cv = 0x7FFFFFFF;
            if (count) cv = count[vz];
            W.cdfH[lane] = ch;
            if (lane == 0) W.cdfH[64] = ch64;
            const long long c = __builtin_amdgcn_readfirstlane(cv);
            nframes = c < F ? c : F;
            wave_fence();
        }
        if (f >= nframes) break;
        const long long src = __builtin_amdgcn_readfirstlane(srcv);
        if constexpr (PROF) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(o1), "+v"(o2), "+v"(Pr)); PBS_STAMP(5); }
        SearchFrame S;
        if constexpr (FUSED) {
            // the frame's row as it lies in memory (two coalesced loads), the front end on it, then y' = y[perm] out of LDS
            const float ya = y[src * 128 + lane], yb = y[src * 128 + 64 + lane];
            W.yrow[lane] = ya; W.yrow[64 + lane] = yb;
            const FrontResult fr = front_device_vals(W.front, __float_as_uint(ya) & 0x7FFFFFFFu, __float_as_uint(yb) & 0x7FFFFFFFu, Gcols, lane);
            wave_fence();
            S = search_prepare_vals<false>(L, W.yrow[fr.o1], W.yrow[fr.o2], fr.o1, fr.o2, fr.Prow, lane);
        } else {
            S = search_prepare_regs<false>(L, y, src, o1, o2, Pr, lane);
        }
        PBS_STAMP(0);
        const float best0 = tep_cost_direct_uniform(L.w, 0.0f, S.d0, lane);
        const PbFrame Fr = pb_frame_setup(L.w, W.q, W.cdfA, P.c4, P.order, P.nmax, lane, best0, &W.pairs[0][0]);
        PBS_STAMP(1);
        wave_fence();
        pb_success_terms(W.q, W.tq, lane);
        wave_fence();
        // lane l <-> TEP {63 - l}, visit index l; valid while its weight is below the smallest weight-2 sum
        // (mode 1, the cross-check route "every frame through the chunk kernel from its first TEP": no head at all)
        const int p = 63 - lane;
        const float rs = L.w[p];
        const float s2min = L.w[62] + L.w[63];
        const u64 vmask = mode == 1 ? 0ull : __ballot(P.order == 1 || rs < s2min);
        const int nhead = (~vmask) ? __builtin_ctzll(~vmask) : 64;
        const bool valid = lane < nhead;
        const u64 D = S.d0 ^ L.P[p];
        const float cost = valid ? tep_cost_direct(L.w, rs, D) : __builtin_inff();
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
        if (sm == 0 && (P.order > 1 || mode == 1)) {   // no rule fired on the head: the chunk kernel takes the frame, with ONE record
            unsigned *const R = recs + f * kPbR1Words;
            R[lane] = __float_as_uint(L.w[lane]); R[64 + lane] = __float_as_uint(L.w[64 + lane]);
            reinterpret_cast<u64 *>(R + 128)[lane] = L.P[lane];
            if (lane < 2) R[kPbR1Zero + lane] = 0u;
            R[kPbR1Cdf + lane] = __float_as_uint((float)W.cdfA[lane]);
            if (lane < 4) R[kPbR1Cdf + 64 + lane] = lane == 0 ? __float_as_uint((float)W.cdfA[64]) : 0u;
            if (lane < 32) R[kPbR1Perm + lane] = reinterpret_cast<const unsigned *>(L.perm)[lane];
            {   // the head's result: nhead TEPs popped and evaluated, the last improvement among them (if any)
                const u64 nbm = __ballot(newbest);
                float hb = best0;
                u64 hD = S.d0, hE = 0;
                int hidx = 0;
                if (nbm) {
                    const int lb = 63 - __builtin_clzll(nbm);
                    hb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cost), lb));
                    hD = readlane64(D, lb); hE = 1ull << (63 - lb); hidx = lb + 1;
                }
                if (lane == 0) {
                    PbHead h;
                    h.fr = Fr; h.d0 = S.d0; h.hm = S.hm; h.hp = S.hp; h.hbestD = hD; h.hbestE = hE;
                    h.hbest = hb; h.nhead = nhead; h.hsuc2 = __popcll(nbm); h.hbestidx = hidx;
                    *reinterpret_cast<PbHead *>(R + kPbR1Head) = h;
                    const int sl = (int)f & (kPbSub - 1);
                    listA[sl * sub_cap + atomicAdd(&ctl[kPbCtlLenA + kPbCtlLine * sl], 1)] = (int)f;
                }
            }
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
// stage 2: sorted chunks of the visit order, ONE FRAME PER WAVEFRONT (round 3)
//
// Round 2 ran this stage on 256- and 1024-thread workgroups: ~17 workgroup barriers per chunk, ~15 wave
// instructions per TEP, 60 % of the wave-cycles waiting (profiles/r03/pmc_counters_nms10_pb3_snr1.0_baseline_*).
// Here a frame belongs to ONE wavefront from its first chunk to its stop: no barrier anywhere (phases are
// separated by wave fences), the frame's search state lives in registers, ~9 frames are resident per CU and
// the dispatcher balances them (one wavefront per workgroup, compile-time LDS addresses).
//
// Direct enumeration of a sum range.  The MRB positions are sorted by reliability (w[0] >= w[1] >= ...) and float
// addition is monotone, so with the other positions fixed the sum of a TEP is non-increasing in its LAST position m.
// The TEPs are 2017 "items" -- the singles {m}; the pairs {i, m} of one i; the triples {i, j, m} of one (i, j) --
// inside each of which the members appear in the visit order by DESCENDING m.  Every lane owns 32 items:
//   q = 0..30  triples, by the DISTANCE of the two fixed positions: lanes l < 62 - q own (i, j) = (l, l + 1 + q) (distance
//              q + 1), lanes 62 - q .. 62 own (l - 62 + q, l) (distance 62 - q): 62 - q and q + 1 items, 63 together;
//              lane 63 owns none.  In both cases one fixed position is the LANE NUMBER and the other is the lane number
//              plus q + 1 resp. q + 2 modulo 64: the lane's own weight plus a copy of the weights that rotates through the
//              wavefront by one lane per row (DPP wave_rol) -- the fixed sums cost no memory access and no uniform operand.
//              (Rounds 1-3 dealt the rows by i: (q, q + 1 + l) and (61 - q, l): two LDS reads per item for the same sum.)
//   q = 31     lanes 0..62: the pairs of i = l;  lane 63: the singles
// and keeps per item a cursor (members [cursor, 64) are visited; 0 = no member left).  The chunk (lo, T] is produced by a
// WALK: every item's next member (fixed sum + the weight under the cursor) is compared with T, the items that have one are listed,
// those lanes emit it (key = sum bits << 32 | positions; slot = running count + mbcnt of the ballot) and step their
// cursor.  An item costs one compare when it has nothing to give, the members cost one trip each; no binary searches,
// no count-then-write double pass, no block scan.  T can be ANY value -- exactness does not depend on it -- so it is
// sized to the work: first guess from pb_bound_guess / the growth exponent of the last two bounds, a short chunk is
// extended in place (the walk resumes), an overflowing walk is abandoned and retried with a smaller T.
// ---------------------------------------------------------------------------------------
// uniform search state of a frame (wave-uniform values)
struct PbwState {
    float best;
    int j, nlive, cmp, suc1, suc2, bestidx;
    u64 bestD, bestE;
};
// arguments and results of the sorted path (pbw_sorted_chunk: a real function call, made with nothing live across it)
struct PbSortArgs {
    PbwState S;
    PbFrame fr;
    u64 d0;
    float mn, mx, c4;
    int n, order, state, stop, ntep;
};
// ... and of pbw_redo_range: the sums (lo, T] -- n TEPs, `done` visited before them -- once more, in chunks of 8-byte keys
struct PbRedoArgs {
    PbwState S;
    PbFrame fr;
    u64 d0;
    float lo, T, smax, c4;
    int done, n, order, target, state, stop, ntep;
};

constexpr int kPbMaxTie = 16;
constexpr int kPbWaveCap = 384;   // chunk capacity of the chunk kernel (10 KiB of LDS per frame: four wavefronts per SIMD; 512 = 12 KiB = three)

template <int CAP>
struct __attribute__((aligned(16))) PbWaveLds {
    float pre[4];             // pre[3] = NaN: the "weight" under an exhausted cursor (0) -- its sum compares false with any bound
    float w[128];             // |y'|                                         } words [4, 364): the image of the record
    u64 P[64];                // rows of P'                                   } pb_singles_kernel wrote for the frame
    u64 Pzero;                // = 0: "row 64", what the unused positions of a pair's or a single's key read (no select)
    float cdfA[68];           // P[Bin(64, p1) <= b] ROUNDED TO float32 -- the rules only ever read the table through a
                              // (float) cast (pb_not_promising), so storing the rounded value is the same arithmetic
    unsigned char perm[128];  // original bit index of primed position p (for the codeword at the end)
    unsigned rpad[2];         // (the record's image ends here: 360 words)
    float tail[4][17];        // tail[g][c] <= the sum of the c lightest parity weights of quarter g (pbw_cost_floor)
    float qpar[64];           // q_p = sigmoid(c4 |y'_p|) of the parity positions (the success rule, pb_success_q)
    float cdfH[68];           // P[Bin(64, 1/2) <= b], float32 as cdfA
    // the chunk: as walked (slots 0..n-1), then (sorted path only) grouped by bucket and finally in visit order; 64 entries of
    // slack take the overshoot of the walk's last trip and the "never before me" pad of the rank count.  (Rounds 2-3 skewed
    // the array by one pad entry per eight against the two-bank pattern of lane-consecutive 64-bit accesses in the sorted
    // path, and let a trip overshoot by 128: 1.6 KiB that stood between the kernel and a fourth wavefront per SIMD.)
    u64 keys[CAP + 64];
    union {
        int hist[CAP];        // bucket counts, then cursors; the costs of a chunk
        unsigned list[CAP + 64];   // the walk's work list (one entry per member emitted)
    };
    unsigned cur[8][64];      // tentative cursors of the walk: byte q & 3 of cur[q / 4][lane] = item q of that lane
    union {
        struct {
            u64 ck[16], rk[16];   // sort-free chunk pass: improvement candidates / the records among them (key, cost)
            float cc[16], rc[16];
        };
        PbSortArgs sa;        // (the pass has given up on the chunk when the sorted path is called: its words are free)
        PbRedoArgs ra;        // (read into registers on entry, written on exit: the calls in between use the words)
    };
    u64 cw[2];
};
static_assert(sizeof(PbSortArgs) <= 384 && sizeof(PbRedoArgs) <= 384, "the rare paths' arguments borrow the candidate words");

__device__ __forceinline__ int pbw_phys(int i) { return i; }

// The weighted distance of a candidate, two ways.  The chunk kernel keeps no byte LUT (8 KiB of LDS per frame: with it two
// wavefronts fit a SIMD, without it three to four, and the kernel spends half its time waiting):
//   pbw_cost_floor  a LOWER bound from the NUMBER of parity discrepancies in each quarter of the parity part: the candidate
//                   differs from the hard decisions in popcount(D_g) positions of quarter g, which weigh at least as much as
//                   that quarter's popcount(D_g) lightest positions.  The rules need a cost only to know whether it beats
//                   the best so far; past the first chunk the bound settles that for all but ~1 key in 10^3..10^4 (measured
//                   on NMS failures: 0.01 % at 1.0 dB, 0.06 % at 2.5 dB; one popcount over all 64 positions lets 13-17 %
//                   through: a random D has ~32 ones, and the 32 lightest weights are light).  Rounded down twice (table
//                   entries, then the sum) so that it stays below the float32 value of the canonical summation, whose
//                   rounding errors are < 1e-6 relative.
//   pbw_cost_exact  the canonical order of the byte LUT (each byte ascending from 0, bytes added in order; tep_cost),
//                   64 conditional adds: bit-identical to the LUT form.  For the survivors of the bound.
template <int CAP>
__device__ __forceinline__ float pbw_cost_floor(const PbWaveLds<CAP> &L, float mrb, u64 D)
{
    const unsigned lo = (unsigned)D, hi = (unsigned)(D >> 32);
    const float t0 = L.tail[0][__popc(lo & 0xFFFFu)], t1 = L.tail[1][__popc(lo >> 16)];
    const float t2 = L.tail[2][__popc(hi & 0xFFFFu)], t3 = L.tail[3][__popc(hi >> 16)];
    return (((mrb + t0) + t1) + (t2 + t3)) * 0.99999f;
}
template <int CAP>
__device__ __forceinline__ float pbw_cost_exact(const PbWaveLds<CAP> &L, float mrb, u64 D)
{
    float acc = mrb;
#pragma unroll 1
    for (int b = 0; b < 8; ++b) {
        const unsigned v = (unsigned)(D >> (8 * b)) & 255u;
        float bs = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; ++t) bs = ((v >> t) & 1u) ? bs + L.w[64 + 8 * b + t] : bs;
        acc = acc + bs;
    }
    return acc;
}
// cost if it can be below `bound`, +inf otherwise (exact for every use: the rules only compare costs with bests <= bound)
template <int CAP>
__device__ __forceinline__ float pbw_cost(const PbWaveLds<CAP> &L, float mrb, u64 D, float bound)
{
    float c = __builtin_inff();
    if (pbw_cost_floor<CAP>(L, mrb, D) < bound) c = pbw_cost_exact<CAP>(L, mrb, D);
    return c;
}

// positions of a key's low word: p0 | p1 << 8 | p2 << 16 | weight << 24 (ascending positions; an unused position is 64, the
// zero row behind P').  Keys made by the chunk kernel's walk carry, in bits 26-27, the frontier growth of the TEP's pop plus
// one (pb_delta + 1 = 0, 1, 2: the walk knows it from the item's geometry; unpacked from the positions it is ~25 instructions)
__device__ __forceinline__ PbTep pbw_tep(unsigned code)
{
    return PbTep{(int)(code & 255u), (int)((code >> 8) & 255u), (int)((code >> 16) & 255u), (int)((code >> 24) & 3u)};
}
constexpr unsigned kPbUnused1 = 64u << 8, kPbUnused2 = 64u << 16;

// the items of a lane (see above).  base: the members are m in (base, 63]; code / sh: a member's key is code | m << sh
struct PbwItem {
    int i, j, base, sh;
    unsigned code;
};
__device__ __forceinline__ PbwItem pbw_item_rt(int q, int l)
{
    PbwItem it;
    const bool tri = q < 31, first = l < 62 - q;
    it.i = tri ? (first ? l : l - 62 + q) : l;
    it.j = tri ? (first ? l + 1 + q : l) : l;
    it.base = tri ? (l <= 62 ? it.j : 63) : (l <= 62 ? l : -1);
    it.code = tri ? ((3u << 24) | ((unsigned)it.j << 8) | (unsigned)it.i) : (l <= 62 ? ((2u << 24) | kPbUnused2 | (unsigned)l) : ((1u << 24) | kPbUnused2 | kPbUnused1));
    it.sh = tri ? 16 : (l <= 62 ? 8 : 0);
    return it;
}

// In-kernel stamps of the diagnostic instantiation (PROF = true, launched only when LDPC_PB_PROFILE is set): the shader
// clock since the previous stamp is added to slot k.  The product instantiation contains none of this.
enum { kPwSetup = 0, kPwWalk, kPwSort, kPwTie, kPwEval, kPwRules, kPwCombine, kPwFinish, kPwFrames, kPwChunks, kPwWalks, kPwKeys,
       kPwSweepA, kPwSweepB, kPwDense, kPwRounds, kPwTrips, kPwScan, kPwSorted, kPwLoad1, kPwLoad2, kPwStore, kPwSlots };
#define PBW_STAMP(k) do { if constexpr (PROF) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); pt[k] += now__ - plast; plast = now__; } } while (0)

// Walk state.  Registers: the COMMITTED cursors only (one byte per item, four items per register; members [cursor, 64) are
// visited).  LDS: the TENTATIVE cursors L.cur[q / 4][lane] of the chunk being sized -- in the dense phase below a lane works on
// whatever item the list hands it.  The sum of an item's NEXT member is NOT kept (rounds 1-3 held the 32 of them in registers:
// with the chunk's keys that was 225 live VGPRs against the 168 of three wavefronts per SIMD, i.e. 57 registers in scratch,
// re-read and re-written once per walk -- 0.87 GB of HBM writes per launch at 1.0 dB): it is the item's fixed sum plus the weight
// under its cursor, two LDS reads and an add when the walk asks for it.
struct PbWalk {
    unsigned ecur[8];
};

__device__ __forceinline__ float pbw_nan() { return __int_as_float(0x7FC00000); }

template <int CAP>
__device__ __forceinline__ void pbw_cursors_store(PbWaveLds<CAP> &L, const unsigned (&cur)[8], int lane)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) L.cur[k][lane] = cur[k];
}
template <int CAP>
__device__ __forceinline__ void pbw_cursors_load(const PbWaveLds<CAP> &L, unsigned (&cur)[8], int lane)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) cur[k] = L.cur[k][lane];
}

template <int CAP>
__device__ __forceinline__ void pbw_walk_init(PbWaveLds<CAP> &L, PbWalk &W, int order, int lane)
{
    // every cursor at 64 (an item's next member is its last position 63); 0 for the items that do not exist: lane 63's
    // triples, the pairs when the order is 1
#pragma unroll
    for (int k = 0; k < 8; ++k) W.ecur[k] = lane <= 62 ? 0x40404040u : 0u;
    if (lane == 63 || order < 2) W.ecur[7] = lane <= 62 ? 0x00404040u : 0x40000000u;
    pbw_cursors_store<CAP>(L, W.ecur, lane);
}

// w[(lane + 1) & 63] of a register that holds w[lane] in every lane, three ways (ROT: what the context's probe of the
// wave_rol:1 DPP control found: -1 = a lane receives its upper neighbour's value, +1 = its lower neighbour's, 0 = unusable)
template <int ROT>
__device__ __forceinline__ float pbw_rot1(float x)
{
    static_assert(ROT != 0, "no rotation: the caller reads LDS");
    if constexpr (ROT < 0) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x134, 0xF, 0xF, true));   // wave_rol:1
    else return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x13C, 0xF, 0xF, true));                     // wave_ror:1
}

// Emit every member with sum <= T that lies beyond the tentative cursors (L.cur); returns the new running count (> CAP: the
// chunk overflowed, the walk stopped early and the caller puts the committed cursors back).  Two phases:
//   list    one pass over the lane's 32 items.  An item's next member has the sum (w[i] + w[j]) + w[cursor - 1]; one of i, j
//           is the lane number and the other sits q + 1 or q + 2 lanes further (mod 64), so the fixed part is the lane's own
//           weight plus a copy of the weights that moves one lane per row (pbw_rot1); the weight under the cursor is one LDS
//           read (a cursor of 0 reads the NaN in front of the weights: no member left, no separate test); one compare.  The
//           items whose next member is <= T are appended to a work list (ballot + mbcnt, no loop); 86 % have nothing to give;
//   dense   the list, 64 entries per trip, every lane emits one or two members of its entry's item (key = sum bits << 32 |
//           positions; slot = running count + mbcnt), steps that item's cursor in LDS and, if the item's next member is
//           <= T too, appends the entry to the list's tail again -- so a trip runs at full lanes whatever the items' lengths.
// A lane works on whatever item the list hands it, hence the cursors in LDS and the item geometry from run-time (q, lane).
// (Round 3 kept the next-member sums in registers -- 32 VGPRs -- and took them back from the dense phase in a third sweep,
//  "collect", ~9 instructions per item: 57 spilled registers re-read and re-written once per walk.)
// K4 = true: the keys are written as their low words only -- 4 bytes, the positions -- and the pass recomputes a key's sum from
// them (pbw_scan4): the same key memory then holds 2 CAP + 64 keys, and every per-chunk cost (the list pass, the probes, the
// bound arithmetic, the reductions) is paid once per ~680 keys instead of once per ~310.  KCAP: the capacity in keys.
// The work list is a RING of CAP + 64 entries in both forms (an entry is free once its trip has read it).
template <int CAP, bool K4>
struct PbwCaps {
    static constexpr int KCAP = K4 ? 2 * CAP + 64 : CAP;      // keys of a chunk (64 more fit behind them)
    static constexpr int RING = CAP + 64;                      // work-list entries
};
template <int CAP, bool PROF, int ROT, bool K4>
__device__ __forceinline__ int pbw_walk(PbWaveLds<CAP> &L, float T, int cnt, int order, int lane,
                                        unsigned long long (&pt)[kPwSlots])
{
    constexpr int KCAP = PbwCaps<CAP, K4>::KCAP, RING = PbwCaps<CAP, K4>::RING;
    unsigned long long plast = 0;
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();
    static_assert(sizeof(L.keys) / (K4 ? 4 : 8) >= KCAP + 64 && CAP >= 128, "a dense trip may write 63 keys past the capacity");
    static_assert(sizeof(L.list) / 4 >= RING, "work-list ring");
    static_assert(offsetof(PbWaveLds<CAP>, w) >= 4 && offsetof(PbWaveLds<CAP>, w) == offsetof(PbWaveLds<CAP>, pre) + 16, "the NaN sits right in front of the weights");
    unsigned *const list = L.list;      // entry: q | owner lane << 5
    const float *const w = L.w;
    int tail = 0;
    {
        // (an opaque copy of the lane number per walk: otherwise lane-dependent addresses are hoisted out of every loop
        //  around the walk, kept for the whole kernel and spilled)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        unsigned cur[8];
        pbw_cursors_load<CAP>(L, cur, ln);
        const float wl = w[ln];
        const char *const wbytes = reinterpret_cast<const char *>(w) - 4;       // + 4 * cursor = the weight under the cursor
        // eight items at a time: the eight sums first (their LDS reads in flight together -- a branch behind every item made
        // the wavefront wait for each read by itself: ~32 exposed LDS latencies per walk), then the eight ballots and appends
        const auto next_sum = [&](int q, float sb) {
            const unsigned a4 = ((cur[q >> 2] >> (8 * (q & 3))) & 255u) << 2;
            return sb + *reinterpret_cast<const float *>(wbytes + a4);
        };
        const auto append = [&](int q, float sv) {
            const bool pend = sv <= T;
            const u64 act = tail <= RING - 64 ? __ballot(pend) : 0ull;      // (more pending items than the ring takes: an overflow already)
            if (act) {
                const int p = tail + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
                if (pend) list[p] = (unsigned)q | ((unsigned)ln << 5);
                tail += __popcll(act);
            }
        };
        if (order > 2) {
            // Rows that can have a member <= T at all.  The smallest sum of row q is (w[61 - q] + w[62]) + w[63] (every other
            // member of the row has positions at least as reliable, and float addition is monotone); it grows with q, so the
            // rows with something to give are a PREFIX 0 .. qmax - 1: one compare per row in lane q, one ballot.  Deep rows
            // stay empty for most of a search (a search of 3500 TEPs visits 8 % of the table), at 2.5 dB nearly all of them.
            int qmax;
            {
                const float rmin = (w[(61 - ln) & 63] + w[62]) + w[63];
                qmax = __popcll(__ballot(ln < 31 && rmin <= T));
            }
            float r1;                    // w[(lane + q + 1) & 63]
            if constexpr (ROT != 0) r1 = pbw_rot1<ROT>(wl); else r1 = w[(ln + 1) & 63];
            const float s31 = next_sum(31, ln <= 62 ? wl : 0.0f);      // (the pairs / singles row: always looked at, with group 0)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (8 * g >= qmax) break;
                float sv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int q = 8 * g + u;
                    if (q < 31) {
                        float r2;
                        if constexpr (ROT != 0) r2 = pbw_rot1<ROT>(r1); else r2 = w[(ln + q + 2) & 63];
                        sv[u] = next_sum(q, wl + (ln < 62 - q ? r1 : r2));
                        r1 = r2;
                    } else {
                        sv[u] = pbw_nan();
                    }
                }
                asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7]));
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (8 * g + u < 31) append(8 * g + u, sv[u]);
            }
            append(31, s31);
        } else {
            append(31, next_sum(31, ln <= 62 ? wl : 0.0f));
        }
    }
    PBW_STAMP(kPwSweepA);
    if (tail == 0) return cnt;
    if (tail > RING - 64) return KCAP + 1;
    wave_fence();
    const auto ring = [](int p) { p = p >= RING ? p - RING : p; return p >= RING ? p - RING : p; };      // (p < 3 RING: an entry is a key at least)
    static_assert(3 * RING > KCAP + 128, "ring index");
    int head = 0;
    while (head < tail && cnt <= KCAP) {
        if constexpr (PROF) pt[kPwTrips] += 1;
        // A trip serves up to 64 entries, two members each.  When fewer than 33 entries wait -- the tail of a walk: a few long
        // items -- every entry gets 2, 4 or 8 lanes, lane u of its group taking the members 2u and 2u + 1 below the cursor:
        // inside an item the sums rise as the position falls, so the members <= T are a PREFIX and every lane judges its two
        // by itself; the group's leader steps the cursor by what the group emitted.  (Two members per entry and trip whatever
        // the list's length: 9.3 trips per chunk at 1.0 dB, half of them at a tenth of the lanes.)
        const int nent = tail - head;
        const int gs = (nent > 32 || cnt > KCAP - 128) ? 0 : (nent > 16 ? 1 : (nent > 8 ? 2 : 3));      // log2 of the lanes per entry
        const int e = head + (lane >> gs), u = lane & ((1 << gs) - 1);
        const bool has = e < tail;
        const unsigned ent = has ? list[ring(e)] : 0u;
        const int q = (int)(ent & 31u), l = (int)((ent >> 5) & 63u);
        int i, j, base, sh;
        unsigned code;
        if (q < 31) {
            const bool first = l < 62 - q;
            i = first ? l : l - 62 + q; j = first ? l + 1 + q : l; base = j; sh = 16;
            code = (3u << 24) | ((unsigned)j << 8) | (unsigned)i;
        } else {
            i = l; j = l; base = l <= 62 ? l : -1; sh = l <= 62 ? 8 : 0;
            code = l <= 62 ? ((2u << 24) | kPbUnused2 | (unsigned)l) : ((1u << 24) | kPbUnused2 | kPbUnused1);
        }
        const int wt = q < 31 ? 3 : (l <= 62 ? 2 : 1);
        unsigned char *const cb = reinterpret_cast<unsigned char *>(&L.cur[q >> 2][l]) + (q & 3);
        const int a = has ? (int)*cb : 1;
        const float sbv = q < 31 ? w[i] + w[j] : (l <= 62 ? w[l] : 0.0f);
        // this lane's two members (the order of the keys inside a chunk is irrelevant: second members follow the first ones)
        const int m = a - 1 - 2 * u;
        const float s = sbv + w[m & 63], s2 = sbv + w[(m - 1) & 63];
        const bool one = has && m > base && (u == 0 || s <= T);                   // (the group's first member is <= T: that is why the item is listed)
        const bool two = one && m - 1 > base && s2 <= T && cnt <= KCAP - 64;       // (no second members in a trip that may end beyond KCAP + 63)
        const u64 act = __ballot(one), act2 = __ballot(two);
        // what the entry's group emitted (a prefix of the item's members below the cursor), its new cursor, and whether the
        // member behind it is <= T too (then the entry is listed again)
        const int gsh = (lane >> gs) << gs;
        const u64 gm = gs == 0 ? 1ull : ((1ull << (1 << gs)) - 1ull);
        const int k = __popcll((act >> gsh) & gm) + __popcll((act2 >> gsh) & gm);
        const int mlast = a - k;                                                  // the lowest position emitted
        const bool left = mlast > base + 1;                                       // the item has members beyond this trip's
        const bool lead = has && u == 0;
        const bool again = lead && left && sbv + w[(mlast - 1) & 63] <= T;
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, (unsigned)cnt));
        const int pos2 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(act2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act2, (unsigned)(cnt + __popcll(act))));
        const u64 more = __ballot(again);
        const int nt = tail + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(more >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)more, 0u));
        wave_fence();                    // (every lane has read its entry: the slots may be written now)
        if (one) {
            const unsigned tmpl = code;       // (the field of the LAST position is zero in it; unused fields hold 64)
            // frontier growth of the pop (pb_delta): the extended child exists below position 63 and below the order, the
            // adjacent child if the last position can move down by one; + 1, in bits 26-27
            const unsigned g1 = (unsigned)((m < 63 && wt < order) + (m > base + 1)) << 26;
            const unsigned g2 = (unsigned)((wt < order) + (m - 1 > base + 1)) << 26;
            if constexpr (K4) {
                unsigned *const codes = reinterpret_cast<unsigned *>(L.keys);
                codes[pos] = tmpl | g1 | ((unsigned)m << sh);
                if (two) codes[pos2] = tmpl | g2 | ((unsigned)(m - 1) << sh);
            } else {
                L.keys[pos] = ((u64)__float_as_uint(s) << 32) | (tmpl | g1 | ((unsigned)m << sh));
                if (two) L.keys[pos2] = ((u64)__float_as_uint(s2) << 32) | (tmpl | g2 | ((unsigned)(m - 1) << sh));
            }
        }
        if (lead) {
            *cb = (unsigned char)(left ? mlast : 0);        // (0: exhausted -- the list pass then reads the NaN)
            if (again) list[ring(nt)] = ent;
        }
        cnt += __popcll(act) + __popcll(act2);
        {
            const int served = 64 >> gs;
            head = head + served < tail ? head + served : tail;
        }
        tail += __popcll(more);
        wave_fence();
        // (the next trip appends at tail .. tail + 63 while the entries head + 64 .. tail - 1 are still unread)
        if (tail - head > RING - 64) { cnt = KCAP + 1; break; }
    }
    PBW_STAMP(kPwDense);
    return cnt;
}

// Typical bound of the N smallest sums in units of the smallest triple sum m3 = w61 + w62 + w63 (medians over decoding
// failures at 2.5 dB; the ratio is scale-free and tight: +-6 % between the 10th and 90th percentile, where the count
// changes like the ~6th power of the bound).  Only a first guess: pbw_next_chunk corrects it with exact counts.
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

// The next chunk: walks (lo, T] for a T aimed at `target` members, 0 < n <= CAP.  Returns n and T; the chunk's keys are
// L.keys[0..n) and the walk's cursors are committed.  -1: the range cannot be split (massively equal sums: the frame goes
// to the list replay); 0: nothing is left to visit (NaN sums).
// (K4: 4-byte keys, capacity 2 CAP + 64, see pbw_walk.  COMMIT = false: W.ecur keeps the cursors the chunk STARTED from -- the
//  caller commits, pbw_cursors_load, once the chunk is judged, or puts them back, pbw_cursors_store, and redoes the range)
template <int CAP, bool PROF, int ROT, bool K4 = false, bool COMMIT = true>
__device__ __forceinline__ int pbw_next_chunk(PbWaveLds<CAP> &L, PbWalk &W, int order, float lo, int done, int nall, int target, int lane,
                                              float &Tout, float &tprev, float &nprev, int &nwalks, unsigned long long (&pt)[kPwSlots],
                                              float Tcap = __builtin_inff())
{
    constexpr int KCAP = PbwCaps<CAP, K4>::KCAP;
    const float inf = __builtin_inff();
    const float *w = L.w;
    const float m3 = (w[61] + w[62]) + w[63];
    const float want = (float)(done + target);
    float Tl = lo, Th = inf;
    float T = nall - done <= KCAP ? inf : m3 * pb_bound_guess(want);
    if (tprev > 0.0f && nprev > 0.0f && lo > tprev && (float)done > nprev && T < inf) {   // growth exponent of the last two bounds
        const float pe = (__builtin_amdgcn_logf((float)done) - __builtin_amdgcn_logf(nprev)) / (__builtin_amdgcn_logf(lo) - __builtin_amdgcn_logf(tprev));
        if (pe > 1.5f && pe < 20.0f) T = lo * __builtin_amdgcn_exp2f((__builtin_amdgcn_logf(want) - __builtin_amdgcn_logf((float)done)) / pe);
    }
    if (!(T > lo)) T = lo > 0.0f ? lo * 1.05f : w[0];
    if (T > Tcap) T = Tcap;                  // (a caller that wants the chunks to end at a given bound)
    float tp = lo, np_ = (float)done;        // last point with a known count
    int cnt = 0, c_ok = 0;
    float T_ok = lo;
    unsigned a_ok[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < 48; ++it) {
        cnt = pbw_walk<CAP, PROF, ROT, K4>(L, T, cnt, order, lane, pt);
        ++nwalks;
        bool over = false;
        if (cnt > KCAP) {
            if (c_ok > 0) break;
            over = true;
            Th = T;
            pbw_cursors_store<CAP>(L, W.ecur, lane);      // back to the committed cursors
            cnt = 0;
        } else if (cnt > 0 && (!(T < inf) || 5 * cnt >= 2 * target || it >= 3)) {
            c_ok = cnt; T_ok = T;
            break;
        } else if (!(T < inf)) {
            return 0;                         // everything that can be visited has been
        } else {                              // too few so far: keep them and walk on from here
            if (cnt > 0) {
                c_ok = cnt; T_ok = T;
                pbw_cursors_load<CAP>(L, a_ok, lane);
            }
            Tl = T;
        }
        float Tn;
        if (over) {
            Tn = Tl > 0.0f ? Tl + (Th - Tl) * 0.5f : Th * 0.9f;
        } else {
            const float tot = (float)(done + cnt);
            float p = 6.0f;
            if (tp > 0.0f && np_ > 0.0f && tot != np_ && T != tp) {
                const float pe = (__builtin_amdgcn_logf(tot) - __builtin_amdgcn_logf(np_)) / (__builtin_amdgcn_logf(T) - __builtin_amdgcn_logf(tp));
                if (pe > 1.5f && pe < 20.0f) p = pe;
            }
            if (cnt > 0) { tp = T; np_ = tot; }
            Tn = cnt > 0 ? T * __builtin_amdgcn_exp2f((__builtin_amdgcn_logf(want) - __builtin_amdgcn_logf(tot)) / p) : T * 1.1f;
            if (it >= 6 || !(Tn > Tl) || !(Tn < Th)) Tn = Th < inf ? Tl + (Th - Tl) * 0.5f : T * 1.2f;
        }
        if (Tn > Tcap) Tn = Tcap;
        if (!(Tn > Tl) || !(Tn < Th)) break;
        T = Tn;
    }
    if (c_ok == 0) { pbw_cursors_store<CAP>(L, W.ecur, lane); return -1; }
    if (cnt != c_ok) pbw_cursors_store<CAP>(L, a_ok, lane);   // an overflow (or a dead end) after a usable shorter chunk: back to that one
    if constexpr (COMMIT) pbw_cursors_load<CAP>(L, W.ecur, lane);                     // commit
    tprev = lo; nprev = (float)done;
    Tout = T_ok;
    return c_ok;
}

// Sort-free pass over a chunk (the n keys as the walk left them, in no particular order).  The visit order matters to the
// rules only through (i) "best so far", which changes only at a key whose cost beats the best the chunk STARTED with -- a
// candidate; deep in a search a chunk holds none, or one or two -- and (ii) the frontier size, which matters only when it
// can be 1.  So: every key's cost, frontier growth and rule 1 against the chunk-start best, in parallel and in any order;
//   no candidate:  rule 1 depends on the sum alone, so the search stops at the SMALLEST firing sum, and the number of TEPs
//                  visited is the number of smaller sums: one count, no sort;
//   <= 16 candidates: they are put in visit order among themselves (a handful of comparisons), the records and their
//                  success rule follow sequentially, rule 1 is re-evaluated for the keys behind the first record with the
//                  best they see, and the stop / winner positions are counts again;
//   otherwise -1 and nothing changed: the caller sorts the chunk (pbw_process_chunk).  That is: many candidates (the first
//                  chunk or two), a frontier that may shrink to one entry (the first chunk, the tail of a complete scan),
//                  or a key whose sum EQUALS that of a key a position is counted against (list order would decide; the
//                  pass compares sums only, which keeps it small: it is compared against a handful of keys per chunk).
// Returns 0 = no rule fired (state advanced), 1 = stopped (stop / ntep set), -1 = not handled.
template <int CAP>
__device__ __forceinline__ int pbw_scan_chunk(PbWaveLds<CAP> &L, const PbParams &P, const PbFrame &Fr, u64 d0, int n, float mn, float mx, int lane,
                                              PbwState &S, int &stop, int &ntep)
{
    constexpr int PER = CAP / 64, STEP = PER % 4 == 0 ? 4 : 3;
    static_assert(CAP % 64 == 0 && PER % STEP == 0, "the pass reads its keys STEP slices at a time");
    if (S.nlive <= 1) return -1;      // (the first chunk: one entry in the frontier, its pops are counted one by one)
    const float best0 = S.best;
    // Rule 1 by probes.  With the best fixed, the rule's left-hand side bs = H[beta] + (A[beta] - H[beta]) w1 falls as the sum
    // rises (w1 = exp(c4 rs) spl falls, beta -- a floor of a float quotient, monotone as computed -- falls, A >= H); the
    // float32 evaluation follows that to a few units in the last place.  Lane l evaluates it at mn + (mx - mn)(l + 1) / 64:
    // below the last probe that still clears the threshold by 0.1 % no key of the chunk can fire, and none is evaluated --
    // every chunk of a search but its last.  Keys above it get the exact evaluation.
    float r_safe;
    {
        const float rp = lane == 63 ? mx : mn + (mx - mn) * ((float)(lane + 1) * (1.0f / 64.0f));
        float w1;
        const float bs = pb_promising_bs(rp, best0, Fr, P.c4, L.cdfA, L.cdfH, w1);
        const u64 unsafe = ~__ballot((double)bs > Fr.p_t_pro * 1.001);
        const int u = unsafe ? __builtin_ctzll(unsafe) : 64;
        r_safe = u == 0 ? -1.0f : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rp), u - 1));
    }
    const auto parity = [&](const PbTep &t) {
        u64 D = d0 ^ L.P[t.p0];
        if (t.wt > 1) D ^= L.P[t.p1];
        if (t.wt > 2) D ^= L.P[t.p2];
        return D;
    };
    u64 kq[PER];
    unsigned npneed = 0, npmask = 0, survmask = 0;
    int sumf = 0, neg = 0, nsurv = 0;
    unsigned short *const slist = reinterpret_cast<unsigned short *>(L.list);     // keys the cost bound could not rule out
    // Branch-free, STEP keys of the lane side by side: the keys, then their three P' rows each (an unused position reads the
    // zero row: no select), then the bound's four table entries each -- three LDS round trips per STEP keys.  An empty slot
    // holds a key whose sum is a NaN (every comparison false), whose positions read in-range garbage and whose growth field
    // says 0: no validity mask anywhere.  (Round 3's form compiled to a branch and a wait behind every single LDS read.)
    constexpr u64 kEmpty = 0xFFFFFFFFF7FFFFFFull;
    const char *const Pb = reinterpret_cast<const char *>(L.P);
    const char *const tb = reinterpret_cast<const char *>(L.tail);
    const unsigned d0l = (unsigned)d0, d0h = (unsigned)(d0 >> 32);
#pragma unroll
    for (int k0 = 0; k0 < PER; k0 += STEP) {
#pragma unroll
        for (int u = 0; u < STEP; ++u) { const int i = lane + 64 * (k0 + u); kq[k0 + u] = i < n ? L.keys[i] : kEmpty; }
        uint2 r0[STEP], r1[STEP], r2[STEP];
#pragma unroll
        for (int u = 0; u < STEP; ++u) {
            const unsigned code = (unsigned)kq[k0 + u];
            r0[u] = *reinterpret_cast<const uint2 *>(Pb + ((code & 255u) << 3));
            r1[u] = *reinterpret_cast<const uint2 *>(Pb + (((code >> 8) & 255u) << 3));
            r2[u] = *reinterpret_cast<const uint2 *>(Pb + (((code >> 16) & 255u) << 3));
        }
        float t0[STEP], t1[STEP], t2[STEP], t3[STEP];
#pragma unroll
        for (int u = 0; u < STEP; ++u) {
            const unsigned lo = __builtin_amdgcn_bitop3_b32(r0[u].x, r1[u].x, r2[u].x, 0x96) ^ d0l;
            const unsigned hi = __builtin_amdgcn_bitop3_b32(r0[u].y, r1[u].y, r2[u].y, 0x96) ^ d0h;
            t0[u] = *reinterpret_cast<const float *>(tb + (__popc(lo & 0xFFFFu) << 2));
            t1[u] = *reinterpret_cast<const float *>(tb + 68 + (__popc(lo >> 16) << 2));
            t2[u] = *reinterpret_cast<const float *>(tb + 136 + (__popc(hi & 0xFFFFu) << 2));
            t3[u] = *reinterpret_cast<const float *>(tb + 204 + (__popc(hi >> 16) << 2));
        }
#pragma unroll
        for (int u = 0; u < STEP; ++u) {
            const int k = k0 + u;
            const unsigned code = (unsigned)kq[k];
            const float rs = __uint_as_float((unsigned)(kq[k] >> 32));
            const bool surv = (((rs + t0[u]) + t1[u]) + (t2[u] + t3[u])) * 0.99999f < best0;      // (pbw_cost_floor)
            survmask |= surv ? 1u << k : 0u;
            npneed |= rs > r_safe ? 1u << k : 0u;
            const int fld = (int)((code >> 26) & 3u);     // growth + 1
            sumf += fld; neg += fld == 0;
        }
        asm volatile("" : "+v"(survmask), "+v"(npneed), "+v"(sumf), "+v"(neg) : : "memory");
    }
    const int sumdel = sumf - PER;       // (every slot, empty or not, carried a + 1)
    if (__ballot(survmask != 0)) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const bool surv = (survmask >> k) & 1u;
            const u64 sm = __ballot(surv);
            if (sm) {
                if (surv) slist[nsurv + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u))] = (unsigned short)(lane + 64 * k);
                nsurv += __popcll(sm);
            }
        }
    }
    const int negtot = wave_add_i32(neg), deltot = wave_add_i32(sumdel);
    if (S.nlive - negtot <= 1) return -1;
    // the survivors' costs, 64 at a time (the canonical summation, no LUT: ~200 instructions, but per BATCH); those that beat the
    // chunk-start best are the candidates
    int ncand = 0;
    if (nsurv) {
        wave_fence();
        for (int b0 = 0; b0 < nsurv && ncand <= 16; b0 += 64) {
            const bool has = b0 + lane < nsurv;
            const u64 key = has ? L.keys[slist[b0 + lane]] : 0ull;
            const float c = has ? pbw_cost_exact<CAP>(L, __uint_as_float((unsigned)(key >> 32)), parity(pbw_tep((unsigned)key))) : __builtin_inff();
            const bool cand = c < best0;
            const u64 cm = __ballot(cand);
            if (cm) {
                const int idx = ncand + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
                if (cand && idx < 16) { L.ck[idx] = key; L.cc[idx] = c; }
                ncand += __popcll(cm);
            }
        }
        if (ncand > 16) return -1;
    }
    // rule 1 for the keys above the last safe probe (the last chunk of a search; nothing elsewhere)
    if (__ballot(npneed != 0)) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const bool need = (npneed >> k) & 1u;
            if (__ballot(need)) {
                float w1;
                if (need && pb_not_promising(__uint_as_float((unsigned)(kq[k] >> 32)), best0, Fr, P.c4, L.cdfA, L.cdfH, w1)) npmask |= 1u << k;
            }
        }
    }
    // (sums are >= +0: their bit patterns order like the floats; an invalid slot holds all ones)
    const auto sumbits = [](u64 key) { return (unsigned)(key >> 32); };
    bool tie = false;
    // number of my keys with a smaller sum than `ref`; a different key with the same sum is a tie
    const auto count_before = [&](u64 ref) {
        int c = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            c += sumbits(kq[k]) < sumbits(ref);
            tie |= sumbits(kq[k]) == sumbits(ref) && kq[k] != ref;
        }
        return wave_add_i32(c);
    };
    int nrec = 0, stop2 = 0;     // records among the candidates; stop2: the success rule fired on the last of them
    if (ncand > 0) {
        // ---- the candidates, in visit order
        wave_fence();
        {
            const u64 my = L.ck[lane & 15];
            const float myc = L.cc[lane & 15];
            int r = 0;
            for (int d = 0; d < ncand; ++d) { const u64 o = L.ck[d]; r += sumbits(o) < sumbits(my); tie |= lane < ncand && sumbits(o) == sumbits(my) && o != my; }
            wave_fence();
            if (lane < ncand) { L.ck[r] = my; L.cc[r] = myc; }
            wave_fence();
        }
        if (__ballot(tie)) return -1;
        // ---- records and the success rule, sequentially (every lane runs the same arithmetic on the same values)
        float before = best0;
        for (int t = 0; t < ncand && !stop2; ++t) {
            const u64 key = L.ck[t];
            const float c = L.cc[t];
            if (c < before) {
                if (lane == 0) { L.rk[nrec] = key; L.rc[nrec] = c; }
                ++nrec;
                const float w1 = det_expf(P.c4 * __uint_as_float((unsigned)(key >> 32))) * Fr.spl;
                if (pb_success_q(parity(pbw_tep((unsigned)key)), w1, L.qpar, Fr)) stop2 = 1;
                before = c;
            }
        }
        wave_fence();
        // ---- rule 1 again for the keys behind the first record, with the best they see
        if (nrec > 0) {
            const unsigned s0 = sumbits(L.rk[0]);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                if (lane + 64 * k < n && sumbits(kq[k]) >= s0 && kq[k] != L.rk[0]) {
                    int t = 0;
                    for (int u = 0; u < nrec; ++u) { const u64 r = L.rk[u]; t += sumbits(r) < sumbits(kq[k]); tie |= sumbits(r) == sumbits(kq[k]) && r != kq[k]; }
                    if (t > 0) {
                        float w1;
                        const bool np = pb_not_promising(__uint_as_float(sumbits(kq[k])), L.rc[t - 1], Fr, P.c4, L.cdfA, L.cdfH, w1);
                        npmask = (npmask & ~(1u << k)) | (np ? 1u << k : 0u);
                    }
                }
            }
        }
    }
    // ---- the smallest sum on which rule 1 fires: every key of that sum sees the same best, so the first of them stops
    unsigned fs = 0x7FFFFFFFu;
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (((npmask >> k) & 1u) && sumbits(kq[k]) < fs) fs = sumbits(kq[k]);
    const unsigned sF = (unsigned)wave_min_i32((int)fs);
    // ---- the stop: the earlier of rule 1's first key and the record on which rule 2 fired
    int reason = 0;
    unsigned sstop = 0;
    if (sF != 0x7FFFFFFFu) { reason = 1; sstop = sF; }
    if (stop2) {
        const unsigned sR = sumbits(L.rk[nrec - 1]);
        if (reason == 1 && sR == sF) tie = true;
        if (reason == 0 || sR < sF) { reason = 2; sstop = sR; }
    }
    int nbefore = nrec;           // records that really happened: those before the stop (and the stop itself for rule 2)
    if (reason) {
        nbefore = 0;
        for (int u = 0; u < nrec; ++u) nbefore += sumbits(L.rk[u]) < sstop;
        nbefore += reason == 2;
    }
    int rank_best = 0, rank_stop = 0;
    if (nbefore > 0) rank_best = count_before(L.rk[nbefore - 1]);
    if (reason == 2) rank_stop = rank_best;
    if (reason == 1) {   // (the keys of the stopping sum all fire: the first of them in list order is at this position, whichever it is)
        int c = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) c += sumbits(kq[k]) < sstop;
        rank_stop = wave_add_i32(c);
    }
    if (__ballot(tie)) return -1;
    // ---- commit
    if (nbefore > 0) {
        const u64 bk = L.rk[nbefore - 1];
        const PbTep t = pbw_tep((unsigned)bk);
        u64 E = 1ull << t.p0;
        if (t.wt > 1) E |= 1ull << t.p1;
        if (t.wt > 2) E |= 1ull << t.p2;
        S.best = L.rc[nbefore - 1]; S.bestD = parity(t); S.bestE = E;
        S.bestidx = S.j + rank_best + 1;
    }
    S.suc2 += nbefore;
    if (reason) {
        S.cmp += 2 * (rank_stop + 1);             // (the frontier never holds a single entry here: no one-comparison pops)
        S.suc1 += reason == 1 ? rank_stop : rank_stop + 1;
        stop = reason; ntep = S.j + rank_stop + 1;
        return 1;
    }
    S.cmp += 2 * n; S.suc1 += n;
    S.j += n; S.nlive += deltot;
    return 0;
}

// pbw_scan_chunk for a chunk of 4-BYTE keys (pbw_walk<K4>: up to 2 CAP + 64 of them): the same sort-free pass, with a key's
// sum recomputed from its positions wherever it is needed -- (w[p0] + w[p1]) + w[p2], the order the walk formed it in, three LDS
// reads and two adds -- and NO key kept in registers: the ordinary chunk reads each key once; the rare paths (a rule fires,
// improvement candidates) read them again.  Same results as pbw_scan_chunk on the same keys; -1 leaves the state untouched and
// the caller redoes the chunk's sum range with 8-byte keys (pbw_redo_range).
template <int CAP>
__device__ __forceinline__ int pbw_scan4(PbWaveLds<CAP> &L, const PbParams &P, const PbFrame &Fr, u64 d0, int n, float mn, float mx, int lane,
                                         PbwState &S, int &stop, int &ntep)
{
    constexpr int STEP = 2;
    if (S.nlive <= 1) return -1;      // (the first chunk of a frame without its head: one entry in the frontier)
    const float best0 = S.best;
    float r_safe;                     // rule 1 by probes (pbw_scan_chunk)
    {
        const float rp = lane == 63 ? mx : mn + (mx - mn) * ((float)(lane + 1) * (1.0f / 64.0f));
        float w1;
        const float bs = pb_promising_bs(rp, best0, Fr, P.c4, L.cdfA, L.cdfH, w1);
        const u64 unsafe = ~__ballot((double)bs > Fr.p_t_pro * 1.001);
        const int u = unsafe ? __builtin_ctzll(unsafe) : 64;
        r_safe = u == 0 ? -1.0f : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rp), u - 1));
    }
    const unsigned *const codes = reinterpret_cast<const unsigned *>(L.keys);
    const char *const Pb = reinterpret_cast<const char *>(L.P);
    const char *const wb = reinterpret_cast<const char *>(L.w);
    const char *const tb = reinterpret_cast<const char *>(L.tail);
    const unsigned d0l = (unsigned)d0, d0h = (unsigned)(d0 >> 32);
    constexpr unsigned kEmpty = 0xF7FFFFFFu;       // an empty slot: growth field 0 + 1, positions that read in range; its sum is forced to NaN
    // sum bits of a key (all ones for an empty slot: a NaN, larger than every sum as an integer)
    const auto sum_of = [&](unsigned code, bool valid) {
        const float w0 = *reinterpret_cast<const float *>(wb + ((code & 255u) << 2));
        const float w1 = *reinterpret_cast<const float *>(wb + (((code >> 8) & 255u) << 2));
        const float w2 = *reinterpret_cast<const float *>(wb + (((code >> 16) & 255u) << 2));
        const unsigned wt = (code >> 24) & 3u;
        float rs = wt > 1u ? w0 + w1 : w0;
        rs = wt > 2u ? rs + w2 : rs;
        return valid ? __float_as_uint(rs) : 0xFFFFFFFFu;
    };
    const auto key_at = [&](int k, unsigned &code, unsigned &sb) {
        const int i = k * 64 + lane;
        code = i < n ? codes[i] : kEmpty;
        sb = sum_of(code, i < n);
    };
    const auto parity = [&](const PbTep &t) {
        u64 D = d0 ^ L.P[t.p0];
        if (t.wt > 1) D ^= L.P[t.p1];
        if (t.wt > 2) D ^= L.P[t.p2];
        return D;
    };
    int sumf = 0, neg = 0, nsurv = 0;
    unsigned fs = 0x7FFFFFFFu;     // the smallest sum on which rule 1 fires (against the chunk-start best)
    unsigned short *const slist = reinterpret_cast<unsigned short *>(L.list);     // keys the cost bound could not rule out
    const int nsl = (n + 63) >> 6;
#pragma unroll 1
    for (int k0 = 0; k0 < nsl; k0 += STEP) {       // (a rolled loop: unrolled over the 14 slices it is 12 KiB of code and the kernel spills)
        unsigned code[STEP], sb[STEP];
        uint2 r0[STEP], r1[STEP], r2[STEP];
#pragma unroll
        for (int u = 0; u < STEP; ++u) {
            key_at(k0 + u, code[u], sb[u]);
            r0[u] = *reinterpret_cast<const uint2 *>(Pb + ((code[u] & 255u) << 3));
            r1[u] = *reinterpret_cast<const uint2 *>(Pb + (((code[u] >> 8) & 255u) << 3));
            r2[u] = *reinterpret_cast<const uint2 *>(Pb + (((code[u] >> 16) & 255u) << 3));
        }
        float t0[STEP], t1[STEP], t2[STEP], t3[STEP];
#pragma unroll
        for (int u = 0; u < STEP; ++u) {
            const unsigned lo = __builtin_amdgcn_bitop3_b32(r0[u].x, r1[u].x, r2[u].x, 0x96) ^ d0l;
            const unsigned hi = __builtin_amdgcn_bitop3_b32(r0[u].y, r1[u].y, r2[u].y, 0x96) ^ d0h;
            t0[u] = *reinterpret_cast<const float *>(tb + (__popc(lo & 0xFFFFu) << 2));
            t1[u] = *reinterpret_cast<const float *>(tb + 68 + (__popc(lo >> 16) << 2));
            t2[u] = *reinterpret_cast<const float *>(tb + 136 + (__popc(hi & 0xFFFFu) << 2));
            t3[u] = *reinterpret_cast<const float *>(tb + 204 + (__popc(hi >> 16) << 2));
        }
#pragma unroll
        for (int u = 0; u < STEP; ++u) {
            const float rs = __uint_as_float(sb[u]);
            const bool surv = (((rs + t0[u]) + t1[u]) + (t2[u] + t3[u])) * 0.99999f < best0;      // (pbw_cost_floor)
            const u64 sm = __ballot(surv);
            if (sm) {
                if (surv) slist[nsurv + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u))] = (unsigned short)((k0 + u) * 64 + lane);
                nsurv += __popcll(sm);
            }
            const int fld = (int)((code[u] >> 26) & 3u);     // growth + 1
            sumf += fld; neg += fld == 0;
            const bool need = rs > r_safe;      // above the last safe probe (the last chunk of a search): the rule itself
            if (__ballot(need)) {
                float w1;
                if (need && pb_not_promising(rs, best0, Fr, P.c4, L.cdfA, L.cdfH, w1) && sb[u] < fs) fs = sb[u];
            }
        }
    }
    const int negtot = wave_add_i32(neg), deltot = wave_add_i32(sumf) - 64 * STEP * ((nsl + STEP - 1) / STEP);     // (every slot looked at carried a + 1)
    if (S.nlive - negtot <= 1) return -1;
    // the survivors' exact costs, 64 at a time; those that beat the chunk-start best are the candidates (as 8-byte keys)
    int ncand = 0;
    if (nsurv) {
        wave_fence();
        for (int b0 = 0; b0 < nsurv && ncand <= 16; b0 += 64) {
            const bool has = b0 + lane < nsurv;
            const unsigned code = has ? codes[slist[b0 + lane]] : kEmpty;
            const unsigned sbits = sum_of(code, has);
            const float c = has ? pbw_cost_exact<CAP>(L, __uint_as_float(sbits), parity(pbw_tep(code))) : __builtin_inff();
            const bool cand = c < best0;
            const u64 cm = __ballot(cand);
            if (cm) {
                const int idx = ncand + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
                if (cand && idx < 16) { L.ck[idx] = ((u64)sbits << 32) | code; L.cc[idx] = c; }
                ncand += __popcll(cm);
            }
        }
        if (ncand > 16) return -1;
    }
    const auto sumbits = [](u64 key) { return (unsigned)(key >> 32); };
    unsigned sF = (unsigned)wave_min_i32((int)fs);
    if (ncand == 0) {
        if (sF == 0x7FFFFFFFu) {     // no candidate, no key fires: the whole chunk is visited and nothing else happens
            S.cmp += 2 * n; S.suc1 += n;
            S.j += n; S.nlive += deltot;
            return 0;
        }
        // no candidate, rule 1 fires: the search stops at the first key of the smallest firing sum (every key of that sum fires)
        int cs = 0;
        for (int k = 0; k < nsl; ++k) { unsigned code, sb; key_at(k, code, sb); cs += sb < sF; }
        const int rank_stop = wave_add_i32(cs);
        S.cmp += 2 * (rank_stop + 1);
        S.suc1 += rank_stop;
        stop = 1; ntep = S.j + rank_stop + 1;
        return 1;
    }
    // ---- the candidates, in visit order; records and the success rule, sequentially (every lane the same arithmetic)
    bool tie = false;
    int nrec = 0, stop2 = 0;
    wave_fence();
    {
        const u64 my = L.ck[lane & 15];
        const float myc = L.cc[lane & 15];
        int r = 0;
        for (int d = 0; d < ncand; ++d) { const u64 o = L.ck[d]; r += sumbits(o) < sumbits(my); tie |= lane < ncand && sumbits(o) == sumbits(my) && o != my; }
        wave_fence();
        if (lane < ncand) { L.ck[r] = my; L.cc[r] = myc; }
        wave_fence();
    }
    if (__ballot(tie)) return -1;
    {
        float before = best0;
        for (int t = 0; t < ncand && !stop2; ++t) {
            const u64 key = L.ck[t];
            const float c = L.cc[t];
            if (c < before) {
                if (lane == 0) { L.rk[nrec] = key; L.rc[nrec] = c; }
                ++nrec;
                const float w1 = det_expf(P.c4 * __uint_as_float((unsigned)(key >> 32))) * Fr.spl;
                if (pb_success_q(parity(pbw_tep((unsigned)key)), w1, L.qpar, Fr)) stop2 = 1;
                before = c;
            }
        }
    }
    wave_fence();
    // Rule 1 again for the keys from the first record on, each with the best it really sees (the record before it).  A lower
    // best fires sooner, so a key that the LAST record's cost does not stop is stopped by none: probes with that cost leave
    // the keys beyond the last safe probe to evaluate -- usually none.  Keys before the first record keep what the
    // chunk-start best said (fs, if it lies before the first record).
    if (nrec > 0) {
        const unsigned s0 = sumbits(L.rk[0]);
        float r_safe2;
        {
            const float rp = lane == 63 ? mx : mn + (mx - mn) * ((float)(lane + 1) * (1.0f / 64.0f));
            float w1;
            const float bs = pb_promising_bs(rp, L.rc[nrec - 1], Fr, P.c4, L.cdfA, L.cdfH, w1);
            const u64 unsafe = ~__ballot((double)bs > Fr.p_t_pro * 1.001);
            const int u = unsafe ? __builtin_ctzll(unsafe) : 64;
            r_safe2 = u == 0 ? -1.0f : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rp), u - 1));
        }
        fs = fs < s0 ? fs : 0x7FFFFFFFu;
        for (int k = 0; k < nsl; ++k) {
            unsigned code, sb;
            key_at(k, code, sb);
            const u64 key = ((u64)sb << 32) | code;
            const bool behind = sb != 0xFFFFFFFFu && sb >= s0;
            int t = 0;
            if (behind)
                for (int u = 0; u < nrec; ++u) { const u64 r = L.rk[u]; t += sumbits(r) < sb; tie |= sumbits(r) == sb && r != key; }
            const bool need = behind && __uint_as_float(sb) > r_safe2;
            if (__ballot(need)) {
                float w1;      // (the first record itself is judged with the chunk-start best: t = 0)
                if (need && pb_not_promising(__uint_as_float(sb), t > 0 ? L.rc[t - 1] : best0, Fr, P.c4, L.cdfA, L.cdfH, w1) && sb < fs) fs = sb;
            }
        }
        sF = (unsigned)wave_min_i32((int)fs);
    }
    // ---- the stop: the earlier of rule 1's first key and the record on which rule 2 fired
    int reason = 0;
    unsigned sstop = 0;
    if (sF != 0x7FFFFFFFu) { reason = 1; sstop = sF; }
    if (stop2) {
        const unsigned sR = sumbits(L.rk[nrec - 1]);
        if (reason == 1 && sR == sF) tie = true;
        if (reason == 0 || sR < sF) { reason = 2; sstop = sR; }
    }
    int nbefore = nrec;           // records that really happened: those before the stop (and the stop itself for rule 2)
    if (reason) {
        nbefore = 0;
        for (int u = 0; u < nrec; ++u) nbefore += sumbits(L.rk[u]) < sstop;
        nbefore += reason == 2;
    }
    // positions: the keys below the last record that counts, the keys below the stopping sum; ties
    const u64 bk = nbefore > 0 ? L.rk[nbefore - 1] : 0ull;
    int cb = 0, cs = 0;
    for (int k = 0; k < nsl; ++k) {
        unsigned code, sb;
        key_at(k, code, sb);
        const u64 key = ((u64)sb << 32) | code;
        if (nbefore > 0) { cb += sb < sumbits(bk); tie |= sb == sumbits(bk) && key != bk; }
        if (reason == 1) cs += sb < sstop;
    }
    const int rank_best = wave_add_i32(cb), rank_stop = reason == 2 ? rank_best : wave_add_i32(cs);
    if (__ballot(tie)) return -1;
    // ---- commit
    if (nbefore > 0) {
        const PbTep t = pbw_tep((unsigned)bk);
        u64 E = 1ull << t.p0;
        if (t.wt > 1) E |= 1ull << t.p1;
        if (t.wt > 2) E |= 1ull << t.p2;
        S.best = L.rc[nbefore - 1]; S.bestD = parity(t); S.bestE = E;
        S.bestidx = S.j + rank_best + 1;
    }
    S.suc2 += nbefore;
    if (reason) {
        S.cmp += 2 * (rank_stop + 1);             // (the frontier never holds a single entry here: no one-comparison pops)
        S.suc1 += reason == 1 ? rank_stop : rank_stop + 1;
        stop = reason; ntep = S.j + rank_stop + 1;
        return 1;
    }
    S.cmp += 2 * n; S.suc1 += n;
    S.j += n; S.nlive += deltot;
    return 0;
}

// The n keys of one chunk (all TEPs of a sum range (mn, mx]): sort into visit order, evaluate in parallel, apply the
// sequential rules.  Returns 0 = no rule fired (state advanced), 1 = stopped (stop / ntep set), 2 = a run of more than
// kPbMaxTie equal sums (frame goes to the list replay).
template <int CAP, bool PROF>
__device__ __forceinline__ int pbw_process_chunk(PbWaveLds<CAP> &L, const PbParams &P, const PbFrame &Fr, u64 d0, int n, float mn, float mx,
                                                 int lane, PbwState &S, int &stop, int &ntep, unsigned long long (&pt)[kPwSlots], unsigned long long &plast)
{
    constexpr int PER = CAP / 64;
    // ---- bucket sort: CAP buckets over (mn, mx], counts -> offsets -> scatter (grouped by bucket) -> every key counts the
    // keys of its own bucket that sort before it.  Entries past a bucket's end belong to higher buckets (larger keys), past
    // the chunk's end to the all-ones pad: the count needs no mask and runs to the wave's fullest bucket.
    {
        static_assert(PER % 2 == 0, "a lane's bucket counters are read and written as pairs");
        int2 *h2 = reinterpret_cast<int2 *>(&L.hist[lane * PER]);
#pragma unroll
        for (int k = 0; k < PER / 2; ++k) h2[k] = make_int2(0, 0);
    }
    const float scale = mx > mn ? (float)CAP / (mx - mn) : 0.0f;
    const bool flat = !(scale < 3.0e38f);           // denormally close sums: one bucket
    const auto bucket = [&](u64 key) {
        const float sv = __uint_as_float((unsigned)(key >> 32));
        return flat ? 0 : (int)__builtin_fminf((sv - mn) * scale, (float)(CAP - 1));
    };
    u64 kreg[PER];
    int breg[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = lane + 64 * k;
        kreg[k] = i < n ? L.keys[i] : ~0ull;
        breg[k] = bucket(kreg[k]);
    }
    wave_fence();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (lane + 64 * k < n) atomicAdd(&L.hist[breg[k]], 1);
    wave_fence();
    int maxsize;
    {
        int c[PER], local = 0, cmax = 0;
        const int2 *h2 = reinterpret_cast<const int2 *>(&L.hist[lane * PER]);
#pragma unroll
        for (int k = 0; k < PER / 2; ++k) { const int2 v = h2[k]; c[2 * k] = v.x; c[2 * k + 1] = v.y; }
#pragma unroll
        for (int k = 0; k < PER; ++k) { local += c[k]; cmax = c[k] > cmax ? c[k] : cmax; }
        int run = wave_incl_add_dpp(local) - local;
        maxsize = wave_max_i32(cmax);
#pragma unroll
        for (int k = 0; k < PER; ++k) { const int t = c[k]; c[k] = run; run += t; }
        int2 *o2 = reinterpret_cast<int2 *>(&L.hist[lane * PER]);
#pragma unroll
        for (int k = 0; k < PER / 2; ++k) o2[k] = make_int2(c[2 * k], c[2 * k + 1]);
    }
    wave_fence();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (lane + 64 * k < n) L.keys[pbw_phys(atomicAdd(&L.hist[breg[k]], 1))] = kreg[k];     // every lane holds its keys: in place
    L.keys[pbw_phys(n + lane)] = ~0ull;
    wave_fence();   // hist[b] is now the END of bucket b
    const int per = (n + 63) >> 6;
    const int i0 = lane * per;
    u64 kq[PER];
    {
        int st[PER], rk[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const bool valid = k < per && i0 + k < n;
            kq[k] = valid ? L.keys[pbw_phys(i0 + k)] : ~0ull;
            const int b = bucket(kq[k]);
            st[k] = valid ? (b > 0 ? L.hist[b > 0 ? b - 1 : 0] : 0) : n;
            rk[k] = 0;
        }
        if (maxsize <= 64) {
            for (int t = 0; t < maxsize; ++t) {
#pragma unroll
                for (int k = 0; k < PER; ++k) rk[k] += L.keys[pbw_phys(st[k] + t)] < kq[k];
            }
        } else {      // a crowded bucket (clustered sums): same count with the reads clamped to the pad
            for (int t = 0; t < maxsize; ++t) {
#pragma unroll
                for (int k = 0; k < PER; ++k) { const int x = st[k] + t; rk[k] += L.keys[pbw_phys(x < n ? x : n)] < kq[k]; }
            }
        }
        wave_fence();
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (k < per && i0 + k < n) L.keys[pbw_phys(st[k] + rk[k])] = kq[k];
        wave_fence();
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) kq[k] = (k < per && i0 + k < n) ? L.keys[pbw_phys(i0 + k)] : 0ull;
    PBW_STAMP(kPwSort);
    // ---- equal sums: list order (pb_visit_less).  A lane looks at its own entries (registers) and at the two entries next
    // to them; the lane that owns the first entry of a run of equal sums puts the run in order.  (Not rare: a deep chunk
    // spans ~2 % of a binade, 377 sums among ~170 k floats collide in one chunk out of three.)
    {
        const unsigned sprev = i0 > 0 && i0 < n ? (unsigned)(L.keys[pbw_phys(i0 - 1)] >> 32) : 0xFFFFFFFFu;
        const unsigned snext = i0 + per < n ? (unsigned)(L.keys[pbw_phys(i0 + per)] >> 32) : 0xFFFFFFFFu;
        unsigned starts = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = i0 + k;
            if (k < per && i + 1 < n) {
                const unsigned sk = (unsigned)(kq[k] >> 32);
                const unsigned sn = (k + 1 < per) ? (unsigned)(kq[k + 1 < PER ? k + 1 : k] >> 32) : snext;
                const unsigned sp = k > 0 ? (unsigned)(kq[k > 0 ? k - 1 : 0] >> 32) : sprev;
                if (sn == sk && (i == 0 || sp != sk)) starts |= 1u << k;
            }
        }
        if (__ballot(starts != 0)) {
            bool degenerate = false;
            for (unsigned m = starts; m; m &= m - 1) {
                const int i = i0 + __builtin_ctz(m);
                const unsigned si = (unsigned)(L.keys[pbw_phys(i)] >> 32);
                int g = 2;
                while (i + g < n && g <= kPbMaxTie && (unsigned)(L.keys[pbw_phys(i + g)] >> 32) == si) ++g;
                if (g > kPbMaxTie) { degenerate = true; continue; }
                for (int a = 1; a < g; ++a) {
                    const u64 ka = L.keys[pbw_phys(i + a)];
                    const PbTep ta = pbw_tep((unsigned)ka);
                    int b = a;
                    while (b > 0 && pb_visit_less(L.w, ta, pbw_tep((unsigned)L.keys[pbw_phys(i + b - 1)]))) { L.keys[pbw_phys(i + b)] = L.keys[pbw_phys(i + b - 1)]; --b; }
                    L.keys[pbw_phys(i + b)] = ka;
                }
            }
            if (__ballot(degenerate)) return 2;
            wave_fence();
#pragma unroll
            for (int k = 0; k < PER; ++k) kq[k] = (k < per && i0 + k < n) ? L.keys[pbw_phys(i0 + k)] : 0ull;
        }
    }
    PBW_STAMP(kPwTie);
    // ---- evaluate: lane l owns the entries [l per, (l + 1) per) of the sorted chunk.  Rolled loops with the per-entry
    // values in LDS (the costs go where the bucket counters were): as register arrays, fully unrolled, they and the 64 LUT
    // reads the scheduler then hoists cost ~390 VGPRs -- one wavefront per SIMD.
    float *const costs = reinterpret_cast<float *>(L.hist);
    const auto parity = [&](const PbTep &t) {
        u64 D = d0 ^ L.P[t.p0];
        if (t.wt > 1) D ^= L.P[t.p1];
        if (t.wt > 2) D ^= L.P[t.p2];
        return D;
    };
    float tmin = __builtin_inff();
    int tdel = 0;
    // (rolled loops over the lane's entries, read back from LDS: this path only runs for the first chunk or two of a frame
    //  -- pbw_scan_chunk takes the others -- and unrolled it is 8 k instructions of a kernel that should fit the I-cache)
#pragma unroll 1
    for (int k = 0; k < per; ++k) {
        const int i = i0 + k;
        if (i < n) {
            const u64 key = L.keys[pbw_phys(i)];
            const PbTep t = pbw_tep((unsigned)key);
            const float c = pbw_cost<CAP>(L, __uint_as_float((unsigned)(key >> 32)), parity(t), S.best);   // (+inf if it cannot beat the best)
            costs[i] = c;
            tmin = __builtin_fminf(tmin, c);
            tdel += pb_delta(t, P.order);
        }
    }
    // exclusive scans over the lanes: min of the costs / sum of the frontier growth before my entries
    const float imin = wave_incl_min_dpp(tmin);
    const int iadd = wave_incl_add_dpp(tdel);
    float before = __shfl_up(imin, 1, 64);
    if (lane == 0) before = __builtin_inff();
    before = __builtin_fminf(before, S.best);
    int nlb = iadd - tdel + S.nlive;
    const int tot_del = __builtin_amdgcn_readlane(iadd, 63);
    PBW_STAMP(kPwEval);
    // ---- the sequential rules on my entries, assuming no earlier stop (`before` is the running best)
    int ones = 0, nev = 0, nnb = 0, lnb = -1, lstop = 0x7FFFFFFF, lreason = 0;
    float lbest = 0.0f;
    u64 lD = 0;
    unsigned lcode = 0;
#pragma unroll 1
    for (int k = 0; k < per; ++k) {
        const int i = i0 + k;
        if (i < n && lstop == 0x7FFFFFFF) {
            const u64 key = L.keys[pbw_phys(i)];
            const float c = costs[i];
            const PbTep t = pbw_tep((unsigned)key);
            float w1;
            const bool np = pb_not_promising(__uint_as_float((unsigned)(key >> 32)), before, Fr, P.c4, L.cdfA, L.cdfH, w1);
            ones += nlb == 1;
            nlb += pb_delta(t, P.order);
            if (np) { lstop = i; lreason = 1; }
            else {
                ++nev;
                if (c < before) {
                    const u64 D = parity(t);
                    before = c; lnb = i; ++nnb; lbest = c; lD = D; lcode = (unsigned)key;
                    if (pb_success_q(D, w1, L.qpar, Fr)) { lstop = i; lreason = 2; }
                }
            }
        }
    }
    PBW_STAMP(kPwRules);
    const int gstop = wave_min_i32(lstop);
    {   // my entries count if they lie before (or contain) the first stop
        const bool mine = i0 < n && i0 <= gstop;
        const int o = wave_add_i32(mine ? ones : 0), e = wave_add_i32(mine ? nev : 0), b = wave_add_i32(mine ? nnb : 0);
        const int l = wave_max_i32(mine ? lnb : -1);
        const int npop = gstop != 0x7FFFFFFF ? gstop + 1 : n;
        S.cmp += 2 * npop - o; S.suc1 += e; S.suc2 += b;
        if (l >= 0) {   // the last improvement before the stop
            const int src = __builtin_ctzll(__ballot(mine && lnb == l));
            const unsigned code = (unsigned)__builtin_amdgcn_readlane((int)lcode, src);
            const PbTep t = pbw_tep(code);
            u64 E = 1ull << t.p0;
            if (t.wt > 1) E |= 1ull << t.p1;
            if (t.wt > 2) E |= 1ull << t.p2;
            S.best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lbest), src));
            S.bestD = readlane64(lD, src);
            S.bestE = E;
            S.bestidx = S.j + l + 1;
        }
        if (gstop != 0x7FFFFFFF) {
            const int src = __builtin_ctzll(__ballot(lstop == gstop));
            stop = __builtin_amdgcn_readlane(lreason, src);
            ntep = S.j + gstop + 1;
            PBW_STAMP(kPwCombine);
            return 1;
        }
    }
    S.j += n; S.nlive += tot_del;
    PBW_STAMP(kPwCombine);
    return 0;
}

// The sorted path as a FUNCTION (not inlined): it runs for the first chunk or two of a frame that arrives without its head,
// for 0.05 % of the chunks otherwise, and inlined it sets the register peak of every kernel that contains it (key, bucket, start
// and rank arrays of eight entries each).  Arguments and results travel through L.sa, so that nothing is live across the call
// but what the caller chooses to keep (the committed cursors are re-read from L.cur, which holds them after a commit).
template <int CAP>
__device__ __noinline__ void pbw_sorted_chunk(PbWaveLds<CAP> &L)
{
    const int lane = threadIdx.x & 63;
    unsigned long long pt[kPwSlots], plast = 0;
    PbParams P;
    P.order = L.sa.order; P.c4 = L.sa.c4;
    const PbFrame Fr = L.sa.fr;
    const u64 d0 = L.sa.d0;
    const int n = L.sa.n;
    const float mn = L.sa.mn, mx = L.sa.mx;
    PbwState S = L.sa.S;
    int stop = L.sa.stop, ntep = L.sa.ntep;
    wave_fence();
    const int state = pbw_process_chunk<CAP, false>(L, P, Fr, d0, n, mn, mx, lane, S, stop, ntep, pt, plast);
    wave_fence();
    if (lane == 0) { L.sa.S = S; L.sa.state = state; L.sa.stop = stop; L.sa.ntep = ntep; }
    wave_fence();
}
// caller's side: park, call, take back
template <int CAP>
__device__ __forceinline__ int pbw_sorted_call(PbWaveLds<CAP> &L, const PbParams &P, const PbFrame &Fr, u64 d0, int n, float mn, float mx, int lane,
                                               PbwState &S, int &stop, int &ntep)
{
    wave_fence();
    if (lane == 0) {
        L.sa.S = S; L.sa.fr = Fr; L.sa.d0 = d0; L.sa.mn = mn; L.sa.mx = mx; L.sa.c4 = P.c4; L.sa.n = n; L.sa.order = P.order;
        L.sa.stop = stop; L.sa.ntep = ntep;
    }
    wave_fence();
    pbw_sorted_chunk<CAP>(L);
    wave_fence();
    S = L.sa.S; stop = L.sa.stop; ntep = L.sa.ntep;
    return L.sa.state;
}

// The sums (lo, T] once more with 8-byte keys: chunks of <= CAP keys, the sort-free pass with the keys in registers
// (pbw_scan_chunk) and, where that cannot settle a chunk either, the sorted path.  For a chunk of 4-byte keys that pbw_scan4
// gave up on (many improvement candidates, a tie against a reference key, a frontier that may shrink to one entry) and for the
// workgroup kernel's fallback (coop_solo_range).  L.cur holds the cursors of the range's start; arguments and results in L.ra
// (state 0 / 1 / 2 as in pb_wave_kernel).  A real function: its code (two more passes, the sort) stays out of the hot loop.
template <int CAP>
__device__ __noinline__ void pbw_redo_range(PbWaveLds<CAP> &L)
{
    const int lane = threadIdx.x & 63;
    unsigned long long pt[kPwSlots];
    PbParams P;
    P.order = L.ra.order; P.c4 = L.ra.c4;
    const PbFrame Fr = L.ra.fr;
    const u64 d0 = L.ra.d0;
    const float Tcap = L.ra.T, smax = L.ra.smax;
    const int target = L.ra.target;
    float lo = L.ra.lo;
    int done = L.ra.done;
    const int end = done + L.ra.n;
    const int nall = P.order > 2 ? kPbTabSize : (P.order > 1 ? kPbTriples0 : kPbPairs0);
    PbwState S = L.ra.S;
    int stop = L.ra.stop, ntep = L.ra.ntep;
    wave_fence();
    PbWalk W;
    pbw_cursors_load<CAP>(L, W.ecur, lane);     // (the walk needs nothing but the cursors)
    float tprev = 0.0f, nprev = 0.0f;
    int state = 0;
    while (state == 0 && done < end) {
        float T;
        int nwalks = 0;
        const int n = pbw_next_chunk<CAP, false, 0>(L, W, P.order, lo, done, nall, target, lane, T, tprev, nprev, nwalks, pt, Tcap);
        if (n <= 0) { state = 2; break; }
        wave_fence();
        const float cmn = lo < 0.0f ? L.w[63] : lo, cmx = T < __builtin_inff() ? T : smax;
        state = pbw_scan_chunk<CAP>(L, P, Fr, d0, n, cmn, cmx, lane, S, stop, ntep);
        if (state < 0) { state = pbw_sorted_call<CAP>(L, P, Fr, d0, n, cmn, cmx, lane, S, stop, ntep); pbw_cursors_load<CAP>(L, W.ecur, lane); }
        lo = T;
        done += n;
    }
    wave_fence();
    if (lane == 0) { L.ra.S = S; L.ra.state = state; L.ra.stop = stop; L.ra.ntep = ntep; }
    wave_fence();
}
// caller's side: park, call, take back (the caller has put the range's starting cursors into L.cur)
template <int CAP>
__device__ __forceinline__ int pbw_redo_call(PbWaveLds<CAP> &L, const PbParams &P, const PbFrame &Fr, u64 d0, float lo, float T, float smax, int done, int n,
                                             int lane, PbwState &S, int &stop, int &ntep)
{
    wave_fence();
    if (lane == 0) {
        L.ra.S = S; L.ra.fr = Fr; L.ra.d0 = d0; L.ra.lo = lo; L.ra.T = T; L.ra.smax = smax; L.ra.c4 = P.c4; L.ra.done = done; L.ra.n = n;
        L.ra.order = P.order; L.ra.target = CAP * 13 / 16; L.ra.stop = stop; L.ra.ntep = ntep;
    }
    wave_fence();
    pbw_redo_range<CAP>(L);
    wave_fence();
    S = L.ra.S; stop = L.ra.stop; ntep = L.ra.ntep;
    return L.ra.state;
}

// A long search handed from the chunk kernel to the workgroup kernel: ONE record per frame with everything the search needs,
// so that the receiving workgroup starts after a single wide load (its 1024 threads copy the record into LDS side by side)
// instead of the chain frame number -> source index -> permutation -> y that the chunk kernel went through:
//   words [0, 496)      the frame's tables as they stand in PbWaveLds (pad, w, P, zero row, cdfA, perm, tail, q), verbatim
//   words [496, 1008)   the committed cursors, [8][64]
//   words [1008, ...)   PbCarry: the search state (sums <= lo are visited) and the frame's scalars
struct PbCarry {
    float lo, best;
    int j, nlive, cmp, suc1, suc2, bestidx;
    u64 bestD, bestE;
    PbFrame fr;
    u64 d0, hm, hp;
    long long f;
    float tprev, nprev;      // the last chunk's lower bound and the TEPs before it (the growth exponent for the next bound)
};
constexpr int kPbFarProbe = 8;
constexpr int kPbRecPrefix = 496, kPbRecCur = 496, kPbRecScalars = 1008, kPbRecWords = 1040;
constexpr int kPbRecPerm = 330;       // (the permutation bytes inside the prefix: PbWaveLds::perm)
static_assert(kPbRecScalars * 4 % 8 == 0 && kPbRecScalars * 4 + sizeof(PbCarry) <= kPbRecWords * 4, "record layout");
static_assert(offsetof(PbWaveLds<kPbWaveCap>, cdfH) == kPbRecPrefix * 4 && offsetof(PbWaveLds<kPbWaveCap>, perm) == kPbRecPerm * 4, "the record's first part is the head of PbWaveLds");

// One frame of list A per wavefront, from its first TEP to its stop (or to the end of the table); massive ties go to
// list B (list replay).  Workgroup b serves sub-list b mod 16, entries b / 16, b / 16 + grid / 16, ... -- with the grid the
// launcher uses, ONE frame per workgroup: the hardware dispatcher then hands the next frame to whichever slot frees first.
// (Measured and dropped in round 4: persistent workgroups that draw their frames by ticket and fetch the next frame's list entry
//  and record while the current one is searched.  In-kernel stamps had put ~45 % of a wavefront's life into the dependent round
//  trips list entry -> record at a frame's start and the permutation / result stores at its end, and with the prefetch those
//  phases do vanish from the stamps -- but the launch got SLOWER: 359 -> 423 us at 2.5 dB, 5.03 -> 5.07 ms at 1.0 dB.  A frame
//  bound early to a wavefront that is busy with a long search starts late, and the launch is its tail: 3.4 frames per
//  resident wavefront at 2.5 dB; the other wavefronts of the SIMD had been hiding those round trips anyway.)
// (10 080 B of LDS per frame: 16 workgroups per CU; four wavefronts per SIMD asked of the register allocator: 128 VGPRs)
template <int CAP, bool PROF, int ROT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void pb_wave_kernel(PbParams P,
                                                     const double *__restrict__ cdf_half, int *__restrict__ ctl,
                                                     const int *__restrict__ listA, int *__restrict__ listB, int sub_cap,
                                                     unsigned *__restrict__ carry,
                                                     const unsigned *__restrict__ recs, PbOut O, unsigned long long *__restrict__ prof_out)
{
    __shared__ PbWaveLds<CAP> L;
    unsigned long long pt[kPwSlots] = {0}, plast = 0;
    if constexpr (PROF) plast = __builtin_amdgcn_s_memtime();
    const int lane0 = threadIdx.x;
    const int sub = blockIdx.x & (kPbSub - 1);
    // (the sub-list's length and this workgroup's entry of it are asked for together -- entry k0 < sub_cap exists whatever the
    //  length says --, through vector loads: a scalar load whose result meets a branch is waited for where it is issued)
    int vz = 0;
    asm volatile("" : "+v"(vz));
    const int k0 = blockIdx.x >> 4;
    const int lenv = ctl[kPbCtlLenA + kPbCtlLine * sub + vz], f0v = listA[sub * sub_cap + k0 + vz];
    const int len = __builtin_amdgcn_readfirstlane(lenv);
    const int nall = P.order > 2 ? kPbTabSize : (P.order > 1 ? kPbTriples0 : kPbPairs0);      // TEPs of weight 1..order
    // TEPs after which a search may leave for the workgroup kernel: the fewer frames search, the sooner (a lone wavefront
    // takes ~30 us per chunk of ~400 TEPs, the workgroup ~12 us per chunk of ~2700; the schedule and its measurements: launch_pb)
    const int budget0 = len < 128 ? P.budget_s : (len < 448 ? P.budget_m : (len < 1400 ? P.budget : (len < 3000 ? P.budget_l : P.budget_xl)));
    bool have_cdfh = false;
    for (int k = k0; k < len; k += gridDim.x >> 4) {
        // (an opaque copy of the lane number per frame: otherwise every lane-dependent constant of the frame's code -- item
        //  geometry, the rank sort's tie masks -- is hoisted out of this loop, kept in registers for the whole kernel and spilled)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        if (!have_cdfh) {
            L.cdfH[lane] = (float)cdf_half[lane];
            if (lane == 0) L.cdfH[64] = (float)cdf_half[64];
            if (lane < 4) L.pre[lane] = pbw_nan();
            have_cdfh = true;
        }
        // The launch's TAIL: once the dispatcher has no frame left to hand out, every search that goes on here keeps its CU
        // from idling only by itself -- the launch then lasts as long as the longest of them (measured with the frames sorted by
        // search length, which no product path can do: 260 -> 204 us at 2.5 dB, 3.62 -> 3.18 ms at 1.0 dB).  So the frames count
        // themselves out as they finish, and a search that finds (after a chunk) that fewer frames of its sub-list are unfinished
        // than late_pct % of the chip's wavefront slots -- the sub-lists advance side by side: workgroup b serves entry b / 16 of
        // sub-list b mod 16 -- leaves for the workgroup kernel after budget / late_div TEPs instead of budget.  WHERE a search is
        // handed on depends on timing then; what it returns does not (tests/test_gpu_osd_pb.py runs the kernels under several
        // schedules).
        int *const finished = &ctl[kPbCtlStartA + kPbCtlLine * sub];
        const bool tail_rule = len * kPbSub > P.late_min && len < P.late_maxlen;
        const int tail_budget = budget0 / P.late_div;
        const long long tail_slots = 4096ll * P.late_pct;       // (256 CUs x 16 wavefronts of this kernel, in per cent)
        const long long f = k == k0 ? __builtin_amdgcn_readfirstlane(f0v) : listA[sub * sub_cap + k];
        // ---- per-frame set-up: ONE wide load of the record pb_singles_kernel wrote (|y'|, P', the CDF table, the permutation: 356
        // words, copied into LDS as they are), the frame's scalars by scalar loads; derived here: the cost-bound table, the
        // success-rule factors
        const unsigned *const rec = recs + f * kPbR1Words;
        {
            const uint4 *const r4 = reinterpret_cast<const uint4 *>(rec);
            uint4 *const l4 = reinterpret_cast<uint4 *>(L.w);
            const uint4 a = r4[lane];
            uint4 b = make_uint4(0, 0, 0, 0);
            if (lane < 26) b = r4[64 + lane];
            if constexpr (PROF) { unsigned t = a.x; asm volatile("s_waitcnt vmcnt(0)" : "+v"(t)); PBW_STAMP(kPwLoad1); }
            l4[lane] = a;
            if (lane < 26) l4[64 + lane] = b;
        }
        const PbHead &H = *reinterpret_cast<const PbHead *>(rec + kPbR1Head);
        const PbFrame Fr = H.fr;
        const u64 d0 = H.d0;
        wave_fence();
        PBW_STAMP(kPwLoad2);
        {   // pbw_cost_floor's table: every quarter's 16 parity weights in ascending order (rank sort inside the 16-lane row,
            // ties by position; no assumption on the order the caller's front end left them in), their running sums
            const float v = L.w[64 + lane];
            const int g0 = lane & 48;
            int r = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) { const float o = L.w[64 + g0 + u]; r += (o < v) || (o == v && g0 + u < lane); }
            float *const srt = reinterpret_cast<float *>(L.keys);
            srt[g0 + r] = v;
            L.qpar[lane] = 1.0f / (1.0f + det_expf(-(P.c4 * v)));            // sigmoid(c4 |y'_p|), as pb_frame_setup computes it
            wave_fence();
            float acc = srt[lane];
            acc = acc + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x111, 0xF, 0xF, true));   // row_shr:1,2,4,8:
            acc = acc + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x112, 0xF, 0xF, true));   // running sums
            acc = acc + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x114, 0xF, 0xF, true));   // inside a row
            acc = acc + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x118, 0xF, 0xF, true));
            L.tail[lane >> 4][(lane & 15) + 1] = acc * 0.99999f;
            if ((lane & 15) == 0) L.tail[lane >> 4][0] = 0.0f;
        }
        wave_fence();
        PbwState S;
        S.best = H.hbest;        // (the order-0 metric, or what the head made of it)
        S.j = 0; S.nlive = 1; S.cmp = 0; S.suc1 = 0; S.suc2 = 0; S.bestidx = 0; S.bestD = d0; S.bestE = 0;
        PbWalk W;
        pbw_walk_init<CAP>(L, W, P.order, lane);
        float lo = -1.0f;
        int done = 0;
        {   // go on where pb_singles_kernel stopped: the nhead least reliable singles are visited (every other sum is larger)
            const int nh = H.nhead;
            if (nh > 0) {
                S.j = nh; S.nlive = nh; S.cmp = 2 * nh - (nh < 2 ? nh : 2); S.suc1 = nh; S.suc2 = H.hsuc2;
                if (H.hsuc2 > 0) { S.bestidx = H.hbestidx; S.bestD = H.hbestD; S.bestE = H.hbestE; }
                lo = L.w[64 - nh]; done = nh;
                if (lane == 63)     // the singles are item 31 of lane 63: its cursor
                    W.ecur[7] = (W.ecur[7] & 0x00FFFFFFu) | ((unsigned)(64 - nh) << 24);
                L.cur[7][lane] = W.ecur[7];
            }
        }
        PBW_STAMP(kPwSetup);
        const float smax = P.order > 2 ? (L.w[0] + L.w[1]) + L.w[2] : (P.order > 1 ? L.w[0] + L.w[1] : L.w[0]);
        int stop = 0, ntep = P.nmax, state = 0;   // state: 0 = searching, 1 = a rule fired, 2 = to the list replay, 3 = to the workgroup kernel
        bool asked = false, firstc = true;       // (firstc: the frame's first chunk here -- its size is tuned apart, many searches end in it)
        float tprev = 0.0f, nprev = 0.0f;
        while (state == 0 && done < nall) {
            float T;
            int nwalks = 0;
            // (the tail rule's counter, asked for here and looked at after the chunk: the round trip hides behind the walk)
            int nstarted = 0;
            if (tail_rule && !asked) nstarted = __hip_atomic_load(finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // a chunk of 4-byte keys (up to 2 CAP + 64 of them); W.ecur keeps the cursors it started from until it is judged
            const int n = pbw_next_chunk<CAP, PROF, ROT, true, false>(L, W, P.order, lo, done, nall, firstc ? P.t1 : P.t2, lane, T, tprev, nprev, nwalks, pt);
            firstc = false;
            PBW_STAMP(kPwWalk);
            if constexpr (PROF) { pt[kPwChunks] += 1; pt[kPwWalks] += nwalks; pt[kPwKeys] += n > 0 ? n : 0; }
            if (n < 0) { state = 2; break; }
            if (n == 0) break;
            wave_fence();
            const float cmn = lo < 0.0f ? L.w[63] : lo, cmx = T < __builtin_inff() ? T : smax;
            state = pbw_scan4<CAP>(L, P, Fr, d0, n, cmn, cmx, lane, S, stop, ntep);
            PBW_STAMP(kPwScan);
            if (state < 0) {     // not settled without a sort: the same sums once more, in chunks of 8-byte keys (a function call)
                if constexpr (PROF) pt[kPwSorted] += 1;
                pbw_cursors_store<CAP>(L, W.ecur, lane);
                state = pbw_redo_call<CAP>(L, P, Fr, d0, lo, T, smax, done, n, lane, S, stop, ntep);
            }
            pbw_cursors_load<CAP>(L, W.ecur, lane);          // commit (L.cur: the cursors behind the chunk, whoever walked it)
            lo = T;
            done += n;
            const int budget = (tail_rule && (long long)(len - nstarted) * kPbSub * 100 <= tail_slots) ? tail_budget : budget0;
            if (state == 0 && done >= budget && !asked && done < nall && len < P.handoff_maxlen) {
                // a long search: the workgroup kernel takes it over if it still has room (at most kPbCoopHalf frames per half of list C and call)
                asked = true;
                // The workgroup kernel's launch is as long as its last frame: it serves first the searches that will run far.
                // Rule 1 (acquire_prob_promising) against the best so far, at 64 sums between here and the largest sum: a
                // search whose rule cannot fire in the first kPbFarProbe of them goes into the first half of list C (an
                // improvement of the best only shortens a search: the long ones are all there).  Per search call, all in one
                // half / threshold 6 / 8 / 12 (means over four batches x 12 launches: single launches scatter by +-15 %, where a
                // search is handed on depends on timing): 2.5 dB 0.479 / 0.469 / 0.475 / 0.468 ms, 2.0 dB 0.988 / 0.963 / 0.955 / 0.970,
                // 1.0 dB 3.80 / 3.76 / 3.75 / 3.75.
                int far;
                {
                    const float rp = lo + (smax - lo) * ((float)(lane + 1) * (1.0f / 64.0f));
                    float w1;
                    const float bs = pb_promising_bs(rp, S.best, Fr, P.c4, L.cdfA, L.cdfH, w1);
                    const u64 fires = __ballot((double)bs < Fr.p_t_pro);
                    far = (fires ? __builtin_ctzll(fires) : 64) >= kPbFarProbe;
                }
                int slot = 0;
                if (lane == 0) slot = atomicAdd(&ctl[far ? kPbCtlLenC : kPbCtlLenC2], 1);
                slot = __builtin_amdgcn_readfirstlane(slot);
                if (slot < kPbCoopHalf) {
                    if (!far) slot += kPbCoopHalf;
                    unsigned *const crec = carry + (long long)slot * kPbRecWords;
                    const unsigned *const Lw = reinterpret_cast<const unsigned *>(&L);
                    for (int k = lane; k < kPbRecPrefix; k += 64) crec[k] = Lw[k];
#pragma unroll
                    for (int k = 0; k < 8; ++k) crec[kPbRecCur + k * 64 + lane] = W.ecur[k];
                    if (lane == 0) {
                        PbCarry c;
                        c.lo = lo; c.best = S.best; c.j = S.j; c.nlive = S.nlive; c.cmp = S.cmp; c.suc1 = S.suc1; c.suc2 = S.suc2;
                        c.bestidx = S.bestidx; c.bestD = S.bestD; c.bestE = S.bestE;
                        c.fr = Fr; c.d0 = d0; c.hm = H.hm; c.hp = H.hp; c.f = f; c.tprev = tprev; c.nprev = nprev;
                        *reinterpret_cast<PbCarry *>(crec + kPbRecScalars) = c;
                    }
                    state = 3;
                }
            }
        }
        if (state == 2) {   // massive ties: the literal list replay decodes this frame
            if (lane == 0) listB[atomicAdd(&ctl[kPbCtlLenB], 1)] = (int)f;
        } else if (state != 3) {   // candidate (E = flipped MRB positions, D = parity discrepancy) -> codeword in ORIGINAL bit order
            if (lane < 2) L.cw[lane] = 0;
            wave_fence();
            const int o1 = L.perm[lane], o2 = L.perm[64 + lane];       // (original bit index of primed positions lane, 64 + lane)
            const u64 mrb_bits = H.hm ^ S.bestE, par_bits = S.bestD ^ H.hp;
            if ((mrb_bits >> lane) & 1) atomicOr(&L.cw[o1 >> 6], 1ull << (o1 & 63));
            if ((par_bits >> lane) & 1) atomicOr(&L.cw[o2 >> 6], 1ull << (o2 & 63));
            wave_fence();
            PBW_STAMP(kPwFinish);
            if (lane < 2) O.cw[f * 2 + lane] = L.cw[lane];
            if (lane == 0) {
                if (O.metric) O.metric[f] = S.best;
                if (O.best) O.best[f] = S.bestidx;
                if (O.ntep) O.ntep[f] = ntep;
                if (O.aux) { O.aux[f * 4] = S.cmp; O.aux[f * 4 + 1] = S.suc1; O.aux[f * 4 + 2] = S.suc2; O.aux[f * 4 + 3] = stop; }
            }
            PBW_STAMP(kPwStore);
            if constexpr (PROF) pt[kPwFrames] += 1;
        }
        if (lane == 0) atomicAdd(finished, 1);
        wave_fence();
    }
    if constexpr (PROF) { if (lane0 == 0 && pt[kPwFrames]) for (int k = 0; k < kPwSlots; ++k) atomicAdd(&prof_out[k], pt[k]); }
}

// ---------------------------------------------------------------------------------------
// stage 2c: long searches, ONE WORKGROUP OF NW WAVEFRONTS PER FRAME (round 3).  The chunk pass above needs no sort, so a
// chunk is embarrassingly parallel over its keys, and the walk is parallel over the item rows: wavefront v owns 32 / NW of
// the 32 rows (dealt so that light and heavy rows pair up), walks them into the workgroup's ONE key buffer (slots reserved by
// an LDS atomic per trip, so the buffer fills evenly whatever the rows give) and scans every NW-th 64-key slice of it.
// What the wavefronts exchange per chunk is a handful of numbers (counts, frontier growth, the smallest firing sum,
// positions) through LDS words and ~7 workgroup barriers; the order of the keys in the buffer depends on the wavefronts'
// timing, the results do not (the pass takes counts, minima and the candidates ordered by sum, a tie leaves it).  Everything a
// wavefront decides from the exchanged numbers it decides like the others (same values, same arithmetic), so the barriers
// line up by construction: no barrier stands inside control flow that depends on a wavefront's own keys.
// A chunk the sort-free pass cannot settle (a tie against a reference key, > 64 candidates, a frontier that may shrink to
// one entry) is redone by wavefront 0 alone with the single-wavefront code above (sorted path, sub-chunks of <= 512 keys up
// to the same bound), then the co-operation resumes.  Frames arrive from pb_wave_kernel with their search state and cursors.
// ---------------------------------------------------------------------------------------
constexpr int kCoopCap = 4096;        // keys of a chunk, all wavefronts together (8192 with a target of 6144 measured in round 4: fewer, longer
                                      // chunks -- per search call 0.428 -> 0.411 ms at 2.5 dB, 0.926 -> 0.941 at 2.0 dB, 3.75 -> 3.84 at 1.0 dB)
constexpr int kCoopMaxCand = 64;
// Wavefronts per frame: 16 (one frame per CU) or 8 (two).  Measured with 8 (round 4), workgroup kernel per
// launch / PB kernels per search call: 2.5 dB 120 -> 151 us / 0.426 -> 0.451 ms (the launch is its longest search, and a chunk
// takes eight wavefronts 1.25 x as long), 2.0 dB 270 -> 200 us / 0.945 -> 0.905 ms (throughput-bound: ~1900 searches),
// 1.5 dB 1.89 -> 1.89 ms, 1.0 dB 245 -> 252 us / 3.75 -> 3.74 ms (its tail again), 3.0 / 3.5 dB 0.287 -> 0.292 / 0.177 -> 0.186 ms.
constexpr int kPbCoopW = 16;
constexpr int kPbCoopGrid = 256 * 16 / kPbCoopW;      // workgroups (frames are handed out by ticket): what the chip holds at once

template <int NW>
struct __attribute__((aligned(16))) PbCoopLds {
    static constexpr int NI = 32 / NW;                    // items per lane
    static constexpr int WK = 2 * kCoopCap / NW;          // keys of a chunk from one wavefront (an NW-th of the items +- 10 % when the chunk is full: half of this)
    PbWaveLds<kPbWaveCap> one;        // the frame's tables (tail, P, w, tq, cdf); the rest of it is wavefront 0's when it works alone
    u64 keys[kCoopCap];
    unsigned short slist[NW][WK];              // a wavefront's keys that the cost bound could not rule out
    uint4 desc[NW][64];                                    // a wavefront's items with members in the chunk (emission)
    u64 ck[kCoopMaxCand], rk[kCoopMaxCand];
    float cc[kCoopMaxCand], rc[kCoopMaxCand];
    int red[8][NW], red2[8][NW];
    int ncand[2], nrec, stop2, tie0, ticket[2];
    PbwState bs;                      // wavefront 0 -> all, after it worked alone
    int bstate, bstop, bntep;
    // What wavefront 0 holds in registers when it leaves for coop_solo_range, parked here and read back afterwards: a value
    // that is live across the call costs a callee-saved register or a scratch slot on EVERY path through the kernel (the
    // call alone took the kernel from 119 VGPRs and no scratch to 128 VGPRs and 101 spilled ones, reloaded inside the walk's
    // loops), a value that is stored before it and loaded after it costs nothing anywhere else.
    int sv[32 / NW][64];
    struct {
        PbFrame fr;
        u64 d0;
        float lo, T, tprev, nprev, smax;
        int done, n, seq, par, it, tk, next_tk;
    } su;
    PbParams sP;
    unsigned long long prof[32];      // diagnostic build only
};

// in-kernel stamps of the diagnostic instantiation (thread 0 of the workgroup), as PBW_STAMP above
enum { kPcSetup = 0, kPcWalk, kPcScan, kPcSolo, kPcOut, kPcFrames, kPcChunks, kPcSolos, kPcSoloKeys, kPcKeys,
       kPcWalks, kPcTrips, kPcWList, kPcWDense, kPcWCollect, kPcWBarrier, kPcWPick, kPcSProbe, kPcSKeys, kPcSBar1, kPcSSurv, kPcSBar2, kPcSCand, kPcSMin, kPcSPos, kPcSlots };
struct PbcProf {
    unsigned long long *pc;      // the workgroup's counters (LDS): registers would cost the diagnostic build ~50 VGPRs and distort it
    unsigned long long last;
};
#define PBC_STAMP(k) do { if constexpr (PROF) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) Q.pc[k] += now__ - Q.last; Q.last = now__; } } while (0)
#define PBC_COUNT(k, v) do { if constexpr (PROF) { if (threadIdx.x == 0) Q.pc[k] += (v); } } while (0)

template <int NW>
struct CoopRed {
    int (*buf)[NW];
    int seq, wave, lane, par;
};
// sum / minimum of one int per wavefront, the same value in every lane of every wavefront (one barrier; eight rotating slots:
// a slot is rewritten seven barriers after its last reader)
template <int NW>
__device__ __forceinline__ int coop_sum(CoopRed<NW> &R, int x)
{
    int *const s = R.buf[R.seq++ & 7];
    if (R.lane == 0) s[R.wave] = x;
    __syncthreads();
    int a = 0;
#pragma unroll
    for (int v = 0; v < NW; ++v) a += s[v];
    return a;
}
template <int NW>
__device__ __forceinline__ int coop_min(CoopRed<NW> &R, int x)
{
    int *const s = R.buf[R.seq++ & 7];
    if (R.lane == 0) s[R.wave] = x;
    __syncthreads();
    int a = s[0];
#pragma unroll
    for (int v = 1; v < NW; ++v) a = s[v] < a ? s[v] : a;
    return a;
}

// The 2048 items (row q, lane l) are dealt to the wavefronts by (5 l + 3 q) mod 16: every wavefront holds four items of
// every row, spread over the lanes, and a chunk's keys come from the sixteen wavefronts within ~10 % of each other
// (dealing whole rows: the fullest wavefront emits 1.4-1.9 times the mean, the others wait for it at every exchange).
// Item j (0, 1) of lane `lane` in wavefront v:
template <int NW>
__device__ __forceinline__ void coop_item(int v, int j, int lane, int &q, int &l)
{
    static_assert(NW == 16 || NW == 8, "the dealing is written for sixteen or eight wavefronts");
    if constexpr (NW == 16) {
        q = 16 * j + (lane >> 2);
        l = ((13 * (v - 3 * q)) & 15) + 16 * (lane & 3);        // 5 l = v - 3 q (mod 16), 5 * 13 = 1
    } else {            // (5 l + 3 q) mod 8: eight items of every row, four rows of a lane
        q = 8 * j + (lane >> 3);
        l = ((5 * (v - 3 * q)) & 7) + 8 * (lane & 7);           // 5 l = v - 3 q (mod 8), 5 * 5 = 1
    }
}

// A wavefront's items in registers: item j of this lane is (row q, lane l) = coop_item(v, j, lane); sb = the sum of its fixed
// positions, members are m in (base, 63] visited downwards from a - 1 (a = cursor: [a, 64) are visited), key = code | m << sh.
template <int NI>
struct CoopItems {
    float sb[NI];
    int base[NI], sh[NI], a[NI];
    unsigned code[NI];
};
template <int NW>
__device__ __forceinline__ void coop_items_init(CoopItems<32 / NW> &I, const float *w, int order, int lane, int wave)
{
#pragma unroll
    for (int j = 0; j < 32 / NW; ++j) {
        int q, l;
        coop_item<NW>(wave, j, lane, q, l);
        const PbwItem it = pbw_item_rt(q, l);
        const bool live = q < 31 ? order > 2 : (order > 1 || l == 63);
        I.sb[j] = q < 31 ? w[it.i] + w[it.j] : (l <= 62 ? w[l] : 0.0f);
        I.base[j] = live ? it.base : 64;      // (an item the order excludes: no members)
        I.sh[j] = it.sh; I.code[j] = it.code;
    }
}

// The members of every item with sum <= T beyond its cursor: sums fall as the position rises, so they are the positions
// [first, a) and `first` is found by bisection (7 steps of one LDS read, the items of a lane side by side).  Nothing is
// written: the count of a bound is a pure function of the cursors, a bound that gives too many or too few keys costs one
// more count, and the keys are generated once, for the bound that is taken (the round-3 list walk emitted as it went and
// rolled the cursors back: a chain of four dependent LDS round trips per 64 entries, ~2000 cycles each, and a list per wavefront).
template <int NI>
__device__ __forceinline__ void coop_count(const CoopItems<NI> &I, const float *w, float T, int (&first)[NI])
{
    int lo[NI], hi[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { lo[j] = I.base[j] + 1; hi[j] = I.a[j]; }
#pragma unroll
    for (int s = 0; s < 7; ++s) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const bool act = lo[j] < hi[j];
            const int mid = (lo[j] + hi[j]) >> 1;
            const bool f = I.sb[j] + w[mid & 63] <= T;
            hi[j] = act && f ? mid : hi[j];
            lo[j] = act && !f ? mid + 1 : lo[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) first[j] = hi[j];
}

// The next chunk of the workgroup: the same bound T in every wavefront, sized by the SUM of their counts (one barrier per
// count), then the keys of (lo, T] into L.keys[0 .. n) -- wavefront v's behind those of the wavefronts before it, a lane's
// behind those of the lanes before it -- and the cursors advanced.  Returns n, -1: cannot be split, 0: nothing left;
// kbase / kcount: this wavefront's own keys (which it also scans: no barrier between the two).
template <int NW, bool PROF>
__device__ __forceinline__ int coop_next_chunk(PbCoopLds<NW> &L, CoopRed<NW> &R, CoopItems<32 / NW> &I, float lo, int done,
                                               int nall, int target, int lane, int wave, float &Tout, float &tprev, float &nprev, int &kbase, int &kcount,
                                               PbcProf &Q)
{
    constexpr int NI = 32 / NW;
    const float inf = __builtin_inff();
    const float *w = L.one.w;
    const float m3 = (w[61] + w[62]) + w[63];
    const float want = (float)(done + target);
    float Tl = lo, Th = inf;
    float T = nall - done <= kCoopCap / 2 ? inf : m3 * pb_bound_guess(want);
    if (tprev > 0.0f && nprev > 0.0f && lo > tprev && (float)done > nprev && T < inf) {
        const float pe = (__builtin_amdgcn_logf((float)done) - __builtin_amdgcn_logf(nprev)) / (__builtin_amdgcn_logf(lo) - __builtin_amdgcn_logf(tprev));
        if (pe > 1.5f && pe < 20.0f) T = lo * __builtin_amdgcn_exp2f((__builtin_amdgcn_logf(want) - __builtin_amdgcn_logf((float)done)) / pe);
    }
    if (!(T > lo)) T = lo > 0.0f ? lo * 1.05f : w[0];
    float tp = lo, np_ = (float)done;
    int first[NI], slot = 0, tot = 0;
    float T_ok = lo;          // the largest bound counted so far that fits (tot_ok keys)
    int tot_ok = 0;
    bool last_ok = false;     // first / slot / tot describe T_ok
    for (int it = 0; it < 48; ++it) {
        PBC_STAMP(kPcWPick);
        PBC_COUNT(kPcWalks, 1);
        coop_count<NI>(I, w, T, first);
        int c = 0;
#pragma unroll
        for (int j = 0; j < NI; ++j) c += I.a[j] - first[j];
        PBC_STAMP(kPcWDense);
        slot = R.seq & 7;
        {
            const int cw = wave_add_i32(c);
            tot = coop_sum<NW>(R, cw > PbCoopLds<NW>::WK ? cw + (1 << 24) : cw);     // (a wavefront scans its own keys: at most WK)
        }
        PBC_STAMP(kPcWBarrier);
        last_ok = false;
        bool over = false;
        if (tot > kCoopCap) {
            over = true;
            Th = T;
            if (tot_ok > 0 && it >= 3) break;        // (a usable smaller chunk is known: take it rather than search on)
        } else if (tot > 0 && (!(T < inf) || 5 * tot >= 2 * target || it >= 3)) {
            T_ok = T; tot_ok = tot; last_ok = true;
            break;
        } else if (!(T < inf)) {
            return 0;                                // everything that can be visited has been
        } else {
            if (tot > 0) { T_ok = T; tot_ok = tot; last_ok = true; }
            Tl = T;
        }
        float Tn;
        if (over) {
            Tn = Tl > 0.0f ? Tl + (Th - Tl) * 0.5f : Th * 0.9f;
        } else {
            const float totf = (float)(done + tot);
            float p = 6.0f;
            if (tp > 0.0f && np_ > 0.0f && totf != np_ && T != tp) {
                const float pe = (__builtin_amdgcn_logf(totf) - __builtin_amdgcn_logf(np_)) / (__builtin_amdgcn_logf(T) - __builtin_amdgcn_logf(tp));
                if (pe > 1.5f && pe < 20.0f) p = pe;
            }
            if (tot > 0) { tp = T; np_ = totf; }
            Tn = tot > 0 ? T * __builtin_amdgcn_exp2f((__builtin_amdgcn_logf(want) - __builtin_amdgcn_logf(totf)) / p) : T * 1.1f;
            if (it >= 6 || !(Tn > Tl) || !(Tn < Th)) Tn = Th < inf ? Tl + (Th - Tl) * 0.5f : T * 1.2f;
        }
        if (!(Tn > Tl) || !(Tn < Th)) break;
        T = Tn;
    }
    if (tot_ok == 0) return -1;
    if (!last_ok) {     // the bound taken is not the one counted last: count it again (rare)
        coop_count<NI>(I, w, T_ok, first);
        int c = 0;
#pragma unroll
        for (int j = 0; j < NI; ++j) c += I.a[j] - first[j];
        slot = R.seq & 7;
        tot = coop_sum<NW>(R, wave_add_i32(c));
    }
    kbase = 0;
#pragma unroll
    for (int v = 0; v < NW; ++v) kbase += v < wave ? L.red[slot][v] : 0;
    kcount = L.red[slot][wave];
    // ---- the keys.  Offsets: wavefronts before mine (the counts just exchanged), lanes before mine, my items in turn.
    int off = kbase;
    int o[NI], cj[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        cj[j] = I.a[j] - first[j];
        const int inc = wave_incl_add_dpp(cj[j]);
        o[j] = off + inc - cj[j];
        off += __builtin_amdgcn_readlane(inc, 63);
    }
    // Emission.  In a chunk ~26 of a wavefront's 128 items have members, ~6.5 each, at most ~15 (at 2.5 dB): a loop over the
    // member number would run at a quarter of the lanes, a loop over the items at a tenth.  So the items with members leave
    // a 16-byte descriptor in LDS (ballot + mbcnt), and every pass serves EIGHT of them, eight lanes each, a member per lane.
    uint4 *const desc = L.desc[wave];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const u64 am = __ballot(cj[j] > 0);
        const int na = __popcll(am);
        if (cj[j] > 0)
            desc[(int)__builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u))] =
                make_uint4((unsigned)I.a[j] | ((unsigned)cj[j] << 8) | ((unsigned)I.sh[j] << 16), (unsigned)o[j], __float_as_uint(I.sb[j]), I.code[j]);
        wave_fence();
        for (int p0 = 0; p0 < na; p0 += 8) {
            const int it = p0 + (lane >> 3);
            if (it < na) {
                const uint4 d = desc[it];
                const int a_ = (int)(d.x & 255u), c_ = (int)((d.x >> 8) & 255u), sh_ = (int)(d.x >> 16);
                const float sb_ = __uint_as_float(d.z);
                for (int k = lane & 7; k < c_; k += 8) {
                    const int m = a_ - 1 - k;
                    L.keys[(int)d.y + k] = ((u64)__float_as_uint(sb_ + w[m]) << 32) | (d.w | ((unsigned)m << sh_));
                }
            }
        }
        wave_fence();
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) I.a[j] = first[j];
    PBC_STAMP(kPcWCollect);
    tprev = lo; nprev = (float)done;
    Tout = T_ok;
    return tot_ok;
}

// pbw_scan_chunk for the workgroup: every wavefront scans the keys it generated (L.keys[kbase .. kbase + kcount)); the
// reductions go through LDS.  Returns 0 / 1 / -1 like pbw_scan_chunk, the same value in every wavefront; -1 leaves the
// state untouched.  The keys are not kept in registers: the ordinary chunk reads each once, the rare paths read them again.
template <int NW, bool PROF>
__device__ __forceinline__ int coop_scan_chunk(PbCoopLds<NW> &L, CoopRed<NW> &R, const PbParams &P, const PbFrame &Fr, u64 d0, int n, int kbase, int kcount,
                                               float mn, float mx, int lane, int wave, PbwState &S, int &stop, int &ntep, PbcProf &Q)
{
    constexpr int CAP = kPbWaveCap, PER = PbCoopLds<NW>::WK / 64;
    const PbWaveLds<CAP> &T0 = L.one;
    if (S.nlive <= 1) return -1;
    const float best0 = S.best;
    float r_safe;     // rule 1 by probes (pbw_scan_chunk)
    {
        const float rp = lane == 63 ? mx : mn + (mx - mn) * ((float)(lane + 1) * (1.0f / 64.0f));
        float w1;
        const float bs = pb_promising_bs(rp, best0, Fr, P.c4, T0.cdfA, T0.cdfH, w1);
        const u64 unsafe = ~__ballot((double)bs > Fr.p_t_pro * 1.001);
        const int u = unsafe ? __builtin_ctzll(unsafe) : 64;
        r_safe = u == 0 ? -1.0f : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rp), u - 1));
    }
    PBC_STAMP(kPcSProbe);
    const auto parity = [&](const PbTep &t) {
        u64 D = d0 ^ T0.P[t.p0];
        if (t.wt > 1) D ^= T0.P[t.p1];
        if (t.wt > 2) D ^= T0.P[t.p2];
        return D;
    };
    const auto sumbits = [](u64 key) { return (unsigned)(key >> 32); };
    // a is visited before b (a != b): by sum, equal sums by the list-slot order (pb_visit_less: through the parents).  Round 4: a tie
    // against a reference key used to send the whole chunk, ~2700 keys, to wavefront 0 alone -- 1-4 times per launch at 2.5 dB,
    // 35-135 us each where the launch takes ~150.
    const auto visited_before = [&](u64 a, u64 b) {
        if (sumbits(a) != sumbits(b)) return sumbits(a) < sumbits(b);
        return pb_visit_less(T0.w, pbw_tep((unsigned)a), pbw_tep((unsigned)b));
    };
    const auto key_at = [&](int k) { const int i = k * 64 + lane; return i < kcount ? L.keys[kbase + i] : ~0ull; };   // (an empty slot: all ones, sum NaN)
    int sumdel = 0, neg = 0, nsurv = 0;
    unsigned fs = 0x7FFFFFFFu;     // the smallest sum on which rule 1 fires (against the chunk-start best)
    unsigned short *const slist = L.slist[wave];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (k * 64 < kcount) {      // (uniform: a wavefront holds ~n / NW keys, four or five slices of the eight)
            const u64 key = key_at(k);
            const bool valid = k * 64 + lane < kcount;
            const PbTep t = pbw_tep((unsigned)key);
            const float rs = __uint_as_float(sumbits(key));
            const bool surv = valid && pbw_cost_floor<CAP>(T0, rs, parity(t)) < best0;
            const u64 sm = __ballot(surv);
            if (sm) {
                if (surv) slist[nsurv + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, 0u))] = (unsigned short)(kbase + k * 64 + lane);
                nsurv += __popcll(sm);
            }
            const int dl = valid ? pb_delta(t, P.order) : 0;
            sumdel += dl; neg += dl < 0;
            const bool need = rs > r_safe;      // above the last safe probe (the last chunk of a search): the rule itself
            if (__ballot(need)) {
                float w1;
                if (need && pb_not_promising(rs, best0, Fr, P.c4, T0.cdfA, T0.cdfH, w1) && sumbits(key) < fs) fs = sumbits(key);
            }
        }
    }
    PBC_STAMP(kPcSKeys);
    // One exchange settles the ordinary chunk.  (The OTHER candidate counter is reset here: it was last read before the barrier
    // of this chunk's count; this chunk's was reset a chunk ago.)
    const int par = R.par++ & 1;
    if (wave == 0 && lane == 0) L.ncand[par ^ 1] = 0;
    // the survivors' exact costs; candidates (cost below the chunk-start best) go to the workgroup's list
    if (nsurv) {
        // (a wavefront holds one or two survivors, rarely more: eight at a time, eight lanes each -- a lane sums one byte of the
        //  discrepancy pattern in the canonical order (pbw_cost_exact), then the eight byte sums are added in order down the row)
        wave_fence();
        for (int b0 = 0; b0 < nsurv; b0 += 8) {
            const int sidx = b0 + (lane >> 3), byte = lane & 7;
            const bool has = sidx < nsurv;
            const u64 key = has ? L.keys[slist[has ? sidx : 0]] : 0ull;
            const u64 D = parity(pbw_tep((unsigned)key));
            const unsigned v = (unsigned)(D >> (8 * byte)) & 255u;
            float bs = 0.0f;
#pragma unroll
            for (int t = 0; t < 8; ++t) bs = ((v >> t) & 1u) ? bs + T0.w[64 + 8 * byte + t] : bs;
            // acc_b = acc_(b-1) + bs_b, acc_(-1) = the key's sum: seven dependent steps, lane b takes lane b - 1's value (row_shr:1)
            float acc = (byte == 0 ? __uint_as_float((unsigned)(key >> 32)) : 0.0f) + (byte == 0 ? bs : 0.0f);
#pragma unroll
            for (int st = 1; st < 8; ++st) {
                const float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x111, 0xF, 0xF, true));
                acc = byte == st ? prev + bs : acc;
            }
            const float c = has && byte == 7 ? acc : __builtin_inff();
            const bool cand = c < best0;
            const u64 cm = __ballot(cand);
            if (cm) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&L.ncand[par], __popcll(cm));
                base = __builtin_amdgcn_readfirstlane(base);
                const int idx = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
                if (cand && idx < kCoopMaxCand) { L.ck[idx] = key; L.cc[idx] = c; }
            }
        }
    }
    PBC_STAMP(kPcSSurv);
    // the exchange: frontier growth and the pops that shrink it (per wavefront |growth| <= 2 * WK <= 2^11), the smallest firing sum
    int negtot, deltot;
    unsigned sF;
    {
        const int slot = R.seq++ & 7;
        const int x1 = (wave_add_i32(neg) << 18) + (wave_add_i32(sumdel) + 4096), x2 = wave_min_i32((int)fs);
        if (lane == 0) { L.red[slot][wave] = x1; L.red2[slot][wave] = x2; }
        __syncthreads();
        int a = 0, m = 0x7FFFFFFF;
#pragma unroll
        for (int v = 0; v < NW; ++v) { a += L.red[slot][v]; const int x = L.red2[slot][v]; m = x < m ? x : m; }
        negtot = a >> 18; deltot = (a & 0x3FFFF) - NW * 4096; sF = (unsigned)m;
    }
    PBC_STAMP(kPcSBar2);
    const int ncand = L.ncand[par];
    if (S.nlive - negtot <= 1) return -1;
    if (ncand > kCoopMaxCand) return -1;
    if (ncand == 0) {
        if (sF == 0x7FFFFFFFu) {     // no candidate, no key fires: the whole chunk is visited and nothing else happens
            S.cmp += 2 * n; S.suc1 += n;
            S.j += n; S.nlive += deltot;
            return 0;
        }
        // no candidate, rule 1 fires: the search stops at the first key of the smallest firing sum (every key of that sum fires)
        int cs = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (k * 64 < kcount) cs += sumbits(key_at(k)) < sF;
        const int rank_stop = coop_sum<NW>(R, wave_add_i32(cs));
        PBC_STAMP(kPcSPos);
        S.cmp += 2 * (rank_stop + 1);
        S.suc1 += rank_stop;
        stop = 1; ntep = S.j + rank_stop + 1;
        return 1;
    }
    // ---- candidates: wavefront 0 orders them and finds the records; the others wait
    bool tie = false;
    if (wave == 0) {
        const u64 my = lane < ncand ? L.ck[lane] : ~0ull;
        const float myc = lane < ncand ? L.cc[lane] : 0.0f;
        int rk0 = 0;
        bool t0 = false;
        for (int d = 0; d < ncand; ++d) { const u64 o = L.ck[d]; rk0 += lane < ncand && o != my && visited_before(o, my); }
        wave_fence();
        if (lane < ncand) { L.ck[rk0] = my; L.cc[rk0] = myc; }
        wave_fence();
        int nrec = 0, stop2 = 0;
        if (!__ballot(t0)) {
            float before = best0;
            for (int t = 0; t < ncand && !stop2; ++t) {
                const u64 key = L.ck[t];
                const float c = L.cc[t];
                if (c < before) {
                    if (lane == 0) { L.rk[nrec] = key; L.rc[nrec] = c; }
                    ++nrec;
                    const float w1 = det_expf(P.c4 * __uint_as_float((unsigned)(key >> 32))) * Fr.spl;
                    if (pb_success_q(parity(pbw_tep((unsigned)key)), w1, T0.qpar, Fr)) stop2 = 1;
                    before = c;
                }
            }
        }
        if (lane == 0) { L.nrec = nrec; L.stop2 = stop2; L.tie0 = __ballot(t0) ? 1 : 0; }
    }
    __syncthreads();
    const int nrec = L.nrec, stop2 = L.stop2;
    tie |= L.tie0 != 0;
    // Rule 1 again for the keys from the first record on, each with the best it really sees (the record before it).  A lower
    // best fires sooner (beta falls with it), so a key that the LAST record's cost does not stop is stopped by none: probes
    // with that cost (as above) leave the keys beyond the last safe probe to evaluate -- usually none.  Keys before the first
    // record keep what the chunk-start best said (fs, if it lies before the first record).
    if (nrec > 0) {
        const unsigned s0 = sumbits(L.rk[0]);
        float r_safe2;
        {
            const float rp = lane == 63 ? mx : mn + (mx - mn) * ((float)(lane + 1) * (1.0f / 64.0f));
            float w1;
            const float bs = pb_promising_bs(rp, L.rc[nrec - 1], Fr, P.c4, T0.cdfA, T0.cdfH, w1);
            const u64 unsafe = ~__ballot((double)bs > Fr.p_t_pro * 1.001);
            const int u = unsafe ? __builtin_ctzll(unsafe) : 64;
            r_safe2 = u == 0 ? -1.0f : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rp), u - 1));
        }
        fs = fs < s0 ? fs : 0x7FFFFFFFu;
#pragma unroll 1
        for (int k = 0; k < PER; ++k) {
            if (k * 64 < kcount) {
                const u64 key = key_at(k);
                const bool behind = key != ~0ull && sumbits(key) >= s0;
                int t = 0;
                if (behind)
                    for (int u = 0; u < nrec; ++u) { const u64 r = L.rk[u]; t += r != key && visited_before(r, key); }
                const bool need = behind && __uint_as_float(sumbits(key)) > r_safe2;
                if (__ballot(need)) {
                    float w1;      // (the first record itself is judged with the chunk-start best: t = 0)
                    if (need && pb_not_promising(__uint_as_float(sumbits(key)), t > 0 ? L.rc[t - 1] : best0, Fr, P.c4, T0.cdfA, T0.cdfH, w1) && sumbits(key) < fs) fs = sumbits(key);
                }
            }
        }
    }
    PBC_STAMP(kPcSCand);
    sF = (unsigned)coop_min<NW>(R, wave_min_i32((int)fs));
    PBC_STAMP(kPcSMin);
    int reason = 0;
    unsigned sstop = 0;
    if (sF != 0x7FFFFFFFu) {
        reason = 1; sstop = sF;
        // ("the first key of the smallest firing sum stops" holds when every key of that sum sees the same best: not with a
        //  record among them)
        for (int u = 0; u < nrec; ++u) tie |= sumbits(L.rk[u]) == sF;
    }
    if (stop2) {
        const unsigned sR = sumbits(L.rk[nrec - 1]);
        if (reason == 0 || sR < sF) { reason = 2; sstop = sR; }
    }
    // the records that count: all of them without a stop and when the LAST record stops by rule 2 (the records are in visit order,
    // equal sums included -- counting "sums below the stopping record's" dropped a record that ties with it: found by
    // tests/tools/pb_long_fuzz.py on quantised channel values); those below the firing sum when rule 1 stops (a record ON that
    // sum has left the pass above)
    int nbefore = nrec;
    if (reason == 1) {
        nbefore = 0;
        for (int u = 0; u < nrec; ++u) nbefore += sumbits(L.rk[u]) < sstop;
    }
    // positions: the keys below the last record that counts, the keys below the stopping sum; ties (one exchange: 13 + 13 + 1 bits)
    const u64 bk = nbefore > 0 ? L.rk[nbefore - 1] : 0ull;
    const float bcost = nbefore > 0 ? L.rc[nbefore - 1] : 0.0f;
    int cb = 0, cs = 0;
#pragma unroll 1
    for (int k = 0; k < PER; ++k) {
        if (k * 64 < kcount) {
            const u64 key = key_at(k);
            if (nbefore > 0) cb += key != bk && visited_before(key, bk);
            if (reason == 1) cs += sumbits(key) < sstop;
        }
    }
    const int pos = coop_sum<NW>(R, wave_add_i32(cb) + (wave_add_i32(cs) << 13) + (__ballot(tie) ? 1 << 26 : 0));
    // (no barrier here: the candidate / record words are next written behind the NEXT chunk's count exchange, which every
    //  wavefront reaches only after it has left this function)
    PBC_STAMP(kPcSPos);
    if (pos >> 26) return -1;
    const int rank_best = pos & 0x1FFF, rank_stop = reason == 2 ? rank_best : (pos >> 13) & 0x1FFF;
    if (nbefore > 0) {
        const PbTep t = pbw_tep((unsigned)bk);
        u64 E = 1ull << t.p0;
        if (t.wt > 1) E |= 1ull << t.p1;
        if (t.wt > 2) E |= 1ull << t.p2;
        S.best = bcost; S.bestD = parity(t); S.bestE = E;
        S.bestidx = S.j + rank_best + 1;
    }
    S.suc2 += nbefore;
    if (reason) {
        S.cmp += 2 * (rank_stop + 1);
        S.suc1 += reason == 1 ? rank_stop : rank_stop + 1;
        stop = reason; ntep = S.j + rank_stop + 1;
        return 1;
    }
    S.cmp += 2 * n; S.suc1 += n;
    S.j += n; S.nlive += deltot;
    return 0;
}

// Wavefront 0 alone over the sums (lo, T] that hold n TEPs: the sorted path, in sub-chunks.  L.one.cur holds the cursors of
// the range's start (all 32 rows); arguments in L.su / L.bs / L.sP, results in L.bs / L.bstate / L.bstop / L.bntep
// (state 0 / 1 / 2 as in pb_wave_kernel).
template <int NW>
__device__ __forceinline__ void coop_solo_range(PbCoopLds<NW> &L)
{
    constexpr int CAP = kPbWaveCap;
    const int lane = threadIdx.x & 63;
    PbwState S = L.bs;
    int stop = L.bstop, ntep = L.bntep;
    PbParams P;
    P.order = L.sP.order; P.c4 = L.sP.c4;
    const int state = pbw_redo_call<CAP>(L.one, P, L.su.fr, L.su.d0, L.su.lo, L.su.T, L.su.smax, L.su.done, L.su.n, lane, S, stop, ntep);
    wave_fence();
    if (lane == 0) { L.bs = S; L.bstate = state; L.bstop = stop; L.bntep = ntep; }
}

template <int NW, bool PROF>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4))) void pb_coop_kernel(PbParams P, const double *__restrict__ cdf_half, int *__restrict__ ctl, int *__restrict__ listB,
                                                          const unsigned *__restrict__ carry, PbOut O, unsigned long long *__restrict__ prof_out)
{
    // (dynamic LDS: a hipGraph kernel node with more than 64 KiB of STATIC LDS aborts at replay on ROCm 7.2; the size is
    //  registered once in pb_ctx_init)
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_coop_lds[];
    PbcProf Q;
    PbCoopLds<NW> &L0 = *reinterpret_cast<PbCoopLds<NW> *>(pb_coop_lds);
    Q.pc = L0.prof;
    if constexpr (PROF) { if (threadIdx.x == 0) for (int k = 0; k < kPcSlots; ++k) Q.pc[k] = 0; Q.last = __builtin_amdgcn_s_memtime(); }
    PbCoopLds<NW> &L = *reinterpret_cast<PbCoopLds<NW> *>(pb_coop_lds);
    constexpr int NI = 32 / NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // list C: ticket t < nfar -> record t (the searches expected to run far), the others -> record kPbCoopHalf + (t - nfar)
    const int lenc = ctl[kPbCtlLenC], lenc2 = ctl[kPbCtlLenC2];
    const int nfar = lenc < kPbCoopHalf ? lenc : kPbCoopHalf;
    const int nlist = nfar + (lenc2 < kPbCoopHalf ? lenc2 : kPbCoopHalf);
    if ((int)blockIdx.x >= nlist) return;      // (more workgroups than frames -- or nothing handed on at all: leave before any set-up)
    const int nall = P.order > 2 ? kPbTabSize : (P.order > 1 ? kPbTriples0 : kPbPairs0);
    if (wave == 0) {
        L.one.cdfH[lane] = (float)cdf_half[lane];
        if (lane == 0) { L.one.cdfH[64] = (float)cdf_half[64]; L.ticket[0] = atomicAdd(&ctl[kPbCtlTicketC], 1); L.sP = P; }
    }
    CoopRed<NW> R{L.red, 0, wave, lane, 0};
    __syncthreads();
    for (int it = 0;; ++it) {
        // frames are handed out by ticket; the NEXT frame's ticket is drawn now and is back long before it is needed
        int tk = L.ticket[it & 1];
        if (tk >= nlist) break;
        int next_tk = 0;      // (drawn now, parked in LDS at the frame's end: the wavefront does not wait for the atomic here)
        if (tid == 0) { next_tk = atomicAdd(&ctl[kPbCtlTicketC], 1); L.ncand[0] = 0; L.ncand[1] = 0; }
        const unsigned *rec = carry + (long long)(tk < nfar ? tk : kPbCoopHalf + (tk - nfar)) * kPbRecWords;
        // the frame's tables: one load per thread (wavefront 0 may still be writing the previous frame's codeword out)
        if (tid >= 64)
            for (int i = tid - 64; i < kPbRecPrefix; i += 64 * NW - 64) reinterpret_cast<unsigned *>(&L.one)[i] = rec[i];
        const PbCarry &c = *reinterpret_cast<const PbCarry *>(rec + kPbRecScalars);
        unsigned po = wave == 0 ? rec[kPbRecPerm + (lane >> 2)] : 0u, po2 = wave == 0 ? rec[kPbRecPerm + 16 + (lane >> 2)] : 0u;      // (the permutation, for the codeword at the end: in flight from here)
        PbFrame Fr = c.fr;
        u64 d0 = c.d0;
        PbwState S;
        S.best = c.best; S.j = c.j; S.nlive = c.nlive; S.cmp = c.cmp; S.suc1 = c.suc1; S.suc2 = c.suc2; S.bestidx = c.bestidx;
        S.bestD = c.bestD; S.bestE = c.bestE;
        float lo = c.lo;
        int done = S.j;
        int a0[NI];            // the cursors the record holds
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            int q, l;
            coop_item<NW>(wave, j, lane, q, l);
            a0[j] = (int)((rec[kPbRecCur + (q >> 2) * 64 + l] >> (8 * (q & 3))) & 255u);
        }
        __syncthreads();
        CoopItems<NI> I;
        coop_items_init<NW>(I, L.one.w, P.order, lane, wave);
#pragma unroll
        for (int j = 0; j < NI; ++j) I.a[j] = a0[j];
        float smax = P.order > 2 ? (L.one.w[0] + L.one.w[1]) + L.one.w[2] : (P.order > 1 ? L.one.w[0] + L.one.w[1] : L.one.w[0]);
        int stop = 0, ntep = P.nmax, state = 0;
        float tprev = c.tprev, nprev = c.nprev;
        PBC_STAMP(kPcSetup);
        while (state == 0 && done < nall) {
            int aprev[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) aprev[j] = I.a[j];
            float T;
            int kbase = 0, kcount = 0;
            int n = coop_next_chunk<NW, PROF>(L, R, I, lo, done, nall, P.t3, lane, wave, T, tprev, nprev, kbase, kcount, Q);
            PBC_STAMP(kPcWalk);
            if (n < 0) { state = 2; break; }
            if (n == 0) break;
            PBC_COUNT(kPcChunks, 1); PBC_COUNT(kPcKeys, n);
            const float cmn = lo < 0.0f ? L.one.w[63] : lo, cmx = T < __builtin_inff() ? T : smax;
            wave_fence();
            state = coop_scan_chunk<NW, PROF>(L, R, P, Fr, d0, n, kbase, kcount, cmn, cmx, lane, wave, S, stop, ntep, Q);
            PBC_STAMP(kPcScan);
            if (state < 0) {
                // (One wavefront redoing ~2700 keys alone takes 35-135 us where the whole launch takes ~150, in-kernel stamps of
                //  round 4.  Measured and dropped: walking the same sums again with a third of the target, all wavefronts
                //  together, until <= 768 keys are left for wavefront 0 -- the failing chunk is walked ~8 times on the way down,
                //  the solo stamp falls from 630 k to 180 k cycles per launch and the launch gets LONGER: 0.479 -> 0.500 ms per
                //  search call at 2.5 dB, no change at 2.0 / 1.0 dB.)
                PBC_COUNT(kPcSolos, 1); PBC_COUNT(kPcSoloKeys, n);
                // wavefront 0 redoes the chunk alone from the cursors the chunk started with
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    int q, l;
                    coop_item<NW>(wave, j, lane, q, l);
                    // (the chunk kernel's code marks an item without members by a cursor of 0; here the cursor stops at base + 1)
                    reinterpret_cast<unsigned char *>(&L.one.cur[q >> 2][l])[q & 3] = (unsigned char)(aprev[j] > I.base[j] + 1 ? aprev[j] : 0);
                }
                __syncthreads();
                if (wave == 0) {
                    // (everything this wavefront needs afterwards is parked in LDS around the call: see PbCoopLds::sv)
#pragma unroll
                    for (int j = 0; j < NI; ++j) L.sv[j][lane] = I.a[j];
                    if (lane == 0) {
                        L.su.fr = Fr; L.su.d0 = d0; L.su.lo = lo; L.su.T = T; L.su.tprev = tprev; L.su.nprev = nprev; L.su.smax = smax;
                        L.su.done = done; L.su.n = n; L.su.seq = R.seq; L.su.par = R.par; L.su.it = it; L.su.tk = tk; L.su.next_tk = next_tk;
                        L.bs = S; L.bstop = stop; L.bntep = ntep;
                    }
                    wave_fence();
                    coop_solo_range<NW>(L);
                    wave_fence();
                    coop_items_init<NW>(I, L.one.w, P.order, lane, wave);
#pragma unroll
                    for (int j = 0; j < NI; ++j) I.a[j] = L.sv[j][lane];
                    Fr = L.su.fr; d0 = L.su.d0; lo = L.su.lo; T = L.su.T; tprev = L.su.tprev; nprev = L.su.nprev; smax = L.su.smax;
                    done = L.su.done; n = L.su.n; R.seq = L.su.seq; R.par = L.su.par; it = L.su.it; tk = L.su.tk;
                    rec = carry + (long long)(tk < nfar ? tk : kPbCoopHalf + (tk - nfar)) * kPbRecWords;
                    po = rec[kPbRecPerm + (lane >> 2)]; po2 = rec[kPbRecPerm + 16 + (lane >> 2)];
                    next_tk = L.su.next_tk;
                }
                __syncthreads();
                S = L.bs; state = L.bstate; stop = L.bstop; ntep = L.bntep;
                PBC_STAMP(kPcSolo);
            }
            lo = T;
            done += n;
        }
        if (tid == 0) L.ticket[(it + 1) & 1] = next_tk;
        __syncthreads();                        // (every wavefront is through with the frame's tables)
        if (wave == 0) {
            if (state == 2) {   // massive ties: the literal list replay decodes this frame
                if (lane == 0) listB[atomicAdd(&ctl[kPbCtlLenB], 1)] = (int)reinterpret_cast<const PbCarry *>(rec + kPbRecScalars)->f;
            } else {
                const PbCarry &c2 = *reinterpret_cast<const PbCarry *>(rec + kPbRecScalars);
                const u64 hm = c2.hm, hp = c2.hp;
                const long long f = c2.f;
                const int o1 = (int)((po >> (8 * (lane & 3))) & 255u), o2 = (int)((po2 >> (8 * (lane & 3))) & 255u);
                if (lane < 2) L.one.cw[lane] = 0;
                wave_fence();
                const u64 mrb_bits = hm ^ S.bestE, par_bits = S.bestD ^ hp;
                if ((mrb_bits >> lane) & 1) atomicOr(&L.one.cw[o1 >> 6], 1ull << (o1 & 63));
                if ((par_bits >> lane) & 1) atomicOr(&L.one.cw[o2 >> 6], 1ull << (o2 & 63));
                wave_fence();
                if (lane < 2) O.cw[f * 2 + lane] = L.one.cw[lane];
                if (lane == 0) {
                    if (O.metric) O.metric[f] = S.best;
                    if (O.best) O.best[f] = S.bestidx;
                    if (O.ntep) O.ntep[f] = ntep;
                    if (O.aux) { O.aux[f * 4] = S.cmp; O.aux[f * 4 + 1] = S.suc1; O.aux[f * 4 + 2] = S.suc2; O.aux[f * 4 + 3] = stop; }
                }
            }
        }
        PBC_STAMP(kPcOut);
        PBC_COUNT(kPcFrames, 1);
    }
    if constexpr (PROF) { if (tid == 0 && Q.pc[kPcFrames]) for (int k = 0; k < kPcSlots; ++k) atomicAdd(&prof_out[k], Q.pc[k]); }
}

// ---------------------------------------------------------------------------------------
// stage 3: literal replay of the frontier list, one frame of list B per wavefront (frames whose sums tie
// massively, or every frame when the caller asks for this path as a cross-check)
// ---------------------------------------------------------------------------------------
// (recs: the singles kernel's records, when the front end ran inside it and left nothing in a workspace: a frame of list B is
//  then set up from its record -- |y'|, P', the permutation, the hard decisions)
__global__ __launch_bounds__(256) void pb_seq_kernel(const float *__restrict__ y, const int *__restrict__ index,
                                                     const unsigned char *__restrict__ perm_in,
                                                     const u64 *__restrict__ parity_in, const unsigned *__restrict__ recs, PbParams P,
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
        SearchFrame S;
        if (recs) {
            const unsigned *const R = recs + f * kPbR1Words;
            const PbHead &H = *reinterpret_cast<const PbHead *>(R + kPbR1Head);
            const unsigned char *const pb = reinterpret_cast<const unsigned char *>(R + kPbR1Perm);
            S.o1 = pb[lane]; S.o2 = pb[64 + lane];
            L.perm[lane] = (unsigned char)S.o1; L.perm[lane + 64] = (unsigned char)S.o2;
            L.w[lane] = __uint_as_float(R[lane]); L.w[lane + 64] = __uint_as_float(R[64 + lane]);
            L.P[lane] = reinterpret_cast<const u64 *>(R + 128)[lane];
            if (lane < 2) L.cw[lane] = 0;
            S.hm = H.hm; S.hp = H.hp; S.d0 = H.d0;
            wave_fence();
            build_byte_luts<8>(L.lut, &L.w[64], lane);
            wave_fence();
        } else {
            const long long src = index ? index[f] : f;
            S = search_prepare(L, y, src, perm_in, parity_in, f, lane);
        }
        const PbFrame Fr = pb_frame_setup(L.w, B.q, B.cdfA, P.c4, P.order, P.nmax, lane);
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
                if (p_suc > (float)p_t_suc) { stop = 2; ntep = j + 1; break; }
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

// chunk targets: the first chunk's count is only guessed (+-40 %), the others follow the growth of the counts
ldpc_pb_tuning pb_default_tuning()
{
    ldpc_pb_tuning t;
    t.budget = 4096; t.budget_s = t.budget / 8; t.budget_m = t.budget / 4; t.budget_l = 2 * t.budget; t.budget_xl = 6 * t.budget;
    t.t1 = 320; t.t2 = 600; t.t3 = 3072;        // (t2 measured at 1.0 / 2.5 dB: 512: 4.16 / 0.548 ms, 600: 4.08 / 0.532, 676: 4.11 / 0.554, 760: 4.37 / 0.561)
    // The tail rule.  One search call at a time (means over four batches x 12 launches, ms at 2.5 / 2.0 / 1.0 dB): off 0.484 / 1.020 / 4.159,
    // 10 %: 0.451 / 0.967 / 3.852, 20 %: 0.431 / 0.948 / 3.790, 30 %: 0.424 / 0.929 / 3.775, 40 %: 0.436 / 0.926 / 3.753 (budget / 16; / 8 and / 4
    // within 1 %).  Four batches in flight (bench.py's graph, 10^8 frames/s at 2.5 / 2.0 dB, 10^7 at 1.0 dB), where the tails hide behind
    // the other batches and the total work counts: off 2.715 / 1.335 / 3.30, 15 %: 2.708 / 1.324 / 3.357, 25 %: 2.670 / 1.311 / 3.343,
    // 40 %: 2.626 / - / 3.31.  20 %: a tenth off a lone call, one per cent off the throughput at 2.5 dB, one per cent on it at 1.0 dB.
    t.late_min = 4608; t.late_maxlen = 1 << 30; t.late_pct = 20; t.late_div = 16;
    t.handoff_maxlen = 1 << 30;
    return t;
}

int pb_ctx_init(ldpc_ctx *ctx)
{
    state(ctx)->pb_tuning = pb_default_tuning();
    static_assert(sizeof(PbCoopLds<kPbCoopW>) * (16 / kPbCoopW) <= 160 * 1024, "16 wavefronts of this kernel per CU");
    LDPC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_coop_kernel<kPbCoopW, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(PbCoopLds<kPbCoopW>)));
    LDPC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pb_coop_kernel<kPbCoopW, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)sizeof(PbCoopLds<kPbCoopW>)));
    return LDPC_OK;
}

// list replay: the list is append-only, at most 1 + 2 (N_max - 1) slots; spilled slots first, spilled chunk minima after
static int pb_spill_layout(ldpc_ctx *ctx, int order, int64_t *spill_slots, int64_t *stride)
{
    const int64_t nmax = state(ctx)->ntep[order];
    const int64_t slots = 2 * nmax + 2;
    if (slots > (int64_t)kPbSuper * 4096) return fail(LDPC_E_UNSUPPORTED, "ldpc_osd_decode: PB-OSD list of %lld slots exceeds the kernel's limit", (long long)slots);
    *spill_slots = slots > kPbLdsSlots ? slots - kPbLdsSlots : 0;
    *stride = *spill_slots + (slots / 64 + 2) + 2;
    return LDPC_OK;
}

// PB-OSD part of a stream's workspace: control words, the two frame lists, the list replay's spill areas
static int stream_ws_pb(ldpc_ctx *ctx, hipStream_t s, int64_t frames, int64_t spill_stride, StreamWs **out)
{
    OsdState *st = state(ctx);
    std::lock_guard<std::mutex> lock(st->mu);
    StreamWs &w = st->ws[s];
    const bool grow_list = frames > w.pb_cap || !w.d_pb_ctl, grow_spill = spill_stride > w.pb_spill_stride;
    const bool capturing = stream_capturing(s);
    if ((grow_list || grow_spill) && capturing)
        return fail(LDPC_E_NOMEM, "PB-OSD workspace of this stream must be sized before capturing (ldpc_osd_reserve_stream with the "
                    "PB-OSD parameters, or one eager call on the stream)");
    if ((grow_list || grow_spill) && w.captured)
        return fail(LDPC_E_NOMEM, "PB-OSD workspace of this stream is referenced by a captured graph and cannot grow: destroy the graph and "
                    "call ldpc_osd_release_stream, or reserve the larger size before capturing");
    if (capturing) w.captured = true;
    if (grow_list) {
        (void)hipFree(w.d_pb_list); w.d_pb_list = nullptr; w.pb_cap = 0;
        if (!w.d_pb_ctl && hipMalloc((void **)&w.d_pb_ctl, sizeof(int) * kPbCtlInts) != hipSuccess)
            return fail(LDPC_E_NOMEM, "PB-OSD control words could not be allocated");
        (void)hipFree(w.d_pb_prep); w.d_pb_prep = nullptr;
        (void)hipFree(w.d_pb_carry); w.d_pb_carry = nullptr;
        const int64_t sub_cap = (frames + kPbSub - 1) / kPbSub;
        if (hipMalloc((void **)&w.d_pb_list, sizeof(int) * (2 * (size_t)kPbSub * (size_t)sub_cap)) != hipSuccess ||
            hipMalloc(&w.d_pb_carry, sizeof(unsigned) * kPbRecWords * (size_t)kPbHeavyCap) != hipSuccess ||
            hipMalloc(&w.d_pb_prep, sizeof(unsigned) * kPbR1Words * (size_t)frames) != hipSuccess)
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

// ldpc_osd_reserve_stream for algo = PB: everything launch_pb would allocate for `frames` frames of this order
int pb_reserve(ldpc_ctx *ctx, hipStream_t s, int64_t frames, int order)
{
    int64_t spill_slots, stride;
    if (int rc = pb_spill_layout(ctx, order, &spill_slots, &stride)) return rc;
    StreamWs *w;
    return stream_ws_pb(ctx, s, frames, stride, &w);
}

int launch_pb(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
              const unsigned char *d_perm, const u64 *d_parity, const ldpc_osd_params *p, uint64_t *d_cw, float *d_metric,
              int32_t *d_best, int32_t *d_ntep, hipStream_t s)
{
    OsdState *st = state(ctx);
    const int64_t nmax = st->ntep[p->order];
    int64_t spill_slots, stride;
    int rc = pb_spill_layout(ctx, p->order, &spill_slots, &stride);
    if (rc) return rc;
    StreamWs *w;
    if ((rc = stream_ws_pb(ctx, s, F, stride, &w))) return rc;
    PbParams pp;
    pp.order = p->order; pp.nmax = (int)nmax; pp.cmin_off = spill_slots;
    // When a search leaves the chunk kernel for the workgroup kernel: after `budget` TEPs, the budget chosen ON THE DEVICE from the
    // length of the frame's sub-list of list A (a sixteenth of the frames that search beyond the weight-1 head), so that the
    // workgroup kernel gets the tails, 1000-3500 frames a call, and never the bulk (its list holds 2 x kPbCoopHalf frames).
    // Measured per 131 072-frame step (round 3), PB kernels with the budget of the schedule / the neighbouring ones / no hand-over:
    //   3.5 dB (length 42)    512: 0.20 ms              2.0 dB  (1750)   8192: 1.50 | 4096: 1.52 | 16384: 1.61 | none 1.80
    //   3.0 dB (177)         1024: 0.34                 1.75 dB (2560)   8192: 2.20 | 16384: 2.26 | 4096: 2.91 | none 2.50
    //   2.5 dB (626)         4096: 0.70 | 2048: 0.70    1.5 dB  (3475)  16384: 3.22 | 8192: 3.78 | none 3.41
    //   2.25 dB (1090)       4096: 1.02 | none 1.45     1.0 dB  (5300)  16384: 6.26 | 8192: 6.85 | none 6.44
    // (serial kernel sums; with four batches in flight -- bench.py's graph -- 24576 beats 16384 and none at 1.0 dB: 2.15 / 2.12 / 2.14 x 10^7
    //  frames/s, and ties with 16384 at 1.5 dB: 4.25 x 10^7 against 4.18 without)
    // The schedule is a property of the CONTEXT (ldpc_ctx_set_pb_tuning, validated there; defaults: pb_default_tuning) and is
    // copied here under the state's lock: no environment variable is read on the decode path (rounds 1-3 read thirteen).
    ldpc_pb_tuning tn;
    {
        std::lock_guard<std::mutex> lock(st->mu);
        tn = st->pb_tuning;
    }
    pp.budget = tn.budget; pp.budget_s = tn.budget_s; pp.budget_m = tn.budget_m; pp.budget_l = tn.budget_l; pp.budget_xl = tn.budget_xl;
    pp.t1 = tn.t1; pp.t2 = tn.t2; pp.t3 = tn.t3;
    pp.late_min = tn.late_min; pp.late_maxlen = tn.late_maxlen; pp.late_pct = tn.late_pct; pp.late_div = tn.late_div;
    pp.handoff_maxlen = tn.handoff_maxlen;
    pp.c4 = (float)(-4.0 * (1.0 / pow(10.0, (double)p->snr_db / 10.0)));    // -4 * noise_variance, pb_testing.py:50-52
    const int mode = (p->reserved & 4) ? 2 : ((p->reserved & 2) ? 1 : 0);
    PbOut O{reinterpret_cast<u64 *>(d_cw), d_metric, d_best, d_ntep, reinterpret_cast<int *>(p->d_aux)};
    const int64_t list_len = (int64_t)kPbSub * w->pb_sub_cap;   // >= pb_cap
    int *listA = w->d_pb_list, *listB = w->d_pb_list + list_len;
    unsigned *carry = reinterpret_cast<unsigned *>(w->d_pb_carry);      // [kPbHeavyCap] records of kPbRecWords words
    const int sub_cap = (int)w->pb_sub_cap;
    unsigned *recs = reinterpret_cast<unsigned *>(w->d_pb_prep);      // [pb_cap] records of kPbR1Words words (singles -> chunk kernel)
    hipLaunchKernelGGL(pb_ctl_clear_kernel, dim3(1), dim3(64), 0, s, w->d_pb_ctl);
    const int64_t want = (F + 3) / 4;
    const unsigned g1 = (unsigned)(F < 1 ? 1 : (F < 32768 ? F : 32768));
    static const bool profile_s = getenv("LDPC_PB_PROFILE") != nullptr;
    // d_perm == nullptr (ldpc_osd_decode's route): the front end runs inside the singles kernel, nothing goes through a workspace
    // -- the frames of list B are then set up from the singles records (mode 2, every frame to the list replay, writes none:
    // the caller keeps the two-kernel route for it)
    const bool fused_front = d_perm == nullptr;
    if (fused_front && (mode == 2 || d_parity != nullptr))
        return fail(LDPC_E_ARG, "PB-OSD: the fused front end needs both front-end buffers absent and is not the list-replay route");
    const u64 *const Gcols = reinterpret_cast<const u64 *>(ctx->d_Gcols);
    if (fused_front) {
        hipLaunchKernelGGL((pb_singles_kernel<false, true>), dim3(g1), dim3(64), 0, s, d_y, d_index, d_count, (long long)F, d_perm, d_parity, Gcols, pp, mode,
                           st->d_cdf_half, w->d_pb_ctl, listA, listB, sub_cap, recs, O, (unsigned long long *)nullptr);
    } else if (!profile_s) {
        hipLaunchKernelGGL((pb_singles_kernel<false, false>), dim3(g1), dim3(64), 0, s, d_y, d_index, d_count, (long long)F, d_perm, d_parity, Gcols, pp, mode,
                           st->d_cdf_half, w->d_pb_ctl, listA, listB, sub_cap, recs, O, (unsigned long long *)nullptr);
    } else {
        static unsigned long long *d_ps = nullptr;
        if (!d_ps) LDPC_HIP(hipMalloc((void **)&d_ps, sizeof(unsigned long long) * 8));
        LDPC_HIP(hipMemsetAsync(d_ps, 0, sizeof(unsigned long long) * 8, s));
        hipLaunchKernelGGL((pb_singles_kernel<true, false>), dim3(g1), dim3(64), 0, s, d_y, d_index, d_count, (long long)F, d_perm, d_parity, Gcols, pp, mode,
                           st->d_cdf_half, w->d_pb_ctl, listA, listB, sub_cap, recs, O, d_ps);
        unsigned long long h[8];
        LDPC_HIP(hipMemcpyAsync(h, d_ps, sizeof(h), hipMemcpyDeviceToHost, s));
        LDPC_HIP(hipStreamSynchronize(s));
        fprintf(stderr, "[LDPC_PB_PROFILE] singles kernel, cycles of lane 0 summed over wavefronts: start+perm/P' loads=%llu y loads+LUT=%llu frame_setup=%llu rules=%llu hand_on=%llu write=%llu\n",
                h[5], h[0], h[1], h[2], h[3], h[4]);
    }
    // chunk kernel: one workgroup (= one wavefront) per (sub-list, entry); a multiple of 16 workgroups, at most 65 536 (a
    // workgroup then takes every 4096th entry of its sub-list).  Workgroups beyond their sub-list's length leave at once.
    // (three instantiations by what the context's probe of the wave_rol:1 DPP control found -- the walk rotates a copy of the
    //  weights through the wavefront: -1 = a lane receives its upper neighbour's value, +1 = its lower neighbour's, 0 = not a
    //  rotation: LDS reads instead)
    const int rot = ctx->dpp_wave_rol_dir;
#define PB_WAVE_LAUNCH(PROFILED, prof_ptr)                                                                                           \
    do {                                                                                                                             \
        if (rot < 0) hipLaunchKernelGGL((pb_wave_kernel<kPbWaveCap, PROFILED, -1>), dim3(g2), dim3(64), 0, s, pp, st->d_cdf_half, w->d_pb_ctl, \
                                        listA, listB, sub_cap, carry, recs, O, prof_ptr);                                            \
        else if (rot > 0) hipLaunchKernelGGL((pb_wave_kernel<kPbWaveCap, PROFILED, 1>), dim3(g2), dim3(64), 0, s, pp, st->d_cdf_half, w->d_pb_ctl, \
                                             listA, listB, sub_cap, carry, recs, O, prof_ptr);                                       \
        else hipLaunchKernelGGL((pb_wave_kernel<kPbWaveCap, PROFILED, 0>), dim3(g2), dim3(64), 0, s, pp, st->d_cdf_half, w->d_pb_ctl, \
                                listA, listB, sub_cap, carry, recs, O, prof_ptr);                                                    \
    } while (0)
    // chunk kernel: one workgroup (= one wavefront) per (sub-list, entry); a multiple of 16 workgroups, at most 65 536 (a
    // workgroup then takes every 4096th entry of its sub-list).  Workgroups beyond their sub-list's length leave at once.
    const int64_t g2w = ((F + kPbSub - 1) / kPbSub) * kPbSub;
    const unsigned g2 = (unsigned)(g2w < 65536 ? g2w : 65536);
    if (!profile_s) {
        PB_WAVE_LAUNCH(false, (unsigned long long *)nullptr);
    } else {
        static unsigned long long *d_pw = nullptr;
        if (!d_pw) LDPC_HIP(hipMalloc((void **)&d_pw, sizeof(unsigned long long) * kPwSlots));
        LDPC_HIP(hipMemsetAsync(d_pw, 0, sizeof(unsigned long long) * kPwSlots, s));
        PB_WAVE_LAUNCH(true, d_pw);
        unsigned long long h[kPwSlots];
        LDPC_HIP(hipMemcpyAsync(h, d_pw, sizeof(h), hipMemcpyDeviceToHost, s));
        LDPC_HIP(hipStreamSynchronize(s));
        static const char *names[kPwSlots] = {"setup", "walk", "sort", "tie", "eval", "rules", "combine", "finish", "FRAMES", "CHUNKS", "WALKS", "KEYS", "sweepA", "sweepB", "dense", "ROUNDS", "TRIPS", "scan", "SORTEDCHUNKS", "load1", "load2", "store"};
        fprintf(stderr, "[LDPC_PB_PROFILE] chunk kernel, shader-clock ticks summed over wavefronts:");
        for (int q = 0; q < kPwSlots; ++q) fprintf(stderr, " %s=%llu", names[q], h[q]);
        fprintf(stderr, "\n");
    }
#undef PB_WAVE_LAUNCH
    // the long searches the chunk kernel handed on (at most kPbHeavyCap, in two halves; the workgroups find an empty list otherwise)
    const unsigned g4 = (unsigned)(F < kPbCoopGrid ? F : kPbCoopGrid);
    if (!profile_s) {
        hipLaunchKernelGGL((pb_coop_kernel<kPbCoopW, false>), dim3(g4), dim3(64 * kPbCoopW), sizeof(PbCoopLds<kPbCoopW>), s,
                           pp, st->d_cdf_half, w->d_pb_ctl, listB, carry, O, (unsigned long long *)nullptr);
    } else {
        static unsigned long long *d_pc = nullptr;
        if (!d_pc) LDPC_HIP(hipMalloc((void **)&d_pc, sizeof(unsigned long long) * kPcSlots));
        LDPC_HIP(hipMemsetAsync(d_pc, 0, sizeof(unsigned long long) * kPcSlots, s));
        hipLaunchKernelGGL((pb_coop_kernel<kPbCoopW, true>), dim3(g4), dim3(64 * kPbCoopW), sizeof(PbCoopLds<kPbCoopW>), s,
                           pp, st->d_cdf_half, w->d_pb_ctl, listB, carry, O, d_pc);
        unsigned long long h[kPcSlots];
        LDPC_HIP(hipMemcpyAsync(h, d_pc, sizeof(h), hipMemcpyDeviceToHost, s));
        LDPC_HIP(hipStreamSynchronize(s));
        static const char *names[kPcSlots] = {"setup", "walk", "scan", "solo", "out", "FRAMES", "CHUNKS", "SOLOS", "SOLOKEYS", "KEYS", "COUNTS", "-", "-", "count", "generate", "count_exchange", "bound", "s_probe", "s_keys", "-", "s_survivors", "s_exchange", "s_candidates", "s_min", "s_positions"};
        fprintf(stderr, "[LDPC_PB_PROFILE] workgroup kernel, shader-clock ticks of thread 0 summed over workgroups:");
        for (int q = 0; q < kPcSlots; ++q) fprintf(stderr, " %s=%llu", names[q], h[q]);
        fprintf(stderr, "\n");
    }
    const unsigned g3 = (unsigned)(want < kPbSeqBlocks ? (want < 1 ? 1 : want) : kPbSeqBlocks);
    hipLaunchKernelGGL(pb_seq_kernel, dim3(g3), dim3(256), 0, s, d_y, d_index, d_perm, d_parity, fused_front ? recs : (const unsigned *)nullptr, pp, st->d_cdf_half,
                       reinterpret_cast<PbEntry *>(w->d_pb_spill), (long long)w->pb_spill_stride, w->d_pb_ctl, listB, O);
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

}  // namespace ldpc

extern "C" {

int ldpc_ctx_get_pb_tuning(ldpc_ctx *ctx, ldpc_pb_tuning *out)
{
    using namespace ldpc;
    if (!ctx || !out || !ctx->osd_state) return fail(LDPC_E_ARG, "ldpc_ctx_get_pb_tuning: null argument");
    OsdState *st = state(ctx);
    std::lock_guard<std::mutex> lock(st->mu);
    *out = st->pb_tuning;
    return LDPC_OK;
}

int ldpc_ctx_set_pb_tuning(ldpc_ctx *ctx, const ldpc_pb_tuning *t)
{
    using namespace ldpc;
    if (!ctx || !ctx->osd_state) return fail(LDPC_E_ARG, "ldpc_ctx_set_pb_tuning: null context");
    ldpc_pb_tuning v = t ? *t : pb_default_tuning();
    const int budgets[5] = {v.budget_s, v.budget_m, v.budget, v.budget_l, v.budget_xl};
    for (int b : budgets)
        if (b < 1) return fail(LDPC_E_ARG, "ldpc_ctx_set_pb_tuning: hand-over budgets must be >= 1 TEP (got %d)", b);
    constexpr int kcap = PbwCaps<kPbWaveCap, true>::KCAP;
    if (v.t1 < 32 || v.t1 > kcap || v.t2 < 32 || v.t2 > kcap)
        return fail(LDPC_E_ARG, "ldpc_ctx_set_pb_tuning: chunk targets t1 / t2 must lie in [32, %d] (got %d, %d)", kcap, v.t1, v.t2);
    if (v.t3 < 256 || v.t3 > kCoopCap) return fail(LDPC_E_ARG, "ldpc_ctx_set_pb_tuning: t3 must lie in [256, %d] (got %d)", kCoopCap, v.t3);
    if (v.late_div < 1 || v.late_pct < 0 || v.late_min < 0 || v.late_maxlen < 0 || v.handoff_maxlen < 0)
        return fail(LDPC_E_ARG, "ldpc_ctx_set_pb_tuning: late_div must be >= 1 and the other fields >= 0");
    OsdState *st = state(ctx);
    std::lock_guard<std::mutex> lock(st->mu);
    st->pb_tuning = v;
    return LDPC_OK;
}

}  // extern "C"
