// H-form ordered-statistics primitives for the DL-OSD stage, (128,64) codes on gfx950 (MI355X).
//
// Reference (paths relative to LDPC_128/DL_OSD_Testing_serial/ of the reference):
//   mag_input_gen / check_matrix_reorder   ordered_statistics_decoding.py:25-41
//   identify_mrb / full_gf2elim            ordered_statistics_decoding.py:43-80, 222-257
//   acquire_min / sliding_osd              ordered_statistics_decoding.py:153-186
//
// The dual of ldpc_osd.hip: positions are sorted by ASCENDING reliability, the elimination runs
// on the permuted H (-> [I | M], the identity part on the least reliable independent positions),
// a candidate is [M . mrb, mrb], and the scan is organised in caller-defined TEP blocks whose
// minima feed the reference's sliding-window network on the host.  Two separate LLR inputs: one
// orders the positions and supplies the starting MRB hard decisions (the CNN-refined values in
// the reference), the other (trajectory row 0) supplies the metric.  One frame per wavefront.
#include "ldpc_internal.h"
#include "ldpc_wave.h"

namespace ldpc {

struct __attribute__((aligned(16))) HFrontLds {
    RankLds rank;              // reliability sort (bucket_ranks, ldpc_wave.h)
    u64 colbuf[64];            // M columns in updated MRB order, bit = physical row
    unsigned mask[4];          // which sorted positions ended up in the MRB
    unsigned char lri[128];    // sorted position -> original bit
};

__global__ __launch_bounds__(256) void hosd_front_kernel(const float *__restrict__ x, long long F,
                                                         const u64 *__restrict__ Hcols,
                                                         unsigned char *__restrict__ lri_out,
                                                         unsigned char *__restrict__ uidx_out, u64 *__restrict__ M_out,
                                                         int *__restrict__ nswaps)
{
    __shared__ HFrontLds lds[4];
    const int lane = threadIdx.x & 63;
    HFrontLds &L = lds[threadIdx.x >> 6];
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);

    for (long long f = wave; f < F; f += (long long)gridDim.x * 4) {
        // ---- mag_input_gen (:25-28): rank in ascending |x|, ties -> lower index ------------
        // ascending order of (|x| bits, index) = descending order of the complemented key, buckets mirrored
        const unsigned a1 = __float_as_uint(x[f * 128 + lane]) & 0x7FFFFFFFu, a2 = __float_as_uint(x[f * 128 + 64 + lane]) & 0x7FFFFFFFu;
        const float bs = bucket_scale(a1, a2);
        int r1, r2;
        bucket_ranks(L.rank, ~(((u64)a1 << 32) | (unsigned)lane), ~(((u64)a2 << 32) | (unsigned)(lane + 64)), 63 - bucket_of(a1, bs),
                     63 - bucket_of(a2, bs), lane, r1, r2);
        L.lri[r1] = (unsigned char)lane;
        L.lri[r2] = (unsigned char)(lane + 64);
        if (lane < 4) L.mask[lane] = 0;
        wave_fence();
        // ---- H with columns in sorted order (:38-40), column-major, then full_gf2elim (:222-257)
        const int s1 = L.lri[lane], s2 = L.lri[lane + 64];
        u64 C1 = Hcols[s1];
        u64 C2 = Hcols[s2];
        int rho = lane, idx1 = lane, idx2 = lane + 64;
        const int ns = ge_columns(C1, C2, rho, idx1, idx2, lane, nullptr);
        // ---- identify_mrb (:58-69): LRB = index_order[:64] as left by the elimination, MRB sorted
        atomicOr(&L.mask[idx2 >> 5], 1u << (idx2 & 31));
        wave_fence();
        const unsigned m[4] = {L.mask[0], L.mask[1], L.mask[2], L.mask[3]};
        const int rankM = below_mask(m, idx2);          // updated MRB position of column 64 + lane
        L.colbuf[rankM] = C2;
        wave_fence();
        // rows of updated_M: transpose to (lane = physical row, bit = MRB position), then logical row
        // r = the row whose pivot sits in column r = physical row rho[r]
        const u64 R = transpose64(L.colbuf[lane], lane);
        M_out[f * 64 + lane] = shfl64(R, rho);
        lri_out[f * 128 + lane] = (unsigned char)s1;
        lri_out[f * 128 + 64 + lane] = (unsigned char)s2;
        uidx_out[f * 128 + lane] = (unsigned char)idx1;
        uidx_out[f * 128 + 64 + rankM] = (unsigned char)idx2;
        if (nswaps && lane == 0) nswaps[f] = ns;
        wave_fence();
    }
}

constexpr int kHosdMaxBlocks = 1024;   // keys kept in LDS (8 KiB); the reference's paths have 30 blocks
constexpr int kHosdLdsTeps = 2304;     // a TEP table up to this size (256 threads x 9) of weight <= 2 in <= kHosdRunBlocks blocks is kept in LDS,
constexpr int kHosdRunBlocks = 128;    // two bytes per TEP, and scanned in per-thread runs (the reference's order-2 paths: 2081 TEPs, 28 blocks)
constexpr int kHosdChunk = 256;        // larger tables: TEPs per work item (a block larger than this is split)

// RUNS: the per-thread-run form (a small table: 23 KiB of LDS, 6 workgroups per CU -- the kernel answers to occupancy: 0.34 ms at
// 4 workgroups per CU with 4-byte TEPs and 1024 block keys, 0.31 at 5); otherwise rounds 1-2's work items (25.7 KiB).
template <bool RUNS>
struct __attribute__((aligned(16))) HSearchLds {
    float lut[16][256];   // lut[b][v]: partial metric of discrepancy byte b (updated positions 8b..8b+7)
    u64 Mcol[64];         // column j of updated_M (bit r = M[r][j])
    float w[128];         // |metric_llr| in updated order
    u64 keys[RUNS ? kHosdRunBlocks : kHosdMaxBlocks];   // per block: (metric bits << 32) | TEP index, minimum = first minimum
    u64 hgL, hgM, mrb0, DL0, best, cw[2];
    int ticket, heavy;
    unsigned char o[128]; // original bit index of updated position p
    int boff[RUNS ? kHosdRunBlocks + 1 : 1];            // block_off, once per workgroup
    unsigned short tl[RUNS ? kHosdLdsTeps : 2];         // the TEP table, once per workgroup: x | y << 6 | weight << 12
};

template <class LDS>
__device__ __forceinline__ float hosd_cost(const LDS &L, u64 DL, u64 DM)
{
    float acc = lut_byte<0>(L.lut, DL);
    acc = acc + lut_byte<1>(L.lut, DL); acc = acc + lut_byte<2>(L.lut, DL); acc = acc + lut_byte<3>(L.lut, DL);
    acc = acc + lut_byte<4>(L.lut, DL); acc = acc + lut_byte<5>(L.lut, DL); acc = acc + lut_byte<6>(L.lut, DL);
    acc = acc + lut_byte<7>(L.lut, DL);
    acc = acc + lut_byte<0>(&L.lut[8], DM); acc = acc + lut_byte<1>(&L.lut[8], DM); acc = acc + lut_byte<2>(&L.lut[8], DM);
    acc = acc + lut_byte<3>(&L.lut[8], DM); acc = acc + lut_byte<4>(&L.lut[8], DM); acc = acc + lut_byte<5>(&L.lut[8], DM);
    acc = acc + lut_byte<6>(&L.lut[8], DM); acc = acc + lut_byte<7>(&L.lut[8], DM);
    return acc;
}

template <class LDS>
__device__ __forceinline__ void hosd_apply(const LDS &L, uchar4 s, u64 &DL, u64 &DM)
{
    if (s.w > 0) { DL ^= L.Mcol[s.x]; DM ^= 1ull << s.x; }
    if (s.w > 1) { DL ^= L.Mcol[s.y]; DM ^= 1ull << s.y; }
    if (s.w > 2) { DL ^= L.Mcol[s.z]; DM ^= 1ull << s.z; }
}

// One frame per 256-thread workgroup: the four wavefronts share the frame's LUTs.  Every thread scans ITS OWN run of
// consecutive TEPs (ceil(N / 256), made odd: LDS banks) and keeps a running first minimum that it flushes -- a 64-bit LDS
// atomicMin on the block's (metric bits, TEP index) key -- when its run crosses into the next block and at its end: ~1.2
// atomics per thread and frame, no per-block wave reductions (round 1-2: work items pulled from an LDS ticket, a wave
// arg-min and an atomic per item -- 40 % of the kernel's instructions for the reference's 28 blocks of ~75 TEPs).
// Metrics are sums of magnitudes, so their bit patterns order like the floats and the smallest key is the FIRST minimum.
template <bool RUNS>
__global__ __launch_bounds__(256) void hosd_search_kernel(const float *__restrict__ xo, const float *__restrict__ xm,
                                                          long long F, const unsigned char *__restrict__ lri,
                                                          const unsigned char *__restrict__ uidx,
                                                          const u64 *__restrict__ Mrows, const uchar4 *__restrict__ teps,
                                                          const int *__restrict__ block_off, int nblk,
                                                          const u64 *__restrict__ label, float *__restrict__ block_min,
                                                          int *__restrict__ block_arg, float *__restrict__ truth,
                                                          u64 *__restrict__ cw_out, float *__restrict__ metric_out,
                                                          int *__restrict__ best_out)
{
    __shared__ HSearchLds<RUNS> L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Both instantiations are launched and decide alike from device data which of them works (the other returns here): the
    // per-thread-run form takes a table of <= kHosdLdsTeps TEPs of weight <= 2 in <= kHosdRunBlocks blocks.
    int run0 = 0, run1 = 0, b0 = 0;
    const int ntep = nblk > 0 ? block_off[nblk] : 0;
    bool runs = ntep <= kHosdLdsTeps && nblk <= kHosdRunBlocks;
    if (runs) {         // (uniform) any TEP of weight 3?
        if (tid == 0) L.heavy = 0;
        __syncthreads();
        bool h = false;
        for (int t = tid; t < ntep; t += 256) h |= teps[t].w > 2;
        if (h) L.heavy = 1;
        __syncthreads();
        runs = L.heavy == 0;
    }
    if (runs != RUNS) return;
    if constexpr (RUNS) {
        // once per workgroup: the block offsets and the table in LDS, this thread's run and its first block
        for (int b = tid; b <= nblk; b += 256) L.boff[b] = block_off[b];
        for (int t = tid; t < ntep; t += 256) { const uchar4 s = teps[t]; L.tl[t] = (unsigned short)(s.x | (s.y << 6) | (s.w << 12)); }
        __syncthreads();
        const int per = ((ntep + 255) / 256) | 1;      // (odd: a lane's run starts on its own LDS bank)
        run0 = tid * per < ntep ? tid * per : ntep; run1 = run0 + per < ntep ? run0 + per : ntep;
        int lo = 0, hi = nblk;       // the block that holds TEP run0: the last b with boff[b] <= run0 (blocks may be empty)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (L.boff[mid] <= run0) lo = mid; else hi = mid; }
        b0 = lo;
    }

    for (long long f = blockIdx.x; f < F; f += gridDim.x) {
        // ---- phase 0: values in updated order (:172-176), hard decisions, M columns, key reset --------
        if (tid < 128) {
            const int o = lri[f * 128 + uidx[f * 128 + tid]];
            const float m = xm[f * 128 + o];
            L.o[tid] = (unsigned char)o;
            L.w[tid] = __builtin_fabsf(m);
            const u64 hg = __ballot(!(m > 0.0f));                       // order_hard_original (:180)
            if (wave == 0) { if (lane == 0) L.hgL = hg; }
            else {
                const u64 h0 = __ballot(!(xo[f * 128 + o] > 0.0f));      // initial_mrb (:186-187)
                if (lane == 0) { L.hgM = hg; L.mrb0 = h0; }
            }
        } else if (wave == 2) {
            L.Mcol[lane] = transpose64(Mrows[f * 64 + lane], lane);
        } else {
            if (lane == 0) { L.ticket = 0; L.best = ~0ull; L.cw[0] = 0; L.cw[1] = 0; }
        }
        for (int b = tid; b < nblk; b += 256) L.keys[b] = ~0ull;
        __syncthreads();
        // ---- phase 1: byte LUTs (four per wavefront), order-0 discrepancy of the LRB part (:155,:159) --
        build_byte_luts<4>(&L.lut[4 * wave], &L.w[32 * wave], lane);
        if (wave == 0) {
            const u64 d = wave_xor64(((L.mrb0 >> lane) & 1) ? L.Mcol[lane] : 0ull) ^ L.hgL;
            if (lane == 0) L.DL0 = d;
        }
        __syncthreads();
        const u64 DL0 = L.DL0, DM0 = L.mrb0 ^ L.hgM;
        if (truth && wave == 1) {
            const u64 l0 = label[f * 2], l1 = label[f * 2 + 1];
            const int o1 = L.o[lane], o2 = L.o[64 + lane];
            const u64 labL = __ballot(((o1 < 64 ? l0 : l1) >> (o1 & 63)) & 1);
            const u64 labM = __ballot(((o2 < 64 ? l0 : l1) >> (o2 & 63)) & 1);
            const float t = hosd_cost(L, labL ^ L.hgL, labM ^ L.hgM);     // (:181-183)
            if (lane == 0) truth[f] = t;
        }
        if constexpr (RUNS) {
            // ---- phase 2: the scan: this thread's run, a flush per block boundary -----------------------------
            int b = b0;
            float best = INFINITY;
            int bestt = 0;
            for (int t = run0; t < run1; ++t) {
                while (t >= L.boff[b + 1]) {      // the run leaves block b (the next ones may be empty)
                    if (best < INFINITY) atomicMin(&L.keys[b], ((u64)(unsigned)__float_as_int(best) << 32) | (unsigned)bestt);
                    best = INFINITY;
                    ++b;
                }
                u64 DL = DL0, DM = DM0;
                const unsigned tv = L.tl[t];
                hosd_apply(L, make_uchar4((unsigned char)(tv & 63u), (unsigned char)((tv >> 6) & 63u), 0, (unsigned char)(tv >> 12)), DL, DM);
                const float c = hosd_cost(L, DL, DM);
                if (c < best) { best = c; bestt = t; }                     // ascending t: first minimum
            }
            if (best < INFINITY) atomicMin(&L.keys[b], ((u64)(unsigned)__float_as_int(best) << 32) | (unsigned)bestt);
        } else {
            // ---- phase 2: the scan; items are numbered block by block, slice by slice -----------------------
            {
                int item = 0, b = 0, s = nblk > 0 ? block_off[0] : 0, t1 = nblk > 0 ? block_off[1] : 0;
                for (;;) {
                    int want = 0;
                    if (lane == 0) want = atomicAdd(&L.ticket, 1);
                    want = __builtin_amdgcn_readfirstlane(want);
                    // advance (b, s) to item `want`; empty blocks own no item
                    while (b < nblk) {
                        if (s >= t1) { ++b; if (b < nblk) { s = block_off[b]; t1 = block_off[b + 1]; } continue; }
                        if (item == want) break;
                        ++item; s += kHosdChunk;
                    }
                    if (b >= nblk) break;
                    const int e = s + kHosdChunk < t1 ? s + kHosdChunk : t1;
                    float best = INFINITY;
                    int bestt = 0x7FFFFFFF;
                    for (int t = s + lane; t < e; t += 64) {
                        u64 DL = DL0, DM = DM0;
                        hosd_apply(L, teps[t], DL, DM);
                        const float c = hosd_cost(L, DL, DM);
                        if (c < best) { best = c; bestt = t; }                 // ascending t per lane: first minimum
                    }
                    const int wl = wave_argmin_lane(best, bestt);
                    if (lane == wl)
                        atomicMin(&L.keys[b], ((u64)(unsigned)__float_as_int(best) << 32) | (unsigned)bestt);
                    ++item; s += kHosdChunk;
                }
            }
        }
        __syncthreads();
        // ---- phase 3: per-block results, the overall first minimum, its codeword -------------------------
        u64 mine = ~0ull;
        for (int b = tid; b < nblk; b += 256) {
            const u64 k = L.keys[b];
            const bool empty = k == ~0ull;
            block_min[f * nblk + b] = empty ? INFINITY : __int_as_float((int)(k >> 32));
            if (block_arg) block_arg[f * nblk + b] = empty ? -1 : (int)(unsigned)k;
            mine = k < mine ? k : mine;
        }
        if (mine != ~0ull) atomicMin(&L.best, mine);
        __syncthreads();
        if (wave == 0) {
            const u64 k = L.best;
            const bool none = k == ~0ull;
            if (lane == 0) {
                if (metric_out) metric_out[f] = none ? INFINITY : __int_as_float((int)(k >> 32));
                if (best_out) best_out[f] = none ? -1 : (int)(unsigned)k;
            }
            if (cw_out) {
                u64 DL = DL0, DM = DM0;
                if (!none) hosd_apply(L, teps[(unsigned)k], DL, DM);
                const u64 bitsL = DL ^ L.hgL, bitsM = DM ^ L.hgM;
                const int o1 = L.o[lane], o2 = L.o[64 + lane];
                if ((bitsL >> lane) & 1) atomicOr(&L.cw[o1 >> 6], 1ull << (o1 & 63));
                if ((bitsM >> lane) & 1) atomicOr(&L.cw[o2 >> 6], 1ull << (o2 & 63));
                wave_fence();
                if (lane < 2) cw_out[f * 2 + lane] = L.cw[lane];
            }
        }
        __syncthreads();
    }
}

int hosd_ctx_init(ldpc_ctx *ctx)
{
    const ldpc_code &c = ctx->code;
    ctx->hosd_ok = false;
    if (c.n != 128 || c.m != 64 || c.k != 64) return LDPC_OK;   // entry points will report UNSUPPORTED
    std::vector<u64> cols(128, 0);
    for (int r = 0; r < 64; ++r)
        for (int v = 0; v < 128; ++v)
            if (c.H[(size_t)r * 128 + v]) cols[v] |= 1ull << r;
    LDPC_HIP(hipMalloc((void **)&ctx->d_Hcols, sizeof(u64) * 128));
    LDPC_HIP(hipMemcpy(ctx->d_Hcols, cols.data(), sizeof(u64) * 128, hipMemcpyHostToDevice));
    ctx->hosd_ok = true;
    return LDPC_OK;
}

void hosd_ctx_release(ldpc_ctx *ctx)
{
    (void)hipFree(ctx->d_Hcols);
    ctx->d_Hcols = nullptr;
}

static unsigned grid_for(int64_t F, int waves_per_block)
{
    const int64_t want = (F + waves_per_block - 1) / waves_per_block;
    return (unsigned)(want < 1 ? 1 : (want < 8192 ? want : 8192));
}

}  // namespace ldpc

using namespace ldpc;

extern "C" {

int ldpc_hosd_front(ldpc_ctx *ctx, const float *d_order_llr, int64_t F, uint8_t *d_lri, uint8_t *d_uidx,
                    uint64_t *d_M, int32_t *d_nswaps, void *stream)
{
    if (!ctx || F < 0 || (F > 0 && (!d_order_llr || !d_lri || !d_uidx || !d_M)))
        return fail(LDPC_E_ARG, "ldpc_hosd_front: bad arguments");
    if (!ctx->hosd_ok)
        return fail(LDPC_E_UNSUPPORTED, "H-form OSD kernels need n=128, m=k=64; this code is n=%d m=%d k=%d", ctx->code.n,
                    ctx->code.m, ctx->code.k);
    if (F == 0) return LDPC_OK;
    hipLaunchKernelGGL(hosd_front_kernel, dim3(grid_for(F, 4)), dim3(256), 0, (hipStream_t)stream, d_order_llr, (long long)F,
                       reinterpret_cast<const u64 *>(ctx->d_Hcols), d_lri, d_uidx, reinterpret_cast<u64 *>(d_M), d_nswaps);
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

int ldpc_hosd_search(ldpc_ctx *ctx, const float *d_order_llr, const float *d_metric_llr, int64_t F,
                     const uint8_t *d_lri, const uint8_t *d_uidx, const uint64_t *d_M, const uint8_t *d_teps,
                     const int32_t *d_block_off, int32_t nblk, const uint64_t *d_label_bits, float *d_block_min,
                     int32_t *d_block_arg, float *d_truth, uint64_t *d_cw, float *d_metric, int32_t *d_best,
                     void *stream)
{
    if (!ctx || F < 0 || nblk < 0 ||
        (F > 0 && (!d_order_llr || !d_metric_llr || !d_lri || !d_uidx || !d_M || !d_block_off || (nblk > 0 && !d_block_min))))
        return fail(LDPC_E_ARG, "ldpc_hosd_search: bad arguments");
    if (d_truth && !d_label_bits) return fail(LDPC_E_ARG, "ldpc_hosd_search: d_truth needs d_label_bits");
    if (!ctx->hosd_ok)
        return fail(LDPC_E_UNSUPPORTED, "H-form OSD kernels need n=128, m=k=64; this code is n=%d m=%d k=%d", ctx->code.n,
                    ctx->code.m, ctx->code.k);
    if (nblk > kHosdMaxBlocks) return fail(LDPC_E_UNSUPPORTED, "ldpc_hosd_search: %d TEP blocks, at most %d", nblk, kHosdMaxBlocks);
    if (F == 0) return LDPC_OK;
    // (the TEP table's size is device data: both forms are launched and one of them returns at once)
    hipLaunchKernelGGL(hosd_search_kernel<true>, dim3(grid_for(F, 1)), dim3(256), 0, (hipStream_t)stream, d_order_llr, d_metric_llr,
                       (long long)F, d_lri, d_uidx, reinterpret_cast<const u64 *>(d_M),
                       reinterpret_cast<const uchar4 *>(d_teps), d_block_off, (int)nblk,
                       reinterpret_cast<const u64 *>(d_label_bits), d_block_min, d_block_arg, d_truth,
                       reinterpret_cast<u64 *>(d_cw), d_metric, d_best);
    hipLaunchKernelGGL(hosd_search_kernel<false>, dim3(grid_for(F, 1)), dim3(256), 0, (hipStream_t)stream, d_order_llr, d_metric_llr,
                       (long long)F, d_lri, d_uidx, reinterpret_cast<const u64 *>(d_M),
                       reinterpret_cast<const uchar4 *>(d_teps), d_block_off, (int)nblk,
                       reinterpret_cast<const u64 *>(d_label_bits), d_block_min, d_block_arg, d_truth,
                       reinterpret_cast<u64 *>(d_cw), d_metric, d_best);
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

}  // extern "C"
