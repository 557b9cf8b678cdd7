// Context management and the NMS entry point of the C ABI (include/ldpc_osd.h).
#include <string.h>

#include "ldpc_internal.h"

using namespace ldpc;

namespace ldpc {
int osd_ctx_init(ldpc_ctx *ctx);      // ldpc_osd.hip
void osd_ctx_release(ldpc_ctx *ctx);  // ldpc_osd.hip
int hosd_ctx_init(ldpc_ctx *ctx);     // ldpc_hosd.hip
void hosd_ctx_release(ldpc_ctx *ctx); // ldpc_hosd.hip
}  // namespace ldpc

template <typename T>
static int upload(const std::vector<T> &v, T **dst)
{
    LDPC_HIP(hipMalloc((void **)dst, sizeof(T) * (v.size() ? v.size() : 1)));
    if (!v.empty()) LDPC_HIP(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return LDPC_OK;
}

extern "C" {

int ldpc_ctx_create(const ldpc_code *code, int32_t device, ldpc_ctx **out)
{
    if (!code || !out) return fail(LDPC_E_ARG, "ldpc_ctx_create: null argument");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(LDPC_E_HIP, "ldpc_ctx_create: no HIP device visible (%s) -- this library has no CPU path",
                    hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(LDPC_E_ARG, "ldpc_ctx_create: device %d of %d", device, ndev);
    int prev = 0;
    LDPC_HIP(hipGetDevice(&prev));
    LDPC_HIP(hipSetDevice(device));
    ldpc_ctx *ctx = new ldpc_ctx();
    ctx->device = device;
    ctx->code = *code;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->cu_count = prop.multiProcessorCount;
    }
    int rc = LDPC_OK;
    do {
        if ((rc = upload(code->chk_ptr, &ctx->d_chk_ptr))) break;
        if ((rc = upload(code->chk_var, &ctx->d_chk_var))) break;
        if ((rc = upload(code->var_ptr, &ctx->d_var_ptr))) break;
        if ((rc = upload(code->var_edge, &ctx->d_var_edge))) break;
        if ((rc = probe_dpp(&ctx->dpp_ror_up, &ctx->dpp_wave_rol_dir))) break;
        // event pool of ldpc_pipeline_run's timing slots: created (and recorded once: the first record of an
        // event sets up its signal and is slow) here, so that decode calls never create anything.  The warm-up
        // records go to a private stream -- the legacy NULL stream would synchronise with every blocking stream
        // of the process and break a capture in progress elsewhere.
        ctx->timing = new hipEvent_t[LDPC_TIMING_SLOTS * 6]();
        hipStream_t warm = nullptr;
        if (hipStreamCreateWithFlags(&warm, hipStreamNonBlocking) != hipSuccess) { rc = fail(LDPC_E_HIP, "ldpc_ctx_create: stream creation failed"); break; }
        for (int i = 0; i < LDPC_TIMING_SLOTS * 6 && !rc; ++i) {
            hipError_t he = hipEventCreate(&ctx->timing[i]);
            if (he == hipSuccess) he = hipEventRecord(ctx->timing[i], warm);
            if (he != hipSuccess) rc = hip_fail(he, "timing event pool");
        }
        if (!rc && hipStreamSynchronize(warm) != hipSuccess) rc = fail(LDPC_E_HIP, "ldpc_ctx_create: stream sync failed");
        (void)hipStreamDestroy(warm);
        if (rc) break;
        if ((rc = osd_ctx_init(ctx))) break;
        if ((rc = hosd_ctx_init(ctx))) break;
    } while (0);
    (void)hipSetDevice(prev);
    if (rc) { ldpc_ctx_destroy(ctx); return rc; }
    *out = ctx;
    return LDPC_OK;
}

void ldpc_ctx_destroy(ldpc_ctx *ctx)
{
    if (!ctx) return;
    osd_ctx_release(ctx);
    hosd_ctx_release(ctx);
    (void)hipFree(ctx->d_chk_ptr); (void)hipFree(ctx->d_chk_var);
    (void)hipFree(ctx->d_var_ptr); (void)hipFree(ctx->d_var_edge);
    if (ctx->timing) {
        for (int i = 0; i < LDPC_TIMING_SLOTS * 6; ++i) if (ctx->timing[i]) (void)hipEventDestroy(ctx->timing[i]);
        delete[] ctx->timing;
    }
    delete ctx;
}

int ldpc_ctx_nms_kernel(const ldpc_ctx *ctx)
{
    if (!ctx) return fail(LDPC_E_ARG, "ldpc_ctx_nms_kernel: null ctx");
    return ctx->code.qc16_ccsds ? LDPC_NMS_QC16 : LDPC_NMS_GENERIC;
}

int ldpc_nms_decode(ldpc_ctx *ctx, const float *d_llr, int64_t B, int32_t T, const float *alpha, float w_in,
                    float w_out, float *d_soft, float *d_traj, uint64_t *d_hard, uint8_t *d_fail, int32_t kernel,
                    void *stream)
{
    if (!ctx || B < 0 || (!d_llr && B > 0)) return fail(LDPC_E_ARG, "ldpc_nms_decode: bad arguments");
    if (T < 0 || T > kMaxIters) return fail(LDPC_E_ARG, "ldpc_nms_decode: T=%d outside 0..%d", T, kMaxIters);
    if (T > 0 && !alpha) return fail(LDPC_E_ARG, "ldpc_nms_decode: alpha is NULL");
    if (B == 0) return LDPC_OK;
    return launch_nms(ctx, d_llr, B, T, alpha, w_in, w_out, d_soft, d_traj, d_hard, d_fail, kernel, (hipStream_t)stream);
}

int ldpc_nms_traj_rows(ldpc_ctx *ctx, const float *d_llr, const int32_t *d_index, const int32_t *d_count, int64_t F, int32_t T,
                       const float *alpha, float w_in, float w_out, float *d_rows, int32_t kernel, void *stream)
{
    if (!ctx || F < 0 || ((!d_llr || !d_index || !d_rows) && F > 0)) return fail(LDPC_E_ARG, "ldpc_nms_traj_rows: bad arguments");
    if (T < 0 || T > kMaxIters) return fail(LDPC_E_ARG, "ldpc_nms_traj_rows: T=%d outside 0..%d", T, kMaxIters);
    if (T > 0 && !alpha) return fail(LDPC_E_ARG, "ldpc_nms_traj_rows: alpha is NULL");
    if (!d_count) return fail(LDPC_E_ARG, "ldpc_nms_traj_rows: d_count is NULL (the number of listed frames is device data)");
    if (F == 0) return LDPC_OK;
    return launch_nms(ctx, d_llr, F, T, alpha, w_in, w_out, nullptr, nullptr, nullptr, nullptr, kernel, (hipStream_t)stream, d_index, d_count, d_rows);
}

int ldpc_pipeline_run(ldpc_ctx *ctx, const ldpc_pipeline *p, void *stream)
{
    if (!ctx || !p) return fail(LDPC_E_ARG, "ldpc_pipeline_run: null argument");
    if (!p->d_hard || !p->d_fail) return fail(LDPC_E_ARG, "ldpc_pipeline_run: d_hard and d_fail are required");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t *ev = nullptr;
    unsigned *rec = nullptr;            // which of the slot's events THIS run records (ldpc_pipeline_timing reads only those)
    if (p->timing_slot >= 0) {
        if (p->timing_slot >= LDPC_TIMING_SLOTS || !ctx->timing) return fail(LDPC_E_ARG, "ldpc_pipeline_run: timing_slot %d", p->timing_slot);
        ev = ctx->timing + p->timing_slot * 6;
        rec = &ctx->timing_recorded[p->timing_slot];
        *rec = 0;
    }
#define LDPC_EV(i) do { if (ev) { LDPC_HIP(hipEventRecord(ev[i], s)); *rec |= 1u << (i); } } while (0)
    int rc;
    LDPC_EV(0);
    if ((rc = ldpc_nms_decode(ctx, p->d_llr, p->B, p->T, p->alpha, p->w_in, p->w_out, p->d_soft, nullptr, p->d_hard,
                              p->d_fail, p->nms_kernel, stream))) return rc;
    LDPC_EV(1);
    const bool want_eval = p->d_label_bits && p->d_nms_counts;
    if (p->osd_enable) {
        if (!p->d_index || !p->d_count || !p->d_cw)
            return fail(LDPC_E_ARG, "ldpc_pipeline_run: OSD stage needs d_index, d_count, d_cw");
        if (want_eval && p->B > 0) {   // counters and the compaction's counting pass share one kernel
            if ((rc = eval_and_compact(ctx, p->d_hard, p->d_label_bits, p->d_fail, p->B, p->d_nms_counts, p->d_index,
                                       p->d_count, s))) return rc;
        } else if ((rc = ldpc_compact(ctx, p->d_fail, p->B, p->d_index, p->d_count, stream))) return rc;
        LDPC_EV(2);
        if (p->d_perm && p->d_parity) {   // caller wants the front-end results: two kernels
            if ((rc = ldpc_osd_front(ctx, p->d_llr, p->d_index, p->d_count, p->B, p->d_perm, p->d_parity, nullptr, stream))) return rc;
            LDPC_EV(3);
            // the search kernel counts its own successes where it can (order-2 scan); otherwise the counting launch follows
            // the search's closing event, so that ms[2] is the search alone for every algorithm
            const bool counted = p->d_label_bits && p->d_osd_counts;
            bool fused = false;
            if ((rc = osd_search_counted(ctx, p->d_llr, p->d_index, p->d_count, p->B, p->d_perm, p->d_parity, &p->osd, p->d_cw,
                                         p->d_metric, p->d_best, p->d_ntep, counted ? p->d_label_bits : nullptr,
                                         counted ? p->d_osd_counts : nullptr, s, &fused))) return rc;
            LDPC_EV(4);
            if (counted && !fused &&
                (rc = ldpc_osd_counts(ctx, p->d_cw, p->d_label_bits, p->d_index, p->d_count, p->d_ntep, p->B, p->d_osd_counts, stream))) return rc;
            return LDPC_OK;
        } else {                          // ldpc_osd_decode: one fused kernel (conventional order 2) or the stream's workspace
            LDPC_EV(3);
            if (!ctx->osd_ok) return fail(LDPC_E_UNSUPPORTED, "OSD kernels need an (n=128, k=64) code; this one is (%d,%d)", ctx->code.n, ctx->code.k);
            if (p->osd.order < 0 || p->osd.order > 3 || p->osd.algo < 0 || p->osd.algo > 2 || (p->osd.algo == LDPC_OSD_PB && p->osd.order < 1))
                return fail(LDPC_E_ARG, "ldpc_pipeline_run: OSD order %d / algorithm %d", p->osd.order, p->osd.algo);
            const bool counted = p->d_label_bits && p->d_osd_counts;
            bool fused = false;
            if (p->B > 0 && (rc = osd_decode_counted(ctx, p->d_llr, p->d_index, p->d_count, p->B, &p->osd, p->d_cw, p->d_metric, p->d_best,
                                                     p->d_ntep, counted ? p->d_label_bits : nullptr, counted ? p->d_osd_counts : nullptr, s, &fused))) return rc;
            LDPC_EV(4);
            if (counted && !fused &&
                (rc = ldpc_osd_counts(ctx, p->d_cw, p->d_label_bits, p->d_index, p->d_count, p->d_ntep, p->B, p->d_osd_counts, stream))) return rc;
            return LDPC_OK;
        }
        LDPC_EV(4);
        if (p->d_label_bits && p->d_osd_counts &&
            (rc = ldpc_osd_counts(ctx, p->d_cw, p->d_label_bits, p->d_index, p->d_count, p->d_ntep, p->B, p->d_osd_counts,
                                  stream))) return rc;
    } else if (want_eval && (rc = ldpc_eval_counts(ctx, p->d_hard, p->d_label_bits, p->d_fail, p->B, p->d_nms_counts, stream)))
        return rc;
    return LDPC_OK;
}

int ldpc_pipeline_timing(ldpc_ctx *ctx, int32_t slot, float *ms)
{
    if (!ctx || !ms || slot < 0 || slot >= LDPC_TIMING_SLOTS || !ctx->timing)
        return fail(LDPC_E_ARG, "ldpc_pipeline_timing: no timed run in slot %d", slot);
    hipEvent_t *ev = ctx->timing + slot * 6;
    const unsigned rec = ctx->timing_recorded[slot];     // events of the LAST run that used the slot (all were recorded once at creation)
    ms[0] = ms[1] = ms[2] = 0.0f;
    if ((rec & 3u) != 3u) return fail(LDPC_E_ARG, "ldpc_pipeline_timing: no timed run in slot %d", slot);
    LDPC_HIP(hipEventElapsedTime(&ms[0], ev[0], ev[1]));
    if ((rec & 0xCu) == 0xCu && hipEventElapsedTime(&ms[1], ev[2], ev[3]) != hipSuccess) { ms[1] = 0.0f; (void)hipGetLastError(); }
    if ((rec & 0x18u) == 0x18u && hipEventElapsedTime(&ms[2], ev[3], ev[4]) != hipSuccess) { ms[2] = 0.0f; (void)hipGetLastError(); }
    return LDPC_OK;
}

}  // extern "C"
