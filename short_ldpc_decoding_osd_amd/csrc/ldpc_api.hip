// Context management and the NMS entry point of the C ABI (include/ldpc_osd.h).
#include <string.h>

#include "ldpc_internal.h"

using namespace ldpc;

namespace ldpc {
int osd_ctx_init(ldpc_ctx *ctx);      // ldpc_osd.hip
void osd_ctx_release(ldpc_ctx *ctx);  // ldpc_osd.hip
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
    int rc = LDPC_OK;
    do {
        if ((rc = upload(code->chk_ptr, &ctx->d_chk_ptr))) break;
        if ((rc = upload(code->chk_var, &ctx->d_chk_var))) break;
        if ((rc = upload(code->var_ptr, &ctx->d_var_ptr))) break;
        if ((rc = upload(code->var_edge, &ctx->d_var_edge))) break;
        ctx->blocksum_cap = 1 << 20;  // compaction scratch: 2^20 blocks x 2048 frames
        if (hipMalloc((void **)&ctx->d_blocksum, sizeof(int32_t) * ctx->blocksum_cap) != hipSuccess) {
            rc = fail(LDPC_E_NOMEM, "ldpc_ctx_create: scratch allocation failed");
            break;
        }
        if ((rc = probe_dpp(&ctx->dpp_ror_up))) break;
        if ((rc = osd_ctx_init(ctx))) break;
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
    (void)hipFree(ctx->d_chk_ptr); (void)hipFree(ctx->d_chk_var);
    (void)hipFree(ctx->d_var_ptr); (void)hipFree(ctx->d_var_edge);
    (void)hipFree(ctx->d_blocksum);
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

}  // extern "C"
