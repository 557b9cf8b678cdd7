// placeholder -- replaced by the OSD kernels
#include "ldpc_internal.h"
using namespace ldpc;
namespace ldpc {
int osd_ctx_init(ldpc_ctx *) { return LDPC_OK; }
void osd_ctx_release(ldpc_ctx *) {}
}
extern "C" {
int ldpc_osd_ge(ldpc_ctx *, const uint64_t *, int64_t, uint64_t *, uint8_t *, int32_t *, void *) { return fail(LDPC_E_UNSUPPORTED, "not built yet"); }
int ldpc_osd_front(ldpc_ctx *, const float *, const int32_t *, const int32_t *, int64_t, uint8_t *, uint64_t *, int32_t *, void *) { return fail(LDPC_E_UNSUPPORTED, "not built yet"); }
int ldpc_osd_decode(ldpc_ctx *, const float *, const int32_t *, const int32_t *, int64_t, const ldpc_osd_params *, uint64_t *, float *, int32_t *, int32_t *, void *) { return fail(LDPC_E_UNSUPPORTED, "not built yet"); }
int ldpc_osd_counts(ldpc_ctx *, const uint64_t *, const uint64_t *, const int32_t *, const int32_t *, const int32_t *, int64_t, int64_t *, void *) { return fail(LDPC_E_UNSUPPORTED, "not built yet"); }
}
