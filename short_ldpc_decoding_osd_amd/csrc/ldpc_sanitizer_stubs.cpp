// Sanitizer build only (python -m short_ldpc_decoding_osd_amd.build --asan): the device entry points of include/ldpc_osd.h
// as stubs, so that a HOST-ONLY library (ldpc_host.cpp under -fsanitize=address,undefined) still exports every symbol the
// binding checks at load time.  Never part of libldpcosd.so.  Each stub fails loudly: there is no CPU decode path.
namespace ldpc { int fail(int code, const char *fmt, ...); }
#define LDPC_STUB(name) extern "C" int name(...) { return ldpc::fail(-5, "sanitizer build: " #name " exists only in the HIP build"); }
LDPC_STUB(ldpc_ctx_create)
LDPC_STUB(ldpc_ctx_destroy)
LDPC_STUB(ldpc_ctx_nms_kernel)
LDPC_STUB(ldpc_nms_decode)
LDPC_STUB(ldpc_eval_counts)
LDPC_STUB(ldpc_compact)
LDPC_STUB(ldpc_pack_bits)
LDPC_STUB(ldpc_unpack_bits)
LDPC_STUB(ldpc_osd_ge)
LDPC_STUB(ldpc_osd_front)
LDPC_STUB(ldpc_osd_reserve)
LDPC_STUB(ldpc_osd_reserve_stream)
LDPC_STUB(ldpc_osd_release_stream)
LDPC_STUB(ldpc_osd_decode)
LDPC_STUB(ldpc_osd_index_errors)
LDPC_STUB(ldpc_osd_search)
LDPC_STUB(ldpc_osd_tep_eval)
LDPC_STUB(ldpc_osd_counts)
LDPC_STUB(ldpc_hosd_front)
LDPC_STUB(ldpc_hosd_search)
LDPC_STUB(ldpc_pipeline_run)
LDPC_STUB(ldpc_pipeline_timing)
