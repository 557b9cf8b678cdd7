// Internal declarations shared by the translation units of libldpcosd.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ldpc_osd.h"

namespace ldpc {

// thread-local error text behind ldpc_last_error()
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_fail(hipError_t e, const char *what);

#define LDPC_HIP(call)                                      \
    do {                                                    \
        hipError_t e__ = (call);                            \
        if (e__ != hipSuccess) return ::ldpc::hip_fail(e__, #call); \
    } while (0)

constexpr int kMaxIters = 64;  // alpha[] travels by value in the kernel arguments

struct AlphaArg {
    float a[kMaxIters];
};

// QC structure of a code whose H is an array of 16x16 circulants (CCSDS (128,64) shape):
// check (br, i) is joined to variable (bc, (i + s) mod 16) for every term (br, bc, s).
struct QcTerm {
    int br, bc, s;
};

}  // namespace ldpc

struct ldpc_code {
    int n = 0, m = 0, k = 0, max_chk_degree = 0, max_var_degree = 0, E = 0;
    std::vector<int32_t> H;         // [m][n]
    std::vector<int32_t> G;         // [k][n]
    std::vector<int32_t> chk_ptr;   // [m+1]   edges are numbered check-major
    std::vector<int32_t> chk_var;   // [E]     variable of edge e
    std::vector<int32_t> var_ptr;   // [n+1]
    std::vector<int32_t> var_edge;  // [E]     edge ids of variable v, ascending check index
    bool qc16_ccsds = false;        // H equals the compiled-in CCSDS (128,64) circulant table
};

struct ldpc_ctx {
    int device = 0, cu_count = 256;
    ldpc_code code;
    // generic NMS tables
    int32_t *d_chk_ptr = nullptr, *d_chk_var = nullptr, *d_var_ptr = nullptr, *d_var_edge = nullptr;
    // OSD constants (n = 128, k = 64 only)
    uint64_t *d_Gcols = nullptr;   // [128] column v of G as a 64-bit word (bit r = G[r][v])
    uint8_t *d_tep = nullptr;      // TEP supports, order <= 3: [43745][4] (i, j, l, weight)
    uint64_t *d_Hcols = nullptr;   // [128] column v of H as a 64-bit word (bit r = H[r][v]); n = 128, m = 64 only
    bool dpp_ror_up = true;        // probed: row_ror:n moves data towards higher lanes
    int dpp_wave_rol_dir = 0;      // probed: wave_rol:1 -- +1 lane j receives lane j-1, -1 lane j+1, 0 unusable
    bool osd_ok = false;
    bool hosd_ok = false;
    void *osd_state = nullptr;     // ldpc::OsdState (TEP tables; per-stream workspaces behind its mutex)
    hipEvent_t *timing = nullptr;  // [LDPC_TIMING_SLOTS][6] events of ldpc_pipeline_run, created with the context
    unsigned timing_recorded[LDPC_TIMING_SLOTS] = {};   // bit i: event i of the slot was recorded by the last run that used it
};

namespace ldpc {

// host helpers (ldpc_host.cpp)
int gf2elim(int32_t *M, int m, int n, std::vector<int32_t> *swaps);  // returns rows left
int build_code(ldpc_code &c);  // fills G, graph tables, qc flag from c.H/m/n
int64_t tep_table(int k, int order, uint8_t *supports, int64_t *boundaries);
int64_t tep_table_fs(int k, int w, uint8_t *supports);
int64_t hosd_pattern_teps(int nseg, const int32_t *bounds, const int32_t *pattern, uint8_t *teps);

// launchers (one per .hip file)
int launch_nms(ldpc_ctx *ctx, const float *d_llr, int64_t B, int T, const float *alpha, float w_in, float w_out,
               float *d_soft, float *d_traj, uint64_t *d_hard, uint8_t *d_fail, int kernel, hipStream_t st,
               const int32_t *d_index = nullptr, const int32_t *d_count = nullptr, float *d_rows = nullptr);
int probe_dpp(bool *ror_up, int *wave_rol_dir);
int osd_search_counted(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                       const uint8_t *d_perm, const uint64_t *d_parity, const ldpc_osd_params *p, uint64_t *d_cw, float *d_metric,
                       int32_t *d_best, int32_t *d_ntep, const uint64_t *d_label, int64_t *d_counts, hipStream_t s,
                       bool *counted_by_search = nullptr);
int osd_decode_counted(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                       const ldpc_osd_params *p, uint64_t *d_cw, float *d_metric, int32_t *d_best, int32_t *d_ntep,
                       const uint64_t *d_label, int64_t *d_counts, hipStream_t s, bool *counted_by_search);
int eval_and_compact(ldpc_ctx *ctx, const uint64_t *d_hard, const uint64_t *d_label, const uint8_t *d_fail, int64_t B,
                     int64_t *d_counts, int32_t *d_index, int32_t *d_count, hipStream_t st);

}  // namespace ldpc
