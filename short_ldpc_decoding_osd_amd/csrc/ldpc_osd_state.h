// Host-side OSD state of a context: TEP tables and the per-stream workspaces (ldpc_osd.hip, ldpc_osd_pb.hip).
#pragma once

#include "ldpc_wave.h"

namespace ldpc {

// Everything a decode call writes besides the caller's buffers lives in a workspace that belongs to the
// STREAM the call is issued on (created on that stream's first call, grown on demand, found under a mutex):
// calls on different streams of one context never share scratch, so they may overlap freely.
struct StreamWs {
    unsigned char *d_perm = nullptr;  // ldpc_osd_decode: front-end results [cap][128]
    u64 *d_parity = nullptr;          //                                    [cap][64]
    int64_t cap = 0;
    int *d_index_safe = nullptr;      // ldpc_osd_params.y_frames: the sanitised copy of the caller's frame list [index_cap]
    int64_t index_cap = 0;
    int *d_pb_ctl = nullptr;          // PB-OSD: list lengths and tickets (kPbCtlInts ints, zeroed per call)
    int *d_pb_list = nullptr;         // PB-OSD: [2][kPbSub * pb_sub_cap] frames handed on: list A (chunk kernel), B (list replay)
    void *d_pb_carry = nullptr;       // PB-OSD: [kPbHeavyCap] records of the long searches handed to the workgroup kernel ("list C")
    void *d_pb_prep = nullptr;        // PB-OSD: [pb_cap] records of the frames the singles kernel hands on (1536 B each: |y'|, P', CDF table, permutation, scalars)
    int64_t pb_cap = 0, pb_sub_cap = 0;
    void *d_pb_spill = nullptr;       // PB-OSD sequential kernel: frontier overflow [waves][stride]
    int64_t pb_spill_stride = 0;
    bool captured = false;            // a hipGraph was captured on this stream: its nodes hold these pointers, so the
                                      // buffers can no longer be moved (growth is refused until ldpc_osd_release_stream)
};
// A device-scope atomic on ONE word saturates at ~88 returning fetch-adds per us (MI355X_MICROARCH.md): 11 k frames
// appended to one list through one counter kept the first PB stage at 150 us whatever else it did.  List A is
// therefore 16 sub-lists (frame f -> sub-list f mod 16), every length on its own 128-byte line; workgroup b of the
// chunk kernel serves entry b / 16 of sub-list b mod 16 (no tickets).
constexpr int kPbSub = 16, kPbCtlLine = 32;
constexpr int kPbCtlLenA = 0, kPbCtlLenB = kPbSub * kPbCtlLine, kPbCtlTicketB = kPbCtlLenB + kPbCtlLine;
constexpr int kPbCtlLenC = kPbCtlTicketB + kPbCtlLine, kPbCtlTicketC = kPbCtlLenC + kPbCtlLine;
constexpr int kPbCtlStartA = kPbCtlTicketC + kPbCtlLine;      // [kPbSub] lines: frames of a sub-list the chunk kernel has STARTED (its tail rule)
constexpr int kPbCtlLenC2 = kPbCtlStartA + kPbSub * kPbCtlLine;   // list C's second half: the searches expected to be short
constexpr int kPbCtlInts = kPbCtlLenC2 + kPbCtlLine;
constexpr int kPbCoopHalf = 4096;     // long searches a call may hand to the workgroup-per-frame kernel: this many expected to run
constexpr int kPbHeavyCap = 2 * kPbCoopHalf;      // to the end of the table (served first) + this many others
constexpr int kPbSeqBlocks = 64;      // grid of the sequential PB kernel (each of its 4 x 64 waves owns a spill area)

struct OsdState {
    int64_t ntep[4] = {0, 0, 0, 0};
    uchar4 *d_tep_fs = nullptr;       // FS visit order, weight classes 1..3 back to back
    int *d_base2 = nullptr;           // order-2 ranks: number of index pairs with a larger sum
    double *d_cdf_half = nullptr;     // PB-OSD: P[Bin(64, 1/2) <= b], b = 0..64
    unsigned long long *d_index_errors = nullptr;   // ldpc_osd_params.y_frames: out-of-range entries met so far
    int fs_off[4] = {0, 0, 0, 0}, fs_cnt[4] = {0, 0, 0, 0};
    ldpc_pb_tuning pb_tuning;         // PB-OSD hand-over schedule and chunk targets (ldpc_ctx_set_pb_tuning); read under `mu`
    std::mutex mu;                    // guards `ws`, `reserve_frames` and `pb_tuning`
    std::unordered_map<hipStream_t, StreamWs> ws;
    int64_t reserve_frames = 0;       // ldpc_osd_reserve: smallest capacity any workspace is created with
};

static inline OsdState *state(ldpc_ctx *ctx) { return reinterpret_cast<OsdState *>(ctx->osd_state); }

// ldpc_osd_pb.hip
int pb_ctx_init(ldpc_ctx *ctx);
int launch_pb(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
              const unsigned char *d_perm, const u64 *d_parity, const ldpc_osd_params *p, uint64_t *d_cw, float *d_metric,
              int32_t *d_best, int32_t *d_ntep, hipStream_t s);
int pb_reserve(ldpc_ctx *ctx, hipStream_t s, int64_t frames, int order);
// ldpc_osd.hip
bool stream_capturing(hipStream_t s);

}  // namespace ldpc
