/*
 * ldpc_osd.h -- C ABI of libldpcosd.so: MI355X (gfx950) short-LDPC NMS + OSD decoder.
 *
 * This is the drop-in boundary for the reference's decoding hot path.  The reference
 * (lgw-frank/Short_LDPC_Decoding_OSD) has no FFI of its own -- its boundary is a Python
 * call surface -- so every entry point cites the reference function it replaces
 * (file:line relative to LDPC_128/ in the reference tree).  The Python mirror of that
 * surface lives in short_ldpc_decoding_osd_amd/ and binds these symbols with ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - pointers named d_* are DEVICE pointers owned by the caller (e.g. torch tensors'
 *     data_ptr()); all other pointers are host memory
 *   - every call returns 0 on success or a negative LDPC_E_* code; ldpc_last_error()
 *     returns a thread-local message.  Nothing aborts or throws across the ABI.
 *   - device calls are asynchronous on `stream`; no allocation and no host sync happens
 *     inside a decode call (graph-capturable)
 *   - bit packing: bit v of a frame lives in word v/64, bit position v%64 (LSB first)
 */
#ifndef LDPC_OSD_H
#define LDPC_OSD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDPC_OSD_ABI_VERSION 4

enum {
    LDPC_OK = 0,
    LDPC_E_ARG = -1,         /* bad argument (null pointer, size, order ...)          */
    LDPC_E_IO = -2,          /* cannot read / parse a file                            */
    LDPC_E_CODE = -3,        /* H.G^T != 0, rank problems                             */
    LDPC_E_HIP = -4,         /* a HIP runtime call failed (message has the details)   */
    LDPC_E_UNSUPPORTED = -5, /* shape not handled by the kernels (e.g. OSD on n != 128) */
    LDPC_E_NOMEM = -6
};

/* OSD search algorithms for ldpc_osd_decode */
enum {
    LDPC_OSD_CONVENTIONAL = 0, /* FS_OSD/convention_osd.py:49-76  convention_osd_main */
    LDPC_OSD_FS = 1,           /* FS_OSD/fs_testing.py:129-161    fs_osd               */
    LDPC_OSD_PB = 2            /* PB_OSD/pb_testing.py:100-149    pb_osd               */
};

/* NMS kernel selection (0 = let the library choose) */
enum {
    LDPC_NMS_AUTO = 0,
    LDPC_NMS_GENERIC = 1, /* any code: one frame per wavefront, messages staged in LDS  */
    LDPC_NMS_QC16 = 2     /* 16x16-circulant codes of the CCSDS (128,64) shape: one frame
                             per 16-lane DPP row, messages in registers                  */
};

typedef struct ldpc_code ldpc_code; /* host-side code definition (H, G, Tanner graph) */
typedef struct ldpc_ctx ldpc_ctx;   /* per-device constants + scratch                  */

const char *ldpc_last_error(void);
int ldpc_abi_version(void);

/* ---------------------------------------------------------------------------------------
 * Code definition (host, one-time).
 * Replaces Ldpc_128_testing/fill_matrix_info.py: Code.load_code :70-129 (alist -> H),
 * Code.gf2elim :7-42 and Code.generator_matrix :44-69 (systematic G, H.G^T = 0 check).
 * ------------------------------------------------------------------------------------- */
int ldpc_code_from_alist(const char *path, ldpc_code **out);
int ldpc_code_from_dense(const int32_t *H /*[m][n] row-major 0/1*/, int32_t m, int32_t n, ldpc_code **out);
void ldpc_code_destroy(ldpc_code *code);
int ldpc_code_dims(const ldpc_code *code, int32_t *n, int32_t *m, int32_t *k, int32_t *max_chk_degree);
int ldpc_code_get_H(const ldpc_code *code, int32_t *H /*[m][n]*/);
int ldpc_code_get_G(const ldpc_code *code, int32_t *G /*[k][n]*/);

/* GF(2) Gauss-Jordan with the reference's pivot rule on one host matrix, in place.
 * Replaces full_gf2elim, PB_OSD/pb_testing.py:231-266 (used for code construction; the
 * per-frame eliminations of the OSD run on the device, see ldpc_osd_ge).
 * swaps: [n][2] pairs (j, col); *rows_out = rows left after deleting all-zero rows.     */
int ldpc_gf2elim_host(int32_t *M /*[m][n]*/, int32_t m, int32_t n, int32_t *swaps, int32_t *nswaps,
                      int32_t *rows_out);

/* Test-error-pattern table, FS_OSD/convention_osd.py:13-47 (generate_teps, query_boundary):
 * all supports of weight 0..order over k positions, each weight class in lexicographic
 * order stably re-sorted by descending index sum.  supports: [count][3] uint8, 0xFF padded
 * (NULL = only return the count).  boundaries: [order+1] cumulative counts (may be NULL).
 * Returns the number of patterns or a negative error.  order <= 3.                        */
int64_t ldpc_tep_table(int32_t k, int32_t order, uint8_t *supports, int64_t *boundaries);

/* FS-OSD visit order of one weight class, FS_OSD/fs_testing.py:32-49 (generate_sequential_teps):
 * lexicographic combinations of range(k) with the indicator vector reversed (support {k-1-p}).
 * supports: [C(k,weight)][3] uint8 ascending positions, 0xFF padded (NULL = only the count).   */
int64_t ldpc_tep_table_fs(int32_t k, int32_t weight, uint8_t *supports);

/* CRC-32C of a host buffer: the record checksum of the TFRecord files the reference's stages
 * exchange (Ldpc_128_testing/data_generating.py:16-26, read_TFdata.py:18-29).                   */
uint32_t ldpc_crc32c(const void *data, uint64_t len);

/* ---------------------------------------------------------------------------------------
 * Device context: uploads the packed H/G, Tanner-graph tables and TEP tables of one code
 * to one GPU; one ctx per device.  The constants (and the event pool of ldpc_pipeline_run's
 * timing slots) are created here and never change afterwards.  The only state that decode calls
 * touch is scratch memory, and that is kept PER STREAM: a mutex-guarded map stream -> workspace
 * (front-end results of ldpc_osd_decode, PB-OSD frame lists and frontier areas), created on a
 * stream's first call and grown on demand.  Hence decode calls issued on DIFFERENT streams of one
 * context -- from one host thread or several -- may run concurrently, for every algorithm;
 * calls on the SAME stream are ordered by the stream.  ldpc_compact needs no scratch at all.
 * (ldpc_pipeline_timing slots are shared: concurrent pipelines must use different slots.)
 * Captured graphs: a hipGraph captured on a stream bakes in the addresses of THAT stream's workspace and
 * may later be launched on any stream.  The graph therefore owns the capture stream's workspace: do not
 * replay it concurrently with other work that uses the same workspace (eager calls on the capture stream,
 * a second replay of the same or of another graph captured on that stream); the library refuses to grow
 * (= move) a workspace once a capture has used it (LDPC_E_NOMEM: reserve enough before capturing), and
 * ldpc_osd_release_stream frees it when the graphs are gone or the stream is destroyed.
 * ------------------------------------------------------------------------------------- */
int ldpc_ctx_create(const ldpc_code *code, int32_t device, ldpc_ctx **out);
void ldpc_ctx_destroy(ldpc_ctx *ctx);
/* which NMS kernel LDPC_NMS_AUTO resolves to for this code (LDPC_NMS_GENERIC / _QC16) */
int ldpc_ctx_nms_kernel(const ldpc_ctx *ctx);

/* ---------------------------------------------------------------------------------------
 * Normalised min-sum BP, fixed T flooding iterations, no early stop.
 * Replaces Decoder_Layer.call / belief_propagation_op / compute_vc / compute_cv2 /
 * marginalize, Ldpc_128_testing/ms_test.py:99-242, and the hard decision + syndrome of
 * Decoding_model.get_eval :39,:45.
 *   d_llr     [B][n] f32 channel values (BPSK 0 -> +1, unscaled)
 *   alpha     host [T] effective check normalisers = softplus(stored weight) (:207-208);
 *             NMS-1 passes T copies of one value
 *   w_in/w_out effective bit weights of NMS-2/3 (:127-131, :222-225); 1.0f for NMS-1
 *   d_soft    [B][n] f32 posterior after iteration T (nullable)
 *   d_traj    [T][B][n] f32 posterior after iterations 1..T (nullable; the reference's
 *             soft_output_list without its slot 0, which is d_llr itself)
 *   d_hard    [B][ceil(n/64)] u64 packed hard decisions (soft > 0 ? 0 : 1) (nullable)
 *   d_fail    [B] u8, 1 = non-zero syndrome (nullable)
 * ------------------------------------------------------------------------------------- */
int ldpc_nms_decode(ldpc_ctx *ctx, const float *d_llr, int64_t B, int32_t T, const float *alpha, float w_in,
                    float w_out, float *d_soft, float *d_traj, uint64_t *d_hard, uint8_t *d_fail, int32_t kernel,
                    void *stream);

/* The trajectories of LISTED frames only, in the layout the reference buffers them: collect_failed_output_selective,
 * ms_test.py:55-64 (driven by Decoding_model.call :30-34 with the index of get_eval :51) -- T + 1 rows per failed
 * frame, row 0 the channel values (soft_output_list[0]), row t the posterior after iteration t.
 *   d_index / d_count  the frame list as ldpc_compact writes it; the number of frames is min(*d_count, F), read ON THE
 *                      DEVICE; F is the capacity of d_rows
 *   d_rows             [F][T+1][n] f32
 * The frames are decoded again (same kernel, same arithmetic: every row equals the corresponding row of
 * ldpc_nms_decode's d_traj); nothing else is written.  At 2.5 dB a quarter of the frames fail: 1.4 KiB of rows per
 * input frame instead of the 5 KiB of a full [T][B][n] trajectory.                                                   */
int ldpc_nms_traj_rows(ldpc_ctx *ctx, const float *d_llr, const int32_t *d_index, const int32_t *d_count, int64_t F,
                       int32_t T, const float *alpha, float w_in, float w_out, float *d_rows, int32_t kernel, void *stream);

/* Error statistics, Decoding_model.get_eval, ms_test.py:36-54.
 * d_counts[5] += {frames, frames_in_error, bit_errors, undetected, syndrome_failures}.
 * The caller zeroes d_counts; d_fail may be NULL (then undetected/syndrome are not counted). */
int ldpc_eval_counts(ldpc_ctx *ctx, const uint64_t *d_hard, const uint64_t *d_label_bits, const uint8_t *d_fail,
                     int64_t B, int64_t *d_counts, void *stream);

/* Stream compaction of the failed-frame flags, the `tf.where(syndrome != 0)` of
 * ms_test.py:51: d_index[0..count) = ascending frame numbers with d_flag != 0.
 * d_index: [B] i32, d_count: [1] i32.                                                     */
int ldpc_compact(ldpc_ctx *ctx, const uint8_t *d_flag, int64_t B, int32_t *d_index, int32_t *d_count,
                 void *stream);

/* Bit (un)packing between the reference's one-integer-per-bit labels and packed words.
 * elem_size: bytes per element of d_bits (1 = u8, 4 = i32, 8 = i64).                      */
int ldpc_pack_bits(ldpc_ctx *ctx, const void *d_bits /*[B][n]*/, int32_t elem_size, int64_t B,
                   uint64_t *d_words /*[B][ceil(n/64)]*/, void *stream);
int ldpc_unpack_bits(ldpc_ctx *ctx, const uint64_t *d_words, int64_t B, void *d_bits, int32_t elem_size,
                     void *stream);

/* ---------------------------------------------------------------------------------------
 * OSD (n = 128, k = 64 codes).  Frames are addressed as d_y[ d_index ? d_index[f] : f ].
 * If d_count is non-NULL the number of frames is min(*d_count, F) read ON THE DEVICE (so a
 * compaction can feed the OSD without a host round trip); F is then the capacity.
 * The entries of d_index are frame numbers of d_y and are NOT range-checked by default (the entry points do
 * not know how many frames d_y holds): a caller-made list must stay inside d_y; ldpc_compact /
 * ldpc_pipeline_run write only valid, ascending frame numbers.  Debug aid: ldpc_osd_params.y_frames.
 * ------------------------------------------------------------------------------------- */

/* Per-frame GF(2) elimination on the device: full_gf2elim, PB_OSD/pb_testing.py:231-266.
 * d_rows_in/out: [F][64][2] u64 (row r of frame f, columns 0..127); d_swaps: [F][64][2] u8
 * recorded (j, col) pairs; d_nswaps: [F] i32.                                              */
int ldpc_osd_ge(ldpc_ctx *ctx, const uint64_t *d_rows_in, int64_t F, uint64_t *d_rows_out, uint8_t *d_swaps,
                int32_t *d_nswaps, void *stream);

/* OSD front end: swapped_info + identify_mrb, PB_OSD/pb_testing.py:268-320
 * (reliability sort, column permutation of G, elimination, MRB/LRB bookkeeping).
 *   d_perm   [F][128] u8 : original bit index at primed position p (pi_1 o pi_2)
 *   d_parity [F][64] u64 : row r of P' in G' = [I | P']  (bit c = P'[r][c])
 *   d_nswaps [F] i32 (nullable)                                                            */
int ldpc_osd_front(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                   uint8_t *d_perm, uint64_t *d_parity, int32_t *d_nswaps, void *stream);

typedef struct ldpc_osd_params {
    int32_t order;       /* 0..3                                                          */
    int32_t algo;        /* LDPC_OSD_*                                                    */
    float snr_db;        /* PB-OSD: noise_variance = 10^(-snr/10), pb_testing.py:50-52     */
    float fs_beta;       /* FS-OSD beta, Main_FS_OSD.py:20 (0.1)                          */
    float fs_tau_e;      /* FS-OSD floor(d_min-1)/2 as the reference evaluates it (6.5)    */
    float fs_tau_psc;    /* FS-OSD tau_psc, FS_OSD/globalmap.py:50 (30)                    */
    int32_t fs_reference_quirk; /* 1: keep optimal_codeword un-updated on a tau_e hit (fs_testing.py:145) */
    int32_t reserved;    /* 0; cross-check switches: bit 0 = conventional order 2 through the table-driven scan instead of
                            the register-resident kernels, bit 3 = the register-resident kernel with v_readlane
                            pairing instead of the rotation-paired persistent one; PB-OSD: bit 0 = ldpc_osd_decode runs the front
                            end inside the first PB kernel (no workspace traffic; 4-5 % slower), bit 1 = every frame
                            through the sorted-chunk kernel from its first TEP, bit 2 = every frame through the
                            literal list replay                                                               */
    void *d_aux;         /* optional DEVICE [F][4] i32, PB-OSD statistics per frame: {frontier comparisons
                            (memory_sum, pb_testing.py:122), suc counter 1 (:138), suc counter 2 (:144),
                            stop reason 0 = none / 1 = promising rule (:129) / 2 = success rule (:145)} */
    int64_t y_frames;    /* 0 = d_index is trusted (default).  > 0 = debug bound: the number of frames d_y holds;
                            ldpc_osd_decode / ldpc_osd_search then run on a sanitised copy of d_index (entries outside
                            [0, y_frames) replaced by 0) and count the offenders, see ldpc_osd_index_errors.
                            It guards CALLER-MADE lists: ldpc_pipeline_run writes d_index itself (its compaction emits
                            valid, ascending frame numbers only), so there the field checks nothing that can be wrong, and
                            the pipeline's counting launch reads the list the library wrote              */
} ldpc_osd_params;

/* PB-OSD tuning of a context (ABI 4; rounds 1-3 read these from LDPC_PB_* environment variables on every call).
 * None of it changes a result -- TEP counts, stop reasons, winners and metrics are the same for every setting (the
 * tests run the kernels under several) -- only how the searches of pb_osd (PB_OSD/pb_testing.py:100-149) are cut into
 * chunks and when a long search moves from the one-wavefront kernel to the one-workgroup kernel:
 *   budget_s / budget_m / budget / budget_l / budget_xl
 *                  TEPs after which a search may be handed to the workgroup kernel, chosen on the device by the number of
 *                  frames still searching after the weight-1 head (a sixteenth of them: < 128 / < 448 / < 1400 / < 3000 /
 *                  more); >= 1.  Defaults 512 / 1024 / 4096 / 8192 / 24576 (measurements: DESIGN.md 3.4).
 *   t1, t2         target size of a frame's first / later chunks in the one-wavefront kernel, 32..832 (320, 600)
 *   t3             target chunk size of the workgroup kernel, 256..4096 (3072)
 *   late_min, late_maxlen, late_pct, late_div
 *                  the launch's tail: once the frames of a sub-list that have not finished are fewer than late_pct % of
 *                  the chip's wavefront slots (a sixteenth of 4096), a search leaves after budget / late_div TEPs, for
 *                  lists of more than late_min frames and sub-lists shorter than late_maxlen; late_div = 1 switches
 *                  it off.  Defaults 4608, 2^30, 20, 16.  With it, WHERE a search is handed on depends on timing.
 *   handoff_maxlen no search is handed on when the sub-list holds this many frames or more (default 2^30)
 * ldpc_ctx_set_pb_tuning validates (LDPC_E_ARG, nothing changed) and stores a copy; NULL restores the defaults.  It
 * applies to decode calls ISSUED afterwards (a captured graph keeps the values it was captured with); call it from
 * one thread, not concurrently with decode calls that should see a particular setting.                            */
typedef struct ldpc_pb_tuning {
    int32_t budget, budget_s, budget_m, budget_l, budget_xl;
    int32_t t1, t2, t3;
    int32_t late_min, late_maxlen, late_pct, late_div;
    int32_t handoff_maxlen;
} ldpc_pb_tuning;
int ldpc_ctx_get_pb_tuning(ldpc_ctx *ctx, ldpc_pb_tuning *out);
int ldpc_ctx_set_pb_tuning(ldpc_ctx *ctx, const ldpc_pb_tuning *tuning);

/* Pre-size OSD workspaces (640 B per frame: permutation + reduced parity rows) for up to max_frames
 * frames per ldpc_osd_decode call: ldpc_osd_reserve sizes the NULL stream's workspace immediately and every
 * other stream's when it is created.  Decode calls grow their stream's workspace on demand, which allocates
 * -- so before capturing calls on a stream into a hipGraph, size that stream's workspace with
 * ldpc_osd_reserve_stream: params = NULL sizes the front-end workspace only; with params it also sizes what
 * the search of (params->algo, params->order) needs (PB-OSD: frame lists, one 1536-byte record per frame, list-replay
 * areas, ~1.5 KiB per frame + 180 MB), so that the FIRST call on the stream may be the captured one.
 * ldpc_osd_release_stream frees the workspace of `stream` (idle, not capturing; graphs captured on it must
 * not be launched afterwards) -- for destroyed streams, whose handle the runtime may hand out again.     */
int ldpc_osd_reserve(ldpc_ctx *ctx, int64_t max_frames);
int ldpc_osd_reserve_stream(ldpc_ctx *ctx, int64_t max_frames, const ldpc_osd_params *params, void *stream);
int ldpc_osd_release_stream(ldpc_ctx *ctx, void *stream);

/* Ordered-statistics decoding of F frames (front end + search).
 *   d_cw      [F][2] u64  best codeword, ORIGINAL bit order
 *   d_metric  [F] f32     its weighted Hamming distance  sum_p (c_p xor h_p) |y_p|
 *   d_best    [F] i32     index of the winning TEP in the reference's table order
 *                         (conventional), or its rank in visit order (FS/PB; 0 = all-zero TEP)
 *   d_ntep    [F] i32     number of TEPs evaluated (FS: num_teps, fs_testing.py:141; PB: cost_tep_num
 *                         or N_max when no rule fired, pb_testing.py:152-155)
 * Any of d_metric/d_best/d_ntep may be NULL.
 * Concurrency: calls on different streams may overlap, whatever the algorithm (scratch is per stream,
 * see ldpc_ctx_create).                                                                       */
int ldpc_osd_decode(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                    const ldpc_osd_params *params, uint64_t *d_cw, float *d_metric, int32_t *d_best,
                    int32_t *d_ntep, void *stream);

/* Number of d_index entries found outside [0, y_frames) by calls that carried ldpc_osd_params.y_frames > 0, since the
 * context was created.  Synchronises the device (a debug aid, not for captured or timed regions).            */
int ldpc_osd_index_errors(ldpc_ctx *ctx, int64_t *count);

/* The search alone, on front-end results supplied by the caller (ldpc_osd_front, or any
 * (perm, P') pair: with perm = identity and d_y already in the primed order this is exactly
 * convention_osd_main / the fs_osd / pb_osd inner loops applied to (updated_inputs, reduced_G)). */
int ldpc_osd_search(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                    const uint8_t *d_perm /*[F][128]*/, const uint64_t *d_parity /*[F][64]*/,
                    const ldpc_osd_params *params, uint64_t *d_cw, float *d_metric, int32_t *d_best, int32_t *d_ntep,
                    void *stream);

/* One GIVEN test error pattern per frame on front-end results: one_tep_compare, FS_OSD/fs_testing.py:51-64
 * (re-encode the MRB hard decisions with the positions of d_mask[f] flipped; its (:58-62) Hamming and weighted
 * distances to the hard decisions of y').  Any weight.  Same LUT evaluation and float order as the searches.
 *   d_mask   [F] u64   bit p = flip MRB position p (primed order)
 *   d_cw     [F][2] u64 candidate codeword, ORIGINAL bit order;  d_metric [F] f32 (nullable);  d_hd [F] i32 (nullable) */
int ldpc_osd_tep_eval(ldpc_ctx *ctx, const float *d_y, const int32_t *d_index, const int32_t *d_count, int64_t F,
                      const uint8_t *d_perm, const uint64_t *d_parity, const uint64_t *d_mask, uint64_t *d_cw,
                      float *d_metric, int32_t *d_hd, void *stream);

/* OSD statistics against labels: d_counts[3] += {frames, frames_wrong, teps_total}.
 * (the success test of convention_osd.py:65-66 / pb_testing.py:158 / fs_testing.py:162)   */
int ldpc_osd_counts(ldpc_ctx *ctx, const uint64_t *d_cw, const uint64_t *d_label_bits, const int32_t *d_index,
                    const int32_t *d_count, const int32_t *d_ntep, int64_t F, int64_t *d_counts, void *stream);

/* ---------------------------------------------------------------------------------------
 * H-form OSD primitives for the DL-OSD stage (n = 128, m = k = 64, full-rank H):
 * DL_OSD_Testing_serial/ordered_statistics_decoding.py.  The trained networks of that stage stay
 * on the host; these entry points do the per-frame sort / elimination / candidate scan.
 * ------------------------------------------------------------------------------------- */

/* Host: the TEPs of one order pattern, osd.error_pattern_gen (:81-98): pattern[s] flips inside
 * segment [bounds[s], bounds[s+1]) of the 64 MRB positions, for s = 0..nseg-1; itertools.product
 * over the segments (leftmost slowest) of itertools.combinations inside each.
 * teps: [count][4] u8 {p0, p1, p2, weight} (ascending positions, unused = 0), or NULL to get the
 * count.  Patterns of total weight > 3 are LDPC_E_UNSUPPORTED.                               */
int64_t ldpc_hosd_pattern_teps(int32_t nseg, const int32_t *bounds, const int32_t *pattern, uint8_t *teps);

/* check_matrix_reorder + identify_mrb (:25-80, full_gf2elim :222-257) per frame:
 * positions sorted by ASCENDING |d_order_llr| (ties: lower index first), the columns of H gathered in
 * that order, Gauss-Jordan with the reference's pivot rule -> [I | M], MRB (last 64) sorted ascending.
 *   d_lri    [F][128] u8  : lri_p, original bit index at sorted position s                 (:34)
 *   d_uidx   [F][128] u8  : updated_index_order, sorted position at updated position p       (:67-68)
 *   d_M      [F][64] u64  : row r of updated_M (bit j = M[r][j], j = updated MRB position)   (:69)
 *   d_nswaps [F] i32 (nullable): recorded column exchanges, -1 = rank-deficient H (outputs undefined)
 * The original bit index at updated position p is d_lri[d_uidx[p]]; positions 0..63 are the LRB
 * (identity part, in the order the elimination left them), 64..127 the MRB.                  */
int ldpc_hosd_front(ldpc_ctx *ctx, const float *d_order_llr, int64_t F, uint8_t *d_lri, uint8_t *d_uidx,
                    uint64_t *d_M, int32_t *d_nswaps, void *stream);

/* Block minima of sliding_osd / acquire_min (:153-186): for every frame and every TEP block b
 * (TEPs d_teps[d_block_off[b] .. d_block_off[b+1]) ), the smallest weighted Hamming distance of
 *   candidate = [ M . (e xor mrb0), e xor mrb0 ],  mrb0 = hard(d_order_llr) on the MRB     (:154-156,186-187)
 * to the hard decisions of d_metric_llr, weights |d_metric_llr|                              (:159-160,180-182).
 * Float order of the metric: positions in updated order, bytes of 8, each byte summed ascending
 * from 0, the 16 byte sums added ascending (oracle/np_oracle.py hosd_cost).
 *   d_block_min [F][nblk] f32, d_block_arg [F][nblk] i32 (nullable; index into d_teps of the first minimum)
 *   d_truth  [F] f32 (nullable, needs d_label_bits [F][2] u64): the metric of the label      (:181-183)
 *   d_cw     [F][2] u64 (nullable): best candidate over all blocks, ORIGINAL bit order
 *   d_metric [F] f32, d_best [F] i32 (nullable): its metric and index into d_teps (first minimum)
 * d_teps: DEVICE [ntep][4] u8 as ldpc_hosd_pattern_teps writes them; d_block_off: DEVICE [nblk+1] i32. */
int ldpc_hosd_search(ldpc_ctx *ctx, const float *d_order_llr, const float *d_metric_llr, int64_t F,
                     const uint8_t *d_lri, const uint8_t *d_uidx, const uint64_t *d_M, const uint8_t *d_teps,
                     const int32_t *d_block_off, int32_t nblk, const uint64_t *d_label_bits, float *d_block_min,
                     int32_t *d_block_arg, float *d_truth, uint64_t *d_cw, float *d_metric, int32_t *d_best,
                     void *stream);

/* ---------------------------------------------------------------------------------------
 * One batch through the whole path with a single host call: the body of the reference drivers'
 * per-batch loops (ldpc_128_testing.py:117-131 then pb_testing.py / fs_testing.py per failed frame):
 *   NMS-T -> error counters -> failed-frame compaction -> OSD (front end + search) on the failures
 *   -> OSD counters.
 * The results of the sequence ldpc_nms_decode, ldpc_eval_counts, ldpc_compact, ldpc_osd_front,
 * ldpc_osd_search, ldpc_osd_counts on one stream (the counters ride in the compaction's counting pass), without a host round trip in between (the OSD
 * kernels read the failure count on the device).  Nullable members switch their stage off.
 * With timing_slot >= 0 the library brackets the three hot kernels with its own HIP events on
 * `stream`; after synchronising, ldpc_pipeline_timing returns their durations.
 * ------------------------------------------------------------------------------------- */
typedef struct ldpc_pipeline {
    const float *d_llr;            /* [B][n]                                               */
    int64_t B;
    int32_t T, nms_kernel;
    const float *alpha;            /* host [T]                                             */
    float w_in, w_out;
    float *d_soft;                 /* [B][n] nullable                                      */
    uint64_t *d_hard;              /* [B][n/64]                                            */
    uint8_t *d_fail;               /* [B]                                                  */
    const uint64_t *d_label_bits;  /* [B][n/64] nullable: no counters                      */
    int64_t *d_nms_counts;         /* [5] accumulated, nullable                            */
    int32_t osd_enable, timing_slot;
    ldpc_osd_params osd;
    int32_t *d_index, *d_count;    /* [B], [1]                                             */
    uint8_t *d_perm;               /* [B][128] nullable: both NULL = ldpc_osd_decode on the     */
    uint64_t *d_parity;            /* [B][64]  context workspace (no per-kernel OSD timing)     */
    uint64_t *d_cw;                /* [B][2]                                               */
    float *d_metric;               /* [B] nullable                                         */
    int32_t *d_best, *d_ntep;      /* [B] nullable / [B]                                   */
    int64_t *d_osd_counts;         /* [3] accumulated, nullable                            */
} ldpc_pipeline;

#define LDPC_TIMING_SLOTS 64
int ldpc_pipeline_run(ldpc_ctx *ctx, const ldpc_pipeline *p, void *stream);
/* ms[3] = {NMS, OSD front end, OSD search} of the run that used `slot` (call after the stream is idle);
 * when d_perm/d_parity were NULL the OSD ran through ldpc_osd_decode: ms[1] = 0, ms[2] = whole OSD stage */
int ldpc_pipeline_timing(ldpc_ctx *ctx, int32_t slot, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_OSD_H */
