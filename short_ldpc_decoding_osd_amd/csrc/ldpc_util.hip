// Small streaming kernels around the decoders: error statistics, failed-frame compaction,
// bit (un)packing.  All are HBM-bound byte movers; they read/write every byte once with
// coalesced accesses and reduce in registers / LDS before touching an atomic.
//
// Reference (paths relative to LDPC_128/): Ldpc_128_testing/ms_test.py:36-54 (get_eval),
// :51 (failure index), Ldpc_128_testing/data_generating.py / read_TFdata.py (labels are one
// int64 per bit).
#include "ldpc_internal.h"

namespace ldpc {

// ---------------------------------------------------------------------------------------
// get_eval counters.  One thread per frame; per-wave reduction by shuffles, one atomic set
// per wave.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void eval_counts_kernel(const unsigned long long *__restrict__ hard,
                                                          const unsigned long long *__restrict__ label,
                                                          const unsigned char *__restrict__ fail, long long B,
                                                          int words, unsigned long long *__restrict__ counts)
{
    __shared__ unsigned long long part[4][5];
    unsigned long long ferr = 0, berr = 0, und = 0, sf = 0, cnt = 0;
    for (long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x; f < B; f += (long long)gridDim.x * blockDim.x) {
        int e = 0;
        for (int w = 0; w < words; ++w) e += __popcll(hard[f * words + w] ^ label[f * words + w]);
        const int bad = fail ? fail[f] != 0 : 0;
        cnt += 1; berr += e; ferr += e != 0; sf += bad; und += (fail && !bad && e != 0);
    }
    cnt = wave_sum(cnt); ferr = wave_sum(ferr); berr = wave_sum(berr); und = wave_sum(und); sf = wave_sum(sf);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[wave][0] = cnt; part[wave][1] = ferr; part[wave][2] = berr; part[wave][3] = und; part[wave][4] = sf;
    }
    __syncthreads();
    // one atomic per counter per block (a few hundred in all): contended same-line atomics
    // serialise at the memory side, so keep them rare
    if (threadIdx.x < 5)
        atomicAdd(&counts[threadIdx.x], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------
// Ordered compaction in ONE launch and with NO scratch memory (so calls on different streams of one
// context cannot disturb each other).  The flags are cut into at most kMaxSeg contiguous segments, one
// per block.  A block first counts the set flags BEFORE its segment itself -- 16 flags per load, the flag
// array is L2-resident (1 B per frame: the last of 64 blocks re-reads 126 KiB at B = 131 072) -- and then
// scans its own segment 2048 or 8192 flags at a time, scattering ascending frame numbers.  The redundant prefix
// reads grow with B / 2 per block, i.e. stay ~1 % of the NMS time of the same batch at any B.
// The block that owns the last segment writes the total.
// ---------------------------------------------------------------------------------------
constexpr int kMaxSeg = 256;   // segments (= blocks) per launch

__device__ __forceinline__ int nonzero_bytes(unsigned long long v)
{
    const unsigned long long k7 = 0x7F7F7F7F7F7F7F7Full;
    return __popcll((((v & k7) + k7) | v) & ~k7);
}

// FPT consecutive flags of one thread as a bit mask
template <int FPT>
__device__ __forceinline__ unsigned flags_of(const unsigned char *flag, long long base, long long B, bool aligned)
{
    unsigned bits = 0;
    if (FPT == 8 && aligned && base + 8 <= B) {
        unsigned long long v = *reinterpret_cast<const unsigned long long *>(flag + base);
#pragma unroll
        for (int i = 0; i < 8; ++i) bits |= (unsigned)(((v >> (8 * i)) & 0xFF) != 0) << i;
    } else {
#pragma unroll
        for (int i = 0; i < FPT; ++i) if (base + i < B) bits |= (unsigned)(flag[base + i] != 0) << i;
    }
    return bits;
}

constexpr int kCompactThreads = 1024, kCompactWaves = kCompactThreads / 64;

__device__ __forceinline__ int block_sum(int v, int *wsum /*[kCompactWaves] shared*/)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();                         // wsum may still be read from an earlier use
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < kCompactWaves; ++w) t += wsum[w];
    return t;
}

// EVAL: the get_eval counters of the same frames ride along (ldpc_pipeline_run: the flags are the
// syndrome flags either way), so the pipeline needs neither eval_counts_kernel nor a second pass.
// 1024 threads per block: the prefix count of the LAST block is a chain of (flags before it) / 16 KiB dependent
// trips of ~0.4 us each (the flags were written by other XCDs: L2 misses) -- 8 trips at B = 131 072.
// FPT flags per thread and step: 2 for batches up to 524 288 frames (2048 flags per block), 8 beyond.
template <bool EVAL, int FPT>
__global__ __launch_bounds__(kCompactThreads) void compact_kernel(const unsigned char *__restrict__ flag, long long B, long long seg,
                                                                  int *__restrict__ index, int *__restrict__ count,
                                                                  const unsigned long long *__restrict__ hard,
                                                                  const unsigned long long *__restrict__ label, int words,
                                                                  unsigned long long *__restrict__ counts)
{
    constexpr int NT = kCompactThreads, kStep = NT * FPT;
    __shared__ int wsum[kCompactWaves];
    __shared__ unsigned long long epart[kCompactWaves][4];
    const long long lo = (long long)blockIdx.x * seg;
    const long long hi = lo + seg < B ? lo + seg : B;
    const bool al16 = (reinterpret_cast<unsigned long long>(flag) & 15) == 0;
    // ---- set flags before this segment (lo is a multiple of 16): independent loads, four in flight per thread
    int acc = 0;
    if (al16) {
        long long i = (long long)threadIdx.x * 16;
        for (; i + 3LL * NT * 16 < lo; i += 4LL * NT * 16) {
            const uint4 v0 = *reinterpret_cast<const uint4 *>(flag + i), v1 = *reinterpret_cast<const uint4 *>(flag + i + NT * 16LL);
            const uint4 v2 = *reinterpret_cast<const uint4 *>(flag + i + 2LL * NT * 16), v3 = *reinterpret_cast<const uint4 *>(flag + i + 3LL * NT * 16);
            acc += nonzero_bytes(((unsigned long long)v0.y << 32) | v0.x) + nonzero_bytes(((unsigned long long)v0.w << 32) | v0.z);
            acc += nonzero_bytes(((unsigned long long)v1.y << 32) | v1.x) + nonzero_bytes(((unsigned long long)v1.w << 32) | v1.z);
            acc += nonzero_bytes(((unsigned long long)v2.y << 32) | v2.x) + nonzero_bytes(((unsigned long long)v2.w << 32) | v2.z);
            acc += nonzero_bytes(((unsigned long long)v3.y << 32) | v3.x) + nonzero_bytes(((unsigned long long)v3.w << 32) | v3.z);
        }
        for (; i < lo; i += NT * 16LL) {
            const uint4 v = *reinterpret_cast<const uint4 *>(flag + i);
            acc += nonzero_bytes(((unsigned long long)v.y << 32) | v.x) + nonzero_bytes(((unsigned long long)v.w << 32) | v.z);
        }
    } else {
        for (long long i = threadIdx.x; i < lo; i += NT) acc += flag[i] != 0;
    }
    int base = block_sum(acc, wsum);
    // ---- own segment
    unsigned long long ferr = 0, berr = 0, und = 0, cnt = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long c0 = lo; c0 < hi; c0 += kStep) {
        const long long fb = c0 + threadIdx.x * FPT;
        const unsigned bits = fb < hi ? flags_of<FPT>(flag, fb, hi, al16) : 0;
        const int mine = __popc(bits);
        if constexpr (EVAL) {   // frame c0 + q NT + thread: coalesced loads of the hard words and labels
            if (words == 2) {   // n = 128: every load of the step is issued before the first is used (one memory latency)
                ulonglong2 hv[FPT], lv[FPT];
                unsigned char fv[FPT];
#pragma unroll
                for (int q = 0; q < FPT; ++q) {
                    const long long f = c0 + q * NT + threadIdx.x, fc = f < hi ? f : lo;
                    hv[q] = *reinterpret_cast<const ulonglong2 *>(hard + fc * 2);
                    lv[q] = *reinterpret_cast<const ulonglong2 *>(label + fc * 2);
                    fv[q] = flag[fc];
                }
#pragma unroll
                for (int q = 0; q < FPT; ++q) {
                    if (c0 + q * NT + threadIdx.x < hi) {
                        const int e = __popcll(hv[q].x ^ lv[q].x) + __popcll(hv[q].y ^ lv[q].y);
                        cnt += 1; berr += e; ferr += e != 0; und += (fv[q] == 0 && e != 0);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < FPT; ++q) {
                    const long long f = c0 + q * NT + threadIdx.x;
                    if (f < hi) {
                        int e = 0;
                        for (int w = 0; w < words; ++w) e += __popcll(hard[f * words + w] ^ label[f * words + w]);
                        cnt += 1; berr += e; ferr += e != 0; und += (flag[f] == 0 && e != 0);
                    }
                }
            }
        }
        int incl = mine;   // exclusive scan over the threads: in-wave inclusive scan, then wave offsets
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        __syncthreads();
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kCompactWaves; ++w) { woff += w < wave ? wsum[w] : 0; tot += wsum[w]; }
        int pos = base + woff + incl - mine;
#pragma unroll
        for (int i = 0; i < FPT; ++i)
            if ((bits >> i) & 1) index[pos++] = (int)(fb + i);
        base += tot;
    }
    if (hi == B && threadIdx.x == 0) *count = base;
    if constexpr (EVAL) {   // counts[] = {frames, frame_err, bit_err, undetected, synd_fail}
        cnt = wave_sum(cnt); ferr = wave_sum(ferr); berr = wave_sum(berr); und = wave_sum(und);
        if (lane == 0) { epart[wave][0] = cnt; epart[wave][1] = ferr; epart[wave][2] = berr; epart[wave][3] = und; }
        const int before = block_sum(acc, wsum);   // (also the barrier that publishes epart)
        if (threadIdx.x < 4) {
            unsigned long long t = 0;
#pragma unroll
            for (int w = 0; w < kCompactWaves; ++w) t += epart[w][threadIdx.x];
            atomicAdd(&counts[threadIdx.x], t);
        }
        if (threadIdx.x == 4) atomicAdd(&counts[4], (unsigned long long)(base - before));
    }
}

static void compact_geometry(int64_t B, int fpt, unsigned *blocks, long long *seg)
{
    const int64_t step = (int64_t)kCompactThreads * fpt;
    const int64_t chunks = (B + step - 1) / step;
    const int64_t per = (chunks + kMaxSeg - 1) / kMaxSeg;   // steps per segment
    *seg = (long long)per * step;
    *blocks = (unsigned)((B + *seg - 1) / *seg);
}

template <bool EVAL>
static void launch_compact(const uint8_t *d_flag, int64_t B, int32_t *d_index, int32_t *d_count, const unsigned long long *hard,
                           const unsigned long long *label, int words, unsigned long long *counts, hipStream_t st)
{
    unsigned blocks; long long seg;
    if (B <= (int64_t)kMaxSeg * kCompactThreads * 2) {
        compact_geometry(B, 2, &blocks, &seg);
        hipLaunchKernelGGL((compact_kernel<EVAL, 2>), dim3(blocks), dim3(kCompactThreads), 0, st, d_flag, (long long)B, seg, d_index, d_count, hard, label, words, counts);
    } else {
        compact_geometry(B, 8, &blocks, &seg);
        hipLaunchKernelGGL((compact_kernel<EVAL, 8>), dim3(blocks), dim3(kCompactThreads), 0, st, d_flag, (long long)B, seg, d_index, d_count, hard, label, words, counts);
    }
}

// ---------------------------------------------------------------------------------------
// bit packing: one wavefront packs 64 bits per step with a ballot
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_bits_kernel(const T *__restrict__ bits, long long B, int n, int words,
                                                        unsigned long long *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long total = B * words;
    for (long long t = wave; t < total; t += (long long)gridDim.x * 4) {
        const long long f = t / words;
        const int v = (int)(t % words) * 64 + lane;
        const unsigned long long m = __ballot(v < n && (bits[f * n + v] & 1));
        if (lane == 0) out[t] = m;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_bits_kernel(const unsigned long long *__restrict__ in, long long B,
                                                          int n, int words, T *__restrict__ bits)
{
    const long long total = B * n;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long f = t / n;
        const int v = (int)(t % n);
        bits[t] = (T)((in[f * words + (v >> 6)] >> (v & 63)) & 1);
    }
}

static unsigned grid_for(long long items, int per_block, unsigned cap = 4096)
{
    long long g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (unsigned)(g < cap ? g : cap);
}

}  // namespace ldpc

using namespace ldpc;

extern "C" {

int ldpc_eval_counts(ldpc_ctx *ctx, const uint64_t *d_hard, const uint64_t *d_label, const uint8_t *d_fail,
                     int64_t B, int64_t *d_counts, void *stream)
{
    if (!ctx || !d_hard || !d_label || !d_counts || B < 0) return fail(LDPC_E_ARG, "ldpc_eval_counts: bad arguments");
    if (B == 0) return LDPC_OK;
    const int words = (ctx->code.n + 63) / 64;
    hipLaunchKernelGGL(eval_counts_kernel, dim3(grid_for(B, 1024, 128)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long *>(d_hard),
                       reinterpret_cast<const unsigned long long *>(d_label), d_fail, (long long)B, words,
                       reinterpret_cast<unsigned long long *>(d_counts));
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

}  // extern "C"

namespace ldpc {
// (a kernel, not hipMemsetAsync: a memset node captured into a hipGraph did not run on later replays, ROCm 7.2)
__global__ void zero_word_kernel(int32_t *p) { *p = 0; }
}  // namespace ldpc

extern "C" {

int ldpc_compact(ldpc_ctx *ctx, const uint8_t *d_flag, int64_t B, int32_t *d_index, int32_t *d_count, void *stream)
{
    if (!ctx || !d_count || B < 0 || B > 0x7FFFFFFFLL || (B > 0 && (!d_flag || !d_index)))
        return fail(LDPC_E_ARG, "ldpc_compact: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) { hipLaunchKernelGGL(ldpc::zero_word_kernel, dim3(1), dim3(1), 0, st, d_count); LDPC_HIP(hipGetLastError()); return LDPC_OK; }
    launch_compact<false>(d_flag, B, d_index, d_count, nullptr, nullptr, 0, nullptr, st);
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

}  // extern "C"

namespace ldpc {

// ldpc_eval_counts + ldpc_compact on the same flags in one launch (ldpc_pipeline_run)
int eval_and_compact(ldpc_ctx *ctx, const uint64_t *d_hard, const uint64_t *d_label, const uint8_t *d_fail, int64_t B,
                     int64_t *d_counts, int32_t *d_index, int32_t *d_count, hipStream_t st)
{
    if (!ctx || !d_hard || !d_label || !d_fail || !d_counts || !d_index || !d_count || B <= 0 || B > 0x7FFFFFFFLL)
        return fail(LDPC_E_ARG, "eval_and_compact: bad arguments");
    launch_compact<true>(d_fail, B, d_index, d_count, reinterpret_cast<const unsigned long long *>(d_hard),
                         reinterpret_cast<const unsigned long long *>(d_label), (ctx->code.n + 63) / 64,
                         reinterpret_cast<unsigned long long *>(d_counts), st);
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

}  // namespace ldpc

extern "C" {

int ldpc_pack_bits(ldpc_ctx *ctx, const void *d_bits, int32_t elem_size, int64_t B, uint64_t *d_words, void *stream)
{
    if (!ctx || !d_bits || !d_words || B < 0) return fail(LDPC_E_ARG, "ldpc_pack_bits: bad arguments");
    if (B == 0) return LDPC_OK;
    const int n = ctx->code.n, words = (n + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    auto *out = reinterpret_cast<unsigned long long *>(d_words);
    const dim3 g(grid_for(B * words, 4, 8192)), b(256);
    switch (elem_size) {
    case 1: hipLaunchKernelGGL(pack_bits_kernel<unsigned char>, g, b, 0, st, (const unsigned char *)d_bits, (long long)B, n, words, out); break;
    case 4: hipLaunchKernelGGL(pack_bits_kernel<int>, g, b, 0, st, (const int *)d_bits, (long long)B, n, words, out); break;
    case 8: hipLaunchKernelGGL(pack_bits_kernel<long long>, g, b, 0, st, (const long long *)d_bits, (long long)B, n, words, out); break;
    default: return fail(LDPC_E_ARG, "ldpc_pack_bits: elem_size %d (want 1, 4 or 8)", elem_size);
    }
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

int ldpc_unpack_bits(ldpc_ctx *ctx, const uint64_t *d_words, int64_t B, void *d_bits, int32_t elem_size, void *stream)
{
    if (!ctx || !d_bits || !d_words || B < 0) return fail(LDPC_E_ARG, "ldpc_unpack_bits: bad arguments");
    if (B == 0) return LDPC_OK;
    const int n = ctx->code.n, words = (n + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    auto *in = reinterpret_cast<const unsigned long long *>(d_words);
    const dim3 g(grid_for(B * n, 256, 8192)), b(256);
    switch (elem_size) {
    case 1: hipLaunchKernelGGL(unpack_bits_kernel<unsigned char>, g, b, 0, st, in, (long long)B, n, words, (unsigned char *)d_bits); break;
    case 4: hipLaunchKernelGGL(unpack_bits_kernel<int>, g, b, 0, st, in, (long long)B, n, words, (int *)d_bits); break;
    case 8: hipLaunchKernelGGL(unpack_bits_kernel<long long>, g, b, 0, st, in, (long long)B, n, words, (long long *)d_bits); break;
    default: return fail(LDPC_E_ARG, "ldpc_unpack_bits: elem_size %d (want 1, 4 or 8)", elem_size);
    }
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

}  // extern "C"
