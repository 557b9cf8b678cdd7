// Normalised min-sum belief propagation kernels for gfx950 (MI355X).
//
// Reference math (paths relative to LDPC_128/ of the reference): Ldpc_128_testing/ms_test.py
//   compute_vc :124-137, compute_cv2 :180-210, marginalize :220-228, fixed-T loop :106-121,
//   hard decision + syndrome of get_eval :39,:45.   Per iteration:
//     tot[v]  = (sum of cv over the checks of v, ascending check index) + w_in * y[v]
//     vc      = tot[v] - cv[c,v]
//     S[c]    = prod sign(vc)  (sign(0) = 0),   m1 <= m2 = two smallest min(|vc|, 1e30)
//     cv[c,v] = (alpha * ((|vc| > m1) ? m1 : m2)) * S[c] * sign(vc)
//     out[v]  = (sum of the new cv, same order) + w_out * y[v]
//   The summation order is the one a sequential dense reduce_sum produces; both kernels and
//   the CPU oracle follow it, so soft outputs agree bit for bit (compiled -ffp-contract=off).
//
// Two kernels:
//   nms_generic : any Tanner graph; one frame per wavefront, edge messages + variable totals
//                 staged in LDS, lanes sweep variables, then checks.
//   nms_qc16    : H made of 16x16 circulants in the CCSDS (128,64) arrangement.  A frame
//                 occupies ONE 16-lane DPP row (4 frames per wavefront): lane j of the row is
//                 check (br, j) for the 4 block rows and variable (bc, j) for the 8 block
//                 columns, so every circulant shift is a `row_ror` DPP rotation fused into the
//                 add/sub that consumes it.  All 32 edge messages per lane live in VGPRs;
//                 the iteration loop touches neither LDS nor memory.  No MFMA: there is no
//                 contraction here, the kernel is VALU-issue bound.
#include "ldpc_internal.h"

namespace ldpc {

// =======================================================================================
// generic kernel
// =======================================================================================
// magnitude of `mag` with the sign bit of `s`: copysign lowers to one v_bfi_b32
__device__ __forceinline__ unsigned sign_insert(unsigned mag, unsigned s)
{
    return __float_as_uint(__builtin_copysignf(__uint_as_float(mag), __uint_as_float(s)));
}

__device__ __forceinline__ void wave_lds_fence()
{
    // one wavefront = one frame: LDS operations of a wave execute in program order, so only
    // the compiler has to be kept from reordering across the phase boundary.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// (index / count / rows: the failed-frame form, ldpc_nms_traj_rows -- frame f of the launch is llr[index[f]], f < min(*count, B),
//  and the only output is rows[f][0..T][n]: row 0 the channel values, row t the posterior after iteration t)
__global__ __launch_bounds__(256) void nms_generic_kernel(
    const float *__restrict__ llr, long long B, int T, AlphaArg alpha, float w_in, float w_out,
    float *__restrict__ soft, float *__restrict__ traj, unsigned long long *__restrict__ hard,
    unsigned char *__restrict__ fail, const int *__restrict__ chk_ptr, const int *__restrict__ chk_var,
    const int *__restrict__ var_ptr, const int *__restrict__ var_edge, int n, int m, int E,
    const int *__restrict__ index, const int *__restrict__ count, float *__restrict__ rows)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *cv = smem + (size_t)wave * (E + 2 * n);
    float *tot = cv + E;
    float *yb = tot + n;
    const int words = (n + 63) >> 6;

    if (count) { const long long c = *count; B = c < B ? c : B; }
    for (long long f = (long long)blockIdx.x * 4 + wave; f < B; f += (long long)gridDim.x * 4) {
        const long long src = index ? index[f] : f;
        for (int v = lane; v < n; v += 64) {
            float y = llr[src * n + v];
            yb[v] = y;
            tot[v] = y;  // T == 0: the posterior is the channel value
            if (rows) rows[(f * (T + 1)) * n + v] = y;
        }
        for (int e = lane; e < E; e += 64) cv[e] = 0.0f;
        wave_lds_fence();
        for (int it = 0; it < T; ++it) {
            const float a_it = alpha.a[it];
            for (int v = lane; v < n; v += 64) {
                float acc = 0.0f;
                for (int q = var_ptr[v]; q < var_ptr[v + 1]; ++q) acc = acc + cv[var_edge[q]];
                tot[v] = acc + yb[v] * w_in;
            }
            wave_lds_fence();
            for (int c = lane; c < m; c += 64) {
                const int e0 = chk_ptr[c], e1 = chk_ptr[c + 1];
                float m1 = __builtin_inff(), m2 = __builtin_inff();
                unsigned sx = 0;
                for (int e = e0; e < e1; ++e) {
                    float vc = tot[chk_var[e]] - cv[e];
                    cv[e] = vc;
                    float a = __builtin_fabsf(vc);
                    m2 = __builtin_fminf(m2, __builtin_fmaxf(m1, a));
                    m1 = __builtin_fminf(m1, a);
                    sx ^= __float_as_uint(vc);
                }
                // clamping the order statistics == clamping every |vc| (monotone); a zero
                // minimum means sign(0) = 0 wipes the whole check row (ms_test.py:187-191)
                const float m1s = a_it * __builtin_fminf(m1, 1e30f);
                const float m2s = (m1 == 0.0f) ? 0.0f : a_it * __builtin_fminf(m2, 1e30f);
                for (int e = e0; e < e1; ++e) {
                    float vc = cv[e];
                    float mag = (__builtin_fabsf(vc) > m1) ? m1s : m2s;
                    cv[e] = __uint_as_float(sign_insert(__float_as_uint(mag), sx ^ __float_as_uint(vc)));
                }
            }
            wave_lds_fence();
            if (traj || rows || it == T - 1) {
                for (int v = lane; v < n; v += 64) {
                    float acc = 0.0f;
                    for (int q = var_ptr[v]; q < var_ptr[v + 1]; ++q) acc = acc + cv[var_edge[q]];
                    float o = acc + w_out * yb[v];
                    if (traj) traj[((long long)it * B + f) * n + v] = o;
                    if (rows) rows[(f * (T + 1) + it + 1) * n + v] = o;
                    if (it == T - 1) tot[v] = o;
                }
                wave_lds_fence();
            }
        }
        // outputs: posterior, packed hard decision (soft > 0 ? 0 : 1), syndrome flag
        if (soft)
            for (int v = lane; v < n; v += 64) soft[f * n + v] = tot[v];
        if (hard)
            for (int w = 0; w < words; ++w) {
                int v = w * 64 + lane;
                unsigned long long bits = __ballot(v < n && !(tot[v] > 0.0f));
                if (lane == 0) hard[f * words + w] = bits;
            }
        if (fail) {
            int bad = 0;
            for (int c = lane; c < m; c += 64) {
                int par = 0;
                for (int e = chk_ptr[c]; e < chk_ptr[c + 1]; ++e) par ^= !(tot[chk_var[e]] > 0.0f);
                bad |= par;
            }
            unsigned long long anybad = __ballot(bad);
            if (lane == 0) fail[f] = anybad != 0;
        }
        wave_lds_fence();
    }
}

// =======================================================================================
// QC-16 kernel (CCSDS (128,64) arrangement)
// =======================================================================================
struct Term {
    int br, bc, s;
};
// check (br, i) -- variable (bc, (i + s) mod 16); check-major, 8 terms per block row.
// Same table as kCcsds128 in ldpc_host.cpp, which verifies the loaded H against it.
constexpr Term kTerms[32] = {
    {0, 0, 0}, {0, 0, 7}, {0, 1, 2}, {0, 2, 14}, {0, 3, 6}, {0, 5, 0}, {0, 6, 13}, {0, 7, 0},
    {1, 0, 6}, {1, 1, 0}, {1, 1, 15}, {1, 2, 0}, {1, 3, 1}, {1, 4, 0}, {1, 6, 0}, {1, 7, 7},
    {2, 0, 4}, {2, 1, 1}, {2, 2, 0}, {2, 2, 15}, {2, 3, 14}, {2, 4, 11}, {2, 5, 0}, {2, 7, 3},
    {3, 0, 0}, {3, 1, 1}, {3, 2, 9}, {3, 3, 0}, {3, 3, 13}, {3, 4, 14}, {3, 5, 1}, {3, 6, 0},
};

// terms of block column BC in ascending block row (= ascending check index up to the
// ordering inside a doubled circulant, resolved per lane below)
struct VarPlan {
    int cnt;
    int e[5];
};
__host__ __device__ constexpr VarPlan var_plan(int bc)
{
    VarPlan p{0, {-1, -1, -1, -1, -1}};
    for (int e = 0; e < 32; ++e)
        if (kTerms[e].bc == bc) p.e[p.cnt++] = e;
    return p;
}

// result[j] = x[(j - S) mod 16] inside each 16-lane row.  UP says whether the hardware's
// row_ror:n moves data towards higher lane numbers (probed once per context).
template <int S, bool UP>
__device__ __forceinline__ int rot_i(int x)
{
    constexpr int s = ((S % 16) + 16) % 16;
    if constexpr (s == 0) return x;
    else {
        constexpr int amount = UP ? s : 16 - s;
        return __builtin_amdgcn_update_dpp(0, x, 0x120 + amount, 0xF, 0xF, true);
    }
}
template <int S, bool UP>
__device__ __forceinline__ float rot_f(float x)
{
    return __int_as_float(rot_i<S, UP>(__float_as_int(x)));
}

// Ordered sum of the check->variable messages of block column BC for variable (BC, j).
template <int BC, int POS, bool UP>
__device__ __forceinline__ float var_sum_from(const float (&cv)[32], int j, float acc)
{
    constexpr VarPlan P = var_plan(BC);
    if constexpr (POS >= P.cnt) return acc;
    else {
        constexpr int ea = P.e[POS];
        constexpr bool paired = (POS + 1 < P.cnt) && (kTerms[P.e[POS + 1 < 5 ? POS + 1 : 4]].br == kTerms[ea].br);
        const float ta = rot_f<kTerms[ea].s, UP>(cv[ea]);
        if constexpr (paired) {
            // two circulants in one block: the variable's two checks are (br, j - sa) and
            // (br, j - sb); the smaller check index is summed first
            constexpr int eb = P.e[POS + 1];
            const float tb = rot_f<kTerms[eb].s, UP>(cv[eb]);
            if constexpr (POS == 0) return var_sum_from<BC, POS + 2, UP>(cv, j, ta + tb);
            else {
                const bool a_first = ((j - kTerms[ea].s) & 15) < ((j - kTerms[eb].s) & 15);
                const float first = a_first ? ta : tb, second = a_first ? tb : ta;
                return var_sum_from<BC, POS + 2, UP>(cv, j, (acc + first) + second);
            }
        } else {
            if constexpr (POS == 0) return var_sum_from<BC, POS + 1, UP>(cv, j, ta);
            else return var_sum_from<BC, POS + 1, UP>(cv, j, acc + ta);
        }
    }
}

template <bool UP, int BC = 0>
__device__ __forceinline__ void var_sums(const float (&cv)[32], int j, float (&S)[8])
{
    if constexpr (BC < 8) {
        S[BC] = var_sum_from<BC, 0, UP>(cv, j, 0.0f);
        var_sums<UP, BC + 1>(cv, j, S);
    }
}

// One block row of check nodes: lane j is check (BR, j).
template <int BR, bool UP>
__device__ __forceinline__ void check_row(float (&cv)[32], const float (&tot)[8], float a_it, unsigned signv)
{
    float vc[8];
    // vc = tot[var] - cv ; variable (bc, j + s) sits s lanes above: rotate down by s
#define LDPC_VC(t) vc[t] = rot_f<16 - kTerms[BR * 8 + t].s, UP>(tot[kTerms[BR * 8 + t].bc]) - cv[BR * 8 + t];
    LDPC_VC(0) LDPC_VC(1) LDPC_VC(2) LDPC_VC(3) LDPC_VC(4) LDPC_VC(5) LDPC_VC(6) LDPC_VC(7)
#undef LDPC_VC
    // "minimum of the OTHER seven magnitudes" per edge == ((|vc| > m1) ? m1 : m2) of ms_test.py:200-206,
    // ties included, computed as a shared tree of v_min3: 4 pair minima (the 1e30 clip of :196 rides along as
    // their third operand -- every output contains at least one clipped node, so the clip reaches all of them),
    // 2 quad minima, then one v_min3 per edge = 14 min-class instructions per check instead of the 30 of a
    // running (min1, min2) + compare + select.  min/cmp/cndmask/DPP issue at half the rate of add/mul/xor on
    // gfx950 (profiles/r01/ubench_valu_issue2.txt), so the per-edge scaling is done with a v_mul instead.
    float a[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) a[t] = __builtin_fabsf(vc[t]);
    const float p01 = __builtin_fminf(__builtin_fminf(a[0], a[1]), 1e30f), p23 = __builtin_fminf(__builtin_fminf(a[2], a[3]), 1e30f);
    const float p45 = __builtin_fminf(__builtin_fminf(a[4], a[5]), 1e30f), p67 = __builtin_fminf(__builtin_fminf(a[6], a[7]), 1e30f);
    const float q03 = __builtin_fminf(p01, p23), q47 = __builtin_fminf(p45, p67);
    // sign(0) = 0 wipes the whole check row (:187-191): a zero minimum zeroes the scale factor
    const float aeff = (__builtin_fminf(q03, q47) == 0.0f) ? 0.0f : a_it;
    float o[8];
    o[0] = __builtin_fminf(__builtin_fminf(a[1], p23), q47);
    o[1] = __builtin_fminf(__builtin_fminf(a[0], p23), q47);
    o[2] = __builtin_fminf(__builtin_fminf(a[3], p01), q47);
    o[3] = __builtin_fminf(__builtin_fminf(a[2], p01), q47);
    o[4] = __builtin_fminf(__builtin_fminf(a[5], p67), q03);
    o[5] = __builtin_fminf(__builtin_fminf(a[4], p67), q03);
    o[6] = __builtin_fminf(__builtin_fminf(a[7], p45), q03);
    o[7] = __builtin_fminf(__builtin_fminf(a[6], p45), q03);
    // parity of the eight sign bits: three 3-input XORs (v_bitop3_b32 0x96) and one 2-input instead of seven XORs
    const unsigned x012 = __builtin_amdgcn_bitop3_b32(__float_as_uint(vc[0]), __float_as_uint(vc[1]), __float_as_uint(vc[2]), 0x96);
    const unsigned x345 = __builtin_amdgcn_bitop3_b32(__float_as_uint(vc[3]), __float_as_uint(vc[4]), __float_as_uint(vc[5]), 0x96);
    const unsigned x67 = __float_as_uint(vc[6]) ^ __float_as_uint(vc[7]);
    const unsigned sx = __builtin_amdgcn_bitop3_b32(x012, x345, x67, 0x96);
    // cv = alpha * mag * (parity of all signs) * sign(vc): the row parity is folded into the scale factor once
    // per check (a float product by -alpha is the negated product by alpha, bit for bit), which leaves one
    // multiply and one "x ^ (vc & sign bit)" per edge instead of multiply + XOR + sign insert
    const float arow = __uint_as_float(__float_as_uint(aeff) ^ (sx & signv));
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float mag = arow * o[t];
        cv[BR * 8 + t] = __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(vc[t]) & signv));
    }
}

template <int BR, bool UP>
__device__ __forceinline__ int syndrome_row(const int (&h)[8])
{
    int s = 0;
#define LDPC_SY(t) s ^= rot_i<16 - kTerms[BR * 8 + t].s, UP>(h[kTerms[BR * 8 + t].bc]);
    LDPC_SY(0) LDPC_SY(1) LDPC_SY(2) LDPC_SY(3) LDPC_SY(4) LDPC_SY(5) LDPC_SY(6) LDPC_SY(7)
#undef LDPC_SY
    return s;
}

// ROWS = false: the decoder (soft / traj / hard / fail of frames 0 .. B).  ROWS = true: the failed-frame form of
// collect_failed_output_selective (ms_test.py:55-64): frame f of the launch is llr[index[f]], f < min(*count, B), and the only
// output is traj[f][0..T][128] -- row 0 the channel values, row t the posterior after iteration t: the reference's buffer order.
// The reference keeps T + 1 rows of the FAILED frames only; writing [T][B][128] for all frames (round 3's surface) was
// 5.1 KiB per input frame against 1.4 KiB at 2.5 dB.  Same arithmetic, same order: the rows equal the full trajectory's.
template <bool UP, bool ROWS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void nms_qc16_kernel(const float *__restrict__ llr, long long B, int T,
                                                       AlphaArg alpha, float w_in, float w_out,
                                                       float *__restrict__ soft, float *__restrict__ traj,
                                                       unsigned long long *__restrict__ hard,
                                                       unsigned char *__restrict__ fail,
                                                       const int *__restrict__ index = nullptr, const int *__restrict__ count = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, r = lane >> 4;
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long f = wave_id * 4 + r;
    if constexpr (ROWS) {
        const long long c = *count;
        B = c < B ? c : B;
        if (wave_id * 4 >= B) return;      // (the grid is sized for the capacity; the count is device data)
    }
    const bool live = f < B;
    const long long fl = live ? f : B - 1;  // dead rows compute on a valid frame, store nothing

    float y[8], yin[8], yout[8], S[8], cv[32];
    const float *src = llr + (ROWS ? (long long)index[fl] : fl) * 128 + j;
#pragma unroll
    for (int bc = 0; bc < 8; ++bc) {
        y[bc] = src[bc * 16];
        yin[bc] = y[bc] * w_in;
        yout[bc] = w_out * y[bc];
        S[bc] = 0.0f;
    }
    if constexpr (ROWS) {
        if (live) {
            float *dst = traj + fl * (long long)(T + 1) * 128 + j;
#pragma unroll
            for (int bc = 0; bc < 8; ++bc) dst[bc * 16] = y[bc];
        }
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) cv[e] = 0.0f;

    unsigned signv;  // 0x80000000 held in a VGPR on purpose (see check_row)
    asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(signv));
    for (int it = 0; it < T; ++it) {
        float a_it;      // alpha[it] copied to a VGPR: VALU ops with an SGPR operand issue at half rate
        asm volatile("v_mov_b32 %0, %1" : "=v"(a_it) : "s"(alpha.a[it]));
        float tot[8];
#pragma unroll
        for (int bc = 0; bc < 8; ++bc) tot[bc] = S[bc] + yin[bc];
        check_row<0, UP>(cv, tot, a_it, signv);
        check_row<1, UP>(cv, tot, a_it, signv);
        check_row<2, UP>(cv, tot, a_it, signv);
        check_row<3, UP>(cv, tot, a_it, signv);
        var_sums<UP>(cv, j, S);
        if (ROWS || traj) {
            float *dst = ROWS ? traj + (fl * (long long)(T + 1) + it + 1) * 128 + j : traj + ((long long)it * B + fl) * 128 + j;
            if (live) {
#pragma unroll
                for (int bc = 0; bc < 8; ++bc) dst[bc * 16] = S[bc] + yout[bc];
            }
        }
    }

    if constexpr (ROWS) return;
    float out[8];
    int h[8];
#pragma unroll
    for (int bc = 0; bc < 8; ++bc) {
        out[bc] = (T > 0) ? S[bc] + yout[bc] : y[bc];
        h[bc] = !(out[bc] > 0.0f);
    }
    if (soft && live) {
        float *dst = soft + fl * 128 + j;
#pragma unroll
        for (int bc = 0; bc < 8; ++bc) dst[bc * 16] = out[bc];
    }
    if (hard) {
        unsigned long long b[8];
#pragma unroll
        for (int bc = 0; bc < 8; ++bc) b[bc] = __ballot(h[bc]);
        if (j < 2 && live) {
            unsigned long long w = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long src_b = j ? b[4 + q] : b[q];
                w |= ((src_b >> (16 * r)) & 0xFFFFull) << (16 * q);
            }
            hard[fl * 2 + j] = w;
        }
    }
    if (fail) {
        int s = syndrome_row<0, UP>(h) | syndrome_row<1, UP>(h) | syndrome_row<2, UP>(h) | syndrome_row<3, UP>(h);
        const unsigned long long bal = __ballot(s != 0);
        if (j == 0 && live) fail[fl] = ((bal >> (16 * r)) & 0xFFFFull) != 0;
    }
}

__global__ void dpp_probe_kernel(int *out)
{
    int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(0, lane, 0x121, 0xF, 0xF, true);        // row_ror:1
    out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x134, 0xF, 0xF, false);  // wave_rol:1
}

int probe_dpp(bool *ror_up, int *wave_rol_dir)
{
    int *d = nullptr, h[128];
    LDPC_HIP(hipMalloc(&d, sizeof(h)));
    hipLaunchKernelGGL(dpp_probe_kernel, dim3(1), dim3(64), 0, 0, d);
    hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "dpp probe");
    if (h[1] == 0 && h[0] == 15) *ror_up = true;        // lane j received lane j-1
    else if (h[1] == 2 && h[0] == 1) *ror_up = false;   // lane j received lane j+1
    else return fail(LDPC_E_HIP, "unexpected row_ror behaviour: lane0=%d lane1=%d", h[0], h[1]);
    // wave_rol:1 over all 64 lanes: +1 = lane j received lane j-1, -1 = lane j received lane j+1, 0 = not a rotation
    // (then the order-2 scan keeps its v_readlane pairing)
    int up = 1, down = 1;
    for (int j = 0; j < 64; ++j) { up &= h[64 + j] == ((j + 63) & 63); down &= h[64 + j] == ((j + 1) & 63); }
    *wave_rol_dir = up ? 1 : (down ? -1 : 0);
    return LDPC_OK;
}

int launch_nms(ldpc_ctx *ctx, const float *d_llr, int64_t B, int T, const float *alpha, float w_in, float w_out,
               float *d_soft, float *d_traj, uint64_t *d_hard, uint8_t *d_fail, int kernel, hipStream_t st,
               const int32_t *d_index, const int32_t *d_count, float *d_rows)
{
    const ldpc_code &c = ctx->code;
    if (kernel == LDPC_NMS_AUTO) kernel = c.qc16_ccsds ? LDPC_NMS_QC16 : LDPC_NMS_GENERIC;
    if (kernel == LDPC_NMS_QC16 && !c.qc16_ccsds)
        return fail(LDPC_E_UNSUPPORTED, "QC16 kernel requested for a code that is not the CCSDS (128,64) circulant array");
    AlphaArg a;
    for (int i = 0; i < kMaxIters; ++i) a.a[i] = i < T ? alpha[i] : 0.0f;
    auto *hard = reinterpret_cast<unsigned long long *>(d_hard);
    if (kernel == LDPC_NMS_QC16) {
        const unsigned blocks = (unsigned)((B + 15) / 16);
        if (d_rows) {
            if (ctx->dpp_ror_up)
                hipLaunchKernelGGL((nms_qc16_kernel<true, true>), dim3(blocks), dim3(256), 0, st, d_llr, (long long)B, T, a, w_in, w_out,
                                   (float *)nullptr, d_rows, (unsigned long long *)nullptr, (unsigned char *)nullptr, d_index, d_count);
            else
                hipLaunchKernelGGL((nms_qc16_kernel<false, true>), dim3(blocks), dim3(256), 0, st, d_llr, (long long)B, T, a, w_in, w_out,
                                   (float *)nullptr, d_rows, (unsigned long long *)nullptr, (unsigned char *)nullptr, d_index, d_count);
        } else if (ctx->dpp_ror_up)
            hipLaunchKernelGGL((nms_qc16_kernel<true, false>), dim3(blocks), dim3(256), 0, st, d_llr, (long long)B, T, a, w_in,
                               w_out, d_soft, d_traj, hard, d_fail, (const int *)nullptr, (const int *)nullptr);
        else
            hipLaunchKernelGGL((nms_qc16_kernel<false, false>), dim3(blocks), dim3(256), 0, st, d_llr, (long long)B, T, a, w_in,
                               w_out, d_soft, d_traj, hard, d_fail, (const int *)nullptr, (const int *)nullptr);
    } else if (kernel == LDPC_NMS_GENERIC) {
        const size_t lds = sizeof(float) * 4 * ((size_t)c.E + 2 * (size_t)c.n);
        if (lds > 160 * 1024) return fail(LDPC_E_UNSUPPORTED, "code too large for the generic NMS kernel (%zu B of LDS)", lds);
        if (lds > 64 * 1024) {   // beyond the default dynamic-LDS limit: opt in (gfx950 has 160 KiB per CU)
            static thread_local size_t granted = 0;
            if (lds > granted) {
                LDPC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nms_generic_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                granted = lds;
            }
        }
        long long want = (B + 3) / 4;
        const unsigned blocks = (unsigned)(want < 8192 ? want : 8192);
        hipLaunchKernelGGL(nms_generic_kernel, dim3(blocks), dim3(256), lds, st, d_llr, (long long)B, T, a, w_in, w_out,
                           d_soft, d_traj, hard, d_fail, ctx->d_chk_ptr, ctx->d_chk_var, ctx->d_var_ptr,
                           ctx->d_var_edge, c.n, c.m, c.E, d_index, d_count, d_rows);
    } else {
        return fail(LDPC_E_ARG, "unknown NMS kernel id %d", kernel);
    }
    LDPC_HIP(hipGetLastError());
    return LDPC_OK;
}

}  // namespace ldpc
