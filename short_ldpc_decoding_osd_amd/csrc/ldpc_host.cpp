// Host side of libldpcosd.so: code definition (alist -> H -> G), Tanner-graph tables,
// TEP tables, error reporting.  One-time work; the per-frame hot path is in the .hip files.
//
// Reference behaviour reproduced here (paths relative to LDPC_128/ of the reference):
//   alist parsing      Ldpc_128_testing/fill_matrix_info.py:70-104
//   GF(2) elimination  Ldpc_128_testing/fill_matrix_info.py:7-42 (== PB_OSD/pb_testing.py:231-266)
//   generator matrix   Ldpc_128_testing/fill_matrix_info.py:44-69
//   TEP table          FS_OSD/convention_osd.py:13-47
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <fstream>
#include <sstream>

#include "ldpc_internal.h"

namespace ldpc {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    return fail(LDPC_E_HIP, "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
}

// ---------------------------------------------------------------------------------------
// Bit-row matrix used for the one-time eliminations.
// ---------------------------------------------------------------------------------------
struct BitMat {
    int rows, cols, words;
    std::vector<uint64_t> w;
    BitMat(int r, int c) : rows(r), cols(c), words((c + 63) / 64), w((size_t)r * ((c + 63) / 64), 0) {}
    uint64_t *row(int r) { return &w[(size_t)r * words]; }
    bool get(int r, int c) { return (row(r)[c >> 6] >> (c & 63)) & 1; }
    void flip(int r, int c) { row(r)[c >> 6] ^= 1ull << (c & 63); }
    void swap_cols(int a, int b)
    {
        for (int r = 0; r < rows; ++r)
            if (get(r, a) != get(r, b)) { flip(r, a); flip(r, b); }
    }
    void drop_row(int r)
    {
        w.erase(w.begin() + (size_t)r * words, w.begin() + (size_t)(r + 1) * words);
        --rows;
    }
};

// Gauss-Jordan with the reference's pivot choice: first row at or below the diagonal that
// has a 1 in the current column; if none, exchange the current column with the first later
// column in which the *diagonal row* has a 1 (recorded), or delete the row if it has none.
static void eliminate(BitMat &M, std::vector<int32_t> *swaps)
{
    int i = 0, j = 0;
    while (i < M.rows && j < M.cols) {
        int piv = -1;
        for (int r = i; r < M.rows; ++r)
            if (M.get(r, j)) { piv = r; break; }
        if (piv < 0) {
            int col = -1;
            for (int c = j; c < M.cols; ++c)
                if (M.get(i, c)) { col = c; break; }
            if (col < 0) { M.drop_row(i); continue; }
            M.swap_cols(j, col);
            if (swaps) { swaps->push_back(j); swaps->push_back(col); }
        } else if (piv != i) {
            std::swap_ranges(M.row(piv), M.row(piv) + M.words, M.row(i));
        }
        // rows above the diagonal are already clear left of j, so whole-row XOR == XOR of columns j..
        const uint64_t *p = M.row(i);
        for (int r = 0; r < M.rows; ++r) {
            if (r == i || !M.get(r, j)) continue;
            uint64_t *q = M.row(r);
            for (int t = j >> 6; t < M.words; ++t) q[t] ^= p[t];
        }
        ++i; ++j;
    }
}

int gf2elim(int32_t *Mi, int m, int n, std::vector<int32_t> *swaps)
{
    BitMat M(m, n);
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < n; ++c)
            if (Mi[(size_t)r * n + c] & 1) M.flip(r, c);
    // a pivot row may still hold ones left of j in rows that were never a pivot row's target;
    // the reference XORs only columns j.. -- identical here because those columns are zero
    // in the pivot row (it sits below all earlier pivots and was cleared by them).
    eliminate(M, swaps);
    for (int r = 0; r < M.rows; ++r)
        for (int c = 0; c < n; ++c) Mi[(size_t)r * n + c] = M.get(r, c);
    return M.rows;
}

static const QcTerm kCcsds128[32] = {
    // CCSDS 131.1-O (128,64): 4 x 8 array of 16x16 circulants, M = 16
    {0, 0, 0}, {0, 0, 7}, {0, 1, 2}, {0, 2, 14}, {0, 3, 6}, {0, 5, 0}, {0, 6, 13}, {0, 7, 0},
    {1, 0, 6}, {1, 1, 0}, {1, 1, 15}, {1, 2, 0}, {1, 3, 1}, {1, 4, 0}, {1, 6, 0}, {1, 7, 7},
    {2, 0, 4}, {2, 1, 1}, {2, 2, 0}, {2, 2, 15}, {2, 3, 14}, {2, 4, 11}, {2, 5, 0}, {2, 7, 3},
    {3, 0, 0}, {3, 1, 1}, {3, 2, 9}, {3, 3, 0}, {3, 3, 13}, {3, 4, 14}, {3, 5, 1}, {3, 6, 0},
};

static bool matches_ccsds128(const ldpc_code &c)
{
    if (c.n != 128 || c.m != 64 || c.E != 512) return false;
    std::vector<int32_t> ref((size_t)64 * 128, 0);
    for (const QcTerm &t : kCcsds128)
        for (int i = 0; i < 16; ++i) ref[(size_t)(t.br * 16 + i) * 128 + t.bc * 16 + (i + t.s) % 16] ^= 1;
    return ref == c.H;
}

int build_code(ldpc_code &c)
{
    const int m = c.m, n = c.n;
    // generator: GE -> [I | H2]; G = [H2^T | I]; undo the recorded column exchanges backwards
    BitMat R(m, n);
    for (int r = 0; r < m; ++r)
        for (int v = 0; v < n; ++v)
            if (c.H[(size_t)r * n + v]) R.flip(r, v);
    std::vector<int32_t> sw;
    eliminate(R, &sw);
    const int rank = R.rows, k = n - rank;
    if (k <= 0) return fail(LDPC_E_CODE, "H has no null space (rank %d, n %d)", rank, n);
    c.k = k;
    c.G.assign((size_t)k * n, 0);
    for (int a = 0; a < k; ++a) {
        for (int r = 0; r < rank; ++r) c.G[(size_t)a * n + r] = R.get(r, rank + a);
        c.G[(size_t)a * n + rank + a] = 1;
    }
    for (int s = (int)sw.size() / 2 - 1; s >= 0; --s)
        for (int a = 0; a < k; ++a) std::swap(c.G[(size_t)a * n + sw[2 * s]], c.G[(size_t)a * n + sw[2 * s + 1]]);
    for (int r = 0; r < m; ++r)
        for (int a = 0; a < k; ++a) {
            int acc = 0;
            for (int v = 0; v < n; ++v) acc ^= c.H[(size_t)r * n + v] & c.G[(size_t)a * n + v];
            if (acc) return fail(LDPC_E_CODE, "generator check failed: H.G^T != 0 at (%d,%d)", r, a);
        }
    // Tanner graph, edges numbered check-major; per variable the edges in ascending check order
    c.chk_ptr.assign(m + 1, 0);
    c.chk_var.clear();
    c.max_var_degree = 0;
    int maxc = 0;
    for (int r = 0; r < m; ++r) {
        c.chk_ptr[r] = (int32_t)c.chk_var.size();
        for (int v = 0; v < n; ++v)
            if (c.H[(size_t)r * n + v]) c.chk_var.push_back(v);
        maxc = std::max(maxc, (int)c.chk_var.size() - c.chk_ptr[r]);
    }
    c.chk_ptr[m] = (int32_t)c.chk_var.size();
    c.E = (int)c.chk_var.size();
    if (c.max_chk_degree < maxc) c.max_chk_degree = maxc;
    c.var_ptr.assign(n + 1, 0);
    c.var_edge.clear();
    for (int v = 0; v < n; ++v) {
        c.var_ptr[v] = (int32_t)c.var_edge.size();
        for (int r = 0; r < m; ++r)
            for (int e = c.chk_ptr[r]; e < c.chk_ptr[r + 1]; ++e)
                if (c.chk_var[e] == v) c.var_edge.push_back(e);
        c.max_var_degree = std::max(c.max_var_degree, (int)c.var_edge.size() - c.var_ptr[v]);
    }
    c.var_ptr[n] = (int32_t)c.var_edge.size();
    c.qc16_ccsds = matches_ccsds128(c);
    return LDPC_OK;
}

static int64_t binom(int n, int r)
{
    int64_t v = 1;
    for (int i = 1; i <= r; ++i) v = v * (n - r + i) / i;
    return v;
}

int64_t tep_table(int k, int order, uint8_t *supports, int64_t *boundaries)
{
    if (k < 1 || k > 255 || order < 0 || order > 3) return fail(LDPC_E_ARG, "tep_table: k=%d order=%d", k, order);
    int64_t total = 0;
    for (int w = 0; w <= order; ++w) {
        total += binom(k, w);
        if (boundaries) boundaries[w] = total;
    }
    if (!supports) return total;
    int64_t base = 0;
    for (int w = 0; w <= order; ++w) {
        // all weight-w supports in lexicographic order, then a stable sort by descending index sum
        std::vector<std::array<uint8_t, 3>> cls;
        cls.reserve((size_t)binom(k, w));
        std::array<uint8_t, 3> cur = {0xFF, 0xFF, 0xFF};
        if (w == 0) cls.push_back(cur);
        for (int a = 0; w >= 1 && a < k; ++a) {
            if (w == 1) { cls.push_back({(uint8_t)a, 0xFF, 0xFF}); continue; }
            for (int b = a + 1; b < k; ++b) {
                if (w == 2) { cls.push_back({(uint8_t)a, (uint8_t)b, 0xFF}); continue; }
                for (int d = b + 1; d < k; ++d) cls.push_back({(uint8_t)a, (uint8_t)b, (uint8_t)d});
            }
        }
        auto key = [w](const std::array<uint8_t, 3> &s) { int t = 0; for (int q = 0; q < w; ++q) t += s[q]; return t; };
        std::stable_sort(cls.begin(), cls.end(), [&](const auto &x, const auto &y) { return key(x) > key(y); });
        for (const auto &s : cls) { memcpy(supports + 3 * base, s.data(), 3); ++base; }
    }
    return total;
}

// FS-OSD visit order of ONE weight class (generate_sequential_teps, FS_OSD/fs_testing.py:32-49):
// lexicographic combinations of range(k) with the indicator vector reversed, i.e. support {k-1-p}.
// supports: [count][3] ascending positions, 0xFF padded.  Returns C(k, w).
int64_t tep_table_fs(int k, int w, uint8_t *supports)
{
    if (k < 1 || k > 255 || w < 1 || w > 3) return fail(LDPC_E_ARG, "tep_table_fs: k=%d w=%d", k, w);
    const int64_t total = binom(k, w);
    if (!supports) return total;
    int c[3] = {0, 1, 2};
    for (int64_t t = 0; t < total; ++t) {
        uint8_t *dst = supports + 3 * t;
        dst[0] = dst[1] = dst[2] = 0xFF;
        for (int q = 0; q < w; ++q) dst[w - 1 - q] = (uint8_t)(k - 1 - c[q]);  // reversed => ascending
        int q = w - 1;
        while (q >= 0 && c[q] == k - w + q) --q;
        if (q < 0) break;
        ++c[q];
        for (int z = q + 1; z < w; ++z) c[z] = c[z - 1] + 1;
    }
    return total;
}


// osd.error_pattern_gen (DL_OSD_Testing_serial/ordered_statistics_decoding.py:81-98): product over
// the segments (leftmost slowest) of the lexicographic combinations inside each segment
int64_t hosd_pattern_teps(int nseg, const int32_t *bounds, const int32_t *pattern, uint8_t *teps)
{
    if (nseg < 1 || nseg > 64 || !bounds || !pattern) return fail(LDPC_E_ARG, "hosd_pattern_teps: bad arguments");
    int weight = 0;
    int64_t total = 1;
    for (int s = 0; s < nseg; ++s) {
        const int len = bounds[s + 1] - bounds[s];
        if (bounds[s] < 0 || bounds[s + 1] > 64 || len < 0 || pattern[s] < 0)
            return fail(LDPC_E_ARG, "hosd_pattern_teps: segment %d = [%d, %d), %d flips", s, bounds[s], bounds[s + 1], pattern[s]);
        weight += pattern[s];
        total *= pattern[s] > len ? 0 : binom(len, pattern[s]);   // combinations() of too many items is empty
    }
    if (weight > 3) return fail(LDPC_E_UNSUPPORTED, "hosd_pattern_teps: pattern weight %d > 3", weight);
    if (!teps || total == 0) return total;
    // odometer over the flipped positions: slot q belongs to segment seg[q], slots of one segment ascend
    int seg[3], pos[3];
    int w = 0;
    for (int s = 0; s < nseg; ++s)
        for (int q = 0; q < pattern[s]; ++q) { seg[w] = s; pos[w] = bounds[s] + q; ++w; }
    for (int64_t t = 0; t < total; ++t) {
        uint8_t *dst = teps + 4 * t;
        dst[0] = dst[1] = dst[2] = 0;
        for (int q = 0; q < w; ++q) dst[q] = (uint8_t)pos[q];
        dst[3] = (uint8_t)w;
        // advance: rightmost slot that can still move inside its segment
        int q = w - 1;
        while (q >= 0) {
            int after = 0;   // slots of the same segment to the right of q
            for (int z = q + 1; z < w && seg[z] == seg[q]; ++z) ++after;
            if (pos[q] < bounds[seg[q] + 1] - 1 - after) break;
            --q;
        }
        if (q < 0) break;
        ++pos[q];
        for (int z = q + 1; z < w; ++z) pos[z] = seg[z] == seg[z - 1] ? pos[z - 1] + 1 : bounds[seg[z]];
    }
    return total;
}

}  // namespace ldpc

using namespace ldpc;

extern "C" {

const char *ldpc_last_error(void) { return g_err; }
int ldpc_abi_version(void) { return LDPC_OSD_ABI_VERSION; }

int ldpc_code_from_dense(const int32_t *H, int32_t m, int32_t n, ldpc_code **out)
{
    if (!H || !out || m <= 0 || n <= m) return fail(LDPC_E_ARG, "ldpc_code_from_dense: bad arguments");
    ldpc_code *c = new ldpc_code();
    c->m = m; c->n = n;
    c->H.resize((size_t)m * n);
    for (size_t t = 0; t < (size_t)m * n; ++t) c->H[t] = H[t] & 1;
    int rc = build_code(*c);
    if (rc) { delete c; return rc; }
    *out = c;
    return LDPC_OK;
}

int ldpc_code_from_alist(const char *path, ldpc_code **out)
{
    if (!path || !out) return fail(LDPC_E_ARG, "ldpc_code_from_alist: null argument");
    std::ifstream in(path);
    if (!in) return fail(LDPC_E_IO, "cannot open alist file '%s'", path);
    std::vector<std::vector<long>> rows;
    std::string line;
    while (std::getline(in, line)) {
        std::vector<long> toks;
        std::istringstream ls(line);
        long v;
        while (ls >> v) toks.push_back(v);
        rows.push_back(std::move(toks));
    }
    if (rows.size() < 4 || rows[0].size() < 2 || rows[1].size() < 2)
        return fail(LDPC_E_IO, "'%s': not an alist file (header)", path);
    const long n = rows[0][0], m = rows[0][1];
    if (n <= 0 || m <= 0 || m >= n || (long)rows.size() < 4 + n)
        return fail(LDPC_E_IO, "'%s': bad alist sizes n=%ld m=%ld lines=%zu", path, n, m, rows.size());
    std::vector<int32_t> H((size_t)m * n, 0);
    for (long v = 0; v < n; ++v)
        for (long chk : rows[4 + v]) {
            if (chk == 0) continue;  // padding entry
            if (chk < 0 || chk > m) return fail(LDPC_E_IO, "'%s': check index %ld out of range at variable %ld", path, chk, v + 1);
            H[(size_t)(chk - 1) * n + v] = 1;
        }
    ldpc_code *c = nullptr;
    int rc = ldpc_code_from_dense(H.data(), (int32_t)m, (int32_t)n, &c);
    if (rc) return rc;
    c->max_chk_degree = (int)rows[1][1];  // the reference keeps the header value (fill_matrix_info.py:87,124)
    *out = c;
    return LDPC_OK;
}

void ldpc_code_destroy(ldpc_code *code) { delete code; }

int ldpc_code_dims(const ldpc_code *c, int32_t *n, int32_t *m, int32_t *k, int32_t *mcd)
{
    if (!c) return fail(LDPC_E_ARG, "ldpc_code_dims: null code");
    if (n) *n = c->n;
    if (m) *m = c->m;
    if (k) *k = c->k;
    if (mcd) *mcd = c->max_chk_degree;
    return LDPC_OK;
}

int ldpc_code_get_H(const ldpc_code *c, int32_t *H)
{
    if (!c || !H) return fail(LDPC_E_ARG, "ldpc_code_get_H: null argument");
    memcpy(H, c->H.data(), sizeof(int32_t) * c->H.size());
    return LDPC_OK;
}

int ldpc_code_get_G(const ldpc_code *c, int32_t *G)
{
    if (!c || !G) return fail(LDPC_E_ARG, "ldpc_code_get_G: null argument");
    memcpy(G, c->G.data(), sizeof(int32_t) * c->G.size());
    return LDPC_OK;
}

int ldpc_gf2elim_host(int32_t *M, int32_t m, int32_t n, int32_t *swaps, int32_t *nswaps, int32_t *rows_out)
{
    if (!M || m <= 0 || n <= 0) return fail(LDPC_E_ARG, "ldpc_gf2elim_host: bad arguments");
    std::vector<int32_t> sw;
    int rows = gf2elim(M, m, n, &sw);
    if (swaps && !sw.empty()) memcpy(swaps, sw.data(), sizeof(int32_t) * sw.size());   // (an empty vector has a null data(): UB for memcpy -- found by the sanitizer build)
    if (nswaps) *nswaps = (int32_t)(sw.size() / 2);
    if (rows_out) *rows_out = rows;
    return LDPC_OK;
}

int64_t ldpc_tep_table(int32_t k, int32_t order, uint8_t *supports, int64_t *boundaries)
{
    return tep_table(k, order, supports, boundaries);
}

// CRC-32C (Castagnoli), the checksum of the TFRecord framing (tensorflow/core/lib/hash/crc32c)
uint32_t ldpc_crc32c(const void *data, uint64_t len)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int b = 0; b < 8; ++b) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
            table[i] = c;
        }
        ready = true;
    }
    const uint8_t *p = static_cast<const uint8_t *>(data);
    uint32_t c = 0xFFFFFFFFu;
    for (uint64_t i = 0; i < len; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

int64_t ldpc_tep_table_fs(int32_t k, int32_t weight, uint8_t *supports) { return tep_table_fs(k, weight, supports); }

int64_t ldpc_hosd_pattern_teps(int32_t nseg, const int32_t *bounds, const int32_t *pattern, uint8_t *teps)
{
    return hosd_pattern_teps(nseg, bounds, pattern, teps);
}

}  // extern "C"
