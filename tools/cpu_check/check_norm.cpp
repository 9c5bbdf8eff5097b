// Runs the normalisation kernels (videoswap_amd/csrc/norm.hip: GroupNorm statistics / finalize / apply with the two-source
// concat and SiLU, LayerNorm with the temporal positional encoding, the row statistics of the LayerNorm fold) on the CPU from
// their real source and compares with double-precision references.  Same emulation as check_train.cpp (hip/hip_runtime.h).
#include "hip/hip_runtime.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
#include "common.h"

// wave shuffle towards lane 0 (hip/hip_runtime.h has __shfl / __shfl_xor): lanes whose partner lies past the wave keep their value
static inline float __shfl_down(float v, int delta, int width = 64) {
    (void)width;
    const int lane = (int)(cpuhip::ctx.tid.x & 63);
    cpuhip::ctx.wave_slots[lane] = v;
    cpuhip::ctx.wave_bar->arrive_and_wait();
    const float r = lane + delta < 64 ? cpuhip::ctx.wave_slots[lane + delta] : v;
    cpuhip::ctx.wave_bar->arrive_and_wait();
    return r;
}
static long g_gn_fuse = 1;
namespace vsxg {
long gemm_option(const char*) { return g_gn_fuse; }      // the option table lives in gemm.hip: only "gn_fuse" is asked here
}

int vsx_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
    return code;
}
int vsx_check_launch(const char*) { return 0; }

#include "norm.hip"


static unsigned rng_state = 2024u;
static float frand() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}
static std::vector<half_t> randh(size_t n, float scale = 1.0f, float shift = 0.f) {
    std::vector<half_t> v(n);
    for (auto& x : v) x = (half_t)(frand() * scale + shift);
    return v;
}
static int n_bad = 0;
static void report(const char* name, int rc, const std::vector<double>& want, const std::vector<double>& got, double tol) {
    double num = 0, den = 0;
    for (size_t i = 0; i < want.size(); ++i) {
        num += (got[i] - want[i]) * (got[i] - want[i]);
        den += want[i] * want[i];
    }
    const double rel = sqrt(num / (den > 0 ? den : 1));
    const bool ok = rc == 0 && rel < tol;
    printf("%-64s rc %d rel-L2 %.2e %s\n", name, rc, rel, ok ? "ok" : "FAIL");
    if (!ok) ++n_bad;
}
template <typename T>
static std::vector<double> dbl(const std::vector<T>& v) {
    std::vector<double> o(v.size());
    for (size_t i = 0; i < v.size(); ++i) o[i] = (double)v[i];
    return o;
}

// GroupNorm over [nimg, rows, C1 (+ C2)] (5-D scope: nimg = batch, rows = frames x pixels; 4-D: nimg = batch x frames)
static void run_groupnorm(const char* name, long nimg, long rows, long C1, long C2, long groups, bool silu) {
    const long C = C1 + C2, cpg = C / groups;
    auto x1 = randh((size_t)nimg * rows * C1, 1.5f, 0.4f), x2 = randh((size_t)nimg * rows * (C2 ? C2 : 8), 0.7f, -0.2f);
    auto gamma = randh(C, 0.3f, 1.0f), beta = randh(C, 0.2f);
    std::vector<double> want((size_t)nimg * rows * C);
    auto at = [&](long i, long r, long c) { return c < C1 ? (double)x1[(i * rows + r) * C1 + c] : (double)x2[(i * rows + r) * C2 + c - C1]; };
    for (long i = 0; i < nimg; ++i)
        for (long g = 0; g < groups; ++g) {
            double s = 0, q = 0;
            for (long r = 0; r < rows; ++r)
                for (long c = g * cpg; c < (g + 1) * cpg; ++c) { const double v = at(i, r, c); s += v; q += v * v; }
            const double n = (double)rows * cpg, mean = s / n, rstd = 1.0 / sqrt(q / n - mean * mean + 1e-5);
            for (long r = 0; r < rows; ++r)
                for (long c = g * cpg; c < (g + 1) * cpg; ++c) {
                    double y = (at(i, r, c) - mean) * rstd * (double)gamma[c] + (double)beta[c];
                    if (silu) y = y / (1.0 + exp(-y));
                    want[(i * rows + r) * C + c] = y;
                }
        }
    const long nchunks = vsx_groupnorm_chunks(rows, nimg);
    std::vector<float> partial((size_t)nimg * nchunks * groups * 2, -1.f), stats((size_t)nimg * groups * 2);
    std::vector<half_t> y((size_t)nimg * rows * C, (half_t)-7.f);
    int rc = vsx_groupnorm_stats(x1.data(), C2 ? x2.data() : nullptr, nimg, rows, C1, C2, groups, partial.data(), nullptr);
    if (rc == 0)
        rc = vsx_groupnorm_apply(x1.data(), C2 ? x2.data() : nullptr, nimg, rows, C1, C2, groups, partial.data(), nchunks, rows,
                                 gamma.data(), beta.data(), 1e-5f, silu ? 1 : 0, stats.data(), y.data(), nullptr);
    report(name, rc, want, dbl(y), 2e-3);
    // few chunks per image: the apply kernel finalized the statistics itself (GN_FUSE_MAX_CHUNKS); with the stand-alone finalize
    // kernel (option gn_fuse = 0) statistics and output must come out bit for bit the same
    if (rc == 0 && nchunks <= 64) {
        std::vector<float> stats2((size_t)nimg * groups * 2, -3.f);
        std::vector<half_t> y2((size_t)nimg * rows * C, (half_t)-7.f);
        g_gn_fuse = 0;
        rc = vsx_groupnorm_apply(x1.data(), C2 ? x2.data() : nullptr, nimg, rows, C1, C2, groups, partial.data(), nchunks, rows,
                                 gamma.data(), beta.data(), 1e-5f, silu ? 1 : 0, stats2.data(), y2.data(), nullptr);
        g_gn_fuse = 1;
        const bool same = rc == 0 && memcmp(stats.data(), stats2.data(), stats.size() * sizeof(float)) == 0 &&
                          memcmp(y.data(), y2.data(), y.size() * sizeof(half_t)) == 0;
        printf("%-64s %s\n", "  fused finalize (in the apply kernel) == stand-alone finalize, bit for bit", same ? "ok" : "FAIL");
        if (!same) ++n_bad;
    } else if (rc == 0) {
        printf("%-64s (%ld chunks per image: stand-alone finalize)\n", "  ", nchunks);
    }
}

static void run_layernorm(const char* name, long M, long C, bool pe, long rows_per_frame, long frames, long frame_offset) {
    auto x = randh((size_t)M * C, 2.0f, 0.5f), gamma = randh(C, 0.3f, 1.0f), beta = randh(C, 0.2f);
    auto pos = randh((size_t)(frames + frame_offset + 1) * C, 0.5f);
    std::vector<double> want((size_t)M * C), wstats((size_t)M * 2);
    for (long m = 0; m < M; ++m) {
        double s = 0, q = 0;
        for (long c = 0; c < C; ++c) { const double v = (double)x[m * C + c]; s += v; q += v * v; }
        const double mean = s / C, rstd = 1.0 / sqrt(q / C - mean * mean + 1e-5);
        wstats[2 * m] = rstd;
        wstats[2 * m + 1] = -rstd * mean;
        for (long c = 0; c < C; ++c) {
            double y = ((double)x[m * C + c] - mean) * rstd * (double)gamma[c] + (double)beta[c];
            if (pe) y += (double)pos[(((m / rows_per_frame) % frames) + frame_offset) * C + c];
            want[m * C + c] = y;
        }
    }
    std::vector<half_t> y((size_t)M * C, (half_t)-7.f);
    const int rc = vsx_layernorm(x.data(), M, C, gamma.data(), beta.data(), 1e-5f, pe ? pos.data() : nullptr, rows_per_frame, frames,
                                 frame_offset, y.data(), nullptr);
    report(name, rc, want, dbl(y), 2e-3);
    if (!pe) {
        std::vector<float> st((size_t)M * 2, -9.f);
        char nm[128];
        snprintf(nm, sizeof nm, "  row statistics (rstd, -rstd * mean) of the same rows");
        report(nm, vsx_row_stats(x.data(), M, C, 1e-5f, st.data(), nullptr), wstats, dbl(st), 1e-5);
        // the same pairs from partial sums over column parts of 64, 64, 32, ... columns (what a producing GEMM writes)
        std::vector<long> edges = {0};
        while (edges.back() < C) edges.push_back(std::min(C, edges.back() + ((edges.size() % 3) == 0 ? 32 : 64)));
        const long np = (long)edges.size() - 1;
        std::vector<float> parts((size_t)M * np * 2);
        for (long m = 0; m < M; ++m)
            for (long q = 0; q < np; ++q) {
                float s1 = 0.f, s2 = 0.f;
                for (long c = edges[q]; c < edges[q + 1]; ++c) { const float v = (float)x[m * C + c]; s1 += v; s2 += v * v; }
                parts[(m * np + q) * 2] = s1;
                parts[(m * np + q) * 2 + 1] = s2;
            }
        std::fill(st.begin(), st.end(), -9.f);
        snprintf(nm, sizeof nm, "  ... combined from %ld partial sums per row", np);
        report(nm, vsx_row_stats_combine(parts.data(), M, np, C, 1e-5f, st.data(), nullptr), wstats, dbl(st), 1e-4);
    }
}

static void run_softmax(const char* name, long nrows, long ncols, long ld, long rows_per_seq) {
    auto S = randh((size_t)nrows * ld, 4.0f);
    std::vector<double> want((size_t)nrows * ld, 0.0);
    for (long r = 0; r < nrows; ++r) {
        const long lim = rows_per_seq ? std::min(ncols, r % rows_per_seq + 1) : ncols;       // causal: keys 0 .. row index
        double mx = -1e30, sum = 0;
        for (long c = 0; c < lim; ++c) mx = std::max(mx, (double)S[r * ld + c]);
        for (long c = 0; c < lim; ++c) sum += exp((double)S[r * ld + c] - mx);
        for (long c = 0; c < ncols; ++c) want[r * ld + c] = c < lim ? exp((double)S[r * ld + c] - mx) / sum : 0.0;
        for (long c = ncols; c < ld; ++c) want[r * ld + c] = (double)S[r * ld + c];          // padding is left alone
    }
    const int rc = rows_per_seq ? vsx_softmax_rows_causal(S.data(), nrows, ncols, ld, rows_per_seq, nullptr)
                                : vsx_softmax_rows(S.data(), nrows, ncols, ld, nullptr);
    report(name, rc, want, dbl(S), 2e-3);
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    if (only < 0 || only == 0) run_groupnorm("GroupNorm 4-D scope 3 x 40 rows x 320, SiLU", 3, 40, 320, 0, 32, true);
    if (only < 0 || only == 1) run_groupnorm("GroupNorm two sources 2 x 24 x (192 + 128), groups straddle", 2, 24, 192, 128, 32, true);
    if (only < 0 || only == 2) run_groupnorm("GroupNorm 5-D scope 1 x 600 rows x 64 (many chunks), no SiLU", 1, 600, 64, 0, 32, false);
    if (only < 0 || only == 3) run_groupnorm("GroupNorm C = 2560 (320 vectors per row: one row per pass)", 1, 9, 1280, 1280, 32, true);
    if (only < 0 || only == 4) run_layernorm("LayerNorm 37 x 320", 37, 320, false, 0, 0, 0);
    if (only < 0 || only == 5) run_layernorm("LayerNorm 19 x 640", 19, 640, false, 0, 0, 0);
    if (only < 0 || only == 6) run_layernorm("LayerNorm 11 x 1280", 11, 1280, false, 0, 0, 0);
    if (only < 0 || only == 7) run_layernorm("LayerNorm + temporal PE 48 x 320 (6 rows per frame, 4 frames, offset 2)", 48, 320, true, 6, 4, 2);
    if (only < 0 || only == 8) run_softmax("softmax rows 21 x 77 (ld 80)", 21, 77, 80, 0);
    if (only < 0 || only == 9) run_softmax("softmax rows 9 x 256", 9, 256, 256, 0);
    if (only < 0 || only == 10) run_softmax("causal softmax 2 x 12 rows x 12 (ld 16)", 24, 12, 16, 12);
    printf(n_bad ? "%d check(s) FAILED\n" : "all checks passed\n", n_bad);
    return n_bad ? 1 : 0;
}
