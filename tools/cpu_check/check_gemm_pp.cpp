// Runs the persistent ping-pong GEMM (videoswap_amd/csrc/gemm_pp.hip) on the CPU from its real source — the 2-D and the
// linear tile walk, both tile heights, every epilogue kind (plain / addend ring / folded LayerNorm / GEGLU) — and
// compares with a double-precision GEMM; the walks must also agree bit for bit, and the 2-D tile walk
// must be a permutation of the tiles.  See hip_gemm.h for what the emulation covers (addressing, LDS layout, MFMA fragment layout, epilogue) and
// what it cannot (the asynchronous ordering of the LDS-DMA).
#define CPUHIP_DYNAMIC_LDS_ONLY
#include "hip/hip_runtime.h"
#include "hip_gemm.h"

#include <stdarg.h>

#include "common.h"

int vsx_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
    return code;
}
int vsx_check_launch(const char*) { return 0; }

static long g_sched = 0;
#include "gemm_common.h"
namespace vsxg {
long gemm_option(const char*) { return g_sched; }
namespace {
CPUHIP_DEFINE_LDS
}
}
#include "gemm_pp.hip"

static unsigned rng_state = 777u;
static float frand() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}
static std::vector<half_t> randh(size_t n, float scale = 1.0f) {
    std::vector<half_t> v(n);
    for (auto& x : v) x = (half_t)(frand() * scale);
    return v;
}
static double gelu(double x) { return 0.5 * x * (1.0 + erf(x / sqrt(2.0))); }

struct Case { const char* name; long M, N, K; bool res, geglu; int bm; bool ln = false; long vt = 0; };     // vt: rows per image of the transposed store (c_mode 1), 0 = row-major C

static int n_bad = 0;
static std::vector<long> g_scheds = {0, 8, 4, 12, 16, 32};      // pp_sched bits: 8 = linear tile walk instead of the 2-D one; 4 = conv K order tap-major (default: taps of a channel slab back to back); 16 = a private A slab per tap (default where the shape allows: one A slab per filter row, read at three row offsets); 32 = every CU issues the pieces of a slab in the same order (default: from a per-CU starting point)

static std::vector<half_t> pack_b(const std::vector<half_t>& w, long rows, long K) {      // [N/8][K/64][8][64]
    std::vector<half_t> out(w.size());
    const long nk = K / 64;
    for (long r = 0; r < rows; ++r)
        for (long k = 0; k < K; ++k)
            out[((r / 8) * nk + k / 64) * 512 + (r % 8) * 64 + (k % 64)] = w[r * K + k];
    return out;
}

static void run_case(const Case& c) {
    using namespace vsxg;
    const long brows = c.geglu ? 2 * c.N : c.N;
    auto A = randh(c.M * c.K), B = randh(brows * c.K, 1.0f / sqrtf((float)c.K)), bias = randh(brows);
    auto R = randh(c.M * c.N);
    if (getenv("CPUHIP_ONES")) {
        for (long m = 0; m < c.M; ++m) for (long k = 0; k < c.K; ++k) A[m * c.K + k] = (half_t)(k == atoi(getenv("CPUHIP_ONES")) ? 1.f : 0.f);
        for (long n = 0; n < brows; ++n) for (long k = 0; k < c.K; ++k) B[n * c.K + k] = (half_t)(float)(k);
        for (auto& b : bias) b = (half_t)0.f;
    }
    // LayerNorm folded into the GEMM (rowscale / colvec): out = rs[m] * acc + rt[m] * c1[n] + bias ...
    std::vector<float> rowscale, colvec;
    if (c.ln) {
        rowscale.resize(2 * c.M);
        colvec.resize(brows);
        for (long m = 0; m < c.M; ++m) { rowscale[2 * m] = 0.5f + 0.01f * (float)(m % 37); rowscale[2 * m + 1] = -0.3f + 0.02f * (float)(m % 11); }
        for (long n = 0; n < brows; ++n) { double s = 0; for (long k = 0; k < c.K; ++k) s += (double)B[n * c.K + k]; colvec[n] = (float)s; }
    }
    // reference
    std::vector<double> want(c.M * c.N);
    for (long m = 0; m < c.M; ++m)
        for (long n = 0; n < c.N; ++n) {
            auto dot = [&](long row) {
                double s = 0;
                for (long k = 0; k < c.K; ++k) s += (double)A[m * c.K + k] * (double)B[row * c.K + k];
                if (c.ln) s = (double)rowscale[2 * m] * s + (double)rowscale[2 * m + 1] * (double)colvec[row];
                return s + (double)bias[row];
            };
            double v = c.geglu ? dot(n) * gelu(dot(c.N + n)) : dot(n);
            if (c.res) v += (double)R[m * c.N + n];
            // transposed store: V^T[img][n][row in image], image stride N * rows_per_img (ldvt = rows_per_img here)
            want[c.vt ? ((m / c.vt) * c.N + n) * c.vt + m % c.vt : m * c.N + n] = v;
        }
    std::vector<half_t> first;
    for (long sched : g_scheds) {
        std::vector<half_t> C(c.M * c.N, (half_t)-7.f);
        GemmParams p{};
        p.A = A.data(); p.B = B.data(); p.C = C.data(); p.bias = bias.data();
        p.residual = c.res ? R.data() : nullptr;
        p.rowscale = c.ln ? rowscale.data() : nullptr;
        p.colvec = c.ln ? colvec.data() : nullptr;
        p.M = c.M; p.N = c.N; p.K = c.K;
        p.lda = c.K; p.ldb = c.K; p.ldc = c.N; p.ldr = c.N;
        p.batch1 = 1; p.rows_per_vec = 1; p.geglu = c.geglu; p.alpha = 1.0f;
        p.vec4 = p.vec8 = 1; p.rvec8 = c.res ? 1 : 0;
        p.a_bytes = (unsigned)(((c.M - 1) * c.K + c.K) * 2);
        p.b_bytes = (unsigned)(((brows - 1) * c.K + c.K) * 2);
        p.splitk = 1;
        if (c.vt) { p.c_mode = 1; p.ldc = c.vt; p.c_rows_per_img = c.vt; p.c_img_stride = c.N * c.vt; p.c_pack4 = 1; }
        if (!pp_supported(p)) { printf("%s: not eligible\n", c.name); return; }
        // row statistics of the output (EPI_STATS: plain / addend epilogues only): 6 parts per 320 columns
        const bool stats = pp_rowstats_ok(p);
        const long nparts = stats ? (c.N / 320) * 6 : 0;
        std::vector<float> parts((size_t)c.M * nparts * 2, -1.f);
        if (stats) { p.rowstats = parts.data(); p.rowstats_parts = (int)nparts; }
        g_sched = sched;
        cpuhip_oob_reads = 0;
        const int rc = launch_pp(p, c.bm, nullptr);
        double num = 0, den = 0;
        for (size_t i = 0; i < want.size(); ++i) {
            const double d = (double)C[i] - want[i];
            num += d * d;
            den += want[i] * want[i];
        }
        double rel = sqrt(num / den);
        if (stats) {            // every part = (sum, sum of squares) of the ROUNDED outputs over its columns: widths 64, 64, 32 per 160
            double worst = 0;
            for (long m = 0; m < c.M; ++m)
                for (long q = 0; q < nparts; ++q) {
                    const long col0 = (q / 3) * 160 + (q % 3) * 64, w = (q % 3) == 2 ? 32 : 64;
                    double s1 = 0, s2 = 0;
                    for (long n = col0; n < col0 + w; ++n) { const double v = (double)C[m * c.N + n]; s1 += v; s2 += v * v; }
                    worst = std::max(worst, fabs((double)parts[(m * nparts + q) * 2] - s1) / (1.0 + fabs(s1)));
                    worst = std::max(worst, fabs((double)parts[(m * nparts + q) * 2 + 1] - s2) / (1.0 + fabs(s2)));
                }
            if (worst > 1e-5) { printf("%s: row statistics off by %.2e\n", c.name, worst); rel = 1.0; }
        }
        bool same = true;
        if (first.empty()) first = C;
        else same = memcmp(first.data(), C.data(), C.size() * sizeof(half_t)) == 0;
        if (getenv("CPUHIP_ROW")) {
            const long m = atol(getenv("CPUHIP_ROW"));
            for (long n = 0; n < c.N; ++n) printf("%g%s", (double)C[m * c.N + n], (n % 32 == 31) ? "\n" : " ");
            exit(0);
        }
        if (getenv("CPUHIP_DEBUG") && rel > 3e-3) {
            int shown = 0;
            for (long m = 0; m < c.M && shown < 24; ++m)
                for (long n = 0; n < c.N && shown < 24; ++n) {
                    const double d = (double)C[m * c.N + n] - want[m * c.N + n];
                    if (fabs(d) > 0.05 + 0.02 * fabs(want[m * c.N + n])) {
                        printf("   m %ld n %ld got %.3f want %.3f\n", m, n, (double)C[m * c.N + n], want[m * c.N + n]);
                        ++shown;
                    }
                }
        }
        const bool ok = rc == 0 && rel < 3e-3 && same && cpuhip_oob_reads == 0;
        printf("%-34s bm %3d sched %2ld: rc %d rel-L2 %.2e %s%s%s\n", c.name, c.bm, sched, rc, rel,
               same ? "" : "DIFFERS from schedule 0 ", cpuhip_oob_reads ? "reads past the tensor " : "", ok ? "ok" : "FAIL");
        if (!ok) ++n_bad;
    }
}

// 3x3 convolution (implicit GEMM, two concatenated sources, optional stride 2 / nearest-2x input) + bias + row vector
static void run_conv(const char* name, int nimg, int H, int W, int C1, int C2, int Cout, int stride, int ups, int bm,
                     bool with_rowvec = true) {
    using namespace vsxg;
    const int ks = 3, pad = 1;
    const int Hs = ups ? H / 2 : H, Ws = ups ? W / 2 : W;               // stored resolution of the sources
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    const long M = (long)nimg * Ho * Wo, K = (long)ks * ks * (C1 + C2);
    auto X1 = randh((size_t)nimg * Hs * Ws * C1), X2 = randh((size_t)nimg * Hs * Ws * (C2 ? C2 : 1));
    auto Wt = randh((size_t)Cout * K, 1.0f / sqrtf((float)K)), bias = randh(Cout), rowvec = randh((size_t)nimg * Cout);
    std::vector<double> want((size_t)M * Cout);
    for (int i = 0; i < nimg; ++i)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo)
                for (int co = 0; co < Cout; ++co) {
                    double s = (double)bias[co] + (with_rowvec ? (double)rowvec[(size_t)i * Cout + co] : 0.0);
                    for (int kh = 0; kh < ks; ++kh)
                        for (int kw = 0; kw < ks; ++kw) {
                            const int h = ho * stride - pad + kh, w = wo * stride - pad + kw;
                            if (h < 0 || h >= H || w < 0 || w >= W) continue;
                            const int hs = ups ? h / 2 : h, wsrc = ups ? w / 2 : w;
                            const half_t* wrow = Wt.data() + (size_t)co * K + (size_t)(kh * ks + kw) * (C1 + C2);
                            const size_t pix = ((size_t)i * Hs + hs) * Ws + wsrc;
                            for (int c = 0; c < C1; ++c) s += (double)X1[pix * C1 + c] * (double)wrow[c];
                            for (int c = 0; c < C2; ++c) s += (double)X2[pix * C2 + c] * (double)wrow[C1 + c];
                        }
                    want[((size_t)(i * Ho + ho) * Wo + wo) * Cout + co] = s;
                }
    std::vector<half_t> firsts[2];              // per K order (pp_sched & 4: tap-major instead of taps-inner): another fp32 summation order
    for (long sched : g_scheds) {
        std::vector<half_t>& first = firsts[(sched >> 2) & 1];
        std::vector<half_t> C((size_t)M * Cout, (half_t)-7.f);
        GemmParams p{};
        p.A = X1.data(); p.A2 = C2 ? X2.data() : nullptr; p.B = Wt.data(); p.C = C.data();
        p.bias = bias.data(); p.rowvec = with_rowvec ? rowvec.data() : nullptr; p.rows_per_vec = with_rowvec ? (long)Ho * Wo : 1;
        p.M = M; p.N = Cout; p.K = K; p.ldb = K; p.ldc = Cout; p.batch1 = 1; p.alpha = 1.0f;
        p.a_mode = 1; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2; p.Ho = Ho; p.Wo = Wo; p.ks = ks; p.stride = stride;
        p.ups = ups; p.pad = pad;
        p.vec4 = p.vec8 = 1;
        p.a_bytes = (unsigned)((size_t)nimg * Hs * Ws * C1 * 2);
        p.a2_bytes = (unsigned)((size_t)nimg * Hs * Ws * C2 * 2);
        p.b_bytes = (unsigned)((size_t)Cout * K * 2);
        p.splitk = 1;
        if (!pp_supported(p)) { printf("%s: not eligible\n", name); return; }
        g_sched = sched;
        cpuhip_oob_reads = 0;
        const int rc = launch_pp(p, bm, nullptr);
        double num = 0, den = 0;
        for (size_t i = 0; i < want.size(); ++i) {
            const double d = (double)C[i] - want[i];
            num += d * d;
            den += want[i] * want[i];
        }
        const double rel = sqrt(num / den);
        bool same = true;
        if (first.empty()) first = C;
        else same = memcmp(first.data(), C.data(), C.size() * sizeof(half_t)) == 0;
        const bool ok = rc == 0 && rel < 3e-3 && same && cpuhip_oob_reads == 0;
        printf("%-34s bm %3d sched %2ld: rc %d rel-L2 %.2e %s%s%s%s\n", name, bm, sched, rc, rel,
               (p.pp_flags & PP_CONV_ASHIFT_ON) ? "(shared A slab) " : "",
               same ? "" : "DIFFERS from the first schedule ", cpuhip_oob_reads ? "reads past the tensor " : "",
               ok ? "ok" : "FAIL");
        if (!ok) ++n_bad;
    }
}

// Sub-pixel form of the nearest-2x convolution (upsample = 2): four classes of output pixels, each a 2x2-tap window on the
// source with its own [N, 9 C] weight matrix (only the class's four taps are read); one GEMM over M = 4 Mc rows whose
// epilogue scatters row (class, i, j) to pixel (2 i + ph, 2 j + pw).
static void run_subpixel(const char* name, int nimg, int Hs, int Ws, int C, int Cout, int bm) {
    using namespace vsxg;
    const long Mc = (long)nimg * Hs * Ws, M = 4 * Mc, K9 = 9L * C;
    auto X = randh((size_t)nimg * Hs * Ws * C);
    auto Wt = randh((size_t)4 * Cout * K9, 1.0f / sqrtf((float)(4 * C))), bias = randh(Cout);
    std::vector<double> want((size_t)M * Cout);
    for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        for (int im = 0; im < nimg; ++im)
            for (int i = 0; i < Hs; ++i)
                for (int j = 0; j < Ws; ++j)
                    for (int co = 0; co < Cout; ++co) {
                        double s = (double)bias[co];
                        for (int kh = ph; kh < ph + 2; ++kh)
                            for (int kw = pw; kw < pw + 2; ++kw) {
                                const int h = i + kh - 1, w = j + kw - 1;
                                if (h < 0 || h >= Hs || w < 0 || w >= Ws) continue;
                                const half_t* wrow = Wt.data() + ((size_t)cls * Cout + co) * K9 + (size_t)(kh * 3 + kw) * C;
                                const size_t pix = ((size_t)im * Hs + h) * Ws + w;
                                for (int c = 0; c < C; ++c) s += (double)X[pix * C + c] * (double)wrow[c];
                            }
                        want[(((size_t)im * 2 * Hs + 2 * i + ph) * 2 * Ws + 2 * j + pw) * Cout + co] = s;
                    }
    }
    std::vector<half_t> first;
    for (long sched : {0L, 8L, 32L}) {
        std::vector<half_t> C_((size_t)M * Cout, (half_t)-7.f);
        GemmParams p{};
        p.A = X.data(); p.B = Wt.data(); p.C = C_.data(); p.bias = bias.data();
        p.M = M; p.N = Cout; p.K = 4L * C; p.ldb = K9; p.ldc = Cout; p.batch1 = 1; p.alpha = 1.0f;
        p.a_mode = 1; p.H = Hs; p.W = Ws; p.C1 = C; p.C2 = 0; p.Ho = Hs; p.Wo = Ws; p.ks = 3; p.stride = 1; p.ups = 0; p.pad = 1;
        p.vec4 = p.vec8 = 1; p.rows_per_vec = 1;
        p.a_bytes = (unsigned)(X.size() * 2);
        p.b_bytes = (unsigned)(Wt.size() * 2);
        p.splitk = 1;
        p.sp_Mc = (int)Mc;
        if (!pp_supported(p) || Mc % bm) { printf("%s: not eligible\n", name); ++n_bad; return; }
        g_sched = sched;
        cpuhip_oob_reads = 0;
        const int rc = launch_pp(p, bm, nullptr);
        double num = 0, den = 0;
        for (size_t i = 0; i < want.size(); ++i) {
            const double d = (double)C_[i] - want[i];
            num += d * d;
            den += want[i] * want[i];
        }
        const double rel = sqrt(num / den);
        bool same = true;
        if (first.empty()) first = C_;
        else same = memcmp(first.data(), C_.data(), C_.size() * sizeof(half_t)) == 0;
        const bool ok = rc == 0 && rel < 3e-3 && same && cpuhip_oob_reads == 0;
        printf("%-34s bm %3d sched %2ld: rc %d rel-L2 %.2e %s%s%s\n", name, bm, sched, rc, rel,
               same ? "" : "DIFFERS from the first schedule ", cpuhip_oob_reads ? "reads past the tensor " : "", ok ? "ok" : "FAIL");
        if (!ok) ++n_bad;
    }
}

int main(int argc, char** argv) {
    const Case cases[] = {
        {"plain 256x320x64", 256, 320, 64, false, false, 256},                 // one tile, one slab
        {"plain 640x320x192 (+res)", 640, 320, 192, true, false, 256},        // 3 tiles incl. a ragged one, 3 slabs
        {"plain 1536x640x128", 1536, 640, 128, false, false, 256},            // 12 tiles on 8 workgroups: 2 tiles each
        {"K tail 512x320x72", 512, 320, 72, true, false, 256},                // 2 slabs, the second 8 wide
        {"geglu 512x160x128", 512, 160, 128, false, true, 256},
        {"plain 384x320x64 (+res) 128-row", 384, 320, 64, true, false, 128},  // one slab per tile
        {"geglu 300x320x192 128-row", 300, 320, 192, false, true, 128},
        {"wide 768x3840x64 (2-D walk, 36 tiles)", 768, 3840, 64, false, false, 256},     // tiles_n = 12: blocks of 8 x 4
        {"rowscale 600x320x128 (+res)", 600, 320, 128, true, false, 256, true},         // LayerNorm identity, ragged M
        {"rowscale geglu 300x160x64 128-row", 300, 160, 64, false, true, 128, true},
        {"rowscale 700x320x64", 700, 320, 64, false, false, 256, true},                  // LayerNorm identity alone
        {"rowscale 200x640x128 (+res) 128-row", 200, 640, 128, true, false, 128, true},
        {"plain 300x320x64 128-row", 300, 320, 64, false, false, 128},                  // the remaining kernel kinds
        {"rowscale 300x320x64 128-row", 300, 320, 64, false, false, 128, true},
        {"rowscale geglu 512x160x64", 512, 160, 64, false, true, 256, true},
    };
    // transposed store of the persistent kernel (round 5; cases 100, 101 so that the numbering above stays): V^T[img][n][row],
    // 256 / 512 rows per image, plain and with the LayerNorm identity, several tiles per workgroup
    const Case vt_cases[] = {
        {"V^T 512x320x128, 256 rows per image", 512, 320, 128, false, false, 256, false, 256},
        {"V^T rowscale 1024x640x64, 512 rows per image", 1024, 640, 64, false, false, 256, true, 512},
    };
    // usage: check_gemm_pp [case index | -1 = all] [comma-separated schedules]
    cpuhip_num_cus = 24;                  // three workgroups per emulated XCD: blockIdx.x >> 3 takes the values 0, 1, 2
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    if (argc > 2) {
        g_scheds.clear();
        for (char* tok = strtok(argv[2], ","); tok; tok = strtok(nullptr, ",")) g_scheds.push_back(atol(tok));
    }
    const int ncases = (int)(sizeof(cases) / sizeof(cases[0]));
    for (int i = 0; i < ncases; ++i)
        if (only < 0 || only == i) run_case(cases[i]);
    for (int i = 0; i < 2; ++i)
        if (only < 0 || only == 100 + i) run_case(vt_cases[i]);
    if (only < 0 || only == ncases) run_conv("conv3x3 2x8x8 64+64->320", 2, 8, 8, 64, 64, 320, 1, 0, 256);
    if (only < 0 || only == ncases + 1) run_conv("conv3x3 3x16x8 128->320 /s2 128-row", 3, 16, 8, 128, 0, 320, 2, 0, 128);    // 32 rows per vector
    if (only < 0 || only == ncases + 2) run_conv("conv3x3 2x8x8 64->320 nearest-2x", 2, 8, 8, 64, 0, 320, 1, 1, 128);
    if (only < 0 || only == ncases + 3) run_conv("conv3x3 1x16x16 128+64->320", 1, 16, 16, 128, 64, 320, 1, 0, 256);
    if (only < 0 || only == ncases + 4)      // 48 rows per vector: 32-row blocks that meet two row vectors
        run_conv("conv3x3 3x6x8 64->320", 3, 6, 8, 64, 0, 320, 1, 0, 256);
    if (only < 0 || only == ncases + 5) run_conv("conv3x3 2x8x8 64->320, bias only", 2, 8, 8, 64, 0, 320, 1, 0, 256, false);
    if (only < 0 || only == ncases + 6) run_conv("conv3x3 2x8x8 64->320, bias only, 128-row", 2, 8, 8, 64, 0, 320, 1, 0, 128, false);
    // shared A slab (W a power of two >= 32): image rows of 32 / 64 pixels, several channel slabs, two sources, a ragged last tile
    if (only < 0 || only == ncases + 7) run_conv("conv3x3 1x16x32 64->320", 1, 16, 32, 64, 0, 320, 1, 0, 256);
    if (only < 0 || only == ncases + 8) run_conv("conv3x3 2x8x32 128+64->320", 2, 8, 32, 128, 64, 320, 1, 0, 256);
    if (only < 0 || only == ncases + 9) run_conv("conv3x3 1x8x64 64->320", 1, 8, 64, 64, 0, 320, 1, 0, 256);
    if (only < 0 || only == ncases + 10) run_conv("conv3x3 1x12x32 128->320 128-row", 1, 12, 32, 128, 0, 320, 1, 0, 128);
    if (only < 0 || only == ncases + 11) run_conv("conv3x3 3x4x32 64->320 (ragged)", 3, 4, 32, 64, 0, 320, 1, 0, 256, false);
    if (only < 0 || only == ncases + 12) run_conv("conv3x3 1x6x64 64+64->320 128-row", 1, 6, 64, 64, 64, 320, 1, 0, 128);
    // image rows as long as half a tile / a whole tile: the edge lanes are those of the tile's first and last 32-row block only
    if (only < 0 || only == ncases + 15) run_conv("conv3x3 1x4x128 64->320", 1, 4, 128, 64, 0, 320, 1, 0, 256);
    if (only < 0 || only == ncases + 16) run_conv("conv3x3 1x3x256 64->320 (ragged)", 1, 3, 256, 64, 0, 320, 1, 0, 256, false);
    if (only < 0 || only == ncases + 17) run_conv("conv3x3 1x5x128 64->320 128-row", 1, 5, 128, 64, 0, 320, 1, 0, 128);
    // sub-pixel form of the nearest-2x convolution: 2 tiles per class (8 x 32 source), two channel slabs; one 128-row tile per class
    if (only < 0 || only == ncases + 18) run_subpixel("subpixel 2x8x32 128->320", 2, 8, 32, 128, 320, 256);
    if (only < 0 || only == ncases + 19) run_subpixel("subpixel 1x8x16 64->320 128-row", 1, 8, 16, 64, 320, 128);
    if (only < 0 || only == ncases + 20) run_subpixel("subpixel 1x14x24 64->640 (W = 24), 128-row", 2, 8, 24, 64, 640, 128);
    // ... and three tiles per workgroup (8 workgroups, 24 tiles): the A ring's parity across tile boundaries for an odd and an
    // even number of A slabs per tile, next to the epilogue's staging area
    if (only == ncases + 13 || only == ncases + 14) {
        cpuhip_num_cus = 8;
        if (only == ncases + 13) run_conv("conv3x3 1x192x32 64->320, 3 tiles/wg", 1, 192, 32, 64, 0, 320, 1, 0, 256);
        else run_conv("conv3x3 1x96x64 64+64->320, 3 tiles/wg", 1, 96, 64, 64, 64, 320, 1, 0, 256);
        cpuhip_num_cus = 24;
    }
    if (only < 0) {          // the 2-D tile walk visits every tile exactly once (any tile count, ragged last super-row)
        int bad = 0;
        for (int tn : {1, 2, 6, 8, 12, 16, 24, 32, 40, 44})
            for (int tm = 1; tm <= 41; ++tm) {
                std::vector<int> seen(tn * tm, 0);
                for (int t = 0; t < tn * tm; ++t) {
                    int a = -1, b = -1;
                    vsxg::tile_coords(t, tn, tm, true, a, b);
                    if (a < 0 || a >= tm || b < 0 || b >= tn || seen[a * tn + b]++) ++bad;
                }
            }
        printf("2-D tile walk is a permutation for all (tiles_m, tiles_n) tried: %s\n", bad ? "FAIL" : "ok");
        n_bad += bad ? 1 : 0;
    }
    printf(n_bad ? "%d check(s) FAILED\n" : "all checks passed\n", n_bad);
    return n_bad ? 1 : 0;
}
