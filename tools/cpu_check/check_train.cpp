// Runs the backward kernels of videoswap_amd/csrc/train.hip on the CPU (hip/hip_runtime.h next to this file)
// and compares every output with a plain double-precision restatement of the same formula.
//   make -C tools/cpu_check        (or: clang++ -std=c++20 -O1 -pthread -I tools/cpu_check -I include -I videoswap_amd/csrc ...)
#include <stdio.h>
#include <stdlib.h>

#include <vector>

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

#include "train.hip"

static unsigned rng_state = 12345u;
static float frand() {                      // uniform in [-1, 1)
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}
static std::vector<half_t> randh(size_t n, float scale = 1.0f, float shift = 0.0f) {
    std::vector<half_t> v(n);
    for (auto& x : v) x = (half_t)(frand() * scale + shift);
    return v;
}
static double gelu(double x) { return 0.5 * x * (1.0 + erf(x / sqrt(2.0))); }
static double dgelu(double x) { return 0.5 * (1.0 + erf(x / sqrt(2.0))) + x * exp(-0.5 * x * x) / sqrt(2.0 * M_PI); }
static double dsilu(double z) { const double s = 1.0 / (1.0 + exp(-z)); return s + z * s * (1.0 - s); }

static int n_bad = 0;
static void compare(const char* what, const std::vector<half_t>& got, const std::vector<double>& want, double tol) {
    double num = 0, den = 0;
    for (size_t i = 0; i < want.size(); ++i) {
        const double d = (double)got[i] - want[i];
        num += d * d;
        den += want[i] * want[i];
    }
    const double rel = sqrt(num / (den > 0 ? den : 1));
    printf("%-28s rel-L2 %.3e  %s\n", what, rel, rel < tol ? "ok" : "FAIL");
    if (!(rel < tol)) ++n_bad;
}

int main() {
    {   // GEGLU
        const long M = 37; const int N = 24;
        auto y2 = randh(M * 2 * N, 2.0f), dout = randh(M * N);
        std::vector<half_t> out(M * N), dy2(M * 2 * N);
        vsx_geglu_fwd(y2.data(), out.data(), M, N, nullptr);
        vsx_geglu_bwd(dout.data(), y2.data(), dy2.data(), M, N, nullptr);
        std::vector<double> wo(M * N), wd(M * 2 * N);
        for (long m = 0; m < M; ++m)
            for (int c = 0; c < N; ++c) {
                const double h = (double)y2[m * 2 * N + c], g = (double)y2[m * 2 * N + N + c], d = (double)dout[m * N + c];
                wo[m * N + c] = h * gelu(g);
                wd[m * 2 * N + c] = d * gelu(g);
                wd[m * 2 * N + N + c] = d * h * dgelu(g);
            }
        compare("geglu_fwd", out, wo, 2e-3);
        compare("geglu_bwd", dy2, wd, 2e-3);
    }
    {   // SiLU backward
        const long n = 1000;
        auto x = randh(n, 3.0f), dy = randh(n);
        std::vector<half_t> dx(n);
        vsx_silu_bwd(dy.data(), x.data(), dx.data(), n, nullptr);
        std::vector<double> w(n);
        for (long i = 0; i < n; ++i) w[i] = (double)dy[i] * dsilu((double)x[i]);
        compare("silu_bwd", dx, w, 2e-3);
    }
    for (int variant = 0; variant < 3; ++variant) {   // GroupNorm backward: plain / +SiLU / +SiLU + concat
        const long nimg = 2, rows = 23; const int C1 = variant == 2 ? 16 : 24, C2 = variant == 2 ? 8 : 0, groups = 4;
        const int C = C1 + C2, cpg = C / groups; const int silu = variant > 0; const float eps = 1e-5f;
        auto x1 = randh(nimg * rows * C1, 1.5f, 0.3f), x2 = randh(nimg * rows * (C2 ? C2 : 1)), dy = randh(nimg * rows * C);
        auto gamma = randh(C, 0.3f, 1.0f), beta = randh(C, 0.2f);
        std::vector<half_t> dx1(nimg * rows * C1), dx2(nimg * rows * (C2 ? C2 : 1));
        std::vector<float> ws(vsx_groupnorm_bwd_workspace(nimg, rows, groups));
        vsx_groupnorm_bwd(dy.data(), x1.data(), C2 ? x2.data() : nullptr, nimg, rows, C1, C2, groups, gamma.data(),
                          beta.data(), eps, silu, ws.data(), dx1.data(), C2 ? dx2.data() : nullptr, nullptr);
        std::vector<double> w1(nimg * rows * C1), w2(nimg * rows * (C2 ? C2 : 1));
        auto X = [&](long i, long r, int c) { return c < C1 ? (double)x1[(i * rows + r) * C1 + c] : (double)x2[(i * rows + r) * C2 + c - C1]; };
        for (long i = 0; i < nimg; ++i)
            for (int g = 0; g < groups; ++g) {
                const double n = (double)rows * cpg;
                double mean = 0, var = 0, s1 = 0, s2 = 0;
                for (long r = 0; r < rows; ++r) for (int c = g * cpg; c < (g + 1) * cpg; ++c) mean += X(i, r, c);
                mean /= n;
                for (long r = 0; r < rows; ++r) for (int c = g * cpg; c < (g + 1) * cpg; ++c) var += (X(i, r, c) - mean) * (X(i, r, c) - mean);
                const double rstd = 1.0 / sqrt(var / n + eps);
                auto GZ = [&](long r, int c) {
                    const double xh = (X(i, r, c) - mean) * rstd;
                    double gz = (double)dy[(i * rows + r) * C + c];
                    if (silu) gz *= dsilu(xh * (double)gamma[c] + (double)beta[c]);
                    return gz * (double)gamma[c];
                };
                for (long r = 0; r < rows; ++r) for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s1 += GZ(r, c); s2 += GZ(r, c) * (X(i, r, c) - mean) * rstd; }
                s1 /= n; s2 /= n;
                for (long r = 0; r < rows; ++r)
                    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                        const double v = rstd * (GZ(r, c) - s1 - (X(i, r, c) - mean) * rstd * s2);
                        if (c < C1) w1[(i * rows + r) * C1 + c] = v; else w2[(i * rows + r) * C2 + c - C1] = v;
                    }
            }
        const char* names[3] = {"groupnorm_bwd", "groupnorm_bwd +silu", "groupnorm_bwd +silu +concat"};
        compare(names[variant], dx1, w1, 3e-3);
        if (C2) compare("   (second source)", dx2, w2, 3e-3);
    }
    {   // LayerNorm backward
        const long M = 11; const int C = 72; const float eps = 1e-5f;
        auto x = randh(M * C, 2.0f, 0.5f), dy = randh(M * C), gamma = randh(C, 0.3f, 1.0f);
        std::vector<half_t> dx(M * C);
        vsx_layernorm_bwd(dy.data(), x.data(), gamma.data(), eps, dx.data(), M, C, nullptr);
        std::vector<double> w(M * C);
        for (long m = 0; m < M; ++m) {
            double mean = 0, var = 0, s1 = 0, s2 = 0;
            for (int c = 0; c < C; ++c) mean += (double)x[m * C + c];
            mean /= C;
            for (int c = 0; c < C; ++c) var += ((double)x[m * C + c] - mean) * ((double)x[m * C + c] - mean);
            const double rstd = 1.0 / sqrt(var / C + eps);
            for (int c = 0; c < C; ++c) { const double gz = (double)dy[m * C + c] * (double)gamma[c]; s1 += gz; s2 += gz * ((double)x[m * C + c] - mean) * rstd; }
            s1 /= C; s2 /= C;
            for (int c = 0; c < C; ++c) {
                const double xh = ((double)x[m * C + c] - mean) * rstd;
                w[m * C + c] = rstd * ((double)dy[m * C + c] * (double)gamma[c] - s1 - xh * s2);
            }
        }
        compare("layernorm_bwd", dx, w, 3e-3);
    }
    {   // softmax backward on a row-padded buffer
        const long nrows = 9; const int ncols = 13; const long ld = 16; const float scale = 0.25f;
        std::vector<half_t> P(nrows * ld, (half_t)0.f), dP(nrows * ld, (half_t)0.f);
        std::vector<double> w(nrows * ld, 0.0);
        for (long r = 0; r < nrows; ++r) {
            double sum = 0; std::vector<double> e(ncols);
            for (int c = 0; c < ncols; ++c) { e[c] = exp(2.0 * frand()); sum += e[c]; }
            for (int c = 0; c < ncols; ++c) { P[r * ld + c] = (half_t)(e[c] / sum); dP[r * ld + c] = (half_t)frand(); }
            double dot = 0;
            for (int c = 0; c < ncols; ++c) dot += (double)P[r * ld + c] * (double)dP[r * ld + c];
            for (int c = 0; c < ncols; ++c) w[r * ld + c] = scale * (double)P[r * ld + c] * ((double)dP[r * ld + c] - dot);
        }
        vsx_softmax_bwd(P.data(), dP.data(), nrows, ncols, ld, scale, nullptr);
        compare("softmax_bwd (+ padding kept)", dP, w, 3e-3);
    }
    {   // 2x2 sum pool
        const long n = 2; const int h = 3, w = 5, c = 16;
        auto x = randh(n * 2 * h * 2 * w * c);
        std::vector<half_t> y(n * h * w * c);
        vsx_sum_pool2x2(x.data(), y.data(), n, h, w, c, nullptr);
        std::vector<double> want(n * h * w * c);
        for (long i = 0; i < n; ++i) for (int yy = 0; yy < h; ++yy) for (int xx = 0; xx < w; ++xx) for (int k = 0; k < c; ++k) {
            double s = 0;
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) s += (double)x[((i * 2 * h + 2 * yy + a) * 2 * w + 2 * xx + b) * c + k];
            want[((i * h + yy) * w + xx) * c + k] = s;
        }
        compare("sum_pool2x2", y, want, 2e-3);
    }
    {   // adapter gather
        const int F = 3, P = 4, C = 40, h = 6, w = 7; const float rate = 8.f, out_scale = 0.5f;
        std::vector<float> tracks(F * P * 2);
        for (auto& t : tracks) t = (frand() * 0.5f + 0.5f) * 50.f;
        tracks[(1 * P + 2) * 2] = -1.f;
        std::vector<int32_t> sel = {1, 0, 1, 1};
        auto dmap = randh((size_t)F * h * w * C);
        std::vector<half_t> dfeat(P * C, (half_t)0.f);
        vsx_adapter_gather(tracks.data(), sel.data(), dmap.data(), dfeat.data(), F, P, C, h, w, rate, out_scale, nullptr);
        std::vector<double> want(P * C, 0.0);
        auto r16 = [](double v) { return (double)(half_t)(float)v; };
        for (int p = 0; p < P; ++p) {
            if (!sel[p]) continue;
            for (int f = 0; f < F; ++f) {
                const double px = tracks[(f * P + p) * 2], py = tracks[(f * P + p) * 2 + 1];
                if (px < 0 || py < 0) continue;
                const double x = r16(r16(px) / rate), y = r16(r16(py) / rate);
                int x1 = (int)x, y1 = (int)y, x2 = x1 + 1, y2 = y1 + 1;
                const double xf = r16(x - x1), yf = r16(y - y1);
                x1 = std::max(std::min(x1, w - 1), 0); x2 = std::max(std::min(x2, w - 1), 0);
                y1 = std::max(std::min(y1, h - 1), 0); y2 = std::max(std::min(y2, h - 1), 0);
                const double xm = r16(1 - xf), ym = r16(1 - yf);
                const double wg[4] = {r16(xm * ym), r16(xf * ym), r16(xm * yf), r16(xf * yf)};
                const int xs[4] = {x1, x2, x1, x2}, ys[4] = {y1, y1, y2, y2};
                for (int k = 0; k < 4; ++k)
                    for (int c = 0; c < C; ++c) want[p * C + c] += wg[k] * (double)dmap[((size_t)(f * h + ys[k]) * w + xs[k]) * C + c];
            }
            for (int c = 0; c < C; ++c) want[p * C + c] *= out_scale;
        }
        compare("adapter_gather", dfeat, want, 2e-3);
    }
    printf(n_bad ? "%d check(s) FAILED\n" : "all checks passed\n", n_bad);
    return n_bad ? 1 : 0;
}
