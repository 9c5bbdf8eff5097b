// Runs the flash-attention backward kernels (videoswap_amd/csrc/attention_bwd.hip: dQ kernel, dK / dV kernel, delta kernel) on
// the CPU from their real source (tools/cpu_check/hip: a workgroup = OS threads, the 32x32x16 MFMA and the LDS-DMA with its
// descriptor range check emulated with the hardware's data layout) and compares with double-precision gradients of
// softmax(scale Q K^T) V.  What it covers: tile indexing, the row permutation of the LDS tiles against the C/D -> B operand
// register mapping, partial query / key tiles handled by zero fill, shared (text) K / V.  What it cannot see: anything about
// asynchronous ordering or speed.
#include "hip/hip_runtime.h"
#include "hip_gemm.h"
#undef smem

#include <stdarg.h>

#define __builtin_amdgcn_exp2f(x) exp2f(x)
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

#include "attention_bwd.hip"

static unsigned rng_state = 4242u;
static float frand() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}
static std::vector<half_t> randh(size_t n, float scale = 1.0f) {
    std::vector<half_t> v(n);
    for (auto& x : v) x = (half_t)(frand() * scale);
    return v;
}
static int n_bad = 0;
static double rel(const std::vector<double>& want, const std::vector<half_t>& got) {
    double num = 0, den = 0;
    for (size_t i = 0; i < want.size(); ++i) {
        const double d = (double)got[i] - want[i];
        num += d * d;
        den += want[i] * want[i];
    }
    return sqrt(num / (den > 0 ? den : 1));
}

static std::vector<half_t> transposed(const std::vector<half_t>& x, long nimg, long n, long C, long ld) {
    std::vector<half_t> t((size_t)nimg * C * ld, (half_t)0.f);
    for (long b = 0; b < nimg; ++b)
        for (long r = 0; r < n; ++r)
            for (long c = 0; c < C; ++c) t[(b * C + c) * ld + r] = x[(b * n + r) * C + c];
    return t;
}

static void run(const char* name, long nb, long kv_div, long heads, long nq, long nk, long d, bool need_kv) {
    const long C = heads * d, nkvb = nb / kv_div;
    const double scale = 1.0 / sqrt((double)d);
    auto Q = randh((size_t)nb * nq * C, 1.5f), K = randh((size_t)nkvb * nk * C, 1.5f), V = randh((size_t)nkvb * nk * C),
         dO = randh((size_t)nb * nq * C);
    const long lds = (nq + 63) / 64 * 64, ldtq = (nq + 7) / 8 * 8, ldtk = (nk + 7) / 8 * 8;
    std::vector<half_t> O((size_t)nb * nq * C);
    std::vector<float> lse((size_t)nb * heads * lds, 0.f), delta((size_t)nb * heads * lds, -7.f);
    std::vector<double> wdq((size_t)nb * nq * C, 0.0), wdk((size_t)nkvb * nk * C, 0.0), wdv((size_t)nkvb * nk * C, 0.0), p(nk), dp(nk);
    for (long b = 0; b < nb; ++b)
        for (long h = 0; h < heads; ++h)
            for (long q = 0; q < nq; ++q) {
                const long kb = b / kv_div;
                const half_t* qr = &Q[(b * nq + q) * C + h * d];
                const half_t* gr = &dO[(b * nq + q) * C + h * d];
                double mx = -1e30, sum = 0;
                for (long k = 0; k < nk; ++k) {
                    double s = 0;
                    for (long e = 0; e < d; ++e) s += (double)qr[e] * (double)K[(kb * nk + k) * C + h * d + e];
                    p[k] = s * scale;
                    mx = std::max(mx, p[k]);
                }
                for (long k = 0; k < nk; ++k) { p[k] = exp(p[k] - mx); sum += p[k]; }
                lse[(b * heads + h) * lds + q] = (float)((mx + log(sum)) * 1.4426950408889634);
                double dl = 0;
                std::vector<double> o(d, 0.0);
                for (long k = 0; k < nk; ++k) {
                    p[k] /= sum;
                    double a = 0;
                    for (long e = 0; e < d; ++e) {
                        o[e] += p[k] * (double)V[(kb * nk + k) * C + h * d + e];
                        a += (double)gr[e] * (double)V[(kb * nk + k) * C + h * d + e];
                    }
                    dp[k] = a;
                }
                for (long e = 0; e < d; ++e) {
                    O[(b * nq + q) * C + h * d + e] = (half_t)o[e];
                    dl += (double)gr[e] * (double)O[(b * nq + q) * C + h * d + e];     // delta from the ROUNDED output, as the kernel
                }
                for (long k = 0; k < nk; ++k) {
                    const double ds = scale * p[k] * (dp[k] - dl);
                    for (long e = 0; e < d; ++e) {
                        wdq[(b * nq + q) * C + h * d + e] += ds * (double)K[(kb * nk + k) * C + h * d + e];
                        wdk[(kb * nk + k) * C + h * d + e] += ds * (double)qr[e];
                        wdv[(kb * nk + k) * C + h * d + e] += p[k] * (double)gr[e];
                    }
                }
            }
    auto QT = transposed(Q, nb, nq, C, ldtq), KT = transposed(K, nkvb, nk, C, ldtk), dOT = transposed(dO, nb, nq, C, ldtq);
    std::vector<half_t> dQ((size_t)nb * nq * C, (half_t)-7.f), dK((size_t)nkvb * nk * C, (half_t)-7.f), dV((size_t)nkvb * nk * C, (half_t)-7.f);
    const int rc = vsx_attention_bwd_f16(Q.data(), K.data(), V.data(), O.data(), dO.data(), need_kv ? QT.data() : nullptr, KT.data(),
                                         need_kv ? dOT.data() : nullptr, lse.data(), delta.data(), dQ.data(),
                                         need_kv ? dK.data() : nullptr, need_kv ? dV.data() : nullptr, nb, heads, nq, nk, d, ldtq,
                                         ldtk, lds, kv_div, (float)scale, nullptr);
    const double eq = rel(wdq, dQ), ek = need_kv ? rel(wdk, dK) : 0, ev = need_kv ? rel(wdv, dV) : 0;
    const bool ok = rc == 0 && eq < 6e-3 && ek < 6e-3 && ev < 6e-3 && cpuhip_oob_reads == 0;
    printf("%-72s rc %d dQ %.2e dK %.2e dV %.2e %s%s\n", name, rc, eq, ek, ev, cpuhip_oob_reads ? "reads past a tensor " : "",
           ok ? "ok" : "FAIL");
    if (!ok) ++n_bad;
    cpuhip_oob_reads = 0;
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    if (only < 0 || only == 0) run("bwd d = 40, 2 heads, 200 x 200 (ragged query and key tiles)", 1, 1, 2, 200, 200, 40, true);
    if (only < 0 || only == 1) run("bwd d = 40, text K / V shared by 2 images, 150 x 77: dQ only", 2, 2, 2, 150, 77, 40, false);
    if (only < 0 || only == 2) run("bwd d = 80, 130 x 130", 1, 1, 1, 130, 130, 80, true);
    if (only < 0 || only == 3) run("bwd d = 64, 128 x 64 (exact tiles), 2 images", 2, 1, 1, 128, 64, 64, true);
    printf(n_bad ? "%d check(s) FAILED\n" : "all checks passed\n", n_bad);
    return n_bad ? 1 : 0;
}
