// Runs the attention kernels (videoswap_amd/csrc/attention.hip: flash attention on 32x32x16 MFMAs with LDS-DMA K / V^T tiles,
// the temporal attention with (site, head) problems packed into MFMA tiles, its long-clip form, and the VALU fallback) on the
// CPU from their real source and compares with a double-precision softmax(Q K^T) V.  The source is compiled as
// attention_cpu.hip = attention.hip with two textual rewrites made by the Makefile / test (tools/cpu_check/attention_cpu.sed):
//   * the `extern __shared__` declaration of the fallback kernel's dynamic LDS becomes a pointer to a static array
//     (block-scope `extern static` is not C++);
//   * a wave rendezvous in front of every V^T fragment read of the temporal kernels: a wave scatters the tile into its private
//     LDS slice and reads it back with no barrier — in order on the hardware, where the 64 lanes run in lockstep, a race here,
//     where they are threads.
#include "hip/hip_runtime.h"
#include "hip_gemm.h"
#undef smem                      // (hip_gemm.h maps the GEMM kernels' dynamic LDS; here the arrays are static __shared__)

#include <stdarg.h>
#include <string.h>

alignas(16) static unsigned char cpuhip_dyn_lds[160 * 1024];
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __log2f(x) log2f(x)
static inline int __all(int pred) {                      // wave vote (all 64 lanes take part, as on the hardware)
    const int lane = (int)(cpuhip::ctx.tid.x & 63);
    cpuhip::ctx.wave_slots[lane] = pred ? 1.f : 0.f;
    cpuhip::ctx.wave_bar->arrive_and_wait();
    int r = 1;
    for (int l = 0; l < 64; ++l) r &= cpuhip::ctx.wave_slots[l] != 0.f;
    cpuhip::ctx.wave_bar->arrive_and_wait();
    return r;
}

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

#include "attention_cpu.hip"


static unsigned rng_state = 31337u;
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
static void report(const char* name, int rc, const std::vector<double>& want, const std::vector<half_t>& got, double tol = 4e-3) {
    double num = 0, den = 0;
    for (size_t i = 0; i < want.size(); ++i) {
        const double d = (double)got[i] - want[i];
        num += d * d;
        den += want[i] * want[i];
    }
    const double rel = sqrt(num / (den > 0 ? den : 1));
    const bool ok = rc == 0 && rel < tol && cpuhip_oob_reads == 0;
    printf("%-64s rc %d rel-L2 %.2e %s%s\n", name, rc, rel, cpuhip_oob_reads ? "reads past a tensor " : "", ok ? "ok" : "FAIL");
    if (!ok) ++n_bad;
    cpuhip_oob_reads = 0;
}

// softmax(scale Q K^T) V per (batch, head); K / V batches are shared by kv_div query batches (CFG halves, frames of a clip)
static long g_attn_var = 0, g_attn_o16 = 1;      // (the product's default is 0; the checks run the 16-row form unless a case says otherwise)
namespace vsxg {
// the option table lives in gemm.hip: "attn_qb" (query blocks per wave) and "attn_o16" (16-row O^T tiles at d = 40) are asked here
long gemm_option(const char* name) { return strcmp(name, "attn_o16") == 0 ? g_attn_o16 : g_attn_var; }
}

static void run_flash(const char* name, long nb, long kv_div, long heads, long nq, long nk, long d) {
    const long C = heads * d, nkvb = nb / kv_div, ldvt = (nk + 7) / 8 * 8;
    const float scale = 1.0f / sqrtf((float)d);
    auto Q = randh((size_t)nb * nq * C, 1.5f), K = randh((size_t)nkvb * nk * C, 1.5f), V = randh((size_t)nkvb * nk * C);
    std::vector<half_t> VT((size_t)nkvb * C * ldvt, (half_t)0.f);
    for (long b = 0; b < nkvb; ++b)
        for (long k = 0; k < nk; ++k)
            for (long c = 0; c < C; ++c) VT[(b * C + c) * ldvt + k] = V[(b * nk + k) * C + c];
    std::vector<double> want((size_t)nb * nq * C), sc(nk);
    for (long b = 0; b < nb; ++b)
        for (long h = 0; h < heads; ++h)
            for (long q = 0; q < nq; ++q) {
                const long kb = b / kv_div;
                double mx = -1e30, sum = 0;
                for (long k = 0; k < nk; ++k) {
                    double s = 0;
                    for (long e = 0; e < d; ++e) s += (double)Q[(b * nq + q) * C + h * d + e] * (double)K[(kb * nk + k) * C + h * d + e];
                    sc[k] = s * scale;
                    mx = std::max(mx, sc[k]);
                }
                for (long k = 0; k < nk; ++k) { sc[k] = exp(sc[k] - mx); sum += sc[k]; }
                for (long e = 0; e < d; ++e) {
                    double o = 0;
                    for (long k = 0; k < nk; ++k) o += sc[k] * (double)V[(kb * nk + k) * C + h * d + e];
                    want[(b * nq + q) * C + h * d + e] = o / sum;
                }
            }
    std::vector<half_t> O((size_t)nb * nq * C, (half_t)-7.f);
    const int rc = vsx_attention_f16(Q.data(), K.data(), VT.data(), O.data(), nb, heads, nq, nk, d, C, C, ldvt, C, nq * C, nk * C,
                                     C * ldvt, nq * C, kv_div, scale, nullptr);
    report(name, rc, want, O);
}

// attention along the frame axis at every site: Q [B, fq, hw, heads*d], K / V [B, fk, hw, heads*d]
static void run_temporal(const char* name, long B, long fq, long fk, long hw, long heads, long d) {
    const long C = heads * d;
    const float scale = 1.0f / sqrtf((float)d);
    auto Q = randh((size_t)B * fq * hw * C, 1.5f), K = randh((size_t)B * fk * hw * C, 1.5f), V = randh((size_t)B * fk * hw * C);
    std::vector<double> want((size_t)B * fq * hw * C), sc(fk);
    for (long b = 0; b < B; ++b)
        for (long s = 0; s < hw; ++s)
            for (long h = 0; h < heads; ++h)
                for (long f = 0; f < fq; ++f) {
                    double mx = -1e30, sum = 0;
                    for (long g = 0; g < fk; ++g) {
                        double a = 0;
                        for (long e = 0; e < d; ++e)
                            a += (double)Q[((b * fq + f) * hw + s) * C + h * d + e] * (double)K[((b * fk + g) * hw + s) * C + h * d + e];
                        sc[g] = a * scale;
                        mx = std::max(mx, sc[g]);
                    }
                    for (long g = 0; g < fk; ++g) { sc[g] = exp(sc[g] - mx); sum += sc[g]; }
                    for (long e = 0; e < d; ++e) {
                        double o = 0;
                        for (long g = 0; g < fk; ++g) o += sc[g] * (double)V[((b * fk + g) * hw + s) * C + h * d + e];
                        want[((b * fq + f) * hw + s) * C + h * d + e] = o / sum;
                    }
                }
    std::vector<half_t> O((size_t)B * fq * hw * C, (half_t)-7.f);
    const int rc = vsx_temporal_attention_f16(Q.data(), K.data(), V.data(), O.data(), B, fq, fk, hw, heads, d, C, C, C, scale, nullptr);
    report(name, rc, want, O);
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    // d = 40 runs on 16-row O^T tiles (16 x 16 x 32 MFMAs, P^T relayout by v_permlane16_swap) unless attn_o16 = 0
    if (only < 0 || only == 0) run_flash("flash d = 40, 16-row O tiles, 2 heads, 200 x 200 (ragged query and key tiles)", 1, 1, 2, 200, 200, 40);
    if (only < 0 || only == 15) run_flash("flash d = 40, 16-row O tiles, 2 batches share K/V, 96 x 333 (6 key tiles: rescales)", 2, 2, 1, 96, 333, 40);
    g_attn_o16 = 0;
    if (only < 0 || only == 16) run_flash("flash d = 40, 32-row O tiles (attn_o16 = 0), 200 x 200", 1, 1, 2, 200, 200, 40);
    g_attn_o16 = 1;
    if (only < 0 || only == 1) run_flash("flash d = 80, cross-attention 150 x 77, K/V shared by 2 batches", 2, 2, 2, 150, 77, 80);
    if (only < 0 || only == 2) run_flash("flash d = 160, 64 x 64", 1, 1, 1, 64, 64, 160);
    if (only < 0 || only == 3) run_flash("flash d = 64 (CLIP / VAE head), 130 x 130", 1, 1, 2, 130, 130, 64);
    // 64 queries per wave (option attn_qb = 2: whatever the launch size)
    g_attn_var = 2;
    if (only < 0 || only == 12) run_flash("flash d = 40, 64 queries per wave, 300 x 200 (ragged)", 1, 1, 2, 300, 200, 40);
    if (only < 0 || only == 13) run_flash("flash d = 80, 64 queries per wave, 2 batches share K/V, 260 x 77", 2, 2, 2, 260, 77, 80);
    if (only < 0 || only == 14) run_flash("flash d = 40, 64 queries per wave, 256 x 128", 1, 1, 1, 256, 128, 40);
    g_attn_var = 0;
    if (only < 0 || only == 4) run_temporal("temporal d = 40, 8 heads, 16 x 16 frames, 3 sites, B = 2", 2, 16, 16, 3, 8, 40);
    if (only < 0 || only == 5) run_temporal("temporal d = 80, 8 heads, 16 x 16", 1, 16, 16, 2, 8, 80);
    if (only < 0 || only == 6) run_temporal("temporal d = 160, 8 heads, 24 x 24 (T = 24: one head per tile)", 1, 24, 24, 2, 8, 160);
    if (only < 0 || only == 7) run_temporal("temporal d = 40, 4 frames (8 heads per tile), 5 heads", 1, 4, 4, 2, 5, 40);
    if (only < 0 || only == 8) run_temporal("temporal long clip d = 40: 16 local query frames x 64 key frames", 1, 16, 64, 2, 8, 40);
    if (only < 0 || only == 9) run_temporal("temporal long clip d = 80: 8 x 40", 1, 8, 40, 1, 8, 80);
    if (only < 0 || only == 10) run_temporal("temporal fallback d = 24 (VALU kernel), 6 x 9 frames", 1, 6, 9, 2, 3, 24);
    printf(n_bad ? "%d check(s) FAILED\n" : "all checks passed\n", n_bad);
    return n_bad ? 1 : 0;
}
