// Do VALU instructions of one wave overlap with MFMAs of ANOTHER wave on the same SIMD?  The flash-attention kernel
// issues 448 cycles of MFMA and ~440 cycles of VALU (exp2 / max / cvt) per 64-key tile and measures their SUM
// (profiles/r01_probes/attention_timing.txt) although four waves share every SIMD.  This probe separates the cases:
//
//   mfma        every wave: NM independent 32x32x16 MFMAs per iteration
//   valu        every wave: NV VALU ops per iteration (fma chains + v_exp_f32 in the proportion of the softmax)
//   same-wave   every wave: both, interleaved (what the kernel does)
//   split       even waves of a SIMD run only the MFMAs, odd waves only the VALU work (same total work per SIMD as
//               `same-wave` with half the waves doing each kind): if the two pipes overlap across waves, `split` takes
//               max(mfma, valu) instead of their sum — the case for a warp-specialised attention kernel.
//
// One workgroup per CU, WAVES waves (WAVES / 4 per SIMD).  Cycles per iteration per SIMD from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int NM = 14;      // MFMAs per iteration (one 64-key tile of the d = 40 kernel)
constexpr int NEXP = 32;    // v_exp_f32 per iteration
constexpr int NFMA = 64;    // other VALU ops per iteration (max3 / fma / cvt stand-ins)

template <int MODE, int WAVES>   // 0 mfma, 1 valu, 2 same wave, 3 split by wave parity within a SIMD, 4 same wave interleaved
__global__ __launch_bounds__(64 * WAVES) void probe(float* __restrict__ sink, long* __restrict__ cycles, int iters) {
    // readfirstlane: the wave index must be UNIFORM for the compiler, otherwise `if (do_mfma)` is a divergent branch and
    // both sides execute under an EXEC mask (round-3 run 1 measured exactly that: `split` with 1 wave per SIMD, where
    // every wave only does MFMAs, took as long as `same-wave`)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves w, w+4, w+8 ... share SIMD w % 4; "parity within the SIMD" = (wave / 4) & 1
    const bool split = MODE == 3 || MODE == 5 || MODE == 6;
    const bool do_mfma = MODE == 0 || MODE == 2 || MODE == 4 || (split && ((wave >> 2) & 1) == 0);
    const bool do_valu = MODE == 1 || MODE == 2 || MODE == 4 || (split && ((wave >> 2) & 1) == 1);
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
    f16v acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.01f * (lane + e);
    __syncthreads();
    const long t0 = (long)__builtin_amdgcn_s_memtime();
    if (MODE == 3 || MODE == 5 || MODE == 6) {
        if (MODE == 5 && !do_mfma) __builtin_amdgcn_s_setprio(3);
        if (MODE == 6 && do_mfma) __builtin_amdgcn_s_setprio(3);
        // role fixed per wave OUTSIDE the loop: two tight loops (round-3 run 2: with the role tested inside one shared loop
        // an MFMA-only wave took 1145 ticks per iteration instead of 449 — the branch-around structure itself cost more
        // than the work, so that table said nothing about overlap)
        if (do_mfma) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < NEXP; ++k) v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7] * 0.5f - 1.0f);
#pragma unroll
                for (int k = 0; k < NFMA; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 0.999f, 0.001f * v[(k + 3) & 7]);
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
        }
        if (MODE == 4) {
            // program order forced to 1 MFMA : 2 exp : 5 fma groups (the in-order wave can issue its VALU work while its
            // own MFMA occupies the matrix pipe only if the instructions ALTERNATE in the instruction stream)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            }
        }
        if (do_valu) {
#pragma unroll
            for (int k = 0; k < NEXP; ++k) v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7] * 0.5f - 1.0f);
#pragma unroll
            for (int k = 0; k < NFMA; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 0.999f, 0.001f * v[(k + 3) & 7]);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const long t1 = (long)__builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int e = 0; e < 8; ++e) s += v[e];
    if (s == 123.456f) sink[threadIdx.x] = s;
    if (lane == 0) cycles[blockIdx.x * WAVES + wave] = t1 - t0;
}

static double g_role[2];

template <int MODE, int WAVES>
static double run(float* sink, long* cyc, int blocks, int iters) {
    hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, sink, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<long> h(blocks * WAVES);
    (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(long), hipMemcpyDeviceToHost);
    double mx = 0;
    g_role[0] = g_role[1] = 0;
    for (size_t i = 0; i < h.size(); ++i) {
        const double c = (double)h[i];
        mx = c > mx ? c : mx;
        const int role = ((i % WAVES) >> 2) & 1;                 // split mode: 0 = MFMA wave, 1 = VALU wave
        g_role[role] = c / iters > g_role[role] ? c / iters : g_role[role];
    }
    return mx / iters;          // s_memtime ticks
}

template <int WAVES>
static void table(float* sink, long* cyc, int blocks, int iters) {
    // s_memtime counts a fixed-frequency clock: report ratios, which are what matters
    const double m = run<0, WAVES>(sink, cyc, blocks, iters), v = run<1, WAVES>(sink, cyc, blocks, iters);
    const double b = run<2, WAVES>(sink, cyc, blocks, iters), s = run<3, WAVES>(sink, cyc, blocks, iters);
    const double rm = g_role[0], rv = g_role[1];
    const double il = run<4, WAVES>(sink, cyc, blocks, iters);
    const double s5 = run<5, WAVES>(sink, cyc, blocks, iters);
    const double rm5 = g_role[0], rv5 = g_role[1];
    const double s6 = run<6, WAVES>(sink, cyc, blocks, iters);
    const double rm6 = g_role[0], rv6 = g_role[1];
    // split does half of the `mfma` work and half of the `valu` work per SIMD: no overlap -> (mfma + valu) / 2,
    // perfect overlap across waves -> max(mfma, valu) / 2
    printf("%2d waves per CU (%d per SIMD): mfma %8.2f  valu %8.2f  same-wave %8.2f (%.2f x (mfma + valu))  "
           "interleaved same-wave %8.2f  split %8.2f [MFMA waves %.2f, VALU waves %.2f] (no overlap would be %.2f, perfect "
           "overlap %.2f)\n", WAVES, WAVES / 4, m, v, b, b / (m + v), il, s, rm, rv, WAVES >= 8 ? (m + v) / 2 : m,
           WAVES >= 8 ? (m > v ? m : v) / 2 : m);
    if (WAVES >= 8)
        printf("      split with the VALU waves at s_setprio 3: %8.2f [MFMA waves %.2f, VALU waves %.2f];  with the MFMA waves at "
               "s_setprio 3: %8.2f [MFMA waves %.2f, VALU waves %.2f]\n", s5, rm5, rv5, s6, rm6, rv6);
}

int main() {
    float* sink;
    long* cyc;
    const int blocks = 256, iters = 2000;
    (void)hipMalloc(&sink, 64 * 16 * sizeof(float));
    (void)hipMalloc(&cyc, blocks * 16 * sizeof(long));
    printf("# ticks per iteration (slowest wave); one iteration = %d MFMA 32x32x16 + %d v_exp + %d fma per wave that does that kind\n",
           NM, NEXP, NFMA);
    table<4>(sink, cyc, blocks, iters);
    table<8>(sink, cyc, blocks, iters);
    table<16>(sink, cyc, blocks, iters);
    return 0;
}
