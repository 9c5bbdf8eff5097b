// Shared between gemm.hip (workgroup-per-tile kernels) and gemm_pp.hip (persistent ping-pong kernel): the launch
// parameters and the LDS-DMA helpers.
#pragma once
#include "common.h"

namespace vsxg {

constexpr int BK = 64;          // K slab (halfs); LDS rows are 128 B = 8 16-byte slots = one full L2 line per row

struct GemmParams {
    const half_t* A;
    const half_t* A2;
    const half_t* B;
    half_t* C;
    const half_t* bias;
    const half_t* rowvec;
    const half_t* residual;
    // LayerNorm folded into the GEMM (vsx.h: rowscale / colvec): out = rs[m] * acc + rt[m] * c1[n] (+ bias ...)
    const float* rowscale;      // [M][2] = (rstd, -rstd * mean) of the A rows, or nullptr
    const float* colvec;        // [N] (geglu: [2N]) = sum_k B[n][k]
    float* rowstats;            // [M][rowstats_parts][2] partial (sum, sum of squares) of the rounded outputs, or nullptr
    int rowstats_parts;
    long M, N, K;
    long lda, ldb, ldc, ldr;
    long a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1, r_bs0, r_bs1;
    long c_rows_per_img, c_img_stride, rows_per_vec;
    int batch1;
    int a_mode, H, W, C1, C2, Ho, Wo, ks, stride, ups, pad;     // pad: zero rows / columns before the image
    int geglu, c_mode, c_pack4, vec4, vec8, rvec8;
    int tiles_n;
    unsigned a_bytes, a2_bytes, b_bytes;   // buffer extents (per batch slice) for the SRD bounds check
    int splitk, nk_per;                    // split-K: grid.z = splitk slices of nk_per slabs, fp32 partials to `ws`
    float* ws;
    float alpha;
    int tiles_total;                       // persistent kernel: number of output tiles
    int pp_flags;                          // persistent kernel: PP_* option bits (tile walk)
    int sp_Mc;                             // sub-pixel form of the nearest-2x convolution (upsample = 2): GEMM rows per (ph, pw) class, else 0
    // workgroup-per-tile kernels: XCD block grid (gemm.hip, "XCD BLOCK GRID").  xcd_gm = 0: the linear walk (grid.z = K slices)
    int xcd_gm, tiles_m;
};

// pp_flags: option bits of "pp_sched" / VSX_PP_SCHED.  Round 3 measured five candidates on the GPU
// (profiles/r03_gemm_option_ab*_b2.txt, r03_gemm_sched_ab_b2.txt) and kept one: the 2-D tile walk inside an XCD is the
// default (same speed, 2.5x less HBM traffic on the wide-N GEMMs); 8 restores the linear walk for A/B runs.  Dropped:
// other cut points of the DMA piece schedule (+-1 %), conv slab order "taps of a channel slab back to back" with a full
// address recomputation per slab (8-13 % slower; round 4's loader keeps three VALU operations per piece and slab and made
// that order the default, PP_CONV_TAP_MAJOR below), no
// s_setprio / s_setprio on the LOAD phase (+-1 %), K-start stagger per workgroup (+5-9 % in tools/ubench/gemm_loop.hip,
// -6 ... +10 % in the kernel: no net gain), start-time stagger of the workgroups by quarters of a tile period, so that
// the epilogues (HBM writes) of some CUs fall under the main loops of others (profiles/r03_gemm_stagger_ab_b2.txt: +-2 %
// at K = 320, slower everywhere else: the CUs are not in lockstep to begin with).
constexpr int PP_TILES_LINEAR = 8;
// Convolution K order of the persistent kernel.  Default since round 4: the ks * ks taps of one 64-channel slab back to back —
// their windows are shifts of the same input rows, so the taps after the first hit L2 and the input of a tile is fetched from
// the fabric once instead of once per tap: 64x64 / 32x32 convolutions 3.8 - 7.1x -> 1.3 - 2.0x their algorithmic bytes, the
// launch mix 1.88x -> 1.67x, at the same GEMM time per forward (profiles/r04_gemm_traffic_by_shape*.txt; single convolutions
// 0 - 3 % slower in isolation, profiles/r04_gemm_conv_order_ab_b2.txt).  Bit 4 restores the tap-major order (all channel slabs
// of a tap, then the next tap), which sums K in the order of the tile kernels: bit-identical results, used by the equality
// tests and for A/B runs.
constexpr int PP_CONV_TAP_MAJOR = 4;
// Shared A slab of the stride-1 3x3 convolutions (gemm_pp.hip, "SHARED A SLAB": the three taps of a filter row read one
// A slab at three row offsets).  On wherever the shape allows it; option bit 16 keeps a private A slab per tap (A/B runs,
// the equality test).  PP_CONV_ASHIFT_ON is set by launch_pp, never by the option.
constexpr int PP_CONV_PRIVATE_A = 16;
constexpr int PP_CONV_ASHIFT_ON = 1 << 16;
// The A (and, in the plain GEMMs, B) pieces of a slab are issued from a per-CU starting point (gemm_pp.hip, "PIECE ROTATION");
// bit 32 restores the common order (A/B runs).
constexpr int PP_COMMON_ORDER = 32;
// Workgroup-per-tile kernels: bit 64 of the same option issues the residual prefetch in front of the K loop again (round 4's
// placement; A/B runs).  Default: behind the last slab's pieces (gemm.hip, "RES_LATE").
constexpr int TILE_RES_EARLY = 64;

typedef __attribute__((address_space(3))) void* lptr_t;

// tools/cpu_check: the host model of the vmcnt queue is told about ordinary loads that stay in flight across counted waits
#ifndef VSX_VMEM_NOTE
#define VSX_VMEM_NOTE(n) do { } while (0)
#endif

#ifdef __HIPCC__
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#endif  // __HIPCC__

// Tuning / test switches (vsx_set_option; initial values from the environment): "gemm_pp" (VSX_GEMM_PP: 0 = never use
// the persistent kernel, 1 = where it is expected to win, 2 = wherever the shape is eligible), "pp_sched"
// (VSX_PP_SCHED: PP_* bits).
long gemm_option(const char* name);

// gemm_pp.hip: persistent ping-pong kernel (256x320 / 128x320 tiles).  `bm` selects the row tile.
bool pp_supported(const GemmParams& p);
bool pp_rowstats_ok(const GemmParams& p);   // can the persistent kernel's epilogue of this problem emit row statistics?
int launch_pp(GemmParams& p, int bm, hipStream_t stream);
// gemm_pp.hip: weight-stationary kernel for the byte-bound K = N = 320 projections (32-row blocks streamed past register-resident weights)
bool ws_supported(const GemmParams& p);
int launch_ws(GemmParams& p, hipStream_t stream);
int ws_waves();                              // 5 or 10: waves per workgroup = row-statistics parts per 320 columns

}  // namespace vsxg
