// K1/K2 (big problems) — persistent "ping-pong" fp16 MFMA GEMM / implicit-GEMM convolution for gfx950.
//
// Same contract, operand staging (LDS-DMA with buffer descriptors, 128-byte-row K slabs of 64 halfs, XOR-swizzled LDS
// rows, implicit im2col / stride / nearest-2x / two-source A loader) and epilogue arithmetic as gemm.hip; what differs
// is the work decomposition and the main-loop schedule, both chosen from the measurements of the
// workgroup-per-tile kernel (profiles/r01_*):
//
//  * 8 waves (2 per SIMD), each owning a (TM*32) x 160 block of the 256x320 (TM = 2) or 128x320 (TM = 1) tile:
//    with 64x160 per wave a k-step reads 7 KiB of fragments for 10 MFMAs (the 16-wave 32x160 layout read 6 KiB for
//    5): the tile's LDS read traffic drops from 384 to 224 KiB per slab, and the 160 accumulator registers fit the
//    256-register budget of 2 waves per SIMD.
//  * PING-PONG: the workgroup is two groups of four waves (one wave of each group per SIMD).  Every k-step is a
//    LOAD phase (fragment ds_reads for that k-step + this wave's share of the next slab's LDS-DMA pieces) followed
//    by an MFMA phase (10 back-to-back MFMAs), with one s_barrier after every phase.  Group 1 runs one phase behind
//    group 0, so at any time one wave of a SIMD feeds the matrix pipe while its partner does everything else; the
//    DMA-issue stalls (a 1-KiB piece blocks its wave for 60-180 cycles) and the LDS latency never sit in front of an
//    MFMA of the same wave.
//  * PERSISTENT: one workgroup per CU walks its XCD's contiguous range of output tiles; the operand stream (slabs
//    in order: tile after tile) runs one slab ahead of the MFMAs ACROSS tile boundaries, so the first slab of the
//    next tile lands under the epilogue of the current one.  K = 320 problems (five slabs per tile) no longer pay a
//    cold prologue per tile.
//
// Phase timeline (|B| = workgroup barrier; L/M = load / MFMA phase of k-step s of slab t):
//     G0:  L0 |B| M0 |B| L1 |B| M1 |B| L2 |B| M2 |B| L3 |B| M3 |B| L0' ...
//     G1:     |B| L0 |B| M0 |B| L1 |B| M1 |B| L2 |B| M2 |B| L3 |B| M3  ...
// Ring of two slabs.  Slab t+1 is written into the slot slab t-1 occupied; its last reader is G1's L3(t-1), which
// has waited for its ds_reads (lgkmcnt(0)) before the barrier that precedes G0's L0(t), the first phase that issues
// a piece of slab t+1.  Every wave waits for its own pieces (vmcnt(0)) before the barrier that precedes G0's L0(t+1):
// G0 at the end of M3(t), G1 at the end of L3(t).
// At a tile boundary both groups run the epilogue in the SAME barrier interval and then resynchronise
// (G0: |B| E |B| L0' |B| ..., G1: M3 E |B| |B| L0' ...).
//
// Register budget (256 per lane, nothing may spill: a scratch reload is a VMEM op and would queue behind the DMA
// pieces in flight): 160 accumulators + 28 fragment registers; row offsets are NOT kept per piece — a piece's row base
// is wave-uniform and travels in the scalar offset of the buffer load, the lane keeps only (row-in-piece, k-slot)
// offsets; the launch parameters are read from the kernarg segment where they are used (s_load) instead of living
// in SGPRs across the tile loop.
#include "gemm_common.h"

#include <stdlib.h>

// -DVSX_PP_ABLATE=<bits> (python -m videoswap_amd.build --define <name> -DVSX_PP_ABLATE=n -> lib/libvsx_<name>.so, tools/gemm_ab.py --libs):
// ablation builds that answer "what does the epilogue cost" by leaving parts of it out — 1: no global stores of C, 2: no GELU
// (GEGLU writes h * g).  WRONG RESULTS by construction; never defined in the product build (profiles/r06_gemm_epilogue_ablation.txt).
// (Leaving the whole epilogue out is not a measurement: the compiler then drops the MFMAs whose accumulators nobody reads.)
#ifndef VSX_PP_ABLATE
#define VSX_PP_ABLATE 0
#endif

namespace vsxg {
#ifdef VSX_GEMM_TIMING
// -DVSX_GEMM_TIMING (tools/gemm_timing.py pp ...): per-wave cycle totals long[workgroup][wave][4] = main loop (all K slabs of
// all tiles), epilogue, barrier after the epilogue, tiles
__device__ long* g_pp_dbg = nullptr;
#endif
namespace {

typedef const __attribute__((address_space(4))) GemmParams* kparams_t;
typedef float f2v __attribute__((ext_vector_type(2)));

// The lane id from the execution mask instead of from a register that would have to stay live: v_mbcnt_lo / _hi count the
// lanes below this one (all lanes are active wherever this is called).  `volatile`: never hoisted, never merged with an
// earlier copy — which is the point (see epilogue_pp).  tools/cpu_check strips asm statements: there the hint passes through.
__device__ __forceinline__ int fresh_lane(const int lane_hint) {
    int l = lane_hint;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ kparams_t kernarg_params() {
    kparams_t kp = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();   // GemmParams is the first kernel argument
    asm volatile("" : "+s"(kp));                 // opaque: fields are (re)loaded where they are used
    return kp;
}

// Epilogue of one wave: TM x 5 C^T accumulator tiles (acc[j][i]: lane (l31, hi) owns output row mrow0 + i*32 + l31;
// register quad g of tile (j, i) holds the 4 consecutive columns ncol0 + j*32 + 8g + 4hi ..).
//
// Writing C straight from that layout costs more than the main loop of a K <= 1280 problem: every store instruction
// touches 32 different rows with 2 x 16 bytes each (64 partial lines; measured 4.5 B/clk/CU, 36 k cycles per 256x320
// tile against 13 k for its five K slabs), and the residual is read the same way.  So the tile goes through a
// wave-private LDS staging area (fp32, 32 rows x 64 columns at a time, rows padded by 16 B: conflict-free
// ds_write_b128 from the fragment layout) and is read back ROW-MAJOR: a lane then owns 8 consecutive columns of one
// row, 8 lanes cover one full 128-byte line, and bias / row vector / residual are added in that layout — same
// fp32 operation order as gemm.hip's epilogue, so the results stay bit-identical to the tile kernels.
// The host routes a problem here only when every row / column octet is 16-byte aligned and N is a multiple of 160
// (GEGLU) / 320, so there is no column edge; rows >= M are skipped.
constexpr int EP_BYTES = 10240;     // staging bytes per wave: 32 rows x (64 + 4) floats = 8704

// Tile index -> (tile_m, tile_n).  Narrow problems (and PP_TILES_LINEAR): N fastest.  tiles_n a multiple of 4 and > 8: the walk goes
// block by block through super-rows of RB row tiles, a block being RB x CB tiles (RB * CB = 32 = the workgroups an XCD
// runs at a time), so that at any moment an XCD's CUs share RB activation panels and CB weight panels in L2 instead of
// 1 and 32 (geglu 1280->5120: every XCD used to stream the whole 26 MB weight matrix once per row tile: 945 MB of HBM
// traffic per launch against 131 MB algorithmic, 378 MB with this walk at the same speed — profiles/r03_gemm_traffic_by_shape_*.txt).
__device__ __forceinline__ void tile_coords(const int tile, const int tiles_n, const int tiles_m, const bool blocked,
                                            int& tile_m, int& tile_n) {
    const int CB = (tiles_n & 7) == 0 ? 8 : 4;
    if (!blocked || tiles_n <= 8 || (tiles_n % CB) != 0) {
        tile_n = tile % tiles_n;
        tile_m = tile / tiles_n;
        return;
    }
    const int RB = 32 / CB;
    const int per_sr = RB * tiles_n;
    const int sr = tile / per_sr;
    const int rem = tile - sr * per_sr;
    const int rows = min(RB, tiles_m - sr * RB);
    const int cb = rem / (rows * CB);
    const int within = rem - cb * rows * CB;
    const int r = within / CB;
    tile_m = sr * RB + r;
    tile_n = cb * CB + (within - r * CB);
}

// ---- What the epilogue may NOT do: wait for a global load it has just issued.  vmcnt counts loads and stores in issue
// order, so a wait for the newest load is vmcnt(0): it also waits for every store before it to be acknowledged.  The first
// staged epilogue loaded bias, row vector and residual where it used them, pass by pass: 24 x "load, wait for it and
// for the last store, store" per tile with 16 bytes per lane in flight (measured: the +res epilogue of a K = 320 tile took
// twice as long as its five K slabs, ~2 TB/s of residual reads chip-wide).  Now:
//  * the per-COLUMN constants of the wave's 160 columns (bias, and c1 of a folded LayerNorm) are loaded once per tile,
//    together with the per-row LayerNorm pairs and ahead of everything else, and parked as fp32 in the slack of the
//    wave's staging area; the chunks read them from LDS;
//  * the ADDEND (residual, or the row vector when there is no residual) is fetched through a ring of PD (4; 3 in the
//    convolution kernel, 2 beside the LayerNorm registers: what fits without spilling) 16-byte registers per lane.  A
//    wave's epilogue is a flat sequence of 10 * TM row passes (per 32-row block: 4 + 4 passes of 8 rows x 64 columns,
//    then 2 passes of 16 rows x 32 columns); pass f consumes ring entry f % PD and refills it with the addend of pass
//    f + PD, so the wait before a pass is vmcnt(PD - 1 + stores since) and PD KiB per wave stay in flight.
constexpr int EP_CONST = 8704 / 4;  // float offset of the column constants in a wave's staging area: [160] bias, [160] c1
static_assert(8704 + 2 * 160 * 4 <= EP_BYTES, "column constants fit behind the staged chunk");

struct PreSrc {
    const half_t* src;      // residual or row vector
    bool is_rowvec;
    unsigned pitch;         // ldr or N
    unsigned v0[2], bnd[2]; // row vector: index of the vector of a 32-row block's first row, first row of the next vector
    int mrow0, ncol0, Mlast;
};

template <int PD>
__device__ __forceinline__ void pre_fetch(const PreSrc& ps, const int lane, const int f, h8* pre) {
    const int i = f / 10, r = f % 10;
    const int c = r < 4 ? 0 : (r < 8 ? 1 : 2);
    const int pass = r - 4 * c;
    const int lpr = c < 2 ? 8 : 4, rpp = c < 2 ? 8 : 16;
    const int n = ps.ncol0 + c * 64 + (lane % lpr) * 8;
    const unsigned m = (unsigned)min(ps.mrow0 + i * 32 + pass * rpp + lane / lpr, ps.Mlast);   // rows past M: the last row
    // row vector: rows_per_vec >= 32, so the block meets at most two vectors (no per-lane division)
    const unsigned row = ps.is_rowvec ? ps.v0[i] + (m >= ps.bnd[i] ? 1u : 0u) : m;
    pre[f % PD] = *reinterpret_cast<const h8*>(ps.src + (row * ps.pitch + (unsigned)n));
}

// One staged chunk (32 rows x CW columns, fp32, row-major in `stg`) -> C.  cst: the wave's column constants + the float
// offset of this chunk's first column in them (nullptr: no bias / LayerNorm, the GEGLU path has applied them).  PD > 0:
// a prefetched addend exists (fbase: flat index of the chunk's first pass, nf: passes of the whole epilogue).  LN:
// LayerNorm folded into the GEMM — lane l holds (rstd, -rstd * mean) of row mblk + l % 32 in st_rs / st_rt; a pass
// fetches its row's pair with two cross-lane reads.  Same fp32 operation order as gemm.hip's epilogue (scale, LayerNorm
// identity, bias, row vector | residual).
// STATS: the pass also writes (sum, sum of squares) of its row's ROUNDED outputs over the chunk's columns into
// p->rowstats[m][part] (vsx.h, ABI 8): the LPR lanes that share a row add up with log2(LPR) cross-lane steps.
// SUBPIX (sub-pixel form of the nearest-2x convolution, gemm_pp_kernel CONV = 3): GEMM row m = class * Mc + r is the output
// pixel (2 i + ph, 2 j + pw) of source pixel r = (image row i', j): output row 4 Ws i' + 2 Ws ph + 2 j + pw.
template <int CW, int PD, bool LN, bool STATS = false, bool SUBPIX = false>   // CW: chunk width in output columns: 64, 32 or 16; PD: ring depth (0: no addend)
__device__ __forceinline__ void epilogue_rows(kparams_t p, const float* stg, const int lane, const int mblk,
                                              const int ncol, const float* cst, const PreSrc& ps, h8* pre,
                                              const int fbase, const int nf, const float st_rs = 1.f,
                                              const float st_rt = 0.f, const int part = 0) {
    constexpr bool ADD = PD > 0;
    constexpr int STRIDE = CW + 4;                  // floats per staged row
    constexpr int LPR = CW / 8;                     // lanes per row (8 columns each)
    constexpr int RPP = 64 / LPR;                   // rows per pass
    const int Mi = (int)p->M;
    const int col8 = (lane % LPR) * 8, r0 = lane / LPR;
    const int n = ncol + col8;
    float bv[8], cv[LN ? 8 : 1];
    if (cst) {
        const f4v b0 = *reinterpret_cast<const f4v*>(cst + col8), b1 = *reinterpret_cast<const f4v*>(cst + col8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[e + 4] = b1[e]; }
        if constexpr (LN) {
            const f4v c0 = *reinterpret_cast<const f4v*>(cst + 160 + col8), c1 = *reinterpret_cast<const f4v*>(cst + 164 + col8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { cv[e] = c0[e]; cv[e + 4] = c1[e]; }
        }
    }
    half_t* cp = p->C;
    const unsigned ldc = (unsigned)p->ldc;
#pragma unroll
    for (int pass = 0; pass < 32 / RPP; ++pass) {
        const int row = pass * RPP + r0;
        const int m = mblk + row;
        const f4v a = *reinterpret_cast<const f4v*>(stg + row * STRIDE + col8);
        const f4v b = *reinterpret_cast<const f4v*>(stg + row * STRIDE + col8 + 4);
        float o[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        if constexpr (LN) {
            const float rs = __shfl(st_rs, row, 64), rt = __shfl(st_rt, row, 64);   // (all lanes take part, also those past M)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(rs, o[e], rt * cv[e]);
        }
        if (cst) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += bv[e];
        }
        if constexpr (ADD) {
            const h8 add = pre[(fbase + pass) % PD];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += (float)add[e];
            if (fbase + pass + PD < nf) pre_fetch<PD>(ps, lane, fbase + pass + PD, pre);
        }
        h8 pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[e] = (half_t)o[e];
        if constexpr (SUBPIX) {
            const unsigned Mc = (unsigned)p->sp_Mc, Ws = (unsigned)p->Wo;
            const unsigned cls = (unsigned)m / Mc, r = (unsigned)m - cls * Mc;
            const unsigned q = r / Ws, j = r - q * Ws;
            const unsigned orow = 4u * Ws * q + 2u * j + 2u * Ws * (cls >> 1) + (cls & 1u);
            if (m < Mi) *reinterpret_cast<h8*>(cp + (orow * ldc + (unsigned)n)) = pk;
        } else if ((VSX_PP_ABLATE & 1) ? m < -Mi : m < Mi) *reinterpret_cast<h8*>(cp + ((unsigned)m * ldc + (unsigned)n)) = pk;
        if constexpr (STATS) {
            // v_dot2_f32_f16 on the packed pairs: sum and sum of squares of the rounded values without converting them back
            // (8 instructions per pass instead of 24; fp16 x fp16 products are exact in fp32)
            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
            const h2v one2 = {(half_t)1.f, (half_t)1.f};
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const h2v v = {pk[e], pk[e + 1]};
                s1 = __builtin_amdgcn_fdot2(v, one2, s1, false);
                s2 = __builtin_amdgcn_fdot2(v, v, s2, false);
            }
#pragma unroll
            for (int d = LPR / 2; d > 0; d >>= 1) {       // the LPR lanes of a row are adjacent: fixed-shape tree
                s1 += __shfl_xor(s1, d, 64);
                s2 += __shfl_xor(s2, d, 64);
            }
            if (col8 == 0 && m < Mi)
                *reinterpret_cast<f2v*>(p->rowstats + ((size_t)m * (unsigned)p->rowstats_parts + (unsigned)part) * 2) = f2v{s1, s2};
        }
    }
}

// EPI: what the epilogue does besides scale and bias — compiled per combination, so that a launch carries only the
// registers and branches of its own epilogue (the K loop runs at 253 of 256 registers).
constexpr int EPI_ADD = 1;          // + residual, or + row vector (one of the two: prefetched ring)
constexpr int EPI_LN = 2;           // LayerNorm folded into the GEMM (rowscale / colvec)
constexpr int EPI_GEGLU = 4;        // h * gelu(g)
constexpr int EPI_STATS = 8;        // row statistics of the output for a LayerNorm that follows (not with LN / GEGLU)
constexpr int EPI_VT = 16;          // transposed store (c_mode 1: V^T for the attention kernels); bias / folded LayerNorm only, 256-row tiles

// Transposed epilogue of one wave (TM = 2, round 5).  C^T[n][row in image] is what the attention kernels read (vsx.h, c_mode 1); a wave's
// 64 rows are exactly ONE 128-byte line of every V^T row n of its 160 columns, and a 256-row tile lies inside one image (the host
// checks rows_per_img % 256 == 0).  Writing that from the accumulator layout costs 32 different V^T rows x 8 bytes per store
// instruction — the workgroup-per-tile kernels do, and run the V^T projections at 0.27 - 0.47 PF/s where the same GEMM with a
// row-major C reaches 0.46 - 0.73.  Here a chunk of 32 columns goes through the wave's staging area TRANSPOSED ([n][m], fp32) and is
// read back with a lane owning 8 consecutive m of one n: 8 lanes store one full line.  Pitch 68 floats and a lane -> (n, octet)
// assignment that follows the hardware's 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32): every
// group reads two n rows of opposite parity, 8 octets each, i.e. all 64 banks once.  Same fp32 operation order as the tile kernels'
// transposed store (scale, LayerNorm identity, bias): bit-identical results.
template <bool LN>
__device__ __forceinline__ void epilogue_vt(f16v (&acc)[5][2], float* stg, const int mrow0, const int ncol0, const int lane) {
    kparams_t p = kernarg_params();
    constexpr int PITCH = 68;
    static_assert(32 * PITCH * 4 <= EP_BYTES, "the transposed chunk fits the wave's staging area");
    const int l31 = lane & 31, hi = lane >> 5;
    const float alpha = p->alpha;
    // read-back coordinates of this lane: position inside its 16-lane ds_read_b128 group -> (n row of the pass, octet of m)
    const bool in_a = l31 < 4 || (l31 >= 12 && l31 < 16) || (l31 >= 20 && l31 < 28);
    const int pos = in_a ? (l31 < 4 ? l31 : (l31 < 16 ? l31 - 8 : l31 - 12)) : (l31 < 12 ? l31 - 4 : (l31 < 20 ? l31 - 8 : l31 - 16));
    const int nl = 2 * (2 * hi + (in_a ? 0 : 1)) + (pos >> 3);       // 0 .. 7
    const int m8 = (pos & 7) * 8;
    float rs[LN ? 8 : 1], rt[LN ? 8 : 1];
    if constexpr (LN) {
        const float* rsp = p->rowscale + 2 * (size_t)(unsigned)(mrow0 + m8);       // (rstd, -rstd * mean) of this lane's 8 rows
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4v v = *reinterpret_cast<const f4v*>(rsp + 4 * q);
            rs[2 * q] = v[0]; rt[2 * q] = v[1]; rs[2 * q + 1] = v[2]; rt[2 * q + 1] = v[3];
        }
    }
    const half_t* bias = p->bias;
    const float* cvp = p->colvec;
    const unsigned rpi = (unsigned)p->c_rows_per_img;
    const unsigned img = (unsigned)mrow0 / rpi;                         // wave-uniform: the tile lies inside one image
    half_t* base = p->C + (long)img * p->c_img_stride + ((unsigned)mrow0 - img * rpi) + m8;
    const long ldc = p->ldc;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    stg[(8 * g + 4 * hi + e) * PITCH + i * 32 + l31] = acc[j][i][4 * g + e] * alpha;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int nloc = pass * 8 + nl;
            const int n = ncol0 + j * 32 + nloc;
            const f4v a = *reinterpret_cast<const f4v*>(stg + nloc * PITCH + m8);
            const f4v b = *reinterpret_cast<const f4v*>(stg + nloc * PITCH + m8 + 4);
            float o[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            if constexpr (LN) {
                const float cvn = cvp[n];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(rs[e], o[e], rt[e] * cvn);
            }
            const float bv = bias ? (float)bias[n] : 0.f;
            h8 pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = (half_t)(o[e] + bv);
            *reinterpret_cast<h8*>(base + (long)n * ldc) = pk;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int TM, int EPI, int PD, bool SUBPIX = false>
__device__ __forceinline__ void epilogue_pp(f16v (&acc)[5][TM], float* stg, const int mrow0, const int ncol0,
                                            const int gcol0, const int lane_in) {
    // Everything the epilogue derives from the lane id (4 * lane, 16 * lane, column octets ...) is loop-invariant over the
    // tile loop: the compiler hoists it in front of the K loop, runs out of registers there and SPILLS it — and the reload
    // in here is a VMEM load whose wait (vmcnt(0)) also waits for the addend ring (round 5's code objects: 1 - 5 such
    // values in the kinds with an addend; videoswap_amd/build.py refuses a library whose hand-scheduled kernels touch
    // scratch).  A lane id recomputed per tile (fresh_lane) keeps those values local to the epilogue: a handful of VALU operations
    // per tile, and not even the lane id itself has to survive the K loop.
    const int lane = fresh_lane(lane_in);
    constexpr bool ADD = (EPI & EPI_ADD) != 0, LN = (EPI & EPI_LN) != 0, GEGLU = (EPI & EPI_GEGLU) != 0;
    constexpr bool STATS = (EPI & EPI_STATS) != 0;
    static_assert(!STATS || (!LN && !GEGLU), "row statistics come from the plain / addend epilogues");
    if constexpr ((EPI & EPI_VT) != 0) {
        static_assert(TM == 2 && !ADD && !GEGLU && !STATS && !SUBPIX, "the transposed store: 256-row tiles, bias / folded LayerNorm only");
        epilogue_vt<LN>(acc, stg, mrow0, ncol0, lane);
        return;
    }
    kparams_t p = kernarg_params();
    // statistics part of this wave's first chunk: 6 parts per 320 columns = (column half wc) x (chunks of 64, 64, 32 columns)
    const int part0 = STATS ? (ncol0 / 160) * 3 : 0;
    const int l31 = lane & 31, hi = lane >> 5;
    const float alpha = p->alpha;
    // ---- loads of the tile's constants, issued before anything else.  Column constants: lane l < 40 owns 4 of the wave's
    // 160 columns (plain: ncol0 + 4l; GEGLU: h columns gcol0 + 4l for l < 20, the matching g columns N + gcol0 + 4(l - 20))
    const half_t* bias = p->bias;
    const int Ni = (int)p->N;
    const int ccol = GEGLU ? (lane < 20 ? gcol0 + 4 * lane : Ni + gcol0 + 4 * (lane - 20)) : ncol0 + 4 * lane;
    h4 cb = {};
    f4v cc = {};
    if (lane < 40) {
        if (bias) cb = *reinterpret_cast<const h4*>(bias + ccol);
        if constexpr (LN) cc = *reinterpret_cast<const f4v*>(p->colvec + ccol);
    }
    // LayerNorm folded into the GEMM: the (rstd, -rstd * mean) pairs of this wave's rows, all blocks at once
    float st_rs[TM], st_rt[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { st_rs[i] = 1.f; st_rt[i] = 0.f; }
    if constexpr (LN) {
        const float* rsp = p->rowscale;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned mr = (unsigned)min(mrow0 + i * 32 + l31, (int)p->M - 1);
            const f2v v = *reinterpret_cast<const f2v*>(rsp + 2 * mr);
            st_rs[i] = v[0];
            st_rt[i] = v[1];
        }
    }
    PreSrc ps = {};
    constexpr int NF = 10 * TM;
    h8 pre[ADD ? PD : 1];
    if constexpr (ADD) {
        const half_t* resp = p->residual;
        ps.src = resp ? resp : p->rowvec;
        ps.is_rowvec = resp == nullptr;
        ps.pitch = ps.is_rowvec ? (unsigned)p->N : (unsigned)p->ldr;
        ps.mrow0 = mrow0;
        ps.ncol0 = ncol0;
        ps.Mlast = (int)p->M - 1;
        if (ps.is_rowvec) {
            const unsigned rpv = (unsigned)p->rows_per_vec;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ps.v0[i] = (unsigned)min(mrow0 + i * 32, ps.Mlast) / rpv;       // wave-uniform
                ps.bnd[i] = (ps.v0[i] + 1u) * rpv;
            }
        }
#pragma unroll
        for (int f = 0; f < PD; ++f) pre_fetch<PD>(ps, lane, f, pre);
    }
    float* cst = stg + EP_CONST;
    auto park_constants = [&]() {               // after the first chunk's ds_writes: the loads above have had that long
        if (lane < 40) {
            f4v bf;
#pragma unroll
            for (int e = 0; e < 4; ++e) bf[e] = (float)cb[e];
            *reinterpret_cast<f4v*>(cst + 4 * lane) = bf;
            if constexpr (LN) *reinterpret_cast<f4v*>(cst + 160 + 4 * lane) = cc;
        }
    };
    if constexpr (GEGLU) {
        // tile j: registers 0-7 are h of 16 output columns, registers 8-15 the matching g.  The activation (which
        // needs the bias first) is computed in the fragment layout; chunks: j = 0..3 (64 columns), j = 4 (16)
        park_constants();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mblk = mrow0 + i * 32;
            const float rs = st_rs[i], rt = st_rt[i];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j0 = c * 4, nj = c ? 1 : 4;
                const int stride = nj * 16 + 4;
#pragma unroll
                for (int jj = 0; jj < nj; ++jj) {
                    const int j = j0 + jj;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int co = j * 16 + 8 * q + 4 * hi;         // column offset inside the wave's 80 outputs
                        float hv[4], gv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {       // alpha = 1 (pp_supported: a GEGLU launch with another scale stays on the
                            hv[e] = acc[j][i][4 * q + e];    // tile kernels): x * 1.0f is the identity, 160 packed multiplications per
                            gv[e] = acc[j][i][8 + 4 * q + e];    // wave and tile fewer in an epilogue that is bound by VALU issue
                        }
                        if constexpr (LN) {
                            const f4v ch = *reinterpret_cast<const f4v*>(cst + 160 + co);
                            const f4v cg = *reinterpret_cast<const f4v*>(cst + 240 + co);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                hv[e] = __builtin_fmaf(rs, hv[e], rt * ch[e]);
                                gv[e] = __builtin_fmaf(rs, gv[e], rt * cg[e]);
                            }
                        }
                        if (bias) {
                            const f4v bh = *reinterpret_cast<const f4v*>(cst + co);
                            const f4v bg = *reinterpret_cast<const f4v*>(cst + 80 + co);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { hv[e] += bh[e]; gv[e] += bg[e]; }
                        }
                        const vsx_f2 g01 = (VSX_PP_ABLATE & 2) ? vsx_f2{gv[0], gv[1]} : gelu_erf_f2(vsx_f2{gv[0], gv[1]});
                        const vsx_f2 g23 = (VSX_PP_ABLATE & 2) ? vsx_f2{gv[2], gv[3]} : gelu_erf_f2(vsx_f2{gv[2], gv[3]});
                        const f4v o = {hv[0] * g01[0], hv[1] * g01[1], hv[2] * g23[0], hv[3] * g23[1]};
                        *reinterpret_cast<f4v*>(stg + l31 * stride + jj * 16 + 8 * q + 4 * hi) = o;
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one accumulator tile at a time (register pressure)
                }
                if (c == 0) epilogue_rows<64, 0, false>(p, stg, lane, mblk, gcol0, nullptr, ps, nullptr, 0, 0);
                else epilogue_rows<16, 0, false>(p, stg, lane, mblk, gcol0 + 64, nullptr, ps, nullptr, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        // plain: chunks j = {0, 1}, {2, 3} (64 columns: one full 128-byte line per row), {4} (32 columns)
        const bool use_cst = LN || bias != nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mblk = mrow0 + i * 32;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int j0 = c * 2, nj = c < 2 ? 2 : 1;
                const int stride = nj * 32 + 4;
#pragma unroll
                for (int jj = 0; jj < nj; ++jj) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f4v o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = acc[j0 + jj][i][4 * g + e] * alpha;
                        *reinterpret_cast<f4v*>(stg + l31 * stride + jj * 32 + 8 * g + 4 * hi) = o;
                    }
                }
                if (i == 0 && c == 0) park_constants();
                __builtin_amdgcn_sched_barrier(0);
                const float* cc_ = use_cst ? cst + c * 64 : nullptr;
                if (c < 2) epilogue_rows<64, ADD ? PD : 0, LN, STATS, SUBPIX>(p, stg, lane, mblk, ncol0 + j0 * 32, cc_, ps, pre, i * 10 + c * 4, NF, st_rs[i], st_rt[i], part0 + c);
                else epilogue_rows<32, ADD ? PD : 0, LN, STATS, SUBPIX>(p, stg, lane, mblk, ncol0 + j0 * 32, cc_, ps, pre, i * 10 + 8, NF, st_rs[i], st_rt[i], part0 + c);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// TM: 32-row blocks per wave (tile = 128*TM x 320).  CONV: 0 = plain row-major A, 1 = implicit-GEMM A loader (a_mode 1),
// 2 = the same with one A slab per filter row ("SHARED A SLAB" below: stride-1 3x3, taps-inner order).
// EPI: epilogue kind (EPI_* bits).  This wave's 2*TM + 5 DMA pieces of a slab are issued in the load phases
// L0 [0, CUT1), L1 [CUT1, CUT2), L2 [CUT2, ..) (A pieces first: they come from HBM / Infinity Cache, the weights from
// L2); L3 issues nothing.  (Other cut points were measured in round 3 — profiles/r03_gemm_sched_ab_b2.txt — and dropped.)
template <int TM, int CONV, int EPI>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmParams p_unused, const int tiles_total) {
    constexpr int CUT1 = 3, CUT2 = TM == 2 ? 6 : 5;
    constexpr int TN = 5;
    constexpr int BM = 128 * TM, BN = 320;
    constexpr int WM = 32 * TM, WN = 160;
    constexpr int GA = 2 * TM;                  // 8-row (1 KiB) A pieces per wave and slab
    constexpr int GB = 5;                       // B pieces per wave and slab
    constexpr int NPIECE = GA + GB;
    constexpr int STAGE = (BM + BN) * 128;      // bytes per ring slot
    constexpr int NFIT = STAGE / EP_BYTES;      // waves whose epilogue staging area fits a ring slot
    constexpr int OOB_OFF = (int)0x80000000;
    constexpr int ZERO_OFF = 2 * STAGE + (8 - NFIT) * EP_BYTES;     // 128 zero bytes behind the staging areas (shared A slab)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    kparams_t p = kernarg_params();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int G = wave >> 2;                    // ping-pong group: waves w and w + 4 share a SIMD
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- this workgroup's tiles: XCD x (workgroups b with b % 8 == x) owns a contiguous range of the
    // (tile_m, tile_n) space — its CUs share one L2 — and the XCD's workgroups stride through it together ----
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int q8 = tiles_total >> 3, r8 = tiles_total & 7;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int wgx = (nwg - xcd + 7) >> 3;       // workgroups placed on this XCD
    const int n_my = local < cnt ? (cnt - local + wgx - 1) / wgx : 0;
    if (n_my == 0) return;
    const int tile0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;   // + k * wgx

    // ---- LDS-DMA coordinates (see gemm.hip): lane -> (row lrow of an 8-row piece, physical 16-byte slot pslot);
    // the XOR swizzle slot ^= (tile_row >> 1) & 7 is applied to the SOURCE column ----
    const int lrow = lane >> 3;
    const int pslot = lane & 7;
    const int kofs_e = (pslot ^ (lrow >> 1)) * 8;           // even pieces
    const int kofs_o = (pslot ^ (4 | (lrow >> 1))) * 8;     // odd pieces

    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p->A), 0, (int)p->a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p->B), 0, (int)p->b_bytes, 0x00020000);
    // second conv source (skip concat): built here, not where it is used — an s_load in a load phase would put an
    // lgkmcnt(0) (and with it the fragment ds_reads) in front of the DMA issue
    // (the sub-pixel form, CONV = 3, is single-source by contract — gemm.hip checks C2 == 0 — and carries no third descriptor:
    // its four extra SGPRs were spilled as a 16-byte stack object, the only private segment of that kernel)
    __amdgpu_buffer_rsrc_t rsrcA2 = rsrcA;
    if constexpr (CONV == 1 || CONV == 2)
        rsrcA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p->A2 ? p->A2 : p->A), 0, (int)p->a2_bytes, 0x00020000);

    const int Mi = (int)p->M;
    const int tiles_n = p->tiles_n;
    const int flags = p->pp_flags;
    const bool blocked = (flags & PP_TILES_LINEAR) == 0;
    const int tiles_m = (Mi + BM - 1) / BM;
    constexpr bool geglu = (EPI & EPI_GEGLU) != 0;
    const int K = (int)p->K;
    const int nk = (K + BK - 1) / BK;
    const int ktail_from = K - (nk - 1) * BK;            // last slab: k offsets >= this are beyond K
    const unsigned lda2 = (unsigned)p->lda * 2u, ldb2 = (unsigned)p->ldb * 2u;   // row pitches in bytes
    // PIECE ROTATION.  The CUs of an XCD run in step and share panels: the column tiles of one row tile read the same A
    // rows, the row tiles of one column tile the same weights.  Every CU therefore starts its walk over the BM / 8 A pieces
    // and (plain GEMMs) the 40 B pieces of a slab somewhere else — an even rotation, so that the swizzle parity of a piece
    // stays the wave's own — and the CUs do not ask their L2 for the same lines at the same moment.  Position-balanced A/B,
    // B = 2 (profiles/r04_gemm_rotation_ab_b2.txt): A rotation - 4.5 % on qkv 320->960, - 5.5 % on geglu 320->1280 (A panel
    // shared by 3 / 8 column tiles), neutral where A is private; B rotation - 2 ... 3.6 % on ff2 1280->320, proj 1280->1280,
    // ff2 5120->1280, + 1 ... 2 % on the convolutions (which keep the common B order); GEMM time of a forward - 1.15 %.
    // PP_COMMON_ORDER switches both off.  (The same rotation in the tile kernels changed nothing: call r04l.)
    const bool rot_on = (p->pp_flags & PP_COMMON_ORDER) == 0;
    const int w5 = wave * GB + (rot_on && !CONV ? 2 * (((int)(blockIdx.x >> 3) * 7) % 20) : 0);       // this wave's first B piece
    const int wa = wave * GA + (rot_on ? 2 * (((int)(blockIdx.x >> 3) * 5) % (BM / 16)) : 0);         // ... and A piece
    auto a_piece = [&](const int q) { return wa + q < BM / 8 ? wa + q : wa + q - BM / 8; };
    const int Ngeglu = geglu ? (int)p->N : 0;    // (only the GEGLU row interleave needs it: M and N are neighbours in the kernarg
                                                 // segment, and a merged 16-byte load keeps four SGPRs alive for two values)

    // per-lane parts of the operand offsets (bytes): row-in-piece * pitch + swizzled k slot
    const int vb_e = (int)((unsigned)lrow * ldb2) + kofs_e * 2;
    const int vb_o = (int)((unsigned)lrow * ldb2) + kofs_o * 2;
    int va_e = 0, va_o = 0;                      // plain A
    // conv A, per piece of this lane: a_px = source pixel of the window's first tap (may lie outside the image), a_mk = bit
    // 3 kh + kw: that tap is inside the image; bits 9 / 10: h0 / w0 odd (nearest-2x source: which taps share a source pixel),
    // a_vb = byte offset of pixel a_px in the CURRENT source (channel count C1 or C2) + the lane's swizzled k slot.  A slab's
    // offset is a_vb + a tap term that is uniform (or one of two uniform values by parity): three VALU operations per piece
    // and slab, whatever the slab order
    int a_px[CONV ? GA : 1], a_mk[CONV ? GA : 1], a_vb[CONV ? GA : 1];
    if constexpr (!CONV) {
        va_e = (int)((unsigned)lrow * lda2) + kofs_e * 2;
        va_o = (int)((unsigned)lrow * lda2) + kofs_o * 2;
    }

    // ---- issue side: the operand stream (tile after tile, slab after slab), one slab ahead of the MFMAs ----
    int i_t = 0, i_kt = 0, i_g = 0;              // tile (index into my sequence), slab in tile, slabs issued so far
    bool need_setup = true;
    int t_kh = 0, t_kw = 0, t_c = 0;             // conv: filter tap / channel base of the next slab
    bool t_second = false, t_dirty = true, t_src_dirty = true;
    const bool tap_inner = CONV >= 2 || (CONV && (flags & PP_CONV_TAP_MAJOR) == 0);
    // SHARED A SLAB (stride-1 3x3, W a power of two in [8, BM]; launch_pp decides).  The windows of the taps (kh, 0..2) are
    // one-pixel shifts of each other, and a tile starts at the first pixel of an image row: the A slab of tap (kh, 1) is
    // streamed ONCE per (channel slab, kh) and the MFMAs of kw = 0 / 2 read their fragments one LDS row lower / higher; the
    // lanes whose pixel sits at the left / right image edge read a zero row instead.  A slab (32 KiB of 72) then arrives with
    // every third slab only: 152 instead of 216 KiB into LDS per three slabs of a kernel that is bound by operand delivery.
    // The A ring runs on its own parity: A slab k of a tile lives in slot (first slab of the tile + k) & 1 — the slot of the
    // tile's first B slab for k = 0, so that the epilogue's staging area (the slot of the LAST slab) never holds the next
    // tile's operands; k + 1 never shares a slot with k, nor the next tile's k = 0 with this tile's last (nk / 3 - 1 and nk
    // differ in parity for every nk = 3q).
    constexpr bool ashift = CONV == 2;
    // SUB-PIXEL FORM of the nearest-2x convolution (CONV = 3; Upsample3D, resnet.py:54,66).  A 3x3 window on the upsampled image
    // meets only 2 x 2 source pixels: output pixel (2 i + ph, 2 j + pw) reads source rows i - 1 + ph, i + ph with the filter rows
    // {0} | {1, 2} (ph = 0) or {0, 1} | {2} (ph = 1) added up, and the same along the columns.  The host adds the taps up once
    // (B = four [N, 9 C] matrices, one per (ph, pw) class, whose taps (ph.., pw..) hold the sums), and the launch runs the four
    // classes as one GEMM over M = 4 Mc rows: a tile belongs to one class (Mc a multiple of the tile height), walks the class's
    // four taps of a PLAIN 3x3 window on the source — 4 / 9 of the slabs — and its epilogue scatters the rows to their pixels.
    constexpr bool subpix = CONV == 3;
    int sp_kh0 = 0, sp_kw0 = 0;
    const int wmask = ashift ? p->W - 1 : 0;
    int i_aslot = 0, i_sbA = 0;
    bool i_hasA = true;
    int t_bit = 0, t_re = 0, t_ro = 0, t_ce = 0, t_co = 0;     // tap bit; byte delta of the tap's row / column for an even / odd window origin
    int i_m0 = 0;                                // first row of the tile being staged
    unsigned i_rowB = 0;                         // (first B row of the tile) * pitch

    auto tile_setup = [&](const int tile) {
        int tile_n, tile_m;
        tile_coords(tile, tiles_n, tiles_m, blocked, tile_m, tile_n);
        i_m0 = tile_m * BM;
        i_rowB = (unsigned)(geglu ? tile_n * (BN / 2) : tile_n * BN) * ldb2;
        int sp_m0 = 0;                          // first GEMM row of the tile's class
        if constexpr (subpix) {
            const int cls = i_m0 / p->sp_Mc;
            sp_m0 = cls * p->sp_Mc;
            sp_kh0 = cls >> 1;
            sp_kw0 = cls & 1;
            i_rowB += (unsigned)(cls * (int)p->N) * ldb2;
        }
        if constexpr (CONV) {
            // CONV = 2 (shared A slab) and 3 (sub-pixel form) are stride-1 3x3 windows with one zero row / column around a
            // source that is not upsampled, output size = input size (launch_pp / gemm.hip route nothing else here): constants,
            // not launch parameters — fewer scalars alive across the K loop (the kernels spill 40 - 70 SGPRs into VGPR lanes)
            constexpr bool fixed = CONV >= 2;
            const int H = p->H, W = p->W;
            const unsigned Wo = fixed ? (unsigned)W : (unsigned)p->Wo, hw = (fixed ? (unsigned)H : (unsigned)p->Ho) * Wo;
            const int stride = fixed ? 1 : p->stride, pad = fixed ? 1 : p->pad, ks = fixed ? 3 : p->ks, ups = fixed ? 0 : p->ups;
            const int Hs = ups ? (H >> 1) : H, Ws = ups ? (W >> 1) : W;
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                const int m = i_m0 + a_piece(i) * 8 + lrow - (subpix ? sp_m0 : 0);      // (sub-pixel form: the row inside its class)
                const unsigned mm = m < Mi ? (unsigned)m : 0u;
                const unsigned img = mm / hw;
                const unsigned rem = mm - img * hw;
                const unsigned ho = rem / Wo;
                const int h0 = (int)ho * stride - pad;
                const int w0 = (int)(rem - ho * Wo) * stride - pad;
                int mk = 0;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        if (kh < ks && kw < ks && m < Mi && (unsigned)(h0 + kh) < (unsigned)H && (unsigned)(w0 + kw) < (unsigned)W)
                            mk |= 1 << (3 * kh + kw);
                a_mk[i] = mk | ((h0 & 1) << 9) | ((w0 & 1) << 10);
                a_px[i] = ((int)img * Hs + (ups ? (h0 >> 1) : h0)) * Ws + (ups ? (w0 >> 1) : w0);
            }
            t_kh = subpix ? sp_kh0 : 0; t_kw = subpix ? sp_kw0 : 0; t_c = 0; t_second = false; t_dirty = true;
            t_src_dirty = true;
            i_aslot = (i_g & 1) ^ 1;            // the tile's first slab toggles it to the slot of its B slab
        }
    };

    int i_sb = 0;                                // LDS byte offset of the slot being filled
    bool i_tail = false, i_second = false, i_on = true;
    unsigned i_soffA = 0, i_soffB = 0;
    auto slab_prep = [&]() {                    // fixes the offsets of stream slab i_g and advances the stream
        if (need_setup) {
            tile_setup(tile0 + i_t * wgx);
            need_setup = false;
        }
        const int kt = i_kt;
        if constexpr (CONV) {
            const int C1 = p->C1, C2 = CONV == 3 ? 0 : p->C2, ks = CONV >= 2 ? 3 : p->ks;
            const int csz = t_second ? C2 : C1;
            if (t_src_dirty) {                  // first slab of a source: the pixel offsets in that source's channel count
#pragma unroll
                for (int i = 0; i < GA; ++i) a_vb[i] = (a_px[i] * csz + ((i & 1) ? kofs_o : kofs_e)) * 2;
                t_src_dirty = false;
            }
            i_hasA = !ashift || t_kw == 0;
            if (i_hasA) i_aslot ^= 1;
            if (t_dirty) {                      // first slab of a tap (a slab never straddles a tap or the two sources)
                const int ups = CONV >= 2 ? 0 : p->ups;
                const int Ws = ups ? (p->W >> 1) : p->W;
                const int kw_e = ashift ? 1 : t_kw;         // shared A slab: the window of the centre column
                t_bit = 3 * t_kh + kw_e;
                // nearest-2x source: tap k of a window whose origin has parity b reads source row / column (b + k) >> 1
                t_re = ((ups ? t_kh >> 1 : t_kh) * Ws * csz) * 2;
                t_ro = ((ups ? (t_kh + 1) >> 1 : t_kh) * Ws * csz) * 2;
                t_ce = ((ups ? kw_e >> 1 : kw_e) * csz) * 2;
                t_co = ((ups ? (kw_e + 1) >> 1 : kw_e) * csz) * 2;
                t_dirty = false;
            }
            i_second = t_second;
            i_soffA = (unsigned)t_c * 2u;
            // K index of this slab in the weight rows (OHWI: tap-major, then source 1 | source 2 channels)
            const unsigned k0 = (unsigned)((t_kh * ks + t_kw) * (C1 + C2) + (t_second ? C1 : 0) + t_c);
            i_soffB = i_rowB + k0 * 2u;
            if (tap_inner) {
                // the ks * ks taps of one 64-channel slab back to back: their windows are shifts of the same input rows, so the
                // taps after the first hit L2 (tap-major order re-reads the input from the fabric once per tap: 3.8 - 7.7 x the
                // algorithmic bytes on the convolutions, profiles/r03_gemm_traffic_by_shape.txt)
                t_dirty = true;
                if (++t_kw == (subpix ? sp_kw0 + 2 : ks)) {
                    t_kw = subpix ? sp_kw0 : 0;
                    if (++t_kh == (subpix ? sp_kh0 + 2 : ks)) {
                        t_kh = subpix ? sp_kh0 : 0;
                        t_c += BK;
                        if (t_c >= csz) { t_c = 0; t_second = true; t_src_dirty = true; }      // (past source 2: the tile is done)
                    }
                }
            } else {
                t_c += BK;
                if (t_c >= csz) {                           // next source or next tap
                    t_c = 0;
                    t_dirty = true;
                    if (!t_second && C2 > 0) {
                        t_second = true;
                        t_src_dirty = true;
                    } else {
                        t_src_dirty = t_second;             // back to source 1 (two-source convolutions only)
                        t_second = false;
                        if (++t_kw == ks) { t_kw = 0; ++t_kh; }
                    }
                }
            }
        } else {
            i_soffA = (unsigned)i_m0 * lda2 + (unsigned)kt * (BK * 2);
            i_soffB = i_rowB + (unsigned)kt * (BK * 2);
        }
        i_sb = (i_g & 1) * STAGE;
        i_sbA = ashift ? i_aslot * STAGE : i_sb;
        i_tail = (kt == nk - 1) && ktail_from < BK;
        ++i_g;
        if (++i_kt == nk) { i_kt = 0; ++i_t; need_setup = true; }
    };
    auto issue_piece = [&](const int q) {           // q is a compile-time constant at every call site
        if (q < GA) {
            const int kofs = (q & 1) ? kofs_o : kofs_e;             // wave * GA and the rotation are even
            const int pa = a_piece(q);
            lptr_t dst = (lptr_t)(smem + (CONV ? i_sbA : i_sb) + pa * 1024);
            if constexpr (CONV) {
                if (!i_hasA) return;            // shared A slab: it came with the slab of kw = 0
                const int mk = a_mk[q < GA ? q : 0];
                const int dlt = ashift ? t_re + t_ce          // no nearest-2x source: one row / column term
                                       : ((mk & 512) ? t_ro : t_re) + ((mk & 1024) ? t_co : t_ce);
                const bool in = ((mk >> t_bit) & 1) != 0 && !(i_tail && kofs >= ktail_from);
                const int v = in ? a_vb[q < GA ? q : 0] + dlt : OOB_OFF;
                if (CONV != 3 && i_second) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA2, dst, 16, v, (int)i_soffA, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, dst, 16, v, (int)i_soffA, 0, 0);
                }
            } else {
                const bool ok = (i_m0 + pa * 8 + lrow < Mi) && !(i_tail && kofs >= ktail_from);
                const int v = ok ? ((q & 1) ? va_o : va_e) : OOB_OFF;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, dst, 16, v, (int)(i_soffA + (unsigned)(pa * 8) * lda2),
                                                         0, 0);
            }
        } else {
            const int j = q - GA;
            const int pj = w5 + j < BN / 8 ? w5 + j : w5 + j - BN / 8;      // piece index inside the 320-row B tile
            const bool odd = (pj & 1) != 0;
            const int kofs = odd ? kofs_o : kofs_e;
            // row of the weight matrix behind tile row pj*8 + lrow: plain n0 + pj*8 + lrow; GEGLU interleaves h and g
            // rows 16 + 16 inside every 32-row MFMA tile (rows 0-15 = h columns, 16-31 = the matching g columns)
            const unsigned srow = geglu ? (unsigned)((pj >> 2) * 16 + (pj & 1) * 8 + ((pj >> 1) & 1) * Ngeglu)
                                        : (unsigned)(pj * 8);
            const int v = (i_tail && kofs >= ktail_from) ? OOB_OFF : (odd ? vb_o : vb_e);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lptr_t)(smem + i_sb + BM * 128 + pj * 1024), 16, v,
                                                     (int)(i_soffB + srow * ldb2), 0, 0);
        }
    };
    auto issue_range = [&](const int lo, const int hi_) {   // compile-time bounds at every call site
        if (!i_on) return;
#pragma unroll
        for (int q = 0; q < NPIECE; ++q)
            if (q >= lo && q < hi_) issue_piece(q);
    };

    f16v acc[TN][TM];
    // fragment addresses: row (.. + l31) * 128 B, logical 16-byte slot ks*2 + hi, swizzled by (row >> 1) & 7:
    // ((ks*2 + hi) ^ swz) * 16 = ((hi ^ swz) * 16) ^ (ks * 32), and the row bases are multiples of 128
    const int fr = ((hi ^ ((l31 >> 1) & 7)) * 16);
    const int a_addr = (wr * WM + l31) * 128 + fr;
    const int b_addr = BM * 128 + (wc * WN + l31) * 128 + fr;

    h8 af[TM], bf[TN];
    // shared A slab: absolute A fragment offsets of the slab being multiplied (A slot, shifted row or the zero row), set per slab
    int aoff[ashift ? TM : 1];
    int c_kw = 0, c_aslot = 0;
    auto shared_a_setup = [&]() {
        const int ln = fresh_lane(lane);                         // (per slab: the pieces of `base` must not be hoisted and spilled)
        const int l31 = ln & 31, hi = ln >> 5;
        const int r = l31 + c_kw - 1;                            // -1 .. 32: the rows of an edge lane are never read
        const int base = c_aslot * STAGE + (wr * WM + r) * 128 + ((hi ^ ((r >> 1) & 7)) * 16);
#pragma unroll
        for (int i = 0; i < (ashift ? TM : 1); ++i) {
            const int row0 = wr * WM + i * 32;                   // tile row of the block's first row (a tile starts at w = 0)
            // image column of this lane's pixel (a tile starts at w = 0 and W divides the tile height): the left neighbour of
            // column 0 and the right neighbour of column W - 1 are padding.  (Image rows of 8 / 16 pixels — the 8 x 8 / 16 x 16
            // levels, on this kernel since several clips are denoised together — have such lanes inside a 32-row block, not
            // only at its ends.)
            const int col = (row0 + l31) & wmask;
            const bool edge = (c_kw == 0 && col == 0) || (c_kw == 2 && col == wmask);
            aoff[i] = edge ? ZERO_OFF : base + i * 4096;
        }
    };
    auto ldfrag = [&](const int slot_off, const int ks_) {
        // ks * 32 through an opaque scalar: `a_addr ^ (ks * 32)` is loop-invariant for ks = 1, 2, 3, and the compiler keeps all six
        // pre-XORed fragment addresses in registers across the whole kernel otherwise (six VGPRs the epilogues with an addend ring
        // do not have: they spilled).  One v_xor per fragment address and k-step instead.
        int ks = ks_;
        asm volatile("" : "+s"(ks));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (ashift) af[i] = *reinterpret_cast<const h8*>(smem + (aoff[i] ^ (ks * 32)));
            else af[i] = *reinterpret_cast<const h8*>(smem + slot_off + ((a_addr ^ (ks * 32)) + i * 4096));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bf[j] = *reinterpret_cast<const h8*>(smem + slot_off + ((b_addr ^ (ks * 32)) + j * 4096));
    };
    auto mma = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[j][i], 0, 0, 0);
    };
    auto mma_first = [&]() {                    // first k-step of a tile: C = 0 (inline constant, no zero-fill pass)
        const f16v zero = {};                   // all-zero C operand (an inline constant on the device)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], zero, 0, 0, 0);
    };
    auto lgkm0 = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); };     // lgkmcnt(0)
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (ashift) {
        if (tid < 32) reinterpret_cast<float*>(smem + ZERO_OFF)[tid] = 0.f;
        lgkm0();
    }
    // ---- prologue: slab 0 of the first tile, all pieces at once ----
    slab_prep();
    issue_range(0, NPIECE);
    wait_vmcnt<0>();
    bar();
    if (G == 1) bar();                          // group 1 runs one phase behind

    int c_g = 0;
#ifdef VSX_GEMM_TIMING
    long t_seg[3] = {0, 0, 0}, t_last = (long)__builtin_amdgcn_s_memtime();
#define PPSTAMP(i) do { const long t_now = (long)__builtin_amdgcn_s_memtime(); t_seg[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define PPSTAMP(i) do { } while (0)
#endif
    for (int c_t = 0; c_t < n_my; ++c_t) {
        for (int kt = 0; kt < nk; ++kt, ++c_g) {
            const int so = (c_g & 1) * STAGE;
            i_on = i_t < n_my;
            if (i_on) slab_prep();              // the slab that streams in while this one is multiplied
            if constexpr (ashift) {
                if (kt == 0) { c_kw = 0; c_aslot = c_g & 1; }
                else if (++c_kw == 3) { c_kw = 0; c_aslot ^= 1; }
                shared_a_setup();
            }
            // ---- k-step 0 ----
            ldfrag(so, 0);
            __builtin_amdgcn_sched_barrier(0);
            issue_range(0, CUT1);
            lgkm0();
            bar();
            __builtin_amdgcn_s_setprio(1);
            if (kt == 0) mma_first(); else mma();
            __builtin_amdgcn_s_setprio(0);
            bar();
            // ---- k-step 1 ----
            ldfrag(so, 1);
            __builtin_amdgcn_sched_barrier(0);
            issue_range(CUT1, CUT2);
            lgkm0();
            bar();
            __builtin_amdgcn_s_setprio(1);
            mma();
            __builtin_amdgcn_s_setprio(0);
            bar();
            // ---- k-step 2 ----
            ldfrag(so, 2);
            __builtin_amdgcn_sched_barrier(0);
            issue_range(CUT2, NPIECE);
            lgkm0();
            bar();
            __builtin_amdgcn_s_setprio(1);
            mma();
            __builtin_amdgcn_s_setprio(0);
            bar();
            // ---- k-step 3 ----
            ldfrag(so, 3);
            __builtin_amdgcn_sched_barrier(0);
            if (G == 1) wait_vmcnt<0>();        // G1's pieces of the next slab (issued in its L0..L2) have landed
            lgkm0();
            bar();
            __builtin_amdgcn_s_setprio(1);
            mma();
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (G == 0) wait_vmcnt<0>();        // G0's pieces: waited behind its last MFMAs
            if (kt + 1 < nk) bar();
        }
        // ---- tile boundary: both groups run the epilogue in the same barrier interval (G0: |B| E, G1: M3 E), then
        // meet at a barrier — the epilogue stages C through the ring slot the tile's last slab occupied (waves
        // 0..NFIT-1) and the LDS behind the ring (the others), and G0's next load phase issues DMA into that slot —
        // and G1 drops one phase behind again.  (The other slot holds slab 0 of the next tile, already landed.) ----
        if (G == 0) bar();
        PPSTAMP(0);
        {
            const int tile = tile0 + c_t * wgx;
            int tile_n, tile_m;
            tile_coords(tile, tiles_n, tiles_m, blocked, tile_m, tile_n);
            const int n0 = geglu ? tile_n * (BN / 2) : tile_n * BN;
            const int stg_off = wave < NFIT ? ((c_g - 1) & 1) * STAGE + wave * EP_BYTES
                                            : 2 * STAGE + (wave - NFIT) * EP_BYTES;
            constexpr int PD = (EPI & EPI_LN) ? 2 : ((CONV || (EPI & EPI_STATS)) ? 3 : 4);      // addend ring depth (what fits without scratch)
            epilogue_pp<TM, EPI, PD, CONV == 3>(acc, reinterpret_cast<float*>(smem + stg_off), tile_m * BM + wr * WM, n0 + wc * WN,
                                 n0 + wc * TN * 16, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        lgkm0();
        PPSTAMP(1);
        bar();
        if (G == 1) bar();
        PPSTAMP(2);
    }
#ifdef VSX_GEMM_TIMING
    if (lane == 0 && g_pp_dbg) {
        long* o_dbg = g_pp_dbg + ((long)blockIdx.x * 8 + wave) * 4;
        o_dbg[0] = t_seg[0]; o_dbg[1] = t_seg[1]; o_dbg[2] = t_seg[2]; o_dbg[3] = n_my;
    }
#endif
    if (G == 0) bar();                          // matches group 1's extra barrier at the start
}

template <int TM, int CONV, int EPI>
int launch_one(const GemmParams& p, hipStream_t stream) {
    constexpr size_t stage = (size_t)(128 * TM + 320) * 128;
    constexpr size_t smem = 2 * stage + (8 - stage / EP_BYTES) * EP_BYTES + (CONV == 2 ? 128 : 0);     // ring + the staging areas behind it (+ the zero row)
    static_assert(smem <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<TM, CONV, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "gemm_pp: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return vsx_fail(VSX_E_LAUNCH, "gemm_pp: cannot query the device");
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = p.tiles_total < n_cu ? p.tiles_total : n_cu;      // one persistent workgroup per CU
    hipLaunchKernelGGL((gemm_pp_kernel<TM, CONV, EPI>), dim3((unsigned)grid), dim3(512), smem, stream, p, p.tiles_total);
    return vsx_check_launch("vsx_gemm_f16 (persistent)");
}


// ---------------------------------------------------------------------------------------------------------------------
// WEIGHT-STATIONARY kernel for K = N = 320 (round 6): the byte-bound projections of the 64 x 64 level (`proj 320->320 +res`,
// proj_in with row statistics: 3.8 ms of a 79-ms step pair at 0.55 of the copy rate).  In the persistent ping-pong kernel such a
// launch costs K loop + epilogue — the stores are 24 - 27 % of it (profiles/r06_gemm_epilogue_ablation.txt) — because the two waves
// of a SIMD own ONE tile whose 160 accumulators leave no room for a second one.  Here the roles are swapped: the WEIGHTS stay in
// registers (a wave owns 64 output columns: 2 x 20 fragments of v_mfma_f32_32x32x16_f16 = 160 VGPRs), the activation streams
// through LDS in blocks of 32 rows (20 KB, three-slot ring filled by LDS-DMA, every HBM line fetched once), and a wave's whole
// result for a block is 32 accumulator registers — so its epilogue (staging, residual, stores of full 128-byte lines) is over
// in a few hundred cycles and the next block's operands are already in LDS; nothing waits for the stores, whose count is part
// of the counted vmcnt waits.  The residual arrives by LDS-DMA as well (wave-private 32 x 64 tiles, two slots), so no register
// holds it across the K loop.  5 waves (320 threads), one workgroup per CU, 150 KB of LDS.
// Same arithmetic as the other back ends — same instruction and operand roles, k ascending in steps of 16, and the staged chunk
// goes through epilogue_rows itself — so the results (and the row statistics, per 64-column part) are bit-identical to the
// persistent kernel's except that a row has 5 statistic parts instead of 6.
// vmcnt bookkeeping (in-order retirement; M % 32 == 0, so every store instruction is issued): per block a wave issues
// LD = 4 (+ 4 with a residual) DMA loads at the top and ST = 4 (+ 4 with statistics) stores in its epilogue.  Top of block j:
// the X pieces of block j were issued at the top of block j - 2, ahead of [R(j - 1)], ST(j - 2), LD(j - 1), ST(j - 1):
// vmcnt(LD - 4 + LD + 2 ST).  Before the epilogue of block j: R(j) was the last load of the top of block j - 1, ahead of
// ST(j - 1) and LD(j): vmcnt(ST + LD).  Blocks past the end are issued as out-of-range loads, so the counts hold to the last block.
// The prologue (X(0), then the weights in two LDS-DMA rounds drained to vmcnt(0), then X(1) and R(0)) shortens two of the counts:
// block 1's X wait has one block of stores behind it, block 0's R wait none.
// NW waves per workgroup (round 6, second form): 5 waves of 64 columns, or 10 waves of 32 columns — half the registers per wave (80 of weight
// fragments), 2 - 3 waves per SIMD instead of 1 - 2, so that one wave's barrier, fragment reads and epilogue hide under another's MFMAs (with 5
// waves a block costs ~3 us whatever its bytes: only the residual's traffic hid that).
constexpr int WS_NST = 3;
constexpr int WS_XST = 32 * 640;                     // bytes per X slot: 5 K slabs x [32 rows][128 B], XOR-swizzled like every slab
constexpr int WS_R0 = WS_NST * WS_XST;               // addend slots: 2 x [the waves' 32-row tiles of their columns] = 2 x 20 KB
constexpr int WS_RST = 32 * 640;
constexpr int WS_S0 = WS_R0 + 2 * WS_RST;            // staging areas: NW x WsGeo<NW>::EPW, then the row pairs of a folded LayerNorm (2 slots x 1 KB, 256 B used)
template <int NW>
struct WsGeo {
    static constexpr int COLS = 320 / NW, NB = COLS / 32, LPR = COLS / 8, RPP = 64 / LPR, PASSES = 32 / RPP;
    static constexpr int PITCH = NB * 32 + 4;                          // floats per staged row
    static constexpr int SCST = NW == 5 ? EP_CONST : 32 * PITCH;       // float offset of the column constants behind the staged chunk
    static constexpr int EPW = NW == 5 ? EP_BYTES : (SCST + 160 + COLS) * 4;
    static constexpr int RS0 = WS_S0 + NW * EPW;
    static constexpr int LDS = RS0 + 2 * 1024;
    static constexpr int WREG = 3 * COLS * 128;                        // prologue: three K slabs of the wave's weight rows
    static_assert(LDS <= 160 * 1024 && WS_XST + NW * WREG <= LDS, "LDS budget of the weight-stationary kernel");
};

template <int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_ws320_kernel(const GemmParams p_unused, const int nblocks, const int nslices) {
    typedef WsGeo<NW> G_;
    constexpr int COLS = G_::COLS, NB = G_::NB, LPR = G_::LPR, RPP = G_::RPP, PASSES = G_::PASSES, XP = 20 / NW;
    constexpr bool ADD = (EPI & EPI_ADD) != 0, STATS = (EPI & EPI_STATS) != 0, LN = (EPI & EPI_LN) != 0;
    // VMEM loads issued at the top of a block (X pieces, addend pieces, the row pairs), stores of its epilogue
    constexpr int LDT = XP + (ADD ? PASSES : 0) + (LN ? 1 : 0), ST = STATS ? 2 * PASSES : PASSES;
    constexpr int OOB_OFF = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    kparams_t p = kernarg_params();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int lrow = lane >> 3, pslot = lane & 7;
    const unsigned lda2 = (unsigned)p->lda * 2u, ldb = (unsigned)p->ldb;
    // (slice, chain) of this workgroup: the workgroups of XCD x are b = x + 8 * local; `nslices` neighbours share a chain
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int slice = local % nslices, grp = local / nslices;
    const int ngroups = ((int)gridDim.x >> 3) / nslices * 8;      // the host launches a multiple of 8 * nslices workgroups
    const int chain = grp * 8 + xcd;
    const int ncol_s = 320 * slice;
    const bool is_rowvec = ADD && p->residual == nullptr;
    const unsigned ldr2 = is_rowvec ? (unsigned)p->N * 2u : (unsigned)p->ldr * 2u;
    const unsigned rpv = is_rowvec ? (unsigned)p->rows_per_vec : 1u;

    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p->A), 0, (int)p->a_bytes, 0x00020000);
    const unsigned r_rows = is_rowvec ? (unsigned)((p->M + (long)rpv - 1) / (long)rpv) : (unsigned)p->M;
    const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(ADD ? (is_rowvec ? p->rowvec : p->residual) : p->A), 0,
        ADD ? (int)((r_rows - 1u) * ldr2 + (unsigned)p->N * 2u) : 0, 0x00020000);
    // DMA lane offsets: a piece is 8 rows x 128 B, the lane's 16-byte slot XOR-swizzled on the SOURCE side by (row >> 1) & 7
    const int vx_e = (int)((unsigned)lrow * lda2) + (pslot ^ (lrow >> 1)) * 16;            // pieces 0, 2 (rows 0-7, 16-23)
    const int vx_o = (int)((unsigned)lrow * lda2) + (pslot ^ (4 | (lrow >> 1))) * 16;      // pieces 1, 3
    const int rrow = lane / LPR, rslot = lane % LPR;                      // addend piece: RPP rows x (COLS halfs = LPR 16-byte slots)
    const int vr = (int)((unsigned)rrow * ldr2) + rslot * 16;

    int nmine = 0;                                   // row blocks chain + i * ngroups
    if (grp < ngroups / 8 && chain < nblocks) nmine = (nblocks - 1 - chain) / ngroups + 1;
    if (nmine == 0) return;                          // (whole groups leave together: no barrier is left waiting)
    auto issue_x = [&](const int i) {                // this wave's share of X block i: XP of the 20 pieces (K slab q / 4, rows 8 (q % 4) ..)
        const bool on = i < nmine;
        const unsigned m0 = (unsigned)(chain + (on ? i : 0) * ngroups) * 32u;
        const int slot = (i % WS_NST) * WS_XST;
#pragma unroll
        for (int k = 0; k < XP; ++k) {
            const int q = wave * XP + k, sl = q >> 2, pc = q & 3;          // (XP is 4 or 2: pc & 1 == k & 1)
            const int v = on ? ((k & 1) ? vx_o : vx_e) : OOB_OFF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lptr_t)(smem + slot + sl * 4096 + pc * 1024), 16, v,
                                                     (int)((m0 + 8u * pc) * lda2 + (unsigned)sl * 128u), 0, 0);
        }
    };
    auto issue_r = [&](const int i) {                // the addend of this wave's 32 x COLS outputs of block i: residual rows, or row vectors
        if constexpr (ADD) {
            const bool on = i < nmine;
            const unsigned m0 = (unsigned)(chain + (on ? i : 0) * ngroups) * 32u;
            const int slot = WS_R0 + (i & 1) * WS_RST + wave * (PASSES * 1024);
            const unsigned col2 = (unsigned)(ncol_s + COLS * wave) * 2u;
            if (is_rowvec) {                         // rows_per_vec >= 32: the block meets the vector of its first row and at most the next one
                const unsigned v0 = m0 / rpv, bnd = (v0 + 1u) * rpv;
#pragma unroll
                for (int pc = 0; pc < PASSES; ++pc) {
                    const unsigned vec = v0 + ((m0 + (unsigned)(RPP * pc + rrow)) >= bnd ? 1u : 0u);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcR, (lptr_t)(smem + slot + pc * 1024), 16,
                                                             on ? (int)(vec * ldr2) + rslot * 16 : OOB_OFF, (int)col2, 0, 0);
                }
            } else {
#pragma unroll
                for (int pc = 0; pc < PASSES; ++pc)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcR, (lptr_t)(smem + slot + pc * 1024), 16, on ? vr : OOB_OFF,
                                                             (int)((m0 + (unsigned)(RPP * pc)) * ldr2 + col2), 0, 0);
            }
        }
    };
    const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(LN ? p->rowscale : reinterpret_cast<const float*>(p->A)), 0, LN ? (int)((unsigned)p->M * 8u) : 0, 0x00020000);
    auto issue_rs = [&](const int i) {               // EPI_LN: the (rstd, -rstd * mean) pairs of block i's 32 rows = 256 contiguous bytes, by LDS-DMA
        if constexpr (LN) {                          // too (lanes 0-15; the others fetch zeros): a register load would make the compiler wait for everything before it
            const bool on = i < nmine && lane < 16;
            const unsigned m0 = (unsigned)(chain + (i < nmine ? i : 0) * ngroups) * 32u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcS, (lptr_t)(smem + G_::RS0 + (i & 1) * 1024), 16,        // (every wave fetches the same 256 bytes into the same place: its own vmcnt covers its own copy)
                                                     on ? lane * 16 : OOB_OFF, (int)(m0 * 8u), 0, 0);
        }
    };
    issue_x(0);

    // The weights: rows n = 320 slice + 64 wave + 32 nb + l31 of B [N][ldb], 16 k per fragment (A operand: row l31, k = 8 hi ..).  Loaded
    // straight from memory a fragment is 32 rows x 32 bytes — a quarter of every line it touches, 40 such instructions per wave: 6 of the first
    // version's 16 us of fixed cost per launch.  So they come through LDS like every operand: the wave's 64 rows as swizzled 128-byte
    // K slabs by LDS-DMA (full lines) into a wave-private 26-KB region behind X slot 0 (nothing else lives there yet), three slabs,
    // then two, read back as fragments.
    h8 wf[NB][20];
    {
        const __amdgpu_buffer_rsrc_t rsrcB =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p->B), 0, (int)p->b_bytes, 0x00020000);
        const unsigned ldb2 = ldb * 2u;
        const int vw_e = (int)((unsigned)lrow * ldb2) + (pslot ^ (lrow >> 1)) * 16;
        const int vw_o = (int)((unsigned)lrow * ldb2) + (pslot ^ (4 | (lrow >> 1))) * 16;
        unsigned char* wreg = smem + WS_XST + wave * G_::WREG;                // NW regions of three slabs behind X slot 0
        const int fw = l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) * 16);
        auto w_round = [&](const int sl0, const int nsl) {
#pragma unroll
            for (int sl = 0; sl < 3; ++sl)
                if (sl < nsl)
#pragma unroll
                    for (int pc = 0; pc < COLS / 8; ++pc)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lptr_t)(wreg + sl * (COLS * 128) + pc * 1024), 16, (pc & 1) ? vw_o : vw_e,
                                                                 (int)((unsigned)(ncol_s + COLS * wave + 8 * pc) * ldb2 + (unsigned)(sl0 + sl) * 128u), 0, 0);
            wait_vmcnt<0>();                     // (also X(0): it was issued first)
            __builtin_amdgcn_sched_barrier(0);   // (tools/cpu_check: the lanes of a wave meet here — on the device they are in lockstep anyway)
        };
        w_round(0, 3);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int t = 0; t < 12; ++t)
                wf[nb][t] = *reinterpret_cast<const h8*>(wreg + (t >> 2) * (COLS * 128) + nb * 4096 + (fw ^ ((t & 3) * 32)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments are in registers before the region is overwritten
        __builtin_amdgcn_sched_barrier(0);
        w_round(3, 2);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int t = 12; t < 20; ++t)
                wf[nb][t] = *reinterpret_cast<const h8*>(wreg + ((t >> 2) - 3) * (COLS * 128) + nb * 4096 + (fw ^ ((t & 3) * 32)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // column constants of the wave's COLS columns: bias, and c1 = sum_k W'[n][k] of the folded LayerNorm
    const half_t* bias = p->bias;
    h4 cb = {};
    f4v cc = {};
    if (lane < COLS / 4) {
        if (bias != nullptr) cb = *reinterpret_cast<const h4*>(bias + ncol_s + COLS * wave + 4 * lane);
        if constexpr (LN) cc = *reinterpret_cast<const f4v*>(p->colvec + ncol_s + COLS * wave + 4 * lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                    // every wave has its weights: the ring, the addend slots and the staging areas are free
    __builtin_amdgcn_sched_barrier(0);
    issue_x(1);
    issue_r(0);
    issue_rs(0);
    float* stg = reinterpret_cast<float*>(smem + WS_S0 + wave * G_::EPW);
    float* cst = stg + G_::SCST;
    const bool use_cst = LN || bias != nullptr;
    if (use_cst && lane < COLS / 4) {                // as fp32, where epilogue_rows looks for them: bias at [0 ..), c1 at [160 ..)
        f4v bf;
#pragma unroll
        for (int e = 0; e < 4; ++e) bf[e] = (float)cb[e];
        *reinterpret_cast<f4v*>(cst + 4 * lane) = bf;
        if constexpr (LN) *reinterpret_cast<f4v*>(cst + 160 + 4 * lane) = cc;
    }
    const float alpha = p->alpha;
    const int fa = l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) * 16);
    PreSrc ps;
    ps.src = nullptr; ps.is_rowvec = false; ps.pitch = 0; ps.v0[0] = ps.v0[1] = 0; ps.bnd[0] = ps.bnd[1] = 0;
    ps.mrow0 = 0; ps.ncol0 = 0; ps.Mlast = 0;

    int xslot = 0;                                   // j % WS_NST
    for (int j = 0; j < nmine; ++j) {
        // this wave's pieces of X(j) have landed (X(0) did in the prologue; X(1) was issued behind the barrier of the prologue, with only
        // ONE block's stores after it)
        if (j == 1) wait_vmcnt<2 * LDT - 4 + ST>();
        else wait_vmcnt<2 * LDT - 4 + 2 * ST>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                // ... everybody's have, and nobody reads X(j - 1) any more
        __builtin_amdgcn_sched_barrier(0);
        issue_x(j + 2);
        issue_r(j + 1);
        issue_rs(j + 1);
        const unsigned char* xs = smem + xslot * WS_XST;
        f16v acc[NB];
        {
            const f16v zero = {};
            const h8 xf = *reinterpret_cast<const h8*>(xs + fa);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[nb][0], xf, zero, 0, 0, 0);
        }
#pragma unroll
        for (int t = 1; t < 20; ++t) {
            const h8 xf = *reinterpret_cast<const h8*>(xs + (t >> 2) * 4096 + (fa ^ ((t & 3) * 32)));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[nb][t], xf, acc[nb], 0, 0, 0);
        }
        // ---- epilogue of this wave's 32 x COLS block: the staged chunk of epilogue_pp (NB 32-column tiles, pitch 32 NB + 4) ----
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f4v o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[nb][4 * g + e] * alpha;
                *reinterpret_cast<f4v*>(stg + l31 * G_::PITCH + nb * 32 + 8 * g + 4 * hi) = o;
            }
        __builtin_amdgcn_sched_barrier(0);
        const int m0 = (chain + j * ngroups) * 32;
        h8 pre[ADD ? PASSES : 1];
        f2v rs_cur = {1.f, 0.f};
        if constexpr (ADD || LN) {
            if (j == 0) wait_vmcnt<LDT>();           // R(j) / the row pairs have landed (wave-private: no barrier); block 0's have no stores behind them
            else wait_vmcnt<ST + LDT>();
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (ADD) {
            const unsigned char* rs = smem + WS_R0 + (j & 1) * WS_RST + wave * (PASSES * 1024) + lane * 16;
#pragma unroll
            for (int ps_ = 0; ps_ < PASSES; ++ps_) pre[ps_] = *reinterpret_cast<const h8*>(rs + ps_ * 1024);
        }
        if constexpr (LN) rs_cur = *reinterpret_cast<const f2v*>(smem + G_::RS0 + (j & 1) * 1024 + l31 * 8);
        epilogue_rows<COLS, ADD ? PASSES : 0, LN, STATS>(p, stg, lane, m0, ncol_s + COLS * wave, use_cst ? cst : nullptr, ps, pre, 0, PASSES,
                                                        rs_cur[0], rs_cur[1], NW * slice + wave);
        VSX_VMEM_NOTE(ST);
        __builtin_amdgcn_sched_barrier(0);
        xslot = xslot == WS_NST - 1 ? 0 : xslot + 1;
    }
}

template <int EPI, int NW>
int launch_ws_one(const GemmParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws320_kernel<EPI, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, WsGeo<NW>::LDS);
        if (e != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "gemm_ws: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return vsx_fail(VSX_E_LAUNCH, "gemm_ws: cannot query the device");
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // one workgroup per CU; the S workgroups of a chain are neighbours on one XCD (b % 8): a multiple of 8 S workgroups, no more chains than row blocks
    const int nblocks = (int)(p.M / 32), S = (int)(p.N / 320);
    int per_xcd = (n_cu / 8) / S;                    // chains per XCD (32 / S)
    if (per_xcd < 1) per_xcd = 1;
    const int want = (nblocks + 7) / 8;              // chains per XCD that have a block at all
    if (per_xcd > want) per_xcd = want;
    const int grid = 8 * S * per_xcd;
    hipLaunchKernelGGL((gemm_ws320_kernel<EPI, NW>), dim3((unsigned)grid), dim3(NW * 64), WsGeo<NW>::LDS, stream, p, nblocks, S);
    return vsx_check_launch("vsx_gemm_f16 (weight-stationary)");
}

}  // namespace

#ifdef VSX_GEMM_TIMING
extern "C" int vsx_pp_debug_buffer(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pp_dbg), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

bool pp_rowstats_ok(const GemmParams& p) {
    return !p.geglu && p.rowscale == nullptr && p.a_mode == 0 && p.c_mode == 0 && pp_supported(p);
}

bool pp_supported(const GemmParams& p) {
    // no column edge (N a multiple of the tile), 16-byte epilogue accesses, fast-tap convolutions, 32-bit offsets
    const long cols = p.geglu ? 2 * p.N : p.N;
    if (cols % 320 != 0 || p.splitk > 1) return false;
    if (p.c_mode == 1) {
        // transposed store (epilogue_vt): a 256-row tile inside one image, one full 128-byte line of every V^T row per wave
        if (p.a_mode != 0 || p.geglu || p.residual || p.rowvec || p.rowstats || p.c_rows_per_img % 256 != 0 || p.M % 256 != 0 ||
            p.ldc % 8 != 0 || p.c_img_stride % 8 != 0 || !vsx_aligned16(p.C) || !vsx_aligned16(p.rowscale))
            return false;
    } else if (p.c_mode != 0) {
        return false;
    }
    if (!p.vec8 || (p.residual && !p.rvec8)) return false;
    if (!vsx_aligned16(p.bias) || !vsx_aligned16(p.rowvec) || !vsx_aligned16(p.colvec)) return false;      // 16-byte epilogue loads
    // one prefetched addend: residual or row vector (not both; not under GEGLU); a 32-row block meets <= 2 row vectors
    if (p.rowvec && (p.residual || p.geglu || p.rows_per_vec < 32)) return false;
    if (p.geglu && p.residual) return false;
    if (p.geglu && p.alpha != 1.f) return false;       // the GEGLU epilogue of the persistent kernel does not multiply by alpha
    if (p.rowscale && p.a_mode == 1) return false;
    if (p.a_mode == 1) {
        const int ctot = p.C1 + p.C2;
        if (ctot % BK != 0 || p.C1 % BK != 0) return false;      // a slab must not straddle a tap or a source
        if (p.H >= 32768 || p.W >= 32768) return false;          // packed (h0, w0)
    }
    return p.M < (1L << 30) && p.K < (1L << 30);
}

int launch_pp(GemmParams& p, int bm, hipStream_t stream) {
    const long cols = p.geglu ? 2 * p.N : p.N;
    p.tiles_n = (int)(cols / 320);
    p.tiles_total = (int)(((p.M + bm - 1) / bm) * p.tiles_n);
    // option "pp_sched" (env VSX_PP_SCHED): PP_* bits (tile walk)
    p.pp_flags = (int)(gemm_option("pp_sched") & (PP_TILES_LINEAR | PP_CONV_TAP_MAJOR | PP_CONV_PRIVATE_A | PP_COMMON_ORDER));
    int conv = p.a_mode == 1 ? 1 : 0;
    // shared A slab (gemm_pp_kernel: "SHARED A SLAB"): stride-1 3x3 convolutions in the taps-inner order whose image rows
    // are a power of two of at least 8 pixels and at most one tile, so that every tile starts at w = 0
    if (conv && (p.pp_flags & (PP_CONV_TAP_MAJOR | PP_CONV_PRIVATE_A)) == 0 && p.ks == 3 && p.stride == 1 && p.ups == 0 &&
        p.pad == 1 && p.Wo == p.W && p.Ho == p.H && p.W >= 8 && p.W <= bm && (p.W & (p.W - 1)) == 0)
        p.pp_flags |= PP_CONV_ASHIFT_ON, conv = 2;
    if (conv && p.sp_Mc > 0) conv = 3;
    const int epi = (p.geglu ? EPI_GEGLU : 0) | (p.rowscale ? EPI_LN : 0) | (p.residual || p.rowvec ? EPI_ADD : 0) |
                    (p.rowstats ? EPI_STATS : 0) | (p.c_mode == 1 ? EPI_VT : 0);
    if (p.c_mode == 1 && bm != 256) return vsx_fail(VSX_E_UNSUPPORTED, "gemm_pp: the transposed store runs on 256-row tiles");
#define VSX_PP_CASE(TM_, CONV_, EPI_) \
    if ((bm == 256) == (TM_ == 2) && conv == CONV_ && epi == (EPI_)) return launch_one<TM_, CONV_, (EPI_)>(p, stream);
#define VSX_PP_CASES(TM_)                             \
    VSX_PP_CASE(TM_, 3, 0)                        \
    VSX_PP_CASE(TM_, 2, 0)                        \
    VSX_PP_CASE(TM_, 2, EPI_ADD)                  \
    VSX_PP_CASE(TM_, 1, 0)                        \
    VSX_PP_CASE(TM_, 1, EPI_ADD)                  \
    VSX_PP_CASE(TM_, 0, 0)                        \
    VSX_PP_CASE(TM_, 0, EPI_ADD)                  \
    VSX_PP_CASE(TM_, 0, EPI_LN)                   \
    VSX_PP_CASE(TM_, 0, EPI_LN | EPI_ADD)         \
    VSX_PP_CASE(TM_, 0, EPI_GEGLU)                \
    VSX_PP_CASE(TM_, 0, EPI_GEGLU | EPI_LN)       \
    VSX_PP_CASE(TM_, 0, EPI_STATS)                \
    VSX_PP_CASE(TM_, 0, EPI_STATS | EPI_ADD)
    VSX_PP_CASES(2)
    VSX_PP_CASE(2, 0, EPI_VT)
    VSX_PP_CASE(2, 0, EPI_VT | EPI_LN)
    VSX_PP_CASES(1)
#undef VSX_PP_CASES
#undef VSX_PP_CASE
    return vsx_fail(VSX_E_UNSUPPORTED, "gemm_pp: no kernel for this epilogue (pp_supported must be asked first)");
}

// Weight-stationary kernel (gemm_ws320_kernel): K = 320, N = 320 / 640 / 960, plain row-major A, whole 32-row blocks; bias, residual OR row
// vector, folded LayerNorm, row statistics (not with the LayerNorm).
bool ws_supported(const GemmParams& p) {
    if (p.K != 320 || p.N % 320 != 0 || p.N > 960 || p.a_mode != 0 || p.geglu || p.c_mode != 0 || p.splitk > 1) return false;
    if (p.rowvec && (p.residual || p.rows_per_vec < 32)) return false;
    if (p.rowscale && (p.rowstats || p.colvec == nullptr || !vsx_aligned16(p.colvec))) return false;
    if (p.M % 32 != 0 || p.M < 32 || p.M >= (1L << 26) || p.batch1 != 1) return false;
    if (!p.vec8 || (p.residual && !p.rvec8) || !vsx_aligned16(p.bias) || !vsx_aligned16(p.rowvec) || !vsx_aligned16(p.A) || !vsx_aligned16(p.B)) return false;
    if (p.lda % 8 != 0 || p.ldb % 8 != 0 || p.lda < 320 || p.ldb < 320) return false;
    return (unsigned long)p.M * (unsigned long)p.lda * 2ul < (1ul << 31) && (!p.residual || (unsigned long)p.M * (unsigned long)p.ldr * 2ul < (1ul << 31));
}

// option "ws_waves" / VSX_WS_WAVES: 5 waves of 64 columns or 10 waves of 32 (the row statistics then come in 10 parts per 320 columns)
int ws_waves() { return gemm_option("ws_waves") == 5 ? 5 : 10; }

int launch_ws(GemmParams& p, hipStream_t stream) {
    static const bool trace = getenv("VSX_WS_TRACE") != nullptr;      // (tests: which launches took this kernel)
    if (trace) fprintf(stderr, "[vsx] weight-stationary: M=%ld N=%ld res=%d rowvec=%d ln=%d stats=%d\n", p.M, p.N, p.residual != nullptr, p.rowvec != nullptr, p.rowscale != nullptr, p.rowstats != nullptr);
    const int epi = (p.residual || p.rowvec ? EPI_ADD : 0) | (p.rowstats ? EPI_STATS : 0) | (p.rowscale ? EPI_LN : 0);
    const bool w10 = ws_waves() == 10;
#define VSX_WS_CASE(E) case (E): return w10 ? launch_ws_one<(E), 10>(p, stream) : launch_ws_one<(E), 5>(p, stream);
    switch (epi) {
        VSX_WS_CASE(0)
        VSX_WS_CASE(EPI_ADD)
        VSX_WS_CASE(EPI_STATS)
        VSX_WS_CASE(EPI_ADD | EPI_STATS)
        VSX_WS_CASE(EPI_LN)
        VSX_WS_CASE(EPI_LN | EPI_ADD)
        default: return vsx_fail(VSX_E_UNSUPPORTED, "gemm_ws: no kernel for this epilogue (ws_supported must be asked first)");
    }
#undef VSX_WS_CASE
}

}  // namespace vsxg
