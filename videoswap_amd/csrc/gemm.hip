// K1/K2 — fp16 MFMA GEMM and implicit-GEMM convolution for gfx950 (MI355X).
//
//   C[m, n] = epilogue(alpha * sum_k A[m, k] * B[n, k])
//
// Design (CDNA4):
//  * v_mfma_f32_32x32x16_f16, fp32 accumulators.  Workgroup = WAVES_M x WAVES_N wave64, every wave owns a 32x160 (or
//    smaller) sub-tile: 256x320 (16 waves), 128x320 (8), 128x160 (4, two workgroups per CU), 128x128 / 64x128 / 64x64.
//    N = 320 divides every channel count of the SD-1.5 UNet (320/640/1280 and the 2560/5120/10240 GEGLU rows), so no
//    MFMA work is wasted on column padding.
//  * Operand staging is LDS-DMA (`buffer_load_dwordx4 ... offen lds`): no VGPR round trip, no ds_write.  K advances in
//    slabs of BK = 64 halfs, so every row contributes one full 128-byte L2 line per slab (64-byte half lines capped the
//    L2 -> LDS stream at 11.5 TB/s chip-wide, full lines reach 17-24 TB/s: tools/ubench/stage2.hip).  Ring of 2 slabs
//    for the big tiles, 4 for the small ones, COUNTED `s_waitcnt vmcnt(N)` and one raw `s_barrier` per slab.
//  * Buffer descriptors do the edge handling: a row's per-lane byte offset is loop-invariant, K advances in the
//    scalar offset, and an out-of-range offset makes the hardware return zeros (M/N edges, conv zero padding, K tail).
//  * LDS rows are 128 B (8 x 16-B slots) XOR-swizzled with (row >> 1) & 7; the DMA destination must be lane-linear,
//    so the swizzle is applied to the SOURCE column of each lane and again when fragments are read: conflict-free
//    ds_read_b128 in the hardware's 16-lane groups.
//  * Main loop: fragments are register-double-buffered across the four k-steps of a slab, a slab's 1-KiB DMA pieces
//    are issued one at a time between MFMAs, and the two waves of a SIMD stage in different k-steps.
//  * Workgroup ids are remapped so that each XCD (own L2) walks a contiguous range of tiles.
//
// The A operand has two loaders:
//   a_mode 0: plain row-major [M, K];
//   a_mode 1: implicit im2col of a channels-last image [nimg, H, W, C1(+C2)] for a ks x ks conv (stride 1/2, zero
//             pad ks/2), optionally reading a half-resolution source as a nearest-2x upsample, optionally
//             concatenating two sources on the channel axis.  This removes the reference's F.interpolate tensor
//             (resnet.py:54), torch.cat (unet_blocks.py:618) and the two rearrange copies per conv (resnet.py:14-16).
// Epilogue: alpha, bias[n], per-image row vector (time embedding, resnet.py:172-176), residual, GEGLU (h/g weight
// rows interleaved 16+16 inside every 32-row MFMA tile so both halves of a column land in the same lane),
// transposed store (V^T for the attention kernel).  Accumulators hold C^T (SWAP) so a lane owns one output row and 4
// consecutive columns per register quad: 8-byte vector loads/stores in the epilogue.
#include "gemm_common.h"
#include <algorithm>
#include <cstdlib>
#include <vector>
#include <stdlib.h>
#include <string.h>

using namespace vsxg;

// Diagnostics only (tools/tile_probe.py, tools/gemm_timing.py, tools/small_m_sweep.py): option "tile_tune" (initial value
// from VSX_TUNE_TILE) = tile + 16 * deep + 256 * splits.  tile 1|2|3 forces the 128x320 / 128x160 / 256x320 tile for problems
// whose column count is a multiple of 320, 4|5|6 the 128x128 / 64x128 / 64x64 tile for any; deep = 1: four ring slots instead of
// two for the 128x160 tile; splits > 0: that many K slices (128x320 tiles + combine) where split-K is possible at all.
static long tile_tune() { return vsxg::gemm_option("tile_tune"); }
static long force_tile() { return tile_tune() & 15; }
static bool tune_deep() { return (tile_tune() & 16) != 0; }
static int tune_splits() { return (int)((tile_tune() >> 8) & 15); }

namespace {

// BK (K slab of 64 halfs: LDS rows are 128 B = one full L2 line per row), GemmParams, lptr_t, wait_vmcnt: gemm_common.h
// LDS ring depth NSTAGE is a template parameter: 2 slots for the big tiles (57-74 KiB per slab), 4 for the small ones

// -DVSX_GEMM_TIMING (tools/gemm_timing.py builds its own copy of the library): per-wave cycle totals of the main-loop
// segments, written to the workspace as long[block][wave][4].  Never defined in the product build.
#ifdef VSX_GEMM_TIMING
#define TSTAMP(i) do { const long t_now = (long)clock64(); t_seg[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

// SWAP = true : accumulators hold C^T tiles (MFMA A operand = weight rows, B operand = activation rows), so a lane owns
//               one output row m and 4 consecutive columns per register quad -> 8-byte epilogue loads/stores.
// SWAP = false: accumulators hold C tiles; a lane owns one column n and 4 consecutive rows -> used by the transposed
//               (V^T) store, where rows are the contiguous axis.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool SWAP, int NSTAGE>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_kernel(const GemmParams p) {
    constexpr int PREFETCH = NSTAGE - 1;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_BLKS = BM / 8, B_BLKS = BN / 8;             // 8-row (8 x 128 B = 1 KiB) LDS-DMA pieces per operand
    static_assert(A_BLKS % NW == 0, "A blocks must split evenly over the waves");
    constexpr int GA = A_BLKS / NW;
    constexpr int GB_LO = B_BLKS / NW, GB_HI = (B_BLKS + NW - 1) / NW;
    constexpr int N_HI = B_BLKS - GB_LO * NW;                   // waves [0, N_HI) issue GB_HI B blocks, the rest GB_LO
    constexpr int STAGE = (BM + BN) * 128;                      // bytes per ring slot

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int l31 = lane & 31, hi = lane >> 5;
#ifdef VSX_GEMM_TIMING
    const long t_begin = (long)clock64();      // absolute stamps (long[block][wave][8], entries 4-7): entry, first / last slab, exit
#endif

    // XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (each XCD has its own 4 MiB L2), so give
    // every XCD a CONTIGUOUS range of the (tile_m, tile_n) space.  Bijective for any grid size; placement only affects
    // speed, never results.
    int tile_n, tile_m, split;
    if (p.xcd_gm > 0) {
        // XCD BLOCK GRID (unbatched launches whose work items number a multiple of 8; the host picks gm, launch()).  The work
        // items are (tile_m, np) with np = K slice * tiles_n + tile_n, one operand panel each way: A rows tile_m (of the
        // slice), weight rows np.  The eight XCDs form a gm x (8 / gm) grid over that space and an XCD's workgroups walk its
        // block rows fastest, so that an XCD fetches tiles_m / gm activation panels and NP * gm / 8 weight panels through
        // its L2 instead of (linear walk) tiles_m / 8 and ALL of the weights — which is what the small-M, long-K launches of
        // the 16x16 / 8x8 levels move: every XCD streamed the whole weight matrix, 5 - 8 x the algorithmic bytes, and the
        // 8x8 convolutions sat on the fabric's read rate (profiles/r04_gemm_traffic_by_shape_tap_inner.txt).
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        const int gm = p.xcd_gm;
        const int RB = p.tiles_m / gm;
        const int CB = (p.tiles_n * (p.splitk > 1 ? p.splitk : 1)) / (8 / gm);
        const int lr = local % RB;
        const int np = (xcd / gm) * CB + local / RB;
        tile_m = (xcd % gm) * RB + lr;
        tile_n = np % p.tiles_n;
        split = np / p.tiles_n;
    } else {
        int wg = blockIdx.x;
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = wg & 7, local = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
        tile_n = wg % p.tiles_n;
        tile_m = wg / p.tiles_n;
        split = p.splitk > 1 ? (int)blockIdx.z : 0;
    }
    const long m0 = (long)tile_m * BM;
    // geglu: a BN-row tile of B (h and g rows interleaved) produces BN/2 output columns
    const long n0 = p.geglu ? (long)tile_n * (BN / 2) : (long)tile_n * BN;

    const int z = (p.splitk > 1 || p.xcd_gm > 0) ? 0 : (int)blockIdx.z;
    const int z0 = z / p.batch1, z1 = z - z0 * p.batch1;
    const half_t* Ab = p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const half_t* Bb = p.B + z0 * p.b_bs0 + z1 * p.b_bs1;

    // ---- LDS-DMA coordinates: lane -> (row lrow of an 8-row piece, physical 16-byte slot pslot).  The LDS image is
    // lane-linear, so the XOR swizzle (slot ^= (tile_row >> 1) & 7) is applied to the SOURCE column: a lane's k offset
    // inside a slab depends on its row and on the parity of the piece (tile_row = piece * 8 + lrow) ----
    constexpr int OOB_OFF = (int)0x80000000;
    const int lrow = lane >> 3;
    const int pslot = lane & 7;
    const int kofs_e = (pslot ^ (lrow >> 1)) * 8;           // even pieces
    const int kofs_o = (pslot ^ (4 | (lrow >> 1))) * 8;     // odd pieces

    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Ab), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcA2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.A2 ? p.A2 : p.A), 0, (int)p.a2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Bb), 0, (int)p.b_bytes, 0x00020000);

    bool a_ok[GA];
    int a_h0[GA], a_w0[GA], a_ibase[GA], a_kofs[GA];
    int va[GA];                                   // current voffset of each A row (plain: fixed; conv: per tap)
    const int Hs = p.ups ? (p.H >> 1) : p.H;
    const int Ws = p.ups ? (p.W >> 1) : p.W;
    const int pad = p.pad;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const long m = m0 + (wave * GA + i) * 8 + lrow;
        a_ok[i] = m < p.M;
        a_h0[i] = 0; a_w0[i] = 0; a_ibase[i] = 0;
        a_kofs[i] = ((wave * GA + i) & 1) ? kofs_o : kofs_e;
        va[i] = a_ok[i] ? (int)((m * p.lda + a_kofs[i]) * 2) : OOB_OFF;
        if (p.a_mode == 1) {
            const long hw = (long)p.Ho * p.Wo;
            const long mm = a_ok[i] ? m : 0;
            const int img = (int)(mm / hw);
            const int rem = (int)(mm - (long)img * hw);
            const int ho = rem / p.Wo;
            a_h0[i] = ho * p.stride - pad;
            a_w0[i] = (rem - ho * p.Wo) * p.stride - pad;
            a_ibase[i] = img * Hs;
        }
    }
    // this wave's B blocks: [b_blk0, b_blk0 + gbw)
    const int gbw = wave < N_HI ? GB_HI : GB_LO;
    const int b_blk0 = wave < N_HI ? wave * GB_HI : N_HI * GB_HI + (wave - N_HI) * GB_LO;
    int vb[GB_HI], b_kofs[GB_HI];
#pragma unroll
    for (int j = 0; j < GB_HI; ++j) {
        const int jr = (b_blk0 + j) * 8 + lrow;       // row inside the B tile
        b_kofs[j] = ((b_blk0 + j) & 1) ? kofs_o : kofs_e;
        long n;
        bool ok;
        if (p.geglu) {   // inside every 32-row MFMA tile: rows 0-15 = h columns, rows 16-31 = the matching g columns
            const long oc = n0 + (jr >> 5) * 16 + (jr & 15);
            ok = oc < p.N;
            n = oc + ((jr & 16) ? p.N : 0);
        } else {
            n = n0 + jr;
            ok = n < p.N;
        }
        vb[j] = (ok && j < gbw) ? (int)((n * p.ldb + b_kofs[j]) * 2) : OOB_OFF;
    }

    const int Ctot = p.C1 + p.C2;
    const int nk = (int)((p.K + BK - 1) / BK);
    const int ktail_from = (int)(p.K - (long)(nk - 1) * BK);         // last slab: k offsets >= this are beyond K
    // conv fast path: a K slab never straddles a filter tap or the two concatenated sources, so (kh, kw, source,
    // channel base) are wave-uniform running counters and the row offsets change only when the tap does
    const bool fast_tap = p.a_mode == 1 && (Ctot % BK == 0) && (p.C1 % BK == 0);
    const int kt_begin = split * p.nk_per;
    const int kt_end = p.splitk > 1 ? min(nk, kt_begin + p.nk_per) : nk;
    int t_kh = 0, t_kw = 0, t_c = 0;          // state of the NEXT slab to issue (slabs are issued in order)
    bool t_second = false, t_dirty = true;
    if (fast_tap && kt_begin > 0) {           // split-K slice: start the running tap counters at slab kt_begin
        const int k0 = kt_begin * BK;
        const int tap = k0 / Ctot, rem = k0 - tap * Ctot;
        t_kh = tap / p.ks;
        t_kw = tap - t_kh * p.ks;
        t_second = rem >= p.C1;
        t_c = t_second ? rem - p.C1 : rem;
    }

    // A slab is staged in two steps so that its 1-KiB DMA pieces can be issued one at a time BETWEEN the MFMAs of the
    // main loop: issue_prep() advances the (wave-uniform) tap state and fixes the slab's offsets, issue_piece(q)
    // issues piece q (A pieces first, then this wave's B pieces).  Issuing all pieces of all 8 waves in one burst
    // right after the barrier kept the CU's address path busy for ~500 cycles per slab with the matrix pipes idle.
    unsigned char* i_sb = smem;
    bool i_tail = false, i_second = false;
    int i_soffA = 0, i_soffB = 0;
    auto issue_prep = [&](int kt, int stage) {
        unsigned char* sb = smem + stage * STAGE;
        const bool last_tail = (kt == nk - 1) && ktail_from < BK;
        int soffA = kt * (BK * 2), soffB = kt * (BK * 2);
        bool second = false;
        if (p.a_mode == 1) {
            if (fast_tap) {
                if (t_dirty) {
                    const int cs = t_second ? p.C2 : p.C1;
#pragma unroll
                    for (int i = 0; i < GA; ++i) {
                        const int hh = a_h0[i] + t_kh, ww = a_w0[i] + t_kw;
                        const bool ok = a_ok[i] && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W;
                        const int hsrc = p.ups ? (hh >> 1) : hh, wsrc = p.ups ? (ww >> 1) : ww;
                        va[i] = ok ? (((a_ibase[i] + hsrc) * Ws + wsrc) * cs + a_kofs[i]) * 2 : OOB_OFF;
                    }
                    t_dirty = false;
                }
                second = t_second;
                soffA = t_c * 2;
                t_c += BK;
                if (t_c >= (t_second ? p.C2 : p.C1)) {          // next source or next tap
                    t_c = 0;
                    t_dirty = true;
                    if (!t_second && p.C2 > 0) {
                        t_second = true;
                    } else {
                        t_second = false;
                        if (++t_kw == p.ks) { t_kw = 0; ++t_kh; }
                    }
                }
            } else {           // generic path (conv_in: 8 padded input channels): per-lane tap decode, single source
#pragma unroll
                for (int i = 0; i < GA; ++i) {
                    const long k = (long)kt * BK + a_kofs[i];
                    const int kk = k < p.K ? (int)k : 0;
                    const int tap = kk / Ctot;
                    const int ci = kk - tap * Ctot;
                    const int kh = tap / p.ks, kw = tap - kh * p.ks;
                    const int hh = a_h0[i] + kh, ww = a_w0[i] + kw;
                    const bool ok = a_ok[i] && k < p.K && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W;
                    const int hsrc = p.ups ? (hh >> 1) : hh, wsrc = p.ups ? (ww >> 1) : ww;
                    va[i] = ok ? (((a_ibase[i] + hsrc) * Ws + wsrc) * p.C1 + ci) * 2 : OOB_OFF;
                }
                soffA = 0;
            }
        }
        i_sb = sb; i_tail = last_tail; i_second = second; i_soffA = soffA; i_soffB = soffB;
    };
    constexpr int NPIECE = GA + GB_HI;
    bool i_on = true;                               // false: nothing left to stage (the tail of the K loop)
    bool i_gate = true;                             // this call site's share of the staggered issue (see mma_issue)
    auto issue_piece = [&](const int q) {           // q is a compile-time constant at every call site
        if (!(i_on && i_gate)) return;
        if (q < GA) {
            const int qi = q < GA ? q : 0;
            const int v = (i_tail && a_kofs[qi] >= ktail_from) ? OOB_OFF : va[qi];
            lptr_t dst = (lptr_t)(i_sb + (wave * GA + q) * 1024);
            if (i_second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA2, dst, 16, v, i_soffA, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, dst, 16, v, i_soffA, 0, 0);
        } else {
            const int j = q - GA;
            if (j < gbw) {   // wave-uniform
                const int jj = j < GB_HI ? j : 0;
                const int v = (i_tail && b_kofs[jj] >= ktail_from) ? OOB_OFF : vb[jj];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lptr_t)(i_sb + BM * 128 + (b_blk0 + j) * 1024), 16, v,
                                                         i_soffB, 0, 0);
            }
        }
    };
    auto issue = [&](int kt, int stage) {
        issue_prep(kt, stage);
        i_on = true;
        i_gate = true;
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) issue_piece(q);
    };

    constexpr int NACC = SWAP ? TN : TM, MACC = SWAP ? TM : TN;
    f16v acc[NACC][MACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < MACC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: row (.. + l31) * 128 B, logical slot s*2 + hi of k-step s, swizzled by (row >> 1) & 7.
    // A ds_read_b128 is served in 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, same + 32): their 8 even and
    // 8 odd rows take 8 distinct (row >> 1) & 7 values, i.e. 16 distinct 16-B slots of the 256-B bank row.
    const int swz = (l31 >> 1) & 7;
    const int a_row = (wr * WM + l31) * 128;
    const int b_row = BM * 128 + (wc * WN + l31) * 128;
    int offs[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) offs[ks] = ((ks * 2 + hi) ^ swz) * 16;

    const int nloc = kt_end - kt_begin;
#pragma unroll
    for (int s = 0; s < PREFETCH; ++s)
        if (s < nloc) issue(kt_begin + s, s);

    // Residual prefetch (<= 8-wave tiles only: the 16-wave tile has no registers to spare): the epilogue's residual quads are
    // requested while the K loop runs, so their HBM latency hides under it instead of stalling every wave at the end of a short
    // loop.  WHEN they are requested matters as much: vmcnt retires in issue order, so loads queued behind the first slabs make
    // the first slab wait wait for them too — round 5's phase stamps (profiles/r05_gemm_tile_phase_timing.txt) showed a 128x160
    // workgroup of `proj 1280->1280 +res` at M = 4096 spending 8.9 us between its entry and its first MFMA: all 256 workgroups
    // were fetching their residual tiles (10.5 MB) from HBM in front of the first operand slab.  The loads now go out BEHIND
    // the last slab's pieces (RES_LATE: full 32-column tiles with 16-byte accesses, an exact count of NRES loads per wave, rows
    // past M clamped), and every slab wait after that point allows NRES more entries in flight.
    constexpr bool PRE_RES = SWAP && (NW <= 8) && (TM * TN <= 5);
    constexpr int NRES = TM * TN * 2;
    h4 rpre[PRE_RES ? TM * TN * 4 : 1];
    const half_t* Rb0 = p.residual ? p.residual + z0 * p.r_bs0 + z1 * p.r_bs1 : nullptr;
    const bool pre_res = PRE_RES && Rb0 != nullptr && p.vec4 && !p.geglu && p.splitk <= 1;
    const bool res_late = pre_res && p.rvec8 && (int)n0 + BN <= (int)p.N;
    bool r_inflight = false;                        // the NRES residual loads sit behind every slab piece issued so far
    auto issue_residual = [&]() {                   // exactly NRES 16-byte loads per wave (wave-uniform control flow)
        if constexpr (PRE_RES) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned m = (unsigned)min((int)m0 + wr * WM + i * 32 + l31, (int)p.M - 1);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int nc = (int)n0 + wc * WN + j * 32 + 16 * a + 8 * hi;
                        const uint4 v = *reinterpret_cast<const uint4*>(Rb0 + m * (unsigned)p.ldr + nc);
                        rpre[(i * TN + j) * 4 + 2 * a] = __builtin_bit_cast(h4, make_uint2(v.x, v.y));
                        rpre[(i * TN + j) * 4 + 2 * a + 1] = __builtin_bit_cast(h4, make_uint2(v.z, v.w));
                    }
            }
        }
    };
    // ONE issue site, inside the K loop (a second one in front of the loop would make the compiler guard the registers against
    // each other with a vmcnt(0)): iteration kt_r is the one whose first k-step sends the last pieces of the last slab; launches
    // whose slabs are all in flight when the loop starts (nloc <= PREFETCH + 1) issue in iteration 0, and a single-slab launch
    // (no loop iteration at all) reads its residual in the epilogue.  TILE_RES_EARLY: iteration 0 whatever the K length, without
    // the exact count (the slab waits then wait for the residual as well: round 4's behaviour, for A/B runs).
    const bool res_early = (p.pp_flags & TILE_RES_EARLY) != 0;
    const int kt_r = res_early ? 0 : max(nloc - 1 - PREFETCH, 0);
    bool r_loaded = false;
    // slab kt must have landed; later slabs (at most PREFETCH-1 of them) stay in flight, and so do the residual loads once they
    // have been issued behind the last slab (r_inflight)
    auto wait_slab = [&](const int kt) {
        const int later = min(nloc - 1 - kt, PREFETCH - 1);
        if (PRE_RES && r_inflight) {
            if (wave < N_HI) {
                constexpr int G = GA + GB_HI;
                if (PREFETCH >= 3 && later >= 2) wait_vmcnt<2 * G + NRES>();
                else if (PREFETCH >= 2 && later == 1) wait_vmcnt<G + NRES>();
                else wait_vmcnt<NRES>();
            } else {
                constexpr int G = GA + GB_LO;
                if (PREFETCH >= 3 && later >= 2) wait_vmcnt<2 * G + NRES>();
                else if (PREFETCH >= 2 && later == 1) wait_vmcnt<G + NRES>();
                else wait_vmcnt<NRES>();
            }
            return;
        }
        if (wave < N_HI) {
            constexpr int G = GA + GB_HI;
            if (PREFETCH >= 3 && later >= 2) wait_vmcnt<2 * G>();
            else if (PREFETCH >= 2 && later == 1) wait_vmcnt<G>();
            else wait_vmcnt<0>();
        } else {
            constexpr int G = GA + GB_LO;
            if (PREFETCH >= 3 && later >= 2) wait_vmcnt<2 * G>();
            else if (PREFETCH >= 2 && later == 1) wait_vmcnt<G>();
            else wait_vmcnt<0>();
        }
    };
    auto ldfrag = [&](const unsigned char* sb, const int off, h8 (&af)[TM], h8 (&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const h8*>(sb + a_row + i * 4096 + off);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const h8*>(sb + b_row + j * 4096 + off);
    };
    auto mma = [&](const h8 (&af)[TM], const h8 (&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (SWAP)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[j][i], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
    };

    // The MFMAs of one k-step with one DMA piece of the slab being staged after each of them.  The CU's address/data
    // path moves 64 B/clk: a 56-KiB slab keeps it busy ~900 cycles of the ~1300 the slab's MFMAs need, and a wave whose
    // DMA instruction meets a full queue stalls (tools/gemm_timing.py: 560 of 2170 cycles per slab), during which only
    // the OTHER wave of its SIMD can feed the matrix pipe.  So the two waves of a SIMD (w and w + NW/2) take turns: the
    // first half of the workgroup stages in the last k-step of a slab, the second half in the first k-step of the
    // next one.  (Spreading every wave's pieces over three k-steps was measured too: better for a lone workgroup,
    // worse under load — the pieces land later and the K = 320 loops are only five slabs long.)
    auto mma_issue = [&](const h8 (&af)[TM], const h8 (&bf)[TN], const bool gate) {
        i_gate = gate;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (SWAP)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[j][i], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                if (i * TN + j < NPIECE) issue_piece(i * TN + j);
            }
#pragma unroll
        for (int q = TM * TN; q < NPIECE; ++q) issue_piece(q);
    };

    // Register-double-buffered fragments (<= 8 waves: the 16-wave tile has 128 registers per lane and relies on its four
    // waves per SIMD instead).  The fragments of k-step s+1 are requested from LDS BEFORE the MFMAs of k-step s are
    // issued, so the LDS round trip always sits behind 5-10 MFMAs of the same wave; the first version read each
    // fragment immediately before the MFMA that consumed it (s_waitcnt lgkmcnt(0) in front of every MFMA) and a lone
    // workgroup ran at 38 % of the MFMA rate (tools/tile_probe.py).
#ifdef VSX_GEMM_TIMING
    long t_seg[4] = {0, 0, 0, 0}, t_last = 0, t_loop0 = 0;
#endif
    constexpr bool PIPE = (NW <= 8) && (TM * TN < 10);
    if constexpr (PIPE) {
        h8 af0[TM], bf0[TN], af1[TM], bf1[TN];
        wait_slab(0);
        __builtin_amdgcn_s_barrier();
        if (PREFETCH < nloc) issue(kt_begin + PREFETCH, PREFETCH & (NSTAGE - 1));
        ldfrag(smem, offs[0], af0, bf0);
        const bool late = NW >= 8 && wave >= NW / 2;      // second wave of each SIMD: stages one k-step later
        i_on = false;
#ifdef VSX_GEMM_TIMING
        t_last = t_loop0 = (long)clock64();
#endif
        for (int kt = 0; kt + 1 < nloc; ++kt) {
            const unsigned char* sb = smem + (kt & (NSTAGE - 1)) * STAGE;
            ldfrag(sb, offs[1], af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mma_issue(af0, bf0, late);                    // (late waves: the slab prepared in the previous iteration)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PRE_RES) {
                if (res_late && kt == kt_r) {                     // every piece of every slab is out: the residual behind them
                    issue_residual();
                    VSX_VMEM_NOTE(NRES);
                    r_inflight = !res_early;
                    r_loaded = true;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ldfrag(sb, offs[2], af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            mma(af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            ldfrag(sb, offs[3], af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mma(af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            TSTAMP(0);
            wait_slab(kt + 1);
            // every fragment read of slab kt has returned before its ring slot is handed back to the DMA engine
            __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)
            TSTAMP(1);
            __builtin_amdgcn_s_barrier();
            TSTAMP(2);
            i_on = kt + 1 + PREFETCH < nloc;
            if (i_on) issue_prep(kt_begin + kt + 1 + PREFETCH, (kt + 1 + PREFETCH) & (NSTAGE - 1));
            ldfrag(smem + ((kt + 1) & (NSTAGE - 1)) * STAGE, offs[0], af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            mma_issue(af1, bf1, !late);
            __builtin_amdgcn_sched_barrier(0);
            TSTAMP(3);
        }
        if (nloc > 0) {                                  // last slab: nothing left to stage
            const unsigned char* sb = smem + ((nloc - 1) & (NSTAGE - 1)) * STAGE;
            ldfrag(sb, offs[1], af1, bf1);
            mma(af0, bf0);
            ldfrag(sb, offs[2], af0, bf0);
            mma(af1, bf1);
            ldfrag(sb, offs[3], af1, bf1);
            mma(af0, bf0);
            mma(af1, bf1);
        }
    } else {
        const bool late16 = NW >= 16 && ((wave >> 2) & 1);
#ifdef VSX_GEMM_TIMING
        t_last = t_loop0 = (long)clock64();
#endif
        for (int kt = 0; kt < nloc; ++kt) {
            wait_slab(kt);
            TSTAMP(1);
            __builtin_amdgcn_s_barrier();   // everybody's part of slab kt is in LDS; slot (kt-1)%NSTAGE is free
            TSTAMP(2);
            // half of the waves of every SIMD (w>>2 even) stage right after the barrier, the other half one k-step
            // later: the CU's DMA queue is fed by 8 waves at a time while the other 8 keep the matrix pipes busy
            i_on = kt + PREFETCH < nloc;
            i_gate = true;
            if (i_on) issue_prep(kt_begin + kt + PREFETCH, (kt + PREFETCH) & (NSTAGE - 1));
            if (!late16) {
#pragma unroll
                for (int q = 0; q < NPIECE; ++q) issue_piece(q);
            }
            TSTAMP(3);
            const unsigned char* sb = smem + (kt & (NSTAGE - 1)) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                h8 af[TM], bf[TN];
                ldfrag(sb, offs[ks], af, bf);
                mma(af, bf);
                if (ks == 0 && late16) {
#pragma unroll
                    for (int q = 0; q < NPIECE; ++q) issue_piece(q);
                }
            }
            TSTAMP(0);
        }
    }

#ifdef VSX_GEMM_TIMING
    long* const t_out = p.ws && p.splitk <= 1 ? reinterpret_cast<long*>(p.ws) + ((long)blockIdx.x * NW + wave) * 8 : nullptr;
    if (lane == 0 && t_out) {
        for (int k = 0; k < 4; ++k) t_out[k] = t_seg[k];
        t_out[4] = t_begin; t_out[5] = t_loop0; t_out[6] = (long)clock64();
    }
#define TSTAMP_END() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0 && t_out) t_out[7] = (long)clock64(); } while (0)
#else
#define TSTAMP_END() do { } while (0)
#endif
    // ---------------- epilogue (32-bit element offsets: the host guarantees M*ldc, M*ldr < 2^31) ----------------
    half_t* Cb = p.C + z0 * p.c_bs0 + z1 * p.c_bs1;
    const half_t* Rb = p.residual ? p.residual + z0 * p.r_bs0 + z1 * p.r_bs1 : nullptr;
    const int Mi = (int)p.M, Ni = (int)p.N;
    const int ldc = (int)p.ldc, ldr = (int)p.ldr;
    const int mb = (int)m0, nb0 = (int)n0;

    if constexpr (SWAP) {
        if (p.splitk > 1) {
            // split-K slice: raw fp32 partial sums to ws[split][m][n]; vsx's reduce kernel applies the epilogue
            float* wsl = p.ws + (size_t)split * (size_t)Mi * (size_t)Ni;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mb + wr * WM + i * 32 + l31;
                if (m >= Mi) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nb = nb0 + wc * WN + j * 32 + 8 * g + 4 * hi;
                        if (nb + 3 < Ni) {
                            f4v o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = acc[j][i][4 * g + e] * p.alpha;
                            *reinterpret_cast<f4v*>(wsl + (size_t)m * Ni + nb) = o;
                        }
                    }
            }
            return;
        }
        // lane owns row m (column of the C^T tile); register quad g of tile (j, i) holds 4 consecutive columns.
        // The bias quads of the wave's full 32-column tiles are loaded HERE, all of them before the first store: a load
        // issued between the stores of two tiles is waited for with vmcnt(0), i.e. together with the stores in flight
        // (gemm_pp.hip has the measurements).  Small waves only (the register budget of the 16-wave kernel is 128).
        constexpr bool PRE_BIAS = (NW <= 8) && (TM * TN <= 5);
        h4 bpre[PRE_BIAS ? TN : 1][4];
        const bool pre_bias = PRE_BIAS && p.bias != nullptr && p.vec8 && !p.geglu;
        if constexpr (PRE_BIAS) {
            if (pre_bias) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nb = nb0 + wc * WN + j * 32 + 8 * g + 4 * hi;
                        bpre[j][g] = nb + 3 < Ni ? *reinterpret_cast<const h4*>(p.bias + nb) : h4{};
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mb + wr * WM + i * 32 + l31;
            if (m >= Mi) continue;
            // Row bases as 32-bit BYTE offsets from wave-uniform pointers (the host guarantees M * ldc, M * ldr < 2^31 elements):
            // the accesses take the scalar-base form, and a row costs three registers instead of three 64-bit pointers — the
            // 16-wave kernel (128 registers) spilled one of them (videoswap_amd/build.py refuses that now).
            const unsigned cbyte = (unsigned)m * (unsigned)ldc * 2u;
            const unsigned rbyte = (unsigned)m * (unsigned)ldr * 2u;
            const unsigned vbyte = p.rowvec ? ((unsigned)m / (unsigned)p.rows_per_vec) * (unsigned)Ni * 2u : 0u;
            const bool rrow = Rb != nullptr, rv = p.rowvec != nullptr;
            auto crow_at = [&](const int n) { return reinterpret_cast<half_t*>(reinterpret_cast<char*>(Cb) + (cbyte + 2u * (unsigned)n)); };
            auto rrow_at = [&](const int n) { return reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(Rb) + (rbyte + 2u * (unsigned)n)); };
            auto rv_at = [&](const int n) { return reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(p.rowvec) + (vbyte + 2u * (unsigned)n)); };
            // LayerNorm folded into the GEMM: out = rs * acc + rt * c1[n] (then bias / row vector / residual as usual)
            const float* cvp = p.rowscale ? p.colvec : nullptr;
            const float rs = cvp ? p.rowscale[2 * (unsigned)m] : 1.f, rt = cvp ? p.rowscale[2 * (unsigned)m + 1] : 0.f;
            if (p.geglu) {
                // tile j: registers 0-7 are h of 16 output columns, registers 8-15 the matching g
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int nb = nb0 + (wc * TN + j) * 16 + 8 * q + 4 * hi;
                        if (nb >= Ni) continue;
                        float hv[4], gv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hv[e] = acc[j][i][4 * q + e] * p.alpha;
                            gv[e] = acc[j][i][8 + 4 * q + e] * p.alpha;
                        }
                        if (cvp) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (nb + e < Ni) {
                                    hv[e] = __builtin_fmaf(rs, hv[e], rt * cvp[nb + e]);
                                    gv[e] = __builtin_fmaf(rs, gv[e], rt * cvp[Ni + nb + e]);
                                }
                            }
                        }
                        if (p.vec4 && nb + 3 < Ni) {
                            if (p.bias) {
                                const h4 bh = *reinterpret_cast<const h4*>(p.bias + nb);
                                const h4 bg = *reinterpret_cast<const h4*>(p.bias + Ni + nb);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { hv[e] += (float)bh[e]; gv[e] += (float)bg[e]; }
                            }
                            const vsx_f2 g01 = gelu_erf_f2(vsx_f2{gv[0], gv[1]}), g23 = gelu_erf_f2(vsx_f2{gv[2], gv[3]});
                            float o[4] = {hv[0] * g01[0], hv[1] * g01[1], hv[2] * g23[0], hv[3] * g23[1]};
                            if (rrow) {
                                const h4 b = *reinterpret_cast<const h4*>(rrow_at(nb));
#pragma unroll
                                for (int e = 0; e < 4; ++e) o[e] += (float)b[e];
                            }
                            h4 pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pk[e] = (half_t)o[e];
                            *reinterpret_cast<h4*>(crow_at(nb)) = pk;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int n = nb + e;
                                if (n < Ni) {
                                    const float bh = p.bias ? (float)p.bias[n] : 0.f;
                                    const float bg = p.bias ? (float)p.bias[Ni + n] : 0.f;
                                    float o = (hv[e] + bh) * gelu_erf_f(gv[e] + bg);
                                    if (rrow) o += (float)*rrow_at(n);
                                    *crow_at(n) = (half_t)o;
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the 40 register quads from being processed at once
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (p.vec8 && nb0 + wc * WN + j * 32 + 31 < Ni) {
                    // Full 32-column tile: the lane pair (l, l+32) exchanges register quads with v_permlane32_swap so
                    // that each lane owns 8 consecutive columns and stores 16 bytes — two dwordx4 stores per tile and
                    // lane instead of four dwordx2 (the epilogue is store-ISSUE-bound: every store instruction of the
                    // row-per-lane layout touches 32 different rows).
                    unsigned pkw[4][2];
                    h4 rq[4];          // residual register quads (16-byte loads in the exchanged layout, swapped back)
                    if (rrow && p.rvec8) {
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            uint2 lo2, hi2;
                            if (PRE_RES && r_loaded) {
                                lo2 = __builtin_bit_cast(uint2, rpre[PRE_RES ? (i * TN + j) * 4 + 2 * a : 0]);
                                hi2 = __builtin_bit_cast(uint2, rpre[PRE_RES ? (i * TN + j) * 4 + 2 * a + 1 : 0]);
                            } else {
                                const uint4 v = *reinterpret_cast<const uint4*>(rrow_at(nb0 + wc * WN + j * 32 + 16 * a + 8 * hi));
                                lo2 = make_uint2(v.x, v.y);
                                hi2 = make_uint2(v.z, v.w);
                            }
                            const auto s0 = __builtin_amdgcn_permlane32_swap(lo2.x, hi2.x, false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(lo2.y, hi2.y, false, false);
                            rq[2 * a] = __builtin_bit_cast(h4, make_uint2(s0[0], s1[0]));
                            rq[2 * a + 1] = __builtin_bit_cast(h4, make_uint2(s0[1], s1[1]));
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nb = nb0 + wc * WN + j * 32 + 8 * g + 4 * hi;
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = acc[j][i][4 * g + e] * p.alpha;
                        if (cvp) {
                            const f4v c = *reinterpret_cast<const f4v*>(cvp + nb);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(rs, o[e], rt * c[e]);
                        }
                        if (p.bias) {
                            h4 b;
                            if (PRE_BIAS && pre_bias) b = bpre[PRE_BIAS ? j : 0][g];
                            else b = *reinterpret_cast<const h4*>(p.bias + nb);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += (float)b[e];
                        }
                        if (rv) {
                            const h4 b = *reinterpret_cast<const h4*>(rv_at(nb));
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += (float)b[e];
                        }
                        if (rrow) {
                            h4 b;
                            if (p.rvec8) b = rq[g];
                            else if (PRE_RES && r_loaded) b = rpre[PRE_RES ? (i * TN + j) * 4 + g : 0];
                            else b = *reinterpret_cast<const h4*>(rrow_at(nb));
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += (float)b[e];
                        }
                        h4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = (half_t)o[e];
                        const uint2 w = __builtin_bit_cast(uint2, pk);
                        pkw[g][0] = w.x;
                        pkw[g][1] = w.y;
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        // quads 2a (columns 16a + 4hi ..) and 2a+1 (16a + 8 + 4hi ..): after the swap the lower lane
                        // of the pair holds columns 16a .. 16a+7, the upper lane 16a+8 .. 16a+15
                        const auto s0 = __builtin_amdgcn_permlane32_swap(pkw[2 * a][0], pkw[2 * a + 1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(pkw[2 * a][1], pkw[2 * a + 1][1], false, false);
                        const uint4 out = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        const int nc = nb0 + wc * WN + j * 32 + 16 * a + 8 * hi;
                        *reinterpret_cast<uint4*>(crow_at(nc)) = out;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = nb0 + wc * WN + j * 32 + 8 * g + 4 * hi;
                    if (nb >= Ni) continue;
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[j][i][4 * g + e] * p.alpha;
                    if (cvp) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (nb + e < Ni) o[e] = __builtin_fmaf(rs, o[e], rt * cvp[nb + e]);
                    }
                    if (p.vec4 && nb + 3 < Ni) {
                        if (p.bias) {
                            const h4 b = *reinterpret_cast<const h4*>(p.bias + nb);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += (float)b[e];
                        }
                        if (rv) {
                            const h4 b = *reinterpret_cast<const h4*>(rv_at(nb));
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += (float)b[e];
                        }
                        if (rrow) {
                            h4 b;
                            if (PRE_RES && r_loaded) b = rpre[PRE_RES ? (i * TN + j) * 4 + g : 0];
                            else b = *reinterpret_cast<const h4*>(rrow_at(nb));
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += (float)b[e];
                        }
                        h4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = (half_t)o[e];
                        *reinterpret_cast<h4*>(crow_at(nb)) = pk;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int n = nb + e;
                            if (n < Ni) {
                                float v = o[e];
                                if (p.bias) v += (float)p.bias[n];
                                if (rv) v += (float)*rv_at(n);
                                if (rrow) v += (float)*rrow_at(n);
                                *crow_at(n) = (half_t)v;
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // bound the live range of the epilogue's loads (no spills)
            }
        }
    } else {
        // transposed store: lane owns column n, register quad g of tile (i, j) holds rows mg..mg+3
        const unsigned rpi = (unsigned)p.c_rows_per_img;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nb0 + wc * WN + j * 32 + l31;
            if (n >= Ni) continue;
            const float bv = p.bias ? (float)p.bias[n] : 0.f;
            const float cvn = p.rowscale ? p.colvec[n] : 0.f;
            half_t* ccol = Cb + (long)n * p.ldc;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int mg = mb + wr * WM + i * 32 + 8 * g + 4 * hi;
                    if (mg >= Mi) continue;
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][4 * g + e] * p.alpha;
                        if (p.rowscale) {
                            const unsigned mr = (unsigned)min(mg + e, Mi - 1);
                            v = __builtin_fmaf(p.rowscale[2 * mr], v, p.rowscale[2 * mr + 1] * cvn);
                        }
                        o[e] = v + bv;
                    }
                    if (p.c_pack4 && mg + 3 < Mi) {
                        const unsigned img = (unsigned)mg / rpi;
                        const unsigned mm = (unsigned)mg - img * rpi;
                        h4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = (half_t)o[e];
                        *reinterpret_cast<h4*>(ccol + (long)img * p.c_img_stride + mm) = pk;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int m = mg + e;
                            if (m < Mi) {
                                const unsigned img = (unsigned)m / rpi;
                                const unsigned mm = (unsigned)m - img * rpi;
                                ccol[(long)img * p.c_img_stride + mm] = (half_t)o[e];
                            }
                        }
                    }
                }
            }
        }
    }
    TSTAMP_END();
}

// split-K combine: out[m, n] = sum_z ws[z][m][n] + bias + rowvec + residual (fixed summation order: deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;          // one thread per 4 consecutive columns
    const int nq = (int)(p.N >> 2);
    if (q >= p.M * nq) return;
    const int m = (int)(q / nq), nb = (int)(q - (long)m * nq) * 4;
    const size_t slab = (size_t)p.M * (size_t)p.N;
    f4v acc = *reinterpret_cast<const f4v*>(p.ws + (size_t)m * p.N + nb);
    for (int z = 1; z < p.splitk; ++z) {
        const f4v v = *reinterpret_cast<const f4v*>(p.ws + z * slab + (size_t)m * p.N + nb);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
    if (p.rowscale) {
        const float rs = p.rowscale[2 * (size_t)m], rt = p.rowscale[2 * (size_t)m + 1];
        const f4v c = *reinterpret_cast<const f4v*>(p.colvec + nb);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(rs, acc[e], rt * c[e]);
    }
    if (p.bias) {
        const h4 b = *reinterpret_cast<const h4*>(p.bias + nb);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (float)b[e];
    }
    if (p.rowvec) {
        const h4 b = *reinterpret_cast<const h4*>(p.rowvec + (size_t)(m / p.rows_per_vec) * p.N + nb);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (float)b[e];
    }
    if (p.residual) {
        const h4 b = *reinterpret_cast<const h4*>(p.residual + (size_t)m * p.ldr + nb);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (float)b[e];
    }
    h4 pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk[e] = (half_t)acc[e];
    *reinterpret_cast<h4*>(p.C + (size_t)m * p.ldc + nb) = pk;
}

// Split-K plan: small-M, long-K problems (the 8x8 / 16x16 UNet levels: M = 1-4 K rows, K up to 23 040) cannot fill
// 256 CUs with output tiles alone; slicing K gives every CU a 128x320 tile to work on.  Returns the slice count (1 =
// no split) for a problem whose wide tiles would number `tiles`.
inline int plan_splitk(const vsx_gemm_desc* d, long tiles, bool eligible) {
    const long nk = (d->K + BK - 1) / BK;
    if (eligible && tune_splits() > 0) return tune_splits() <= nk ? tune_splits() : (int)nk;
    if (!eligible || tiles >= 160) return 1;
    int s = (int)((256 + tiles - 1) / tiles);
    if (s > 8) s = 8;
    while (s > 1 && nk / s < 12) --s;     // keep >= 12 slabs (768 k) per slice
    return s;
}

int g_last_xcd_gm = 0;       // diagnostics (tools/cpu_check/check_gemm_api.cpp reads it): the arrangement of the last tile-kernel launch

// XCD BLOCK GRID (gemm_kernel): which gm x (8 / gm) arrangement of the XCDs over (tile_m, np) moves the fewest operand bytes
// through the L2s.  Per XCD: (tiles_m / gm) x (K slices its column block meets) activation panels + NP * gm / 8 weight
// panels.  Returns 0 (the linear walk: the same blocks as gm = 8 whenever tiles_m is a multiple of 8) unless another
// arrangement is valid and cheaper.  Option "xcd_walk" (VSX_XCD_WALK) = 0 keeps the linear walk (A/B runs).
int plan_xcd_grid(const GemmParams& p, const long tiles_m, const int bm, const int bn, const long nbatch) {
    if (nbatch != 1 || gemm_option("xcd_walk") == 0) return 0;
    const int splits = p.splitk > 1 ? p.splitk : 1;
    const long np_total = (long)p.tiles_n * splits;
    if ((tiles_m * np_total) % 8 != 0 || tiles_m * np_total < 16) return 0;
    const long nk = (p.K + BK - 1) / BK;
    const double kslice = (double)(splits > 1 ? (long)p.nk_per * BK : nk * BK);
    const double taps = p.a_mode == 1 ? (double)(p.ks * p.ks) / (double)(p.stride * p.stride) : 1.0;    // windows of neighbouring pixels overlap
    const double a_panel = (double)bm * kslice * 2.0 / (taps < 1.0 ? 1.0 : taps);
    const double b_panel = (double)bn * kslice * 2.0;
    auto cost = [&](const int gm) -> double {
        const int gn = 8 / gm;
        if (tiles_m % gm != 0 || np_total % gn != 0) return -1.0;
        const long RB = tiles_m / gm, CB = np_total / gn;
        long nsp = 1;                                   // K slices the widest-spanning column block meets
        for (int b = 0; b < gn; ++b) {
            const long lo = b * CB, hi = lo + CB - 1;
            nsp = std::max(nsp, hi / p.tiles_n - lo / p.tiles_n + 1);
        }
        return (double)(RB * nsp) * a_panel + (double)CB * b_panel;
    };
    const double legacy = (tiles_m % 8 == 0) ? cost(8)
                                             : (double)((tiles_m + 7) / 8 + 1) * splits * a_panel + (double)np_total * b_panel;
    int best = 0;
    double best_cost = legacy;
    for (int gm = 4; gm >= 1; gm >>= 1) {
        const double c = cost(gm);
        if (c >= 0.0 && c < 0.95 * best_cost) { best = gm; best_cost = c; }
    }
    return best;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool SWAP, int NSTAGE>
int launch(GemmParams& p, long tiles_m, long nbatch, hipStream_t stream) {
    constexpr size_t smem = (size_t)NSTAGE * (BM + BN) * 128;
    static_assert(smem <= 160 * 1024, "LDS ring exceeds 160 KiB");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, WAVES_M, WAVES_N, SWAP, NSTAGE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    p.pp_flags = (int)(gemm_option("pp_sched") & TILE_RES_EARLY);
    p.tiles_m = (int)tiles_m;
    p.xcd_gm = g_last_xcd_gm = plan_xcd_grid(p, tiles_m, BM, BN, nbatch);
    dim3 grid((unsigned)(tiles_m * p.tiles_n), 1, (unsigned)(p.splitk > 1 ? p.splitk : nbatch));
    if (p.xcd_gm > 0) grid = dim3((unsigned)(tiles_m * p.tiles_n * (p.splitk > 1 ? p.splitk : 1)), 1, 1);
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WAVES_M, WAVES_N, SWAP, NSTAGE>), grid, dim3(WAVES_M * WAVES_N * 64), smem,
                       stream, p);
    return vsx_check_launch("vsx_gemm_f16");
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_tile(GemmParams& p, long M, long cols, long nbatch, hipStream_t stream, bool deep = false) {
    p.tiles_n = (int)((cols + BN - 1) / BN);
    const long tiles_m = (M + BM - 1) / BM;
    // ring depth: two 128-byte-row slabs for the big tiles (57-74 KiB per slab), four for the small ones
    constexpr int NST = ((BM + BN) * 128 * 4 <= 96 * 1024) ? 4 : 2;
    // `deep` (128x160 only): four slots = one workgroup per CU with three slabs in flight, for launches that put at most
    // one workgroup on a CU anyway (two slots: two workgroups of 72 KiB per CU hide each other's latency)
    if constexpr (BM == 128 && BN == 160) {
        if (deep && p.c_mode != 1) return launch<BM, BN, WAVES_M, WAVES_N, true, 4>(p, tiles_m, nbatch, stream);
    }
    if (p.c_mode == 1) return launch<BM, BN, WAVES_M, WAVES_N, false, NST>(p, tiles_m, nbatch, stream);
    return launch<BM, BN, WAVES_M, WAVES_N, true, NST>(p, tiles_m, nbatch, stream);
}

// ---- instrumentation (bench.py roofline): hipEvent pairs around sampled launches ----
struct ProfState {
    bool on = false;
    bool paused = false;
    long max_samples = 0;
    long stride = 1, seen = 0;
    long n = 0;
    double flop = 0.0;
    std::vector<hipEvent_t>* ev = nullptr;  // 2 per sample
    std::vector<double>* work = nullptr;    // 2 per sample: algorithmic FLOP, algorithmic bytes (vsx_prof_collect_roofline)
};

ProfState g_prof;

}  // namespace

extern "C" int vsx_prof_enable(int64_t on, int64_t max_samples) {
    if (!g_prof.ev) g_prof.ev = new std::vector<hipEvent_t>();
    if (!g_prof.work) g_prof.work = new std::vector<double>();
    g_prof.work->clear();
    g_prof.on = on != 0;
    g_prof.stride = on > 1 ? on : 1;        // on = k > 1: bracket every k-th launch only
    g_prof.seen = 0;
    g_prof.max_samples = max_samples;
    g_prof.n = 0;
    g_prof.flop = 0.0;
    return VSX_OK;
}

// suspend / resume sampling without touching what has been collected (HIP-graph capture and replayed calls)
extern "C" int vsx_prof_pause(int64_t paused) {
    g_prof.paused = paused != 0;
    return VSX_OK;
}

// Per sampled launch: duration t, algorithmic FLOP f and algorithmic bytes b (A once + weights once + C once + residual once; a
// convolution reads every input pixel once).  A launch cannot finish before max(f / peak_flops, b / peak_bytes_per_s): the sum of
// those floors over the samples is what the same launches would take on BOTH rooflines at once (`floor_ms`), and
// `byte_bound_ms` is the measured time of the launches whose byte floor is the larger one.
extern "C" int vsx_prof_collect_roofline(double peak_flops, double peak_bytes_per_s, int64_t* n_launches, double* total_ms,
                                         double* total_flop, double* total_bytes, double* floor_ms, double* byte_bound_ms) {
    double ms = 0.0, bytes = 0.0, floor = 0.0, bb = 0.0;
    if (g_prof.ev) {
        for (long i = 0; i < g_prof.n; ++i) {
            hipEvent_t a = (*g_prof.ev)[2 * i], b = (*g_prof.ev)[2 * i + 1];
            if (hipEventSynchronize(b) != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "prof: event sync failed");
            float t = 0.f;
            if (hipEventElapsedTime(&t, a, b) != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "prof: elapsed failed");
            ms += t;
            if (g_prof.work && (long)g_prof.work->size() >= 2 * (i + 1) && peak_flops > 0.0 && peak_bytes_per_s > 0.0) {
                const double f = (*g_prof.work)[2 * i], by = (*g_prof.work)[2 * i + 1];
                const double tf = f / peak_flops, tb = by / peak_bytes_per_s;
                bytes += by;
                floor += 1e3 * (tf > tb ? tf : tb);
                if (tb > tf) bb += t;
            }
        }
    }
    if (n_launches) *n_launches = g_prof.n;
    if (total_ms) *total_ms = ms;
    if (total_flop) *total_flop = g_prof.flop;
    if (total_bytes) *total_bytes = bytes;
    if (floor_ms) *floor_ms = floor;
    if (byte_bound_ms) *byte_bound_ms = bb;
    g_prof.n = 0;
    g_prof.flop = 0.0;
    if (g_prof.work) g_prof.work->clear();
    return VSX_OK;
}

extern "C" int vsx_prof_collect(int64_t* n_launches, double* total_ms, double* total_flop) {
    return vsx_prof_collect_roofline(0.0, 0.0, n_launches, total_ms, total_flop, nullptr, nullptr, nullptr);
}

namespace vsxg {
namespace {
struct Option { const char* name; const char* env; long value; bool init; };
Option g_options[] = {{"gemm_pp", "VSX_GEMM_PP", 1, false}, {"pp_sched", "VSX_PP_SCHED", 0, false},
                      {"tile_tune", "VSX_TUNE_TILE", 0, false}, {"xcd_walk", "VSX_XCD_WALK", 1, false},
                      {"attn_qb", "VSX_ATTN_QB", 0, false}, {"temporal_out", "VSX_TEMPORAL_OUT", 0, false},
                      {"attn_o16", "VSX_ATTN_O16", 0, false}, {"gn_fuse", "VSX_GN_FUSE", 0, false},
                      {"gemm_ws", "VSX_GEMM_WS", 1, false}, {"ws_waves", "VSX_WS_WAVES", 10, false}};
Option* find_option(const char* name) {
    for (auto& o : g_options)
        if (strcmp(o.name, name) == 0) {
            if (!o.init) {
                const char* e = getenv(o.env);
                if (e) o.value = atol(e);
                o.init = true;
            }
            return &o;
        }
    return nullptr;
}
}  // namespace
long gemm_option(const char* name) {
    Option* o = find_option(name);
    return o ? o->value : 0;
}
}  // namespace vsxg

extern "C" int vsx_set_option(const char* name, int64_t value) {
    auto* o = name ? vsxg::find_option(name) : nullptr;
    if (!o) return vsx_fail(VSX_E_BADSHAPE, "vsx_set_option: unknown option '%s'", name ? name : "(null)");
    o->value = (long)value;
    return VSX_OK;
}

static int pp_mode() { return (int)vsxg::gemm_option("gemm_pp"); }

extern "C" int64_t vsx_gemm_workspace(const vsx_gemm_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    const long nbatch = d->batch0 * d->batch1;
    const long cols = d->geglu ? 2 * d->N : d->N;
    const bool vec4 = d->ldc % 4 == 0 && d->N % 4 == 0 && (!d->residual || d->ldr % 4 == 0);
    const bool eligible = cols % 320 == 0 && nbatch == 1 && !d->geglu && d->c_mode == 0 && vec4;
    const long tiles = ((d->M + 127) / 128) * ((cols + 319) / 320);
    const int s = plan_splitk(d, tiles, eligible);
    return s > 1 ? (int64_t)s * d->M * d->N * (int64_t)sizeof(float) : 0;
}

static int gemm_impl(const vsx_gemm_desc* d, vsx_stream_t stream_, bool dry, int64_t* parts_out);

extern "C" int vsx_gemm_f16(const vsx_gemm_desc* d, vsx_stream_t stream) { return gemm_impl(d, stream, false, nullptr); }

// How many row-statistics parts would this launch write?  The same checks and the same kernel choice as the launch itself
// (gemm_impl stops in front of it); 0 whenever the problem does not go to the persistent kernel with an eligible epilogue.
extern "C" int64_t vsx_gemm_rowstats_parts(const vsx_gemm_desc* d) {
    int64_t parts = 0;
    if (!d || d->M <= 0) return 0;
    vsx_gemm_desc probe = *d;
    probe.rowstats = nullptr;
    probe.rowstats_parts = 0;
    return gemm_impl(&probe, nullptr, true, &parts) == VSX_OK ? parts : 0;
}

static int gemm_impl(const vsx_gemm_desc* d, vsx_stream_t stream_, const bool dry, int64_t* parts_out) {
    hipStream_t stream = (hipStream_t)stream_;
    VSX_REQUIRE(d != nullptr, VSX_E_BADSHAPE, "gemm: null descriptor");
    VSX_REQUIRE(d->M >= 0 && d->N > 0 && d->K > 0, VSX_E_BADSHAPE, "gemm: bad M/N/K %ld/%ld/%ld",
                (long)d->M, (long)d->N, (long)d->K);
    if (d->M == 0) return VSX_OK;
    VSX_REQUIRE(d->A && d->B && d->C, VSX_E_BADSHAPE, "gemm: null operand");
    VSX_REQUIRE(vsx_aligned16(d->A) && vsx_aligned16(d->B) && vsx_aligned16(d->C), VSX_E_BADSHAPE,
                "gemm: operands must be 16-byte aligned");
    VSX_REQUIRE(d->batch0 >= 1 && d->batch1 >= 1, VSX_E_BADSHAPE, "gemm: batch counts must be >= 1");
    VSX_REQUIRE(d->K % 8 == 0 && d->ldb % 8 == 0, VSX_E_BADSHAPE, "gemm: K (%ld) and ldb (%ld) must be multiples of 8",
                (long)d->K, (long)d->ldb);
    VSX_REQUIRE(d->b_bs0 % 8 == 0 && d->b_bs1 % 8 == 0 && d->a_bs0 % 8 == 0 && d->a_bs1 % 8 == 0, VSX_E_BADSHAPE,
                "gemm: batch strides must be multiples of 8 elements");

    GemmParams p{};
    p.A = (const half_t*)d->A;
    p.A2 = (const half_t*)d->A2;
    p.B = (const half_t*)d->B;
    p.C = (half_t*)d->C;
    p.bias = (const half_t*)d->bias;
    p.rowvec = (const half_t*)d->rowvec;
    p.residual = (const half_t*)d->residual;
    p.rowscale = (const float*)d->rowscale;
    p.colvec = (const float*)d->colvec;
    VSX_REQUIRE((d->rowscale == nullptr) == (d->colvec == nullptr), VSX_E_BADSHAPE, "gemm: rowscale and colvec come together");
    VSX_REQUIRE(!d->rowscale || (d->batch0 * d->batch1 == 1 && d->a_mode == 0 && vsx_aligned16(d->colvec)), VSX_E_UNSUPPORTED,
                "gemm: rowscale / colvec (LayerNorm folded into a Linear) need an unbatched plain GEMM and a 16-byte aligned colvec");
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr;
    p.a_bs0 = d->a_bs0; p.a_bs1 = d->a_bs1; p.b_bs0 = d->b_bs0; p.b_bs1 = d->b_bs1;
    p.c_bs0 = d->c_bs0; p.c_bs1 = d->c_bs1; p.r_bs0 = d->r_bs0; p.r_bs1 = d->r_bs1;
    p.batch1 = (int)d->batch1;
    p.a_mode = (int)d->a_mode;
    p.geglu = (int)d->geglu;
    p.c_mode = (int)d->c_mode;
    p.alpha = (float)d->alpha;
    p.rows_per_vec = d->rows_per_vec > 0 ? d->rows_per_vec : 1;
    p.c_rows_per_img = d->c_rows_per_img;
    p.c_img_stride = d->c_img_stride;

    const long nbatch = d->batch0 * d->batch1;
    // (the transposed store of c_mode 1 addresses C with 64-bit offsets, and its ldc is the row length of one image's V^T,
    // not a pitch of the M rows: B = 2 x 64 frames x 4096 tokens must not trip over M * ldc there)
    VSX_REQUIRE(d->M < (1L << 31) && d->N < (1L << 30) && (d->c_mode == 1 || d->M * d->ldc < (1L << 31)) &&
                    (!d->residual || d->M * d->ldr < (1L << 31)),
                VSX_E_UNSUPPORTED, "gemm: M*ldc and M*ldr must be below 2^31 elements");
    {
        const long brows = p.geglu ? 2 * d->N : d->N;
        const long bb = ((brows - 1) * d->ldb + d->K) * 2;
        VSX_REQUIRE(bb < (1L << 31), VSX_E_UNSUPPORTED, "gemm: B operand slice must be smaller than 2 GiB");
        p.b_bytes = (unsigned)bb;
    }
    VSX_REQUIRE(nbatch <= 65535, VSX_E_BADSHAPE, "gemm: batch0*batch1 = %ld exceeds 65535", nbatch);

    if (p.a_mode == 0) {
        VSX_REQUIRE(d->lda % 8 == 0, VSX_E_BADSHAPE, "gemm: lda (%ld) must be a multiple of 8", (long)d->lda);
        const long ab = ((d->M - 1) * d->lda + d->K) * 2;
        VSX_REQUIRE(ab < (1L << 31), VSX_E_UNSUPPORTED, "gemm: A operand slice must be smaller than 2 GiB");
        p.a_bytes = (unsigned)ab;
        p.a2_bytes = 0;
    } else if (p.a_mode == 1) {
        VSX_REQUIRE(nbatch == 1, VSX_E_UNSUPPORTED, "gemm: conv mode does not take a batch");
        VSX_REQUIRE(d->ks == 1 || d->ks == 3, VSX_E_UNSUPPORTED, "gemm: conv kernel size %ld", (long)d->ks);
        VSX_REQUIRE(d->stride == 1 || d->stride == 2, VSX_E_UNSUPPORTED, "gemm: conv stride %ld", (long)d->stride);
        VSX_REQUIRE(d->C1 > 0 && d->C1 % 8 == 0 && d->C2 >= 0 && d->C2 % 8 == 0, VSX_E_BADSHAPE,
                    "gemm: conv channels must be multiples of 8 (C1=%ld C2=%ld)", (long)d->C1, (long)d->C2);
        VSX_REQUIRE((d->C2 == 0) == (d->A2 == nullptr), VSX_E_BADSHAPE, "gemm: A2/C2 mismatch");
        VSX_REQUIRE(d->A2 == nullptr || vsx_aligned16(d->A2), VSX_E_BADSHAPE, "gemm: A2 must be 16-byte aligned");
        VSX_REQUIRE(d->K == d->ks * d->ks * (d->C1 + d->C2), VSX_E_BADSHAPE, "gemm: conv K mismatch");
        VSX_REQUIRE(d->C2 == 0 || (d->C1 % 64 == 0 && d->C2 % 64 == 0), VSX_E_UNSUPPORTED,
                    "gemm: a two-source conv needs both channel counts to be multiples of 64 (one K slab)");
        VSX_REQUIRE(d->H > 0 && d->W > 0, VSX_E_BADSHAPE, "gemm: conv H/W");
        VSX_REQUIRE(!d->upsample || (d->H % 2 == 0 && d->W % 2 == 0), VSX_E_BADSHAPE, "gemm: upsample needs even H/W");
        p.H = (int)d->H; p.W = (int)d->W; p.C1 = (int)d->C1; p.C2 = (int)d->C2;
        p.ks = (int)d->ks; p.stride = (int)d->stride; p.ups = d->upsample ? 1 : 0;
        const bool subpix = d->upsample == 2;
        if (subpix) {
            // sub-pixel form (vsx.h: upsample = 2): internally a plain 3x3 window on the SOURCE image, four taps per class
            VSX_REQUIRE(d->ks == 3 && d->stride == 1 && d->C2 == 0 && d->C1 % 64 == 0 && d->pad_lo < 0 && d->pad_hi < 0 &&
                            !d->rowvec && !d->residual && !d->geglu && d->c_mode == 0 && !d->rowscale && !d->rowstats,
                        VSX_E_UNSUPPORTED, "gemm: the sub-pixel form takes a single-source 3x3 stride-1 convolution with a bias only");
            VSX_REQUIRE(d->M % 4 == 0, VSX_E_BADSHAPE, "gemm: sub-pixel form: M must be a multiple of 4");
            p.ups = 0;
            p.H = (int)d->H / 2;
            p.W = (int)d->W / 2;
            p.sp_Mc = (int)(d->M / 4);
            p.K = 4 * d->C1;                 // slabs actually multiplied (the weight rows stay 9 C long: ldb)
        }
        const bool sym = d->pad_lo < 0 && d->pad_hi < 0;
        VSX_REQUIRE(sym || (d->pad_lo >= 0 && d->pad_hi >= 0 && d->pad_lo < d->ks && d->pad_hi < d->ks), VSX_E_BADSHAPE,
                    "gemm: conv padding (%ld, %ld) for kernel size %ld", (long)d->pad_lo, (long)d->pad_hi, (long)d->ks);
        const int pad_lo = sym ? p.ks / 2 : (int)d->pad_lo, pad_hi = sym ? p.ks / 2 : (int)d->pad_hi;
        p.pad = pad_lo;
        p.Ho = (p.H + pad_lo + pad_hi - p.ks) / p.stride + 1;
        p.Wo = (p.W + pad_lo + pad_hi - p.ks) / p.stride + 1;
        {
            const long pix = (d->M / ((long)p.Ho * p.Wo) / (subpix ? 4 : 1)) * (long)(p.ups ? p.H / 2 : p.H) * (p.ups ? p.W / 2 : p.W);
            VSX_REQUIRE(pix * d->C1 * 2 < (1L << 31) && pix * d->C2 * 2 < (1L << 31), VSX_E_UNSUPPORTED,
                        "gemm: conv source tensors must be smaller than 2 GiB");
            p.a_bytes = (unsigned)(pix * d->C1 * 2);
            p.a2_bytes = (unsigned)(pix * d->C2 * 2);
        }
        VSX_REQUIRE(d->M % ((long)p.Ho * p.Wo) == 0, VSX_E_BADSHAPE, "gemm: conv M (%ld) not a multiple of Ho*Wo (%d*%d)",
                    (long)d->M, p.Ho, p.Wo);
        if (subpix) {                        // B = four [N, 9 C] matrices, one per (ph, pw) class
            const long bb = ((4 * d->N - 1) * d->ldb + d->K) * 2;
            VSX_REQUIRE(bb < (1L << 31), VSX_E_UNSUPPORTED, "gemm: B operand slice must be smaller than 2 GiB");
            p.b_bytes = (unsigned)bb;
        }
    } else {
        return vsx_fail(VSX_E_UNSUPPORTED, "gemm: a_mode %d", p.a_mode);
    }
    if (p.c_mode == 1) {
        VSX_REQUIRE(!p.geglu && !d->rowvec && !d->residual, VSX_E_UNSUPPORTED, "gemm: transposed store takes bias only");
        VSX_REQUIRE(d->c_rows_per_img > 0, VSX_E_BADSHAPE, "gemm: c_rows_per_img");
        p.c_pack4 = (d->c_rows_per_img % 4 == 0 && d->ldc % 4 == 0 && d->c_img_stride % 4 == 0 &&
                     d->c_bs0 % 4 == 0 && d->c_bs1 % 4 == 0) ? 1 : 0;
    }
    if (p.geglu) VSX_REQUIRE(!d->rowvec, VSX_E_UNSUPPORTED, "gemm: geglu with rowvec");
    // 8-byte vector epilogue when every row start / column quad is 8-byte aligned
    auto al8 = [](const void* q) { return (((uintptr_t)q) & 7) == 0; };
    p.vec4 = (d->ldc % 4 == 0 && d->c_bs0 % 4 == 0 && d->c_bs1 % 4 == 0 && al8(d->C) && al8(d->bias) &&
              al8(d->rowvec) && d->N % 4 == 0 &&
              (!d->residual || (al8(d->residual) && d->ldr % 4 == 0 && d->r_bs0 % 4 == 0 && d->r_bs1 % 4 == 0)))
                 ? 1 : 0;
    // 16-byte stores (two lanes of a pair exchange register quads first) when rows and column octets are 16-byte aligned
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    p.vec8 = (p.vec4 && d->ldc % 8 == 0 && d->c_bs0 % 8 == 0 && d->c_bs1 % 8 == 0 && al16(d->C) && d->N % 8 == 0) ? 1 : 0;
    p.rvec8 = (p.vec8 && d->residual && al16(d->residual) && d->ldr % 8 == 0 && d->r_bs0 % 8 == 0 && d->r_bs1 % 8 == 0)
                  ? 1 : 0;

    // tile selection.  cols = rows of B.  The 320-wide tiles need cols % 320 == 0 (no column padding waste) and
    // enough workgroups to cover the 256 CUs; otherwise fall back to the 128/64 tiles (2 workgroups per CU).
    const long cols = p.geglu ? 2 * d->N : d->N;
    auto blocks = [&](long bm, long bn) { return ((d->M + bm - 1) / bm) * ((cols + bn - 1) / bn) * nbatch; };
    int rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool sample = !dry && g_prof.on && !g_prof.paused && g_prof.n < g_prof.max_samples &&
                        (g_prof.seen++ % g_prof.stride) == 0;
    if (sample) {
        if ((long)g_prof.ev->size() < 2 * (g_prof.n + 1)) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
                return vsx_fail(VSX_E_LAUNCH, "prof: hipEventCreate failed");
            g_prof.ev->push_back(a);
            g_prof.ev->push_back(b);
        }
        e0 = (*g_prof.ev)[2 * g_prof.n];
        e1 = (*g_prof.ev)[2 * g_prof.n + 1];
        (void)hipEventRecord(e0, stream);
    }
    const bool wide = (cols % 320 == 0);
    const int splits = plan_splitk(d, blocks(128, 320), wide && nbatch == 1 && !p.geglu && p.c_mode == 0 && p.vec4);
    // Persistent ping-pong kernel (gemm_pp.hip): problems with at least ~one 256x320 (or 128x320) tile per CU.
    // VSX_GEMM_PP=0 disables it (A/B measurements against the workgroup-per-tile kernels), 2 forces it whenever the
    // shape is eligible.
    const int pp = pp_mode();
    const bool pp_ok = pp != 0 && wide && nbatch == 1 && (splits <= 1 || pp >= 2) && !force_tile() && pp_supported(p);
    // option gemm_pp: 0 never, 1 automatic (thresholds from tools/gemm_ab.py: profiles/r03_gemm_ab_b{1,2}.txt), 2 = 256-row
    // tiles wherever >= 64 of them exist (else 128-row), 3 = 128-row tiles wherever >= 32 exist, 4 = 256-row tiles for every
    // eligible problem however small (tests: tools/cpu_check).
    // Automatic: 256-row tiles from 192 of them (0.75 per CU: qkv 1280->3840 at B = 1, 47 instead of 60 us); 128-row
    // tiles for the plain GEMMs with about ONE such tile per CU (the 640 / 1280-wide projections at M = 16 384 / 8 192 and
    // ff2 at those sizes: 7-10 % over the tile kernels; the convolutions of that size stay on the tile kernels)
    const long t128 = blocks(128, 320);
    const bool pp256 = pp_ok && pp != 3 && (blocks(256, 320) >= 192 || (pp == 2 && blocks(256, 320) >= 64) || pp == 4);
    const bool pp128 = !pp256 && pp_ok && p.c_mode != 1 && ((pp >= 2 && t128 >= 32) || (pp == 1 && p.a_mode != 1 && t128 >= 240 && t128 <= 272));
    // Weight-stationary kernel (gemm_pp.hip: gemm_ws320_kernel) for the byte-bound K = 320 projections of the 64 x 64 level; option
    // "gemm_ws" / VSX_GEMM_WS: 0 = never; 1 (default) = the K = N = 320 projections WITH a residual from 65 536 rows (8 blocks of 32 rows
    // per CU) — where it measured at or above the persistent kernel, alone and inside the loop (profiles/r06_gemm_weight_stationary_ab.txt:
    // + 0.5 % frames/s at one clip per step, + 0.6 % at four; without a residual the persistent kernel's epilogue is short enough
    // and wins); 3 = the same from 131 072 rows (A/B runs); 4 = 1 + the LayerNorm-folded projections 320 -> 640 / 960 (column slices);
    // 2 = every eligible problem (tests)
    const long ws_opt = gemm_option("gemm_ws");
    const bool ws_res = d->N == 320 && d->residual != nullptr && d->M >= (ws_opt == 3 ? 131072 : 65536);
    const bool ws_ln = ws_opt == 4 && d->N > 320 && d->rowscale != nullptr && d->M >= 65536;
    const bool ws_st = ws_opt == 5 && d->N == 320 && d->rowstats_parts >= 0 && d->rowscale == nullptr && d->residual == nullptr && d->M >= 131072;     // (5: + the residual-free K = N = 320 launches, A/B runs)
    const bool ws = ws_opt != 0 && pp != 0 && nbatch == 1 && splits <= 1 && !force_tile() && ws_supported(p) && (ws_opt == 2 || ws_res || ws_ln || ws_st);
    // row statistics of the output (vsx.h, ABI 8): only the staged row passes of the persistent kernels produce them
    const long stat_parts = ws ? (p.rowscale ? 0 : (cols / 320) * ws_waves()) : ((pp256 || pp128) && pp_rowstats_ok(p) ? (cols / 320) * 6 : 0);
    if (dry) {
        *parts_out = stat_parts;
        return VSX_OK;
    }
    if (d->rowstats != nullptr) {
        VSX_REQUIRE(stat_parts > 0 && d->rowstats_parts == stat_parts, VSX_E_UNSUPPORTED,
                    "gemm: rowstats with %ld parts, but this launch writes %ld (ask vsx_gemm_rowstats_parts first)",
                    (long)d->rowstats_parts, stat_parts);
        VSX_REQUIRE(vsx_aligned16(d->rowstats), VSX_E_BADSHAPE, "gemm: rowstats must be 16-byte aligned");
        p.rowstats = (float*)d->rowstats;
        p.rowstats_parts = (int)stat_parts;
    }
    if (p.sp_Mc > 0)
        VSX_REQUIRE((pp256 && p.sp_Mc % 256 == 0) || (pp128 && p.sp_Mc % 128 == 0), VSX_E_UNSUPPORTED,
                    "gemm: the sub-pixel form runs on the persistent kernel only (enough tiles, rows per class a multiple of the tile)");
    if (ws) {
        rc = launch_ws(p, stream);
    } else if (pp256) {
        rc = launch_pp(p, 256, stream);
    } else if (pp128) {
        rc = launch_pp(p, 128, stream);
    } else if (force_tile() && (wide || force_tile() >= 4)) {
        p.ws = (float*)d->workspace;
        if (force_tile() == 1) rc = launch_tile<128, 320, 4, 2>(p, d->M, cols, nbatch, stream);
        else if (force_tile() == 2) rc = launch_tile<128, 160, 4, 1>(p, d->M, cols, nbatch, stream, tune_deep());
        else if (force_tile() == 3) rc = launch_tile<256, 320, 8, 2>(p, d->M, cols, nbatch, stream);
        else if (force_tile() == 4) rc = launch_tile<128, 128, 2, 2>(p, d->M, cols, nbatch, stream);
        else if (force_tile() == 5) rc = launch_tile<64, 128, 2, 2>(p, d->M, cols, nbatch, stream);
        else rc = launch_tile<64, 64, 2, 2>(p, d->M, cols, nbatch, stream);
    } else if (splits > 1 && d->workspace != nullptr &&
        d->workspace_bytes >= (int64_t)splits * d->M * d->N * (int64_t)sizeof(float)) {
        const long nk = (d->K + BK - 1) / BK;
        p.splitk = splits;
        p.nk_per = (int)((nk + splits - 1) / splits);
        p.ws = (float*)d->workspace;
        rc = launch_tile<128, 320, 4, 2>(p, d->M, cols, nbatch, stream);
        if (rc == VSX_OK) {
            const long quads = d->M * (d->N / 4);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, p);
            rc = vsx_check_launch("vsx_gemm_f16 (split-K reduce)");
        }
    } else if (wide && blocks(256, 320) >= 240 && (p.a_mode == 1 || p.geglu || p.residual == nullptr)) {
        // 16 waves (4 per SIMD), 32x160 per wave, 142 FLOP per staged byte: the big-M convolutions, the GEGLU
        // projections (their erf epilogue overlaps with the other waves' MFMAs) and the residual-free projections.
        // Measured per shape against the 8- and 4-wave tiles with tools/shape_prof.py (VSX_TUNE_TILE=1|2|3).
        rc = launch_tile<256, 320, 8, 2>(p, d->M, cols, nbatch, stream);
    } else if (cols % 160 == 0 && blocks(128, 160) >= 200) {
        // 4 waves, 32x160 per wave, two workgroups per CU (74 KiB of LDS each): one's epilogue and prologue overlap
        // with the other's main loop, and its registers leave room to prefetch the residual.  Since the 128-byte-row
        // slabs it is at least as fast as the 8-wave 128x320 tile on every UNet shape.
        // at most one workgroup per CU (<= 256 of them): nothing to overlap with, so four ring slots instead of two
        // (profiles/r04_gemm_small_m_sweep.txt: proj 1280->1280 at M = 4096 35.2 -> 32.9 us, qk 1280->2560 at M = 2048 - 7 %)
        rc = launch_tile<128, 160, 4, 1>(p, d->M, cols, nbatch, stream, blocks(128, 160) <= 256);
    } else if (cols % 160 == 0 && blocks(128, 160) >= 176 && !p.geglu && p.c_mode == 0) {
        rc = launch_tile<128, 160, 4, 1>(p, d->M, cols, nbatch, stream, true);      // qkv 1280->3840 at M = 1024: - 9 %
    } else if (wide && blocks(128, 320) >= 200) {
        rc = launch_tile<128, 320, 4, 2>(p, d->M, cols, nbatch, stream);    // 8 waves, 32x160 per wave
    } else if (blocks(128, 128) >= 512) {
        rc = launch_tile<128, 128, 2, 2>(p, d->M, cols, nbatch, stream);
    } else if (blocks(64, 128) >= 512 || (p.geglu && cols >= 128)) {
        rc = launch_tile<64, 128, 2, 2>(p, d->M, cols, nbatch, stream);
    } else {
        rc = launch_tile<64, 64, 2, 2>(p, d->M, cols, nbatch, stream);
    }
    if (sample) {
        (void)hipEventRecord(e1, stream);
        g_prof.n += 1;
        const double flop = 2.0 * (double)d->M * (double)cols * (double)p.K * (double)nbatch;      // (sub-pixel form: the 4 C it multiplies)
        g_prof.flop += flop;
        // algorithmic bytes (tools/pmc_by_shape.py has the same definition): every operand element once
        double a_el = (double)d->M * (double)d->K;
        if (p.a_mode == 1) {
            const double nimg = (double)d->M / ((double)p.Ho * (double)p.Wo) / (p.sp_Mc > 0 ? 4.0 : 1.0);
            const double hin = p.ups ? p.H / 2 : p.H, win = p.ups ? p.W / 2 : p.W;     // (sub-pixel form: p.H, p.W are the source's already)
            a_el = nimg * hin * win * (double)(p.C1 + p.C2);
        }
        const double w_el = (double)cols * (double)d->K * (p.sp_Mc > 0 ? 4.0 : 1.0);
        const double c_el = (double)d->M * (double)d->N;
        g_prof.work->push_back(flop);
        g_prof.work->push_back(2.0 * (double)nbatch * (a_el + w_el + c_el + (d->residual ? c_el : 0.0)));
    }
    return rc;
}
