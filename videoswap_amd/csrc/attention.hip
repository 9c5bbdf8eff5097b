// K5/K6 — fused attention for gfx950: O = softmax(Q K^T * scale) V with online softmax on
// v_mfma_f32_32x32x16_f16, scores never leave the register file.
//
// Work split: workgroup = 4 wave64 = 128 query rows of one (image, head); each wave owns 32
// query rows and walks the keys in tiles of 64.  Both products are computed TRANSPOSED so that
// every per-query quantity (running max m, running sum l, rescale factor) is lane-local:
//
//   S^T[key, q] = K[key, :] . Q[q, :]      A = K tile (LDS), B = Q fragments (registers)
//   O^T[c,   q] = V^T[c, key] . P^T[key,q] A = V^T tile (LDS), B = P (registers, straight from S^T)
//
// In the 32x32 MFMA C/D layout a lane holds column q = lane&31 and 16 of the 32 rows, so the
// lane pair (q, q+32) holds a full score column: row max / row sum are 15 local ops plus one
// cross-lane exchange, and the exponentiated scores are already in B-operand position for the
// second product (the k-slot -> key assignment of P is matched when V^T fragments are read).
// V arrives pre-transposed per image (vsx_gemm_f16 c_mode 1 writes V^T while projecting), which
// keeps every LDS fragment read a contiguous 8/16-byte access.
//
// LDS: K tile [64][DK*16+8] halfs (row = 2*DK+1 16-byte slots, odd => conflict-free b128 reads); LDS row r holds key
//      r with bits 2,3 swapped, which makes a lane's 8 scores per MFMA step 8 consecutive keys;
//      V^T tile [DT*32][72] halfs (144-byte rows = 9 slots, odd => conflict-free b128 reads).
#include "common.h"

#include <type_traits>

namespace vsxg {
long gemm_option(const char* name);      // gemm.hip: the option table of vsx_set_option
}

namespace {

struct AttnParams {
    const half_t* Q;
    const half_t* K;
    const half_t* VT;
    half_t* O;
    int nq, nk, heads;
    long ldq, ldk, ldvt, ldo;
    long q_bs, k_bs, vt_bs, o_bs;
    int kv_div;
    float scale_log2e;
    float* lse;     // optional [nb, heads, lse_ld]: log2 of the softmax denominator in the scaled-score domain (training)
    long lse_ld;
};

// -DVSX_GEMM_TIMING (tools/gemm_timing.py): per-wave cycle totals of the key-loop segments, long[block][wave][6]
#ifdef VSX_GEMM_TIMING
__device__ long* g_attn_dbg = nullptr;
#define ASTAMP(i) do { const long t_now = (long)clock64(); t_seg[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif

constexpr int KV_TILE = 64;
constexpr int VSTR = 72;          // V^T LDS row: 64 keys + one 16-byte dummy slot (odd slot count)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// QB: 32-query column blocks per wave.  With QB = 2 a wave owns 64 queries: every K and V^T fragment read from LDS feeds two
// MFMAs and a workgroup covers 256 queries per key tile, so both the bytes delivered into LDS (16 KB per tile) and the
// LDS fragment reads per FLOP halve; 128 accumulator + 64 score registers (188 VGPRs at d = 40, 249 at d = 80): two waves
// per SIMD instead of four.  Measured in round 4 (profiles/r04_attn_ab*.txt; N = 4096, d = 40, 32 images: 1 055 us):
//   * removing the if-converted mask of the partial tile (100 of 200 VALU instructions per tile): 0 %;
//   * QB = 2: -2 ... -5 % on one box (1 038 us; with the scale-and-subtract of two scores as one v_pk_fma_f32 1 008 us),
//     +3 % on another; d = 80 (N = 1024): -4 ... -5 % on both -> the launch rule takes QB = 2 for d = 80 only;
//   * P rounded toward zero (v_cvt_pkrtz, same issue cost as v_cvt_pk: tools/ubench/valu_rate.hip): no gain, error 2.8e-4
//     -> 3.1e-4: dropped.
// Neither operand delivery nor the VALU instruction count is what bounds this kernel at d = 40: the time is close to
// (MFMA issue + VALU issue) of a SIMD's waves added up, 29 % of the MFMA work being the 40 -> 48 / 40 -> 64 padding.
// O16 (round 6, d = 40 only; option attn_o16, off by default — see launch_attn): the second product on 16 x 16 x 32 MFMA tiles.  With 32 x 32 tiles O^T pads 40 channel rows to 64
// (8 MFMAs of 32 cycles per 64-key tile, 37.5 % of them multiplying padding); with 16-row tiles it pads to 48: 2 key blocks x 3
// row tiles x 2 query tiles = 12 MFMAs of 16 cycles (tools/ubench/mfma16_probe.hip: 17 against 32 ticks per instruction), 192
// instead of 256 matrix-pipe cycles per key tile of a kernel whose time is MFMA issue + VALU issue added up.  What it takes:
//   * P^T as the B operand of v_mfma_f32_16x16x32_f16: lane l supplies query column l % 16 and the 8 keys of k-group l / 16.  From
//     the 32 x 32 C/D layout of S^T (lane = (query q = l & 31, hi = l >> 5); per 32-key block two 8-key chunks s2 = 0 / 1 at keys
//     16 s2 + 8 hi + 0..7) the four 16-lane rows hold (q 0-15, hi 0), (q 16-31, hi 0), (q 0-15, hi 1), (q 16-31, hi 1):
//     v_permlane16_swap(X = chunk s2 = 0, Y = chunk s2 = 1) trades X's odd rows for Y's even rows (probed: X' = {X.row0, Y.row0,
//     X.row2, Y.row2}, Y' = {X.row1, Y.row1, X.row3, Y.row3}), so X' is the whole B operand of queries 0-15 and Y' of queries
//     16-31, both with k-group g holding keys {0, 16, 8, 24}[g] + 0..7 of the block: four swaps per 32 keys, no LDS round trip;
//   * the V^T fragment (A operand: lane l supplies channel row l % 16 of the 16-row tile and the same 8 keys): one ds_read_b128 at
//     key offset {0, 16, 8, 24}[l / 16];
//   * O^T in the 16 x 16 C/D layout: lane l holds query 16 qt + l % 16 of BOTH query tiles and channels 16 ct + 4 (l / 16) + 0..3,
//     so the rescale factor of the online softmax (per query, living in the S^T layout's lanes) is redistributed with one more
//     permlane16_swap(alpha, alpha) -> (alpha of tile 0's queries, alpha of tile 1's), only when a row maximum grew;
//   * the all-ones row that makes the MFMA produce the softmax denominator sits in row 47 (the last of the 48).
template <int D, int QB = 1, bool O16 = false>
__global__ __launch_bounds__(256) void flash_attn_kernel(const AttnParams p) {
    static_assert(!O16 || (QB == 1 && D % 16 != 0 && D <= 48), "16-row O tiles: one query block per wave, a spare padding row");
    constexpr int DK = (D + 15) / 16;   // k-steps of the QK^T product
    constexpr int DT = (D + 31) / 32;   // 32-row tiles of O^T
    constexpr int CT = (D + 15) / 16;   // 16-row tiles of O^T (O16)
    constexpr int VROWS = O16 ? CT * 16 : DT * 32;      // rows of the V^T tile in LDS
    constexpr int KSTR = DK * 16 + 8;   // K LDS row (halfs): 2*DK real/zero slots + one dummy slot (odd slot count)
    constexpr bool HAS_SPARE = (VROWS > D);     // a padding row of the O^T tile can carry the softmax denominator
    constexpr int KSPR = KSTR / 8, VSPR = VSTR / 8;              // 16-byte slots per LDS row
    constexpr int K_UNITS = KV_TILE * KSPR, V_UNITS = D * VSPR;  // 16-byte units per tile (V: rows < D only)
    constexpr int NKI = (K_UNITS + 255) / 256, NVI = (V_UNITS + 255) / 256;   // LDS-DMA instructions per wave
    constexpr int K_BYTES = KV_TILE * KSTR * 2;
    constexpr int V_BYTES = VROWS * VSTR * 2;
    constexpr int STAGE = K_BYTES + V_BYTES;
    constexpr int OOB_OFF = (int)0x80000000;

    // two-slot LDS ring filled by LDS-DMA (`buffer_load_dwordx4 ... lds`): tile j+1 streams in while tile j is
    // consumed; per-lane offsets are loop-invariant, the key position advances in the scalar offset, out-of-range
    // rows / pad slots read as zero through the buffer descriptor
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order (1-D grid): workgroup id b lands on XCD b % 8; give every XCD a contiguous range of
    // (image, head, query-tile) so the query tiles that share one head's K / V^T stream them from the same L2
    int wg = blockIdx.x;
    {
        const int nwg = gridDim.x, qq = nwg >> 3, rr = nwg & 7;
        const int xcd = wg & 7, local = wg >> 3;
        wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + local;
    }
    constexpr int QTILE = 128 * QB;      // queries per workgroup
    const int qtiles = (p.nq + QTILE - 1) / QTILE;
    const int qt = wg % qtiles;
    const int h = (wg / qtiles) % p.heads;
    const long b = wg / (qtiles * p.heads);
    const long kvb = b / p.kv_div;

    const int q0 = qt * QTILE + wave * (32 * QB) + l31;      // query of column block 0; block x: q0 + 32 x

    // Q fragments: B operand of S^T = K Q^T; lane (q, hi) holds Q[q, t*16 + hi*8 .. +7]
    h8 qf[QB][DK];
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        const int q = q0 + 32 * x;
        const bool qok = q < p.nq;
        const half_t* qrow = p.Q + b * p.q_bs + (long)(qok ? q : 0) * p.ldq + h * D;
#pragma unroll
        for (int t = 0; t < DK; ++t) {
            const int d0 = t * 16 + hi * 8;
            qf[x][t] = (qok && d0 < D) ? as_h8(ld16(qrow + d0)) : as_h8(make_uint4(0, 0, 0, 0));
        }
    }

    const half_t* Kb = p.K + kvb * p.k_bs + h * D;
    const half_t* Vb = p.VT + kvb * p.vt_bs + (long)h * D * p.ldvt;
    const __amdgpu_buffer_rsrc_t rsrcK = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(Kb), 0, (int)((((long)p.nk - 1) * p.ldk + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcV = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(Vb), 0, (int)((long)D * p.ldvt * 2), 0x00020000);
    int vk[NKI], vv[NVI];
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
        const int u = (wave * NKI + i) * 64 + lane;
        const int row = u / KSPR, slot = u - row * KSPR;
        // LDS row r of the K tile holds key krow(r) = r with bits 2 and 3 swapped: in the 32x32 C/D layout a lane then
        // owns 8 CONSECUTIVE keys per (tile, half) instead of two groups of 4, so the matching V^T fragment of the
        // second product is one ds_read_b128
        const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
        vk[i] = (u < K_UNITS && slot * 8 < D) ? (int)(((long)key * p.ldk + slot * 8) * 2) : OOB_OFF;
    }
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
        const int u = (wave * NVI + i) * 64 + lane;
        const int row = u / VSPR, slot = u - row * VSPR;
        vv[i] = (u < V_UNITS && slot < 8) ? (int)(((long)row * p.ldvt + slot * 8) * 2) : OOB_OFF;
    }
    // rows >= D of the V^T tiles are never written by the DMA: zero them once, and put the ones row in place
    for (int i = tid; i < 2 * (VROWS - D) * VSTR; i += 256) {
        const int st = i / ((VROWS - D) * VSTR), r = i - st * ((VROWS - D) * VSTR);
        const int row = D + r / VSTR, col = r - (r / VSTR) * VSTR;
        half_t* sv = reinterpret_cast<half_t*>(smem + st * STAGE + K_BYTES);
        // the ones row is the LAST row of the tile: the tail lanes of the final V^T DMA instruction write zeros into
        // the (up to 7) rows right after row D-1, so row D itself is not safe
        sv[row * VSTR + col] = (HAS_SPARE && row == VROWS - 1 && col < KV_TILE) ? (half_t)1.f : (half_t)0.f;
    }

    auto issue = [&](int j0, int stage) {
        unsigned char* sb = smem + stage * STAGE;
        const int soffK = (int)((long)j0 * p.ldk * 2);
        const int soffV = j0 * 2;
        // the descriptor's range check does not cover the scalar offset: in the last, partial tile the rows / key
        // slots past the end are switched off per lane (once per workgroup)
        const bool partial = j0 + KV_TILE > p.nk;
#pragma unroll
        for (int i = 0; i < NKI; ++i)
            if ((wave * NKI + i) * 64 < K_UNITS) {     // wave-uniform
                int v = vk[i];
                if (partial) {
                    const int row = ((wave * NKI + i) * 64 + lane) / KSPR;
                    if (j0 + ((row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1)) >= p.nk) v = OOB_OFF;
                }
                // lanes past the end of the tile are switched off (EXEC): a DMA lane always writes its 16 bytes,
                // zeros included, and would clobber the neighbouring LDS region
                if ((wave * NKI + i) * 64 + lane < K_UNITS)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, (lds_ptr_t)(sb + (wave * NKI + i) * 1024), 16, v,
                                                             soffK, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < NVI; ++i)
            if ((wave * NVI + i) * 64 < V_UNITS) {
                int v = vv[i];
                if (partial) {
                    const int u = (wave * NVI + i) * 64 + lane;
                    if (j0 + (u - (u / VSPR) * VSPR) * 8 >= p.ldvt) v = OOB_OFF;
                }
                if ((wave * NVI + i) * 64 + lane < V_UNITS)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcV, (lds_ptr_t)(sb + K_BYTES + (wave * NVI + i) * 1024),
                                                             16, v, soffV, 0, 0);
            }
    };

    f16v o[QB][O16 ? 1 : DT];
    f4v o16[O16 ? CT : 1][2];             // O16: [16-row channel tile][16-query tile], 4 registers each
    float m_i[QB], l_i[QB];
#pragma unroll
    for (int x = 0; x < QB; ++x) {
#pragma unroll
        for (int t = 0; t < (O16 ? 1 : DT); ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[x][t][r] = 0.f;
        m_i[x] = -INFINITY;
        l_i[x] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < (O16 ? CT : 1); ++t)
#pragma unroll
        for (int x = 0; x < 2; ++x) o16[t][x] = f4v{0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    int stage = 0;
#ifdef VSX_GEMM_TIMING
    long t_seg[6] = {0, 0, 0, 0, 0, 0}, t_last = (long)clock64();
#endif
    // One key tile.  MASKED (a compile-time constant) is the last, partial tile of a key count that is not a multiple of
    // 64: with the test written as a run-time `if (j0 + KV_TILE > nk)` inside one loop body the compiler if-converted the
    // mask — 31 compares, 33 selects and 37 key-index additions per tile in EVERY iteration, as many VALU instructions
    // again as the softmax itself (32 fma, 32 exp, 16 cvt_pk, 20 max: ISA of round 3) — so the loop runs the unmasked
    // body and the partial tile, if any, is peeled.
    auto tile = [&](const int j0, auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ASTAMP(0);
        __syncthreads();   // tile j0 is in LDS for every wave; the other slot is no longer being read
        ASTAMP(1);
        if (j0 + KV_TILE < p.nk) issue(j0 + KV_TILE, stage ^ 1);
        ASTAMP(2);
        const half_t* sK = reinterpret_cast<const half_t*>(smem + stage * STAGE);
        const half_t* sV = reinterpret_cast<const half_t*>(smem + stage * STAGE + K_BYTES);

        // ---- S^T = K Q^T : two 32-key row tiles; a K fragment feeds the QB query blocks ----
        f16v s[QB][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const half_t* krow = sK + (kt * 32 + l31) * KSTR + hi * 8;
            const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < DK; ++t) {
                const h8 kf = *reinterpret_cast<const h8*>(krow + t * 16);
#pragma unroll
                for (int x = 0; x < QB; ++x)
                    s[x][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[x][t], t == 0 ? zero : s[x][kt], 0, 0, 0);
            }
        }
        ASTAMP(3);
        // ---- online softmax (lane-local per query column); VALU budget: max3, fma, exp2, cvt per score ----
#pragma unroll
        for (int x = 0; x < QB; ++x) {
            if constexpr (MASKED) {      // only the last, partial key tile needs masking
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // C/D row (r&3) + 8*(r>>2) + 4*hi of the tile holds key (r&3) + 4*((r>>2)&1) + 8*hi + 16*(r>>3)
                        const int key = j0 + kt * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);
                        if (key >= p.nk) s[x][kt][r] = -INFINITY;
                    }
            }
            float mx = fmaxf(s[x][0][0], s[x][1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[x][0][r]), s[x][1][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * p.scale_log2e;      // scale > 0: max commutes with the scaling
            if (!__all(mx <= m_i[x])) {     // running max grows: rescale O (and the denominator row inside it)
                const float m_new = fmaxf(m_i[x], mx);
                const float alpha = __builtin_amdgcn_exp2f(m_i[x] - m_new);
                m_i[x] = m_new;
                if (!HAS_SPARE) l_i[x] *= alpha;
                if constexpr (O16) {
                    // alpha belongs to query l & 31; the O^T tiles of this lane hold queries l % 16 (tile 0) and 16 + l % 16 (tile 1)
                    const auto ab = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, alpha), __builtin_bit_cast(unsigned, alpha), false, false);
                    // (scalars first: __builtin_bit_cast applied to a vector ELEMENT reads the vector's first element whatever the index)
                    const unsigned u0 = ab[0], u1 = ab[1];
                    const float a0 = __builtin_bit_cast(float, u0), a1 = __builtin_bit_cast(float, u1);
#pragma unroll
                    for (int t = 0; t < CT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { o16[t][0][r] *= a0; o16[t][1][r] *= a1; }
                } else {
#pragma unroll
                    for (int t = 0; t < DT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[x][t][r] *= alpha;
                }
            }
            const float neg_m = -m_i[x];
            float rs = 0.f;
            {   // two scores per v_pk_fma_f32
                const vsx_f2 sc2 = {p.scale_log2e, p.scale_log2e}, nm2 = {neg_m, neg_m};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const vsx_f2 t = vsx_f2{s[x][kt][r], s[x][kt][r + 1]} * sc2 + nm2;
                        const float e0 = __builtin_amdgcn_exp2f(t[0]), e1 = __builtin_amdgcn_exp2f(t[1]);
                        s[x][kt][r] = e0;
                        s[x][kt][r + 1] = e1;
                        if (!HAS_SPARE) rs += e0 + e1;
                    }
            }
            if (!HAS_SPARE) {
                rs += __shfl_xor(rs, 32, 64);
                l_i[x] += rs;
            }
        }

        ASTAMP(4);
        // ---- O^T += V^T P^T; a V^T fragment feeds the QB query blocks ----
        // B operand k-slot jj of (kt, s2) on lane hi carries key kt*32 + 16*s2 + 8*hi + jj (K rows are permuted)
        if constexpr (O16) {
            const int l15 = lane & 15, g = lane >> 4;
            const int koff = ((g & 1) << 4) | ((g >> 1) << 3);          // {0, 16, 8, 24}[g]: the key chunk of k-group g after the swaps
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
                h8 px, py;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) { px[jj] = (half_t)s[0][kt][jj]; py[jj] = (half_t)s[0][kt][8 + jj]; }
                u4v ux = __builtin_bit_cast(u4v, px), uy = __builtin_bit_cast(u4v, py);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(ux[w], uy[w], false, false);
                    ux[w] = sw[0];
                    uy[w] = sw[1];
                }
                const h8 pb0 = __builtin_bit_cast(h8, ux), pb1 = __builtin_bit_cast(h8, uy);      // B operands of queries 0-15 / 16-31
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    const h8 vf = *reinterpret_cast<const h8*>(sV + (t * 16 + l15) * VSTR + kt * 32 + koff);
                    o16[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb0, o16[t][0], 0, 0, 0);
                    o16[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb1, o16[t][1], 0, 0, 0);
                }
            }
        } else
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                h8 pf[QB];
#pragma unroll
                for (int x = 0; x < QB; ++x) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) pf[x][jj] = (half_t)s[x][kt][8 * s2 + jj];
                }
                const int c0 = kt * 32 + 16 * s2 + 8 * hi;
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const h8 vf = *reinterpret_cast<const h8*>(sV + (t * 32 + l31) * VSTR + c0);
#pragma unroll
                    for (int x = 0; x < QB; ++x) o[x][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[x], o[x][t], 0, 0, 0);
                }
            }
        }
        ASTAMP(5);
        stage ^= 1;
    };
    {
        int j0 = 0;
        for (; j0 + KV_TILE <= p.nk; j0 += KV_TILE) tile(j0, std::false_type{});
        if (j0 < p.nk) tile(j0, std::true_type{});
    }

#ifdef VSX_GEMM_TIMING
    if (lane == 0 && g_attn_dbg) {
        long* o_dbg = g_attn_dbg + ((long)blockIdx.x * 4 + wave) * 6;
        for (int k = 0; k < 6; ++k) o_dbg[k] = t_seg[k];
    }
#endif
    if constexpr (O16) {
        // the denominator sits in row 47 = row 15 of the last 16-row tile: register 3 of the lanes 48 .. 63 (k-group 3), for query
        // 16 qt + l % 16; lane l stores channels 16 t + 4 (l / 16) + 0..3 of that query
        const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float den = __shfl(o16[CT - 1][qt][3], 48 + l15, 64);
            const int q = (q0 - l31) + 16 * qt + l15;       // (q0 - l31: the wave's first query)
            if (q < p.nq) {
                const float inv = 1.0f / den;
                half_t* orow = p.O + b * p.o_bs + (long)q * p.ldo + h * D;
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    const int c = t * 16 + 4 * g;
                    if (c < D) {
                        h4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = (half_t)(o16[t][qt][e] * inv);
                        *reinterpret_cast<h4*>(orow + c) = pk;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        if (HAS_SPARE) {
            // the denominator sits in the last row (31) of the last O^T tile: register 15 of the lanes with hi == 1;
            // broadcast it to the lane pair
            constexpr int reg = 15;
            constexpr int owner_hi = 1;
            const float mine = o[x][DT - 1][reg];
            const float other = __shfl_xor(mine, 32, 64);
            l_i[x] = (hi == owner_hi) ? mine : other;
        }
        // ---- normalise and store: lane (q, hi) holds O[q, t*32 + 8*g + 4*hi + 0..3] ----
        const int q = q0 + 32 * x;
        if (q < p.nq) {
            // what the backward pass recomputes P from: P = exp2(s * scale * log2(e) - lse)
            if (p.lse && hi == 0) p.lse[(b * p.heads + h) * p.lse_ld + q] = m_i[x] + __log2f(l_i[x]);
            const float inv = 1.0f / l_i[x];
            half_t* orow = p.O + b * p.o_bs + (long)q * p.ldo + h * D;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = t * 32 + 8 * g + 4 * hi;
                    if (c < D) {
                        h4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = (half_t)(o[x][t][4 * g + e] * inv);
                        *reinterpret_cast<h4*>(orow + c) = pk;
                    }
                }
        }
    }
}

template <int D>
int launch_attn(const AttnParams& p, long nb, hipStream_t stream) {
    if constexpr (D == 40 || D == 80) {
        // 64 queries per wave (option "attn_qb" / VSX_ATTN_QB: 0 = the rule below, 1 / 2 = always that many query blocks
        // per wave: A/B runs and tests).  Three boxes, interleaved rounds (profiles/r04_attn_ab*.txt): d = 80 gains 4 - 5 %
        // every time; d = 40 moves between - 3 % and + 5 % with the box and the image count (four waves per SIMD hide
        // more than two), so it keeps 32 queries per wave.
        const long wg2 = ((p.nq + 255) / 256) * (long)p.heads * nb;
        const long force = vsxg::gemm_option("attn_qb");
        if (force == 2 || (force == 0 && D == 80 && wg2 >= 512 && p.nq >= 256)) {
            hipLaunchKernelGGL((flash_attn_kernel<D, 2>), dim3((unsigned)wg2), dim3(256), 0, stream, p);
            return vsx_check_launch("vsx_attention_f16");
        }
    }
    dim3 grid((unsigned)(((p.nq + 127) / 128) * (long)p.heads * nb));
    if constexpr (D == 40) {
        // 16-row O^T tiles (flash_attn_kernel, O16): option "attn_o16" / VSX_ATTN_O16 = 1.  Built and measured in round 6
        // and NOT the default: 64 of 448 matrix-pipe cycles per key tile fewer, but the kernel does not get faster for it — 613.8 -> 602.3 TF/s
        // at 32 x 8 heads x 4096^2, 629.7 -> 650.6 at 16 images, 642.5 -> 648.1 at 5376 keys, 4.040 -> 4.039 frames/s end to end on one box
        // (profiles/r06_attn_o16_ab.txt): per 64-key tile a wave issues 164 VALU instructions — 33 v_exp_f32 among them, at a quarter of the
        // plain rate — beside its 14 MFMAs, and it is that stream, not the padding of O^T, that sets the tile time at d = 40.
        if (p.lse == nullptr && vsxg::gemm_option("attn_o16") != 0) {
            hipLaunchKernelGGL((flash_attn_kernel<D, 1, true>), grid, dim3(256), 0, stream, p);
            return vsx_check_launch("vsx_attention_f16");
        }
    }
    hipLaunchKernelGGL((flash_attn_kernel<D, 1>), grid, dim3(256), 0, stream, p);
    return vsx_check_launch("vsx_attention_f16");
}

// ---------------------------------------------------------------------------------------------
// K8 — temporal self-attention across frames at each spatial site.  One wave per
// (batch, site, head): Q/K/V rows of the site ([f, d], d contiguous in HBM) are staged in LDS,
// scores and softmax in fp32.  FLOPs are negligible (0.1 % of the UNet); the kernel is bound by
// the strided HBM reads, which adjacent heads of a site turn into full 128-byte lines in L2.
// ---------------------------------------------------------------------------------------------
struct TempParams {
    const half_t* Q;
    const half_t* K;
    const half_t* V;
    half_t* O;
    int fq, fk, hw, heads, d;
    long ldq, ldkv, ldo;
    float scale;
    int direct_out;     // option "temporal_out" = 1: round 4's store from the accumulator layout (A/B runs); default 0: staged through LDS
};

__global__ __launch_bounds__(256) void temporal_attn_kernel(const TempParams p) {
    // 4 waves per workgroup, one (site, head) problem per wave (heads are split over blockIdx.y * 4 + wave); each wave
    // has its own LDS slice, the workgroup barriers only keep the four waves in step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int d = p.d, fq = p.fq, fk = p.fk;
    const int wave = threadIdx.x >> 6;
    const size_t slice = (((size_t)(fq + 2 * fk) * d * sizeof(half_t) + (size_t)fq * fk * sizeof(float)) + 15) & ~(size_t)15;
    unsigned char* base = smem_raw + wave * slice;
    half_t* sQ = reinterpret_cast<half_t*>(base);        // [fq][d]
    half_t* sK = sQ + fq * d;                            // [fk][d]
    half_t* sV = sK + fk * d;                            // [fk][d]
    float* sS = reinterpret_cast<float*>(sV + fk * d);   // [fq][fk]

    const int lane = threadIdx.x & 63;
    const long site = blockIdx.x;
    const int h = min((int)blockIdx.y * 4 + wave, p.heads - 1);   // surplus waves redo the last head (same values)
    const long b = blockIdx.z;
    const int dv = d >> 3;

    // Stage Q, K, V: all of a batch's global loads are issued before the first LDS store (the loop used to wait for
    // every 16-byte load before issuing the next one: one or two requests in flight per wave, 2 TB/s)
    constexpr int UB = 4;
    for (int i0 = lane; i0 < fq * dv; i0 += 64 * UB) {
        uint4 r[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = i0 + 64 * u;
            if (i < fq * dv) {
                const int f = i / dv, c = (i - f * dv) * 8;
                r[u] = ld16(p.Q + ((b * fq + f) * p.hw + site) * p.ldq + h * d + c);
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = i0 + 64 * u;
            if (i < fq * dv) {
                const int f = i / dv, c = (i - f * dv) * 8;
                st16(sQ + f * d + c, r[u]);
            }
        }
    }
    for (int i0 = lane; i0 < fk * dv; i0 += 64 * UB) {
        uint4 rk[UB], rv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = i0 + 64 * u;
            if (i < fk * dv) {
                const int f = i / dv, c = (i - f * dv) * 8;
                const long row = (b * fk + f) * p.hw + site;
                rk[u] = ld16(p.K + row * p.ldkv + h * d + c);
                rv[u] = ld16(p.V + row * p.ldkv + h * d + c);
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = i0 + 64 * u;
            if (i < fk * dv) {
                const int f = i / dv, c = (i - f * dv) * 8;
                st16(sK + f * d + c, rk[u]);
                st16(sV + f * d + c, rv[u]);
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < fq * fk; i += 64) {
        const int f = i / fk, g = i - f * fk;
        float acc = 0.f;
        for (int c = 0; c < d; c += 8) {
            const h8 a = *reinterpret_cast<const h8*>(sQ + f * d + c);
            const h8 k = *reinterpret_cast<const h8*>(sK + g * d + c);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {     // v_dot2_f32_f16: two fp16 products accumulated in fp32
                const h2 a2 = {a[e], a[e + 1]}, k2 = {k[e], k[e + 1]};
                acc = __builtin_amdgcn_fdot2(a2, k2, acc, false);
            }
        }
        sS[i] = acc * p.scale;
    }
    __syncthreads();
    for (int f = lane; f < fq; f += 64) {
        float mx = -INFINITY;
        for (int g = 0; g < fk; ++g) mx = fmaxf(mx, sS[f * fk + g]);
        float sum = 0.f;
        for (int g = 0; g < fk; ++g) {
            const float e = __expf(sS[f * fk + g] - mx);
            sS[f * fk + g] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        for (int g = 0; g < fk; ++g) sS[f * fk + g] *= inv;
    }
    __syncthreads();
    for (int i = lane; i < fq * dv; i += 64) {
        const int f = i / dv, c = (i - f * dv) * 8;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int g = 0; g < fk; ++g) {
            const float pr = sS[f * fk + g];
            const h8 v = *reinterpret_cast<const h8*>(sV + g * d + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += pr * (float)v[e];
        }
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
        st16(p.O + ((b * fq + f) * p.hw + site) * p.ldo + h * d + c, as_u4(o));
    }
}


// V^T tile in LDS: row = channel, 32 keys + 8 pad halfs per row.  The scatter writes 2-byte elements; the row pitch (80 B) times
// the 8 rows between two 16-byte channel groups is a multiple of the 128-byte bank period, so without a swizzle every channel
// group of a key lands on the same bank (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.69 - 0.93 in profiles/r02_pmc_sq.txt).
// The 8-key chunk of a row is XORed with the row's channel group (low two bits): channel groups spread over four chunk positions, and the
// b128 fragment reads (one row per lane, 8 consecutive rows share the XOR) stay conflict-free.
__device__ __forceinline__ int vt_col(const int key, const int chgroup) { return key ^ ((chgroup & 3) << 3); }

// K8 on the matrix cores.  A (site, head) problem is only fq x fk x d (16 x 16 x 40): G = 32 / max(fq, fk) problems —
// consecutive heads of one site — are packed along BOTH dimensions of a 32x32 tile, S^T[(g,key), (g',query)], and the
// off-diagonal blocks (g != g') are masked to zero probability, which makes P block-diagonal so that one
// O^T[c, (g',query)] = V^T[c, (g,key)] . P^T product serves all packed problems.  Operands of the first product are
// loaded straight from HBM into MFMA registers (a lane's 16 bytes = 8 consecutive channels of one (frame, head) row;
// MFMA row r carries packed key swap23(r), so the exponentiated scores are already a B operand, as in the flash
// kernel); V goes through a wave-private LDS tile, written transposed ([channel][key]) and read back as b128.
// One wave per G heads, four waves (independent, no barriers) per workgroup.  The first version did these products
// with per-lane dot products over LDS copies of Q, K, V and was VALU/LDS-bound at 2 TB/s.
template <int D>
__global__ __launch_bounds__(256) void temporal_attn_mfma_kernel(const TempParams p) {
    constexpr int DK = (D + 15) / 16;      // k-steps of S^T = K Q^T
    constexpr int DT = (D + 31) / 32;      // 32-row tiles of O^T
    constexpr int DV = D / 8;              // 16-byte chunks per (frame, head) row
    constexpr int VSTR = 40;               // V^T LDS row: 32 keys + 8 pad halfs (5 slots, odd)
    constexpr int NVL = (32 * DV + 63) / 64;
    __shared__ __attribute__((aligned(16))) half_t smem[4][DT * 32 * VSTR];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int fq = p.fq, fk = p.fk;
    const int G = 32 / max(fq, fk);
    const long site = blockIdx.x;
    const long b = blockIdx.z;
    const int head0 = ((int)blockIdx.y * 4 + wave) * G;
    if (head0 >= p.heads) return;          // waves are independent: no barrier below
    half_t* sVT = smem[wave];

    // ---- V: coalesced 16-byte loads, transposed scatter into LDS (zeros for padding keys / heads) ----
    {
        uint4 raw[NVL];
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const int u = lane + 64 * i;
            const int key = u / DV, ch = u - key * DV;
            const int g = key / fk, f = key - g * fk;
            const bool ok = u < 32 * DV && g < G && head0 + g < p.heads;
            raw[i] = ok ? ld16(p.V + ((b * fk + f) * p.hw + site) * p.ldkv + (long)(head0 + g) * D + ch * 8)
                        : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const int u = lane + 64 * i;
            if (u < 32 * DV) {
                const int key = u / DV, ch = u - key * DV;
                const h8 v = as_h8(raw[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) sVT[(ch * 8 + e) * VSTR + vt_col(key, ch)] = v[e];
            }
        }
    }

    // ---- S^T = K Q^T: MFMA row r <- packed key swap23(r), column <- packed query ----
    const int krow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int kg = krow / fk, kf_ = krow - kg * fk;
    const bool kok = kg < G && head0 + kg < p.heads;
    const int qg = l31 / fq, qf_ = l31 - qg * fq;
    const bool qok = qg < G && head0 + qg < p.heads;
    const half_t* kptr = p.K + ((b * fk + kf_) * p.hw + site) * p.ldkv + (long)(head0 + kg) * D + hi * 8;
    const half_t* qptr = p.Q + ((b * fq + qf_) * p.hw + site) * p.ldq + (long)(head0 + qg) * D + hi * 8;
    h8 ka[DK], qb[DK];
#pragma unroll
    for (int t = 0; t < DK; ++t) {
        const bool in = t * 16 + hi * 8 < D;
        ka[t] = (kok && in) ? as_h8(ld16(kptr + t * 16)) : as_h8(make_uint4(0, 0, 0, 0));
        qb[t] = (qok && in) ? as_h8(ld16(qptr + t * 16)) : as_h8(make_uint4(0, 0, 0, 0));
    }
    const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f16v sc = zero;
#pragma unroll
    for (int t = 0; t < DK; ++t) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[t], qb[t], t == 0 ? zero : sc, 0, 0, 0);

    // ---- softmax over the keys of this column's own problem (register r holds packed key (r&3)+4*((r>>2)&1)+8*hi+16*(r>>3)) ----
    const int lo = qg * fk, up = lo + fk;
    const float sl2 = p.scale * 1.44269504088896340736f;
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);
        const bool v = qok && key >= lo && key < up;
        sc[r] = v ? sc[r] * sl2 : -INFINITY;
        mx = fmaxf(mx, sc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (mx == -INFINITY) mx = 0.f;         // padding column: every probability is exp2(-inf) = 0
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sc[r] = __builtin_amdgcn_exp2f(sc[r] - mx);
        sum += sc[r];
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;

    // ---- O^T = V^T P^T (P block-diagonal, unnormalised, fp16; normalised in fp32 afterwards) ----
    f16v o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = zero;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        h8 pf;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) pf[jj] = (half_t)sc[8 * s2 + jj];
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const h8 vf = *reinterpret_cast<const h8*>(sVT + (t * 32 + l31) * VSTR + vt_col(16 * s2 + 8 * hi, (t * 32 + l31) >> 3));
            o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[t], 0, 0, 0);
        }
    }
    // ---- O.  Stored from the accumulator layout, an instruction writes 8 bytes into each of 32 (frame, head) rows — ten such
    // instructions per wave, and the kernel stalled at instruction issue a third of its time (SQ_WAIT_INST_ANY, profiles/r05_pmc_sq.txt).
    // The tile goes through the wave's own V^T region of LDS instead (dead after the last MFMA: the LDS operations of a wave execute in
    // order), written [query column][channel] and read back so that a lane stores 16 bytes and the G heads of a frame form one
    // contiguous run of G * D halfs.  Staged row pitch D + 4 halfs = (D / 2 + 2) dwords = 2 mod 4: the 8-byte writes of a 16-lane group
    // fall into 16 different bank pairs. ----
    if (p.direct_out) {
        if (qok) {
            half_t* orow = p.O + ((b * fq + qf_) * p.hw + site) * p.ldo + (long)(head0 + qg) * D;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c = t * 32 + 8 * g4 + 4 * hi;
                    if (c < D) {
                        h4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = (half_t)(o[t][4 * g4 + e] * inv);
                        *reinterpret_cast<h4*>(orow + c) = pk;
                    }
                }
        }
        return;
    }
    constexpr int OSTR = D + 4;
    static_assert(32 * OSTR <= DT * 32 * VSTR, "the staged output fits the wave's V^T tile");
    static_assert((D / 2 + 2) % 4 == 2, "staged row pitch");
    half_t* sO = sVT;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int c = t * 32 + 8 * g4 + 4 * hi;
            if (c < D) {
                h4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (half_t)(o[t][4 * g4 + e] * inv);
                *reinterpret_cast<h4*>(sO + l31 * OSTR + c) = pk;
            }
        }
    __builtin_amdgcn_sched_barrier(0);
    const int per_frame = G * DV;                   // 16-byte chunks of a frame's run: G heads x D / 8
    const int total = fq * per_frame;
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
        const int u = lane + 64 * i;
        if (u < total) {
            const int f = u / per_frame, rem = u - f * per_frame;
            const int g = rem / DV, ch = rem - g * DV;
            if (head0 + g < p.heads) {
                const half_t* src = sO + (g * fq + f) * OSTR + ch * 8;          // 8-byte aligned
                const h4 lo4 = *reinterpret_cast<const h4*>(src), up4 = *reinterpret_cast<const h4*>(src + 4);
                const h8 v = {lo4[0], lo4[1], lo4[2], lo4[3], up4[0], up4[1], up4[2], up4[3]};
                st16(p.O + ((b * fq + f) * p.hw + site) * p.ldo + (long)(head0 + g) * D + ch * 8, as_u4(v));
            }
        }
    }
}

// Long-clip form of the matrix-core kernel (frame-sharded mode): fq local query frames (<= 32) against
// fk gathered key frames (<= 128).  G = 32 / fq heads are packed along the QUERY dimension only; for every packed head
// g and every block of 32 keys one S^T tile [32 keys of head g] x [(g', query)] is computed and a column keeps it iff
// g == g'; the softmax then runs over the column's own 32*NKB score registers, and the second product walks the same
// (g, key block) pairs with P zeroed for the other heads' columns.  V^T tiles go through the wave-private LDS tile one
// (head, key block) at a time.
template <int D>
__global__ __launch_bounds__(256) void temporal_attn_mfma_long_kernel(const TempParams p) {
    constexpr int DK = (D + 15) / 16;
    constexpr int DT = (D + 31) / 32;
    constexpr int DV = D / 8;
    constexpr int VSTR = 40;
    constexpr int NVL = (32 * DV + 63) / 64;
    constexpr int NKB = 4;                 // key blocks of 32 held in registers: fk <= 128
    __shared__ __attribute__((aligned(16))) half_t smem[4][DT * 32 * VSTR];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int fq = p.fq, fk = p.fk;
    const int G = 32 / fq;
    const int nkb = (fk + 31) >> 5;
    const long site = blockIdx.x;
    const long b = blockIdx.z;
    const int head0 = ((int)blockIdx.y * 4 + wave) * G;
    if (head0 >= p.heads) return;          // waves are independent: no barrier below
    half_t* sVT = smem[wave];

    const int qg = l31 / fq, qf_ = l31 - qg * fq;
    const bool qok = qg < G && head0 + qg < p.heads;
    const half_t* qptr = p.Q + ((b * fq + qf_) * p.hw + site) * p.ldq + (long)(head0 + qg) * D + hi * 8;
    h8 qb[DK];
#pragma unroll
    for (int t = 0; t < DK; ++t)
        qb[t] = (qok && t * 16 + hi * 8 < D) ? as_h8(ld16(qptr + t * 16)) : as_h8(make_uint4(0, 0, 0, 0));

    const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f16v sc[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) sc[kb] = zero;
    const int krow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);      // MFMA row r carries key swap23(r)
    for (int g = 0; g < G; ++g) {
        if (head0 + g >= p.heads) break;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb >= nkb) continue;
            const int key = kb * 32 + krow;
            const half_t* kptr = p.K + ((b * fk + key) * p.hw + site) * p.ldkv + (long)(head0 + g) * D + hi * 8;
            f16v tile = zero;
#pragma unroll
            for (int t = 0; t < DK; ++t) {
                const h8 ka = (key < fk && t * 16 + hi * 8 < D) ? as_h8(ld16(kptr + t * 16)) : as_h8(make_uint4(0, 0, 0, 0));
                tile = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qb[t], tile, 0, 0, 0);
            }
            if (g == qg) sc[kb] = tile;
        }
    }

    // ---- softmax over the fk keys of this column (register r of block kb holds key 32 kb + (r&3) + 4((r>>2)&1) + 8 hi + 16 (r>>3)) ----
    const float sl2 = p.scale * 1.44269504088896340736f;
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);
            const bool v = qok && kb < nkb && key < fk;
            sc[kb][r] = v ? sc[kb][r] * sl2 : -INFINITY;
            mx = fmaxf(mx, sc[kb][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (mx == -INFINITY) mx = 0.f;
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[kb][r] = __builtin_amdgcn_exp2f(sc[kb][r] - mx);
            sum += sc[kb][r];
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;

    // ---- O^T = sum over (head g, key block) of V^T[c, keys] . P^T[keys, (g', query)], P = 0 where g' != g ----
    f16v o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = zero;
    for (int g = 0; g < G; ++g) {
        if (head0 + g >= p.heads) break;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb >= nkb) continue;
            uint4 raw[NVL];
#pragma unroll
            for (int i = 0; i < NVL; ++i) {
                const int u = lane + 64 * i;
                const int kk = u / DV, ch = u - kk * DV;
                const int key = kb * 32 + kk;
                raw[i] = (u < 32 * DV && key < fk)
                             ? ld16(p.V + ((b * fk + key) * p.hw + site) * p.ldkv + (long)(head0 + g) * D + ch * 8)
                             : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NVL; ++i) {
                const int u = lane + 64 * i;
                if (u < 32 * DV) {
                    const int kk = u / DV, ch = u - kk * DV;
                    const h8 v = as_h8(raw[i]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) sVT[(ch * 8 + e) * VSTR + vt_col(kk, ch)] = v[e];
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                h8 pf;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pf[jj] = g == qg ? (half_t)sc[kb][8 * s2 + jj] : (half_t)0.f;
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const h8 vf = *reinterpret_cast<const h8*>(sVT + (t * 32 + l31) * VSTR + vt_col(16 * s2 + 8 * hi, (t * 32 + l31) >> 3));
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[t], 0, 0, 0);
                }
            }
        }
    }
    if (qok) {
        half_t* orow = p.O + ((b * fq + qf_) * p.hw + site) * p.ldo + (long)(head0 + qg) * D;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c = t * 32 + 8 * g4 + 4 * hi;
                if (c < D) {
                    h4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (half_t)(o[t][4 * g4 + e] * inv);
                    *reinterpret_cast<h4*>(orow + c) = pk;
                }
            }
    }
}

template <int D>
int launch_temporal_mfma_long(const TempParams& p, long B, hipStream_t stream) {
    const int G = 32 / p.fq;
    const int groups = (p.heads + G - 1) / G;
    dim3 grid((unsigned)p.hw, (unsigned)((groups + 3) / 4), (unsigned)B);
    hipLaunchKernelGGL((temporal_attn_mfma_long_kernel<D>), grid, dim3(256), 0, stream, p);
    return vsx_check_launch("vsx_temporal_attention_f16");
}

template <int D>
int launch_temporal_mfma(const TempParams& p, long B, hipStream_t stream) {
    const int G = 32 / (p.fq > p.fk ? p.fq : p.fk);
    const int groups = (p.heads + G - 1) / G;
    dim3 grid((unsigned)p.hw, (unsigned)((groups + 3) / 4), (unsigned)B);
    hipLaunchKernelGGL((temporal_attn_mfma_kernel<D>), grid, dim3(256), 0, stream, p);
    return vsx_check_launch("vsx_temporal_attention_f16");
}

}  // namespace

#ifdef VSX_GEMM_TIMING
extern "C" int vsx_attn_debug_buffer(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

static int attention_fwd(const void* Q, const void* K, const void* VT, void* O, float* lse, int64_t lse_ld, int64_t nb, int64_t heads,
                         int64_t nq, int64_t nk, int64_t d, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo, int64_t q_bs,
                         int64_t k_bs, int64_t vt_bs, int64_t o_bs, int64_t kv_div, float scale, vsx_stream_t stream_);

extern "C" int vsx_attention_f16(const void* Q, const void* K, const void* VT, void* O, int64_t nb, int64_t heads,
                                 int64_t nq, int64_t nk, int64_t d, int64_t ldq, int64_t ldk, int64_t ldvt,
                                 int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t vt_bs, int64_t o_bs,
                                 int64_t kv_div, float scale, vsx_stream_t stream) {
    return attention_fwd(Q, K, VT, O, nullptr, 0, nb, heads, nq, nk, d, ldq, ldk, ldvt, ldo, q_bs, k_bs, vt_bs, o_bs, kv_div, scale,
                         stream);
}

extern "C" int vsx_attention_lse_f16(const void* Q, const void* K, const void* VT, void* O, float* lse, int64_t lse_ld,
                                     int64_t nb, int64_t heads, int64_t nq, int64_t nk, int64_t d, int64_t ldq, int64_t ldk,
                                     int64_t ldvt, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t vt_bs, int64_t o_bs,
                                     int64_t kv_div, float scale, vsx_stream_t stream) {
    VSX_REQUIRE(lse != nullptr && lse_ld >= nq, VSX_E_BADSHAPE, "attention_lse: lse buffer / row length");
    return attention_fwd(Q, K, VT, O, lse, lse_ld, nb, heads, nq, nk, d, ldq, ldk, ldvt, ldo, q_bs, k_bs, vt_bs, o_bs, kv_div, scale,
                         stream);
}

static int attention_fwd(const void* Q, const void* K, const void* VT, void* O, float* lse, int64_t lse_ld, int64_t nb, int64_t heads,
                         int64_t nq, int64_t nk, int64_t d, int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo, int64_t q_bs,
                         int64_t k_bs, int64_t vt_bs, int64_t o_bs, int64_t kv_div, float scale, vsx_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VSX_REQUIRE(Q && K && VT && O, VSX_E_BADSHAPE, "attention: null tensor");
    if (nb == 0 || nq == 0) return VSX_OK;
    VSX_REQUIRE(nb > 0 && heads > 0 && nq > 0 && nk > 0 && kv_div > 0, VSX_E_BADSHAPE, "attention: bad sizes");
    VSX_REQUIRE(nb <= 65535 && heads <= 65535, VSX_E_BADSHAPE, "attention: nb/heads exceed grid limits");
    VSX_REQUIRE(vsx_aligned16(Q) && vsx_aligned16(K) && vsx_aligned16(VT) && vsx_aligned16(O), VSX_E_BADSHAPE,
                "attention: tensors must be 16-byte aligned");
    VSX_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, VSX_E_BADSHAPE,
                "attention: row strides must be multiples of 8 (ldo: 4)");
    VSX_REQUIRE(q_bs % 8 == 0 && k_bs % 8 == 0 && vt_bs % 8 == 0 && o_bs % 4 == 0, VSX_E_BADSHAPE,
                "attention: batch strides must be multiples of 8");
    VSX_REQUIRE(ldvt >= ((nk + 7) / 8) * 8, VSX_E_BADSHAPE, "attention: ldvt (%ld) < round_up(nk=%ld, 8)", (long)ldvt,
                (long)nk);
    AttnParams p;
    p.Q = (const half_t*)Q; p.K = (const half_t*)K; p.VT = (const half_t*)VT; p.O = (half_t*)O;
    p.nq = (int)nq; p.nk = (int)nk; p.heads = (int)heads;
    p.ldq = ldq; p.ldk = ldk; p.ldvt = ldvt; p.ldo = ldo;
    p.q_bs = q_bs; p.k_bs = k_bs; p.vt_bs = vt_bs; p.o_bs = o_bs;
    p.kv_div = (int)kv_div;
    p.scale_log2e = scale * 1.44269504088896340736f;
    p.lse = lse;
    p.lse_ld = lse_ld;
    switch (d) {
        case 8: return launch_attn<8>(p, nb, stream);
        case 16: return launch_attn<16>(p, nb, stream);
        case 32: return launch_attn<32>(p, nb, stream);
        case 40: return launch_attn<40>(p, nb, stream);
        case 64: return launch_attn<64>(p, nb, stream);
        case 80: return launch_attn<80>(p, nb, stream);
        case 128: return launch_attn<128>(p, nb, stream);
        case 160: return launch_attn<160>(p, nb, stream);
        default: return vsx_fail(VSX_E_UNSUPPORTED, "attention: head dim %ld not in {8,16,32,40,64,80,128,160}", (long)d);
    }
}

extern "C" int vsx_temporal_attention_f16(const void* Q, const void* K, const void* V, void* O, int64_t B, int64_t fq,
                                          int64_t fk, int64_t hw, int64_t heads, int64_t d, int64_t ldq, int64_t ldkv,
                                          int64_t ldo, float scale, vsx_stream_t stream) {
    VSX_REQUIRE(Q && K && V && O, VSX_E_BADSHAPE, "temporal_attention: null tensor");
    if (B == 0 || hw == 0) return VSX_OK;
    VSX_REQUIRE(B > 0 && fq > 0 && fk > 0 && hw > 0 && heads > 0 && d > 0, VSX_E_BADSHAPE, "temporal_attention: bad sizes");
    VSX_REQUIRE(d % 8 == 0 && ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0, VSX_E_BADSHAPE,
                "temporal_attention: d and row strides must be multiples of 8");
    VSX_REQUIRE(vsx_aligned16(Q) && vsx_aligned16(K) && vsx_aligned16(V) && vsx_aligned16(O), VSX_E_BADSHAPE,
                "temporal_attention: tensors must be 16-byte aligned");
    VSX_REQUIRE(B <= 65535 && heads <= 65535, VSX_E_BADSHAPE, "temporal_attention: grid limits");
    TempParams p;
    p.Q = (const half_t*)Q; p.K = (const half_t*)K; p.V = (const half_t*)V; p.O = (half_t*)O;
    p.fq = (int)fq; p.fk = (int)fk; p.hw = (int)hw; p.heads = (int)heads; p.d = (int)d;
    p.ldq = ldq; p.ldkv = ldkv; p.ldo = ldo; p.scale = scale;
    p.direct_out = vsxg::gemm_option("temporal_out") == 1 ? 1 : 0;
    if (fq <= 32 && fk <= 32) {            // matrix-core kernel for the UNet's head dims
        if (d == 40) return launch_temporal_mfma<40>(p, B, (hipStream_t)stream);
        if (d == 80) return launch_temporal_mfma<80>(p, B, (hipStream_t)stream);
        if (d == 160) return launch_temporal_mfma<160>(p, B, (hipStream_t)stream);
    } else if (fq <= 32 && fk <= 128) {    // long-clip mode: local query frames against the gathered key frames
        if (d == 40) return launch_temporal_mfma_long<40>(p, B, (hipStream_t)stream);
        if (d == 80) return launch_temporal_mfma_long<80>(p, B, (hipStream_t)stream);
        if (d == 160) return launch_temporal_mfma_long<160>(p, B, (hipStream_t)stream);
    }
    const size_t slice = (((size_t)(fq + 2 * fk) * d * sizeof(half_t) + (size_t)fq * fk * sizeof(float)) + 15) & ~(size_t)15;
    const size_t smem = 4 * slice;
    VSX_REQUIRE(smem <= 160 * 1024, VSX_E_UNSUPPORTED, "temporal_attention: %zu bytes of LDS needed (> 160 KiB)", smem);
    static size_t smem_attr = 64 * 1024;
    if (smem > smem_attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "temporal_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
        smem_attr = 160 * 1024;
    }
    dim3 grid((unsigned)hw, (unsigned)((heads + 3) / 4), (unsigned)B);
    hipLaunchKernelGGL(temporal_attn_kernel, grid, dim3(256), smem, (hipStream_t)stream, p);
    return vsx_check_launch("vsx_temporal_attention_f16");
}
