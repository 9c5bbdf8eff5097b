// K5b — backward of the fused attention (vsx_attention_lse_f16) for gfx950: data gradients dQ, dK, dV without ever
// materialising the [heads, nq, nk] probabilities (the adapter training step, trainer_videoswap.py:33-97, differentiates
// through every self-attention of the frozen UNet: at the 64x64 level that matrix is 4.3 GB per layer, and the round-3
// backward wrote, transposed and re-read it three times).
//
// With s = scale * q.k, P = softmax(s) = exp2(s * log2(e) - lse), dP = dO V^T, delta_q = sum_c dO[q,c] O[q,c]:
//     dS = scale * P o (dP - delta),   dQ = dS K,   dK = dS^T Q,   dV = P^T dO.
// Two kernels on v_mfma_f32_32x32x16_f16, both built like the forward kernel (attention.hip: K / V tiles streamed into a
// two-slot LDS ring by LDS-DMA, every product computed in the transposed form that keeps the per-row softmax quantities
// lane-local, scores and probabilities never leave the register file), no atomics — the result is deterministic:
//
//   attn_bwd_dq_kernel   workgroup = 128 queries of one (image, head), loop over key tiles of 64:
//                          S^T[key,q] = K Q^T, dP^T[key,q] = V dO^T  (A = K / V rows from LDS, B = Q / dO fragments in registers)
//                          dS^T = scale * P^T o (dP^T - delta_q)      (lane-local: a lane owns one query column)
//                          dQ^T[c,q] += K^T[c,key] dS^T[key,q]        (A = K^T tile from LDS, B = dS^T straight from registers)
//   attn_bwd_dkv_kernel  workgroup = 128 keys of one (image, head), loop over query tiles of 64:
//                          S[q,key] = Q K^T, dP[q,key] = dO V^T       (A = Q / dO rows from LDS, B = K / V fragments in registers)
//                          P, dS as above with lse / delta of the tile's queries read from LDS
//                          dV^T[c,key] += dO^T[c,q] P[q,key],  dK^T[c,key] += Q^T[c,q] dS[q,key]
//
// The transposed operand tiles (K^T, Q^T, dO^T: [heads*d, n] per image, rows padded to a multiple of 8 with zeros) are
// handed in by the caller (one strided copy each, a few percent of the attention's bytes); lse comes from the forward
// pass, delta from vsx_attention_bwd_f16 itself (attn_delta_kernel).  Every partial tile is handled by the buffer
// descriptor's zero fill: a query row past the end has Q = dO = 0, hence dS = 0 and a zero column of dO^T; a key past the
// end is masked to P = 0 in the dQ kernel and simply not stored in the dK / dV kernel.
#include "common.h"

#include <type_traits>

namespace {

struct AttnBwdParams {
    const half_t *Q, *K, *V, *dO;       // [nb | nkvb, n, heads*d], contiguous rows of C = heads*d
    const half_t *QT, *KT, *dOT;        // [nb | nkvb, heads*d, ldt]: transposed per image, zero-padded rows
    const float *lse, *delta;           // [nb, heads, lds]: lds >= round_up(nq, 64), zero beyond nq
    half_t *dQ, *dK, *dV;               // row-major like Q / K / V
    int nq, nk, heads, kv_div;
    long C, ldtq, ldtk, lds;
    float scale, scale_log2e;
};

constexpr int BT = 64;            // keys (dQ kernel) / queries (dK dV kernel) per tile
constexpr int TSTR = 72;          // transposed LDS row: 64 columns + one 16-byte dummy slot (odd slot count)
constexpr int OOB_OFF = (int)0x80000000;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// LDS row r of a row-major tile holds source row r with bits 2 and 3 swapped: in the 32x32 C/D layout a lane then owns
// 8 CONSECUTIVE rows per (tile, half), which is the k-slice the B operand of the second product needs (attention.hip)
__device__ __forceinline__ int perm_row(const int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

template <int D>
struct Geo {
    static constexpr int DK = (D + 15) / 16;        // k-steps of a product over the head dim
    static constexpr int DT = (D + 31) / 32;        // 32-row tiles of a [d, n] result
    static constexpr int RSTR = DK * 16 + 8;        // row-major LDS row (halfs): odd number of 16-byte slots
    static constexpr int RSPR = RSTR / 8, TSPR = TSTR / 8;
    static constexpr int R_UNITS = BT * RSPR, T_UNITS = D * TSPR;       // 16-byte units per tile
    static constexpr int NRI = (R_UNITS + 255) / 256, NTI = (T_UNITS + 255) / 256;     // LDS-DMA instructions per wave
    static constexpr int R_BYTES = BT * RSTR * 2, T_BYTES = DT * 32 * TSTR * 2;
};

// Loop-invariant per-lane source offsets of this wave's DMA instructions: row-major tile (row pitch ld elements) ...
template <int D>
__device__ __forceinline__ void row_offsets(int (&v)[Geo<D>::NRI], const int wave, const int lane, const long ld) {
    using G = Geo<D>;
#pragma unroll
    for (int i = 0; i < G::NRI; ++i) {
        const int u = (wave * G::NRI + i) * 64 + lane;
        const int row = u / G::RSPR, slot = u - row * G::RSPR;
        v[i] = (u < G::R_UNITS && slot * 8 < D) ? (int)(((long)perm_row(row) * ld + slot * 8) * 2) : OOB_OFF;
    }
}
// ... and transposed tile (rows = head-dim index, row pitch ldt elements)
template <int D>
__device__ __forceinline__ void tr_offsets(int (&v)[Geo<D>::NTI], const int wave, const int lane, const long ldt) {
    using G = Geo<D>;
#pragma unroll
    for (int i = 0; i < G::NTI; ++i) {
        const int u = (wave * G::NTI + i) * 64 + lane;
        const int row = u / G::TSPR, slot = u - row * G::TSPR;
        v[i] = (u < G::T_UNITS && slot < 8) ? (int)(((long)row * ldt + slot * 8) * 2) : OOB_OFF;
    }
}

// One row-major tile: rows [j0, j0 + 64) of a [n, C] matrix (head slice already in the descriptor base) -> LDS at `dst`.
template <int D>
__device__ __forceinline__ void issue_rows(const __amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, const int (&v)[Geo<D>::NRI],
                                           const int wave, const int lane, const int j0, const int n, const long ld) {
    using G = Geo<D>;
    const int soff = (int)((long)j0 * ld * 2);
    const bool partial = j0 + BT > n;       // the descriptor's range check does not cover the scalar offset
#pragma unroll
    for (int i = 0; i < G::NRI; ++i)
        if ((wave * G::NRI + i) * 64 < G::R_UNITS) {       // wave-uniform
            int o = v[i];
            if (partial && j0 + perm_row(((wave * G::NRI + i) * 64 + lane) / G::RSPR) >= n) o = OOB_OFF;
            // lanes past the end of the tile are switched off (EXEC): a DMA lane always writes its 16 bytes
            if ((wave * G::NRI + i) * 64 + lane < G::R_UNITS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + (wave * G::NRI + i) * 1024), 16, o, soff, 0, 0);
        }
}
// One transposed tile: columns [j0, j0 + 64) of a [D, ldt] matrix -> LDS rows of TSTR halfs.
template <int D>
__device__ __forceinline__ void issue_tr(const __amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, const int (&v)[Geo<D>::NTI],
                                         const int wave, const int lane, const int j0, const long ldt) {
    using G = Geo<D>;
    const int soff = j0 * 2;
    const bool partial = j0 + BT > ldt;
#pragma unroll
    for (int i = 0; i < G::NTI; ++i)
        if ((wave * G::NTI + i) * 64 < G::T_UNITS) {
            int o = v[i];
            if (partial) {
                const int u = (wave * G::NTI + i) * 64 + lane;
                if (j0 + (u - (u / G::TSPR) * G::TSPR) * 8 >= ldt) o = OOB_OFF;
            }
            if ((wave * G::NTI + i) * 64 + lane < G::T_UNITS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + (wave * G::NTI + i) * 1024), 16, o, soff, 0, 0);
        }
}

__device__ __forceinline__ int xcd_order(int wg, const int nwg) {       // attention.hip: contiguous ranges per XCD
    const int qq = nwg >> 3, rr = nwg & 7;
    const int xcd = wg & 7, local = wg >> 3;
    return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + local;
}

// C/D registers of a [64 x 32] pair of score tiles -> the B operand (8 halfs) of k-step (kt, s2) of the second product
__device__ __forceinline__ h8 b_operand(const f16v& t, const int s2) {
    h8 r;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) r[jj] = (half_t)t[8 * s2 + jj];
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnBwdParams p) {
    using G = Geo<D>;
    constexpr int DK = G::DK, DT = G::DT, RSTR = G::RSTR;
    constexpr int STAGE = 2 * G::R_BYTES + G::T_BYTES;          // K rows | V rows | K^T
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wg = xcd_order(blockIdx.x, gridDim.x);
    const int qtiles = (p.nq + 127) / 128;
    const int qt = wg % qtiles;
    const int h = (wg / qtiles) % p.heads;
    const long b = wg / (qtiles * p.heads);
    const long kvb = b / p.kv_div;
    const int q = qt * 128 + wave * 32 + l31;
    const bool qok = q < p.nq;

    // B operands: lane (q, hi) holds Q[q, t*16 + hi*8 .. +7] and dO[q, ...]
    h8 qf[DK], dof[DK];
    {
        const long row = (b * p.nq + (qok ? q : 0)) * p.C + h * D;
#pragma unroll
        for (int t = 0; t < DK; ++t) {
            const int d0 = t * 16 + hi * 8;
            const bool ok = qok && d0 < D;
            qf[t] = ok ? as_h8(ld16(p.Q + row + d0)) : as_h8(make_uint4(0, 0, 0, 0));
            dof[t] = ok ? as_h8(ld16(p.dO + row + d0)) : as_h8(make_uint4(0, 0, 0, 0));
        }
    }
    const long st = (b * p.heads + h) * p.lds + (qok ? q : 0);
    const float lse_q = qok ? p.lse[st] : 0.f;
    const float delta_q = qok ? p.delta[st] : 0.f;

    const half_t* Kb = p.K + kvb * p.nk * p.C + h * D;
    const half_t* Vb = p.V + kvb * p.nk * p.C + h * D;
    const half_t* KTb = p.KT + (kvb * p.C + (long)h * D) * p.ldtk;
    const int rbytes = (int)((((long)p.nk - 1) * p.C + D) * 2);
    const __amdgpu_buffer_rsrc_t rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Kb), 0, rbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Vb), 0, rbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcKT =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(KTb), 0, (int)((long)D * p.ldtk * 2), 0x00020000);
    int vr[G::NRI], vt[G::NTI];
    row_offsets<D>(vr, wave, lane, p.C);
    tr_offsets<D>(vt, wave, lane, p.ldtk);
    // rows >= D of the K^T tiles are never written by the DMA: zero them once
    if constexpr (DT * 32 > D) {
        constexpr int PADN = (DT * 32 - D) * TSTR;
        for (int i = tid; i < 2 * PADN; i += 256) {
            const int sg = i / PADN, r = i - sg * PADN;
            reinterpret_cast<half_t*>(smem + sg * STAGE + 2 * G::R_BYTES)[D * TSTR + r] = (half_t)0.f;
        }
    }
    auto issue = [&](const int j0, const int stage) {
        unsigned char* sb = smem + stage * STAGE;
        issue_rows<D>(rsrcK, sb, vr, wave, lane, j0, p.nk, p.C);
        issue_rows<D>(rsrcV, sb + G::R_BYTES, vr, wave, lane, j0, p.nk, p.C);
        issue_tr<D>(rsrcKT, sb + 2 * G::R_BYTES, vt, wave, lane, j0, p.ldtk);
    };

    f16v dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;

    issue(0, 0);
    int stage = 0;
    auto tile = [&](const int j0, auto masked_c) {
        constexpr bool MASKED = decltype(masked_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (j0 + BT < p.nk) issue(j0 + BT, stage ^ 1);
        const half_t* sK = reinterpret_cast<const half_t*>(smem + stage * STAGE);
        const half_t* sV = reinterpret_cast<const half_t*>(smem + stage * STAGE + G::R_BYTES);
        const half_t* sKT = reinterpret_cast<const half_t*>(smem + stage * STAGE + 2 * G::R_BYTES);
        // ---- S^T = K Q^T and dP^T = V dO^T: two 32-key row tiles each ----
        f16v s[2], dp[2];
        const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int ro = (kt * 32 + l31) * RSTR + hi * 8;
#pragma unroll
            for (int t = 0; t < DK; ++t) {
                const h8 kf = *reinterpret_cast<const h8*>(sK + ro + t * 16);
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[t], t == 0 ? zero : s[kt], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < DK; ++t) {
                const h8 vf = *reinterpret_cast<const h8*>(sV + ro + t * 16);
                dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, dof[t], t == 0 ? zero : dp[kt], 0, 0, 0);
            }
        }
        // ---- dS^T = scale * P^T o (dP^T - delta_q), P^T = exp2(S^T * scale * log2(e) - lse_q) ----
        const float neg_l = -lse_q;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], p.scale_log2e, neg_l));
                if constexpr (MASKED) {
                    // C/D row (r&3) + 8*(r>>2) + 4*hi of the tile holds key (r&3) + 4*((r>>2)&1) + 8*hi + 16*(r>>3)
                    const int key = j0 + kt * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);
                    if (key >= p.nk) pr = 0.f;
                }
                s[kt][r] = pr * (dp[kt][r] - delta_q) * p.scale;
            }
        // ---- dQ^T += K^T dS^T ----
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const h8 pf = b_operand(s[kt], s2);
                const int c0 = kt * 32 + 16 * s2 + 8 * hi;
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const h8 af = *reinterpret_cast<const h8*>(sKT + (t * 32 + l31) * TSTR + c0);
                    dq[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, pf, dq[t], 0, 0, 0);
                }
            }
        stage ^= 1;
    };
    {
        int j0 = 0;
        for (; j0 + BT <= p.nk; j0 += BT) tile(j0, std::false_type{});
        if (j0 < p.nk) tile(j0, std::true_type{});
    }
    // ---- store: lane (q, hi) holds dQ[q, t*32 + 8*g + 4*hi + 0..3] ----
    if (qok) {
        half_t* orow = p.dQ + (b * p.nq + q) * p.C + h * D;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = t * 32 + 8 * g + 4 * hi;
                if (c < D) {
                    h4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (half_t)dq[t][4 * g + e];
                    *reinterpret_cast<h4*>(orow + c) = pk;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dK, dV
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnBwdParams p) {
    using G = Geo<D>;
    constexpr int DK = G::DK, DT = G::DT, RSTR = G::RSTR;
    constexpr int ST_BYTES = 512;                                // lse[64] | delta[64] of the tile's queries (fp32)
    constexpr int STAGE = 2 * G::R_BYTES + 2 * G::T_BYTES + ST_BYTES;      // Q rows | dO rows | Q^T | dO^T | statistics
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wg = xcd_order(blockIdx.x, gridDim.x);
    const int ktiles = (p.nk + 127) / 128;
    const int ktile = wg % ktiles;
    const int h = (wg / ktiles) % p.heads;
    const long b = wg / (ktiles * p.heads);
    const int key = ktile * 128 + wave * 32 + l31;
    const bool kok = key < p.nk;

    // B operands: lane (key, hi) holds K[key, t*16 + hi*8 .. +7] and V[key, ...]
    h8 kf[DK], vf[DK];
    {
        const long row = (b * p.nk + (kok ? key : 0)) * p.C + h * D;
#pragma unroll
        for (int t = 0; t < DK; ++t) {
            const int d0 = t * 16 + hi * 8;
            const bool ok = kok && d0 < D;
            kf[t] = ok ? as_h8(ld16(p.K + row + d0)) : as_h8(make_uint4(0, 0, 0, 0));
            vf[t] = ok ? as_h8(ld16(p.V + row + d0)) : as_h8(make_uint4(0, 0, 0, 0));
        }
    }
    const half_t* Qb = p.Q + b * p.nq * p.C + h * D;
    const half_t* Ob = p.dO + b * p.nq * p.C + h * D;
    const half_t* QTb = p.QT + (b * p.C + (long)h * D) * p.ldtq;
    const half_t* OTb = p.dOT + (b * p.C + (long)h * D) * p.ldtq;
    const float* Lb = p.lse + (b * p.heads + h) * p.lds;
    const float* Db = p.delta + (b * p.heads + h) * p.lds;
    const int rbytes = (int)((((long)p.nq - 1) * p.C + D) * 2);
    const __amdgpu_buffer_rsrc_t rsrcQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Qb), 0, rbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Ob), 0, rbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcQT =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(QTb), 0, (int)((long)D * p.ldtq * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcOT =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(OTb), 0, (int)((long)D * p.ldtq * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Lb), 0, (int)(p.lds * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Db), 0, (int)(p.lds * 4), 0x00020000);
    int vr[G::NRI], vt[G::NTI];
    row_offsets<D>(vr, wave, lane, p.C);
    tr_offsets<D>(vt, wave, lane, p.ldtq);
    if constexpr (DT * 32 > D) {        // rows >= D of the four transposed tiles are never written by the DMA
        constexpr int PADN = (DT * 32 - D) * TSTR;
        for (int i = tid; i < 4 * PADN; i += 256) {
            const int which = i / PADN, r = i - which * PADN;
            reinterpret_cast<half_t*>(smem + (which >> 1) * STAGE + 2 * G::R_BYTES + (which & 1) * G::T_BYTES)[D * TSTR + r] = (half_t)0.f;
        }
    }
    auto issue = [&](const int j0, const int stage) {
        unsigned char* sb = smem + stage * STAGE;
        issue_rows<D>(rsrcQ, sb, vr, wave, lane, j0, p.nq, p.C);
        issue_rows<D>(rsrcO, sb + G::R_BYTES, vr, wave, lane, j0, p.nq, p.C);
        issue_tr<D>(rsrcQT, sb + 2 * G::R_BYTES, vt, wave, lane, j0, p.ldtq);
        issue_tr<D>(rsrcOT, sb + 2 * G::R_BYTES + G::T_BYTES, vt, wave, lane, j0, p.ldtq);
        if (wave == 0 && lane < 16) {       // 64 floats each: the statistics rows are padded to a multiple of 64 with zeros
            unsigned char* ss = sb + 2 * G::R_BYTES + 2 * G::T_BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcL, (lds_ptr_t)ss, 16, lane * 16, j0 * 4, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcD, (lds_ptr_t)(ss + 256), 16, lane * 16, j0 * 4, 0, 0);
        }
    };

    f16v dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }

    issue(0, 0);
    int stage = 0;
    for (int j0 = 0; j0 < p.nq; j0 += BT, stage ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (j0 + BT < p.nq) issue(j0 + BT, stage ^ 1);
        const half_t* sQ = reinterpret_cast<const half_t*>(smem + stage * STAGE);
        const half_t* sO = reinterpret_cast<const half_t*>(smem + stage * STAGE + G::R_BYTES);
        const half_t* sQT = reinterpret_cast<const half_t*>(smem + stage * STAGE + 2 * G::R_BYTES);
        const half_t* sOT = reinterpret_cast<const half_t*>(smem + stage * STAGE + 2 * G::R_BYTES + G::T_BYTES);
        const float* sL = reinterpret_cast<const float*>(smem + stage * STAGE + 2 * G::R_BYTES + 2 * G::T_BYTES);
        // ---- S = Q K^T and dP = dO V^T: two 32-query row tiles each (rows = queries, columns = this wave's keys) ----
        f16v s[2], dp[2];
        const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int ro = (qt * 32 + l31) * RSTR + hi * 8;
#pragma unroll
            for (int t = 0; t < DK; ++t) {
                const h8 af = *reinterpret_cast<const h8*>(sQ + ro + t * 16);
                s[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, kf[t], t == 0 ? zero : s[qt], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < DK; ++t) {
                const h8 af = *reinterpret_cast<const h8*>(sO + ro + t * 16);
                dp[qt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, vf[t], t == 0 ? zero : dp[qt], 0, 0, 0);
            }
        }
        // ---- P = exp2(S * scale * log2(e) - lse_q), dS = scale * P o (dP - delta_q): register r = 8 s2 + jj of tile qt is
        // query qt*32 + 16 s2 + 8 hi + jj (the Q / dO rows are stored permuted), so a lane reads 8 consecutive statistics ----
        // (a query past the end has Q = dO = 0 and zero statistics: P = 1 meets a zero column of dO^T, dS = 0)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int q8 = qt * 32 + 16 * s2 + 8 * hi;
                const f4v l0 = *reinterpret_cast<const f4v*>(sL + q8), l1 = *reinterpret_cast<const f4v*>(sL + q8 + 4);
                const f4v d0 = *reinterpret_cast<const f4v*>(sL + 64 + q8), d1 = *reinterpret_cast<const f4v*>(sL + 64 + q8 + 4);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int r = 8 * s2 + jj;
                    const float lq = jj < 4 ? l0[jj & 3] : l1[jj & 3], dq_ = jj < 4 ? d0[jj & 3] : d1[jj & 3];
                    const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][r], p.scale_log2e, -lq));
                    s[qt][r] = pr;
                    dp[qt][r] = pr * (dp[qt][r] - dq_) * p.scale;
                }
            }
        // ---- dV^T += dO^T P, dK^T += Q^T dS ----
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const h8 pf = b_operand(s[qt], s2), sf = b_operand(dp[qt], s2);
                const int c0 = qt * 32 + 16 * s2 + 8 * hi;
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const h8 ao = *reinterpret_cast<const h8*>(sOT + (t * 32 + l31) * TSTR + c0);
                    dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ao, pf, dv[t], 0, 0, 0);
                    const h8 aq = *reinterpret_cast<const h8*>(sQT + (t * 32 + l31) * TSTR + c0);
                    dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq, sf, dk[t], 0, 0, 0);
                }
            }
    }
    // ---- store: lane (key, hi) holds dK[key, t*32 + 8*g + 4*hi + 0..3] (and dV) ----
    if (kok) {
        half_t* krow = p.dK + (b * p.nk + key) * p.C + h * D;
        half_t* vrow = p.dV + (b * p.nk + key) * p.C + h * D;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = t * 32 + 8 * g + 4 * hi;
                if (c < D) {
                    h4 a, v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = (half_t)dk[t][4 * g + e]; v[e] = (half_t)dv[t][4 * g + e]; }
                    *reinterpret_cast<h4*>(krow + c) = a;
                    *reinterpret_cast<h4*>(vrow + c) = v;
                }
            }
    }
}

// delta[b, h, q] = sum_c dO[b, q, h*d + c] * O[b, q, h*d + c] (fp32), zero for q in [nq, lds).  One thread per (b, h, q).
__global__ __launch_bounds__(256) void attn_delta_kernel(const half_t* __restrict__ dO, const half_t* __restrict__ O, long nb,
                                                         int heads, int nq, int d, long lds, float* __restrict__ delta) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nb * heads * lds) return;
    const int q = (int)(i % lds);
    const int h = (int)((i / lds) % heads);
    const long b = i / (lds * heads);
    float acc = 0.f;
    if (q < nq) {
        const long row = (b * nq + q) * (long)heads * d + (long)h * d;
        for (int c = 0; c < d; c += 8) {
            const h8 a = as_h8(ld16(dO + row + c)), o = as_h8(ld16(O + row + c));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_fmaf((float)a[e], (float)o[e], acc);
        }
    }
    delta[i] = acc;
}

template <int D>
int launch_bwd(const AttnBwdParams& p, long nb, hipStream_t stream) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<D>), dim3((unsigned)(((p.nq + 127) / 128) * (long)p.heads * nb)), dim3(256), 0, stream, p);
    int rc = vsx_check_launch("vsx_attention_bwd_f16 (dQ)");
    if (rc || p.dK == nullptr) return rc;
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<D>), dim3((unsigned)(((p.nk + 127) / 128) * (long)p.heads * nb)), dim3(256), 0, stream, p);
    return vsx_check_launch("vsx_attention_bwd_f16 (dK, dV)");
}

}  // namespace

extern "C" int64_t vsx_attention_bwd_supported(int64_t d) { return d == 40 || d == 64 || d == 80; }

extern "C" int vsx_attention_bwd_f16(const void* Q, const void* K, const void* V, const void* O, const void* dO, const void* QT,
                                     const void* KT, const void* dOT, const float* lse, float* delta, void* dQ, void* dK,
                                     void* dV, int64_t nb, int64_t heads, int64_t nq, int64_t nk, int64_t d, int64_t ldtq,
                                     int64_t ldtk, int64_t lds, int64_t kv_div, float scale, vsx_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VSX_REQUIRE(Q && K && V && O && dO && KT && lse && delta && dQ, VSX_E_BADSHAPE, "attention_bwd: null tensor");
    VSX_REQUIRE((dK == nullptr) == (dV == nullptr), VSX_E_BADSHAPE, "attention_bwd: dK and dV go together");
    VSX_REQUIRE(dK == nullptr || (QT && dOT && kv_div == 1), VSX_E_BADSHAPE,
                "attention_bwd: key / value gradients need Q^T, dO^T and unshared K / V");
    if (nb == 0 || nq == 0) return VSX_OK;
    VSX_REQUIRE(nb > 0 && heads > 0 && nq > 0 && nk > 0 && kv_div > 0 && nb % kv_div == 0, VSX_E_BADSHAPE, "attention_bwd: bad sizes");
    VSX_REQUIRE(vsx_attention_bwd_supported(d), VSX_E_UNSUPPORTED, "attention_bwd: head dim %ld not in {40, 64, 80}", (long)d);
    VSX_REQUIRE(ldtk % 8 == 0 && ldtk >= ((nk + 7) / 8) * 8 && (dK == nullptr || (ldtq % 8 == 0 && ldtq >= ((nq + 7) / 8) * 8)),
                VSX_E_BADSHAPE, "attention_bwd: transposed rows must be padded to a multiple of 8");
    VSX_REQUIRE(lds % 64 == 0 && lds >= nq, VSX_E_BADSHAPE, "attention_bwd: statistics rows must be padded to a multiple of 64");
    VSX_REQUIRE(vsx_aligned16(Q) && vsx_aligned16(K) && vsx_aligned16(V) && vsx_aligned16(O) && vsx_aligned16(dO) &&
                    vsx_aligned16(QT) && vsx_aligned16(KT) && vsx_aligned16(dOT) && vsx_aligned16(lse) && vsx_aligned16(delta) &&
                    vsx_aligned16(dQ) && vsx_aligned16(dK) && vsx_aligned16(dV),
                VSX_E_BADSHAPE, "attention_bwd: tensors must be 16-byte aligned");
    const long C = heads * d;
    VSX_REQUIRE(nq * C < (1L << 30) && nk * C < (1L << 30) && C * ldtq < (1L << 30) && C * ldtk < (1L << 30), VSX_E_UNSUPPORTED,
                "attention_bwd: an image exceeds the 31-bit buffer offsets");
    const long nstat = nb * heads * lds;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((nstat + 255) / 256)), dim3(256), 0, stream, (const half_t*)dO,
                       (const half_t*)O, (long)nb, (int)heads, (int)nq, (int)d, (long)lds, delta);
    int rc = vsx_check_launch("vsx_attention_bwd_f16 (delta)");
    if (rc) return rc;
    AttnBwdParams p;
    p.Q = (const half_t*)Q; p.K = (const half_t*)K; p.V = (const half_t*)V; p.dO = (const half_t*)dO;
    p.QT = (const half_t*)QT; p.KT = (const half_t*)KT; p.dOT = (const half_t*)dOT;
    p.lse = lse; p.delta = delta;
    p.dQ = (half_t*)dQ; p.dK = (half_t*)dK; p.dV = (half_t*)dV;
    p.nq = (int)nq; p.nk = (int)nk; p.heads = (int)heads; p.kv_div = (int)kv_div;
    p.C = C; p.ldtq = ldtq; p.ldtk = ldtk; p.lds = lds;
    p.scale = scale;
    p.scale_log2e = scale * 1.44269504088896340736f;
    switch (d) {
        case 40: return launch_bwd<40>(p, nb, stream);
        case 64: return launch_bwd<64>(p, nb, stream);
        case 80: return launch_bwd<80>(p, nb, stream);
        default: return vsx_fail(VSX_E_UNSUPPORTED, "attention_bwd: head dim %ld", (long)d);
    }
}
