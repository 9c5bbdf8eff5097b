// K1/K2 — fp16 MFMA GEMM and implicit-GEMM convolution for gfx950 (MI355X).
//
//   C[m, n] = epilogue(alpha * sum_k A[m, k] * B[n, k])
//
// Design (CDNA4): 256-thread workgroups = 4 wave64 in a 2x2 grid, each wave owns a
// (BM/2)x(BN/2) sub-tile built from v_mfma_f32_32x32x16_f16 tiles with fp32 accumulators.
// K is walked in BK=64 slabs staged HBM -> registers -> LDS with two LDS buffers and ONE
// barrier per slab: the global loads of slab t+1 are issued before the MFMAs of slab t and are
// written to the other LDS buffer after them.  LDS rows are padded to 72 halfs (144 B = 9
// 16-byte slots, odd) so the ds_read_b128 fragment reads of 16 distinct rows hit 16 distinct
// slots (conflict-free).
//
// The A operand has two loaders:
//   a_mode 0: plain row-major [M, K];
//   a_mode 1: implicit im2col of a channels-last image [nimg, H, W, C1(+C2)] for a ks x ks conv
//             (stride 1/2, zero pad ks/2), optionally reading a half-resolution source as a
//             nearest-2x upsample, optionally concatenating two sources on the channel axis.
//             This removes the reference's F.interpolate tensor (resnet.py:54), torch.cat
//             (unet_blocks.py:618) and the two rearrange copies per conv (resnet.py:14-16).
// Epilogue: alpha, bias[n], per-image row vector (time embedding, resnet.py:172-176), residual,
// GEGLU (h * gelu(g) with h/g column blocks interleaved per wave), transposed store (V^T for
// the attention kernel).
#include "common.h"
#include <vector>

namespace {

constexpr int BK = 64;
constexpr int LSTR = BK + 8;  // LDS row stride in halfs

struct GemmParams {
    const half_t* A;
    const half_t* A2;
    const half_t* B;
    half_t* C;
    const half_t* bias;
    const half_t* rowvec;
    const half_t* residual;
    long M, N, K;
    long lda, ldb, ldc, ldr;
    long a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1, r_bs0, r_bs1;
    long c_rows_per_img, c_img_stride, rows_per_vec;
    int batch1;
    int a_mode, H, W, C1, C2, Ho, Wo, ks, stride, ups;
    int geglu, c_mode, c_pack4;
    int tiles_n;
    float alpha;
};

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AV = BM / 32, BV = BN / 32;  // 16-byte vectors per thread per slab

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* sA = reinterpret_cast<half_t*>(smem_raw);  // [2][BM][LSTR]
    half_t* sB = sA + 2 * BM * LSTR;                   // [2][BN][LSTR]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;

    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const long m0 = (long)tile_m * BM;
    // geglu: a BN-wide tile of B rows produces BN/2 output columns
    const long n0 = p.geglu ? (long)tile_n * (BN / 2) : (long)tile_n * BN;

    const int z = blockIdx.z;
    const int z0 = z / p.batch1, z1 = z - z0 * p.batch1;
    const half_t* Ab = p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const half_t* Bb = p.B + z0 * p.b_bs0 + z1 * p.b_bs1;

    // ---- loader coordinates: thread owns column-vector lcol of rows lrow + 32*v ----
    const int lrow = tid >> 3;
    const int lcol = (tid & 7) * 8;

    // A rows
    bool a_ok[AV];
    long a_off[AV];           // a_mode 0: element offset of the row
    int a_img[AV], a_ho[AV], a_wo[AV];
#pragma unroll
    for (int v = 0; v < AV; ++v) {
        const long m = m0 + lrow + 32 * v;
        a_ok[v] = m < p.M;
        a_off[v] = m * p.lda;
        if (p.a_mode == 1) {
            const long hw = (long)p.Ho * p.Wo;
            const long mm = a_ok[v] ? m : 0;
            a_img[v] = (int)(mm / hw);
            const int rem = (int)(mm - (long)a_img[v] * hw);
            a_ho[v] = rem / p.Wo;
            a_wo[v] = rem - a_ho[v] * p.Wo;
        } else {
            a_img[v] = 0; a_ho[v] = 0; a_wo[v] = 0;
        }
    }
    // B rows (with the GEGLU h/g interleave: 32-column blocks h0 g0 h1 g1)
    bool b_ok[BV];
    long b_off[BV];
#pragma unroll
    for (int v = 0; v < BV; ++v) {
        const int j = lrow + 32 * v;  // row inside the tile
        long n;
        if (p.geglu) {
            const int q32 = j >> 5;  // == v
            const long oc = n0 + (q32 >> 1) * 32 + (j & 31);
            b_ok[v] = oc < p.N;
            n = oc + ((q32 & 1) ? p.N : 0);
        } else {
            n = n0 + j;
            b_ok[v] = n < p.N;
        }
        b_off[v] = n * p.ldb;
    }

    const int Ctot = p.C1 + p.C2;
    const int Hs = p.ups ? (p.H >> 1) : p.H;
    const int Ws = p.ups ? (p.W >> 1) : p.W;
    const int pad = p.ks >> 1;

    uint4 ra[AV], rb[BV];

    auto gload = [&](int kt) {
        const long k = (long)kt * BK + lcol;
        const bool kok = k < p.K;
        if (p.a_mode == 0) {
#pragma unroll
            for (int v = 0; v < AV; ++v) {
                ra[v] = (a_ok[v] && kok) ? ld16(Ab + a_off[v] + k) : make_uint4(0, 0, 0, 0);
            }
        } else {
            const int kk = kok ? (int)k : 0;
            const int tap = kk / Ctot;
            const int ci = kk - tap * Ctot;
            const int kh = tap / p.ks;
            const int kw = tap - kh * p.ks;
            const bool second = ci >= p.C1;
            const half_t* src = second ? p.A2 : p.A;
            const int cs = second ? p.C2 : p.C1;
            const int cc = second ? ci - p.C1 : ci;
#pragma unroll
            for (int v = 0; v < AV; ++v) {
                const int hh = a_ho[v] * p.stride + kh - pad;
                const int ww = a_wo[v] * p.stride + kw - pad;
                const bool ok = a_ok[v] && kok && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
                const int hsrc = p.ups ? (hh >> 1) : hh;
                const int wsrc = p.ups ? (ww >> 1) : ww;
                const long pix = ((long)a_img[v] * Hs + hsrc) * Ws + wsrc;
                ra[v] = ok ? ld16(src + pix * cs + cc) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int v = 0; v < BV; ++v) {
            rb[v] = (b_ok[v] && kok) ? ld16(Bb + b_off[v] + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
        half_t* a = sA + buf * BM * LSTR;
        half_t* b = sB + buf * BN * LSTR;
#pragma unroll
        for (int v = 0; v < AV; ++v) st16(a + (lrow + 32 * v) * LSTR + lcol, ra[v]);
#pragma unroll
        for (int v = 0; v < BV; ++v) st16(b + (lrow + 32 * v) * LSTR + lcol, rb[v]);
    };

    f16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (int)((p.K + BK - 1) / BK);
    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = (kt + 1) < nk;
        if (more) gload(kt + 1);

        const half_t* a = sA + buf * BM * LSTR + (wr * WM + l31) * LSTR + hi * 8;
        const half_t* b = sB + buf * BN * LSTR + (wc * WN + l31) * LSTR + hi * 8;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            h8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const h8*>(a + i * 32 * LSTR + ks * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const h8*>(b + j * 32 * LSTR + ks * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    // lane holds, for tile (i,j): column n = l31, rows (r&3) + 8*(r>>2) + 4*hi, r = 0..15
    half_t* Cb = p.C + z0 * p.c_bs0 + z1 * p.c_bs1;
    const half_t* Rb = p.residual ? p.residual + z0 * p.r_bs0 + z1 * p.r_bs1 : nullptr;

    if (p.geglu) {
        if constexpr (TN == 2) {
            const long n = n0 + wc * 32 + l31;  // output column
            if (n < p.N) {
                const float bh = p.bias ? (float)p.bias[n] : 0.f;
                const float bg = p.bias ? (float)p.bias[p.N + n] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long m = m0 + wr * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (m < p.M) {
                            const float hval = acc[i][0][r] * p.alpha + bh;
                            const float gval = acc[i][1][r] * p.alpha + bg;
                            float o = hval * gelu_erf_f(gval);
                            if (Rb) o += (float)Rb[m * p.ldr + n];
                            Cb[m * p.ldc + n] = (half_t)o;
                        }
                    }
                }
            }
        }
        return;
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long n = n0 + wc * WN + j * 32 + l31;
        const bool nok = n < p.N;
        const float bv = (nok && p.bias) ? (float)p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long mbase = m0 + wr * WM + i * 32 + 4 * hi;
            if (p.c_mode == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long m = mbase + (r & 3) + 8 * (r >> 2);
                    if (nok && m < p.M) {
                        float o = acc[i][j][r] * p.alpha + bv;
                        if (p.rowvec) o += (float)p.rowvec[(m / p.rows_per_vec) * p.N + n];
                        if (Rb) o += (float)Rb[m * p.ldr + n];
                        Cb[m * p.ldc + n] = (half_t)o;
                    }
                }
            } else {
                // transposed store: C[img][n][m % rows]
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const long mg = mbase + 8 * g;  // 4 consecutive rows mg..mg+3
                    if (!nok || mg >= p.M) continue;
                    float o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = acc[i][j][4 * g + q] * p.alpha + bv;
                    if (p.c_pack4 && mg + 3 < p.M) {
                        const long img = mg / p.c_rows_per_img;
                        const long mm = mg - img * p.c_rows_per_img;
                        h4 pk;
                        pk[0] = (half_t)o[0]; pk[1] = (half_t)o[1];
                        pk[2] = (half_t)o[2]; pk[3] = (half_t)o[3];
                        *reinterpret_cast<h4*>(Cb + img * p.c_img_stride + n * p.ldc + mm) = pk;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const long m = mg + q;
                            if (m < p.M) {
                                const long img = m / p.c_rows_per_img;
                                const long mm = m - img * p.c_rows_per_img;
                                Cb[img * p.c_img_stride + n * p.ldc + mm] = (half_t)o[q];
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int BM, int BN>
int launch(const GemmParams& p, long tiles_m, long nbatch, hipStream_t stream) {
    constexpr size_t smem = 2 * (BM + BN) * LSTR * sizeof(half_t);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid((unsigned)(tiles_m * p.tiles_n), 1, (unsigned)nbatch);
    hipLaunchKernelGGL((gemm_kernel<BM, BN>), grid, dim3(256), smem, stream, p);
    return vsx_check_launch("vsx_gemm_f16");
}

// ---- instrumentation (bench.py roofline): hipEvent pairs around sampled launches ----
struct ProfState {
    bool on = false;
    long max_samples = 0;
    long n = 0;
    double flop = 0.0;
    std::vector<hipEvent_t>* ev = nullptr;  // 2 per sample
};

ProfState g_prof;

}  // namespace

extern "C" int vsx_prof_enable(int64_t on, int64_t max_samples) {
    if (!g_prof.ev) g_prof.ev = new std::vector<hipEvent_t>();
    g_prof.on = on != 0;
    g_prof.max_samples = max_samples;
    g_prof.n = 0;
    g_prof.flop = 0.0;
    return VSX_OK;
}

extern "C" int vsx_prof_collect(int64_t* n_launches, double* total_ms, double* total_flop) {
    double ms = 0.0;
    if (g_prof.ev) {
        for (long i = 0; i < g_prof.n; ++i) {
            hipEvent_t a = (*g_prof.ev)[2 * i], b = (*g_prof.ev)[2 * i + 1];
            if (hipEventSynchronize(b) != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "prof: event sync failed");
            float t = 0.f;
            if (hipEventElapsedTime(&t, a, b) != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "prof: elapsed failed");
            ms += t;
        }
    }
    if (n_launches) *n_launches = g_prof.n;
    if (total_ms) *total_ms = ms;
    if (total_flop) *total_flop = g_prof.flop;
    g_prof.n = 0;
    g_prof.flop = 0.0;
    return VSX_OK;
}

extern "C" int vsx_gemm_f16(const vsx_gemm_desc* d, vsx_stream_t stream_) {
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
    VSX_REQUIRE(nbatch <= 65535, VSX_E_BADSHAPE, "gemm: batch0*batch1 = %ld exceeds 65535", nbatch);

    if (p.a_mode == 0) {
        VSX_REQUIRE(d->lda % 8 == 0, VSX_E_BADSHAPE, "gemm: lda (%ld) must be a multiple of 8", (long)d->lda);
    } else if (p.a_mode == 1) {
        VSX_REQUIRE(nbatch == 1, VSX_E_UNSUPPORTED, "gemm: conv mode does not take a batch");
        VSX_REQUIRE(d->ks == 1 || d->ks == 3, VSX_E_UNSUPPORTED, "gemm: conv kernel size %ld", (long)d->ks);
        VSX_REQUIRE(d->stride == 1 || d->stride == 2, VSX_E_UNSUPPORTED, "gemm: conv stride %ld", (long)d->stride);
        VSX_REQUIRE(d->C1 > 0 && d->C1 % 8 == 0 && d->C2 >= 0 && d->C2 % 8 == 0, VSX_E_BADSHAPE,
                    "gemm: conv channels must be multiples of 8 (C1=%ld C2=%ld)", (long)d->C1, (long)d->C2);
        VSX_REQUIRE((d->C2 == 0) == (d->A2 == nullptr), VSX_E_BADSHAPE, "gemm: A2/C2 mismatch");
        VSX_REQUIRE(d->A2 == nullptr || vsx_aligned16(d->A2), VSX_E_BADSHAPE, "gemm: A2 must be 16-byte aligned");
        VSX_REQUIRE(d->K == d->ks * d->ks * (d->C1 + d->C2), VSX_E_BADSHAPE, "gemm: conv K mismatch");
        VSX_REQUIRE(d->H > 0 && d->W > 0, VSX_E_BADSHAPE, "gemm: conv H/W");
        VSX_REQUIRE(!d->upsample || (d->H % 2 == 0 && d->W % 2 == 0), VSX_E_BADSHAPE, "gemm: upsample needs even H/W");
        p.H = (int)d->H; p.W = (int)d->W; p.C1 = (int)d->C1; p.C2 = (int)d->C2;
        p.ks = (int)d->ks; p.stride = (int)d->stride; p.ups = d->upsample ? 1 : 0;
        const int pad = p.ks / 2;
        p.Ho = (p.H + 2 * pad - p.ks) / p.stride + 1;
        p.Wo = (p.W + 2 * pad - p.ks) / p.stride + 1;
        VSX_REQUIRE(d->M % ((long)p.Ho * p.Wo) == 0, VSX_E_BADSHAPE, "gemm: conv M (%ld) not a multiple of Ho*Wo (%d*%d)",
                    (long)d->M, p.Ho, p.Wo);
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

    // tile selection: big tiles when they still fill the 256 CUs (2 workgroups per CU)
    const long cols = p.geglu ? 2 * d->N : d->N;
    const long big = ((d->M + 127) / 128) * ((cols + 127) / 128) * nbatch;
    int rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool sample = g_prof.on && g_prof.n < g_prof.max_samples;
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
        hipEventRecord(e0, stream);
    }
    if (big >= 512 || p.geglu) {
        const bool tall = big >= 512;
        if (tall) {
            p.tiles_n = (int)((cols + 127) / 128);
            rc = launch<128, 128>(p, (d->M + 127) / 128, nbatch, stream);
        } else {
            p.tiles_n = (int)((cols + 127) / 128);
            rc = launch<64, 128>(p, (d->M + 63) / 64, nbatch, stream);
        }
    } else {
        p.tiles_n = (int)((cols + 63) / 64);
        rc = launch<64, 64>(p, (d->M + 63) / 64, nbatch, stream);
    }
    if (sample) {
        hipEventRecord(e1, stream);
        g_prof.n += 1;
        g_prof.flop += 2.0 * (double)d->M * (double)cols * (double)d->K * (double)nbatch;
    }
    return rc;
}
