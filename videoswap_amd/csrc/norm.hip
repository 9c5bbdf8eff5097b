// K3/K4 — GroupNorm (two-kernel, channels-last, optional two-source concat) and LayerNorm
// (+ fused temporal positional encoding) for gfx950.  HBM-bound: 16-byte vector loads/stores,
// fp32 statistics, deterministic reduction order (no atomics).
#include "common.h"

namespace vsxg {
long gemm_option(const char* name);      // gemm.hip: the option table of vsx_set_option
}

namespace {

constexpr int GN_MAX_THREADS = 512;
constexpr int GN_MAX_GROUPS = 64;

// Thread layout of the statistics and the apply kernel: a row is vpr = C / 8 16-byte vectors, a workgroup has
// rp = 512 / vpr (>= 1) rows in flight per pass and rp * vpr threads (480 for C = 320 ... 1920, 320 for C = 2560), so that a
// thread keeps ONE column octet for its whole life: thread t owns vector t % vpr of rows t / vpr, t / vpr + rp, ...  The per-channel
// constants of the apply kernel (scale / shift) therefore live in 16 registers per thread instead of an LDS table that every
// vector read back (64 B of LDS reads per 16 B of HBM), and neither kernel divides anything inside its streaming loop.
__host__ __device__ inline int gn_threads(int C) {
    const int vpr = C >> 3;
    return (GN_MAX_THREADS / vpr) * vpr;
}

// Rows per statistics chunk / per apply workgroup: ~3 (statistics) and ~4 (apply) workgroups per CU whatever the split between
// images and rows (5-D GroupNorm: 2 images x 65536 rows; per-frame GroupNorm: 32 images x 4096 rows; 8x8 level: 1-4 K rows).
// Round 4 launched 2048 workgroups of 64 rows: a 41-KB stream each, behind which the block reduction (statistics) or the
// table build (apply) took about as long as the stream itself — 1.7 and 3.0 TB/s over a forward against the 5 TB/s of a copy.
__host__ __device__ inline int gn_stat_rows(long rows, long nimg) {
    long r = rows * nimg / 768;
    return (int)(r < 8 ? 8 : (r > 1024 ? 1024 : r));
}
__host__ __device__ inline int gn_apply_rows(long rows, long nimg) {
    long r = rows * nimg / 1024;
    return (int)(r < 4 ? 4 : (r > 512 ? 512 : r));
}

// Per-thread partial sums of the statistics kernel in LDS: a thread owns 8 consecutive channels.  Scalar writes in channel order sit
// at a pitch of 8 floats and use an eighth of the banks; channel c = 8 v + e is kept at (e / 4) * C / 2 + 4 v + e % 4 instead, so
// that a thread writes two 16-byte quads and the low quads of all lanes are contiguous.
__device__ __forceinline__ int gn_col(const int c, const int C) { return ((c >> 2) & 1) * (C >> 1) + (c >> 3) * 4 + (c & 3); }

__device__ __forceinline__ const half_t* gn_src(const half_t* x1, const half_t* x2, int c, int C1, int C2, long& pitch) {
    pitch = c < C1 ? C1 : C2;                   // a thread's column octet lies in one of the two concatenated sources
    return c < C1 ? x1 + c : x2 + (c - C1);
}

// grid (nchunks, nimg), block gn_threads(C).
__global__ __launch_bounds__(GN_MAX_THREADS) void gn_stats_kernel(const half_t* __restrict__ x1,
                                                                  const half_t* __restrict__ x2, long rows, int C1,
                                                                  int C2, int groups, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float red_s[4096];
    __shared__ __attribute__((aligned(16))) float red_q[4096];
    const int C = C1 + C2;
    const int vpr = C >> 3;                   // vectors per row (<= 512)
    const int rp = (int)blockDim.x / vpr;     // rows per pass (>= 1): every thread of the block has a row
    const int tid = threadIdx.x;
    const int rl = tid / vpr;
    const int cv = tid - rl * vpr;
    const int chunk = blockIdx.x;
    const long img = blockIdx.y;
    const int rpc = gn_stat_rows(rows, gridDim.y);
    const long r0 = (long)chunk * rpc;
    const long r1 = min(r0 + (long)rpc, rows);
    long pitch;
    const half_t* src = gn_src(x1, x2, cv * 8, C1, C2, pitch) + img * rows * pitch;

    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    // eight row loads in flight per thread
    for (long rb = r0 + rl; rb < r1; rb += 8 * rp) {
        uint4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long r = rb + u * rp;
            raw[u] = r < r1 ? ld16(src + r * pitch) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const h8 v = as_h8(raw[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s[e] += f;
                q[e] += f * f;
            }
        }
    }
    // a thread's 8 channels go out as two 16-byte writes, the low quads of a row's threads contiguous, then the high quads
    *reinterpret_cast<f4v*>(red_s + rl * C + cv * 4) = f4v{s[0], s[1], s[2], s[3]};
    *reinterpret_cast<f4v*>(red_s + rl * C + (C >> 1) + cv * 4) = f4v{s[4], s[5], s[6], s[7]};
    *reinterpret_cast<f4v*>(red_q + rl * C + cv * 4) = f4v{q[0], q[1], q[2], q[3]};
    *reinterpret_cast<f4v*>(red_q + rl * C + (C >> 1) + cv * 4) = f4v{q[4], q[5], q[6], q[7]};
    __syncthreads();
    for (int c = tid; c < C; c += (int)blockDim.x) {
        const int col = gn_col(c, C);           // the column sum stays in its (permuted) column of row 0
        float a = 0.f, b = 0.f;
        for (int r = 0; r < rp; ++r) { a += red_s[r * C + col]; b += red_q[r * C + col]; }
        red_s[col] = a;
        red_q[col] = b;
    }
    __syncthreads();
    if (tid < groups) {
        const int cpg = C / groups;
        float a = 0.f, b = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += red_s[gn_col(c, C)]; b += red_q[gn_col(c, C)]; }
        float* out = partial + ((img * gridDim.x + chunk) * groups + tid) * 2;
        out[0] = a;
        out[1] = b;
    }
}

// mean / rstd from the two sums of an (image, group): ONE spelling for the stand-alone finalize kernel and for the apply kernel's fused
// prologue (explicit fma: the two must round alike whatever the compiler would contract)
__device__ __forceinline__ void gn_mean_rstd(const float sum, const float sumsq, const float inv_count, const float eps, float& mean,
                                             float& rstd) {
    mean = sum * inv_count;
    const float var = fmaxf(__builtin_fmaf(-mean, mean, sumsq * inv_count), 0.f);
    rstd = rsqrtf(var + eps);
}

// The apply kernel can compute the statistics itself when an image has at most this many chunks (the per-frame GroupNorms of the
// transformer / motion-module entries: 13 - 49 chunks per image): one launch fewer per GroupNorm, 36 of 81 per UNet forward (option
// gn_fuse; measured, not the default: vsx_groupnorm_apply).
constexpr int GN_FUSE_MAX_CHUNKS = 64;

// grid (groups, nimg), block 256: mean / rstd of one (image, group) from its per-chunk partial sums.  The 5-D
// GroupNorm has ~1000 chunks per image: every thread issues its (independent) loads back to back and the combine is
// a fixed-shape tree, so the kernel is a few microseconds and the result is deterministic (identical on every rank).
// (The first version walked the chunks serially in 8 threads per group: 13 us per call, 81 calls per forward.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nchunks, int groups,
                                                          float inv_count, float eps, float* __restrict__ stats) {
    __shared__ float s_a[256];
    __shared__ float s_b[256];
    const int tid = threadIdx.x;
    const int g = blockIdx.x;
    const long img = blockIdx.y;
    const float2* pp = reinterpret_cast<const float2*>(partial) + img * nchunks * groups + g;
    float a = 0.f, b = 0.f;
#pragma unroll 8
    for (int ch = tid; ch < nchunks; ch += 256) {
        const float2 v = pp[(long)ch * groups];
        a += v.x;
        b += v.y;
    }
    s_a[tid] = a;
    s_b[tid] = b;
    __syncthreads();
#pragma unroll
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) { s_a[tid] += s_a[tid + w]; s_b[tid] += s_b[tid + w]; }
        __syncthreads();
    }
    if (tid == 0) {
        float mean, rstd;
        gn_mean_rstd(s_a[0], s_b[0], inv_count, eps, mean, rstd);
        stats[(img * groups + g) * 2] = mean;
        stats[(img * groups + g) * 2 + 1] = rstd;
    }
}

// grid (ceil(rows / gn_apply_rows(rows)), nimg), block gn_threads(C).  Per-channel scale / shift (a = rstd * gamma,
// b = beta - mean * a) of the thread's own eight channels are built once, in registers, so the streaming loop is one FMA
// (+ SiLU) per element and touches no LDS (the first version recomputed the group index with an integer division per element
// and was VALU-bound; the second kept the constants in an LDS table and read 64 bytes of it per 16-byte vector).
// FUSED (nchunks <= GN_FUSE_MAX_CHUNKS, no finalize launch): `partial` holds the image's per-chunk sums [nchunks][groups][2]; every
// workgroup reduces them itself — wave w takes groups w, w + nwaves, ..: lane l holds chunk l (zero past nchunks) and the wave adds
// with partners l + 32, l + 16, .., l + 1, which is the tree of gn_finalize_kernel restricted to its first 64 slots (the slots past
// the chunk count hold zeros there, and x + 0 is x): the same sums in the same order, the same gn_mean_rstd — bit-identical
// statistics — and the workgroups with blockIdx.x == 0 still write them to `stats_out` (the C ABI hands them to the caller).
template <bool FUSED>
__global__ __launch_bounds__(GN_MAX_THREADS) void gn_apply_kernel(const half_t* __restrict__ x1, const half_t* __restrict__ x2,
                                                                  long rows, int C1, int C2, int groups,
                                                                  const float* __restrict__ stats,
                                                                  const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                                  int silu, half_t* __restrict__ y, const float* __restrict__ partial,
                                                                  int nchunks, float inv_count, float eps, float* __restrict__ stats_out) {
    const int C = C1 + C2;
    const int cpg = C / groups;
    const int vpr = C >> 3;
    const int rp = (int)blockDim.x / vpr;
    const int tid = threadIdx.x;
    const int rl = tid / vpr;
    const int c0 = (tid - rl * vpr) * 8;
    const long img = blockIdx.y;
    __shared__ float st_sh[FUSED ? 2 * GN_MAX_GROUPS : 2];
    if constexpr (FUSED) {
        const int lane = tid & 63, wave = tid >> 6;
        const int nwaves = (int)blockDim.x >> 6;                    // full waves only (480 threads: 7; the half wave idles here)
        if (wave < nwaves) {
            const float2* pp = reinterpret_cast<const float2*>(partial) + img * nchunks * groups;
            for (int g = wave; g < groups; g += nwaves) {
                float2 v = lane < nchunks ? pp[(long)lane * groups + g] : make_float2(0.f, 0.f);
#pragma unroll
                for (int w = 32; w > 0; w >>= 1) {
                    v.x += __shfl_down(v.x, w, 64);
                    v.y += __shfl_down(v.y, w, 64);
                }
                if (lane == 0) {
                    float mean, rstd;
                    gn_mean_rstd(v.x, v.y, inv_count, eps, mean, rstd);
                    st_sh[2 * g] = mean;
                    st_sh[2 * g + 1] = rstd;
                    if (blockIdx.x == 0) {
                        stats_out[(img * groups + g) * 2] = mean;
                        stats_out[(img * groups + g) * 2 + 1] = rstd;
                    }
                }
            }
        }
        __syncthreads();
    }
    float ca[8], cb[8];
    {
        const h8 gm = as_h8(ld16(gamma + c0)), bt = as_h8(ld16(beta + c0));
        const float* st = FUSED ? st_sh : stats + img * groups * 2;
        int g = c0 / cpg, left = (g + 1) * cpg - c0;      // channels of group g from c0 on
        float mean = st[2 * g], rstd = st[2 * g + 1];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (left == 0) { ++g; left = cpg; mean = st[2 * g]; rstd = st[2 * g + 1]; }
            --left;
            ca[e] = rstd * (float)gm[e];
            cb[e] = (float)bt[e] - mean * ca[e];
        }
    }
    long pitch;
    const half_t* src = gn_src(x1, x2, c0, C1, C2, pitch) + img * rows * pitch;
    half_t* dst = y + img * rows * C + c0;
    const int rpb = gn_apply_rows(rows, gridDim.y);
    const long r0 = (long)blockIdx.x * rpb;
    const long r1 = min(r0 + (long)rpb, rows);
    // four vectors in flight per thread: all loads of a batch are issued before the first is used
    for (long rb = r0 + rl; rb < r1; rb += 4 * rp) {
        uint4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = rb + u * rp;
            raw[u] = r < r1 ? ld16(src + r * pitch) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long r = rb + u * rp;
            if (r >= r1) break;
            const h8 v = as_h8(raw[u]);
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = __builtin_fmaf((float)v[e], ca[e], cb[e]);
                if (silu) f = silu_f(f);
                o[e] = (half_t)f;
            }
            st16(dst + r * C, as_u4(o));
        }
    }
}

// A wave normalises R rows of NV * 64 16-byte vectors at a time (R * NV = 4: four rows of C <= 512, two of C <= 1024,
// one of C <= 2048), 4 waves per block.  All of a wave's row loads are issued before the first reduction, so each wave
// keeps 4 x 1 KiB in flight: the first version (one row per wave, a single 640-byte request in flight at C = 320) ran
// at 3 TB/s against the 5-6 TB/s of a plain device copy of the same tensor.  Per-row arithmetic is unchanged.
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, long M, int C,
                                                        const half_t* __restrict__ gamma,
                                                        const half_t* __restrict__ beta, float eps,
                                                        const half_t* __restrict__ pe, long rows_per_frame, int frames,
                                                        int frame_offset, half_t* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (m0 >= M) return;
    const int vpr = C >> 3;
    uint4 raw[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int v = lane + 64 * k;
            raw[r][k] = (v < vpr && m0 + r < M) ? ld16(x + (m0 + r) * C + v * 8) : make_uint4(0, 0, 0, 0);
        }
    h8 gm[NV], bt[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int v = lane + 64 * k;
        if (v < vpr) {
            gm[k] = as_h8(ld16(gamma + v * 8));
            bt[k] = as_h8(ld16(beta + v * 8));
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long m = m0 + r;
        if (m >= M) break;
        float f[NV][8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const h8 h = as_h8(raw[r][k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[k][e] = (float)h[e]; sum += f[k][e]; }
        }
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (lane + 64 * k < vpr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[k][e] - mean; sq += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
        const half_t* perow = nullptr;
        if (pe) {
            const long fr = (m / rows_per_frame) % frames + frame_offset;
            perow = pe + fr * C;
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int v = lane + 64 * k;
            if (v < vpr) {
                h8 o;
                if (perow) {
                    const h8 pv = as_h8(ld16(perow + v * 8));
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        o[e] = (half_t)((f[k][e] - mean) * rstd * (float)gm[k][e] + (float)bt[k][e] + (float)pv[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        o[e] = (half_t)((f[k][e] - mean) * rstd * (float)gm[k][e] + (float)bt[k][e]);
                }
                st16(y + m * C + v * 8, as_u4(o));
            }
        }
    }
}

// Row statistics only (LayerNorm folded into its consumer GEMM): same loads and the same fp32 arithmetic as
// layernorm_kernel, no normalised output — stats[m] = (rstd, -rstd * mean).
template <int NV, int R>
__global__ __launch_bounds__(256) void row_stats_kernel(const half_t* __restrict__ x, long M, int C, float eps,
                                                        float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (m0 >= M) return;
    const int vpr = C >> 3;
    uint4 raw[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int v = lane + 64 * k;
            raw[r][k] = (v < vpr && m0 + r < M) ? ld16(x + (m0 + r) * C + v * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long m = m0 + r;
        if (m >= M) break;
        float f[NV][8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const h8 h = as_h8(raw[r][k]);
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[k][e] = (float)h[e]; sum += f[k][e]; }
        }
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (lane + 64 * k < vpr) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[k][e] - mean; sq += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
        if (lane == 0) {
            stats[2 * m] = rstd;
            stats[2 * m + 1] = -rstd * mean;
        }
    }
}

// Row statistics from the partial sums a producing GEMM's epilogue wrote (vsx_gemm_desc.rowstats): one thread per row adds its
// nparts (sum, sum of squares) pairs in index order — 8 * nparts contiguous bytes — and writes (rstd, -rstd * mean).
__global__ __launch_bounds__(256) void row_stats_combine_kernel(const float* __restrict__ parts, long M, int nparts, float inv_c,
                                                                float eps, float* __restrict__ stats) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float2* pp = reinterpret_cast<const float2*>(parts) + m * nparts;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < nparts; ++i) {
        const float2 v = pp[i];
        s1 += v.x;
        s2 += v.y;
    }
    const float mean = s1 * inv_c;
    const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv_c), 0.f);
    const float rstd = rsqrtf(var + eps);
    stats[2 * m] = rstd;
    stats[2 * m + 1] = -rstd * mean;
}

// in-place row softmax, one wave per row (fp16 storage, fp32 math).
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* __restrict__ S, long nrows, int ncols, long ld) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    half_t* row = S + r * ld;
    float mx = -INFINITY;
    for (int c = lane; c < ncols; c += 64) mx = fmaxf(mx, (float)row[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < ncols; c += 64) sum += __expf((float)row[c] - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int c = lane; c < ncols; c += 64) row[c] = (half_t)(__expf((float)row[c] - mx) * inv);
}

// causal variant (CLIP text encoder: token q attends to tokens <= q): row r belongs to query q = r % rows_per_seq;
// columns > q get probability 0
__global__ __launch_bounds__(256) void softmax_rows_causal_kernel(half_t* __restrict__ S, long nrows, int ncols,
                                                                  long ld, int rows_per_seq) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    half_t* row = S + r * ld;
    const int live = min(ncols, (int)(r % rows_per_seq) + 1);
    float mx = -INFINITY;
    for (int c = lane; c < live; c += 64) mx = fmaxf(mx, (float)row[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < live; c += 64) sum += __expf((float)row[c] - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int c = lane; c < ncols; c += 64) row[c] = c < live ? (half_t)(__expf((float)row[c] - mx) * inv) : (half_t)0.f;
}

}  // namespace

extern "C" int64_t vsx_groupnorm_chunks(int64_t rows, int64_t nimg) {
    const int rpc = gn_stat_rows(rows, nimg);
    return (rows + rpc - 1) / rpc;
}

static int gn_check(const void* x1, const void* x2, int64_t nimg, int64_t rows, int64_t C1, int64_t C2,
                    int64_t groups) {
    VSX_REQUIRE(x1 && nimg > 0 && rows > 0, VSX_E_BADSHAPE, "groupnorm: empty input");
    VSX_REQUIRE(C1 > 0 && C1 % 8 == 0 && C2 >= 0 && C2 % 8 == 0, VSX_E_BADSHAPE,
                "groupnorm: channels must be multiples of 8 (C1=%ld C2=%ld)", (long)C1, (long)C2);
    VSX_REQUIRE((C2 == 0) == (x2 == nullptr), VSX_E_BADSHAPE, "groupnorm: x2/C2 mismatch");
    const int64_t C = C1 + C2;
    VSX_REQUIRE(C <= 4096, VSX_E_UNSUPPORTED, "groupnorm: C=%ld > 4096", (long)C);
    VSX_REQUIRE(groups > 0 && groups <= GN_MAX_GROUPS && C % groups == 0, VSX_E_BADSHAPE,
                "groupnorm: groups=%ld does not divide C=%ld (max %d groups)", (long)groups, (long)C, GN_MAX_GROUPS);
    VSX_REQUIRE(vsx_aligned16(x1) && vsx_aligned16(x2), VSX_E_BADSHAPE, "groupnorm: inputs must be 16-byte aligned");
    VSX_REQUIRE(nimg <= 65535, VSX_E_BADSHAPE, "groupnorm: nimg=%ld > 65535", (long)nimg);
    return VSX_OK;
}

extern "C" int vsx_groupnorm_stats(const void* x1, const void* x2, int64_t nimg, int64_t rows, int64_t C1,
                                   int64_t C2, int64_t groups, float* partial, vsx_stream_t stream) {
    int rc = gn_check(x1, x2, nimg, rows, C1, C2, groups);
    if (rc) return rc;
    VSX_REQUIRE(partial != nullptr, VSX_E_WORKSPACE, "groupnorm_stats: null partial buffer");
    dim3 grid((unsigned)vsx_groupnorm_chunks(rows, nimg), (unsigned)nimg);
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3((unsigned)gn_threads((int)(C1 + C2))), 0, (hipStream_t)stream, (const half_t*)x1,
                       (const half_t*)x2, (long)rows, (int)C1, (int)C2, (int)groups, partial);
    return vsx_check_launch("vsx_groupnorm_stats");
}

extern "C" int vsx_groupnorm_apply(const void* x1, const void* x2, int64_t nimg, int64_t rows, int64_t C1,
                                   int64_t C2, int64_t groups, const float* partial, int64_t nchunks,
                                   int64_t count_rows, const void* gamma, const void* beta, float eps, int64_t silu,
                                   float* stats, void* y, vsx_stream_t stream) {
    int rc = gn_check(x1, x2, nimg, rows, C1, C2, groups);
    if (rc) return rc;
    VSX_REQUIRE(partial && gamma && beta && y && stats, VSX_E_BADSHAPE, "groupnorm_apply: null argument");
    VSX_REQUIRE(nchunks > 0 && count_rows > 0, VSX_E_BADSHAPE, "groupnorm_apply: nchunks/count_rows");
    VSX_REQUIRE(vsx_aligned16(gamma) && vsx_aligned16(beta) && vsx_aligned16(y), VSX_E_BADSHAPE,
                "groupnorm_apply: gamma/beta/y must be 16-byte aligned");
    const int64_t C = C1 + C2;
    const float inv_count = 1.0f / ((float)count_rows * (float)(C / groups));
    const int rpb = gn_apply_rows(rows, nimg);
    dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)nimg);
    const unsigned threads = (unsigned)gn_threads((int)C);
    // few chunks per image (per-frame GroupNorm): the apply kernel can finalize the statistics itself — one launch fewer; option
    // "gn_fuse" / VSX_GN_FUSE = 1.  Built and measured in round 6 and NOT the default: bit-identical, but 2 % SLOWER over the 60 per-frame
    // GroupNorms of a forward pair (0.874 -> 0.894 ms, profiles/r06_groupnorm_fused_finalize_ab.txt) — a 4-us finalize launch between two
    // kernels costs less than the same reduction on the critical path of every apply workgroup (back-to-back launches overlap their
    // ramp-up and drain; a prologue does not)
    if (nchunks <= GN_FUSE_MAX_CHUNKS && threads >= 64 && vsxg::gemm_option("gn_fuse") != 0) {
        hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(threads), 0, (hipStream_t)stream, (const half_t*)x1, (const half_t*)x2,
                           (long)rows, (int)C1, (int)C2, (int)groups, (const float*)nullptr, (const half_t*)gamma, (const half_t*)beta,
                           (int)silu, (half_t*)y, partial, (int)nchunks, inv_count, eps, stats);
        return vsx_check_launch("vsx_groupnorm_apply");
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)groups, (unsigned)nimg), dim3(256), 0, (hipStream_t)stream, partial,
                       (int)nchunks, (int)groups, inv_count, eps, stats);
    hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(threads), 0, (hipStream_t)stream, (const half_t*)x1, (const half_t*)x2,
                       (long)rows, (int)C1, (int)C2, (int)groups, stats, (const half_t*)gamma, (const half_t*)beta,
                       (int)silu, (half_t*)y, (const float*)nullptr, 0, 0.f, 0.f, (float*)nullptr);
    return vsx_check_launch("vsx_groupnorm_apply");
}

extern "C" int vsx_layernorm(const void* x, int64_t M, int64_t C, const void* gamma, const void* beta, float eps,
                             const void* pe, int64_t rows_per_frame, int64_t frames, int64_t frame_offset, void* y,
                             vsx_stream_t stream) {
    VSX_REQUIRE(x && gamma && beta && y, VSX_E_BADSHAPE, "layernorm: null argument");
    if (M == 0) return VSX_OK;
    VSX_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, VSX_E_BADSHAPE, "layernorm: C=%ld must be a multiple of 8, <= 2048",
                (long)C);
    VSX_REQUIRE(vsx_aligned16(x) && vsx_aligned16(gamma) && vsx_aligned16(beta) && vsx_aligned16(y) && vsx_aligned16(pe),
                VSX_E_BADSHAPE, "layernorm: pointers must be 16-byte aligned");
    if (pe) VSX_REQUIRE(rows_per_frame > 0 && frames > 0 && frame_offset >= 0, VSX_E_BADSHAPE, "layernorm: pe geometry");
    const long rpf = rows_per_frame > 0 ? rows_per_frame : 1;
    const int nfr = (int)(frames > 0 ? frames : 1);
#define VSX_LN_LAUNCH(NV, R)                                                                                            \
    hipLaunchKernelGGL((layernorm_kernel<NV, R>), dim3((unsigned)((M + 4 * (R) - 1) / (4 * (R)))), dim3(256), 0,            \
                       (hipStream_t)stream, (const half_t*)x, (long)M, (int)C, (const half_t*)gamma,                    \
                       (const half_t*)beta, eps, (const half_t*)pe, rpf, nfr, (int)frame_offset, (half_t*)y)
    if (C <= 512) VSX_LN_LAUNCH(1, 4);
    else if (C <= 1024) VSX_LN_LAUNCH(2, 2);
    else VSX_LN_LAUNCH(4, 1);
#undef VSX_LN_LAUNCH
    return vsx_check_launch("vsx_layernorm");
}

extern "C" int vsx_row_stats(const void* x, int64_t M, int64_t C, float eps, float* stats, vsx_stream_t stream) {
    VSX_REQUIRE(x && stats, VSX_E_BADSHAPE, "row_stats: null argument");
    if (M == 0) return VSX_OK;
    VSX_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, VSX_E_BADSHAPE, "row_stats: C=%ld must be a multiple of 8, <= 2048",
                (long)C);
    VSX_REQUIRE(vsx_aligned16(x), VSX_E_BADSHAPE, "row_stats: x must be 16-byte aligned");
#define VSX_RS_LAUNCH(NV, R)                                                                                            \
    hipLaunchKernelGGL((row_stats_kernel<NV, R>), dim3((unsigned)((M + 4 * (R) - 1) / (4 * (R)))), dim3(256), 0,            \
                       (hipStream_t)stream, (const half_t*)x, (long)M, (int)C, eps, stats)
    if (C <= 512) VSX_RS_LAUNCH(1, 4);
    else if (C <= 1024) VSX_RS_LAUNCH(2, 2);
    else VSX_RS_LAUNCH(4, 1);
#undef VSX_RS_LAUNCH
    return vsx_check_launch("vsx_row_stats");
}

extern "C" int vsx_row_stats_combine(const float* parts, int64_t M, int64_t nparts, int64_t C, float eps, float* stats,
                                     vsx_stream_t stream) {
    VSX_REQUIRE(parts && stats, VSX_E_BADSHAPE, "row_stats_combine: null argument");
    if (M == 0) return VSX_OK;
    VSX_REQUIRE(M > 0 && nparts > 0 && nparts <= 64 && C > 0, VSX_E_BADSHAPE, "row_stats_combine: M=%ld nparts=%ld C=%ld", (long)M,
                (long)nparts, (long)C);
    VSX_REQUIRE((((uintptr_t)parts) & 7) == 0 && (((uintptr_t)stats) & 7) == 0, VSX_E_BADSHAPE, "row_stats_combine: 8-byte alignment");
    hipLaunchKernelGGL(row_stats_combine_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, parts, (long)M,
                       (int)nparts, 1.0f / (float)C, eps, stats);
    return vsx_check_launch("vsx_row_stats_combine");
}

extern "C" int vsx_softmax_rows(void* S, int64_t nrows, int64_t ncols, int64_t ld, vsx_stream_t stream) {
    VSX_REQUIRE(S != nullptr && ncols > 0 && ld >= ncols, VSX_E_BADSHAPE, "softmax_rows: bad arguments");
    if (nrows == 0) return VSX_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)S, (long)nrows, (int)ncols, (long)ld);
    return vsx_check_launch("vsx_softmax_rows");
}

extern "C" int vsx_softmax_rows_causal(void* S, int64_t nrows, int64_t ncols, int64_t ld, int64_t rows_per_seq,
                                       vsx_stream_t stream) {
    VSX_REQUIRE(S != nullptr && ncols > 0 && ld >= ncols && rows_per_seq > 0, VSX_E_BADSHAPE,
                "softmax_rows_causal: bad arguments");
    if (nrows == 0) return VSX_OK;
    hipLaunchKernelGGL(softmax_rows_causal_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)S, (long)nrows, (int)ncols, (long)ld, (int)rows_per_seq);
    return vsx_check_launch("vsx_softmax_rows_causal");
}
