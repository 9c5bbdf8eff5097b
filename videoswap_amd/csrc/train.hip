// The backward kernels of the adapter training step (trainer_videoswap.py:33-97).  First run on an
// MI355X in round 3 (gpurun_out r03a: tests/test_autograd.py + tests/test_training.py green, 329 ms per training step at
// 16 x 512^2), then moved from the development library into libvsx.so / include/vsx.h (ABI 5).  The gradient reaches the adapter through the frozen UNet, so these are DATA gradients:
//
//   vsx_geglu_fwd / vsx_geglu_bwd   GEGLU as its own pass over the saved pre-activations (the GEMM epilogue fuses it in
//                                   inference and drops them)
//   vsx_silu_bwd                    adapter MLP activation
//   vsx_groupnorm_bwd               GroupNorm(+SiLU) over [nimg, rows, C1 (+C2 concatenated)], statistics recomputed
//   vsx_layernorm_bwd               one wave per row
//   vsx_softmax_bwd                 dS = scale * P o (dP - rowsum(dP o P)) on the row-padded score buffers
//   vsx_sum_pool2x2                 gradient of the nearest-2x upsampling folded into a conv's loader
//   vsx_adapter_gather              gradient of vsx_adapter_scatter with respect to the point features
//
// The matrix work of the backward pass (linear / conv dgrad, the attention products) runs on vsx_gemm_f16 with
// transposed / flipped weight copies (videoswap_amd/autograd.py).  All kernels here are HBM-bound streaming passes:
// 16-byte accesses, fp32 arithmetic and fp32 reductions in a fixed order (no atomics: results are deterministic).
// The GroupNorm backward is a chunked two-level reduction like the forward (see below).
#include "common.h"

namespace {

constexpr int TR_THREADS = 256;

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// Phi(x) and phi(x) of the standard normal: gelu(x) = x Phi(x), gelu'(x) = Phi(x) + x phi(x)
__device__ __forceinline__ float norm_cdf(float x) { return norm_cdf_fast(x); }
__device__ __forceinline__ float norm_pdf(float x) { return 0.3989422804014327f * __expf(-0.5f * x * x); }

__device__ __forceinline__ float block_sum(float v, float* red) {      // red: >= TR_THREADS / 64 floats of LDS
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < TR_THREADS / 64; ++w) t += red[w];
    return t;
}

// y2 [M, 2N] = (h | g) -> out [M, N]; one thread per 8 output columns
__global__ void geglu_fwd_kernel(const half_t* __restrict__ y2, half_t* __restrict__ out, long M, int N) {
    const long i = (long)blockIdx.x * TR_THREADS + threadIdx.x;
    const int vpr = N >> 3;
    if (i >= M * vpr) return;
    const long m = i / vpr;
    const int c = (int)(i - m * vpr) * 8;
    const h8 h = as_h8(ld16(y2 + m * 2 * N + c));
    const h8 g = as_h8(ld16(y2 + m * 2 * N + N + c));
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const vsx_f2 v = gelu_erf_f2(vsx_f2{(float)g[e], (float)g[e + 1]});
        o[e] = (half_t)((float)h[e] * v[0]);
        o[e + 1] = (half_t)((float)h[e + 1] * v[1]);
    }
    st16(out + m * N + c, as_u4(o));
}

__global__ void geglu_bwd_kernel(const half_t* __restrict__ dout, const half_t* __restrict__ y2,
                                 half_t* __restrict__ dy2, long M, int N) {
    const long i = (long)blockIdx.x * TR_THREADS + threadIdx.x;
    const int vpr = N >> 3;
    if (i >= M * vpr) return;
    const long m = i / vpr;
    const int c = (int)(i - m * vpr) * 8;
    const h8 h = as_h8(ld16(y2 + m * 2 * N + c));
    const h8 g = as_h8(ld16(y2 + m * 2 * N + N + c));
    const h8 d = as_h8(ld16(dout + m * N + c));
    h8 dh, dg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float gf = (float)g[e], df = (float)d[e];
        const float cdf = norm_cdf(gf);
        dh[e] = (half_t)(df * gf * cdf);
        dg[e] = (half_t)(df * (float)h[e] * (cdf + gf * norm_pdf(gf)));
    }
    st16(dy2 + m * 2 * N + c, as_u4(dh));
    st16(dy2 + m * 2 * N + N + c, as_u4(dg));
}

__global__ void silu_bwd_kernel(const half_t* __restrict__ dy, const half_t* __restrict__ x, half_t* __restrict__ dx,
                                long n) {
    const long i = (long)blockIdx.x * TR_THREADS + threadIdx.x;
    if (i >= n) return;
    const float xf = (float)x[i], s = sigmoid_f(xf);
    dx[i] = (half_t)((float)dy[i] * (s + xf * s * (1.0f - s)));
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm backward as a chunked two-level reduction (the first version ran ONE workgroup per (image, group): 64
// workgroups for the 5-D GroupNorm of a resnet, scalar loads, an integer division per element — 1.7 ms per call on average
// and 36 % of the training step's kernel time, profiles/r03_train_kernel_stats.txt).  Group g of image i covers channels
// [g*cpg, (g+1)*cpg) of all `rows` rows; channels < C1 live in x1 / dx1, the others in x2 / dx2 (the skip concat).
//   pass 1  gnb_partial<0>: per (chunk of rows, image) the group sums of x and x^2             -> finalize: mean, rstd
//   pass 2  gnb_partial<1>: per (chunk, image) the group sums of gz and gz * xhat,
//                           gz = dy * gamma (* silu'(z))                                        -> finalize: m1, m2
//   pass 3  gnb_apply:      dx = rstd * (gz - m1 - xhat * m2)
// A thread owns one 16-byte vector of channels (the same one for every row it visits), so its group parameters live in
// registers; rows are visited `rp` at a time by the workgroup, 16-byte loads and stores.  Deterministic: fixed-shape tree
// per chunk, chunks summed in index order.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GNB_THREADS = 512;

__host__ __device__ inline int gnb_chunk_rows(long rows, long nimg) {      // ~2048 workgroups whatever the split
    long r = rows * nimg / 2048;
    return (int)(r < 8 ? 8 : (r > 512 ? 512 : r));
}

__device__ __forceinline__ uint4 gnb_load(const half_t* x1, const half_t* x2, long row, int c, int C1, int C2) {
    const half_t* ptr = (c < C1) ? x1 + row * C1 + c : x2 + row * C2 + (c - C1);
    return ld16(ptr);
}

// gz of one element: the upstream gradient through gamma and (optionally) SiLU
__device__ __forceinline__ float gnb_gz(float dyv, float xh, float ga, float be, int silu) {
    if (silu) {
        const float z = xh * ga + be;
        const float sg = sigmoid_f(z);
        dyv *= sg + z * sg * (1.0f - sg);
    }
    return dyv * ga;
}

// grid (nchunks, nimg).  WHAT 0: (sum x, sum x^2); WHAT 1: (sum gz, sum gz * xhat) — per group, into
// partial[((img * nchunks + chunk) * groups + g) * 2 ..]
template <int WHAT>
__global__ __launch_bounds__(GNB_THREADS) void gnb_partial_kernel(const half_t* __restrict__ dy,
                                                                  const half_t* __restrict__ x1,
                                                                  const half_t* __restrict__ x2, long rows, int C1, int C2,
                                                                  int groups, const half_t* __restrict__ gamma,
                                                                  const half_t* __restrict__ beta, int silu,
                                                                  const float* __restrict__ stats,
                                                                  float* __restrict__ partial) {
    __shared__ float red_a[4096];
    __shared__ float red_b[4096];
    const int C = C1 + C2;
    const int cpg = C / groups;
    const int vpr = C >> 3;
    const int rp = GNB_THREADS / vpr;          // rows per pass (>= 1: C <= 4096)
    const int tid = threadIdx.x;
    const int rl = tid / vpr;
    const int cv = tid - rl * vpr;
    const long img = blockIdx.y;
    const int rpc = gnb_chunk_rows(rows, gridDim.y);
    const long r0 = (long)blockIdx.x * rpc;
    const long r1 = min(r0 + (long)rpc, rows);
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.f; b[e] = 0.f; }
    if (rl < rp) {
        float mu[8], rs[8], ga[8], be[8];
        if (WHAT == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = cv * 8 + e;
                const int g = c / cpg;
                mu[e] = stats[(img * groups + g) * 2];
                rs[e] = stats[(img * groups + g) * 2 + 1];
                ga[e] = (float)gamma[c];
                be[e] = (float)beta[c];
            }
        }
        for (long r = r0 + rl; r < r1; r += rp) {
            const h8 xv = as_h8(gnb_load(x1, x2, img * rows + r, cv * 8, C1, C2));
            if (WHAT == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)xv[e];
                    a[e] += f;
                    b[e] += f * f;
                }
            } else {
                const h8 dv = as_h8(ld16(dy + (img * rows + r) * C + cv * 8));
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)xv[e] - mu[e]) * rs[e];
                    const float gz = gnb_gz((float)dv[e], xh, ga[e], be[e], silu);
                    a[e] += gz;
                    b[e] += gz * xh;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red_a[rl * C + cv * 8 + e] = a[e];
            red_b[rl * C + cv * 8 + e] = b[e];
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += GNB_THREADS) {
        float sa = 0.f, sb = 0.f;
        for (int r = 0; r < rp; ++r) { sa += red_a[r * C + c]; sb += red_b[r * C + c]; }
        red_a[c] = sa;
        red_b[c] = sb;
    }
    __syncthreads();
    if (tid < groups) {
        float sa = 0.f, sb = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { sa += red_a[c]; sb += red_b[c]; }
        float* out = partial + ((img * gridDim.x + blockIdx.x) * groups + tid) * 2;
        out[0] = sa;
        out[1] = sb;
    }
}

// grid (groups, nimg), block 256: chunk partials -> WHAT 0: (mean, rstd); WHAT 1: (mean of gz, mean of gz * xhat)
template <int WHAT>
__global__ __launch_bounds__(256) void gnb_finalize_kernel(const float* __restrict__ partial, int nchunks, int groups,
                                                           float inv_count, float eps, float* __restrict__ out) {
    __shared__ float s_a[256];
    __shared__ float s_b[256];
    const int tid = threadIdx.x;
    const int g = blockIdx.x;
    const long img = blockIdx.y;
    const float* pp = partial + (img * nchunks * groups + g) * 2;
    float a = 0.f, b = 0.f;
    for (int ch = tid; ch < nchunks; ch += 256) {
        a += pp[(long)ch * groups * 2];
        b += pp[(long)ch * groups * 2 + 1];
    }
    s_a[tid] = a;
    s_b[tid] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) { s_a[tid] += s_a[tid + w]; s_b[tid] += s_b[tid + w]; }
        __syncthreads();
    }
    if (tid == 0) {
        float* o = out + (img * groups + g) * 2;
        if (WHAT == 0) {
            const float mean = s_a[0] * inv_count;
            o[0] = mean;
            o[1] = rsqrtf(fmaxf(s_b[0] * inv_count - mean * mean, 0.f) + eps);
        } else {
            o[0] = s_a[0] * inv_count;
            o[1] = s_b[0] * inv_count;
        }
    }
}

// grid (nchunks, nimg): dx = rstd * (gz - m1 - xhat * m2)
__global__ __launch_bounds__(GNB_THREADS) void gnb_apply_kernel(const half_t* __restrict__ dy, const half_t* __restrict__ x1,
                                                                const half_t* __restrict__ x2, long rows, int C1, int C2,
                                                                int groups, const half_t* __restrict__ gamma,
                                                                const half_t* __restrict__ beta, int silu,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ sums, half_t* __restrict__ dx1,
                                                                half_t* __restrict__ dx2) {
    const int C = C1 + C2;
    const int cpg = C / groups;
    const int vpr = C >> 3;
    const int rp = GNB_THREADS / vpr;
    const int tid = threadIdx.x;
    const int rl = tid / vpr;
    const int cv = tid - rl * vpr;
    if (rl >= rp) return;
    const long img = blockIdx.y;
    const int rpc = gnb_chunk_rows(rows, gridDim.y);
    const long r0 = (long)blockIdx.x * rpc;
    const long r1 = min(r0 + (long)rpc, rows);
    float mu[8], rs[8], ga[8], be[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cv * 8 + e;
        const int g = c / cpg;
        mu[e] = stats[(img * groups + g) * 2];
        rs[e] = stats[(img * groups + g) * 2 + 1];
        m1[e] = sums[(img * groups + g) * 2];
        m2[e] = sums[(img * groups + g) * 2 + 1];
        ga[e] = (float)gamma[c];
        be[e] = (float)beta[c];
    }
    const int c0 = cv * 8;
    for (long r = r0 + rl; r < r1; r += rp) {
        const long row = img * rows + r;
        const h8 xv = as_h8(gnb_load(x1, x2, row, c0, C1, C2));
        const h8 dv = as_h8(ld16(dy + row * C + c0));
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = ((float)xv[e] - mu[e]) * rs[e];
            const float gz = gnb_gz((float)dv[e], xh, ga[e], be[e], silu);
            o[e] = (half_t)(rs[e] * (gz - m1[e] - xh * m2[e]));
        }
        if (c0 < C1) st16(dx1 + row * C1 + c0, as_u4(o));
        else st16(dx2 + row * C2 + (c0 - C1), as_u4(o));
    }
}

// LayerNorm backward: one wave per row of C (multiple of 8, <= 2048) elements; 4 rows per workgroup
__global__ __launch_bounds__(256) void ln_bwd_kernel(const half_t* __restrict__ dy, const half_t* __restrict__ x,
                                                     const half_t* __restrict__ gamma, float eps,
                                                     half_t* __restrict__ dx, long M, int C) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const half_t* xr = x + row * C;
    const half_t* dr = dy + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += (float)xr[c];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = (float)xr[c] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float gz = (float)dr[c] * (float)gamma[c];
        s1 += gz;
        s2 += gz * ((float)xr[c] - mean) * rstd;
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float xh = ((float)xr[c] - mean) * rstd;
        const float gz = (float)dr[c] * (float)gamma[c];
        dx[row * C + c] = (half_t)(rstd * (gz - m1 - xh * m2));
    }
}

// one wave per row of ncols (row stride ld): dS over dP
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const half_t* __restrict__ P, half_t* __restrict__ dP,
                                                          long nrows, int ncols, long ld, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const half_t* p = P + row * ld;
    half_t* d = dP + row * ld;
    float s = 0.f;
    for (int c = lane; c < ncols; c += 64) s += (float)p[c] * (float)d[c];
    s = wave_sum(s);
    for (int c = lane; c < ncols; c += 64) d[c] = (half_t)(scale * (float)p[c] * ((float)d[c] - s));
}

// x [n, 2h, 2w, c] -> y [n, h, w, c]; one thread per 8 channels of an output pixel
__global__ void sum_pool2x2_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, long n, int h, int w, int c) {
    const long i = (long)blockIdx.x * TR_THREADS + threadIdx.x;
    const int vpr = c >> 3;
    const long total = n * h * w * vpr;
    if (i >= total) return;
    const int cv = (int)(i % vpr) * 8;
    const long pix = i / vpr;
    const int xw = (int)(pix % w);
    const long t = pix / w;
    const int yh = (int)(t % h);
    const long img = t / h;
    const half_t* src = x + ((img * 2 * h + 2 * yh) * 2 * w + 2 * xw) * c + cv;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dxp = 0; dxp < 2; ++dxp) {
            const h8 v = as_h8(ld16(src + ((long)dy * 2 * w + dxp) * c));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
    st16(y + pix * c + cv, as_u4(o));
}

// One workgroup per point: dfeat[p, :] = out_scale * sum over frames and the 4 corners of weight * dmap[f, y, x, :],
// frames in index order (fixed summation order), same fp16 sub-pixel position and weights as vsx_adapter_scatter.
__global__ void adapter_gather_kernel(const float* __restrict__ tracks, const int* __restrict__ selected,
                                      const half_t* __restrict__ dmap, half_t* __restrict__ dfeat, int F, int P, int C,
                                      int h, int w, float rate, float out_scale) {
    const int pt = blockIdx.x;
    if (!selected[pt]) return;                      // dfeat is zero-filled by the caller
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int f = 0; f < F; ++f) {
            const float px = tracks[((long)f * P + pt) * 2 + 0];
            const float py = tracks[((long)f * P + pt) * 2 + 1];
            if (px < 0.f || py < 0.f) continue;
            const float x = (float)(half_t)((float)(half_t)px / rate), y = (float)(half_t)((float)(half_t)py / rate);
            int x1 = (int)x, y1 = (int)y;
            int x2 = x1 + 1, y2 = y1 + 1;
            const float xf = (float)(half_t)(x - (float)x1), yf = (float)(half_t)(y - (float)y1);
            x1 = max(min(x1, w - 1), 0); x2 = max(min(x2, w - 1), 0);
            y1 = max(min(y1, h - 1), 0); y2 = max(min(y2, h - 1), 0);
            const float xm = (float)(half_t)(1.f - xf), ym = (float)(half_t)(1.f - yf);
            const float wgt[4] = {(float)(half_t)(xm * ym), (float)(half_t)(xf * ym), (float)(half_t)(xm * yf),
                                  (float)(half_t)(xf * yf)};
            const int xs[4] = {x1, x2, x1, x2};
            const int ys[4] = {y1, y1, y2, y2};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                acc += wgt[k] * (float)dmap[(((long)f * h + ys[k]) * w + xs[k]) * C + c];
        }
        dfeat[(long)pt * C + c] = (half_t)(acc * out_scale);
    }
}

inline unsigned blocks_for(long n) { return (unsigned)((n + TR_THREADS - 1) / TR_THREADS); }

}  // namespace

extern "C" int vsx_geglu_fwd(const void* y2, void* out, int64_t M, int64_t N, vsx_stream_t stream) {
    VSX_REQUIRE(y2 && out && M >= 0 && N > 0 && N % 8 == 0, VSX_E_BADSHAPE, "geglu_fwd: bad arguments (N %% 8 == 0)");
    if (M == 0) return VSX_OK;
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(blocks_for(M * (N / 8))), dim3(TR_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)y2, (half_t*)out, (long)M, (int)N);
    return vsx_check_launch("vsx_geglu_fwd");
}

extern "C" int vsx_geglu_bwd(const void* dout, const void* y2, void* dy2, int64_t M, int64_t N, vsx_stream_t stream) {
    VSX_REQUIRE(dout && y2 && dy2 && M >= 0 && N > 0 && N % 8 == 0, VSX_E_BADSHAPE, "geglu_bwd: bad arguments");
    if (M == 0) return VSX_OK;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(blocks_for(M * (N / 8))), dim3(TR_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)dout, (const half_t*)y2, (half_t*)dy2, (long)M, (int)N);
    return vsx_check_launch("vsx_geglu_bwd");
}

extern "C" int vsx_silu_bwd(const void* dy, const void* x, void* dx, int64_t n, vsx_stream_t stream) {
    VSX_REQUIRE(dy && x && dx && n >= 0, VSX_E_BADSHAPE, "silu_bwd: bad arguments");
    if (n == 0) return VSX_OK;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(blocks_for(n)), dim3(TR_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)dy, (const half_t*)x, (half_t*)dx, (long)n);
    return vsx_check_launch("vsx_silu_bwd");
}

extern "C" int64_t vsx_groupnorm_bwd_workspace(int64_t nimg, int64_t rows, int64_t groups) {
    if (nimg <= 0 || rows <= 0 || groups <= 0) return 0;
    const int rpc = gnb_chunk_rows(rows, nimg);
    const int64_t nchunks = (rows + rpc - 1) / rpc;
    return nimg * groups * (4 + 2 * nchunks);            // floats: stats | sums | chunk partials (reused by both passes)
}

extern "C" int vsx_groupnorm_bwd(const void* dy, const void* x1, const void* x2, int64_t nimg, int64_t rows, int64_t C1,
                                 int64_t C2, int64_t groups, const void* gamma, const void* beta, float eps,
                                 int64_t silu, void* ws, void* dx1, void* dx2, vsx_stream_t stream) {
    VSX_REQUIRE(dy && x1 && gamma && beta && ws && dx1 && (C2 == 0 || (x2 && dx2)), VSX_E_BADSHAPE,
                "groupnorm_bwd: null argument");
    const int64_t C = C1 + C2;
    VSX_REQUIRE(nimg > 0 && rows > 0 && C1 > 0 && C2 >= 0 && groups > 0 && groups <= 256 && C % groups == 0 &&
                    nimg <= 65535,
                VSX_E_BADSHAPE, "groupnorm_bwd: bad sizes");
    VSX_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && C <= 4096, VSX_E_UNSUPPORTED,
                "groupnorm_bwd: channel counts must be multiples of 8 and C1 + C2 <= 4096 (C1=%ld C2=%ld)", (long)C1, (long)C2);
    VSX_REQUIRE(vsx_aligned16(dy) && vsx_aligned16(x1) && vsx_aligned16(x2) && vsx_aligned16(dx1) && vsx_aligned16(dx2),
                VSX_E_BADSHAPE, "groupnorm_bwd: tensors must be 16-byte aligned");
    const int rpc = gnb_chunk_rows(rows, nimg);
    const int nchunks = (int)((rows + rpc - 1) / rpc);
    float* stats = (float*)ws;
    float* sums = stats + nimg * groups * 2;
    float* partial = sums + nimg * groups * 2;
    const float inv_count = 1.0f / ((float)rows * (float)(C / groups));
    const dim3 gchunks((unsigned)nchunks, (unsigned)nimg), ggroups((unsigned)groups, (unsigned)nimg);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gnb_partial_kernel<0>, gchunks, dim3(GNB_THREADS), 0, st, (const half_t*)dy, (const half_t*)x1,
                       (const half_t*)x2, (long)rows, (int)C1, (int)C2, (int)groups, (const half_t*)gamma,
                       (const half_t*)beta, (int)silu, (const float*)stats, partial);
    hipLaunchKernelGGL(gnb_finalize_kernel<0>, ggroups, dim3(256), 0, st, (const float*)partial, nchunks, (int)groups,
                       inv_count, eps, stats);
    hipLaunchKernelGGL(gnb_partial_kernel<1>, gchunks, dim3(GNB_THREADS), 0, st, (const half_t*)dy, (const half_t*)x1,
                       (const half_t*)x2, (long)rows, (int)C1, (int)C2, (int)groups, (const half_t*)gamma,
                       (const half_t*)beta, (int)silu, (const float*)stats, partial);
    hipLaunchKernelGGL(gnb_finalize_kernel<1>, ggroups, dim3(256), 0, st, (const float*)partial, nchunks, (int)groups,
                       inv_count, eps, sums);
    hipLaunchKernelGGL(gnb_apply_kernel, gchunks, dim3(GNB_THREADS), 0, st, (const half_t*)dy, (const half_t*)x1,
                       (const half_t*)x2, (long)rows, (int)C1, (int)C2, (int)groups, (const half_t*)gamma,
                       (const half_t*)beta, (int)silu, (const float*)stats, (const float*)sums, (half_t*)dx1,
                       (half_t*)dx2);
    return vsx_check_launch("vsx_groupnorm_bwd");
}

extern "C" int vsx_layernorm_bwd(const void* dy, const void* x, const void* gamma, float eps, void* dx, int64_t M,
                                 int64_t C, vsx_stream_t stream) {
    VSX_REQUIRE(dy && x && gamma && dx && M >= 0 && C > 0, VSX_E_BADSHAPE, "layernorm_bwd: bad arguments");
    if (M == 0) return VSX_OK;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)dy, (const half_t*)x, (const half_t*)gamma, eps, (half_t*)dx, (long)M, (int)C);
    return vsx_check_launch("vsx_layernorm_bwd");
}

extern "C" int vsx_softmax_bwd(const void* P, void* dP, int64_t nrows, int64_t ncols, int64_t ld, float scale,
                               vsx_stream_t stream) {
    VSX_REQUIRE(P && dP && nrows >= 0 && ncols > 0 && ld >= ncols, VSX_E_BADSHAPE, "softmax_bwd: bad arguments");
    if (nrows == 0) return VSX_OK;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)P, (half_t*)dP, (long)nrows, (int)ncols, (long)ld, scale);
    return vsx_check_launch("vsx_softmax_bwd");
}

extern "C" int vsx_sum_pool2x2(const void* x, void* y, int64_t n, int64_t h, int64_t w, int64_t c, vsx_stream_t stream) {
    VSX_REQUIRE(x && y && n >= 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0, VSX_E_BADSHAPE, "sum_pool2x2: bad arguments");
    if (n == 0) return VSX_OK;
    hipLaunchKernelGGL(sum_pool2x2_kernel, dim3(blocks_for(n * h * w * (c / 8))), dim3(TR_THREADS), 0,
                       (hipStream_t)stream, (const half_t*)x, (half_t*)y, (long)n, (int)h, (int)w, (int)c);
    return vsx_check_launch("vsx_sum_pool2x2");
}

extern "C" int vsx_adapter_gather(const float* tracks, const int32_t* selected, const void* dmap, void* dfeat, int64_t F,
                                  int64_t P, int64_t C, int64_t h, int64_t w, float rate, float out_scale,
                                  vsx_stream_t stream) {
    VSX_REQUIRE(tracks && selected && dmap && dfeat, VSX_E_BADSHAPE, "adapter_gather: null argument");
    VSX_REQUIRE(F > 0 && P >= 0 && C > 0 && h > 0 && w > 0 && rate > 0.f, VSX_E_BADSHAPE, "adapter_gather: bad sizes");
    if (P == 0) return VSX_OK;
    hipLaunchKernelGGL(adapter_gather_kernel, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, tracks,
                       (const int*)selected, (const half_t*)dmap, (half_t*)dfeat, (int)F, (int)P, (int)C, (int)h, (int)w,
                       rate, out_scale);
    return vsx_check_launch("vsx_adapter_gather");
}
