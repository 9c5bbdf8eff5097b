// K10/K11 — element-wise glue of the denoising loop for gfx950: SiLU, axpy, latent layout
// conversion at the UNet boundary, fused CFG + DDIM update, Prompt-to-Prompt latent blend and the
// SparsePointAdapter bilinear scatter.  All HBM-bound and tiny next to the UNet itself.
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;

__global__ void silu_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i < n) y[i] = (half_t)silu_f((float)x[i]);
}

// CLIP's activation: x * sigmoid(1.702 x)
__global__ void quick_gelu_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i < n) {
        const float v = (float)x[i];
        y[i] = (half_t)(v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v)));
    }
}

__global__ void axpy_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, float s,
                            half_t* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i < n) y[i] = (half_t)((float)a[i] + s * (float)b[i]);
}

// x[B,Cin,F,HW] -> y[B*F,HW,Cpad]; one thread per output pixel.
__global__ void pack_latents_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int B, int Cin, int F,
                                    long HW, int Cpad) {
    const long i = (long)blockIdx.x * EW_THREADS + threadIdx.x;  // (b*F + f)*HW + s
    const long total = (long)B * F * HW;
    if (i >= total) return;
    const long s = i % HW;
    const long bf = i / HW;
    const int f = (int)(bf % F);
    const int b = (int)(bf / F);
    for (int c = 0; c < Cpad; ++c) {
        half_t v = (half_t)0.f;
        if (c < Cin) v = x[(((long)b * Cin + c) * F + f) * HW + s];
        y[i * Cpad + c] = v;
    }
}

// x[B*F,HW,Cs] -> y[B,Cout,F,HW]; one thread per (pixel), loops channels.
__global__ void unpack_latents_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int B, int Cout, int F,
                                      long HW, int Cs) {
    const long i = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    const long total = (long)B * F * HW;
    if (i >= total) return;
    const long s = i % HW;
    const long bf = i / HW;
    const int f = (int)(bf % F);
    const int b = (int)(bf / F);
    for (int c = 0; c < Cout; ++c) y[(((long)b * Cout + c) * F + f) * HW + s] = x[i * Cs + c];
}

__global__ void cfg_ddim_kernel(const half_t* __restrict__ x, const half_t* __restrict__ eu,
                                const half_t* __restrict__ ec, float g, float sa_t, float s1a_t, float sa_n,
                                float s1a_n, half_t* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= n) return;
    float e = (float)eu[i];
    if (ec) e = e + g * ((float)ec[i] - e);
    const float x0 = ((float)x[i] - s1a_t * e) / sa_t;
    out[i] = (half_t)(sa_n * x0 + s1a_n * e);
}

__global__ void masked_blend_kernel(const half_t* __restrict__ x, const half_t* __restrict__ src,
                                    const half_t* __restrict__ mask, half_t* __restrict__ out, long C, long n_sp) {
    const long i = (long)blockIdx.x * EW_THREADS + threadIdx.x;
    if (i >= C * n_sp) return;
    const long s = i % n_sp;
    const float a = (float)src[i];
    out[i] = (half_t)(a + (float)mask[s] * ((float)x[i] - a));
}

// One workgroup per (frame, point): the 4 corner splats of a point are applied sequentially by
// the same threads (clamped corners may coincide and must accumulate, adapter_model.py:33-45);
// different points of a frame may hit the same pixel, so the accumulation uses fp32 atomics on a
// staging buffer?  No: points are few (P <= 64) and C-vectors are independent per channel, so a
// frame is owned by ONE workgroup that walks its points in index order — the same order as the
// reference's python loop — which makes the fp16 accumulation order identical.
__global__ void adapter_scatter_kernel(const float* __restrict__ tracks, const int* __restrict__ selected,
                                       const half_t* __restrict__ feat, half_t* __restrict__ out, int P, int C,
                                       int h, int w, float rate, float out_scale) {
    const int f = blockIdx.x;
    for (int pt = 0; pt < P; ++pt) {
        if (!selected[pt]) continue;
        const float px = tracks[((long)f * P + pt) * 2 + 0];
        const float py = tracks[((long)f * P + pt) * 2 + 1];
        if (px < 0.f || py < 0.f) continue;
        // the reference holds the tracks in fp16 and divides in fp16 (pipeline_videoswap.py:532-533,
        // adapter_model.py:129): reproduce that quantisation of the sub-pixel position
        const float x = (float)(half_t)((float)(half_t)px / rate), y = (float)(half_t)((float)(half_t)py / rate);
        int x1 = (int)x, y1 = (int)y;
        int x2 = x1 + 1, y2 = y1 + 1;
        const float xf = (float)(half_t)(x - (float)x1), yf = (float)(half_t)(y - (float)y1);
        x1 = max(min(x1, w - 1), 0); x2 = max(min(x2, w - 1), 0);
        y1 = max(min(y1, h - 1), 0); y2 = max(min(y2, h - 1), 0);
        // the reference's weights are products of 0-dim fp16 tensors: every intermediate is rounded to fp16
        const float xm = (float)(half_t)(1.f - xf), ym = (float)(half_t)(1.f - yf);
        const float wgt[4] = {(float)(half_t)(xm * ym), (float)(half_t)(xf * ym), (float)(half_t)(xm * yf),
                              (float)(half_t)(xf * yf)};
        const int xs[4] = {x1, x2, x1, x2};
        const int ys[4] = {y1, y1, y2, y2};
        for (int k = 0; k < 4; ++k) {
            half_t* dst = out + (((long)f * h + ys[k]) * w + xs[k]) * C;
            for (int c = threadIdx.x; c < C; c += blockDim.x) {
                const float v = (float)feat[(long)pt * C + c];
                // reference rounds (value*w) to fp16, then the += rounds again
                const half_t add = (half_t)(v * wgt[k]);
                dst[c] = (half_t)((float)dst[c] + (float)add);
            }
            // same threads own the same channels for every corner/point: no barrier needed
        }
    }
    // t2i_guidance_scale multiplies the FINISHED map (pipeline_videoswap.py:545-546), one more fp16 rounding; a thread
    // owns the same channels of every pixel, so it only touches values it accumulated itself
    if (out_scale != 1.0f) {
        half_t* base = out + (long)f * h * w * C;
        for (long px = 0; px < (long)h * w; ++px)
            for (int c = threadIdx.x; c < C; c += blockDim.x) {
                const half_t v = base[px * C + c];
                if ((float)v != 0.f) base[px * C + c] = (half_t)((float)v * out_scale);
            }
    }
}

inline unsigned blocks_for(long n) { return (unsigned)((n + EW_THREADS - 1) / EW_THREADS); }

}  // namespace

extern "C" int vsx_silu(const void* x, void* y, int64_t n, vsx_stream_t stream) {
    VSX_REQUIRE(x && y && n >= 0, VSX_E_BADSHAPE, "silu: bad arguments");
    if (n == 0) return VSX_OK;
    hipLaunchKernelGGL(silu_kernel, dim3(blocks_for(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const half_t*)x,
                       (half_t*)y, (long)n);
    return vsx_check_launch("vsx_silu");
}

extern "C" int vsx_quick_gelu(const void* x, void* y, int64_t n, vsx_stream_t stream) {
    VSX_REQUIRE(x && y && n >= 0, VSX_E_BADSHAPE, "quick_gelu: bad arguments");
    if (n == 0) return VSX_OK;
    hipLaunchKernelGGL(quick_gelu_kernel, dim3(blocks_for(n)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)x, (half_t*)y, (long)n);
    return vsx_check_launch("vsx_quick_gelu");
}

extern "C" int vsx_axpy(const void* a, const void* b, float s, void* y, int64_t n, vsx_stream_t stream) {
    VSX_REQUIRE(a && b && y && n >= 0, VSX_E_BADSHAPE, "axpy: bad arguments");
    if (n == 0) return VSX_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks_for(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const half_t*)a,
                       (const half_t*)b, s, (half_t*)y, (long)n);
    return vsx_check_launch("vsx_axpy");
}

extern "C" int vsx_pack_latents(const void* x, void* y, int64_t B, int64_t Cin, int64_t F, int64_t HW, int64_t Cpad,
                                vsx_stream_t stream) {
    VSX_REQUIRE(x && y && B > 0 && Cin > 0 && F > 0 && HW > 0 && Cpad >= Cin, VSX_E_BADSHAPE, "pack_latents: bad arguments");
    hipLaunchKernelGGL(pack_latents_kernel, dim3(blocks_for(B * F * HW)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)x, (half_t*)y, (int)B, (int)Cin, (int)F, (long)HW, (int)Cpad);
    return vsx_check_launch("vsx_pack_latents");
}

extern "C" int vsx_unpack_latents(const void* x, void* y, int64_t B, int64_t Cout, int64_t F, int64_t HW, int64_t Cs,
                                  vsx_stream_t stream) {
    VSX_REQUIRE(x && y && B > 0 && Cout > 0 && F > 0 && HW > 0 && Cs >= Cout, VSX_E_BADSHAPE,
                "unpack_latents: bad arguments");
    hipLaunchKernelGGL(unpack_latents_kernel, dim3(blocks_for(B * F * HW)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)x, (half_t*)y, (int)B, (int)Cout, (int)F, (long)HW, (int)Cs);
    return vsx_check_launch("vsx_unpack_latents");
}

extern "C" int vsx_cfg_ddim_step(const void* x, const void* eps_u, const void* eps_c, float guidance, float alpha_t,
                                 float alpha_next, void* out, int64_t n, vsx_stream_t stream) {
    VSX_REQUIRE(x && eps_u && out && n >= 0, VSX_E_BADSHAPE, "cfg_ddim_step: bad arguments");
    VSX_REQUIRE(alpha_t > 0.f && alpha_t <= 1.f && alpha_next > 0.f && alpha_next <= 1.f, VSX_E_BADSHAPE,
                "cfg_ddim_step: alphas must be in (0,1]");
    if (n == 0) return VSX_OK;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(blocks_for(n)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)x, (const half_t*)eps_u, (const half_t*)eps_c, guidance, sqrtf(alpha_t),
                       sqrtf(1.f - alpha_t), sqrtf(alpha_next), sqrtf(1.f - alpha_next), (half_t*)out, (long)n);
    return vsx_check_launch("vsx_cfg_ddim_step");
}

extern "C" int vsx_masked_blend(const void* x, const void* src, const void* mask, void* out, int64_t C, int64_t n_sp,
                                vsx_stream_t stream) {
    VSX_REQUIRE(x && src && mask && out && C > 0 && n_sp > 0, VSX_E_BADSHAPE, "masked_blend: bad arguments");
    hipLaunchKernelGGL(masked_blend_kernel, dim3(blocks_for(C * n_sp)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const half_t*)x, (const half_t*)src, (const half_t*)mask, (half_t*)out, (long)C, (long)n_sp);
    return vsx_check_launch("vsx_masked_blend");
}

extern "C" int vsx_adapter_scatter(const float* tracks, const int32_t* selected, const void* feat, void* out,
                                   int64_t F, int64_t P, int64_t C, int64_t h, int64_t w, float rate, float out_scale,
                                   vsx_stream_t stream) {
    VSX_REQUIRE(tracks && selected && feat && out, VSX_E_BADSHAPE, "adapter_scatter: null argument");
    VSX_REQUIRE(F > 0 && P >= 0 && C > 0 && h > 0 && w > 0 && rate > 0.f, VSX_E_BADSHAPE, "adapter_scatter: bad sizes");
    if (P == 0) return VSX_OK;
    hipLaunchKernelGGL(adapter_scatter_kernel, dim3((unsigned)F), dim3(256), 0, (hipStream_t)stream, tracks,
                       (const int*)selected, (const half_t*)feat, (half_t*)out, (int)P, (int)C, (int)h, (int)w, rate,
                       out_scale);
    return vsx_check_launch("vsx_adapter_scatter");
}
