/*
 * vsx.h — C ABI of libvsx.so, the MI355X (gfx950) denoising-path kernels for VideoSwap.
 *
 * This is the drop-in boundary of the hot path: the Python host classes in
 * videoswap_amd/ (same names and call signatures as the reference's UNet / Attention / pipeline
 * classes) bind exactly these entry points through ctypes.  Nothing here takes a torch type:
 * plain device pointers, sizes, element strides and the hipStream_t to launch on.
 *
 * Contract (all entry points):
 *   - every tensor is fp16 (IEEE binary16) in device memory unless stated; accumulation is fp32;
 *   - the CALLER owns all memory (inputs, outputs, workspaces); the library never allocates or
 *     frees device memory and never synchronises; every call is asynchronous on `stream`;
 *   - return 0 (VSX_OK) on success, a negative VSX_E_* code otherwise; vsx_last_error() returns a
 *     thread-local description of the last failure; no exception crosses the ABI;
 *   - pointers must be 16-byte aligned and row strides multiples of 8 elements (16-byte vector
 *     access) unless a function says otherwise; violations return VSX_E_BADSHAPE;
 *   - activations are channels-last: a video tensor the reference holds as [B,C,F,H,W]
 *     (reference videoswap/models/animatediff_models/resnet.py:12-16) lives here as
 *     [B*F, H, W, C] == token matrix [B*F*H*W, C].
 *
 * Each entry point cites the reference call site it replaces (file:line into showlab/VideoSwap).
 */
#ifndef VSX_H_
#define VSX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSX_ABI_VERSION 10

#define VSX_OK 0
#define VSX_E_BADSHAPE (-1)
#define VSX_E_UNSUPPORTED (-2)
#define VSX_E_LAUNCH (-3)
#define VSX_E_WORKSPACE (-4)

typedef void* vsx_stream_t; /* hipStream_t */

int vsx_abi_version(void);
const char* vsx_last_error(void);
/* sha256 (hex) of the kernel sources + flags the library was built from; the Python loader compares it with the
 * sources it sits next to and refuses a stale binary. */
const char* vsx_source_digest(void);
/* Tuning / test switch (process-wide, not thread-safe against concurrent launches): "gemm_pp" = 0 never / 1 default /
 * 2 always-when-eligible use of the persistent ping-pong GEMM kernel; "pp_sched" = its option bits (8: linear tile walk; 4: convolutions sum K
 * tap-major like the tile kernels instead of taps-inner; 16: a private A slab per convolution tap; 32: common piece order in every CU).  Results
 * are identical for every setting except bit 4 (same arithmetic; bit 4 selects another fp32 summation order of the same products).
 * "tile_tune" (diagnostics, tools/small_m_sweep.py) forces a tile / ring depth / K-slice count of the workgroup-per-tile kernels. */
int vsx_set_option(const char* name, int64_t value);

/* ------------------------------------------------------------------------------------------
 * K1/K2: MFMA GEMM / implicit-GEMM convolution with fused epilogues.
 *   C[z][m, n] = epilogue( alpha * sum_k A[z][m, k] * B[z][n, k] )        (both operands K-major)
 * Replaces: nn.Linear / 1x1 nn.Conv2d (attention.py:65,93; motion_module.py:113,136; diffusers
 * Attention.to_q/k/v/out, FeedForward), InflatedConv3d 3x3 (resnet.py:9-18), Downsample3D
 * (resnet.py:83), Upsample3D nearest-2x + conv (resnet.py:54,66), the channel concat of the up
 * blocks (unet_blocks.py:618,720), torch.baddbmm/bmm of get_attention_scores / probs @ V
 * (attention_register.py:70-76,150-156).
 * All integer fields are 64-bit so the struct has no padding on any ABI.
 * ------------------------------------------------------------------------------------------ */
typedef struct vsx_gemm_desc {
    int64_t M, N, K;     /* N = output columns (for geglu: N = K_out = half of B's rows) */
    int64_t batch0, batch1; /* grid.z = batch0*batch1; z0 = z / batch1, z1 = z % batch1 */

    /* A operand */
    const void* A;       /* a_mode 0: [M,K] row-major, row stride lda; a_mode 1: NHWC image(s) */
    const void* A2;      /* a_mode 1 only: second source concatenated on C (may be NULL) */
    int64_t lda, a_bs0, a_bs1;
    int64_t a_mode;      /* 0 = plain rows, 1 = implicit im2col of a ks x ks conv, pad ks/2 */
    int64_t H, W;        /* a_mode 1: logical input height/width (after the optional 2x upsample) */
    int64_t C1, C2;      /* a_mode 1: channels of A and A2 (C2 = 0 without A2); K = ks*ks*(C1+C2).  Multiples of 8;
                            with A2 both must be multiples of 64 (one K slab never straddles the two sources) */
    int64_t ks, stride;  /* a_mode 1: kernel size 1 or 3, stride 1 or 2 */
    int64_t upsample;    /* a_mode 1: 1 = A/A2 are [.., H/2, W/2, C], read as nearest-2x upsampled.
                            ABI v9: 2 = the same convolution in its SUB-PIXEL form (Upsample3D, resnet.py:54,66): a 3x3 window on
                            the upsampled image meets 2 x 2 source pixels, so output pixel (2i+ph, 2j+pw) is a 2x2-tap window on
                            the source with the coinciding filter rows / columns added up.  B = FOUR [N, 9 C] matrices, class
                            2 ph + pw first to last, whose taps (ph.., pw..) of a pad-1 3x3 window on the source hold the sums
                            (videoswap_amd.ops.subpixel_weights); 4/9 of the multiplications.  ks 3, stride 1, one source with
                            C1 % 64 == 0, bias only, N % 320 == 0, M/4 a multiple of 256 and a launch large enough for the
                            persistent kernel — VSX_E_UNSUPPORTED otherwise (use upsample = 1). */

    /* B operand: [N (or 2N for geglu), K] row-major (PyTorch Linear weight / OHWI conv weight) */
    const void* B;
    int64_t ldb, b_bs0, b_bs1;

    /* C */
    void* C;
    int64_t ldc, c_bs0, c_bs1;
    int64_t c_mode;         /* 0: C[m*ldc+n]; 1: transposed per image: C[img*c_img_stride + n*ldc + m%rows] */
    int64_t c_rows_per_img; /* c_mode 1: rows (tokens) per image */
    int64_t c_img_stride;   /* c_mode 1: element stride between images */

    /* epilogue */
    const void* bias;     /* [N] (geglu: [2N]) or NULL */
    const void* rowvec;   /* [ceil(M/rows_per_vec), N] added to row m from row m/rows_per_vec, or NULL
                             (time-embedding broadcast, resnet.py:172-176) */
    int64_t rows_per_vec;
    const void* residual; /* [M,N] with row stride ldr added after everything else, or NULL */
    int64_t ldr, r_bs0, r_bs1;
    int64_t geglu;        /* 1: out[m,n] = (acc_h+b_h) * gelu_erf(acc_g+b_g), h=row n, g=row N+n of B
                             (diffusers GEGLU used at attention.py:204, motion_module.py:218) */
    double alpha;
    /* optional split-K workspace (fp32 partial sums), caller-allocated: vsx_gemm_workspace(d) bytes, or NULL.  Small-M
       / long-K problems (the 8x8 and 16x16 UNet levels) are then sliced along K so that every CU gets a tile. */
    void* workspace;
    int64_t workspace_bytes;
    /* ABI v4, a_mode 1: zero padding before (top / left) and after (bottom / right) the image, 0..ks-1 each; the
       pair (-1, -1) selects the symmetric ks/2 of nn.Conv2d(padding=ks//2).  diffusers' VAE encoder downsamples with
       F.pad(x, (0, 1, 0, 1)) + a stride-2 conv without padding = (pad_lo, pad_hi) = (0, 1). */
    int64_t pad_lo, pad_hi;
    /* ABI v7: LayerNorm folded into the Linear that consumes it (attention.py:182,199,205 norm1/2/3 -> to_q/k/v, GEGLU;
       motion_module.py:213,219).  With W' = W o gamma as the B operand and A = the RAW activation,
           LN(x) W^T + b  =  rstd_m * acc[m,n]  -  rstd_m * mean_m * c1[n]  +  (beta W^T + b)[n]
       rowscale = fp32 [M][2] = (rstd_m, -rstd_m * mean_m) (vsx_row_stats), colvec = fp32 [N] (geglu: [2N]) = c1[n] =
       sum_k W'[n,k]; the last term is passed as `bias`, the temporal positional encoding pe W^T as `rowvec`.  The
       normalised tensor is never written or re-read.  Plain (a_mode 0), unbatched GEMMs only; NULL / NULL = off. */
    const void* rowscale;
    const void* colvec;
    /* ABI v8: row statistics of the OUTPUT, for a LayerNorm that follows this Linear (the producer side of the fold above:
       every LayerNorm input of the UNet is the output of a GEMM — proj_in, attention to_out + residual; attention.py:182,
       199,205, motion_module.py:213,219).  rowstats = fp32 [M][rowstats_parts][2] = (sum, sum of squares) of the ROUNDED
       fp16 outputs of row m over column part p, written by the epilogue that stores them; vsx_row_stats_combine turns the
       parts into the [M][2] rowscale of the consumer (one pass over 48 bytes per row instead of vsx_row_stats' pass over
       the whole row).  Only the persistent kernel emits them (plain GEMM, N a multiple of 320, epilogue = bias and / or
       residual or row vector): ASK FIRST with vsx_gemm_rowstats_parts(d) — the number of parts this launch will write
       (6 per 320 columns on the persistent kernel, one per wave — 10, or 5 — on the weight-stationary K = 320 kernel), 0 = it will not (keep rowstats NULL and use vsx_row_stats).  NULL / 0 = off. */
    void* rowstats;
    int64_t rowstats_parts;
} vsx_gemm_desc;

int vsx_gemm_f16(const vsx_gemm_desc* d, vsx_stream_t stream);
/* parts per row the launch described by d would write into d->rowstats (see above); the answer does not depend on
   d->rowstats / d->rowstats_parts themselves */
int64_t vsx_gemm_rowstats_parts(const vsx_gemm_desc* d);
/* bytes of `workspace` that make split-K possible for this problem (0: it would not be split) */
int64_t vsx_gemm_workspace(const vsx_gemm_desc* d);

/* ------------------------------------------------------------------------------------------
 * K3: GroupNorm over channels-last activations, two kernels.
 *   x = concat_C(x1[nimg, rows, C1], x2[nimg, rows, C2]); statistics per (img, group) over
 *   rows x (C/groups).  nimg = B and rows = F*H*W reproduces the reference's 5-D GroupNorm
 *   (resnet.py:166,177; unet.py:474: statistics pooled over frames); nimg = B*F, rows = H*W the
 *   per-frame one (attention.py:108; motion_module.py:146).
 * vsx_groupnorm_stats writes fp32 partial sums partial[img][chunk][group][2] (sum, sumsq) with
 * chunk count = vsx_groupnorm_chunks(rows, nimg); vsx_groupnorm_apply reduces them (deterministic
 * order) into stats[nimg][groups][2] = (mean, rstd) (fp32 workspace, caller-allocated) and writes
 * y = (x-mean)*rstd*gamma+beta (optionally SiLU) as one [nimg, rows, C1+C2].
 * In frame-sharded mode the caller all-reduces `partial` between the two calls and passes the
 * GLOBAL element count in `count_rows`.
 * ------------------------------------------------------------------------------------------ */
int64_t vsx_groupnorm_chunks(int64_t rows, int64_t nimg);
int vsx_groupnorm_stats(const void* x1, const void* x2, int64_t nimg, int64_t rows, int64_t C1,
                        int64_t C2, int64_t groups, float* partial, vsx_stream_t stream);
int vsx_groupnorm_apply(const void* x1, const void* x2, int64_t nimg, int64_t rows, int64_t C1,
                        int64_t C2, int64_t groups, const float* partial, int64_t nchunks,
                        int64_t count_rows, const void* gamma, const void* beta, float eps,
                        int64_t silu, float* stats, void* y, vsx_stream_t stream);

/* Row statistics of x[M, C] for a LayerNorm folded into its consumer GEMM (vsx_gemm_desc.rowscale):
 * stats[m] = (rstd_m, -rstd_m * mean_m), rstd = 1/sqrt(var + eps), fp32 statistics as in vsx_layernorm. */
int vsx_row_stats(const void* x, int64_t M, int64_t C, float eps, float* stats, vsx_stream_t stream);
/* the same stats[m] from the partial sums a producing GEMM wrote (vsx_gemm_desc.rowstats): parts [M][nparts][2] fp32 =
 * (sum, sum of squares) over disjoint column parts that cover the C columns; variance = E[x^2] - mean^2 in fp32. */
int vsx_row_stats_combine(const float* parts, int64_t M, int64_t nparts, int64_t C, float eps, float* stats,
                          vsx_stream_t stream);

/* K4: LayerNorm over the last dim of x[M, C] (attention.py:182,199,205; motion_module.py:213,219).
 * If pe != NULL, adds pe[((m / rows_per_frame) % frames) + frame_offset][c] (fp16 [max_len, C]) to
 * the normalised row: the AnimateDiff temporal positional encoding (motion_module.py:253-255,
 * 291-292) fused so that no '(b f) d c -> (b d) f c' copy is needed. */
int vsx_layernorm(const void* x, int64_t M, int64_t C, const void* gamma, const void* beta,
                  float eps, const void* pe, int64_t rows_per_frame, int64_t frames,
                  int64_t frame_offset, void* y, vsx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K5/K6: fused attention O = softmax(Q K^T * scale) V, online softmax, MFMA, never
 * materialising the scores.  Replaces F.scaled_dot_product_attention (diffusers AttnProcessor2_0,
 * the default processor of attention.py:174-194) and xformers.memory_efficient_attention
 * (edlora_util.py:61; attention_register.py:67,147).
 *   Q  [nb, nq, heads*d]  row stride ldq     K [nkvb, nk, heads*d] row stride ldk
 *   VT [nkvb, heads*d, ldvt] = V transposed per image (written by vsx_gemm_f16 c_mode 1);
 *       ldvt >= round_up(nk, 8)
 *   O  [nb, nq, heads*d]  row stride ldo
 * image b uses K/V image b / kv_div (kv_div = frames for cross-attention where the text
 * embedding is shared by all frames: the reference repeats it, attention.py:100-103).
 * d in {8,16,32,40,64,80,128,160}.
 * ------------------------------------------------------------------------------------------ */
int vsx_attention_f16(const void* Q, const void* K, const void* VT, void* O, int64_t nb,
                      int64_t heads, int64_t nq, int64_t nk, int64_t d, int64_t ldq, int64_t ldk,
                      int64_t ldvt, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t vt_bs,
                      int64_t o_bs, int64_t kv_div, float scale, vsx_stream_t stream);

/* Training form of K5/K6 (ABI 8; the adapter training step differentiates through every attention of the frozen UNet,
 * trainer_videoswap.py:33-97):
 *   vsx_attention_lse_f16: the same forward, additionally writing lse[b][h][q] (fp32, row length lse_ld >= nq) = log2 of
 *       the softmax denominator in the scaled-score domain: P[q][k] = exp2(scale * log2(e) * q.k - lse[q]).
 *   vsx_attention_bwd_f16: dQ (and, when dK / dV are given, dK and dV) from dO WITHOUT materialising the [heads, nq, nk]
 *       probabilities: P is recomputed tile by tile from lse, dS = scale * P o (dO V^T - delta), delta[q] = dO[q].O[q]
 *       (computed here into the caller's `delta` buffer), dQ = dS K, dK = dS^T Q, dV = P^T dO; two MFMA kernels, no atomics.
 *       Q, O, dO, dQ [nb, nq, heads*d]; K, V, dK, dV [nb / kv_div, nk, heads*d]: all contiguous rows.
 *       QT, dOT [nb, heads*d, ldtq], KT [nb / kv_div, heads*d, ldtk]: per-image transposes, rows zero-padded to a multiple
 *       of 8 (QT / dOT only for dK / dV); lse, delta [nb, heads, lds], lds a multiple of 64, zero beyond nq.
 *       kv_div > 1 (text K / V shared by the frames of a clip): dQ only.  d in {40, 64, 80}
 *       (vsx_attention_bwd_supported); other head dims keep the materialised path of videoswap_amd/autograd.py. */
int vsx_attention_lse_f16(const void* Q, const void* K, const void* VT, void* O, float* lse, int64_t lse_ld, int64_t nb,
                          int64_t heads, int64_t nq, int64_t nk, int64_t d, int64_t ldq, int64_t ldk, int64_t ldvt,
                          int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t vt_bs, int64_t o_bs, int64_t kv_div,
                          float scale, vsx_stream_t stream);
int64_t vsx_attention_bwd_supported(int64_t d);
int vsx_attention_bwd_f16(const void* Q, const void* K, const void* V, const void* O, const void* dO, const void* QT,
                          const void* KT, const void* dOT, const float* lse, float* delta, void* dQ, void* dK, void* dV,
                          int64_t nb, int64_t heads, int64_t nq, int64_t nk, int64_t d, int64_t ldtq, int64_t ldtk,
                          int64_t lds, int64_t kv_div, float scale, vsx_stream_t stream);

/* K7 helper: in-place row softmax of fp16 scores S[nrows, ld] over the first ncols columns
 * (diffusers Attention.get_attention_scores softmax; probs are then handed to the
 * Prompt-to-Prompt controller, attention_register.py:70-76). */
int vsx_softmax_rows(void* S, int64_t nrows, int64_t ncols, int64_t ld, vsx_stream_t stream);
/* causal variant for the CLIP text encoder (transformers CLIPAttention with causal_attention_mask): row r is query
 * r % rows_per_seq and attends to columns <= that index; the other columns get probability 0. */
int vsx_softmax_rows_causal(void* S, int64_t nrows, int64_t ncols, int64_t ld, int64_t rows_per_seq,
                            vsx_stream_t stream);

/* K8: temporal self-attention across frames at every spatial site (motion_module.py:287-338),
 * computed directly on the [B, F, HW, heads*d] layout (no transposes).  q/k/v/o share row
 * stride ld.  In frame-sharded mode q holds the local fq frames and k/v the gathered fk frames. */
int vsx_temporal_attention_f16(const void* Q, const void* K, const void* V, void* O, int64_t B,
                               int64_t fq, int64_t fk, int64_t hw, int64_t heads, int64_t d,
                               int64_t ldq, int64_t ldkv, int64_t ldo, float scale,
                               vsx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K11: element-wise glue.
 * ------------------------------------------------------------------------------------------ */
/* y = silu(x) (resnet.py:172: nonlinearity(temb)) ; n elements */
int vsx_silu(const void* x, void* y, int64_t n, vsx_stream_t stream);
/* y = x * sigmoid(1.702 x): CLIP's quick_gelu (text encoder MLP; edlora_util.py:144 runs it 16 x per prompt) */
int vsx_quick_gelu(const void* x, void* y, int64_t n, vsx_stream_t stream);
/* y = a + s*b (adapter residual add, unet_blocks.py:399-402; unet.py:434-438) */
int vsx_axpy(const void* a, const void* b, float s, void* y, int64_t n, vsx_stream_t stream);
/* Latent layout conversion at the UNet boundary.
 * pack:   x[B,Cin,F,H,W] -> y[B*F,H,W,Cpad] (channels >= Cin zero-filled), unet.py:411
 * unpack: x[B*F,H,W,Cs] (first Cout channels) -> y[B,Cout,F,H,W], unet.py:476 */
int vsx_pack_latents(const void* x, void* y, int64_t B, int64_t Cin, int64_t F, int64_t HW,
                     int64_t Cpad, vsx_stream_t stream);
int vsx_unpack_latents(const void* x, void* y, int64_t B, int64_t Cout, int64_t F, int64_t HW,
                       int64_t Cs, vsx_stream_t stream);
/* Classifier-free guidance + DDIM update in one pass (pipeline_videoswap.py:578-580,587 and the
 * diffusers DDIMScheduler.step / DDIMInverseScheduler.step arithmetic, eta = 0):
 *   eps = eps_u + g*(eps_c - eps_u)   (eps_c == NULL: eps = eps_u)
 *   x0  = (x - sqrt(1-a_t) eps)/sqrt(a_t);  out = sqrt(a_n) x0 + sqrt(1-a_n) eps
 * with fp32 arithmetic, fp16 storage. */
int vsx_cfg_ddim_step(const void* x, const void* eps_u, const void* eps_c, float guidance,
                      float alpha_t, float alpha_next, void* out, int64_t n, vsx_stream_t stream);
/* Latent blend of the Prompt-to-Prompt SpatialBlender (spatial_blend.py:142):
 * out = src + mask*(x - src); mask fp16 broadcast over channels: x,src [C, n_sp], mask [n_sp]. */
int vsx_masked_blend(const void* x, const void* src, const void* mask, void* out, int64_t C,
                     int64_t n_sp, vsx_stream_t stream);

/* K10: SparsePointAdapter scatter (adapter_model.py:25-47,112-131): for every visible point p in
 * frame f splat feat[p,:]*w onto the 4 bilinear corners (clamped to the edge, accumulating) of
 * out[f, y, x, :] (channels-last, fp16, must be zero-filled by the caller).
 * tracks [F,P,2] fp32 pixel coords (x,y); negative = invisible; selected[P] int32 0/1. */
int vsx_adapter_scatter(const float* tracks, const int32_t* selected, const void* feat, void* out,
                        int64_t F, int64_t P, int64_t C, int64_t h, int64_t w, float rate,
                        float out_scale, vsx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Launch-time instrumentation used by bench.py: when enabled, vsx_gemm_f16 brackets each of its
 * launches with hipEvents on the launch stream (at most `max_samples` launches are sampled;
 * on = k > 1 brackets every k-th launch only: the two event packets cost a few microseconds per
 * launch, 5 % of the whole loop when every launch is bracketed).
 * vsx_prof_collect synchronises those events and returns the number of sampled launches, their
 * summed duration (ms) and summed algorithmic FLOP (2*M*N*K*batch; geglu counts B's 2N rows).
 * ------------------------------------------------------------------------------------------ */
int vsx_prof_enable(int64_t on, int64_t max_samples);
/* suspend (1) / resume (0) the sampling: events cannot be recorded while a stream is being captured into a HIP graph */
int vsx_prof_pause(int64_t paused);
int vsx_prof_collect(int64_t* n_launches, double* total_ms, double* total_flop);
/* ABI 10: the same collection priced against BOTH rooflines.  Per sampled launch the algorithmic bytes are A once + weights
 * once + C once + residual once (a convolution reads every input pixel once); a launch cannot finish before
 * max(FLOP / peak_flops, bytes / peak_bytes_per_s).  floor_ms = the sum of those floors over the sampled launches (so
 * floor_ms / total_ms is the fraction of the ATTAINABLE rate the launch mix reached), byte_bound_ms = the measured time of the
 * launches whose byte floor is the larger of the two.  Like vsx_prof_collect it consumes the samples. */
int vsx_prof_collect_roofline(double peak_flops, double peak_bytes_per_s, int64_t* n_launches, double* total_ms,
                              double* total_flop, double* total_bytes, double* floor_ms, double* byte_bound_ms);

/* ------------------------------------------------------------------------------------------
 * RCCL collectives of the frame-sharded long-clip mode (the reference has no
 * counterpart: it cannot run clips longer than 24 frames, motion_module.py:237-255).  One
 * communicator per process (= per GPU).  Rank 0 calls vsx_comm_unique_id and distributes the 128
 * bytes out of band (torch.distributed store, MPI, a file); every rank then calls vsx_comm_init.
 * librccl is opened with dlopen at the first call: libvsx.so has no link-time dependency on it.
 * Collectives are asynchronous on `stream`; the caller orders them against compute with events.
 *   vsx_allgather_kv:  kv_local [batch][elems_per_batch] fp16 (the K|V rows of this rank's frames,
 *       elems_per_batch = f_local*hw*2C) -> kv_all [batch][nranks][elems_per_batch]: per batch item
 *       the ranks' frame slabs in rank order = the [B, F_total, hw, 2C] tensor
 *       vsx_temporal_attention_f16 reads with fk = F_total (K = columns [0,C), V = [C,2C), ldkv = 2C);
 *   vsx_allgather_f32: fp32 GroupNorm partial sums (vsx_groupnorm_stats) of every rank, rank order;
 *       vsx_groupnorm_apply then reduces them in a fixed order: identical statistics on every rank;
 *   vsx_allreduce_gnstats: in-place fp32 sum over the ranks;
 *   vsx_alltoall_f16: the frames <-> sites re-shard of FrameShard(exchange='sites') (everything between a motion
 *       module's proj_in and proj_out is local to a SITE, motion_module.py:138-162,222-234) as ONE group of
 *       ncclSend / ncclRecv straight from / into the strided activation layouts (no pack / unpack passes): for every
 *       peer p and block (o, i), o < nouter, i < ninner, `block_elems` contiguous fp16 elements travel from
 *           send + p*send_strides[0] + o*send_strides[1] + i*send_strides[2]   on this rank   to
 *           recv + r*recv_strides[0] + o*recv_strides[1] + i*recv_strides[2]   on rank p   (r = this rank; strides in
 *       elements).  frames -> sites ([B, f, P, hw/P, C] -> [B, P, f, hw/P, C]): nouter = B, ninner = f,
 *       block = hw/P*C, send strides (block, f*P*block, P*block), recv strides (f*block, P*f*block, block);
 *       sites -> frames is the same call with the two stride triples exchanged.
 *
 * Recording communicator (ABI 8): vsx_comm_init_recording(rank, nranks) makes this process rank `rank` of `nranks`
 * WITHOUT librccl: the collectives above run their argument checks, stride arithmetic, group structure and peer loop
 * unchanged, but every ncclSend / ncclRecv / ncclAllGather / ncclAllReduce / local copy is appended to a log instead
 * of being executed (no device work, the buffers are never dereferenced).  vsx_comm_recorded(out, capacity) drains the
 * log (out == NULL: number of pending records): 6 int64 per record = op (VSX_COMM_OP_*), peer (-1: none), source
 * offset, destination offset (ELEMENTS from the entry point's first / second buffer argument; -1: n/a), element count,
 * element size.  Replaying the logs of all ranks against each other (NCCL matching: the k-th send of rank a to rank b
 * meets the k-th receive of b from a) reproduces what a P-GPU node would do: tests/test_distributed.py checks the
 * result against FrameShard's layouts for P = 2, 4, 8 — the multi-rank marshalling of this ABI on a box with one GPU
 * or none.  vsx_comm_destroy ends the recording.
 * ------------------------------------------------------------------------------------------ */
#define VSX_COMM_OP_SEND 1
#define VSX_COMM_OP_RECV 2
#define VSX_COMM_OP_ALLGATHER 3   /* destination block of rank q at dst offset + q * count */
#define VSX_COMM_OP_ALLREDUCE 4
#define VSX_COMM_OP_LOCAL_COPY 5
#define VSX_COMM_OP_GROUP_START 6
#define VSX_COMM_OP_GROUP_END 7
int vsx_comm_unique_id(void* id128);
int vsx_comm_init(int64_t rank, int64_t nranks, const void* id128);
int vsx_comm_init_recording(int64_t rank, int64_t nranks);
int64_t vsx_comm_recorded(int64_t* out, int64_t capacity);
int64_t vsx_comm_size(void);
int64_t vsx_comm_rank(void);
int vsx_comm_destroy(void);
int vsx_allgather_kv(const void* kv_local, void* kv_all, int64_t batch, int64_t elems_per_batch,
                     vsx_stream_t stream);
int vsx_allgather_f32(const float* local, float* all, int64_t count, vsx_stream_t stream);
int vsx_allreduce_gnstats(float* partial, int64_t count, vsx_stream_t stream);
int vsx_alltoall_f16(const void* send, void* recv, int64_t nouter, int64_t ninner, int64_t block_elems,
                     const int64_t* send_strides, const int64_t* recv_strides, vsx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Gradient path of the adapter training step (trainer_videoswap.py:33-97 —
 * `accelerator.backward(loss)` through the frozen UNet into the SparsePointAdapter, adapter_model.py:70-107).
 * Only DATA gradients are needed (the UNet's weights are frozen); the matrix work of the backward pass (linear /
 * conv dgrad, attention products, the adapter MLP's weight gradients) runs on vsx_gemm_f16 with transposed /
 * flipped operands (videoswap_amd/autograd.py).  These are the element-wise / reduction passes: fp16 tensors,
 * fp32 arithmetic, fixed-order reductions (no atomics: deterministic).
 *   vsx_geglu_fwd     y2 [M, 2N] (h | g pre-activations, kept for the backward) -> out [M, N] = h * gelu_erf(g)
 *                     (diffusers GEGLU as its own pass; inference fuses it into the GEMM epilogue and drops y2)
 *   vsx_geglu_bwd     dout [M, N], y2 [M, 2N] -> dy2 [M, 2N]
 *   vsx_silu_bwd      dx = dy * silu'(x)   (adapter MLP, adapter_model.py:12-22)
 *   vsx_groupnorm_bwd data gradient of vsx_groupnorm_apply (same arguments; statistics recomputed; dy [.., C1+C2]
 *                     -> dx1 [.., C1], dx2 [.., C2]); ws: vsx_groupnorm_bwd_workspace(nimg, rows, groups) FLOATS
 *                     (chunk partial sums of a two-level reduction, ABI 6)   (resnet.py:166-177)
 *   vsx_layernorm_bwd data gradient of vsx_layernorm (the positional encoding is added after the normalisation)
 *   vsx_softmax_bwd   dS = scale * P o (dP - rowsum(dP o P)) in place over dP, rows padded to ld
 *   vsx_sum_pool2x2   [n, 2h, 2w, c] -> [n, h, w, c]: gradient of the nearest-2x upsampling folded into a conv
 *   vsx_adapter_gather gradient of vsx_adapter_scatter with respect to feat [P, C] (adapter_model.py:25-47)
 * ------------------------------------------------------------------------------------------ */
int vsx_geglu_fwd(const void* y2, void* out, int64_t M, int64_t N, vsx_stream_t stream);
int vsx_geglu_bwd(const void* dout, const void* y2, void* dy2, int64_t M, int64_t N, vsx_stream_t stream);
int vsx_silu_bwd(const void* dy, const void* x, void* dx, int64_t n, vsx_stream_t stream);
int64_t vsx_groupnorm_bwd_workspace(int64_t nimg, int64_t rows, int64_t groups);
int vsx_groupnorm_bwd(const void* dy, const void* x1, const void* x2, int64_t nimg, int64_t rows, int64_t C1,
                      int64_t C2, int64_t groups, const void* gamma, const void* beta, float eps, int64_t silu,
                      void* ws, void* dx1, void* dx2, vsx_stream_t stream);
int vsx_layernorm_bwd(const void* dy, const void* x, const void* gamma, float eps, void* dx, int64_t M, int64_t C,
                      vsx_stream_t stream);
int vsx_softmax_bwd(const void* P, void* dP, int64_t nrows, int64_t ncols, int64_t ld, float scale,
                    vsx_stream_t stream);
int vsx_sum_pool2x2(const void* x, void* y, int64_t n, int64_t h, int64_t w, int64_t c, vsx_stream_t stream);
int vsx_adapter_gather(const float* tracks, const int32_t* selected, const void* dmap, void* dfeat, int64_t F,
                       int64_t P, int64_t C, int64_t h, int64_t w, float rate, float out_scale, vsx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VSX_H_ */
