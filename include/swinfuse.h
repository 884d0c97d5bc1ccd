/*
 * swinfuse.h — C-ABI of the MI355X (gfx950) Swin-UNet image-fusion forward library.
 *
 * The reference (RainbowZL0/swin-unet-image-fusion) has no FFI layer: its boundary is the
 * Python nn.Module API.  This header is the boundary a maintainer binds underneath that API
 * (ctypes stub in INTEGRATION.md).  Each entry point names the reference interface it
 * replaces (file:line relative to the reference root).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - All tensors are DEVICE pointers, fp32, **NHWC** ("token-major": [B][H][W][C]) unless a
 *    function says NCHW.  (B,1,H,W) NCHW and NHWC coincide, so the whole-model entry takes the
 *    reference's input/output tensors as they are.
 *  - The library never allocates, frees or retains device memory: weights and workspaces are
 *    borrowed for the duration of a call.  Workspace sizes come from the
 *    *_workspace_bytes queries.
 *  - Every function enqueues on the caller's stream (hipStream_t passed as void*), never
 *    synchronises, and is safe to capture into a hipGraph.
 *  - Return value: 0 = SWF_OK, negative = swf_status.  Nothing throws or aborts.
 *    swf_last_error_string() gives a thread-local description of the last failure.
 *  - Only the eval()/no_grad forward is provided (dropout p=0 -> identity, BatchNorm running
 *    statistics).
 */
#ifndef SWINFUSE_H
#define SWINFUSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWF_VERSION_MAJOR 0
#define SWF_VERSION_MINOR 1

typedef void* swf_stream_t; /* hipStream_t */

typedef enum swf_status {
    SWF_OK = 0,
    SWF_ERR_NULL = -1,        /* required pointer is NULL */
    SWF_ERR_BAD_SHAPE = -2,   /* sizes inconsistent (e.g. map not a multiple of the window; einops error in the reference) */
    SWF_ERR_PAD = -3,         /* reflect pad >= dimension (torch RuntimeError at a006_PaddingOperation.py:128) */
    SWF_ERR_UNSUPPORTED = -4, /* configuration outside what the kernels cover */
    SWF_ERR_WORKSPACE = -5,   /* workspace too small */
    SWF_ERR_HIP = -6          /* a HIP launch failed */
} swf_status;

/* Arithmetic mode of the window-attention contractions and linear layers.
 *  FP32 : every contraction in exact fp32 (f32-input MFMA == fmaf chain; VALU attention).
 *  FAST : fused window kernels — linear layers as split-bf16 (bf16x3, fp32-grade) MFMA,
 *         QK^T in f16 MFMA, P.V in fp16 MFMA; LayerNorm statistics, softmax, residual
 *         stream and all accumulators stay fp32.  Shapes the fused kernels do not cover
 *         fall back to FP32 kernels (never to the host). */
typedef enum swf_precision { SWF_PREC_FP32 = 0, SWF_PREC_FAST = 1 } swf_precision;

typedef struct swf_linear { const float* weight; /* [out][in] row-major (nn.Linear / 1x1 Conv2d) */
                            const float* bias;   /* [out] or NULL */ } swf_linear;
typedef struct swf_norm   { const float* gamma; const float* beta; } swf_norm; /* LayerNorm over C, eps 1e-5 */

/* ---- WindowAttention (a001_WindowAttention.py:9-20 ctor, :448-474 forward) ---------------- */
typedef struct swf_attn_desc {
    int32_t channels;      /* in_out_dims */
    int32_t heads;         /* num_heads */
    int32_t head_dim;      /* dims_per_head (heads*head_dim need not equal channels) */
    int32_t win_h, win_w;  /* window_size */
    int32_t shift;         /* use_cyclic_shift: roll by (-win_h/2,-win_w/2), mask = -1e10 (a001:217-315) */
} swf_attn_desc;

typedef struct swf_attn_params {
    swf_linear q, k, v;        /* q_for_heads / k_for_heads / v_for_heads: [heads*head_dim][channels] */
    swf_linear proj;           /* linear_projection: [channels][heads*head_dim] */
    const float* bias_table;   /* relative_position_bias_table [(2*win_h-1)][(2*win_w-1)], shared by all heads (a001:72-82) */
} swf_attn_params;

/* out = WindowAttention(q, k, v) [+ residual if non-NULL].  q,k,v,out,residual: [B][H][W][C].
 * H,W must be multiples of the window (SWF_ERR_BAD_SHAPE otherwise). */
int swf_window_attention_fwd(const swf_attn_desc* desc, const swf_attn_params* p,
                             const float* q, const float* k, const float* v, const float* residual,
                             float* out, int32_t B, int32_t H, int32_t W,
                             void* workspace, size_t workspace_bytes, swf_stream_t stream);
/* The same with the arithmetic mode as an argument (swf_precision): SWF_PREC_FAST runs the four projections as split-bf16 MFMA
 * GEMMs and QK^T / P.V on the f16 MFMA attention core (same workspace query); swf_window_attention_fwd == SWF_PREC_FP32. */
int swf_window_attention_fwd_prec(const swf_attn_desc* desc, int32_t precision, const swf_attn_params* p,
                                  const float* q, const float* k, const float* v, const float* residual,
                                  float* out, int32_t B, int32_t H, int32_t W,
                                  void* workspace, size_t workspace_bytes, swf_stream_t stream);
size_t swf_window_attention_workspace_bytes(const swf_attn_desc* desc, int32_t B, int32_t H, int32_t W);

/* ---- BasicBlock (a005_BasicBlock.py:127-145) and its two halves ---------------------------- */
/* Which kernel shapes the fast tier picks where it has a choice (same arithmetic, results differ in the last bits):
 * LATENCY (default): shortest single forward — eight waves per window at level 2, 32-token MLP tiles at level 3;
 * THROUGHPUT: least CU-time per forward, for callers that keep several forwards in flight on separate streams
 * (ShardedFusion lanes) — four waves per window (two windows per CU), 64-token MLP tiles. */
typedef enum swf_schedule { SWF_SCHED_LATENCY = 0, SWF_SCHED_THROUGHPUT = 1 } swf_schedule;

typedef struct swf_block_desc {
    swf_attn_desc attn;
    int32_t hidden;        /* mlp_hidden_dims */
    int32_t cross;         /* use_cross_attr: x'=WA_x(q=x,k=y,v=y), y'=WA_y(q=y,k=x,v=x) (a002_AutoPathWinAtt.py:67-82) */
    int32_t precision;     /* swf_precision */
    int32_t schedule;      /* swf_schedule */
} swf_block_desc;

typedef struct swf_block_stream_params {   /* one modality stream of one BasicBlock */
    swf_norm ln1;              /* stage_1.norm_layer_{1|2}  (a004_AddAndLayerNormWithOtherModule.py:16-18) */
    swf_attn_params attn;      /* auto_path_win_att.window_attention_{x|y} */
    swf_norm ln2;              /* stage_2.norm_layer_{1|2} */
    swf_linear fc1, fc2;       /* auto_path_mlp.mlp_{x|y}_1 [hidden][C], mlp_{x|y}_2 [C][hidden]; ELU(alpha=1) between (a003_AutoPathMLP.py:21-44) */
} swf_block_stream_params;

/* stage_1 of a BasicBlock for both streams: out = in + Attention(LN(in), ...) (a004:29-38 with
 * other_module = AutoPathWinAtt a002:58-82).  py / y_* may be NULL for a single-path block. */
int swf_attn_halfblock_fwd(const swf_block_desc* desc, const swf_block_stream_params* px,
                           const swf_block_stream_params* py,
                           const float* x_in, const float* y_in, float* x_out, float* y_out,
                           int32_t B, int32_t H, int32_t W,
                           void* workspace, size_t workspace_bytes, swf_stream_t stream);
/* stage_2: out = in + MLP(LN(in)) per stream (a004:29-38 with other_module = AutoPathMLP a003:46-50). */
int swf_mlp_halfblock_fwd(const swf_block_desc* desc, const swf_block_stream_params* px,
                          const swf_block_stream_params* py,
                          const float* x_in, const float* y_in, float* x_out, float* y_out,
                          int32_t B, int32_t H, int32_t W,
                          void* workspace, size_t workspace_bytes, swf_stream_t stream);
/* whole BasicBlock.forward(x, y) (a005:127-145); x_out/y_out may alias x_in/y_in. */
int swf_basic_block_fwd(const swf_block_desc* desc, const swf_block_stream_params* px,
                        const swf_block_stream_params* py,
                        const float* x_in, const float* y_in, float* x_out, float* y_out,
                        int32_t B, int32_t H, int32_t W,
                        void* workspace, size_t workspace_bytes, swf_stream_t stream);
size_t swf_basic_block_workspace_bytes(const swf_block_desc* desc, int32_t B, int32_t H, int32_t W);

/* Block-level pre-pack for callers that run the same block many times (fast tier): derive the fused kernel's
 * weight images once into a caller-owned buffer of swf_basic_block_packed_bytes(desc) bytes (0 = this shape has no
 * fused kernel; use swf_basic_block_fwd), then launch with swf_basic_block_fwd_packed — exactly one kernel. */
size_t swf_basic_block_packed_bytes(const swf_block_desc* desc);
int swf_basic_block_pack(const swf_block_desc* desc, const swf_block_stream_params* px,
                         const swf_block_stream_params* py, void* packed, size_t packed_bytes, swf_stream_t stream);
int swf_basic_block_fwd_packed(const swf_block_desc* desc, const void* packed,
                               const float* x_in, const float* y_in, float* x_out, float* y_out,
                               int32_t B, int32_t H, int32_t W, swf_stream_t stream);

/* ---- training side, first stage: backward of one BasicBlock ------------------------------------------------
 * What torch.autograd computes for a005_BasicBlock.py:127-145 inside the reference's training step (a016_train.py:150-196), in
 * exact fp32.  The forward intermediates are recomputed from the block's inputs, so nothing has to be saved by the forward call.
 * gx_out / gy_out: dL/d(block outputs); gx_in / gy_in: dL/d(block inputs) (written); gpx / gpy: where the parameter gradients go —
 * the same fields as swf_block_stream_params, every non-NULL pointer is OVERWRITTEN with the gradient of that tensor (NULL = not
 * wanted).  py / y_* / gpy NULL for a single-path block.  Sums over tokens run in a fixed order (no atomics): bit-reproducible. */
typedef struct swf_linear_grad { float* weight; float* bias; } swf_linear_grad;
typedef struct swf_norm_grad { float* gamma; float* beta; } swf_norm_grad;
typedef struct swf_attn_grads { swf_linear_grad q, k, v, proj; float* bias_table; } swf_attn_grads;
typedef struct swf_block_stream_grads {
    swf_norm_grad ln1; swf_attn_grads attn; swf_norm_grad ln2; swf_linear_grad fc1, fc2;
} swf_block_stream_grads;
typedef struct swf_patch_grads { swf_linear_grad conv; swf_norm_grad ln; } swf_patch_grads;
size_t swf_basic_block_bwd_workspace_bytes(const swf_block_desc* desc, int32_t B, int32_t H, int32_t W);
int swf_basic_block_bwd(const swf_block_desc* desc, const swf_block_stream_params* px, const swf_block_stream_params* py,
                        const float* x_in, const float* y_in, const float* gx_out, const float* gy_out,
                        float* gx_in, float* gy_in, const swf_block_stream_grads* gpx, const swf_block_stream_grads* gpy,
                        int32_t B, int32_t H, int32_t W, void* workspace, size_t workspace_bytes, swf_stream_t stream);

/* The inner modules on their own under autograd (a caller that keeps the reference's BasicBlock and swaps only a001 / a003 / a004):
 * exact fp32, forward intermediates recomputed, fixed-order sums.  Gradient pointers as above (NULL = not wanted).
 * WindowAttention.forward (a001:448-474): q / k / v [B][H][W][C]; gq / gk / gv are three separate buffers — where one tensor was passed
 * as more than one of q, k, v the caller adds them (torch.autograd does). */
size_t swf_window_attention_bwd_workspace_bytes(const swf_attn_desc* desc, int32_t B, int32_t H, int32_t W);
int swf_window_attention_bwd(const swf_attn_desc* desc, const swf_attn_params* p, const float* q, const float* k, const float* v,
                             const float* gout, float* gq, float* gk, float* gv, const swf_attn_grads* gp,
                             int32_t B, int32_t H, int32_t W, void* workspace, size_t workspace_bytes, swf_stream_t stream);
/* one stream of AutoPathMLP.forward (a003:46-50): out = fc2(ELU(fc1(x))), x [tokens][channels] */
size_t swf_mlp_bwd_workspace_bytes(int64_t tokens, int32_t channels, int32_t hidden);
int swf_mlp_bwd(const swf_linear* fc1, const swf_linear* fc2, const float* x, const float* gout, float* gx,
                const swf_linear_grad* gfc1, const swf_linear_grad* gfc2, int64_t tokens, int32_t channels, int32_t hidden,
                void* workspace, size_t workspace_bytes, swf_stream_t stream);
/* my_layer_norm (a004:54-72) */
size_t swf_layernorm_bwd_workspace_bytes(int64_t tokens, int32_t C);
int swf_layernorm_bwd(const swf_norm* ln, const float* x, const float* gout, float* gx, const swf_norm_grad* gp,
                      int64_t tokens, int32_t C, void* workspace, size_t workspace_bytes, swf_stream_t stream);

/* SelfAndCrossBlockPair.forward (a012_SelfAndCrossBlockPair.py:70-78): four BasicBlocks in the
 * order self/normal, self/shifted, cross/normal, cross/shifted (a009:90-109).  `desc` gives the
 * shared dims; shift/cross flags inside it are ignored.  px[4], py[4]. */
int swf_block_pair4_fwd(const swf_block_desc* desc, const swf_block_stream_params* px,
                        const swf_block_stream_params* py,
                        const float* x_in, const float* y_in, float* x_out, float* y_out,
                        int32_t B, int32_t H, int32_t W,
                        void* workspace, size_t workspace_bytes, swf_stream_t stream);

/* ---- PatchMergingAndLinearLayer (a011_PatchOperation.py:244-264) + MyPadding (a006:167-187) -- */
typedef struct swf_patch_params { swf_linear conv; /* mlp_layer_{x|y}: 1x1 conv */ swf_norm ln; /* layer_norm_{x|y} */ } swf_patch_params;

/* Encoder stage front: reflect-pad in (H,W) bottom/right to a multiple of (merge_h,merge_w)
 * (a006:122-131), space-to-depth with channel order (ph*merge_w+pw)*Cin+c (a011:87-93), 1x1 conv
 * 4Cin->Cout, LayerNorm(Cout), ELU (a011:236-239), then reflect-pad the merged map to a multiple
 * of (win_h,win_w).  in: [B][H][W][Cin]; out: [B][Ho][Wo][Cout] with Ho,Wo from swf_merge_out_shape.
 * SWF_ERR_PAD when a reflect pad is >= the dimension it pads (the reference raises RuntimeError). */
int swf_patch_merge_fwd(const swf_patch_params* p, const float* in, float* out,
                        int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                        int32_t merge_h, int32_t merge_w, int32_t win_h, int32_t win_w,
                        void* workspace, size_t workspace_bytes, swf_stream_t stream);
int swf_merge_out_shape(int32_t H, int32_t W, int32_t merge_h, int32_t merge_w, int32_t win_h, int32_t win_w,
                        int32_t* Hm, int32_t* Wm, int32_t* Ho, int32_t* Wo);

/* Decoder stage back: crop the window padding (in is [B][Hp][Wp][Cin], valid part Hm x Wm,
 * a006:143-146), 1x1 conv Cin->mh*mw*Cout, LayerNorm over mh*mw*Cout, depth-to-space (a011:111-117),
 * ELU (order a011:241), crop to (Hout,Wout) (undo of the merge padding), then optionally add
 * `skip` [B][Hout][Wout][Cout] (the U-Net skip add that precedes the next decoder stage,
 * a013_ModelDefinition.py:222-225). */
int swf_patch_unmerge_fwd(const swf_patch_params* p, const float* in, const float* skip, float* out,
                          int32_t B, int32_t Hp, int32_t Wp, int32_t Hm, int32_t Wm, int32_t Cin, int32_t Cout,
                          int32_t merge_h, int32_t merge_w, int32_t Hout, int32_t Wout,
                          void* workspace, size_t workspace_bytes, swf_stream_t stream);
size_t swf_patch_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                 int32_t merge_h, int32_t merge_w, int32_t win_h, int32_t win_w, int32_t encoder);

/* ---- final fusion head (a013:126-152): cat -> conv kxk reflect -> BatchNorm2d(eval) -> ELU -> conv kxk reflect */
typedef struct swf_head_params {
    const float* conv1_w; const float* conv1_b;   /* final_layer.0: [2][2][k][k], [2] */
    const float* bn_gamma; const float* bn_beta; const float* bn_mean; const float* bn_var; /* final_layer.1, eps 1e-5 */
    const float* conv2_w; const float* conv2_b;   /* final_layer.3: [1][2][k][k], [1] */
} swf_head_params;
/* x,y,out: [B][H][W] (single channel).  tmp workspace: 2*B*H*W floats. */
int swf_final_head_fwd(const swf_head_params* p, const float* x, const float* y, float* out,
                       int32_t B, int32_t H, int32_t W, int32_t ksize,
                       void* workspace, size_t workspace_bytes, swf_stream_t stream);

/* Backward of the final head with BatchNorm in eval mode (running statistics: a per-channel affine map).  gout: dL/d(out) [B][H][W];
 * gx / gy: dL/d(x), dL/d(y); gp: parameter gradients (overwritten; NULL pointers skipped; the running statistics have none).  Reads
 * the eight BatchNorm scalars back to the host: synchronises the stream once. */
typedef struct swf_head_grads {
    float* conv1_w; float* conv1_b; float* bn_gamma; float* bn_beta; float* conv2_w; float* conv2_b;
} swf_head_grads;
size_t swf_final_head_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t ksize);
/* batch_stats != 0: p->bn_mean / bn_var hold the BATCH statistics of this forward (training-mode BatchNorm, as the reference trains:
 * a016:137 model.train()), and the normalisation's own gradient is included. */
int swf_final_head_bwd(const swf_head_params* p, const float* x, const float* y, const float* gout, float* gx, float* gy,
                       const swf_head_grads* gp, int32_t B, int32_t H, int32_t W, int32_t ksize, int32_t batch_stats,
                       void* workspace, size_t workspace_bytes, swf_stream_t stream);
/* Training-mode BatchNorm2d of the head (a013:133 under model.train()): the batch mean and biased variance of conv1(cat(x, y)) per
 * channel -> mean[2], var[2] (device; pass them as bn_mean / bn_var of swf_final_head_fwd / _bwd), and, when non-NULL,
 * running = (1 - momentum) running + momentum (mean, unbiased variance) as nn.BatchNorm2d does.  Workspace: swf_final_head_bwd's. */
int swf_final_head_batch_stats(const swf_head_params* p, const float* x, const float* y, float* mean, float* var,
                               float* running_mean, float* running_var, float momentum,
                               int32_t B, int32_t H, int32_t W, int32_t ksize,
                               void* workspace, size_t workspace_bytes, swf_stream_t stream);

/* ---- training side, second stage: patch layers, padding, skip add ------------------------------------------------------------ */
/* Backward of one stream of PatchMergingAndLinearLayer (a011:244-264) as the module runs it (no padding inside: MyPadding is its own
 * module).  H x W = the layer's INPUT map: encoder the full map (divisible by the merging size), decoder the merged map.  in: the
 * forward input [B][H][W][Cin]; gout: dL/d(output) (encoder [B][H/mh][W/mw][Cout], decoder [B][H*mh][W*mw][Cout]); gin: dL/d(input);
 * gp: parameter gradients (overwritten; NULL pointers skipped). */
size_t swf_patch_layer_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t merge_h, int32_t merge_w,
                                           int32_t encoder);
int swf_patch_layer_bwd(const swf_patch_params* p, const float* in, const float* gout, float* gin, const swf_patch_grads* gp,
                        int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t merge_h, int32_t merge_w, int32_t encoder,
                        void* workspace, size_t workspace_bytes, swf_stream_t stream);
/* Adjoint of swf_reflect_pad_fwd (MyPadding encoder side, a006:122-131): gout [B][H+pad_h][W+pad_w][C] -> gin [B][H][W][C]. */
int swf_reflect_pad_bwd(const float* gout, float* gin, int32_t B, int32_t H, int32_t W, int32_t C, int32_t pad_h, int32_t pad_w,
                        swf_stream_t stream);
/* out = a + b (the decoder's skip connection, a013:222-225, when the stages run as separate modules under autograd). */
int swf_add_fwd(const float* a, const float* b, float* out, int64_t count, swf_stream_t stream);

/* AutoPathMLP.forward (a003_AutoPathMLP.py:46-50) on NHWC tokens: out = fc2(ELU(fc1(in))) per stream, no norm, no residual
 * (px / py: only fc1 and fc2 are read; py / y_* NULL for a single path).  SWF_PREC_FAST runs the level-0 width (24 channels,
 * hidden 96 or 4) as ONE launch of the fused block kernel's MLP half, other shapes as split-bf16 GEMMs. */
size_t swf_mlp_workspace_bytes(int32_t precision, int64_t tokens, int32_t channels, int32_t hidden);
int swf_mlp_fwd(int32_t precision, const swf_block_stream_params* px, const swf_block_stream_params* py,
                const float* x_in, const float* y_in, float* x_out, float* y_out,
                int64_t tokens, int32_t channels, int32_t hidden,
                void* workspace, size_t workspace_bytes, swf_stream_t stream);

/* ---- token-level pieces used by the inner reference modules ------------------------------------ */
/* out[tokens][n_out] = act(in[tokens][n_in] . W^T + b) (+ residual); act 0 = none, 1 = ELU(alpha=1).
 * nn.Linear (a001:42-61) and 1x1 nn.Conv2d on NHWC tokens (a003:21-22, a011:60-63).  Exact fp32. */
int swf_linear_fwd(const swf_linear* lin, const float* in, const float* residual, float* out,
                   int64_t tokens, int32_t n_in, int32_t n_out, int32_t act, swf_stream_t stream);
/* The same with the arithmetic mode as an argument: SWF_PREC_FAST = split-bf16 (bf16x3) MFMA, fp32 accumulate (deterministic
 * split-K for deep K: workspace from swf_linear_workspace_bytes, 0 for SWF_PREC_FP32). */
size_t swf_linear_workspace_bytes(int32_t precision, int64_t tokens, int32_t n_in, int32_t n_out);
int swf_linear_fwd_prec(const swf_linear* lin, int32_t precision, const float* in, const float* residual, float* out,
                        int64_t tokens, int32_t n_in, int32_t n_out, int32_t act,
                        void* workspace, size_t workspace_bytes, swf_stream_t stream);
/* my_layer_norm (a004:54-72): LayerNorm over C of [tokens][C], eps 1e-5; elu != 0 applies ELU after. */
int swf_layernorm_fwd(const swf_norm* ln, const float* in, float* out, int64_t tokens, int32_t C, int32_t elu,
                      swf_stream_t stream);
/* MyPadding encoder side (a006:122-131): reflect-pad bottom/right by (pad_h,pad_w); [B][H][W][C] ->
 * [B][H+pad_h][W+pad_w][C].  An NCHW tensor is passed as B*C maps with C=1.  SWF_ERR_PAD if pad >= dim. */
int swf_reflect_pad_fwd(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C,
                        int32_t pad_h, int32_t pad_w, swf_stream_t stream);
/* MyPadding decoder side (a006:133-146): top-left crop [B][Hp][Wp][C] -> [B][H][W][C]. */
int swf_crop_fwd(const float* in, float* out, int32_t B, int32_t Hp, int32_t Wp, int32_t H, int32_t W, int32_t C,
                 swf_stream_t stream);

/* ---- layout helpers for the NCHW module API (a007_utils.py:7-26 are the reference's permutes) */
int swf_nchw_to_nhwc(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W, swf_stream_t stream);
int swf_nhwc_to_nchw(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W, swf_stream_t stream);

/* ---- whole model: MyModel.forward(in_x, in_y) (a013:209-230) -------------------------------- */
#define SWF_MAX_LEVELS 8
typedef struct swf_model_desc {
    int32_t levels;                      /* len(in_dims_list) */
    int32_t in_dims[SWF_MAX_LEVELS];     /* in_dims_list  (a013:22) */
    int32_t out_dims[SWF_MAX_LEVELS];    /* out_dims_list (a013:23) */
    int32_t heads;                       /* att_num_heads */
    int32_t head_dim[SWF_MAX_LEVELS];    /* floor(out_dims[j]*att_dims_per_head_ratio) (a013:174,191) */
    int32_t mlp_ratio;                   /* mlp_hidden_dims_ratio: encoder hidden = out_dims[j]*ratio (a013:177), decoder hidden = in_dims[j]*ratio (a013:196) */
    int32_t win_h, win_w, merge_h, merge_w;
    int32_t head_ksize;                  /* final_conv_layer_kernel_size */
    int32_t precision;                   /* swf_precision */
    int32_t schedule;                    /* swf_schedule */
} swf_model_desc;

/* The weights live in ONE fp32 device arena whose layout the library defines.  Parameter i has
 * the reference's canonical state_dict key (e.g.
 * "encoder_list.0.3.self_att_block.normal_window_block.auto_path_win_att.window_attention_x.q_for_heads.weight"),
 * an element offset into the arena and an element count; the host copies each tensor there once. */
int32_t swf_model_param_count(const swf_model_desc* desc);
int swf_model_param_info(const swf_model_desc* desc, int32_t index, char* name_buf, size_t name_buf_len,
                         int64_t* offset_elems, int64_t* numel);
int64_t swf_model_arena_elems(const swf_model_desc* desc);
size_t swf_model_workspace_bytes(const swf_model_desc* desc, int32_t B, int32_t H, int32_t W);
/* ir, vis, out: [B][1][H][W] == [B][H][W][1] fp32.  out is NOT clamped (callers clamp: a016:153, a017:83). */
int swf_model_forward(const swf_model_desc* desc, const float* arena, const float* ir, const float* vis,
                      float* out, int32_t B, int32_t H, int32_t W,
                      void* workspace, size_t workspace_bytes, swf_stream_t stream);

/* Optional: derive the fused kernels' weight images (split-bf16, pre-scaled Wq, masked bias matrices) ONCE
 * instead of per call.  `packed` is a caller-owned device buffer of swf_model_packed_bytes(desc) bytes
 * (0 when no level of this model uses a fused kernel); repack after the arena changes. */
size_t swf_model_packed_bytes(const swf_model_desc* desc);
int swf_model_pack_weights(const swf_model_desc* desc, const float* arena, void* packed, size_t packed_bytes,
                           swf_stream_t stream);
int swf_model_forward_packed(const swf_model_desc* desc, const float* arena, const void* packed,
                             const float* ir, const float* vis, float* out, int32_t B, int32_t H, int32_t W,
                             void* workspace, size_t workspace_bytes, swf_stream_t stream);

/* First-forward variant: additionally reports, per cross-attention block, whether its two input streams were identical
 * everywhere — the condition on which the reference prints a message and calls exit() on a model's first forward
 * (a005_BasicBlock.py:98-118, `(x == y).all()`).  cross_equal_flags: device int32[4 * levels], index
 * 2*stage + {0: plain-window cross block, 1: shifted-window cross block}, stages = encoder 0..levels-1 then decoder
 * 0..levels-1; 1 = identical inputs, 0 = inputs differ.  `packed` may be NULL (weights are then packed per call).  The
 * caller reads the flags back after synchronising and raises; the forward itself always completes. */
int swf_model_forward_checked(const swf_model_desc* desc, const float* arena, const void* packed,
                              const float* ir, const float* vis, float* out, int32_t B, int32_t H, int32_t W,
                              void* workspace, size_t workspace_bytes, int32_t* cross_equal_flags, swf_stream_t stream);
/* Measurement entry (bench.py's per-level roofline): swf_model_forward_packed with HIP events recorded on `stream` between
 * the stages of a013:209-230.  Unlike every other entry it SYNCHRONISES the stream (and so cannot be graph-captured), then
 * writes the elapsed milliseconds of the 4 * levels + 1 segments to the host array seg_ms: for s = 0 .. levels-1
 * [2s] = the patch-merging layer of encoder stage s (a011), [2s + 1] = its four BasicBlocks (a012); for j = 0 .. levels-1
 * [2 levels + 2j] = the four BasicBlocks of decoder stage j (level levels-1-j), [2 levels + 2j + 1] = its un-merging layer (with
 * the skip add); [4 levels] = the final head (a013:126-152).  seg_count = capacity of seg_ms. */
int swf_model_forward_profiled(const swf_model_desc* desc, const float* arena, const void* packed,
                               const float* ir, const float* vis, float* out, int32_t B, int32_t H, int32_t W,
                               void* workspace, size_t workspace_bytes, float* seg_ms, int32_t seg_count, swf_stream_t stream);
/* *flag (device int32) = 1 if a[i] == b[i] for every i < count, else 0 (torch semantics: NaN != NaN).  The first-call
 * test of BasicBlock / SelfAndCrossBlockPair (a005:111-113) without a device->host copy of the tensors. */
int swf_tensors_equal(const float* a, const float* b, int64_t count, int32_t* flag, swf_stream_t stream);

/* ---- colour-space steps either side of the model in the reference's inference script ------
 * a015_dataset.py:73-93 / a017_test.py:68,83-88.  OpenCV's 8-bit fixed-point BGR->YCrCb and float YCrCb->RGB,
 * restated from its published formulas (cv2 itself is not available to this build). */
/* bgr [B][H][W][3] uint8 (cv2.imread layout) -> y [B][1][H][W], crcb [B][2][H][W], float32 = uint8/255 */
int swf_bgr8_to_ycrcb_fwd(const uint8_t* bgr, float* y, float* crcb, int32_t B, int32_t H, int32_t W, swf_stream_t stream);
/* IR gray uint8 -> float32 / 255 (v2.ToDtype(scale=True), a015:57-60) */
int swf_gray8_to_unit_fwd(const uint8_t* gray, float* out, int64_t count, swf_stream_t stream);
/* fused_y [B][1][H][W] (unclamped model output), crcb [B][2][H][W] -> clamp(Y,0,1), YCrCb->RGB:
 * rgb_f [B][3][H][W] float32 (may be NULL) and/or rgb8 [B][H][W][3] uint8 quantised like torchvision save_image */
int swf_ycrcb_to_rgb_fwd(const float* fused_y, const float* crcb, float* rgb_f, uint8_t* rgb8,
                         int32_t B, int32_t H, int32_t W, swf_stream_t stream);

/* ---- misc ------------------------------------------------------------------------------------ */
int swf_version(void);                     /* major*1000 + minor */
const char* swf_last_error_string(void);   /* thread-local, never NULL */
const char* swf_status_string(int status);

#ifdef __cplusplus
}
#endif
#endif /* SWINFUSE_H */
