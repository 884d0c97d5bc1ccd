// Backward of one BasicBlock in exact fp32 (kernels_bwd.hip): SURVEY.md section 8(f) rank 4, first stage.
#pragma once
#include "swf_common.h"

namespace swf {

size_t basic_block_bwd_ws(const swf_block_desc& d, int nstream, int B, int H, int W);
int basic_block_bwd(const swf_block_desc& d, const swf_block_stream_params* px, const swf_block_stream_params* py, const float* x_in,
                    const float* y_in, const float* gx_out, const float* gy_out, float* gx_in, float* gy_in, const swf_block_stream_grads* gx,
                    const swf_block_stream_grads* gy, int B, int H, int W, void* workspace, size_t workspace_bytes, hipStream_t stream);

// the inner modules on their own: WindowAttention (a001), one MLP stream (a003), one LayerNorm (a004)
size_t window_attention_bwd_ws(const swf_attn_desc& d, int B, int H, int W);
int window_attention_bwd(const swf_attn_desc& d, const swf_attn_params& p, const float* q_in, const float* k_in, const float* v_in, const float* gout,
                         float* gq, float* gk, float* gv, const swf_attn_grads* gp, int B, int H, int W, void* workspace, size_t workspace_bytes,
                         hipStream_t stream);
size_t mlp_bwd_ws(int64_t N, int C, int hid);
int mlp_bwd(const swf_linear& fc1, const swf_linear& fc2, const float* x, const float* gout, float* gx, const swf_linear_grad* g1, const swf_linear_grad* g2,
            int64_t N, int C, int hid, void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t layernorm_bwd_ws(int64_t N, int C);
int layernorm_bwd(const swf_norm& ln, const float* x, const float* gout, float* gx, const swf_norm_grad* gp, int64_t N, int C, void* workspace,
                  size_t workspace_bytes, hipStream_t stream);

// PatchMergingAndLinearLayer (one stream; H x W = the layer's input map), reflect pad, elementwise add
size_t patch_bwd_ws(int B, int H, int W, int Cin, int Cout, int mh, int mw, int encoder);
int patch_bwd(const swf_patch_params& p, const float* in, const float* gout, float* gin, const swf_patch_grads* gp, int B, int H, int W, int Cin,
              int Cout, int mh, int mw, int encoder, void* workspace, size_t workspace_bytes, hipStream_t stream);
int reflect_pad_bwd(const float* g, float* dx, int B, int H, int W, int C, int ph, int pw, hipStream_t stream);
int add_tensors(const float* a, const float* b, float* out, int64_t n, hipStream_t stream);

// final head (BatchNorm in eval mode)
size_t head_bwd_ws(int B, int H, int W, int ks);
int head_bwd(const swf_head_params& p, const float* x, const float* y, const float* gout, float* gx, float* gy, const swf_head_grads* gp, int B,
             int H, int W, int ks, int batch_stats, void* workspace, size_t workspace_bytes, hipStream_t stream);
// batch mean / biased variance of conv1's two output channels (BatchNorm2d in training mode, a013:133) + the running-statistics update
int head_batch_stats(const swf_head_params& p, const float* x, const float* y, float* mean, float* var, float* running_mean, float* running_var,
                     float momentum, int B, int H, int W, int ks, void* workspace, size_t workspace_bytes, hipStream_t stream);

}  // namespace swf
