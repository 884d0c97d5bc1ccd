// Patch layers of the deep levels (fast tier), one launch each (kernels_deeppatch.hip): gather -> conv (bf16x3 MFMA, weights
// pre-packed fragment-major) -> LayerNorm -> ELU (-> depth-to-space scatter + skip).  Shapes: 2x2 merging with
// (K, N) = (384, 192), (768, 384) in the encoder and (384, 768), (192, 384) in the decoder; everything else stays on
// launch_patch_rr / launch_patch_fused / the GEMM path.
#pragma once
#include "kernels_patch.h"

namespace swf {

bool deep_patch_supported(int decoder, int Cin, int Cout, int mh, int mw);
// bytes of the packed conv weight of ONE stream of one layer (0 = shape not covered)
size_t deep_patch_packed_bytes(int decoder, int Cin, int Cout, int mh, int mw);
int pack_deep_patch(int decoder, int Cin, int Cout, int mh, int mw, const float* weight, void* dst, hipStream_t stream);
// The two layers next to the deepest level run conv + bias only, over column slices (deep_patch_raw): the caller passes
// raw_out[s] = [M][N] fp32 and runs LayerNorm (+ scatter) on it as a second launch.
bool deep_patch_raw(int decoder, int Cin, int Cout, int mh, int mw);
// Optional extras of the whole-row (non-raw) layers.  Encoder: LayerNorm of the finished rows with ln_gamma / ln_beta (the next
// block's LN1) written as split-bf16 planes [M][N] (all four pointers of a stream, or none).  Decoder: packed weights to touch at
// the end of the launch (L2 warm-up for the block that runs next), warm_bytes per stream.
struct DeepPatchExtra {
    const float* ln_gamma[2]; const float* ln_beta[2]; unsigned short* ln_hi[2]; unsigned short* ln_lo[2];
    const void* warm[2]; size_t warm_bytes;
};
// d as for launch_patch_fused (d.w is not read); packed[s]: the stream's image written by pack_deep_patch
int launch_deep_patch(const PatchFusedDesc& d, const void* const* packed, int nstream, hipStream_t stream, float* const* raw_out = nullptr,
                      const DeepPatchExtra* extra = nullptr);

// Second launch of a column-sliced layer: LayerNorm (+ ELU, + depth-to-space scatter and skip in the decoder) of the conv rows in
// raw[s], and — when extra carries them — the next block's LN1 of the finished rows / pixels as split-bf16 planes (encoder: N = 384
// rows; decoder: N = 768 -> pixels of 192 channels).  d as for launch_deep_patch.
int launch_deep_patch_finish(const PatchFusedDesc& d, float* const* raw, int nstream, hipStream_t stream, const DeepPatchExtra* extra = nullptr);

// Q/K/V projections of a level-4 block (C = heads * head_dim = 384) on the same kernel: LayerNorm planes [M][384] in, the attention
// core's fp16 operands out (Q pre-scaled by qscale).  Replaces launch_gemm_sp(..., SP_EPI_QKV16).
struct DeepQkvArgs {
    const unsigned short* xn_hi[2]; const unsigned short* xn_lo[2];   // LayerNorm planes (bf16 hi / lo) of each stream
    const unsigned short* w_hi[2]; const unsigned short* w_lo[2];     // Wq | Wk | Wv stacked [3 C][C], fragment-major (DeepWeights::qkvf_*)
    const float* bias[2][3];                                          // q, k, v bias or nullptr
    unsigned short* out[2][3];                                        // fp16 [M][C] each
    float qscale; int cross, M;                                       // cross: K and V of stream s read stream 1 - s
    int C;                                                            // 384, or 192 (16x16-window levels: the fused Q/K/V + attention kernel covers the rest)
};
bool deep_qkv_supported(const swf_block_desc& d);
int launch_deep_qkv(const DeepQkvArgs& q, int nstream, hipStream_t stream);

// Output projection of a deep block (C = heads * head_dim = 192 or 384) on the same kernel: attention output planes [M][C] in,
// out = res + bias + O . Wp^T as fp32 rows.  Replaces the projection launch_gemm_sp where attn_proj_kernel / the folded projection do not apply.
struct DeepProjArgs {
    const unsigned short* o_hi[2]; const unsigned short* o_lo[2];    // attention output planes (bf16 hi / lo)
    const unsigned short* w_hi[2]; const unsigned short* w_lo[2];    // Wproj, fragment-major (DeepWeights::pf_*)
    const float* bias[2]; const float* res[2]; float* out[2];        // bias [C] or nullptr; residual rows; result rows (may alias res)
    int M, C;
};
bool deep_proj_supported(const swf_block_desc& d);
int launch_deep_proj(const DeepProjArgs& q, int nstream, hipStream_t stream);

}  // namespace swf
