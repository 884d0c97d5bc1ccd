// Fast tier for the deep levels (C >= 128: few tokens, large weights): the linears of a BasicBlock as
// separate GEMMs on PRE-SPLIT bf16 operands.  Every GEMM input is written by its producer (LayerNorm,
// attention core, fc1 epilogue, weight pack) as two bf16 planes hi = bf16(v), lo = bf16(v - hi), so the
// GEMM stages 16-byte chunks straight into LDS without conversion work and runs the bf16x3 scheme
// (lo.hi + hi.lo + hi.hi, fp32 accumulate) on v_mfma_f32_32x32x16_bf16 with 128x128 / 128x64 / 64x64 tiles.
#pragma once
#include "kernels_generic.h"

namespace swf {

using bf16_raw = unsigned short;   // storage type of a bf16 plane element in host-visible signatures

struct SpGemmProb {
    const bf16_raw* a_hi; const bf16_raw* a_lo;   // activations [M][K]
    const bf16_raw* w_hi; const bf16_raw* w_lo;   // weights [N][K] (nn.Linear layout)
    const float* bias;                            // [N] or nullptr
    const float* res;                             // SP_EPI_F32: [M][ldo] added to the result, or nullptr
    float* out;                                   // SP_EPI_F32: [M][ldo]
    bf16_raw* o_hi; bf16_raw* o_lo;               // SP_EPI_ELU_SPLIT: planes [M][N] of ELU(acc + bias)
};
struct SpGemmBatch {
    SpGemmProb p[kMaxProb];
    float* scratch = nullptr;          // split-K partials (only used when K >= kSpSplitKMinK)
    int64_t scratch_floats = 0;
    float qscale = 1.0f;               // SP_EPI_QKV16
};
// SP_EPI_QKV16: the operands of the attention core in their final formats, written to o_hi [M][N] (16-bit elements):
// problem p % 3 == 0 -> fp16((acc + bias) * batch.qscale) (Q, scale = d^-0.5 * log2 e), == 1 -> fp16 (K), == 2 -> fp16 (V)
enum { SP_EPI_F32 = 0, SP_EPI_ELU_SPLIT = 1, SP_EPI_QKV16 = 2 };

bool gemm_sp_supported(int N, int K);
// K slices as a function of K alone (batch shards must stay bit-identical): floats of scratch = slices*nprob*M*N
int gemm_sp_splitk_for(int K, int epi);
int launch_gemm_sp(const SpGemmBatch& batch, int nprob, int M, int N, int K, int ldo, int epi, hipStream_t stream);

// fp32 [n] -> planes hi[n], lo[n]
int launch_split_planes(const float* src, bf16_raw* hi, bf16_raw* lo, int64_t n, hipStream_t stream);

// bytes of the pre-split weight image of ONE stream of one block:
// planes hi|lo of [Wq | Wk | Wv | Wproj | Wfc1 | Wfc2], each in nn.Linear layout, then (when the fused MLP kernel
// covers the shape) fragment-major copies of Wfc1 and Wfc2 (and of Wproj where kernels_attnproj.hip covers the shape)
size_t deep_block_packed_bytes(const swf_block_desc& d);
bool deep_block_supported(const swf_block_desc& d);
int pack_deep_block(const swf_block_desc& d, const swf_block_stream_params& p, void* packed, hipStream_t stream);

struct DeepWeights {   // views into one packed image
    const bf16_raw *q_hi, *q_lo, *k_hi, *k_lo, *v_hi, *v_lo, *p_hi, *p_lo, *w1_hi, *w1_lo, *w2_hi, *w2_lo;
    // fc1 / fc2 again in MFMA-fragment-major order for the fused MLP kernel (kernels_mlp.hip), or nullptr:
    // block (row tile rt of 32 rows, k16 step ks) = 64 lanes x 8 bf16, lane = 32*hf + r holds row 32rt+r, k = 16ks+8hf..+7
    const bf16_raw *w1f_hi, *w1f_lo, *w2f_hi, *w2f_lo;
    const bf16_raw *qkvf_hi, *qkvf_lo;   // Wq | Wk | Wv stacked, fragment-major (launch_deep_qkv, kernels_deeppatch.h), or nullptr
    const bf16_raw *pf_hi, *pf_lo;   // Wproj in the same fragment-major order for the attention + projection kernel (kernels_attnproj.hip), or nullptr
    const void* qa;   // section of the fused Q/K/V + attention kernel (kernels_qkvattn.h), or nullptr
};
DeepWeights deep_block_views(const swf_block_desc& d, const void* packed);

}  // namespace swf
