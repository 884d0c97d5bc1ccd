// Deep-level fast tier, attention half in ONE launch: Q/K/V projections + window attention (kernels_qkvattn.hip).
// Input: the LayerNorm planes of both streams (split bf16, [tokens][C]); output: the attention output O as split bf16
// planes [tokens][heads * head_dim], the input format of the projection GEMM (kernels_deep.h).
#pragma once
#include "kernels_deep.h"

namespace swf {

bool qkvattn_supported(const swf_block_desc& d);
// bytes of this kernel's section of the packed image of ONE stream (appended to the deep-level image): Q/K/V weights as
// fragment-major split-bf16 images in per-head virtual-channel order, the bias vectors in that order, the relative-position
// bias matrix in accumulator order, the output projection as B fragments per (head group, 32-channel tile)
size_t qkvattn_packed_bytes(const swf_block_desc& d);
int pack_qkvattn(const swf_block_desc& d, const swf_block_stream_params& p, void* dst, hipStream_t stream);

struct QkvAttnArgs {
    const void* packed[2];                                 // qkvattn sections of the two streams
    const bf16_raw* xn_hi[2]; const bf16_raw* xn_lo[2];     // LN1 planes [B*H*W][C]
    bf16_raw* o_hi[2]; bf16_raw* o_lo[2];                   // attention output planes [B*H*W][heads*head_dim]
    // optional (all or none): [head group 2][stream] fp32 [B*H*W][C].  When given, the kernel also applies the output projection
    // (weights only: no bias, no residual) and writes each head group's partial sum here INSTEAD of the O planes; the consumer
    // adds x + proj bias + part[0] + part[1] in that order (the fused MLP kernel's prologue)
    float* part[2][2];
    int B, H, W, shift, cross;
};
int launch_qkvattn(const swf_block_desc& d, const QkvAttnArgs& a, int nstream, hipStream_t stream);

}  // namespace swf
