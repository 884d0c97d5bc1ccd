// Fused fast tier: one launch = one BasicBlock (both modality streams) on LDS-resident window tiles.
#pragma once
#include "swf_common.h"

namespace swf {

// true when the fused window-block kernel covers this (dims, map) combination
bool window_block_supported(const swf_block_desc& d, int B, int H, int W);
// bytes of the packed (kernel-layout) weights of ONE stream of one block; 0 if unsupported
size_t window_block_packed_bytes(const swf_block_desc& d);
// workspace needed by the block-level entry (packs both streams per call)
size_t window_block_workspace_bytes(const swf_block_desc& d, int B, int H, int W);
// true: launch_window_block needs x_out / y_out distinct from x_in / y_in (the caller routes in-place calls through a temporary)
bool window_block_out_of_place(const swf_block_desc& d);

// fp32 parameters -> kernel layout (split-bf16 hi/lo weight images, pre-scaled Wq, the four
// masked/unmasked relative-position bias matrices).  One launch for both streams.
int pack_window_block(const swf_block_desc& d, const swf_block_stream_params& px, const swf_block_stream_params& py,
                      void* packed_x, void* packed_y, hipStream_t stream);

int launch_window_block(const swf_block_desc& d, const void* packed_x, const void* packed_y,
                        const float* x_in, const float* y_in, float* x_out, float* y_out, int B, int H, int W,
                        hipStream_t stream, const void* next_packed_x = nullptr, const void* next_packed_y = nullptr,
                        size_t next_bytes = 0);
// next_packed_*: packed images of the block that runs next with different weights (or nullptr): this launch ends by touching
// them (next_bytes each; 0 = the size of this block's own image) so that they are L2-resident when that block starts.

// Touch a buffer from every XCD so that it is L2-resident for the next launch (a fused stage that follows a deep-level stage).
int launch_l2_warm(const void* p, size_t bytes, hipStream_t stream);

// MFMA attention core on projection buffers (8x8 or 7x7 windows — `win` —, head_dim in {3,6,12,24,48}); same contract as
// launch_attn_core of the exact tier, fast-tier arithmetic (bf16 QK^T, fp16 PV, fp32 softmax).
bool attn_core_mfma_supported(int wh, int ww, int head_dim);
int launch_attn_core_mfma(const float* const* Q, const float* const* K, const float* const* V, float* const* O,
                          const float* const* table, int nprob, int ldq, int ldk, int ldv, int ldo, int B, int H, int W,
                          int heads, int head_dim, int shift, hipStream_t stream, unsigned short* const* O_hi = nullptr,
                          unsigned short* const* O_lo = nullptr, const unsigned short* const* Q16 = nullptr,
                          const unsigned short* const* K16 = nullptr, const unsigned short* const* V16 = nullptr, int win = 8);
// Q16 / K16 / V16 non-null (head_dim % 4 == 0): the operands come in the core's own formats — Q f16 pre-scaled by
// d^-0.5 * log2(e), K f16, V f16 (row strides ldq / ldk / ldv in elements) — as written by the deep-level Q/K/V GEMM
// epilogue (SP_EPI_QKV16); Q / K / V are then ignored.
// O_hi / O_lo non-null (head_dim % 4 == 0): the output is written as split-bf16 planes hi = bf16(o), lo = bf16(o - hi)
// with row stride ldo instead of fp32 O (input format of the deep-level projection GEMM, kernels_deep.h).

// MFMA attention core for 16x16 windows (256 tokens): online softmax over key tiles.  bias_scratch is unused
// (attn_core_mfma16_scratch_floats returns 0; kept for source compatibility of the callers).
bool attn_core_mfma16_supported(int wh, int ww, int head_dim);
size_t attn_core_mfma16_scratch_floats(int nprob);
int launch_attn_core_mfma16(const float* const* Q, const float* const* K, const float* const* V, float* const* O,
                            const float* const* table, int nprob, int ldq, int ldk, int ldv, int ldo, int B, int H, int W,
                            int heads, int head_dim, int shift, float* bias_scratch, hipStream_t stream, unsigned short* const* O_hi = nullptr,
                            unsigned short* const* O_lo = nullptr, const unsigned short* const* Q16 = nullptr,
                            const unsigned short* const* K16 = nullptr, const unsigned short* const* V16 = nullptr);   // 16-bit operand / plane variants: see launch_attn_core_mfma

}  // namespace swf
