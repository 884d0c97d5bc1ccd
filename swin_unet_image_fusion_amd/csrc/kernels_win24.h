// Level-0 fused BasicBlock kernel (C = 24, 8 heads of 3, 8x8 windows; hidden 96 or 4): register-resident design.
// Same contract as the window_block family in kernels_window.h; kernels_window.hip routes these shapes here.
#pragma once
#include "swf_common.h"

namespace swf {

bool win24_supported(const swf_block_desc& d);
// bytes of the packed weights of ONE stream (fragment-major split-bf16 images, fp32 vectors, bias matrix)
size_t win24_packed_bytes(const swf_block_desc& d);
int pack_win24(const swf_block_desc& d, const swf_block_stream_params& px, const swf_block_stream_params& py,
               void* packed_x, void* packed_y, hipStream_t stream);
int launch_win24(const swf_block_desc& d, const void* packed_x, const void* packed_y, const float* x_in, const float* y_in,
                 float* x_out, float* y_out, int B, int H, int W, hipStream_t stream, const void* next_packed_x,
                 const void* next_packed_y, size_t next_bytes);

constexpr int WIN24_HALF_ATTN = 1, WIN24_HALF_MLP = 2;   // `mode` of launch_win24_half
// packed bytes of ONE stream for a half-block launch (C = 24, hidden 96 or 4; the attention half uses the hidden-96 layout); 0 = not covered
size_t win24_half_packed_bytes(int channels, int hidden);
// The two halves of the block as launches of their own (fast tier of the stand-alone a001 / a003 / a004 module entries):
// mode 1 = attention half, mode 2 = MLP half; raw = 1: no LayerNorm, no residual (WindowAttention.forward / AutoPathMLP.forward).
// pack_win24 packs whatever weights the stream parameters hold (a missing half packs as zeros).  See kernels_win24.hip.
int launch_win24_half(const swf_block_desc& d, int mode, int raw, const void* packed_x, const void* packed_y, const float* x_in,
                      const float* y_in, float* x_out, float* y_out, int B, int H, int W, int ntok_x, int ntok_y, hipStream_t stream);

}  // namespace swf
