// Level-2 fused BasicBlock kernel (C = 96, 8 heads of 12, 8x8 or 7x7 windows; hidden 384 or 192): register-resident design
// (kernels_win96.hip).  Same contract as the window_block family in kernels_window.h, which routes these shapes here.
#pragma once
#include "swf_common.h"

namespace swf {

bool win96_supported(const swf_block_desc& d);
size_t win96_packed_bytes(const swf_block_desc& d);   // ONE stream
int pack_win96(const swf_block_desc& d, const swf_block_stream_params& px, const swf_block_stream_params& py,
               void* packed_x, void* packed_y, hipStream_t stream);
int launch_win96(const swf_block_desc& d, const void* packed_x, const void* packed_y, const float* x_in, const float* y_in,
                 float* x_out, float* y_out, int B, int H, int W, hipStream_t stream, const void* next_packed_x,
                 const void* next_packed_y, size_t next_bytes);

// the two halves of the block as launches of their own (mode 1 = attention half, 2 = MLP half; raw = 1: no LayerNorm, no residual);
// same contract as launch_win24_half (kernels_win24.h)
size_t win96_half_packed_bytes(int channels, int hidden);
int launch_win96_half(const swf_block_desc& d, int mode, int raw, const void* packed_x, const void* packed_y, const float* x_in,
                      const float* y_in, float* x_out, float* y_out, int B, int H, int W, int ntok_x, int ntok_y, hipStream_t stream);

}  // namespace swf
