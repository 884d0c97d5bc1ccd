// Level-1 fused BasicBlock kernel (C = 48, 8 heads of 6, 8x8 windows; hidden 192 or 96): register-resident design
// (kernels_win48.hip).  Same contract as the window_block family in kernels_window.h, which routes these shapes here.
#pragma once
#include "swf_common.h"

namespace swf {

bool win48_supported(const swf_block_desc& d);
size_t win48_packed_bytes(const swf_block_desc& d);   // ONE stream
int pack_win48(const swf_block_desc& d, const swf_block_stream_params& px, const swf_block_stream_params& py,
               void* packed_x, void* packed_y, hipStream_t stream);
int launch_win48(const swf_block_desc& d, const void* packed_x, const void* packed_y, const float* x_in, const float* y_in,
                 float* x_out, float* y_out, int B, int H, int W, hipStream_t stream, const void* next_packed_x,
                 const void* next_packed_y, size_t next_bytes);

}  // namespace swf
