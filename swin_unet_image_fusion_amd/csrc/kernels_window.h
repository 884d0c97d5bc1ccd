// Fused fast tier: one launch = one BasicBlock (both modality streams) on LDS-resident window tiles.
#pragma once
#include "swf_common.h"

namespace swf {

// true when the fused window-block kernel covers this (dims, map) combination
bool window_block_supported(const swf_block_desc& d, int B, int H, int W);
size_t window_block_workspace_bytes(const swf_block_desc& d, int B, int H, int W);
int launch_window_block(const swf_block_desc& d, const swf_block_stream_params& px, const swf_block_stream_params& py,
                        const float* x_in, const float* y_in, float* x_out, float* y_out, int B, int H, int W,
                        void* workspace, size_t workspace_bytes, hipStream_t stream);

}  // namespace swf
