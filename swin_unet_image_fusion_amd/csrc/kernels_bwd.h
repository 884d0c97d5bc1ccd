// Backward of one BasicBlock in exact fp32 (kernels_bwd.hip): SURVEY.md section 8(f) rank 4, first stage.
#pragma once
#include "swf_common.h"

namespace swf {

size_t basic_block_bwd_ws(const swf_block_desc& d, int nstream, int B, int H, int W);
int basic_block_bwd(const swf_block_desc& d, const swf_block_stream_params* px, const swf_block_stream_params* py, const float* x_in,
                    const float* y_in, const float* gx_out, const float* gy_out, float* gx_in, float* gy_in, const swf_block_stream_grads* gx,
                    const swf_block_stream_grads* gy, int B, int H, int W, void* workspace, size_t workspace_bytes, hipStream_t stream);

}  // namespace swf
