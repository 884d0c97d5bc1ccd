// Fused fast tier (placeholder until the MFMA window kernel lands): reports "not supported" so
// every block runs on the exact-fp32 tier.
#include "kernels_window.h"

namespace swf {

bool window_block_supported(const swf_block_desc&, int, int, int) { return false; }
size_t window_block_workspace_bytes(const swf_block_desc&, int, int, int) { return 0; }
int launch_window_block(const swf_block_desc&, const swf_block_stream_params&, const swf_block_stream_params&, const float*,
                        const float*, float*, float*, int, int, int, void*, size_t, hipStream_t) {
    return fail(SWF_ERR_UNSUPPORTED, "fused window block not available");
}

}  // namespace swf
