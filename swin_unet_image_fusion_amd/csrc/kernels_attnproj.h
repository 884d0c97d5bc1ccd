// Level-4 attention tail in one launch (kernels_attnproj.hip): window attention of all 8 heads from the Q/K/V GEMM's 16-bit
// outputs + output projection + bias + residual.  Replaces launch_attn_core_mfma + the projection launch_gemm_sp (and the O planes'
// round trip) where C = heads * head_dim = 384, 8 heads of 48, 8x8 or 7x7 windows.
#pragma once
#include "kernels_deep.h"

namespace swf {

struct AttnProjArgs {
    const bf16_raw* q[2]; const bf16_raw* k[2]; const bf16_raw* v[2];   // fp16 [M][384] as written by SP_EPI_QKV16 (Q pre-scaled by d^-0.5 log2 e)
    const bf16_raw* wp_hi[2]; const bf16_raw* wp_lo[2];                // projection weight, split-bf16, fragment-major (DeepWeights::pf_hi / pf_lo)
    const float* pbias[2];                                             // projection bias [384] or nullptr
    const float* table[2];                                             // relative-position bias table [(2 ws - 1)^2]
    const float* res[2]; float* out[2];                                // residual rows / result rows [M][384]; out may alias res
    int B, H, W, shift;
};

bool attnproj_supported(const swf_block_desc& d);
int launch_attnproj(const swf_block_desc& d, const AttnProjArgs& a, int nstream, hipStream_t stream);

}  // namespace swf
