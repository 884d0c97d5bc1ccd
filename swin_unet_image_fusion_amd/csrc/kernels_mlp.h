// Deep-level fast tier: LN2 + fc1 + ELU + fc2 + residual of a BasicBlock in one launch (kernels_mlp.hip).
#pragma once
#include "kernels_deep.h"

namespace swf {

struct MlpFusedDesc {
    const float* x[2];                              // [M][C] residual stream (input of LN2 and of the residual add)
    float* out[2];                                  // [M][C]; may alias x
    const float* gamma[2]; const float* beta[2];    // LN2
    const bf16_raw* w1_hi[2]; const bf16_raw* w1_lo[2];   // fc1 [HID][C] split planes, FRAGMENT-MAJOR (DeepWeights::w1f_*)
    const bf16_raw* w2_hi[2]; const bf16_raw* w2_lo[2];   // fc2 [C][HID] split planes, FRAGMENT-MAJOR (DeepWeights::w2f_*)
    const float* b1[2]; const float* b2[2];
    float* scratch; int64_t scratch_floats;         // nstream * mlp_fused_splits * M * C floats when splits > 1
    int M, C, HID;
    // optional (all or none): LayerNorm of the finished rows with these parameters — the next block's LN1 — written as split
    // planes [M][C] by the reduce kernel.  Only honoured when mlp_fused_splits(C, HID) > 1 (mlp_fused_writes_ln).
    const float* ln_gamma[2]; const float* ln_beta[2]; bf16_raw* ln_hi[2]; bf16_raw* ln_lo[2];
};

bool mlp_fused_supported(int C, int HID);
int mlp_fused_splits(int C, int HID);
inline bool mlp_fused_writes_ln(int C, int HID) { return mlp_fused_splits(C, HID) > 1; }
int launch_mlp_fused(const MlpFusedDesc& d, int nstream, hipStream_t stream);

}  // namespace swf
