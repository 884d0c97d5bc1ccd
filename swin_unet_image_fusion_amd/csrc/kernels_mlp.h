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
    int schedule;                                   // swf_schedule: THROUGHPUT takes 64-token tiles where LATENCY takes 32 (C = 192)
    // optional (all or none): LayerNorm of the finished rows with these parameters — the next block's LN1 — written as split
    // planes [M][C] by the reduce kernel (hidden splits > 1) or by the fused kernel's own epilogue (unsplit).
    const float* ln_gamma[2]; const float* ln_beta[2]; bf16_raw* ln_hi[2]; bf16_raw* ln_lo[2];
    // optional (all or none): the attention half's tail folded into the prologue.  The rows that enter LN2 (and the residual add) are
    // x + pbias + part0 + part1, summed in that order: the output projection's bias and its two head-group partial sums as
    // written by launch_qkvattn (QkvAttnArgs::part).  x1: [M][C] scratch that receives those rows; must not alias x or out.
    const float* part0[2]; const float* part1[2]; const float* pbias[2]; float* x1[2];
};

bool mlp_fused_supported(int C, int HID);
int mlp_fused_splits(int C, int HID);
inline bool mlp_fused_writes_ln(int, int) { return true; }
int launch_mlp_fused(const MlpFusedDesc& d, int nstream, hipStream_t stream);

}  // namespace swf
