// Fused patch layers of the fast tier: gather -> 1x1 conv (bf16x3 MFMA) -> LayerNorm -> ELU (-> scatter + skip) in one launch.
#pragma once
#include "swf_common.h"

namespace swf {

struct PatchFusedDesc {
    const float* in[2]; float* out[2]; const float* skip[2];     // skip: decoder only (or nullptr)
    const float* w[2]; const float* bias[2]; const float* gamma[2]; const float* beta[2];   // conv [N][K], bias [N], LN [N]
    int decoder;            // 0: PatchMerging (encoder), 1: anti-merging (decoder)
    int B, H, W, Cin;       // input map [B][H][W][Cin] (decoder: the window-padded map Hp x Wp)
    int mh, mw, Hm, Wm;     // merge size; merged map (decoder: cropped map the conv runs on)
    int Ho, Wo;             // encoder: window-padded merged map (output); decoder: output extent Hout x Wout
    int K, N, Cout;         // conv in / out features; decoder: channels per output pixel (N = mh*mw*Cout)
    int64_t M;              // tokens per stream (encoder: B*Ho*Wo, decoder: B*Hm*Wm)
};

bool patch_fused_supported(int K, int N);
int launch_patch_fused(const PatchFusedDesc& d, int nstream, hipStream_t stream);

}  // namespace swf
