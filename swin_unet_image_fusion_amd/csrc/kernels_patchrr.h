// Patch layers of the fast tier, register-resident variant (kernels_patchrr.hip): gather -> 1x1 conv (split-bf16 x3 on the
// 32x32x16 MFMA) -> LayerNorm -> ELU (-> depth-to-space scatter + skip) with no LDS tile and no workgroup barrier.  Needs the conv
// weights pre-packed as MFMA A fragments (pack_patch_rr, once per model); launch_patch_fused (kernels_patch.h) stays the path
// for every other shape and for callers without a packed image.
#pragma once
#include "kernels_patch.h"

namespace swf {

// shapes covered: 2x2 merging; encoder Cin == 1 or Cin % 8 == 0, decoder Cin % 8 == 0 and Cout % 4 == 0 or Cout == 1; K <= 192, N <= 192
bool patch_rr_supported(int decoder, int Cin, int Cout, int mh, int mw);
// bytes of the packed image of ONE stream of one layer (0 = shape not covered)
size_t patch_rr_packed_bytes(int decoder, int Cin, int Cout, int mh, int mw);
int pack_patch_rr(int decoder, int Cin, int Cout, int mh, int mw, const float* weight, const float* bias, const float* gamma,
                  const float* beta, void* dst, hipStream_t stream);
// d as for launch_patch_fused (d.w / d.bias / d.gamma / d.beta are not read); packed[s]: the stream's image written by pack_patch_rr
int launch_patch_rr(const PatchFusedDesc& d, const void* const* packed, int nstream, hipStream_t stream);

}  // namespace swf
