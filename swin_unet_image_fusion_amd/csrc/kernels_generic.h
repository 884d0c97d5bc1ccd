// Launchers of the exact-fp32 ("generic") kernel tier: every shape the reference modules accept.
#pragma once
#include "swf_common.h"

namespace swf {

constexpr int kMaxProb = 6;  // problems batched into one launch via blockIdx.z (streams x {q,k,v})

struct GemmProb {
    const float* A;     // [M][K] row stride lda
    const float* W;     // [N][K] row-major (nn.Linear weight)
    const float* bias;  // [N] or nullptr
    const float* res;   // [M][N] row stride ldo, added to the result, or nullptr
    float* out;         // [M][N] row stride ldo
};
struct GemmBatch {
    GemmProb p[kMaxProb];
    float* scratch = nullptr;        // optional: room for split-K partials (fast tier); nullptr disables split-K
    int64_t scratch_floats = 0;
};

// out = act(A . W^T + bias) (+ res); act: 0 none, 1 ELU(alpha=1).  Exact fp32 (f32-input MFMA).
int launch_gemm_f32(const GemmBatch& batch, int nprob, int M, int N, int K, int lda, int ldo, int act,
                    hipStream_t stream);

// K slices the fast-tier GEMM uses for a given K (scratch need = slices * nprob * M * N floats)
int gemm_splitk_for(int K);
// Same contract in fast-tier arithmetic: split-bf16 (bf16x3) operands on the bf16 MFMA, fp32 accumulate.
int launch_gemm_bf16x3(const GemmBatch& batch, int nprob, int M, int N, int K, int lda, int ldo, int act,
                       hipStream_t stream);
inline int launch_gemm(int fast, const GemmBatch& batch, int nprob, int M, int N, int K, int lda, int ldo, int act,
                       hipStream_t stream) {
    return fast ? launch_gemm_bf16x3(batch, nprob, M, N, K, lda, ldo, act, stream)
                : launch_gemm_f32(batch, nprob, M, N, K, lda, ldo, act, stream);
}

// LayerNorm-prologue GEMM (fast tier): out[M][N] = act(LN(x[M][C]; gamma, beta) . W[N][C]^T + bias)
struct LnGemmProb { const float* x; const float* gamma; const float* beta; const float* W; const float* bias; float* out; };
struct LnGemmBatch { LnGemmProb p[kMaxProb]; };
bool lngemm_supported(int C);
int launch_lngemm_bf16x3(const LnGemmBatch& batch, int nprob, int M, int N, int C, int act, hipStream_t stream);

struct LnProb {
    const float* in; float* out; const float* gamma; const float* beta;
    // optional (vector path only, C % 4 == 0): write the result as split-bf16 planes hi = bf16(v), lo = bf16(v - hi)
    // instead of fp32 `out` (input format of the deep-level GEMMs, kernels_deep.h)
    unsigned short* out_hi = nullptr; unsigned short* out_lo = nullptr;
};
struct LnBatch { LnProb p[2]; };
// LayerNorm over the last dim (eps 1e-5, biased variance), optional ELU on the result.
int launch_layernorm(const LnBatch& batch, int nprob, int64_t tokens, int C, int elu, hipStream_t stream);

struct AttnCoreProb {
    const float* Q; const float* K; const float* V;  // token-major, image order, row strides ldq/ldk/ldv
    float* O;                                        // [tokens][heads*head_dim], row stride ldo
    const float* bias_table;                         // [(2wh-1)(2ww-1)]
};
struct AttnCoreBatch { AttnCoreProb p[2]; };
// softmax(QK^T * d^-0.5 + bias (+mask)) V per (window, head); handles the cyclic shift by index
// arithmetic on load/store.  SWF_ERR_UNSUPPORTED if head_dim > 64 or the K/V tile exceeds LDS.
int launch_attn_core(const AttnCoreBatch& batch, int nprob, int ldq, int ldk, int ldv, int ldo,
                     int B, int H, int W, int wh, int ww, int heads, int head_dim, int shift,
                     hipStream_t stream);

// Encoder gather: A[n][(ph*mw+pw)*Cin+c] for every token of the padded merged map [B][Ho][Wo]
// (reflect pad of the input to a multiple of the merge size, and of the merged map to a multiple of
// the window, both folded into the index).
struct PtrPair { const float* in[2]; float* out[2]; const float* aux[2]; };
int launch_merge_gather(const PtrPair& pp, int nprob, int B, int H, int W, int Cin, int mh, int mw,
                        int Hm, int Wm, int Ho, int Wo, hipStream_t stream);
// [B][H][W][C] -> [B][H+ph][W+pw][C], reflect on the bottom / right edge
int launch_reflect_pad(const float* in, float* out, int B, int H, int W, int C, int ph, int pw, hipStream_t stream);
// [B][Hp][Wp][C] -> [B][Hm][Wm][C] (top-left crop)
int launch_crop(const PtrPair& pp, int nprob, int B, int Hp, int Wp, int Hm, int Wm, int C, hipStream_t stream);
// Decoder scatter: out[b][y][x][c] = ELU(Z[(b,y/mh,x/mw)][((y%mh)*mw+x%mw)*Cout+c]) (+ skip)
// LayerNorm(batch.p[i].in rows of mh*mw*Cout) -> depth-to-space -> ELU (+ pp.aux skip) -> pp.out, one launch; bit-identical to
// launch_layernorm followed by launch_unmerge_scatter.  Needs Cout % 4 == 0, mh*mw*Cout <= 1024, 16-byte aligned tensors.
bool ln_unmerge_scatter_supported(const LnBatch& batch, const PtrPair& pp, int nprob, int Cout, int mh, int mw);
int launch_ln_unmerge_scatter(const LnBatch& batch, const PtrPair& pp, int nprob, int B, int Hm, int Wm, int Cout, int mh, int mw,
                              int Hout, int Wout, hipStream_t stream);
int launch_unmerge_scatter(const PtrPair& pp, int nprob, int B, int Hm, int Wm, int Cout, int mh, int mw,
                           int Hout, int Wout, hipStream_t stream);

// head: tmp[b][y][x][2] = ELU(BN(conv1(cat(x,y))));  out = conv2(tmp); reflect 'same' padding
int launch_head_conv1(const float* x, const float* y, float* tmp, const swf_head_params& p, int B, int H, int W,
                      int ks, hipStream_t stream);
int launch_head_conv2(const float* tmp, float* out, const swf_head_params& p, int B, int H, int W, int ks,
                      hipStream_t stream);
// both convolutions; one fused launch for 3x3 kernels (tmp [B][H][W][2] is only used by the two-kernel path)
int launch_head(const float* x, const float* y, float* tmp, float* out, const swf_head_params& p, int B, int H, int W, int ks,
                hipStream_t stream);

// *flag (device, pre-set to 1 by the caller) is cleared when a[i] != b[i] for any i < n: the reference's first-call
// `(x == y).all()` test in front of a cross-attention block (a005_BasicBlock.py:111-113); NaN != NaN as in torch
int launch_all_equal(const float* a, const float* b, int64_t n, int32_t* flag, hipStream_t stream);

int launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int H, int W, hipStream_t stream);
int launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int H, int W, hipStream_t stream);

}  // namespace swf
