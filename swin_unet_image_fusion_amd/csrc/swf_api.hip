// C-ABI entry points of libswinfuse: argument checks, workspace carving and the composition
// of kernels into the reference's units (WindowAttention, BasicBlock halves, SelfAndCrossBlockPair,
// PatchMergingAndLinearLayer + MyPadding, final head, MyModel.forward).
#include <algorithm>
#include <cmath>
#include <mutex>
#include <string>
#include <vector>

#include "kernels_generic.h"
#include <cstdlib>
#include "kernels_window.h"
#include "kernels_bwd.h"
#include "kernels_win24.h"
#include "kernels_win48.h"
#include "kernels_win96.h"
#include "kernels_deep.h"
#include "kernels_patch.h"
#include "kernels_patchrr.h"
#include "kernels_deeppatch.h"
#include "kernels_mlp.h"
#include "kernels_qkvattn.h"
#include "kernels_attnproj.h"

namespace swf {

char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
int fail(int status, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return status;
}

static int check_attn_desc(const swf_attn_desc* d, int B, int H, int W) {
    if (!d) return fail(SWF_ERR_NULL, "desc is NULL");
    if (d->channels <= 0 || d->heads <= 0 || d->head_dim <= 0 || d->win_h <= 0 || d->win_w <= 0)
        return fail(SWF_ERR_BAD_SHAPE, "non-positive attention dims");
    if (B <= 0 || H <= 0 || W <= 0) return fail(SWF_ERR_BAD_SHAPE, "empty batch or map (%d,%d,%d)", B, H, W);
    if (H % d->win_h || W % d->win_w)
        return fail(SWF_ERR_BAD_SHAPE, "map %dx%d is not a multiple of the window %dx%d", H, W, d->win_h, d->win_w);
    return SWF_OK;
}

// room for the split-K partials of the largest fast-tier GEMM of a unit
static int64_t splitk_need(int K, int64_t nprob_m_n) {
    const int sk = gemm_splitk_for(K);
    return sk > 1 ? sk * nprob_m_n : 0;
}

// ---- generic composition -----------------------------------------------------------------
// Q/K/V projections (three GEMM problems per stream in one launch), attention core, output
// projection (+ residual).  qsrc/ksrc/vsrc are [N][C] token-major inputs per stream.
static int attention_generic(const swf_attn_desc& d, int nstream, const swf_attn_params* const* prm,
                             const float* const* qsrc, const float* const* ksrc, const float* const* vsrc,
                             const float* const* residual, float* const* out, int B, int H, int W, Carver& ws,
                             hipStream_t stream, int fast = 0, const swf_norm* const* qnorm = nullptr,
                             const swf_norm* const* kvnorm = nullptr) {
    // qnorm / kvnorm non-null: q/k/v sources are PRE-LayerNorm tensors and the projections run with the
    // LayerNorm prologue (fast tier only)
    const int64_t N = (int64_t)B * H * W;
    const int C = d.channels, HD = d.heads * d.head_dim;
    float* qkv[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    float* o[2] = {nullptr, nullptr};
    for (int s = 0; s < nstream; ++s) {
        for (int i = 0; i < 3; ++i) qkv[s][i] = ws.floats(N * HD);
        o[s] = ws.floats(N * HD);
    }
    const int64_t sk_floats = fast ? std::max(splitk_need(C, 3 * nstream * N * HD), splitk_need(HD, nstream * N * C)) : 0;
    float* sk = fast ? ws.floats(sk_floats) : nullptr;
    const bool win16 = fast && attn_core_mfma16_supported(d.win_h, d.win_w, d.head_dim);
    float* bias16 = win16 ? ws.floats((int64_t)attn_core_mfma16_scratch_floats(nstream)) : nullptr;
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "attention workspace too small (need %zu B)", ws.used);
    GemmBatch gb{};
    gb.scratch = sk; gb.scratch_floats = sk_floats;
    for (int s = 0; s < nstream; ++s) {
        const swf_linear* lin[3] = {&prm[s]->q, &prm[s]->k, &prm[s]->v};
        const float* src[3] = {qsrc[s], ksrc[s], vsrc[s]};
        for (int i = 0; i < 3; ++i) gb.p[s * 3 + i] = GemmProb{src[i], lin[i]->weight, lin[i]->bias, nullptr, qkv[s][i]};
    }
    if (qnorm) {
        LnGemmBatch lb{};
        for (int s = 0; s < nstream; ++s) {
            const swf_linear* lin[3] = {&prm[s]->q, &prm[s]->k, &prm[s]->v};
            const float* src[3] = {qsrc[s], ksrc[s], vsrc[s]};
            const swf_norm* nrm[3] = {qnorm[s], kvnorm[s], kvnorm[s]};
            for (int i = 0; i < 3; ++i) lb.p[s * 3 + i] = LnGemmProb{src[i], nrm[i]->gamma, nrm[i]->beta, lin[i]->weight, lin[i]->bias, qkv[s][i]};
        }
        SWF_TRY(launch_lngemm_bf16x3(lb, nstream * 3, (int)N, HD, C, 0, stream));
    } else {
        SWF_TRY(launch_gemm(fast, gb, nstream * 3, (int)N, HD, C, C, HD, 0, stream));
    }
    if (win16) {
        const float* qq[2] = {qkv[0][0], qkv[1][0]};
        const float* kk[2] = {qkv[0][1], qkv[1][1]};
        const float* vv[2] = {qkv[0][2], qkv[1][2]};
        const float* tt[2] = {prm[0]->bias_table, nstream == 2 ? prm[1]->bias_table : nullptr};
        SWF_TRY(launch_attn_core_mfma16(qq, kk, vv, o, tt, nstream, HD, HD, HD, HD, B, H, W, d.heads, d.head_dim, d.shift, bias16, stream));
    } else if (fast && attn_core_mfma_supported(d.win_h, d.win_w, d.head_dim)) {
        const float* qq[2] = {qkv[0][0], qkv[1][0]};
        const float* kk[2] = {qkv[0][1], qkv[1][1]};
        const float* vv[2] = {qkv[0][2], qkv[1][2]};
        const float* tt[2] = {prm[0]->bias_table, nstream == 2 ? prm[1]->bias_table : nullptr};
        SWF_TRY(launch_attn_core_mfma(qq, kk, vv, o, tt, nstream, HD, HD, HD, HD, B, H, W, d.heads, d.head_dim, d.shift, stream, nullptr, nullptr,
                                      nullptr, nullptr, nullptr, d.win_h));
    } else {
        AttnCoreBatch ab{};
        for (int s = 0; s < nstream; ++s) ab.p[s] = AttnCoreProb{qkv[s][0], qkv[s][1], qkv[s][2], o[s], prm[s]->bias_table};
        SWF_TRY(launch_attn_core(ab, nstream, HD, HD, HD, HD, B, H, W, d.win_h, d.win_w, d.heads, d.head_dim, d.shift, stream));
    }
    GemmBatch pb{};
    pb.scratch = sk; pb.scratch_floats = sk_floats;
    for (int s = 0; s < nstream; ++s)
        pb.p[s] = GemmProb{o[s], prm[s]->proj.weight, prm[s]->proj.bias, residual ? residual[s] : nullptr, out[s]};
    SWF_TRY(launch_gemm(fast, pb, nstream, (int)N, C, HD, HD, C, 0, stream));
    return SWF_OK;
}

static size_t attention_generic_ws(const swf_attn_desc& d, int nstream, int B, int H, int W) {
    const int64_t N = (int64_t)B * H * W, HD = (int64_t)d.heads * d.head_dim;
    size_t total = 0;
    for (int s = 0; s < nstream; ++s) total += carve_bytes({N * HD, N * HD, N * HD, N * HD});
    return total + carve_bytes({std::max(splitk_need(d.channels, 3 * nstream * N * HD), splitk_need((int)HD, nstream * N * d.channels))}) +
           carve_bytes({(int64_t)attn_core_mfma16_scratch_floats(nstream)});
}

static int check_stream_params(const swf_block_stream_params* p, const char* which, bool need_attn, bool need_mlp) {
    if (!p) return fail(SWF_ERR_NULL, "%s params are NULL", which);
    if (need_attn && (!p->ln1.gamma || !p->ln1.beta || !p->attn.q.weight || !p->attn.k.weight || !p->attn.v.weight ||
                      !p->attn.proj.weight || !p->attn.bias_table))
        return fail(SWF_ERR_NULL, "%s: attention half has a NULL weight", which);
    if (need_mlp && (!p->ln2.gamma || !p->ln2.beta || !p->fc1.weight || !p->fc2.weight))
        return fail(SWF_ERR_NULL, "%s: MLP half has a NULL weight", which);
    return SWF_OK;
}

static int attn_halfblock_generic(const swf_block_desc* desc, const swf_block_stream_params* px,
                                  const swf_block_stream_params* py, const float* x_in, const float* y_in, float* x_out,
                                  float* y_out, int B, int H, int W, Carver& ws, hipStream_t stream) {
    const int nstream = py ? 2 : 1;
    const int64_t N = (int64_t)B * H * W;
    const int C = desc->attn.channels;
    const bool cross = desc->cross && nstream == 2;   // single path ignores cross (a002:83)
    const int fast = desc->precision == SWF_PREC_FAST;
    const swf_attn_params* prm[2] = {&px->attn, py ? &py->attn : nullptr};
    const float* res[2] = {x_in, y_in};
    float* out[2] = {x_out, y_out};
    if (fast && lngemm_supported(C)) {
        // LayerNorm folded into the projection GEMMs: K and V of stream s read the OTHER stream's tokens in a
        // cross block, normalised with that stream's LayerNorm (a004:29-38 then a002:67-82)
        const float* raw[2] = {x_in, y_in};
        const swf_norm* nrm[2] = {&px->ln1, py ? &py->ln1 : nullptr};
        const float* qsrc[2] = {raw[0], raw[1]};
        const float* kvsrc[2] = {cross ? raw[1] : raw[0], cross ? raw[0] : raw[1]};
        const swf_norm* kvn[2] = {cross ? nrm[1] : nrm[0], cross ? nrm[0] : nrm[1]};
        return attention_generic(desc->attn, nstream, prm, qsrc, kvsrc, kvsrc, res, out, B, H, W, ws, stream, fast, nrm, kvn);
    }
    float* xn[2] = {ws.floats(N * C), nstream == 2 ? ws.floats(N * C) : nullptr};
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "attention half-block workspace too small");
    LnBatch lb{};
    lb.p[0] = LnProb{x_in, xn[0], px->ln1.gamma, px->ln1.beta};
    if (nstream == 2) lb.p[1] = LnProb{y_in, xn[1], py->ln1.gamma, py->ln1.beta};
    SWF_TRY(launch_layernorm(lb, nstream, N, C, 0, stream));
    const float* qsrc[2] = {xn[0], xn[1]};
    const float* kvsrc[2] = {cross ? xn[1] : xn[0], cross ? xn[0] : xn[1]};
    return attention_generic(desc->attn, nstream, prm, qsrc, kvsrc, kvsrc, res, out, B, H, W, ws, stream, fast);
}

static int mlp_halfblock_generic(const swf_block_desc* desc, const swf_block_stream_params* px,
                                 const swf_block_stream_params* py, const float* x_in, const float* y_in, float* x_out,
                                 float* y_out, int B, int H, int W, Carver& ws, hipStream_t stream) {
    const int nstream = py ? 2 : 1;
    const int64_t N = (int64_t)B * H * W;
    const int C = desc->attn.channels, hid = desc->hidden;
    const int fast = desc->precision == SWF_PREC_FAST;
    const bool fold_ln = fast && lngemm_supported(C);
    float* xn[2] = {nullptr, nullptr};
    if (!fold_ln) { xn[0] = ws.floats(N * C); if (nstream == 2) xn[1] = ws.floats(N * C); }
    float* hb[2] = {ws.floats(N * hid), nstream == 2 ? ws.floats(N * hid) : nullptr};
    const int64_t sk_floats = fast ? std::max(splitk_need(C, nstream * N * hid), splitk_need(hid, nstream * N * C)) : 0;
    float* sk = fast ? ws.floats(sk_floats) : nullptr;
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "MLP half-block workspace too small");
    const swf_block_stream_params* pp[2] = {px, py};
    const float* res[2] = {x_in, y_in};
    float* out[2] = {x_out, y_out};
    GemmBatch g2{};
    g2.scratch = sk; g2.scratch_floats = sk_floats;
    for (int s = 0; s < nstream; ++s) g2.p[s] = GemmProb{hb[s], pp[s]->fc2.weight, pp[s]->fc2.bias, res[s], out[s]};
    if (fold_ln) {
        LnGemmBatch l1{};
        for (int s = 0; s < nstream; ++s) l1.p[s] = LnGemmProb{res[s], pp[s]->ln2.gamma, pp[s]->ln2.beta, pp[s]->fc1.weight, pp[s]->fc1.bias, hb[s]};
        SWF_TRY(launch_lngemm_bf16x3(l1, nstream, (int)N, hid, C, 1, stream));
    } else {
        LnBatch lb{};
        lb.p[0] = LnProb{x_in, xn[0], px->ln2.gamma, px->ln2.beta};
        if (nstream == 2) lb.p[1] = LnProb{y_in, xn[1], py->ln2.gamma, py->ln2.beta};
        SWF_TRY(launch_layernorm(lb, nstream, N, C, 0, stream));
        GemmBatch g1{};
        g1.scratch = sk; g1.scratch_floats = sk_floats;
        for (int s = 0; s < nstream; ++s) g1.p[s] = GemmProb{xn[s], pp[s]->fc1.weight, pp[s]->fc1.bias, nullptr, hb[s]};
        SWF_TRY(launch_gemm(fast, g1, nstream, (int)N, hid, C, C, hid, 1, stream));
    }
    SWF_TRY(launch_gemm(fast, g2, nstream, (int)N, C, hid, hid, C, 0, stream));
    return SWF_OK;
}

static size_t deep_block_ws(const swf_block_desc* d, int nstream, int B, int H, int W);
static size_t block_generic_ws(const swf_block_desc* d, int nstream, int B, int H, int W) {
    const int64_t N = (int64_t)B * H * W;
    size_t a = 0, m = 0;
    for (int s = 0; s < nstream; ++s) a += carve_bytes({N * d->attn.channels});
    a += attention_generic_ws(d->attn, nstream, B, H, W);
    for (int s = 0; s < nstream; ++s) m += carve_bytes({N * d->attn.channels}) + carve_bytes({N * d->hidden});
    m += carve_bytes({std::max(splitk_need(d->attn.channels, nstream * N * d->hidden), splitk_need(d->hidden, nstream * N * d->attn.channels))});
    return std::max(std::max(a, m), deep_block_ws(d, nstream, B, H, W));
}

// ---- deep-level composition (fast tier, C >= 128): pre-split bf16 planes between the units ----------
// LN1 -> planes | Q/K/V GEMMs -> fp32 | MFMA attention core -> planes | proj GEMM (+x) | LN2 -> planes |
// fc1 GEMM + ELU -> planes | fc2 GEMM (+x, split-K when hidden >= 1024).  `packed_*`: pre-split weight images
// (pack_deep_block) or nullptr, in which case they are derived into the workspace on every call.
static size_t deep_block_ws(const swf_block_desc* d, int nstream, int B, int H, int W) {
    if (!deep_block_supported(*d)) return 0;
    const int64_t N = (int64_t)B * H * W, C = d->attn.channels, HD = (int64_t)d->attn.heads * d->attn.head_dim, hid = d->hidden;
    size_t t = 0;
    for (int s = 0; s < nstream; ++s) t += carve_bytes({N * C / 2, N * C / 2});   // LN1 planes first (deep_ln1_planes)
    for (int s = 0; s < nstream; ++s) {
        t += carve_bytes({(int64_t)deep_block_packed_bytes(*d) / 4});
        t += carve_bytes({N * HD / 2, N * HD / 2, N * HD / 2, N * HD / 2, N * HD / 2, N * hid / 2, N * hid / 2});
    }
    int64_t sk = std::max((int64_t)gemm_sp_splitk_for((int)HD, SP_EPI_F32), (int64_t)gemm_sp_splitk_for((int)hid, SP_EPI_F32));
    if (mlp_fused_supported((int)C, (int)hid)) sk = std::max(sk, (int64_t)mlp_fused_splits((int)C, (int)hid));
    t += carve_bytes({sk > 1 ? sk * nstream * N * C : 0});
    // projection folded into the attention / MLP launches (deep_block_impl): two head-group partials + the attention-residual rows
    if (qkvattn_supported(*d) && mlp_fused_supported((int)C, (int)hid) && HD == C)
        for (int s = 0; s < nstream; ++s) t += carve_bytes({N * C, N * C, N * C});
    return t;
}

// The LN1 planes of a deep-level block sit at the START of its workspace, at offsets that depend on the token count and C only:
// whoever runs before the block on the same workspace (the previous block's MLP reduce, the previous stage's last block, the
// patch-merging kernel) can leave them there and say so through `ln1_ready`.
static void deep_ln1_planes(Carver& ws, int64_t N, int C, int nstream, bf16_raw** hi, bf16_raw** lo) {
    for (int s = 0; s < nstream; ++s) {
        hi[s] = reinterpret_cast<bf16_raw*>(ws.floats(N * C / 2));
        lo[s] = reinterpret_cast<bf16_raw*>(ws.floats(N * C / 2));
    }
}

static int deep_block_impl(const swf_block_desc* desc, const swf_block_stream_params* px, const swf_block_stream_params* py,
                           const float* x_in, const float* y_in, float* x_out, float* y_out, int B, int H, int W,
                           Carver& ws, hipStream_t stream, const void* packed_x, const void* packed_y,
                           const swf_block_stream_params* const* next_p = nullptr, bool* ln1_ready = nullptr) {
    // `next_p`: stream parameters of the block that runs next on the same workspace with the same shapes (or nullptr).  When the
    // fused MLP's reduce kernel can, it also writes that block's LN1 planes; `*ln1_ready` reports it, and on entry says whether the
    // previous block did the same for this one (the planes sit at the same workspace offsets: the carve depends on shapes only).
    const bool ln1_given = ln1_ready && *ln1_ready;
    if (ln1_ready) *ln1_ready = false;
    const int nstream = py ? 2 : 1;
    const int64_t N = (int64_t)B * H * W;
    const int C = desc->attn.channels, HD = desc->attn.heads * desc->attn.head_dim, hid = desc->hidden;
    const bool cross = desc->cross && nstream == 2;
    const swf_block_stream_params* pp[2] = {px, py};
    const float* xin[2] = {x_in, y_in};
    float* xout[2] = {x_out, y_out};
    const void* pk[2] = {packed_x, packed_y};
    auto planes = [&](int64_t n) { return reinterpret_cast<bf16_raw*>(ws.floats(n / 2)); };
    bf16_raw *xn_hi[2], *xn_lo[2], *o_hi[2], *o_lo[2], *h_hi[2], *h_lo[2];
    bf16_raw* qkv[2][3];   // Q (bf16, pre-scaled), K (bf16), V (fp16): the attention core's operand formats
    void* wbuf[2] = {nullptr, nullptr};
    deep_ln1_planes(ws, N, C, nstream, xn_hi, xn_lo);
    for (int s = 0; s < nstream; ++s) {
        wbuf[s] = ws.floats((int64_t)deep_block_packed_bytes(*desc) / 4);
        for (int i = 0; i < 3; ++i) qkv[s][i] = planes(N * HD);
        o_hi[s] = planes(N * HD); o_lo[s] = planes(N * HD);
        h_hi[s] = planes(N * hid); h_lo[s] = planes(N * hid);
    }
    int64_t skn = std::max((int64_t)gemm_sp_splitk_for(HD, SP_EPI_F32), (int64_t)gemm_sp_splitk_for(hid, SP_EPI_F32));
    static const bool no_fused_mlp = debug_env("SWF_NO_FUSED_MLP") != nullptr;   // A/B switch
    const bool fused_mlp = !no_fused_mlp && mlp_fused_supported(C, hid) && N <= INT32_MAX / 2;
    if (fused_mlp) skn = std::max(skn, (int64_t)mlp_fused_splits(C, hid));
    const int64_t sk_floats = skn > 1 ? skn * nstream * N * C : 0;
    float* sk = ws.floats(sk_floats);
    // Output projection folded into its neighbours (C = 192 levels): qkv_attn writes the two head-group partial sums of
    // O . Wp^T, the fused MLP kernel's prologue adds x + bias + both and normalises — no projection GEMM launch, O never
    // reaches HBM.  (Carved last so the LN1 planes keep their offsets from block to block.)
    static const bool no_qkvattn = debug_env("SWF_NO_QKVATTN") != nullptr;     // A/B switches
    static const bool no_projfuse = debug_env("SWF_NO_PROJFUSE") != nullptr;
    const bool fused_attn_shape = !no_qkvattn && qkvattn_supported(*desc) && N <= INT32_MAX / 256;
    // Measured: +0.5 % on the step at B=16 256x256 (16x16 maps at this level), -1.7 % at 512x512 — the extra rows the MLP prologue
    // pulls per workgroup only pay while its grid under-fills the chip.  The rule looks at the map size, never at the batch, so
    // batch shards keep taking the same path (bit-identical rows across shard sizes).
    const bool proj_fused = fused_attn_shape && fused_mlp && !no_projfuse && HD == C && (int64_t)H * W <= 256;
    float *part[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *x1[2] = {nullptr, nullptr};
    if (qkvattn_supported(*desc) && mlp_fused_supported(C, hid) && HD == C)
        for (int s = 0; s < nstream; ++s) { part[0][s] = ws.floats(N * C); part[1][s] = ws.floats(N * C); x1[s] = ws.floats(N * C); }
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "deep block workspace too small (need %zu B)", ws.used);
    DeepWeights wv[2];
    for (int s = 0; s < nstream; ++s) {
        if (!pk[s]) { SWF_TRY(pack_deep_block(*desc, *pp[s], wbuf[s], stream)); pk[s] = wbuf[s]; }
        wv[s] = deep_block_views(*desc, pk[s]);
    }
    // attention half (a004:29-38 around a002:58-82)
    if (!ln1_given) {
        LnBatch l1{};
        for (int s = 0; s < nstream; ++s) l1.p[s] = LnProb{xin[s], nullptr, pp[s]->ln1.gamma, pp[s]->ln1.beta, xn_hi[s], xn_lo[s]};
        SWF_TRY(launch_layernorm(l1, nstream, N, C, 0, stream));
    }
    const bool fused_attn = fused_attn_shape && wv[0].qa && (nstream == 1 || wv[1].qa);
    const bool fold_proj = proj_fused && fused_attn && pp[0]->attn.proj.bias && (nstream == 1 || pp[1]->attn.proj.bias);
    if (fused_attn) {   // Q/K/V projections + window attention (+ output projection partials) in one launch
        QkvAttnArgs qa{};
        for (int s = 0; s < nstream; ++s) {
            qa.packed[s] = wv[s].qa; qa.xn_hi[s] = xn_hi[s]; qa.xn_lo[s] = xn_lo[s]; qa.o_hi[s] = o_hi[s]; qa.o_lo[s] = o_lo[s];
            if (fold_proj) { qa.part[0][s] = part[0][s]; qa.part[1][s] = part[1][s]; }
        }
        qa.B = B; qa.H = H; qa.W = W; qa.shift = desc->attn.shift; qa.cross = cross;
        SWF_TRY(launch_qkvattn(*desc, qa, nstream, stream));
    }
    SpGemmBatch gq{};
    for (int s = 0; s < nstream; ++s) {
        const int kvs = cross ? 1 - s : s;   // K and V of stream s read the other stream's normalised tokens in a cross block
        const swf_linear* lin[3] = {&pp[s]->attn.q, &pp[s]->attn.k, &pp[s]->attn.v};
        const bf16_raw* wh[3] = {wv[s].q_hi, wv[s].k_hi, wv[s].v_hi};
        const bf16_raw* wl[3] = {wv[s].q_lo, wv[s].k_lo, wv[s].v_lo};
        for (int i = 0; i < 3; ++i) {
            const int src = i == 0 ? s : kvs;
            gq.p[s * 3 + i] = SpGemmProb{xn_hi[src], xn_lo[src], wh[i], wl[i], lin[i]->bias, nullptr, nullptr, qkv[s][i], nullptr};
        }
    }
    gq.qscale = 1.4426950408889634f / std::sqrt((float)desc->attn.head_dim);   // d^-0.5 (a001:32-34) and exp -> exp2
    const bool deep_qkv = !fused_attn && HD == C && deep_qkv_supported(*desc) && wv[0].qkvf_hi && (nstream == 1 || wv[1].qkvf_hi);
    if (deep_qkv) {   // level 4: the three projections of both streams in one launch of the rows-times-fragment-major-weights kernel
        DeepQkvArgs dq{};
        for (int s = 0; s < nstream; ++s) {
            dq.xn_hi[s] = xn_hi[s]; dq.xn_lo[s] = xn_lo[s]; dq.w_hi[s] = wv[s].qkvf_hi; dq.w_lo[s] = wv[s].qkvf_lo;
            dq.bias[s][0] = pp[s]->attn.q.bias; dq.bias[s][1] = pp[s]->attn.k.bias; dq.bias[s][2] = pp[s]->attn.v.bias;
            for (int i = 0; i < 3; ++i) dq.out[s][i] = qkv[s][i];
        }
        dq.qscale = gq.qscale; dq.cross = cross; dq.M = (int)N; dq.C = C;
        SWF_TRY(launch_deep_qkv(dq, nstream, stream));
    }
    if (!fused_attn && !deep_qkv) SWF_TRY(launch_gemm_sp(gq, 3 * nstream, (int)N, HD, C, HD, SP_EPI_QKV16, stream));
    // level 4 (C = 384): attention core + output projection + residual in one launch
    const bool attn_proj = !fused_attn && HD == C && attnproj_supported(*desc) && wv[0].pf_hi && (nstream == 1 || wv[1].pf_hi);
    if (attn_proj) {
        AttnProjArgs ap{};
        for (int s = 0; s < nstream; ++s) {
            ap.q[s] = qkv[s][0]; ap.k[s] = qkv[s][1]; ap.v[s] = qkv[s][2];
            ap.wp_hi[s] = wv[s].pf_hi; ap.wp_lo[s] = wv[s].pf_lo; ap.pbias[s] = pp[s]->attn.proj.bias; ap.table[s] = pp[s]->attn.bias_table;
            ap.res[s] = xin[s]; ap.out[s] = xout[s];
        }
        ap.B = B; ap.H = H; ap.W = W; ap.shift = desc->attn.shift;
        SWF_TRY(launch_attnproj(*desc, ap, nstream, stream));
    }
    if (!fused_attn && !attn_proj) {
        const bf16_raw* qq[2] = {qkv[0][0], qkv[1][0]};
        const bf16_raw* kk[2] = {qkv[0][1], qkv[1][1]};
        const bf16_raw* vv[2] = {qkv[0][2], qkv[1][2]};
        const float* tt[2] = {pp[0]->attn.bias_table, nstream == 2 ? pp[1]->attn.bias_table : nullptr};
        if (desc->attn.win_h == 16)
            SWF_TRY(launch_attn_core_mfma16(nullptr, nullptr, nullptr, nullptr, tt, nstream, HD, HD, HD, HD, B, H, W, desc->attn.heads,
                                            desc->attn.head_dim, desc->attn.shift, nullptr, stream, o_hi, o_lo, qq, kk, vv));
        else
            SWF_TRY(launch_attn_core_mfma(nullptr, nullptr, nullptr, nullptr, tt, nstream, HD, HD, HD, HD, B, H, W, desc->attn.heads,
                                          desc->attn.head_dim, desc->attn.shift, stream, o_hi, o_lo, qq, kk, vv, desc->attn.win_h));
    }
    SpGemmBatch gp{};
    gp.scratch = sk; gp.scratch_floats = sk_floats;
    for (int s = 0; s < nstream; ++s)
        gp.p[s] = SpGemmProb{o_hi[s], o_lo[s], wv[s].p_hi, wv[s].p_lo, pp[s]->attn.proj.bias, xin[s], xout[s], nullptr, nullptr};
    const bool deep_proj = !fold_proj && !attn_proj && HD == C && deep_proj_supported(*desc) && wv[0].pf_hi && (nstream == 1 || wv[1].pf_hi);
    if (deep_proj) {   // rows x fragment-major Wproj (+ bias + residual) instead of the plane GEMM
        DeepProjArgs dp{};
        for (int s = 0; s < nstream; ++s) {
            dp.o_hi[s] = o_hi[s]; dp.o_lo[s] = o_lo[s]; dp.w_hi[s] = wv[s].pf_hi; dp.w_lo[s] = wv[s].pf_lo;
            dp.bias[s] = pp[s]->attn.proj.bias; dp.res[s] = xin[s]; dp.out[s] = xout[s];
        }
        dp.M = (int)N; dp.C = C;
        SWF_TRY(launch_deep_proj(dp, nstream, stream));
    }
    if (!fold_proj && !attn_proj && !deep_proj) SWF_TRY(launch_gemm_sp(gp, nstream, (int)N, C, HD, C, SP_EPI_F32, stream));
    // MLP half (a004:29-38 around a003:46-50)
    if (fused_mlp) {   // LN2 + fc1 + ELU + fc2 + residual in one launch (+ a fixed-order reduce over the hidden splits)
        MlpFusedDesc md{};
        for (int s = 0; s < nstream; ++s) {
            md.x[s] = fold_proj ? xin[s] : xout[s]; md.out[s] = xout[s]; md.gamma[s] = pp[s]->ln2.gamma; md.beta[s] = pp[s]->ln2.beta;
            if (fold_proj) { md.part0[s] = part[0][s]; md.part1[s] = part[1][s]; md.pbias[s] = pp[s]->attn.proj.bias; md.x1[s] = x1[s]; }
            md.w1_hi[s] = wv[s].w1f_hi; md.w1_lo[s] = wv[s].w1f_lo; md.w2_hi[s] = wv[s].w2f_hi; md.w2_lo[s] = wv[s].w2f_lo;
            md.b1[s] = pp[s]->fc1.bias; md.b2[s] = pp[s]->fc2.bias;
        }
        md.scratch = sk; md.scratch_floats = sk_floats; md.M = (int)N; md.C = C; md.HID = hid; md.schedule = desc->schedule;
        const bool ln_next = next_p && ln1_ready && next_p[0] && (nstream == 1 || next_p[1]) && mlp_fused_writes_ln(C, hid);
        if (ln_next)
            for (int s = 0; s < nstream; ++s) {
                md.ln_gamma[s] = next_p[s]->ln1.gamma; md.ln_beta[s] = next_p[s]->ln1.beta; md.ln_hi[s] = xn_hi[s]; md.ln_lo[s] = xn_lo[s];
            }
        SWF_TRY(launch_mlp_fused(md, nstream, stream));
        if (ln_next) *ln1_ready = true;
        return SWF_OK;
    }
    LnBatch l2{};
    for (int s = 0; s < nstream; ++s) l2.p[s] = LnProb{xout[s], nullptr, pp[s]->ln2.gamma, pp[s]->ln2.beta, xn_hi[s], xn_lo[s]};
    SWF_TRY(launch_layernorm(l2, nstream, N, C, 0, stream));
    SpGemmBatch g1{};
    for (int s = 0; s < nstream; ++s)
        g1.p[s] = SpGemmProb{xn_hi[s], xn_lo[s], wv[s].w1_hi, wv[s].w1_lo, pp[s]->fc1.bias, nullptr, nullptr, h_hi[s], h_lo[s]};
    SWF_TRY(launch_gemm_sp(g1, nstream, (int)N, hid, C, hid, SP_EPI_ELU_SPLIT, stream));
    SpGemmBatch g2{};
    g2.scratch = sk; g2.scratch_floats = sk_floats;
    for (int s = 0; s < nstream; ++s)
        g2.p[s] = SpGemmProb{h_hi[s], h_lo[s], wv[s].w2_hi, wv[s].w2_lo, pp[s]->fc2.bias, xout[s], xout[s], nullptr, nullptr};
    return launch_gemm_sp(g2, nstream, (int)N, C, hid, C, SP_EPI_F32, stream);
}

// bytes of the pre-packed weight image of ONE stream of a block (fused window kernel or deep-level GEMMs); 0 = none
static size_t block_packed_bytes(const swf_block_desc& d) {
    const size_t pb = window_block_packed_bytes(d);
    return pb ? pb : deep_block_packed_bytes(d);
}

static int check_block(const swf_block_desc* desc, const swf_block_stream_params* px, const swf_block_stream_params* py,
                       const float* x_in, const float* y_in, float* x_out, float* y_out, int B, int H, int W,
                       bool attn, bool mlp) {
    if (!desc) return fail(SWF_ERR_NULL, "desc is NULL");
    SWF_TRY(check_attn_desc(&desc->attn, B, H, W));
    if (mlp && desc->hidden <= 0) return fail(SWF_ERR_BAD_SHAPE, "hidden must be positive");
    SWF_TRY(check_stream_params(px, "x-stream", attn, mlp));
    if (!x_in || !x_out) return fail(SWF_ERR_NULL, "x tensors are NULL");
    if (py) {
        SWF_TRY(check_stream_params(py, "y-stream", attn, mlp));
        if (!y_in || !y_out) return fail(SWF_ERR_NULL, "y tensors are NULL with dual-path params");
    }
    return SWF_OK;
}

static int basic_block_impl(const swf_block_desc* desc, const swf_block_stream_params* px,
                            const swf_block_stream_params* py, const float* x_in, const float* y_in, float* x_out,
                            float* y_out, int B, int H, int W, void* workspace, size_t workspace_bytes,
                            hipStream_t stream, const void* prepacked_x = nullptr, const void* prepacked_y = nullptr,
                            const void* next_x = nullptr, const void* next_y = nullptr, size_t next_bytes = 0,
                            const swf_block_stream_params* const* next_p = nullptr, bool* ln1_ready = nullptr) {
    // next_p / ln1_ready: see deep_block_impl; every other path ignores the planes and leaves *ln1_ready false
    if (desc->precision == SWF_PREC_FAST && py && window_block_supported(*desc, B, H, W)) {
        if (ln1_ready) *ln1_ready = false;
        const bool prepacked = prepacked_x && prepacked_y;   // model path: weights were packed once (swf_model_pack_weights)
        const size_t pb = window_block_packed_bytes(*desc);
        char* w = static_cast<char*>(workspace);
        size_t used = prepacked ? 0 : 2 * pb;
        if (!prepacked && (!workspace || workspace_bytes < used)) return fail(SWF_ERR_WORKSPACE, "fused block workspace too small (need %zu B)", used);
        // a kernel whose workgroups read one stream while others write it (kernels_window.h) gets temporary outputs when called in place
        float *ox = x_out, *oy = y_out;
        const size_t map_bytes = (size_t)B * H * W * desc->attn.channels * 4;
        const bool via_tmp = window_block_out_of_place(*desc) && desc->cross && (x_in == x_out || y_in == y_out || x_in == y_out || y_in == x_out);
        if (via_tmp) {
            used = align_up(used, 256);
            if (!workspace || workspace_bytes < used + 2 * map_bytes)
                return fail(SWF_ERR_WORKSPACE, "fused block workspace too small (need %zu B)", used + 2 * map_bytes);
            ox = reinterpret_cast<float*>(w + used);
            oy = reinterpret_cast<float*>(w + used + map_bytes);
        }
        if (prepacked) {
            SWF_TRY(launch_window_block(*desc, prepacked_x, prepacked_y, x_in, y_in, ox, oy, B, H, W, stream, next_x, next_y, next_bytes));
        } else {   // block-level entry: pack this block's weights into the workspace, then one fused launch
            SWF_TRY(pack_window_block(*desc, *px, *py, w, w + pb, stream));
            SWF_TRY(launch_window_block(*desc, w, w + pb, x_in, y_in, ox, oy, B, H, W, stream));
        }
        if (via_tmp) {
            if (hipMemcpyAsync(x_out, ox, map_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
                hipMemcpyAsync(y_out, oy, map_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return fail(SWF_ERR_HIP, "fused block: copy of the temporary outputs failed");
        }
        return SWF_OK;
    }
    auto aligned16 = [](const swf_block_stream_params* p) {
        if (!p) return true;
        const void* v[] = {p->ln1.gamma, p->ln1.beta, p->ln2.gamma, p->ln2.beta, p->attn.q.weight, p->attn.k.weight, p->attn.v.weight,
                           p->attn.proj.weight, p->fc1.weight, p->fc2.weight, p->attn.q.bias, p->attn.k.bias, p->attn.v.bias,
                           p->attn.proj.bias, p->fc1.bias, p->fc2.bias};
        uintptr_t bits = 0;
        for (const void* q : v) bits |= reinterpret_cast<uintptr_t>(q);
        return bits % 16 == 0;
    };
    const uintptr_t tbits = reinterpret_cast<uintptr_t>(x_in) | reinterpret_cast<uintptr_t>(y_in) | reinterpret_cast<uintptr_t>(x_out) |
                            reinterpret_cast<uintptr_t>(y_out);
    static const bool no_deep = debug_env("SWF_NO_DEEP") != nullptr;   // A/B switch for tools/profile_block.py
    if (deep_block_supported(*desc) && !no_deep && tbits % 16 == 0 && aligned16(px) && aligned16(py)) {
        Carver ws(workspace, workspace_bytes);
        return deep_block_impl(desc, px, py, x_in, y_in, x_out, y_out, B, H, W, ws, stream, prepacked_x, prepacked_y, next_p, ln1_ready);
    }
    if (ln1_ready) *ln1_ready = false;
    {
        Carver ws(workspace, workspace_bytes);
        SWF_TRY(attn_halfblock_generic(desc, px, py, x_in, y_in, x_out, y_out, B, H, W, ws, stream));
    }
    Carver ws(workspace, workspace_bytes);
    return mlp_halfblock_generic(desc, px, py, x_out, y_out, x_out, y_out, B, H, W, ws, stream);
}

// ---- patch layers -------------------------------------------------------------------------
static int pad_amount(int len, int mult) { return (mult - len % mult) % mult; }

static int merge_shapes(int H, int W, int mh, int mw, int wh, int ww, int* Hm, int* Wm, int* Ho, int* Wo) {
    if (H <= 0 || W <= 0 || mh <= 0 || mw <= 0 || wh <= 0 || ww <= 0) return fail(SWF_ERR_BAD_SHAPE, "non-positive size");
    const int ph = pad_amount(H, mh), pw = pad_amount(W, mw);
    // F.pad(reflect) requires pad < dim (a006:128 raises RuntimeError otherwise)
    if (ph >= H || pw >= W) return fail(SWF_ERR_PAD, "reflect pad (%d,%d) >= map (%d,%d) before merging", ph, pw, H, W);
    *Hm = (H + ph) / mh;
    *Wm = (W + pw) / mw;
    const int qh = pad_amount(*Hm, wh), qw = pad_amount(*Wm, ww);
    if (qh >= *Hm || qw >= *Wm)
        return fail(SWF_ERR_PAD, "Padding size should be less than the corresponding input dimension: pad (%d,%d) on a %dx%d map",
                    qh, qw, *Hm, *Wm);
    *Ho = *Hm + qh;
    *Wo = *Wm + qw;
    return SWF_OK;
}

static int patch_merge_impl(const swf_patch_params* const* p, int nstream, const float* const* in, float* const* out,
                            int B, int H, int W, int Cin, int Cout, int mh, int mw, int wh, int ww, void* workspace,
                            size_t workspace_bytes, hipStream_t stream, int fast = 0, const void* const* prr = nullptr,
                            const swf_block_stream_params* const* first_blk = nullptr, bool* ln1_ready = nullptr) {
    // prr: per-stream packed images of the register-resident kernel (pack_patch_rr; the model path has them) or nullptr
    // first_blk / ln1_ready: the parameters of the deep-level block that runs next on this workspace; when the whole-row deep patch
    // kernel runs, it also leaves that block's LN1 planes (deep_ln1_planes) and sets *ln1_ready
    if (ln1_ready) *ln1_ready = false;
    int Hm, Wm, Ho, Wo;
    SWF_TRY(merge_shapes(H, W, mh, mw, wh, ww, &Hm, &Wm, &Ho, &Wo));
    const int64_t N = (int64_t)B * Ho * Wo;
    const int K = mh * mw * Cin;
    static const bool no_fused = debug_env("SWF_NO_FUSED_PATCH") != nullptr;   // A/B switches
    static const bool no_prr = debug_env("SWF_NO_PATCH_RR") != nullptr;
    const bool use_prr = fast && !no_fused && !no_prr && prr && nstream == 2 && patch_rr_supported(0, Cin, Cout, mh, mw) &&
                         (int64_t)B * H * W * Cin < (int64_t(1) << 31);
    const bool use_dp = fast && !no_fused && !use_prr && prr && deep_patch_supported(0, Cin, Cout, mh, mw) && (int64_t)B * H * W * Cin < (int64_t(1) << 31);
    if (use_prr || use_dp || (fast && !no_fused && patch_fused_supported(K, Cout))) {   // one launch: gather -> conv -> LN -> ELU
        PatchFusedDesc d{};
        for (int s = 0; s < nstream; ++s) {
            d.in[s] = in[s]; d.out[s] = out[s]; d.skip[s] = nullptr;
            d.w[s] = p[s]->conv.weight; d.bias[s] = p[s]->conv.bias; d.gamma[s] = p[s]->ln.gamma; d.beta[s] = p[s]->ln.beta;
        }
        d.decoder = 0; d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.mh = mh; d.mw = mw; d.Hm = Hm; d.Wm = Wm; d.Ho = Ho; d.Wo = Wo;
        d.K = K; d.N = Cout; d.Cout = Cout; d.M = N;
        if (use_prr) return launch_patch_rr(d, prr, nstream, stream);
        if (use_dp && deep_patch_raw(0, Cin, Cout, mh, mw)) {   // conv over column slices, then LayerNorm + ELU (+ the next block's LN1) as a second launch
            Carver wz(workspace, workspace_bytes);
            DeepPatchExtra ex{};
            const bool with_ln = first_blk && ln1_ready && first_blk[0] && (nstream == 1 || first_blk[1]);
            if (with_ln) {   // the planes sit at the start of the workspace (deep_ln1_planes), the conv rows behind them
                bf16_raw *hi[2] = {nullptr, nullptr}, *lo[2] = {nullptr, nullptr};
                deep_ln1_planes(wz, N, Cout, nstream, hi, lo);
                for (int s = 0; s < nstream; ++s) {
                    ex.ln_gamma[s] = first_blk[s]->ln1.gamma; ex.ln_beta[s] = first_blk[s]->ln1.beta; ex.ln_hi[s] = hi[s]; ex.ln_lo[s] = lo[s];
                }
            }
            float* zr[2] = {nullptr, nullptr};
            for (int s = 0; s < nstream; ++s) zr[s] = wz.floats(N * Cout);
            if (!wz.ok()) return fail(SWF_ERR_WORKSPACE, "patch-merge workspace too small (need %zu B)", wz.used);
            SWF_TRY(launch_deep_patch(d, prr, nstream, stream, zr));
            SWF_TRY(launch_deep_patch_finish(d, zr, nstream, stream, with_ln ? &ex : nullptr));
            if (with_ln) *ln1_ready = true;
            return SWF_OK;
        }
        if (use_dp) {
            DeepPatchExtra ex{};
            const bool with_ln = first_blk && ln1_ready && first_blk[0] && (nstream == 1 || first_blk[1]) && workspace;
            if (with_ln) {
                Carver wl(workspace, workspace_bytes);
                bf16_raw *hi[2] = {nullptr, nullptr}, *lo[2] = {nullptr, nullptr};
                deep_ln1_planes(wl, N, Cout, nstream, hi, lo);
                if (!wl.ok()) return fail(SWF_ERR_WORKSPACE, "patch-merge workspace too small for the LN1 planes (need %zu B)", wl.used);
                for (int s = 0; s < nstream; ++s) {
                    ex.ln_gamma[s] = first_blk[s]->ln1.gamma; ex.ln_beta[s] = first_blk[s]->ln1.beta; ex.ln_hi[s] = hi[s]; ex.ln_lo[s] = lo[s];
                }
            }
            SWF_TRY(launch_deep_patch(d, prr, nstream, stream, nullptr, with_ln ? &ex : nullptr));
            if (with_ln) *ln1_ready = true;
            return SWF_OK;
        }
        return launch_patch_fused(d, nstream, stream);
    }
    Carver ws(workspace, workspace_bytes);
    float* a[2];
    float* z[2];
    for (int s = 0; s < nstream; ++s) { a[s] = ws.floats(N * K); z[s] = ws.floats(N * Cout); }
    const int64_t sk_floats = fast ? splitk_need(K, nstream * N * Cout) : 0;
    float* sk = fast ? ws.floats(sk_floats) : nullptr;
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "patch-merge workspace too small (need %zu B)", ws.used);
    PtrPair pp{};
    GemmBatch gb{};
    gb.scratch = sk; gb.scratch_floats = sk_floats;
    LnBatch lb{};
    for (int s = 0; s < nstream; ++s) {
        pp.in[s] = in[s]; pp.out[s] = a[s];
        gb.p[s] = GemmProb{a[s], p[s]->conv.weight, p[s]->conv.bias, nullptr, z[s]};
        lb.p[s] = LnProb{z[s], out[s], p[s]->ln.gamma, p[s]->ln.beta};
    }
    SWF_TRY(launch_merge_gather(pp, nstream, B, H, W, Cin, mh, mw, Hm, Wm, Ho, Wo, stream));
    SWF_TRY(launch_gemm(fast, gb, nstream, (int)N, Cout, K, K, Cout, 0, stream));
    return launch_layernorm(lb, nstream, N, Cout, 1, stream);
}

static int patch_unmerge_impl(const swf_patch_params* const* p, int nstream, const float* const* in,
                              const float* const* skip, float* const* out, int B, int Hp, int Wp, int Hm, int Wm, int Cin,
                              int Cout, int mh, int mw, int Hout, int Wout, void* workspace, size_t workspace_bytes,
                              hipStream_t stream, int fast = 0, const void* const* prr = nullptr, const char* warm = nullptr,
                              size_t warm_pb = 0, bool* warmed = nullptr, const swf_block_stream_params* const* next_blk = nullptr,
                              bool* ln1_ready = nullptr) {
    // next_blk / ln1_ready: the first block of the decoder stage that runs next (a deep-level stage on this workspace): the
    // column-sliced layer's second launch also leaves that block's LN1 planes (deep_ln1_planes over the output pixels)
    if (ln1_ready) *ln1_ready = false;
    // warm / warm_pb: packed images (x, then y at + warm_pb) of the block that runs next; the whole-row deep patch kernel touches them
    // at its end and sets *warmed (the caller otherwise spends a launch on it)
    if (warmed) *warmed = false;
    if (Hm <= 0 || Wm <= 0 || Hm > Hp || Wm > Wp) return fail(SWF_ERR_BAD_SHAPE, "crop %dx%d of %dx%d", Hm, Wm, Hp, Wp);
    if (Hout <= 0 || Wout <= 0 || Hout > Hm * mh || Wout > Wm * mw)
        return fail(SWF_ERR_BAD_SHAPE, "output %dx%d larger than the unmerged map %dx%d", Hout, Wout, Hm * mh, Wm * mw);
    const int64_t N = (int64_t)B * Hm * Wm;
    const int Kz = mh * mw * Cout;
    const bool need_crop = (Hm != Hp) || (Wm != Wp);
    static const bool no_fused = debug_env("SWF_NO_FUSED_PATCH") != nullptr;   // A/B switches
    static const bool no_prr = debug_env("SWF_NO_PATCH_RR") != nullptr;
    const bool use_prr = fast && !no_fused && !no_prr && prr && nstream == 2 && patch_rr_supported(1, Cin, Cout, mh, mw) &&
                         (int64_t)B * Hp * Wp * Cin < (int64_t(1) << 31);
    const bool use_dp = fast && !no_fused && !use_prr && prr && deep_patch_supported(1, Cin, Cout, mh, mw) && (int64_t)B * Hp * Wp * Cin < (int64_t(1) << 31);
    if (use_prr || use_dp || (fast && !no_fused && patch_fused_supported(Cin, Kz))) {   // one launch: crop -> conv -> LN -> scatter -> ELU (+ skip)
        PatchFusedDesc d{};
        for (int s = 0; s < nstream; ++s) {
            d.in[s] = in[s]; d.out[s] = out[s]; d.skip[s] = skip ? skip[s] : nullptr;
            d.w[s] = p[s]->conv.weight; d.bias[s] = p[s]->conv.bias; d.gamma[s] = p[s]->ln.gamma; d.beta[s] = p[s]->ln.beta;
        }
        d.decoder = 1; d.B = B; d.H = Hp; d.W = Wp; d.Cin = Cin; d.mh = mh; d.mw = mw; d.Hm = Hm; d.Wm = Wm; d.Ho = Hout; d.Wo = Wout;
        d.K = Cin; d.N = Kz; d.Cout = Cout; d.M = N;
        if (use_prr) return launch_patch_rr(d, prr, nstream, stream);
        if (use_dp && deep_patch_raw(1, Cin, Cout, mh, mw)) {   // conv over column slices, then LayerNorm + scatter + ELU (+ skip) (+ the next block's LN1)
            Carver wz(workspace, workspace_bytes);
            DeepPatchExtra ex{};
            const bool with_ln = next_blk && ln1_ready && next_blk[0] && (nstream == 1 || next_blk[1]);
            if (with_ln) {
                bf16_raw *hi[2] = {nullptr, nullptr}, *lo[2] = {nullptr, nullptr};
                deep_ln1_planes(wz, (int64_t)B * Hout * Wout, Cout, nstream, hi, lo);
                for (int s = 0; s < nstream; ++s) {
                    ex.ln_gamma[s] = next_blk[s]->ln1.gamma; ex.ln_beta[s] = next_blk[s]->ln1.beta; ex.ln_hi[s] = hi[s]; ex.ln_lo[s] = lo[s];
                }
            }
            float* zr[2] = {nullptr, nullptr};
            for (int s = 0; s < nstream; ++s) zr[s] = wz.floats(N * Kz);
            if (!wz.ok()) return fail(SWF_ERR_WORKSPACE, "patch-unmerge workspace too small (need %zu B)", wz.used);
            SWF_TRY(launch_deep_patch(d, prr, nstream, stream, zr));
            SWF_TRY(launch_deep_patch_finish(d, zr, nstream, stream, with_ln ? &ex : nullptr));
            if (with_ln) *ln1_ready = true;
            return SWF_OK;
        } else if (use_dp) {
            DeepPatchExtra ex{};
            if (warm && warm_pb && nstream == 2) { ex.warm[0] = warm; ex.warm[1] = warm + warm_pb; ex.warm_bytes = warm_pb; }
            SWF_TRY(launch_deep_patch(d, prr, nstream, stream, nullptr, ex.warm[0] ? &ex : nullptr));
            if (warmed && ex.warm[0]) *warmed = true;
            return SWF_OK;
        }
        if (!use_dp) return launch_patch_fused(d, nstream, stream);
    }
    Carver ws(workspace, workspace_bytes);
    float* cr[2] = {nullptr, nullptr};
    float* z[2];
    float* zn[2];
    for (int s = 0; s < nstream; ++s) {
        if (need_crop) cr[s] = ws.floats(N * Cin);
        z[s] = ws.floats(N * Kz);
        zn[s] = ws.floats(N * Kz);
    }
    const int64_t sk_floats = fast ? splitk_need(Cin, nstream * N * Kz) : 0;
    float* sk = fast ? ws.floats(sk_floats) : nullptr;
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "patch-unmerge workspace too small (need %zu B)", ws.used);
    PtrPair cp{}, sp{};
    GemmBatch gb{};
    gb.scratch = sk; gb.scratch_floats = sk_floats;
    LnBatch lb{};
    for (int s = 0; s < nstream; ++s) {
        cp.in[s] = in[s]; cp.out[s] = cr[s];
        gb.p[s] = GemmProb{need_crop ? cr[s] : in[s], p[s]->conv.weight, p[s]->conv.bias, nullptr, z[s]};
        lb.p[s] = LnProb{z[s], zn[s], p[s]->ln.gamma, p[s]->ln.beta};
        sp.in[s] = zn[s]; sp.out[s] = out[s]; sp.aux[s] = skip ? skip[s] : nullptr;
    }
    if (need_crop) SWF_TRY(launch_crop(cp, nstream, B, Hp, Wp, Hm, Wm, Cin, stream));
    SWF_TRY(launch_gemm(fast, gb, nstream, (int)N, Kz, Cin, Cin, Kz, 0, stream));
    if (ln_unmerge_scatter_supported(lb, sp, nstream, Cout, mh, mw))   // LN + depth-to-space + ELU (+ skip) in one launch
        return launch_ln_unmerge_scatter(lb, sp, nstream, B, Hm, Wm, Cout, mh, mw, Hout, Wout, stream);
    SWF_TRY(launch_layernorm(lb, nstream, N, Kz, 0, stream));
    return launch_unmerge_scatter(sp, nstream, B, Hm, Wm, Cout, mh, mw, Hout, Wout, stream);
}

static size_t patch_ws(int nstream, int B, int H, int W, int Cin, int Cout, int mh, int mw, int wh, int ww, int encoder) {
    size_t total = 0;
    if (encoder) {
        int Hm, Wm, Ho, Wo;
        if (merge_shapes(H, W, mh, mw, wh, ww, &Hm, &Wm, &Ho, &Wo) != SWF_OK) return 0;
        const int64_t N = (int64_t)B * Ho * Wo;
        for (int s = 0; s < nstream; ++s) total += carve_bytes({N * mh * mw * Cin, N * Cout});
        total += carve_bytes({splitk_need(mh * mw * Cin, nstream * N * Cout)});
    } else {
        const int64_t N = (int64_t)B * H * W;   // upper bound: uncropped map
        for (int s = 0; s < nstream; ++s) total += carve_bytes({N * Cin, N * mh * mw * Cout, N * mh * mw * Cout});
        total += carve_bytes({splitk_need(Cin, (int64_t)nstream * N * mh * mw * Cout)});
    }
    return total;
}

// ---- model layout ---------------------------------------------------------------------------
struct BlockOff { int64_t ln1g, ln1b, wq, bq, wk, bk, wv, bv, wp, bp, table, ln2g, ln2b, w1, b1, w2, b2; };
struct PatchOff { int64_t w, b, g, bt; };
struct ParamEntry { std::string name; int64_t offset, numel; };
struct ModelLayout {
    swf_model_desc desc;
    std::vector<ParamEntry> entries;
    int64_t total = 0;
    BlockOff enc_blk[SWF_MAX_LEVELS][4][2], dec_blk[SWF_MAX_LEVELS][4][2];
    PatchOff enc_patch[SWF_MAX_LEVELS][2], dec_patch[SWF_MAX_LEVELS][2];
    int64_t h_c1w, h_c1b, h_g, h_b, h_m, h_v, h_c2w, h_c2b;

    int64_t add(const std::string& name, int64_t numel) {
        const int64_t off = total;
        entries.push_back({name, off, numel});
        total += (numel + 3) / 4 * 4;   // keep every tensor 16-byte aligned
        return off;
    }
    void add_blocks(const std::string& prefix, int C, int heads, int hd, int hidden, int wh, int ww, BlockOff (*dst)[2]) {
        static const char* grp[2] = {"self_att_block.", "cross_att_block."};
        static const char* blk[2] = {"normal_window_block.", "shifted_window_block."};
        for (int g = 0; g < 2; ++g)
            for (int k = 0; k < 2; ++k)
                for (int s = 0; s < 2; ++s) {
                    const std::string p = prefix + grp[g] + blk[k];
                    const std::string st = s == 0 ? "x" : "y";
                    const std::string nl = s == 0 ? "norm_layer_1." : "norm_layer_2.";
                    const std::string wa = p + "auto_path_win_att.window_attention_" + st + ".";
                    const std::string ml = p + "auto_path_mlp.mlp_" + st;
                    BlockOff& o = dst[g * 2 + k][s];
                    const int HD = heads * hd;
                    o.ln1g = add(p + "stage_1." + nl + "weight", C);
                    o.ln1b = add(p + "stage_1." + nl + "bias", C);
                    o.wq = add(wa + "q_for_heads.weight", (int64_t)HD * C); o.bq = add(wa + "q_for_heads.bias", HD);
                    o.wk = add(wa + "k_for_heads.weight", (int64_t)HD * C); o.bk = add(wa + "k_for_heads.bias", HD);
                    o.wv = add(wa + "v_for_heads.weight", (int64_t)HD * C); o.bv = add(wa + "v_for_heads.bias", HD);
                    o.wp = add(wa + "linear_projection.weight", (int64_t)C * HD); o.bp = add(wa + "linear_projection.bias", C);
                    o.table = add(wa + "relative_position_bias_table", (int64_t)(2 * wh - 1) * (2 * ww - 1));
                    o.ln2g = add(p + "stage_2." + nl + "weight", C);
                    o.ln2b = add(p + "stage_2." + nl + "bias", C);
                    o.w1 = add(ml + "_1.weight", (int64_t)hidden * C); o.b1 = add(ml + "_1.bias", hidden);
                    o.w2 = add(ml + "_2.weight", (int64_t)C * hidden); o.b2 = add(ml + "_2.bias", C);
                }
    }
    void add_patch(const std::string& prefix, int cin, int cout, PatchOff* dst) {
        for (int s = 0; s < 2; ++s) {
            const std::string st = s == 0 ? "x" : "y";
            dst[s].w = add(prefix + "mlp_layer_" + st + ".weight", (int64_t)cin * cout);
            dst[s].b = add(prefix + "mlp_layer_" + st + ".bias", cout);
            dst[s].g = add(prefix + "layer_norm_" + st + ".weight", cout);
            dst[s].bt = add(prefix + "layer_norm_" + st + ".bias", cout);
        }
    }
    void build(const swf_model_desc& d) {
        desc = d;
        const int mm = d.merge_h * d.merge_w;
        for (int s = 0; s < d.levels; ++s) {   // encoder: [pad2, merge(.1), padW, blocks(.3)] (a013:302-311)
            const std::string e = "encoder_list." + std::to_string(s) + ".";
            add_patch(e + "1.", d.in_dims[s] * mm, d.out_dims[s], enc_patch[s]);
            add_blocks(e + "3.", d.out_dims[s], d.heads, d.head_dim[s], d.out_dims[s] * d.mlp_ratio, d.win_h, d.win_w, enc_blk[s]);
        }
        for (int j = 0; j < d.levels; ++j) {   // decoder j serves level L-1-j, module order reversed (a013:313-314)
            const int lvl = d.levels - 1 - j;
            const std::string e = "decoder_list." + std::to_string(j) + ".";
            add_blocks(e + "0.", d.out_dims[lvl], d.heads, d.head_dim[lvl], d.in_dims[lvl] * d.mlp_ratio, d.win_h, d.win_w, dec_blk[j]);
            add_patch(e + "2.", d.out_dims[lvl], d.in_dims[lvl] * mm, dec_patch[j]);
        }
        const int k2 = d.head_ksize * d.head_ksize;
        h_c1w = add("final_layer.0.weight", 2 * 2 * k2); h_c1b = add("final_layer.0.bias", 2);
        h_g = add("final_layer.1.weight", 2); h_b = add("final_layer.1.bias", 2);
        h_m = add("final_layer.1.running_mean", 2); h_v = add("final_layer.1.running_var", 2);
        h_c2w = add("final_layer.3.weight", 2 * k2); h_c2b = add("final_layer.3.bias", 1);
    }
};

static int check_model_desc(const swf_model_desc* d) {
    if (!d) return fail(SWF_ERR_NULL, "model desc is NULL");
    if (d->levels <= 0 || d->levels > SWF_MAX_LEVELS) return fail(SWF_ERR_BAD_SHAPE, "levels %d outside 1..%d", d->levels, SWF_MAX_LEVELS);
    if (d->heads <= 0 || d->mlp_ratio <= 0 || d->win_h <= 0 || d->win_w <= 0 || d->merge_h <= 0 || d->merge_w <= 0)
        return fail(SWF_ERR_BAD_SHAPE, "non-positive model dims");
    if (d->head_ksize <= 0 || d->head_ksize % 2 == 0) return fail(SWF_ERR_UNSUPPORTED, "final conv kernel must be odd");
    for (int s = 0; s < d->levels; ++s) {
        if (d->in_dims[s] <= 0 || d->out_dims[s] <= 0 || d->head_dim[s] <= 0) return fail(SWF_ERR_BAD_SHAPE, "non-positive dims at level %d", s);
        if (s > 0 && d->in_dims[s] != d->out_dims[s - 1])
            return fail(SWF_ERR_BAD_SHAPE, "in_dims[%d]=%d != out_dims[%d]=%d", s, d->in_dims[s], s - 1, d->out_dims[s - 1]);
    }
    if (d->in_dims[0] != 1) return fail(SWF_ERR_UNSUPPORTED, "in_dims[0] must be 1 (single-channel IR / Y inputs, final head takes 2 channels)");
    return SWF_OK;
}

// Layouts are cached per descriptor; the key is a canonical copy (unused levels and struct padding
// zeroed) so that callers need not zero-initialise the struct.
static swf_model_desc canonical_desc(const swf_model_desc* d) {
    swf_model_desc c;
    std::memset(&c, 0, sizeof(c));
    c.levels = d->levels;
    for (int s = 0; s < d->levels; ++s) { c.in_dims[s] = d->in_dims[s]; c.out_dims[s] = d->out_dims[s]; c.head_dim[s] = d->head_dim[s]; }
    c.heads = d->heads; c.mlp_ratio = d->mlp_ratio;
    c.win_h = d->win_h; c.win_w = d->win_w; c.merge_h = d->merge_h; c.merge_w = d->merge_w;
    c.head_ksize = d->head_ksize;
    c.precision = 0;   // the parameter layout does not depend on the arithmetic mode
    return c;
}

static const ModelLayout* get_layout(const swf_model_desc* d_in) {
    const swf_model_desc canon = canonical_desc(d_in);
    const swf_model_desc* d = &canon;
    static std::mutex mu;
    static std::vector<ModelLayout*> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (ModelLayout* l : cache)
        if (std::memcmp(&l->desc, d, sizeof(*d)) == 0) return l;
    ModelLayout* l = new ModelLayout();
    l->build(*d);
    cache.push_back(l);
    return l;
}

static swf_block_stream_params make_stream_params(const float* a, const BlockOff& o) {
    swf_block_stream_params p;
    p.ln1 = {a + o.ln1g, a + o.ln1b};
    p.attn.q = {a + o.wq, a + o.bq}; p.attn.k = {a + o.wk, a + o.bk}; p.attn.v = {a + o.wv, a + o.bv};
    p.attn.proj = {a + o.wp, a + o.bp}; p.attn.bias_table = a + o.table;
    p.ln2 = {a + o.ln2g, a + o.ln2b};
    p.fc1 = {a + o.w1, a + o.b1}; p.fc2 = {a + o.w2, a + o.b2};
    return p;
}

struct LevelShape { int Hin, Win, Hm, Wm, Ho, Wo; };

static int model_shapes(const swf_model_desc* d, int H, int W, LevelShape* ls) {
    int h = H, w = W;
    for (int s = 0; s < d->levels; ++s) {
        ls[s].Hin = h; ls[s].Win = w;
        SWF_TRY(merge_shapes(h, w, d->merge_h, d->merge_w, d->win_h, d->win_w, &ls[s].Hm, &ls[s].Wm, &ls[s].Ho, &ls[s].Wo));
        h = ls[s].Ho; w = ls[s].Wo;
    }
    if (d->head_ksize / 2 >= H || d->head_ksize / 2 >= W) return fail(SWF_ERR_PAD, "final conv reflect pad >= image");
    return SWF_OK;
}

static swf_block_desc level_block_desc(const swf_model_desc* d, int lvl, bool encoder) {
    swf_block_desc b;
    b.attn.channels = d->out_dims[lvl];
    b.attn.heads = d->heads;
    b.attn.head_dim = d->head_dim[lvl];
    b.attn.win_h = d->win_h; b.attn.win_w = d->win_w;
    b.attn.shift = 0;
    b.hidden = (encoder ? d->out_dims[lvl] : d->in_dims[lvl]) * d->mlp_ratio;
    b.cross = 0;
    b.precision = d->precision;
    b.schedule = d->schedule;
    return b;
}

// `packed`: nullptr, or the 8 packed weight images of the stage laid out [block 0..3][stream x,y] at a
// stride of window_block_packed_bytes(desc)
static int block_pair4_impl(const swf_block_desc* desc, const swf_block_stream_params* px, const swf_block_stream_params* py,
                            const float* x_in, const float* y_in, float* x_out, float* y_out, int B, int H, int W,
                            void* workspace, size_t workspace_bytes, hipStream_t stream, const char* packed = nullptr,
                            const char* after = nullptr, size_t after_pb = 0, int32_t* equal_flags = nullptr,
                            const swf_block_stream_params* const* stage_next = nullptr, bool* ln1_io = nullptr) {
    // `stage_next` / `ln1_io` (deep levels): the first block's parameters of the stage that runs next on the same workspace with the
    // same map (the decoder stage behind the deepest encoder stage), so that this stage's last MLP reduce leaves that block's LN1
    // planes; *ln1_io says on entry whether this stage's first block finds its planes in place, on return whether the next does
    // `equal_flags` (device, 2 entries pre-set to 1, or nullptr): cleared when the inputs of the stage's two cross blocks
    // differ somewhere — the reference's first-forward check `(x == y).all()` (a005:111-118)
    // `after`: packed images (x, then y at + after_pb) of the first block of the NEXT stage when that is a fused-kernel stage
    // too: the last block of this stage warms them
    const float* xi = x_in;
    const float* yi = y_in;
    const size_t pb = packed ? block_packed_bytes(*desc) : 0;
    bool ln1_ready = ln1_io ? *ln1_io : false;   // deep levels: block i's MLP reduce also writes block i+1's LN1 planes
    // Kernels that cannot run a cross block in place (window_block_out_of_place): the two cross blocks ping-pong through two
    // temporary maps at the END of the workspace (block 2: maps -> temporaries, block 3: temporaries -> outputs) instead of
    // each going through basic_block_impl's temporary-and-copy route
    float *tx = nullptr, *ty = nullptr;
    size_t ws_left = workspace_bytes;
    {
        swf_block_desc dc = *desc;
        dc.cross = 1;
        const size_t map_bytes = (size_t)B * H * W * desc->attn.channels * 4;
        if (desc->precision == SWF_PREC_FAST && py && window_block_supported(dc, B, H, W) && window_block_out_of_place(dc) && workspace &&
            workspace_bytes >= 2 * map_bytes + 512 + 2 * window_block_packed_bytes(dc)) {
            ws_left = (workspace_bytes - 2 * map_bytes) & ~size_t(255);
            tx = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_left);
            ty = tx + map_bytes / 4;
        }
    }
    for (int i = 0; i < 4; ++i) {
        swf_block_desc d = *desc;
        d.cross = i >= 2;          // self pair first, then cross pair (a012:72-73)
        d.attn.shift = i & 1;      // normal window, then shifted window (a009:102-105)
        const void* pkx = (packed && pb) ? packed + (size_t)(2 * i) * pb : nullptr;
        const void* pky = (packed && pb) ? packed + (size_t)(2 * i + 1) * pb : nullptr;
        const void* nkx = (packed && pb && i < 3) ? packed + (size_t)(2 * i + 2) * pb : (i == 3 ? after : nullptr);   // next block: warmed in L2
        const void* nky = (packed && pb && i < 3) ? packed + (size_t)(2 * i + 3) * pb : (i == 3 && after ? after + after_pb : nullptr);
        const swf_block_stream_params* nxt[2] = {i < 3 ? &px[i + 1] : nullptr, (i < 3 && py) ? &py[i + 1] : nullptr};
        const bool chain_out = i == 3 && stage_next && stage_next[0] && (!py || stage_next[1]);
        if (chain_out) { nxt[0] = stage_next[0]; nxt[1] = py ? stage_next[1] : nullptr; }
        if (equal_flags && d.cross && py)
            SWF_TRY(launch_all_equal(xi, yi, (int64_t)B * H * W * desc->attn.channels, equal_flags + (i - 2), stream));
        float* xo = (tx && i == 2) ? tx : x_out;
        float* yo = (tx && i == 2) ? ty : y_out;
        SWF_TRY(basic_block_impl(&d, &px[i], py ? &py[i] : nullptr, xi, yi, xo, yo, B, H, W, workspace, tx ? ws_left : workspace_bytes, stream, pkx, pky,
                                 nkx, nky, i == 3 ? after_pb : 0, (i < 3 || chain_out) ? nxt : nullptr, &ln1_ready));
        xi = xo; yi = yo;
    }
    if (ln1_io) *ln1_io = ln1_ready;
    return SWF_OK;
}

// byte offset of each stage's packed images inside the model-level packed buffer (encoder stages first)
struct PackedPlan {
    size_t enc[SWF_MAX_LEVELS], dec[SWF_MAX_LEVELS], total;
    bool enc_on[SWF_MAX_LEVELS], dec_on[SWF_MAX_LEVELS];
    // patch layers (register-resident kernel, or the deep-level kernel where that one does not cover the shape): two per-stream
    // images of penc_b / pdec_b bytes each, 0 = layer not covered
    size_t penc[SWF_MAX_LEVELS], pdec[SWF_MAX_LEVELS], penc_b[SWF_MAX_LEVELS], pdec_b[SWF_MAX_LEVELS];
};
static PackedPlan packed_plan(const swf_model_desc* d) {
    PackedPlan p{};
    size_t off = 0;
    for (int s = 0; s < d->levels; ++s) {
        swf_block_desc be = level_block_desc(d, s, true);
        be.precision = SWF_PREC_FAST;
        const size_t pb = block_packed_bytes(be);
        p.enc_on[s] = pb > 0; p.enc[s] = off; off += 8 * pb;
    }
    for (int j = 0; j < d->levels; ++j) {
        swf_block_desc bd = level_block_desc(d, d->levels - 1 - j, false);
        bd.precision = SWF_PREC_FAST;
        const size_t pb = block_packed_bytes(bd);
        p.dec_on[j] = pb > 0; p.dec[j] = off; off += 8 * pb;
    }
    for (int s = 0; s < d->levels; ++s) {
        p.penc_b[s] = patch_rr_packed_bytes(0, d->in_dims[s], d->out_dims[s], d->merge_h, d->merge_w);
        if (!p.penc_b[s]) p.penc_b[s] = deep_patch_packed_bytes(0, d->in_dims[s], d->out_dims[s], d->merge_h, d->merge_w);
        p.penc[s] = off; off += 2 * p.penc_b[s];
    }
    for (int j = 0; j < d->levels; ++j) {
        const int lvl = d->levels - 1 - j;
        p.pdec_b[j] = patch_rr_packed_bytes(1, d->out_dims[lvl], d->in_dims[lvl], d->merge_h, d->merge_w);
        if (!p.pdec_b[j]) p.pdec_b[j] = deep_patch_packed_bytes(1, d->out_dims[lvl], d->in_dims[lvl], d->merge_h, d->merge_w);
        p.pdec[j] = off; off += 2 * p.pdec_b[j];
    }
    p.total = off;
    return p;
}

}  // namespace swf

using namespace swf;

extern "C" {

int swf_version(void) { return SWF_VERSION_MAJOR * 1000 + SWF_VERSION_MINOR; }
const char* swf_last_error_string(void) { return err_buf(); }
const char* swf_status_string(int status) {
    switch (status) {
        case SWF_OK: return "ok";
        case SWF_ERR_NULL: return "null pointer";
        case SWF_ERR_BAD_SHAPE: return "bad shape";
        case SWF_ERR_PAD: return "reflect padding >= dimension";
        case SWF_ERR_UNSUPPORTED: return "unsupported configuration";
        case SWF_ERR_WORKSPACE: return "workspace too small";
        case SWF_ERR_HIP: return "HIP error";
        default: return "unknown status";
    }
}

// ---- fast tier of the stand-alone module entries at level-0 width (C = 24): the block kernel with the other half compiled out ----
// (levels 0-2: C = 24 / 48 / 96, 8 heads of C / 8, 8x8 or 7x7 windows; hidden widths of the encoder / decoder blocks)
static bool half_attn_shape(const swf_attn_desc& a, int H, int W) {
    return (a.channels == 24 || a.channels == 48 || a.channels == 96) && a.heads == 8 && a.head_dim * 8 == a.channels && a.win_h == a.win_w &&
           (a.win_h == 8 || a.win_h == 7) && H % a.win_h == 0 && W % a.win_w == 0;
}
static int half_attn_hidden(int C) { return 4 * C; }   // the attention half runs on the wide-MLP image layout (fc sections unused)
static size_t half_packed_bytes(int C, int hid) {
    return C == 24 ? win24_half_packed_bytes(C, hid) : C == 48 ? win48_half_packed_bytes(C, hid) : C == 96 ? win96_half_packed_bytes(C, hid) : 0;
}
static int half_pack(const swf_block_desc& bd, const swf_block_stream_params& sx, const swf_block_stream_params& sy, char* pk, size_t pb, hipStream_t st) {
    return bd.attn.channels == 24 ? pack_win24(bd, sx, sy, pk, pk + pb, st)
         : bd.attn.channels == 48 ? pack_win48(bd, sx, sy, pk, pk + pb, st) : pack_win96(bd, sx, sy, pk, pk + pb, st);
}
static int half_launch(const swf_block_desc& bd, int mode, int raw, const char* pk, size_t pb, const float* x_in, const float* y_in, float* x_out,
                       float* y_out, int B, int H, int W, int nx, int ny, hipStream_t st) {
    return bd.attn.channels == 24 ? launch_win24_half(bd, mode, raw, pk, pk + pb, x_in, y_in, x_out, y_out, B, H, W, nx, ny, st)
         : bd.attn.channels == 48 ? launch_win48_half(bd, mode, raw, pk, pk + pb, x_in, y_in, x_out, y_out, B, H, W, nx, ny, st)
                                  : launch_win96_half(bd, mode, raw, pk, pk + pb, x_in, y_in, x_out, y_out, B, H, W, nx, ny, st);
}

// MLP half on window24 / window48 / window96_kernel <HID, 8, MLP half, raw>: tokens as flat lists.  Dual path: one list per stream.  Single path: the one
// list is split between the kernel's two stream slots (same weights).  SWF_ERR_UNSUPPORTED = shape not covered (the caller falls back).
static int mlp_half24(int C, int hid, int raw, const swf_block_stream_params* px, const swf_block_stream_params* py, const float* x_in,
                      const float* y_in, float* x_out, float* y_out, int64_t N, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const size_t pb = half_packed_bytes(C, hid);
    if (!pb || N <= 0 || N * C * 4 >= (int64_t(1) << 31) || !workspace || workspace_bytes < 2 * pb) return SWF_ERR_UNSUPPORTED;
    swf_block_desc bd{};
    bd.attn = swf_attn_desc{C, 8, C / 8, 8, 8, 0};
    bd.hidden = hid; bd.cross = 0; bd.precision = SWF_PREC_FAST;
    swf_block_stream_params sx = *px, sy = py ? *py : *px;
    sx.attn = swf_attn_params{}; sy.attn = swf_attn_params{};
    sx.ln1 = sy.ln1 = swf_norm{nullptr, nullptr};
    if (raw) sx.ln2 = sy.ln2 = swf_norm{nullptr, nullptr};
    char* pk = static_cast<char*>(workspace);
    SWF_TRY(half_pack(bd, sx, sy, pk, pb, stream));
    if (py) return half_launch(bd, WIN24_HALF_MLP, raw, pk, pb, x_in, y_in, x_out, y_out, 1, 1, 1, (int)N, (int)N, stream);
    const int64_t n0 = std::min<int64_t>(N, ((N + 1) / 2 + 63) / 64 * 64);
    return half_launch(bd, WIN24_HALF_MLP, raw, pk, pb, x_in, x_in + n0 * C, x_out, x_out + n0 * C, 1, 1, 1, (int)n0, (int)(N - n0), stream);
}

size_t swf_window_attention_workspace_bytes(const swf_attn_desc* desc, int32_t B, int32_t H, int32_t W) {
    if (!desc || B <= 0 || H <= 0 || W <= 0) return 0;
    return std::max(attention_generic_ws(*desc, 1, B, H, W), 2 * half_packed_bytes(desc->channels, half_attn_hidden(desc->channels)) + 512);
}

static int window_attention_impl(const swf_attn_desc* desc, int precision, const swf_attn_params* p, const float* q, const float* k,
                                 const float* v, const float* residual, float* out, int32_t B, int32_t H, int32_t W,
                                 void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    SWF_TRY(check_attn_desc(desc, B, H, W));
    if (precision != SWF_PREC_FP32 && precision != SWF_PREC_FAST) return fail(SWF_ERR_BAD_SHAPE, "window_attention: unknown precision %d", precision);
    if (!p || !q || !k || !v || !out) return fail(SWF_ERR_NULL, "window_attention: NULL tensor or params");
    if (!p->q.weight || !p->k.weight || !p->v.weight || !p->proj.weight || !p->bias_table)
        return fail(SWF_ERR_NULL, "window_attention: NULL weight");
    const int hid_a = half_attn_hidden(desc->channels);
    const size_t pbh = half_packed_bytes(desc->channels, hid_a);
    if (precision == SWF_PREC_FAST && k == v && !residual && half_attn_shape(*desc, H, W) && (int64_t)B * H * W * desc->channels * 4 < (int64_t(1) << 31) &&
        pbh && workspace && workspace_bytes >= 2 * pbh && out != q && out != k) {
        // window24/48_kernel<.., attention half, RAW>: stream 0 = the queries and the output, stream 1 = the key / value tensor
        swf_block_desc bd{*desc, hid_a, 1, SWF_PREC_FAST};
        swf_block_stream_params sp{};
        sp.attn = *p;
        char* pk = static_cast<char*>(workspace);
        SWF_TRY(half_pack(bd, sp, sp, pk, pbh, as_stream(stream)));
        return half_launch(bd, WIN24_HALF_ATTN, 1, pk, pbh, q, k, out, nullptr, B, H, W, 0, 0, as_stream(stream));
    }
    Carver ws(workspace, workspace_bytes);
    const swf_attn_params* prm[2] = {p, nullptr};
    const float* qs[2] = {q, nullptr};
    const float* ks[2] = {k, nullptr};
    const float* vs[2] = {v, nullptr};
    const float* rs[2] = {residual, nullptr};
    float* os[2] = {out, nullptr};
    return attention_generic(*desc, 1, prm, qs, ks, vs, residual ? rs : nullptr, os, B, H, W, ws, as_stream(stream),
                             precision == SWF_PREC_FAST ? 1 : 0);
}

int swf_window_attention_fwd(const swf_attn_desc* desc, const swf_attn_params* p, const float* q, const float* k,
                             const float* v, const float* residual, float* out, int32_t B, int32_t H, int32_t W,
                             void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    return window_attention_impl(desc, SWF_PREC_FP32, p, q, k, v, residual, out, B, H, W, workspace, workspace_bytes, stream);
}

int swf_window_attention_fwd_prec(const swf_attn_desc* desc, int32_t precision, const swf_attn_params* p, const float* q, const float* k,
                                  const float* v, const float* residual, float* out, int32_t B, int32_t H, int32_t W,
                                  void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    return window_attention_impl(desc, precision, p, q, k, v, residual, out, B, H, W, workspace, workspace_bytes, stream);
}

size_t swf_basic_block_workspace_bytes(const swf_block_desc* desc, int32_t B, int32_t H, int32_t W) {
    if (!desc || B <= 0 || H <= 0 || W <= 0) return 0;
    return std::max(std::max(block_generic_ws(desc, 2, B, H, W), window_block_workspace_bytes(*desc, B, H, W)),
                    2 * std::max(half_packed_bytes(desc->attn.channels, half_attn_hidden(desc->attn.channels)), half_packed_bytes(desc->attn.channels, desc->hidden)) + 512);
}

int swf_attn_halfblock_fwd(const swf_block_desc* desc, const swf_block_stream_params* px, const swf_block_stream_params* py,
                           const float* x_in, const float* y_in, float* x_out, float* y_out, int32_t B, int32_t H, int32_t W,
                           void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    SWF_TRY(check_block(desc, px, py, x_in, y_in, x_out, y_out, B, H, W, true, false));
    const int hid_a = half_attn_hidden(desc->attn.channels);
    const size_t pbh = half_packed_bytes(desc->attn.channels, hid_a);
    if (desc->precision == SWF_PREC_FAST && half_attn_shape(desc->attn, H, W) && (int64_t)B * H * W * desc->attn.channels * 4 < (int64_t(1) << 31) &&
        pbh && workspace && workspace_bytes >= 2 * pbh) {
        // window24/48_kernel<.., attention half>: LN1 + Q/K/V + attention + projection + residual of both streams in one launch.  A single-path
        // block runs its stream as stream 0; stream 1 mirrors it with its stores dropped.
        swf_block_desc bd = *desc;
        bd.hidden = hid_a;
        bd.cross = desc->cross && py;   // a single path ignores cross (a002:83)
        swf_block_stream_params sx = *px, sy = py ? *py : *px;
        sx.fc1 = sx.fc2 = swf_linear{nullptr, nullptr}; sy.fc1 = sy.fc2 = swf_linear{nullptr, nullptr};
        sx.ln2 = sy.ln2 = swf_norm{nullptr, nullptr};
        char* pk = static_cast<char*>(workspace);
        SWF_TRY(half_pack(bd, sx, sy, pk, pbh, as_stream(stream)));
        return half_launch(bd, WIN24_HALF_ATTN, 0, pk, pbh, x_in, py ? y_in : x_in, x_out, py ? y_out : nullptr, B, H, W, 0, 0, as_stream(stream));
    }
    Carver ws(workspace, workspace_bytes);
    return attn_halfblock_generic(desc, px, py, x_in, y_in, x_out, y_out, B, H, W, ws, as_stream(stream));
}

int swf_mlp_halfblock_fwd(const swf_block_desc* desc, const swf_block_stream_params* px, const swf_block_stream_params* py,
                          const float* x_in, const float* y_in, float* x_out, float* y_out, int32_t B, int32_t H, int32_t W,
                          void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    SWF_TRY(check_block(desc, px, py, x_in, y_in, x_out, y_out, B, H, W, false, true));
    if (desc->precision == SWF_PREC_FAST) {
        const int st = mlp_half24(desc->attn.channels, desc->hidden, 0, px, py, x_in, y_in, x_out, y_out, (int64_t)B * H * W, workspace, workspace_bytes,
                                  as_stream(stream));
        if (st != SWF_ERR_UNSUPPORTED) return st;
    }
    Carver ws(workspace, workspace_bytes);
    return mlp_halfblock_generic(desc, px, py, x_in, y_in, x_out, y_out, B, H, W, ws, as_stream(stream));
}

size_t swf_mlp_workspace_bytes(int32_t precision, int64_t tokens, int32_t channels, int32_t hidden) {
    if (tokens <= 0 || channels <= 0 || hidden <= 0) return 0;
    size_t generic = carve_bytes({tokens * hidden}) + std::max(swf_linear_workspace_bytes(precision, tokens, channels, hidden),
                                                               swf_linear_workspace_bytes(precision, tokens, hidden, channels));
    return std::max(generic, 2 * half_packed_bytes(channels, hidden) + 512);
}

int swf_mlp_fwd(int32_t precision, const swf_block_stream_params* px, const swf_block_stream_params* py, const float* x_in, const float* y_in,
                float* x_out, float* y_out, int64_t tokens, int32_t channels, int32_t hidden, void* workspace, size_t workspace_bytes,
                swf_stream_t stream) {
    if (precision != SWF_PREC_FP32 && precision != SWF_PREC_FAST) return fail(SWF_ERR_BAD_SHAPE, "mlp: unknown precision %d", precision);
    if (!px || !px->fc1.weight || !px->fc2.weight || !x_in || !x_out) return fail(SWF_ERR_NULL, "mlp: NULL argument");
    if (py && (!py->fc1.weight || !py->fc2.weight || !y_in || !y_out)) return fail(SWF_ERR_NULL, "mlp: NULL y-stream argument");
    if (tokens <= 0 || tokens > INT32_MAX || channels <= 0 || hidden <= 0) return fail(SWF_ERR_BAD_SHAPE, "mlp: bad sizes");
    if (precision == SWF_PREC_FAST) {
        const int st = mlp_half24(channels, hidden, 1, px, py, x_in, y_in, x_out, y_out, tokens, workspace, workspace_bytes, as_stream(stream));
        if (st != SWF_ERR_UNSUPPORTED) return st;
    }
    // two linear layers per stream through the hidden activations in the workspace
    Carver ws(workspace, workspace_bytes);
    float* hid = ws.floats(tokens * hidden);
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "mlp: workspace too small (need %zu B)", (size_t)swf_mlp_workspace_bytes(precision, tokens, channels, hidden));
    char* rest = static_cast<char*>(workspace) + align_up(ws.used, 256);
    const size_t rest_bytes = workspace_bytes > align_up(ws.used, 256) ? workspace_bytes - align_up(ws.used, 256) : 0;
    const swf_block_stream_params* pp[2] = {px, py};
    const float* in[2] = {x_in, y_in};
    float* out[2] = {x_out, y_out};
    for (int s = 0; s < (py ? 2 : 1); ++s) {
        SWF_TRY(swf_linear_fwd_prec(&pp[s]->fc1, precision, in[s], nullptr, hid, tokens, channels, hidden, 1, rest, rest_bytes, stream));
        SWF_TRY(swf_linear_fwd_prec(&pp[s]->fc2, precision, hid, nullptr, out[s], tokens, hidden, channels, 0, rest, rest_bytes, stream));
    }
    return SWF_OK;
}

int swf_basic_block_fwd(const swf_block_desc* desc, const swf_block_stream_params* px, const swf_block_stream_params* py,
                        const float* x_in, const float* y_in, float* x_out, float* y_out, int32_t B, int32_t H, int32_t W,
                        void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    SWF_TRY(check_block(desc, px, py, x_in, y_in, x_out, y_out, B, H, W, true, true));
    return basic_block_impl(desc, px, py, x_in, y_in, x_out, y_out, B, H, W, workspace, workspace_bytes, as_stream(stream));
}

size_t swf_basic_block_packed_bytes(const swf_block_desc* desc) {
    if (!desc || desc->precision != SWF_PREC_FAST) return 0;
    return 2 * window_block_packed_bytes(*desc);
}

int swf_basic_block_pack(const swf_block_desc* desc, const swf_block_stream_params* px, const swf_block_stream_params* py,
                         void* packed, size_t packed_bytes, swf_stream_t stream) {
    if (!desc) return fail(SWF_ERR_NULL, "desc is NULL");
    SWF_TRY(check_stream_params(px, "x-stream", true, true));
    SWF_TRY(check_stream_params(py, "y-stream", true, true));
    const size_t pb = window_block_packed_bytes(*desc);
    if (pb == 0 || desc->precision != SWF_PREC_FAST) return fail(SWF_ERR_UNSUPPORTED, "no fused kernel for C=%d hidden=%d", desc->attn.channels, desc->hidden);
    if (!packed || packed_bytes < 2 * pb) return fail(SWF_ERR_WORKSPACE, "packed buffer too small (need %zu B)", 2 * pb);
    return pack_window_block(*desc, *px, *py, packed, static_cast<char*>(packed) + pb, as_stream(stream));
}

int swf_basic_block_fwd_packed(const swf_block_desc* desc, const void* packed, const float* x_in, const float* y_in,
                               float* x_out, float* y_out, int32_t B, int32_t H, int32_t W, swf_stream_t stream) {
    if (!desc || !packed || !x_in || !y_in || !x_out || !y_out) return fail(SWF_ERR_NULL, "basic_block_fwd_packed: NULL argument");
    SWF_TRY(check_attn_desc(&desc->attn, B, H, W));
    if (!window_block_supported(*desc, B, H, W)) return fail(SWF_ERR_UNSUPPORTED, "no fused kernel for this block shape");
    const size_t pb = window_block_packed_bytes(*desc);
    return launch_window_block(*desc, packed, static_cast<const char*>(packed) + pb, x_in, y_in, x_out, y_out, B, H, W, as_stream(stream));
}

size_t swf_basic_block_bwd_workspace_bytes(const swf_block_desc* desc, int32_t B, int32_t H, int32_t W) {
    if (!desc || B <= 0 || H <= 0 || W <= 0) return 0;
    return basic_block_bwd_ws(*desc, 2, B, H, W);
}

int swf_basic_block_bwd(const swf_block_desc* desc, const swf_block_stream_params* px, const swf_block_stream_params* py, const float* x_in,
                        const float* y_in, const float* gx_out, const float* gy_out, float* gx_in, float* gy_in, const swf_block_stream_grads* gpx,
                        const swf_block_stream_grads* gpy, int32_t B, int32_t H, int32_t W, void* workspace, size_t workspace_bytes,
                        swf_stream_t stream) {
    SWF_TRY(check_block(desc, px, py, x_in, y_in, gx_in, gy_in, B, H, W, true, true));
    if (!gx_out || (py && !gy_out)) return fail(SWF_ERR_NULL, "basic_block_bwd: NULL output gradient");
    return basic_block_bwd(*desc, px, py, x_in, y_in, gx_out, gy_out, gx_in, gy_in, gpx, gpy, B, H, W, workspace, workspace_bytes, as_stream(stream));
}

size_t swf_window_attention_bwd_workspace_bytes(const swf_attn_desc* desc, int32_t B, int32_t H, int32_t W) {
    if (!desc || B <= 0 || H <= 0 || W <= 0 || desc->win_h <= 0 || desc->win_w <= 0 || desc->channels <= 0 || desc->heads <= 0 || desc->head_dim <= 0) return 0;
    return window_attention_bwd_ws(*desc, B, H, W);
}

int swf_window_attention_bwd(const swf_attn_desc* desc, const swf_attn_params* p, const float* q, const float* k, const float* v, const float* gout,
                             float* gq, float* gk, float* gv, const swf_attn_grads* gp, int32_t B, int32_t H, int32_t W, void* workspace,
                             size_t workspace_bytes, swf_stream_t stream) {
    if (!desc || !p || !q || !k || !v || !gout || !gq || !gk || !gv) return fail(SWF_ERR_NULL, "window_attention_bwd: NULL argument");
    if (!p->q.weight || !p->k.weight || !p->v.weight || !p->proj.weight || !p->bias_table) return fail(SWF_ERR_NULL, "window_attention_bwd: NULL parameter");
    if (B <= 0 || H <= 0 || W <= 0 || desc->channels <= 0 || desc->heads <= 0 || desc->head_dim <= 0 || desc->win_h <= 0 || desc->win_w <= 0)
        return fail(SWF_ERR_BAD_SHAPE, "window_attention_bwd: bad sizes");
    if (H % desc->win_h || W % desc->win_w)
        return fail(SWF_ERR_BAD_SHAPE, "window_attention_bwd: map %dx%d is not a multiple of the window %dx%d", H, W, desc->win_h, desc->win_w);
    if (gq == gk || gq == gv || gk == gv) return fail(SWF_ERR_UNSUPPORTED, "window_attention_bwd: gq, gk, gv must be three buffers");
    return window_attention_bwd(*desc, *p, q, k, v, gout, gq, gk, gv, gp, B, H, W, workspace, workspace_bytes, as_stream(stream));
}

size_t swf_mlp_bwd_workspace_bytes(int64_t tokens, int32_t channels, int32_t hidden) {
    if (tokens <= 0 || channels <= 0 || hidden <= 0) return 0;
    return mlp_bwd_ws(tokens, channels, hidden);
}

int swf_mlp_bwd(const swf_linear* fc1, const swf_linear* fc2, const float* x, const float* gout, float* gx, const swf_linear_grad* gfc1,
                const swf_linear_grad* gfc2, int64_t tokens, int32_t channels, int32_t hidden, void* workspace, size_t workspace_bytes,
                swf_stream_t stream) {
    if (!fc1 || !fc2 || !fc1->weight || !fc2->weight || !x || !gout || !gx) return fail(SWF_ERR_NULL, "mlp_bwd: NULL argument");
    if (tokens <= 0 || channels <= 0 || hidden <= 0) return fail(SWF_ERR_BAD_SHAPE, "mlp_bwd: bad sizes");
    return mlp_bwd(*fc1, *fc2, x, gout, gx, gfc1, gfc2, tokens, channels, hidden, workspace, workspace_bytes, as_stream(stream));
}

size_t swf_layernorm_bwd_workspace_bytes(int64_t tokens, int32_t C) {
    if (tokens <= 0 || C <= 0) return 0;
    return layernorm_bwd_ws(tokens, C);
}

int swf_layernorm_bwd(const swf_norm* ln, const float* x, const float* gout, float* gx, const swf_norm_grad* gp, int64_t tokens, int32_t C,
                      void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    if (!ln || !ln->gamma || !ln->beta || !x || !gout || !gx) return fail(SWF_ERR_NULL, "layernorm_bwd: NULL argument");
    if (tokens <= 0 || C <= 0) return fail(SWF_ERR_BAD_SHAPE, "layernorm_bwd: bad sizes");
    return layernorm_bwd(*ln, x, gout, gx, gp, tokens, C, workspace, workspace_bytes, as_stream(stream));
}

size_t swf_patch_layer_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t merge_h, int32_t merge_w, int32_t encoder) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || merge_h <= 0 || merge_w <= 0) return 0;
    return patch_bwd_ws(B, H, W, Cin, Cout, merge_h, merge_w, encoder);
}

int swf_patch_layer_bwd(const swf_patch_params* p, const float* in, const float* gout, float* gin, const swf_patch_grads* gp, int32_t B, int32_t H,
                        int32_t W, int32_t Cin, int32_t Cout, int32_t merge_h, int32_t merge_w, int32_t encoder, void* workspace,
                        size_t workspace_bytes, swf_stream_t stream) {
    if (!p || !p->conv.weight || !p->ln.gamma || !p->ln.beta || !in || !gout || !gin) return fail(SWF_ERR_NULL, "patch_layer_bwd: NULL argument");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || merge_h <= 0 || merge_w <= 0) return fail(SWF_ERR_BAD_SHAPE, "patch_layer_bwd: bad sizes");
    return patch_bwd(*p, in, gout, gin, gp, B, H, W, Cin, Cout, merge_h, merge_w, encoder, workspace, workspace_bytes, as_stream(stream));
}

int swf_reflect_pad_bwd(const float* gout, float* gin, int32_t B, int32_t H, int32_t W, int32_t C, int32_t pad_h, int32_t pad_w, swf_stream_t stream) {
    if (!gout || !gin) return fail(SWF_ERR_NULL, "reflect_pad_bwd: NULL tensor");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(SWF_ERR_BAD_SHAPE, "reflect_pad_bwd: empty tensor");
    return reflect_pad_bwd(gout, gin, B, H, W, C, pad_h, pad_w, as_stream(stream));
}

size_t swf_final_head_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t ksize) {
    if (B <= 0 || H <= 0 || W <= 0 || ksize <= 0) return 0;
    return head_bwd_ws(B, H, W, ksize);
}

int swf_final_head_batch_stats(const swf_head_params* p, const float* x, const float* y, float* mean, float* var, float* running_mean,
                               float* running_var, float momentum, int32_t B, int32_t H, int32_t W, int32_t ksize, void* workspace,
                               size_t workspace_bytes, swf_stream_t stream) {
    if (!p || !p->conv1_w || !x || !y || !mean || !var) return fail(SWF_ERR_NULL, "final_head_batch_stats: NULL argument");
    if (B <= 0 || H <= 0 || W <= 0) return fail(SWF_ERR_BAD_SHAPE, "final_head_batch_stats: empty tensor");
    return head_batch_stats(*p, x, y, mean, var, running_mean, running_var, momentum, B, H, W, ksize, workspace, workspace_bytes, as_stream(stream));
}

int swf_final_head_bwd(const swf_head_params* p, const float* x, const float* y, const float* gout, float* gx, float* gy, const swf_head_grads* gp,
                       int32_t B, int32_t H, int32_t W, int32_t ksize, int32_t batch_stats, void* workspace, size_t workspace_bytes,
                       swf_stream_t stream) {
    if (!p || !p->conv1_w || !p->conv2_w || !p->bn_gamma || !p->bn_beta || !p->bn_mean || !p->bn_var || !x || !y || !gout || !gx || !gy)
        return fail(SWF_ERR_NULL, "final_head_bwd: NULL argument");
    if (B <= 0 || H <= 0 || W <= 0) return fail(SWF_ERR_BAD_SHAPE, "final_head_bwd: empty tensor");
    return head_bwd(*p, x, y, gout, gx, gy, gp, B, H, W, ksize, batch_stats, workspace, workspace_bytes, as_stream(stream));
}

int swf_add_fwd(const float* a, const float* b, float* out, int64_t count, swf_stream_t stream) {
    if (!a || !b || !out) return fail(SWF_ERR_NULL, "add: NULL tensor");
    if (count <= 0) return fail(SWF_ERR_BAD_SHAPE, "add: empty tensor");
    return add_tensors(a, b, out, count, as_stream(stream));
}

int swf_block_pair4_fwd(const swf_block_desc* desc, const swf_block_stream_params* px, const swf_block_stream_params* py,
                        const float* x_in, const float* y_in, float* x_out, float* y_out, int32_t B, int32_t H, int32_t W,
                        void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    if (!px) return fail(SWF_ERR_NULL, "block params are NULL");
    for (int i = 0; i < 4; ++i)
        SWF_TRY(check_block(desc, &px[i], py ? &py[i] : nullptr, x_in, y_in, x_out, y_out, B, H, W, true, true));
    return block_pair4_impl(desc, px, py, x_in, y_in, x_out, y_out, B, H, W, workspace, workspace_bytes, as_stream(stream));
}

int swf_merge_out_shape(int32_t H, int32_t W, int32_t merge_h, int32_t merge_w, int32_t win_h, int32_t win_w, int32_t* Hm,
                        int32_t* Wm, int32_t* Ho, int32_t* Wo) {
    if (!Hm || !Wm || !Ho || !Wo) return fail(SWF_ERR_NULL, "output pointer is NULL");
    return merge_shapes(H, W, merge_h, merge_w, win_h, win_w, Hm, Wm, Ho, Wo);
}

size_t swf_patch_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t merge_h,
                                 int32_t merge_w, int32_t win_h, int32_t win_w, int32_t encoder) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return patch_ws(1, B, H, W, Cin, Cout, merge_h, merge_w, win_h, win_w, encoder);
}

int swf_patch_merge_fwd(const swf_patch_params* p, const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                        int32_t Cout, int32_t merge_h, int32_t merge_w, int32_t win_h, int32_t win_w, void* workspace,
                        size_t workspace_bytes, swf_stream_t stream) {
    if (!p || !in || !out || !p->conv.weight || !p->ln.gamma || !p->ln.beta) return fail(SWF_ERR_NULL, "patch_merge: NULL argument");
    if (B <= 0 || Cin <= 0 || Cout <= 0) return fail(SWF_ERR_BAD_SHAPE, "patch_merge: non-positive dims");
    const swf_patch_params* pp[2] = {p, nullptr};
    const float* ins[2] = {in, nullptr};
    float* outs[2] = {out, nullptr};
    return patch_merge_impl(pp, 1, ins, outs, B, H, W, Cin, Cout, merge_h, merge_w, win_h, win_w, workspace, workspace_bytes,
                            as_stream(stream));
}

int swf_patch_unmerge_fwd(const swf_patch_params* p, const float* in, const float* skip, float* out, int32_t B, int32_t Hp,
                          int32_t Wp, int32_t Hm, int32_t Wm, int32_t Cin, int32_t Cout, int32_t merge_h, int32_t merge_w,
                          int32_t Hout, int32_t Wout, void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    if (!p || !in || !out || !p->conv.weight || !p->ln.gamma || !p->ln.beta) return fail(SWF_ERR_NULL, "patch_unmerge: NULL argument");
    if (B <= 0 || Cin <= 0 || Cout <= 0 || merge_h <= 0 || merge_w <= 0) return fail(SWF_ERR_BAD_SHAPE, "patch_unmerge: non-positive dims");
    const swf_patch_params* pp[2] = {p, nullptr};
    const float* ins[2] = {in, nullptr};
    const float* sk[2] = {skip, nullptr};
    float* outs[2] = {out, nullptr};
    return patch_unmerge_impl(pp, 1, ins, skip ? sk : nullptr, outs, B, Hp, Wp, Hm, Wm, Cin, Cout, merge_h, merge_w, Hout, Wout,
                              workspace, workspace_bytes, as_stream(stream));
}

int swf_final_head_fwd(const swf_head_params* p, const float* x, const float* y, float* out, int32_t B, int32_t H, int32_t W,
                       int32_t ksize, void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    if (!p || !x || !y || !out) return fail(SWF_ERR_NULL, "final_head: NULL argument");
    if (B <= 0 || H <= 0 || W <= 0) return fail(SWF_ERR_BAD_SHAPE, "final_head: empty tensor");
    if (ksize <= 0 || ksize % 2 == 0) return fail(SWF_ERR_UNSUPPORTED, "final_head: even kernel size %d", ksize);
    if (ksize / 2 >= H || ksize / 2 >= W) return fail(SWF_ERR_PAD, "final_head: reflect pad %d >= map %dx%d", ksize / 2, H, W);
    Carver ws(workspace, workspace_bytes);
    float* tmp = ws.floats((int64_t)B * H * W * 2);
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "final_head workspace too small (need %zu B)", ws.used);
    return launch_head(x, y, tmp, out, *p, B, H, W, ksize, as_stream(stream));
}

int swf_linear_fwd(const swf_linear* lin, const float* in, const float* residual, float* out, int64_t tokens, int32_t n_in,
                   int32_t n_out, int32_t act, swf_stream_t stream) {
    if (!lin || !lin->weight || !in || !out) return fail(SWF_ERR_NULL, "linear: NULL argument");
    if (tokens <= 0 || tokens > INT32_MAX || n_in <= 0 || n_out <= 0) return fail(SWF_ERR_BAD_SHAPE, "linear: bad sizes");
    if (act != 0 && act != 1) return fail(SWF_ERR_UNSUPPORTED, "linear: activation %d", act);
    GemmBatch gb{};
    gb.p[0] = GemmProb{in, lin->weight, lin->bias, residual, out};
    return launch_gemm_f32(gb, 1, (int)tokens, n_out, n_in, n_in, n_out, act, as_stream(stream));
}

size_t swf_linear_workspace_bytes(int32_t precision, int64_t tokens, int32_t n_in, int32_t n_out) {
    if (precision != SWF_PREC_FAST || tokens <= 0 || n_in <= 0 || n_out <= 0) return 0;
    return carve_bytes({splitk_need(n_in, tokens * n_out)});
}

int swf_linear_fwd_prec(const swf_linear* lin, int32_t precision, const float* in, const float* residual, float* out, int64_t tokens,
                        int32_t n_in, int32_t n_out, int32_t act, void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    if (precision == SWF_PREC_FP32) return swf_linear_fwd(lin, in, residual, out, tokens, n_in, n_out, act, stream);
    if (precision != SWF_PREC_FAST) return fail(SWF_ERR_BAD_SHAPE, "linear: unknown precision %d", precision);
    if (!lin || !lin->weight || !in || !out) return fail(SWF_ERR_NULL, "linear: NULL argument");
    if (tokens <= 0 || tokens > INT32_MAX || n_in <= 0 || n_out <= 0) return fail(SWF_ERR_BAD_SHAPE, "linear: bad sizes");
    if (act != 0 && act != 1) return fail(SWF_ERR_UNSUPPORTED, "linear: activation %d", act);
    Carver ws(workspace, workspace_bytes);
    GemmBatch gb{};
    gb.scratch_floats = splitk_need(n_in, tokens * n_out);
    gb.scratch = ws.floats(gb.scratch_floats);
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "linear: workspace too small (need %zu B)", ws.used);
    gb.p[0] = GemmProb{in, lin->weight, lin->bias, residual, out};
    return launch_gemm_bf16x3(gb, 1, (int)tokens, n_out, n_in, n_in, n_out, act, as_stream(stream));
}

int swf_layernorm_fwd(const swf_norm* ln, const float* in, float* out, int64_t tokens, int32_t C, int32_t elu, swf_stream_t stream) {
    if (!ln || !ln->gamma || !ln->beta || !in || !out) return fail(SWF_ERR_NULL, "layernorm: NULL argument");
    if (tokens <= 0 || C <= 0) return fail(SWF_ERR_BAD_SHAPE, "layernorm: bad sizes");
    LnBatch lb{};
    lb.p[0] = LnProb{in, out, ln->gamma, ln->beta};
    return launch_layernorm(lb, 1, tokens, C, elu ? 1 : 0, as_stream(stream));
}

int swf_reflect_pad_fwd(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t pad_h, int32_t pad_w,
                        swf_stream_t stream) {
    if (!in || !out) return fail(SWF_ERR_NULL, "reflect_pad: NULL tensor");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || pad_h < 0 || pad_w < 0) return fail(SWF_ERR_BAD_SHAPE, "reflect_pad: bad sizes");
    if (pad_h >= H || pad_w >= W)
        return fail(SWF_ERR_PAD, "Padding size should be less than the corresponding input dimension: pad (%d,%d) on a %dx%d map", pad_h, pad_w, H, W);
    return launch_reflect_pad(in, out, B, H, W, C, pad_h, pad_w, as_stream(stream));
}

int swf_crop_fwd(const float* in, float* out, int32_t B, int32_t Hp, int32_t Wp, int32_t H, int32_t W, int32_t C, swf_stream_t stream) {
    if (!in || !out) return fail(SWF_ERR_NULL, "crop: NULL tensor");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || H > Hp || W > Wp) return fail(SWF_ERR_BAD_SHAPE, "crop: bad sizes");
    PtrPair pp{};
    pp.in[0] = in; pp.out[0] = out;
    return launch_crop(pp, 1, B, Hp, Wp, H, W, C, as_stream(stream));
}

int swf_nchw_to_nhwc(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W, swf_stream_t stream) {
    if (!in || !out) return fail(SWF_ERR_NULL, "layout: NULL tensor");
    return launch_nchw_to_nhwc(in, out, B, C, H, W, as_stream(stream));
}
int swf_nhwc_to_nchw(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W, swf_stream_t stream) {
    if (!in || !out) return fail(SWF_ERR_NULL, "layout: NULL tensor");
    return launch_nhwc_to_nchw(in, out, B, C, H, W, as_stream(stream));
}

// ---- model ------------------------------------------------------------------------------------
int32_t swf_model_param_count(const swf_model_desc* desc) {
    if (check_model_desc(desc) != SWF_OK) return -1;
    return (int32_t)get_layout(desc)->entries.size();
}
int64_t swf_model_arena_elems(const swf_model_desc* desc) {
    if (check_model_desc(desc) != SWF_OK) return -1;
    return get_layout(desc)->total;
}
int swf_model_param_info(const swf_model_desc* desc, int32_t index, char* name_buf, size_t name_buf_len, int64_t* offset_elems,
                         int64_t* numel) {
    SWF_TRY(check_model_desc(desc));
    const ModelLayout* l = get_layout(desc);
    if (index < 0 || index >= (int32_t)l->entries.size()) return fail(SWF_ERR_BAD_SHAPE, "param index %d out of range", index);
    const ParamEntry& e = l->entries[index];
    if (name_buf && name_buf_len) {
        if (e.name.size() + 1 > name_buf_len) return fail(SWF_ERR_WORKSPACE, "name buffer too small (%zu needed)", e.name.size() + 1);
        std::memcpy(name_buf, e.name.c_str(), e.name.size() + 1);
    }
    if (offset_elems) *offset_elems = e.offset;
    if (numel) *numel = e.numel;
    return SWF_OK;
}

// workspace layout: per level two activation maps (x, y), two full-resolution decoder outputs,
// then one scratch region shared by every unit.
static size_t model_scratch_bytes(const swf_model_desc* d, int B, const LevelShape* ls) {
    size_t scratch = 0;
    for (int s = 0; s < d->levels; ++s) {
        swf_block_desc be = level_block_desc(d, s, true), bd = level_block_desc(d, s, false);
        scratch = std::max(scratch, std::max(block_generic_ws(&be, 2, B, ls[s].Ho, ls[s].Wo), block_generic_ws(&bd, 2, B, ls[s].Ho, ls[s].Wo)));
        scratch = std::max(scratch, std::max(window_block_workspace_bytes(be, B, ls[s].Ho, ls[s].Wo), window_block_workspace_bytes(bd, B, ls[s].Ho, ls[s].Wo)));
        scratch = std::max(scratch, patch_ws(2, B, ls[s].Hin, ls[s].Win, d->in_dims[s], d->out_dims[s], d->merge_h, d->merge_w, d->win_h, d->win_w, 1));
        scratch = std::max(scratch, patch_ws(2, B, ls[s].Ho, ls[s].Wo, d->out_dims[s], d->in_dims[s], d->merge_h, d->merge_w, d->win_h, d->win_w, 0));
    }
    scratch = std::max(scratch, carve_bytes({(int64_t)B * ls[0].Hin * ls[0].Win * 2}));
    return scratch;
}

size_t swf_model_workspace_bytes(const swf_model_desc* desc, int32_t B, int32_t H, int32_t W) {
    if (check_model_desc(desc) != SWF_OK || B <= 0) return 0;
    LevelShape ls[SWF_MAX_LEVELS];
    if (model_shapes(desc, H, W, ls) != SWF_OK) return 0;
    size_t total = 0;
    for (int s = 0; s < desc->levels; ++s) total += 2 * carve_bytes({(int64_t)B * ls[s].Ho * ls[s].Wo * desc->out_dims[s]});
    total += 2 * carve_bytes({(int64_t)B * H * W * desc->in_dims[0]});
    return total + model_scratch_bytes(desc, B, ls) + 256;
}

size_t swf_model_packed_bytes(const swf_model_desc* desc) {
    if (check_model_desc(desc) != SWF_OK) return 0;
    return packed_plan(desc).total;
}

int swf_model_pack_weights(const swf_model_desc* desc, const float* arena, void* packed, size_t packed_bytes, swf_stream_t stream_) {
    SWF_TRY(check_model_desc(desc));
    if (!arena) return fail(SWF_ERR_NULL, "model_pack_weights: NULL arena");
    const PackedPlan plan = packed_plan(desc);
    if (plan.total == 0) return SWF_OK;
    if (!packed || packed_bytes < plan.total) return fail(SWF_ERR_WORKSPACE, "packed buffer too small (need %zu B)", plan.total);
    const ModelLayout* L = get_layout(desc);
    hipStream_t stream = as_stream(stream_);
    char* base = static_cast<char*>(packed);
    for (int pass = 0; pass < 2; ++pass)
        for (int k = 0; k < desc->levels; ++k) {
            const bool enc = pass == 0;
            if (!(enc ? plan.enc_on[k] : plan.dec_on[k])) continue;
            swf_block_desc bd = level_block_desc(desc, enc ? k : desc->levels - 1 - k, enc);
            bd.precision = SWF_PREC_FAST;
            const size_t pb = block_packed_bytes(bd);
            const bool fused = window_block_packed_bytes(bd) > 0;
            char* dst = base + (enc ? plan.enc[k] : plan.dec[k]);
            for (int i = 0; i < 4; ++i) {
                const swf_block_stream_params px = make_stream_params(arena, enc ? L->enc_blk[k][i][0] : L->dec_blk[k][i][0]);
                const swf_block_stream_params py = make_stream_params(arena, enc ? L->enc_blk[k][i][1] : L->dec_blk[k][i][1]);
                if (fused) {
                    SWF_TRY(pack_window_block(bd, px, py, dst + (size_t)(2 * i) * pb, dst + (size_t)(2 * i + 1) * pb, stream));
                } else {
                    SWF_TRY(pack_deep_block(bd, px, dst + (size_t)(2 * i) * pb, stream));
                    SWF_TRY(pack_deep_block(bd, py, dst + (size_t)(2 * i + 1) * pb, stream));
                }
            }
        }
    for (int k = 0; k < desc->levels; ++k) {   // patch layers of the register-resident kernel
        const int lvl = desc->levels - 1 - k;
        for (int st = 0; st < 2; ++st) {
            if (plan.penc_b[k]) {
                const PatchOff& o = L->enc_patch[k][st];
                if (patch_rr_packed_bytes(0, desc->in_dims[k], desc->out_dims[k], desc->merge_h, desc->merge_w))
                    SWF_TRY(pack_patch_rr(0, desc->in_dims[k], desc->out_dims[k], desc->merge_h, desc->merge_w, arena + o.w, arena + o.b, arena + o.g,
                                          arena + o.bt, base + plan.penc[k] + st * plan.penc_b[k], stream));
                else
                    SWF_TRY(pack_deep_patch(0, desc->in_dims[k], desc->out_dims[k], desc->merge_h, desc->merge_w, arena + o.w,
                                            base + plan.penc[k] + st * plan.penc_b[k], stream));
            }
            if (plan.pdec_b[k]) {
                const PatchOff& o = L->dec_patch[k][st];
                if (patch_rr_packed_bytes(1, desc->out_dims[lvl], desc->in_dims[lvl], desc->merge_h, desc->merge_w))
                    SWF_TRY(pack_patch_rr(1, desc->out_dims[lvl], desc->in_dims[lvl], desc->merge_h, desc->merge_w, arena + o.w, arena + o.b,
                                          arena + o.g, arena + o.bt, base + plan.pdec[k] + st * plan.pdec_b[k], stream));
                else
                    SWF_TRY(pack_deep_patch(1, desc->out_dims[lvl], desc->in_dims[lvl], desc->merge_h, desc->merge_w, arena + o.w,
                                            base + plan.pdec[k] + st * plan.pdec_b[k], stream));
            }
        }
    }
    return SWF_OK;
}

static int model_forward_impl(const swf_model_desc* desc, const float* arena, const char* packed, const float* ir, const float* vis, float* out,
                              int32_t B, int32_t H, int32_t W, void* workspace, size_t workspace_bytes, swf_stream_t stream_,
                              int32_t* equal_flags = nullptr, hipEvent_t* marks = nullptr) {
    // marks (swf_model_forward_profiled): 4 * levels + 2 events, recorded on the stream at the start and after every segment
    SWF_TRY(check_model_desc(desc));
    const PackedPlan plan = packed_plan(desc);
    if (!arena || !ir || !vis || !out) return fail(SWF_ERR_NULL, "model_forward: NULL tensor");
    if (B <= 0) return fail(SWF_ERR_BAD_SHAPE, "model_forward: empty batch");
    LevelShape ls[SWF_MAX_LEVELS];
    SWF_TRY(model_shapes(desc, H, W, ls));
    const ModelLayout* L = get_layout(desc);
    hipStream_t stream = as_stream(stream_);
    const int n = desc->levels;
    Carver ws(workspace, workspace_bytes);
    float* act[SWF_MAX_LEVELS][2];
    for (int s = 0; s < n; ++s)
        for (int t = 0; t < 2; ++t) act[s][t] = ws.floats((int64_t)B * ls[s].Ho * ls[s].Wo * desc->out_dims[s]);
    float* full[2] = {ws.floats((int64_t)B * H * W * desc->in_dims[0]), ws.floats((int64_t)B * H * W * desc->in_dims[0])};
    const size_t scratch_bytes = model_scratch_bytes(desc, B, ls);
    size_t scratch_off = align_up(ws.used, 256);
    if (!workspace || scratch_off + scratch_bytes > workspace_bytes)
        return fail(SWF_ERR_WORKSPACE, "model workspace too small: have %zu B, need %zu B", workspace_bytes, scratch_off + scratch_bytes);
    void* scratch = static_cast<char*>(workspace) + scratch_off;

    auto patch_params = [&](const PatchOff& o) { return swf_patch_params{{arena + o.w, arena + o.b}, {arena + o.g, arena + o.bt}}; };

    int nmark = 0;
    auto mark = [&]() -> int {
        if (marks && hipEventRecord(marks[nmark++], stream) != hipSuccess) return fail(SWF_ERR_HIP, "model_forward: hipEventRecord failed");
        return SWF_OK;
    };
    SWF_TRY(mark());
    // encoder (a013:215-220)
    const float* cur[2] = {ir, vis};
    bool ln1_carry = false;   // deep levels: the LN1 planes of the next block to run are in place (deep_ln1_planes)
    swf_block_stream_params dpx0 = make_stream_params(arena, L->dec_blk[0][0][0]), dpy0 = make_stream_params(arena, L->dec_blk[0][0][1]);
    const swf_block_stream_params* dec_first[2] = {&dpx0, &dpy0};   // first block of the decoder stage that follows the deepest encoder stage
    for (int s = 0; s < n; ++s) {
        swf_patch_params pm[2] = {patch_params(L->enc_patch[s][0]), patch_params(L->enc_patch[s][1])};
        const swf_patch_params* pmp[2] = {&pm[0], &pm[1]};
        const void* prr_enc[2] = {packed ? packed + plan.penc[s] : nullptr, packed ? packed + plan.penc[s] + plan.penc_b[s] : nullptr};
        swf_block_stream_params px[4], py[4];
        for (int i = 0; i < 4; ++i) { px[i] = make_stream_params(arena, L->enc_blk[s][i][0]); py[i] = make_stream_params(arena, L->enc_blk[s][i][1]); }
        swf_block_desc bd = level_block_desc(desc, s, true);
        const swf_block_stream_params* first_blk[2] = {&px[0], &py[0]};
        const bool deep_stage = packed && desc->precision == SWF_PREC_FAST && window_block_packed_bytes(bd) == 0 && deep_block_supported(bd);
        ln1_carry = false;
        SWF_TRY(patch_merge_impl(pmp, 2, cur, act[s], B, ls[s].Hin, ls[s].Win, desc->in_dims[s], desc->out_dims[s], desc->merge_h,
                                 desc->merge_w, desc->win_h, desc->win_w, scratch, scratch_bytes, stream, desc->precision == SWF_PREC_FAST,
                                 (packed && plan.penc_b[s]) ? prr_enc : nullptr, deep_stage ? first_blk : nullptr, deep_stage ? &ln1_carry : nullptr));
        SWF_TRY(mark());
        // the first block of the next fused-kernel stage is warmed by this stage's last block
        const char* after = nullptr;
        size_t after_pb = 0;
        if (packed && s + 1 < n) {
            swf_block_desc nb = level_block_desc(desc, s + 1, true);
            if (window_block_packed_bytes(nb) && plan.enc_on[s + 1]) { after = packed + plan.enc[s + 1]; after_pb = window_block_packed_bytes(nb); }
        }
        // the deepest stage hands the LN1 planes of the decoder's first block (same level, same map, same workspace) to it
        const bool chain = deep_stage && s == n - 1;
        SWF_TRY(block_pair4_impl(&bd, px, py, act[s][0], act[s][1], act[s][0], act[s][1], B, ls[s].Ho, ls[s].Wo, scratch, scratch_bytes, stream,
                                 (packed && plan.enc_on[s]) ? packed + plan.enc[s] : nullptr, after, after_pb,
                                 equal_flags ? equal_flags + 2 * s : nullptr, chain ? dec_first : nullptr, deep_stage ? &ln1_carry : nullptr));
        if (!chain) ln1_carry = false;
        SWF_TRY(mark());
        cur[0] = act[s][0]; cur[1] = act[s][1];
    }
    // decoder (a013:221-227): the skip add of stage j+1 is folded into stage j's unmerge epilogue,
    // written in place over the encoder activation of the level below.
    for (int j = 0; j < n; ++j) {
        const int lvl = n - 1 - j;
        swf_block_stream_params px[4], py[4];
        for (int i = 0; i < 4; ++i) { px[i] = make_stream_params(arena, L->dec_blk[j][i][0]); py[i] = make_stream_params(arena, L->dec_blk[j][i][1]); }
        swf_block_desc bd = level_block_desc(desc, lvl, false);
        const char* after = nullptr;
        size_t after_pb = 0;
        if (packed && j + 1 < n) {
            swf_block_desc nb = level_block_desc(desc, n - 2 - j, false);
            if (window_block_packed_bytes(nb) && plan.dec_on[j + 1]) { after = packed + plan.dec[j + 1]; after_pb = window_block_packed_bytes(nb); }
        }
        bool ln1_in = ln1_carry;   // planes left by the deepest encoder stage (j == 0) or by the unmerge layer of the stage before
        SWF_TRY(block_pair4_impl(&bd, px, py, act[lvl][0], act[lvl][1], act[lvl][0], act[lvl][1], B, ls[lvl].Ho, ls[lvl].Wo, scratch, scratch_bytes, stream,
                                 (packed && plan.dec_on[j]) ? packed + plan.dec[j] : nullptr, after, after_pb,
                                 equal_flags ? equal_flags + 2 * (n + j) : nullptr, nullptr, &ln1_in));
        SWF_TRY(mark());
        // a deep-level stage cannot warm its successor from inside a block kernel: its patch layer does it (or one small launch)
        const bool warm_next = after && window_block_packed_bytes(bd) == 0;
        bool warmed = false;
        // the next decoder stage, when it is a deep-level one, finds its first block's LN1 planes written by this stage's unmerge layer
        swf_block_stream_params npx0{}, npy0{};
        bool next_deep = false;
        if (packed && desc->precision == SWF_PREC_FAST && j + 1 < n) {
            swf_block_desc nb = level_block_desc(desc, n - 2 - j, false);
            nb.precision = SWF_PREC_FAST;
            next_deep = window_block_packed_bytes(nb) == 0 && deep_block_supported(nb);
            if (next_deep) { npx0 = make_stream_params(arena, L->dec_blk[j + 1][0][0]); npy0 = make_stream_params(arena, L->dec_blk[j + 1][0][1]); }
        }
        const swf_block_stream_params* next_first[2] = {&npx0, &npy0};
        ln1_carry = false;
        swf_patch_params pm[2] = {patch_params(L->dec_patch[j][0]), patch_params(L->dec_patch[j][1])};
        const swf_patch_params* pmp[2] = {&pm[0], &pm[1]};
        const float* ins[2] = {act[lvl][0], act[lvl][1]};
        const float* skip[2] = {lvl > 0 ? act[lvl - 1][0] : nullptr, lvl > 0 ? act[lvl - 1][1] : nullptr};
        float* outs[2] = {lvl > 0 ? act[lvl - 1][0] : full[0], lvl > 0 ? act[lvl - 1][1] : full[1]};
        const void* prr_dec[2] = {packed ? packed + plan.pdec[j] : nullptr, packed ? packed + plan.pdec[j] + plan.pdec_b[j] : nullptr};
        SWF_TRY(patch_unmerge_impl(pmp, 2, ins, lvl > 0 ? skip : nullptr, outs, B, ls[lvl].Ho, ls[lvl].Wo, ls[lvl].Hm, ls[lvl].Wm,
                                   desc->out_dims[lvl], desc->in_dims[lvl], desc->merge_h, desc->merge_w, ls[lvl].Hin, ls[lvl].Win,
                                   scratch, scratch_bytes, stream, desc->precision == SWF_PREC_FAST, (packed && plan.pdec_b[j]) ? prr_dec : nullptr,
                                   warm_next ? after : nullptr, warm_next ? after_pb : 0, &warmed, next_deep ? next_first : nullptr,
                                   next_deep ? &ln1_carry : nullptr));
        if (warm_next && !warmed) SWF_TRY(launch_l2_warm(after, 2 * after_pb, stream));
        SWF_TRY(mark());
    }
    swf_head_params hp{arena + L->h_c1w, arena + L->h_c1b, arena + L->h_g, arena + L->h_b, arena + L->h_m, arena + L->h_v,
                       arena + L->h_c2w, arena + L->h_c2b};
    float* tmp = static_cast<float*>(scratch);
    SWF_TRY(launch_head(full[0], full[1], tmp, out, hp, B, H, W, desc->head_ksize, stream));
    return mark();
}

int swf_model_forward(const swf_model_desc* desc, const float* arena, const float* ir, const float* vis, float* out,
                      int32_t B, int32_t H, int32_t W, void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    return model_forward_impl(desc, arena, nullptr, ir, vis, out, B, H, W, workspace, workspace_bytes, stream);
}

int swf_model_forward_packed(const swf_model_desc* desc, const float* arena, const void* packed, const float* ir, const float* vis,
                             float* out, int32_t B, int32_t H, int32_t W, void* workspace, size_t workspace_bytes, swf_stream_t stream) {
    return model_forward_impl(desc, arena, static_cast<const char*>(packed), ir, vis, out, B, H, W, workspace, workspace_bytes, stream);
}

int swf_model_forward_checked(const swf_model_desc* desc, const float* arena, const void* packed, const float* ir, const float* vis,
                              float* out, int32_t B, int32_t H, int32_t W, void* workspace, size_t workspace_bytes,
                              int32_t* cross_equal_flags, swf_stream_t stream) {
    if (!cross_equal_flags) return fail(SWF_ERR_NULL, "model_forward_checked: NULL flags");
    if (check_model_desc(desc) != SWF_OK) return SWF_ERR_BAD_SHAPE;
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cross_equal_flags), 1, (size_t)4 * desc->levels, as_stream(stream));
    if (e != hipSuccess) return fail(SWF_ERR_HIP, "model_forward_checked: memset: %s", hipGetErrorString(e));
    return model_forward_impl(desc, arena, static_cast<const char*>(packed), ir, vis, out, B, H, W, workspace, workspace_bytes, stream,
                              cross_equal_flags);
}

int swf_model_forward_profiled(const swf_model_desc* desc, const float* arena, const void* packed, const float* ir, const float* vis,
                               float* out, int32_t B, int32_t H, int32_t W, void* workspace, size_t workspace_bytes, float* seg_ms,
                               int32_t seg_count, swf_stream_t stream) {
    if (!seg_ms) return fail(SWF_ERR_NULL, "model_forward_profiled: NULL seg_ms");
    if (check_model_desc(desc) != SWF_OK) return SWF_ERR_BAD_SHAPE;
    const int nseg = 4 * desc->levels + 1;
    if (seg_count < nseg) return fail(SWF_ERR_BAD_SHAPE, "model_forward_profiled: seg_ms holds %d values, need %d", seg_count, nseg);
    hipEvent_t ev[4 * SWF_MAX_LEVELS + 2];
    int made = 0, st = SWF_OK;
    for (; made < nseg + 1; ++made)
        if (hipEventCreate(&ev[made]) != hipSuccess) { st = fail(SWF_ERR_HIP, "model_forward_profiled: hipEventCreate failed"); break; }
    if (st == SWF_OK)
        st = model_forward_impl(desc, arena, static_cast<const char*>(packed), ir, vis, out, B, H, W, workspace, workspace_bytes, stream, nullptr, ev);
    if (st == SWF_OK && hipStreamSynchronize(as_stream(stream)) != hipSuccess) st = fail(SWF_ERR_HIP, "model_forward_profiled: synchronize failed");
    for (int i = 0; st == SWF_OK && i < nseg; ++i)
        if (hipEventElapsedTime(&seg_ms[i], ev[i], ev[i + 1]) != hipSuccess) st = fail(SWF_ERR_HIP, "model_forward_profiled: hipEventElapsedTime failed");
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    return st;
}

int swf_tensors_equal(const float* a, const float* b, int64_t count, int32_t* flag, swf_stream_t stream) {
    if (!flag) return fail(SWF_ERR_NULL, "tensors_equal: NULL flag");
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flag), 1, 1, as_stream(stream));
    if (e != hipSuccess) return fail(SWF_ERR_HIP, "tensors_equal: memset: %s", hipGetErrorString(e));
    return launch_all_equal(a, b, count, flag, as_stream(stream));
}

}  // extern "C"
