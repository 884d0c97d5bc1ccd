// Patch layers of the deep levels in one launch (fast tier; kernels_patchrr.hip covers K, N <= 192):
//   encoder  PatchMerging   gather 2x2 -> conv (K = 4 Cin -> N = Cout) -> LayerNorm(N) -> ELU                       (a010:61-75)
//   decoder  anti-merging   conv (K = Cin -> N = 4 Cout) -> LayerNorm(N) -> ELU -> depth-to-space scatter (+ skip)   (a011:107-117)
// for (K, N) = (384, 192), (768, 384) | (384, 768), (192, 384).  Replaces gather + split-K GEMM + reduce + LayerNorm launches.
//
// One workgroup = 32 MT token rows x ALL N output columns (LayerNorm needs whole rows), NW waves:
//   * the gathered token rows (both reflect pads / the crop are index arithmetic) are split to bf16 hi / lo into an LDS image,
//     K <= 384 columns at a time;
//   * out^T[column][token] on v_mfma_f32_32x32x16_bf16 (bf16x3): a wave owns whole 32-column tiles (+ one half tile when the tile
//     count does not divide), its weight fragments stream from L2 in fragment-major order (one contiguous 1-KB load per fragment,
//     pack_deep_patch) through a register ring, the token fragments come from the LDS image;
//   * the out tile goes through LDS (laid over the token image) and leaves as rows: bias, LayerNorm (fp32, shuffles within the
//     row's threads), ELU, then rows out (encoder) or the depth-to-space scatter with the skip add (decoder).
#include "kernels_deeppatch.h"

#include <cstdlib>

#include "kernels_qkvattn.h"
#include "win_frag.h"

namespace swf {
namespace {

using namespace wf;
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct DpArgs {
    const float* in[2]; float* out[2]; const float* skip[2];
    const bf16* w_hi[2]; const bf16* w_lo[2];
    const float* bias[2]; const float* gamma[2]; const float* beta[2];
    int B, H, W, Hm, Wm, Ho, Wo, M;
    // Q/K/V mode (DEC == 2): LayerNorm planes in, fp16 operands of the attention core out
    const bf16* xh[2]; const bf16* xl[2]; const float* qb[2][3]; f16* qo[2][3]; float qscale; int cross;
    // encoder, whole-row form: LayerNorm of the finished rows with these parameters (the next block's LN1) as split-bf16 planes [M][N], or nullptr
    const float* ln_g[2]; const float* ln_b[2]; bf16* ln_hi[2]; bf16* ln_lo[2];
    // decoder, whole-row form: packed weights of the block that runs next, touched at the end (per-XCD L2 warm-up, kernels_window.hip), or nullptr
    const char* warm[2]; int warm_bytes;
};

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < n ? i : 2 * n - 2 - i; }   // bottom / right only
__device__ __forceinline__ float elu_fast(float v) { return v > 0.f ? v : __builtin_amdgcn_exp2f(v * kLog2e) - 1.0f; }

#ifdef DP_PROBE   // tools/dp_probe.hip: wall-clock stamps (10 ns) of workgroup (DP_PROBE, 0), thread 0
__device__ unsigned long long dp_probe[16];
#define DP_STAMP(i) do { if (blockIdx.x == DP_PROBE && blockIdx.y == 0 && threadIdx.x == 0) dp_probe[i] = wall_clock64(); } while (0)
#else
#define DP_STAMP(i) do { } while (0)
#endif

template <int K, int N, int MT, int NW, int DEC, int NTOT>
__global__ __launch_bounds__(64 * NW) void deep_patch_kernel(DpArgs a) {
    // NTOT == N: the workgroup owns whole conv output rows and finishes them (LayerNorm, ELU, store / scatter).  NTOT > N ("raw"):
    // it owns columns [N z, N z + N) of NTOT (z = blockIdx.z) and writes conv + bias to a.out as rows [M][NTOT]; LayerNorm runs as
    // a second launch.  The two layers next to the deepest level take this form: 1 024 tokens per stream are 16-32 whole-row
    // workgroups, each MFMA-bound for 11 us on an eighth of the chip.
    // DEC == 2 (Q/K/V projections of a level-4 block, raw form): the rows are the LayerNorm planes [M][K] (already split bf16), the
    // NTOT = 3 K weight rows are Wq | Wk | Wv stacked, and the epilogue writes (acc + bias) [* d^-0.5 log2 e for Q] as fp16 — the
    // operand formats of attn_proj_kernel (SP_EPI_QKV16 of the GEMM this replaces).  K and V of a cross block read the other stream.
    // DEC == 3 (output projection of a deep block): rows = the attention output planes [M][K], weights = Wproj fragment-major,
    // epilogue = + bias + residual rows -> out [M][NTOT] (columns [N z, N z + N) per workgroup).
    constexpr bool IS_DEC = DEC == 1, IS_QKV = DEC == 2, IS_PROJ = DEC == 3, RAW = NTOT != N || IS_PROJ;
    static_assert(!IS_QKV || (NTOT == 3 * K && K % N == 0 && K <= 384), "Q/K/V mode");
    static_assert(!IS_PROJ || (NTOT == K && K <= 384), "projection mode");
    constexpr int NT = 64 * NW, MR = 32 * MT, TPR = NT / MR;           // threads, token rows per workgroup, threads per row
    constexpr int KC = K > 384 ? 384 : K, NKC = K / KC, KS = KC / 16;   // columns staged at once, passes, k16 steps per pass
    constexpr int T = N / 32, NF = T / NW, R = T % NW;                  // 32-column tiles; whole tiles per wave; left-over tiles
    static_assert(K % KC == 0 && N % 32 == 0 && (R == 0 || (MT == 2 && 2 * R == NW)), "tile dealing");
    constexpr int NH = R ? 1 : 0, NFR = NF + NH;                        // + one half tile (one 32-token half); fragments per k16 step
    constexpr int NFRAG = KS * NFR;                                     // fragments a wave streams per pass
    constexpr int D = NFRAG % 8 == 0 ? 8 : 6;                           // ring depth
    static_assert(NFRAG % D == 0, "ring depth must divide the fragment count");
    constexpr int LDA = KC + 8, ORS = N + 4;                            // row strides: odd multiples of 16 B
    constexpr int CIN = IS_DEC ? K : K / 4, COUT = IS_DEC ? NTOT / 4 : NTOT;  // channels per input / output pixel (2x2 merging)
    constexpr size_t img_bytes = size_t(2) * MR * LDA * 2, out_bytes = size_t(MR) * ORS * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* a_hi = reinterpret_cast<bf16*>(smem);
    bf16* a_lo = a_hi + MR * LDA;
    float* otile = reinterpret_cast<float*>(smem);   // laid over the token image after the last k step
    (void)img_bytes; (void)out_bytes;

    const int tile = blockIdx.x, s = blockIdx.y, ct0 = RAW ? (int)blockIdx.z * (N / 32) : 0;   // first 32-column tile of this workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;

    // ---- the wave's weight fragment stream (buffer loads: lane offset in a VGPR, fragment offset in an SGPR) ----
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w_hi[s]), 0, NTOT * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w_lo[s]), 0, NTOT * K * 2, 0x00020000);
    const int loff = lane * 16;
    const int half_nt = T - R + (wave >> 1), half_tok = wave & 1;
    u32x4 rh[D], rl[D];
    // fragment f of pass kc: k16 step f / NFR of the pass, the wave's tile f % NFR
    auto frag_load = [&](int f, int kc) {
        const int ks = f / NFR, j = f % NFR;
        const int nt = j < NF ? wave + NW * j : T - R + (wave >> 1);
        const int off = ((ct0 + nt) * (K / 16) + kc * KS + ks) * 1024;
        rh[f % D] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rwh, loff, off, 0));
        rl[f % D] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rwl, loff, off, 0));
    };

    // ---- source rows of this workgroup: thread (row, sub) covers float4 columns sub, sub + TPR, ... of the staged K chunk ----
    const int row = tid / TPR, sub = tid % TPR;
    const int m = min(tile * MR + row, a.M - 1);   // rows past M are computed on a clamped copy and never stored
    int rb = 0, ry = 0, rx = 0;                     // batch index and position of the row's token in its (merged) map
    if (IS_DEC) { rx = m % a.Wm; const int t = m / a.Wm; ry = t % a.Hm; rb = t / a.Hm; }
    else if (!IS_QKV && !IS_PROJ) { rx = m % a.Wo; const int t = m / a.Wo; ry = t % a.Ho; rb = t / a.Ho; }
    const int which = IS_QKV ? (32 * ct0) / K : 0;   // 0 = Q, 1 = K, 2 = V
    constexpr int NV = KC / (4 * TPR);
    static_assert(KC % (4 * TPR) == 0, "row chunks must divide over the row's threads");
    auto stage = [&](int kc) {   // K chunk kc of the rows -> split-bf16 LDS image
        if constexpr (IS_QKV || IS_PROJ) {   // the planes are copied as they are: 16-byte chunks of 8 bf16
            constexpr int NQ = KC / (8 * TPR);
            const int ss = (IS_QKV && which != 0 && a.cross) ? 1 - s : s;
            const bf16* ph = a.xh[ss] + (size_t)m * K;
            const bf16* pl = a.xl[ss] + (size_t)m * K;
            u32x4 vh[NQ], vl[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                vh[i] = *reinterpret_cast<const u32x4*>(ph + 8 * (sub + TPR * i));
                vl[i] = *reinterpret_cast<const u32x4*>(pl + 8 * (sub + TPR * i));
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                *reinterpret_cast<u32x4*>(a_hi + row * LDA + 8 * (sub + TPR * i)) = vh[i];
                *reinterpret_cast<u32x4*>(a_lo + row * LDA + 8 * (sub + TPR * i)) = vl[i];
            }
            return;
        }
        float4 v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int col = kc * KC + 4 * (sub + TPR * i);   // column of the conv input row
            const float* src;
            if (IS_DEC) {
                src = a.in[s] + ((size_t)(rb * a.H + ry) * a.W + rx) * CIN + col;   // crop: only Hm x Wm of the H x W map is read
            } else {
                const int pq = col / CIN, c = col - pq * CIN, ph = pq >> 1, pw = pq & 1;   // 2x2 merging: (ph, pw, channel) order
                const int my = reflect_idx(ry, a.Hm), mx = reflect_idx(rx, a.Wm);            // window pad of the merged map
                const int iy = reflect_idx(2 * my + ph, a.H), ix = reflect_idx(2 * mx + pw, a.W);   // merge pad of the input
                src = a.in[s] + ((size_t)(rb * a.H + iy) * a.W + ix) * CIN + c;
            }
            v[i] = *reinterpret_cast<const float4*>(src);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * (sub + TPR * i);
            const float n[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            bf16x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) { h[j] = (bf16)n[j]; l[j] = (bf16)(n[j] - (float)h[j]); }
            *reinterpret_cast<bf16x4*>(a_hi + row * LDA + c) = h;
            *reinterpret_cast<bf16x4*>(a_lo + row * LDA + c) = l;
        }
    };

    // decoder: the skip rows this thread adds at the very end are requested now (cold encoder activations: ~2 us from HBM)
    constexpr int NO = N / (4 * TPR);
    static_assert(N % (4 * TPR) == 0, "output row chunks must divide over the row's threads");
    f32x4 sk[IS_DEC && !RAW ? NO : 1];
    if constexpr (IS_DEC && !RAW) {
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int c = 4 * (sub + TPR * i), p = c / COUT, cc = c - p * COUT;
            const int y = 2 * ry + (p >> 1), xo = 2 * rx + (p & 1);
            sk[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.skip[s] && y < a.Ho && xo < a.Wo) sk[i] = *reinterpret_cast<const f32x4*>(a.skip[s] + ((size_t)(rb * a.Ho + y) * a.Wo + xo) * COUT + cc);
        }
    }
    DP_STAMP(0);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc[MT * NF + NH];
#pragma unroll
    for (int i = 0; i < MT * NF + NH; ++i) acc[i] = zero16;

#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        if (kc > 0) __syncthreads();   // every wave has read the previous chunk's image
        // the ring's first fragments go out ahead of the row loads: both fly under the split arithmetic
        if (kc == 0) {
#pragma unroll
            for (int f = 0; f < D; ++f) frag_load(f, 0);
        }
        stage(kc);
        DP_STAMP(1 + 3 * kc);
        __syncthreads();
        DP_STAMP(2 + 3 * kc);
        u32x4 bh[MT], bl[MT], xh, xl;
        {
            const int ko = 8 * hf;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                bh[t] = *reinterpret_cast<const u32x4*>(a_hi + (32 * t + r) * LDA + ko);
                bl[t] = *reinterpret_cast<const u32x4*>(a_lo + (32 * t + r) * LDA + ko);
            }
            if constexpr (NH) {
                xh = *reinterpret_cast<const u32x4*>(a_hi + (32 * half_tok + r) * LDA + ko);
                xl = *reinterpret_cast<const u32x4*>(a_lo + (32 * half_tok + r) * LDA + ko);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 wh[NFR], wl[NFR];
#pragma unroll
            for (int j = 0; j < NFR; ++j) {
                const int f = ks * NFR + j;
                wh[j] = rh[f % D]; wl[j] = rl[f % D];
                if (f + D < NFRAG) frag_load(f + D, kc);
                else if (kc + 1 < NKC) frag_load(f + D - NFRAG, kc + 1);
            }
            u32x4 ch[MT], cl[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) { ch[t] = bh[t]; cl[t] = bl[t]; }
            const u32x4 cxh = xh, cxl = xl;
            if (ks + 1 < KS) {   // token fragments one k16 step ahead
                const int ko = 16 * (ks + 1) + 8 * hf;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    bh[t] = *reinterpret_cast<const u32x4*>(a_hi + (32 * t + r) * LDA + ko);
                    bl[t] = *reinterpret_cast<const u32x4*>(a_lo + (32 * t + r) * LDA + ko);
                }
                if constexpr (NH) {
                    xh = *reinterpret_cast<const u32x4*>(a_hi + (32 * half_tok + r) * LDA + ko);
                    xl = *reinterpret_cast<const u32x4*>(a_lo + (32 * half_tok + r) * LDA + ko);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch D fragments ahead (hipcc otherwise sinks it next to its use)
#pragma unroll
            for (int j = 0; j < NFR; ++j) {
                if (j < NF) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[MT * j + t] = mma3(wh[j], wl[j], ch[t], cl[t], acc[MT * j + t]);
                } else {
                    if constexpr (NH) acc[MT * NF] = mma3(wh[j], wl[j], cxh, cxl, acc[MT * NF]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        DP_STAMP(3 + 3 * kc);
    }
    __syncthreads();   // the token image is dead: the out tile takes its place
    DP_STAMP(8);

    // ---- out tile -> LDS rows: register 4g+j of (tile nt, token half t) is column 32 nt + 8g + 4 hf + j of token 32 t + r ----
#pragma unroll
    for (int i = 0; i < MT * NF + NH; ++i) {
        const int nt = i < MT * NF ? wave + NW * (i / MT) : half_nt;
        const int t = i < MT * NF ? (i % MT) : half_tok;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(otile + (32 * t + r) * ORS + 32 * nt + 8 * g + 4 * hf) =
                f32x4{acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
    }
    __syncthreads();
    DP_STAMP(9);

    // ---- rows: + bias, LayerNorm over the N conv outputs, ELU, store / scatter.  TPR threads per row; every global operand of
    //      the row is requested before the arithmetic ----
    const bool live = tile * MR + row < a.M;
    f32x4 v[NO], gm[RAW ? 1 : NO], bt[RAW ? 1 : NO];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        const int c = 4 * (sub + TPR * i);
        v[i] = *reinterpret_cast<const f32x4*>(otile + row * ORS + c);
        if constexpr (IS_QKV) {
            if (a.qb[s][which]) v[i] += *reinterpret_cast<const f32x4*>(a.qb[s][which] + 32 * ct0 - which * K + c);
        } else {
            if (a.bias[s]) v[i] += *reinterpret_cast<const f32x4*>(a.bias[s] + 32 * ct0 + c);
        }
        if constexpr (!RAW) {
            gm[i] = *reinterpret_cast<const f32x4*>(a.gamma[s] + c);
            bt[i] = *reinterpret_cast<const f32x4*>(a.beta[s] + c);
        }
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    if constexpr (IS_PROJ) {
        if (live) {
            const size_t ro = (size_t)m * NTOT + 32 * ct0;
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                const int c = 4 * (sub + TPR * i);
                f32x4 o = v[i];
                if (a.skip[s]) o += *reinterpret_cast<const f32x4*>(a.skip[s] + ro + c);   // residual rows (may alias out: same thread, same element)
                *reinterpret_cast<f32x4*>(a.out[s] + ro + c) = o;
            }
        }
        DP_STAMP(10);
        return;
    }
    if constexpr (IS_QKV) {
        if (live) {
            const float sc = which == 0 ? a.qscale : 1.0f;
            f16* dst = a.qo[s][which] + (size_t)m * K + 32 * ct0 - which * K;
            typedef f16 f16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                const f16x4 h = {(f16)(v[i][0] * sc), (f16)(v[i][1] * sc), (f16)(v[i][2] * sc), (f16)(v[i][3] * sc)};
                *reinterpret_cast<f16x4*>(dst + 4 * (sub + TPR * i)) = h;
            }
        }
        DP_STAMP(10);
        return;
    }
    if constexpr (RAW) {
        if (live) {
#pragma unroll
            for (int i = 0; i < NO; ++i) *reinterpret_cast<f32x4*>(a.out[s] + (size_t)m * NTOT + 32 * ct0 + 4 * (sub + TPR * i)) = v[i];
        }
        DP_STAMP(10);
        return;
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        const float d0 = v[i][0] - mean, d1 = v[i][1] - mean, d2 = v[i][2] - mean, d3 = v[i][3] - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q * (1.0f / N) + 1e-5f);
    if (live) {
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        const int c = 4 * (sub + TPR * i);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = elu_fast((v[i][j] - mean) * rstd * gm[i][j] + bt[i][j]);
        if constexpr (IS_DEC) {
            const int p = c / COUT, cc = c - p * COUT;   // Cout % 4 == 0: a chunk never straddles sub-pixels
            const int y = 2 * ry + (p >> 1), xo = 2 * rx + (p & 1);
            if (y < a.Ho && xo < a.Wo) {
                o += sk[i];
                *reinterpret_cast<f32x4*>(a.out[s] + ((size_t)(rb * a.Ho + y) * a.Wo + xo) * COUT + cc) = o;
            }
        } else {
            *reinterpret_cast<f32x4*>(a.out[s] + (size_t)m * N + c) = o;
            v[i] = o;
        }
    }
    if constexpr (!IS_DEC) {
        if (a.ln_hi[s]) {   // the next block's LN1 over the finished row (same threads, same shuffles), written as planes
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NO; ++i) s2 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) s2 += __shfl_xor(s2, o);
            const float mean2 = s2 * (1.0f / N);
            float q2 = 0.f;
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                const float d0 = v[i][0] - mean2, d1 = v[i][1] - mean2, d2 = v[i][2] - mean2, d3 = v[i][3] - mean2;
                q2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) q2 += __shfl_xor(q2, o);
            const float rstd2 = 1.0f / sqrtf(q2 * (1.0f / N) + 1e-5f);
#pragma unroll
            for (int i = 0; i < NO; ++i) {
                const int c = 4 * (sub + TPR * i);
                const f32x4 g = *reinterpret_cast<const f32x4*>(a.ln_g[s] + c), bb = *reinterpret_cast<const f32x4*>(a.ln_b[s] + c);
                bf16x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float n = (v[i][j] - mean2) * rstd2 * g[j] + bb[j];
                    h[j] = (bf16)n;
                    l[j] = (bf16)(n - (float)h[j]);
                }
                *reinterpret_cast<bf16x4*>(a.ln_hi[s] + (size_t)m * N + c) = h;
                *reinterpret_cast<bf16x4*>(a.ln_lo[s] + (size_t)m * N + c) = l;
            }
        }
    }
    }
    if constexpr (IS_DEC && !RAW) {
        if (a.warm[0]) {   // blocks L, L + 8, ... share an XCD: together they touch the whole image (speed only)
            const int nblk = (int)(gridDim.x * gridDim.y), L = (int)(blockIdx.x + gridDim.x * blockIdx.y);
            const int nsl = max(1, nblk / 8), sl = (L / 8) % nsl;
            const int lines = (a.warm_bytes + 127) / 128, per = (lines + nsl - 1) / nsl, l0 = sl * per, l1 = min(lines, l0 + per);
            unsigned acc = 0;
            for (int s2 = 0; s2 < 2; ++s2)
                if (a.warm[s2])
                    for (int l = l0 + tid; l < l1; l += NT) acc ^= *reinterpret_cast<const unsigned*>(a.warm[s2] + (size_t)l * 128);
            if (acc == 0x9e3779b9u && a.M < 0) a.out[0][0] = 0.f;   // never true: keeps the loads alive
        }
    }
    DP_STAMP(10);
}

// ---- row finishers of the column-sliced ("raw") layers: one wave per conv output row ---------------------------------------
// Encoder (N = 384): LayerNorm + ELU -> out rows, then (optionally) the next block's LN1 of those rows as split-bf16 planes.
// Decoder (N = 768 = 4 sub-pixels x 192): LayerNorm over the whole row + ELU, depth-to-space scatter + skip -> out pixels, then
// (optionally) the next block's LN1 per output pixel as planes.  Lane l holds 16-byte groups l, l + 64, (l + 128): in the decoder
// group index = 48 p + q with p = l >> 4 the sub-pixel, so a pixel's 192 channels live in 16 consecutive lanes.
struct FinArgs {
    const float* z[2]; float* out[2]; const float* skip[2];
    const float* g1[2]; const float* b1[2];
    const float* g2[2]; const float* b2[2]; bf16* hi[2]; bf16* lo[2];   // optional second LayerNorm (planes), or nullptr
    int M, Hm, Wm, Ho, Wo;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void dp_finish_enc_kernel(FinArgs a) {
    constexpr int N = 384, NG = N / 4;   // 96 groups: lanes 0..63 hold group l, lanes 0..31 also group l + 64
    const int s = blockIdx.y, lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const bool two = lane + 64 < NG;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float* zr = a.z[s] + (size_t)m * N;
    f32x4 v0 = *reinterpret_cast<const f32x4*>(zr + 4 * lane), v1 = two ? *reinterpret_cast<const f32x4*>(zr + 4 * (lane + 64)) : zero4;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.g1[s] + 4 * lane), c0 = *reinterpret_cast<const f32x4*>(a.b1[s] + 4 * lane);
    const f32x4 g1 = two ? *reinterpret_cast<const f32x4*>(a.g1[s] + 4 * (lane + 64)) : zero4, c1 = two ? *reinterpret_cast<const f32x4*>(a.b1[s] + 4 * (lane + 64)) : zero4;
    const float mean = wave_sum((v0[0] + v0[1]) + (v0[2] + v0[3]) + (v1[0] + v1[1]) + (v1[2] + v1[3])) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d0 = v0[j] - mean, d1 = two ? v1[j] - mean : 0.f; q += d0 * d0 + d1 * d1; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / N) + 1e-5f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v0[j] = elu_fast((v0[j] - mean) * rstd * g0[j] + c0[j]);
        v1[j] = two ? elu_fast((v1[j] - mean) * rstd * g1[j] + c1[j]) : 0.f;
    }
    float* orow = a.out[s] + (size_t)m * N;
    *reinterpret_cast<f32x4*>(orow + 4 * lane) = v0;
    if (two) *reinterpret_cast<f32x4*>(orow + 4 * (lane + 64)) = v1;
    if (!a.hi[s]) return;
    const float mean2 = wave_sum((v0[0] + v0[1]) + (v0[2] + v0[3]) + (v1[0] + v1[1]) + (v1[2] + v1[3])) * (1.0f / N);
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d0 = v0[j] - mean2, d1 = two ? v1[j] - mean2 : 0.f; q2 += d0 * d0 + d1 * d1; }
    const float rstd2 = 1.0f / sqrtf(wave_sum(q2) * (1.0f / N) + 1e-5f);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k == 1 && !two) break;
        const int c = 4 * (lane + 64 * k);
        const f32x4 v = k ? v1 : v0;
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.g2[s] + c), bb = *reinterpret_cast<const f32x4*>(a.b2[s] + c);
        bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float n = (v[j] - mean2) * rstd2 * g[j] + bb[j];
            h[j] = (bf16)n;
            l[j] = (bf16)(n - (float)h[j]);
        }
        *reinterpret_cast<bf16x4*>(a.hi[s] + (size_t)m * N + c) = h;
        *reinterpret_cast<bf16x4*>(a.lo[s] + (size_t)m * N + c) = l;
    }
}

__global__ __launch_bounds__(256) void dp_finish_dec_kernel(FinArgs a) {
    constexpr int CO = 192, N = 4 * CO, GP = CO / 4;   // 48 groups per sub-pixel: lane (p = l >> 4, q = l & 15) holds groups q, q + 16, q + 32 of sub-pixel p
    const int s = blockIdx.y, lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int p = lane >> 4, q0 = lane & 15;
    const int mx = m % a.Wm, t = m / a.Wm, my = t % a.Hm, b = t / a.Hm;
    const int y = 2 * my + (p >> 1), xo = 2 * mx + (p & 1);
    const bool inside = y < a.Ho && xo < a.Wo;
    const size_t pix = ((size_t)(b * a.Ho + y) * a.Wo + xo) * CO;
    const float* zr = a.z[s] + (size_t)m * N + p * CO;
    f32x4 v[3], sk[3];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = 4 * (q0 + 16 * i);
        v[i] = *reinterpret_cast<const f32x4*>(zr + c);
        sk[i] = (a.skip[s] && inside) ? *reinterpret_cast<const f32x4*>(a.skip[s] + pix + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(sum) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / N) + 1e-5f);
    float sum2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = p * CO + 4 * (q0 + 16 * i);
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.g1[s] + c), bb = *reinterpret_cast<const f32x4*>(a.b1[s] + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = elu_fast((v[i][j] - mean) * rstd * g[j] + bb[j]) + sk[i][j];
        if (inside) *reinterpret_cast<f32x4*>(a.out[s] + pix + 4 * (q0 + 16 * i)) = v[i];
        sum2 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    if (!a.hi[s]) return;
    // LN1 of the output pixel: its 192 channels sit in the 16 lanes of sub-pixel p
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum2 += __shfl_xor(sum2, o);
    const float mean2 = sum2 * (1.0f / CO);
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean2; q2 += d * d; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
    const float rstd2 = 1.0f / sqrtf(q2 * (1.0f / CO) + 1e-5f);
    if (!inside) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = 4 * (q0 + 16 * i);
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.g2[s] + c), bb = *reinterpret_cast<const f32x4*>(a.b2[s] + c);
        bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float n = (v[i][j] - mean2) * rstd2 * g[j] + bb[j];
            h[j] = (bf16)n;
            l[j] = (bf16)(n - (float)h[j]);
        }
        *reinterpret_cast<bf16x4*>(a.hi[s] + pix + c) = h;
        *reinterpret_cast<bf16x4*>(a.lo[s] + pix + c) = l;
    }
    (void)GP;
}

// fp32 [N][K] -> fragment-major split planes: block (32-row tile rt, k16 step ks) = 64 lanes x 8 bf16, lane 32 hf + r holds
// row 32 rt + r, k = 16 ks + 8 hf .. + 7
__global__ __launch_bounds__(256) void dp_pack_kernel(const float* __restrict__ src, bf16* __restrict__ hi, bf16* __restrict__ lo, int N, int K) {
    const int total = N * K, ksteps = K / 16;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int block = e >> 9, within = e & 511, lane = within >> 3, j = within & 7, hf = lane >> 5, r = lane & 31;
        const int ks = block % ksteps, rt = block / ksteps;
        const float v = src[(size_t)(32 * rt + r) * K + 16 * ks + 8 * hf + j];
        const bf16 h = (bf16)v;
        hi[e] = h;
        lo[e] = (bf16)(v - (float)h);
    }
}

struct Shape { int dec, K, N, raw; };   // raw: conv + bias only (column slices over more workgroups); the caller runs LayerNorm
constexpr Shape kShapes[] = {{0, 384, 192, 0}, {0, 768, 384, 1}, {1, 384, 768, 1}, {1, 192, 384, 0}};

int shape_index(int decoder, int Cin, int Cout, int mh, int mw) {
    static const bool off = debug_env("SWF_NO_DEEP_PATCH") != nullptr;   // A/B switch (tools)
    if (off || mh != 2 || mw != 2 || Cin % 4 || Cout % 4) return -1;
    const int K = decoder ? Cin : 4 * Cin, N = decoder ? 4 * Cout : Cout;
    for (int i = 0; i < 4; ++i)
        if (kShapes[i].dec == (decoder ? 1 : 0) && kShapes[i].K == K && kShapes[i].N == N) return i;
    return -1;
}

template <int K, int N, int MT, int NW, int DEC, int NTOT>
int launch_t(const DpArgs& a, int nstream, hipStream_t stream) {
    constexpr int KC = K > 384 ? 384 : K, MR = 32 * MT;
    constexpr size_t img = size_t(2) * MR * (KC + 8) * 2, outb = size_t(MR) * (N + 4) * 4, lds = img > outb ? img : outb;
    static_assert(lds <= 160 * 1024, "LDS");
    static hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_patch_kernel<K, N, MT, NW, DEC, NTOT>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr_err != hipSuccess) return fail(SWF_ERR_HIP, "hipFuncSetAttribute(deep_patch): %s", hipGetErrorString(attr_err));
    hipLaunchKernelGGL((deep_patch_kernel<K, N, MT, NW, DEC, NTOT>), dim3((a.M + MR - 1) / MR, nstream, NTOT / N), dim3(64 * NW), lds, stream, a);
    return check_launch("deep_patch");
}

}  // namespace

int launch_deep_patch_finish(const PatchFusedDesc& d, float* const* raw, int nstream, hipStream_t stream, const DeepPatchExtra* extra) {
    const int i = shape_index(d.decoder, d.Cin, d.Cout, d.mh, d.mw);
    if (i < 0 || !kShapes[i].raw) return fail(SWF_ERR_UNSUPPORTED, "deep_patch_finish: not a column-sliced shape");
    FinArgs a{};
    for (int s = 0; s < nstream; ++s) {
        if (!raw || !raw[s] || !d.out[s] || !d.gamma[s] || !d.beta[s]) return fail(SWF_ERR_NULL, "deep_patch_finish: NULL operand (stream %d)", s);
        a.z[s] = raw[s]; a.out[s] = d.out[s]; a.skip[s] = d.skip[s]; a.g1[s] = d.gamma[s]; a.b1[s] = d.beta[s];
        if (extra && extra->ln_hi[s]) {
            a.g2[s] = extra->ln_gamma[s]; a.b2[s] = extra->ln_beta[s];
            a.hi[s] = reinterpret_cast<bf16*>(extra->ln_hi[s]); a.lo[s] = reinterpret_cast<bf16*>(extra->ln_lo[s]);
        }
    }
    a.M = (int)d.M; a.Hm = d.Hm; a.Wm = d.Wm; a.Ho = d.Ho; a.Wo = d.Wo;
    const dim3 grid((unsigned)((d.M + 3) / 4), nstream);
    if (d.decoder) hipLaunchKernelGGL(dp_finish_dec_kernel, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(dp_finish_enc_kernel, grid, dim3(256), 0, stream, a);
    return check_launch("deep_patch_finish");
}

bool deep_qkv_supported(const swf_block_desc& d) {
    static const bool off = debug_env("SWF_NO_DEEP_QKV") != nullptr;   // A/B switch (tools)
    const int C = d.attn.channels;
    // C = 192 only where the fused Q/K/V + attention kernel does not cover the block (16x16 windows)
    return !off && d.precision == SWF_PREC_FAST && d.attn.heads * d.attn.head_dim == C && (C == 384 || (C == 192 && !qkvattn_supported(d)));
}

bool deep_proj_supported(const swf_block_desc& d) {
    static const bool off = debug_env("SWF_NO_DEEP_PROJ") != nullptr;   // A/B switch (tools)
    const int C = d.attn.channels;
    return !off && d.precision == SWF_PREC_FAST && d.attn.heads * d.attn.head_dim == C && (C == 384 || C == 192);
}

int launch_deep_proj(const DeepProjArgs& q, int nstream, hipStream_t stream) {
    if (q.M <= 0 || q.M > INT32_MAX / 2048) return fail(SWF_ERR_UNSUPPORTED, "deep_proj: token count");
    if (q.C != 192 && q.C != 384) return fail(SWF_ERR_UNSUPPORTED, "deep_proj: C=%d", q.C);
    DpArgs a{};
    for (int s = 0; s < nstream; ++s) {
        if (!q.o_hi[s] || !q.o_lo[s] || !q.w_hi[s] || !q.w_lo[s] || !q.out[s]) return fail(SWF_ERR_NULL, "deep_proj: NULL operand (stream %d)", s);
        a.xh[s] = reinterpret_cast<const bf16*>(q.o_hi[s]); a.xl[s] = reinterpret_cast<const bf16*>(q.o_lo[s]);
        a.w_hi[s] = reinterpret_cast<const bf16*>(q.w_hi[s]); a.w_lo[s] = reinterpret_cast<const bf16*>(q.w_lo[s]);
        a.bias[s] = q.bias[s]; a.skip[s] = q.res[s]; a.out[s] = q.out[s];
    }
    a.M = q.M;
    return q.C == 192 ? launch_t<192, 192, 2, 4, 3, 192>(a, nstream, stream) : launch_t<384, 192, 2, 4, 3, 384>(a, nstream, stream);
}

int launch_deep_qkv(const DeepQkvArgs& q, int nstream, hipStream_t stream) {
    if (q.M <= 0 || q.M > INT32_MAX / 2048) return fail(SWF_ERR_UNSUPPORTED, "deep_qkv: token count");
    DpArgs a{};
    for (int s = 0; s < nstream; ++s) {
        if (!q.xn_hi[s] || !q.xn_lo[s] || !q.w_hi[s] || !q.w_lo[s]) return fail(SWF_ERR_NULL, "deep_qkv: NULL operand (stream %d)", s);
        a.xh[s] = reinterpret_cast<const bf16*>(q.xn_hi[s]); a.xl[s] = reinterpret_cast<const bf16*>(q.xn_lo[s]);
        a.w_hi[s] = reinterpret_cast<const bf16*>(q.w_hi[s]); a.w_lo[s] = reinterpret_cast<const bf16*>(q.w_lo[s]);
        for (int i = 0; i < 3; ++i) {
            if (!q.out[s][i]) return fail(SWF_ERR_NULL, "deep_qkv: NULL output (stream %d)", s);
            a.qb[s][i] = q.bias[s][i]; a.qo[s][i] = reinterpret_cast<f16*>(q.out[s][i]);
        }
    }
    a.M = q.M; a.qscale = q.qscale; a.cross = q.cross && nstream == 2;
    if (q.C == 192) return launch_t<192, 192, 2, 4, 2, 576>(a, nstream, stream);
    if (q.C != 384) return fail(SWF_ERR_UNSUPPORTED, "deep_qkv: C=%d", q.C);
    return launch_t<384, 192, 2, 4, 2, 1152>(a, nstream, stream);
}

bool deep_patch_supported(int decoder, int Cin, int Cout, int mh, int mw) { return shape_index(decoder, Cin, Cout, mh, mw) >= 0; }
bool deep_patch_raw(int decoder, int Cin, int Cout, int mh, int mw) {
    const int i = shape_index(decoder, Cin, Cout, mh, mw);
    return i >= 0 && kShapes[i].raw;
}

size_t deep_patch_packed_bytes(int decoder, int Cin, int Cout, int mh, int mw) {
    const int i = shape_index(decoder, Cin, Cout, mh, mw);
    return i < 0 ? 0 : align_up((size_t)kShapes[i].K * kShapes[i].N * 4, 256);   // hi plane | lo plane
}

int pack_deep_patch(int decoder, int Cin, int Cout, int mh, int mw, const float* weight, void* dst, hipStream_t stream) {
    const int i = shape_index(decoder, Cin, Cout, mh, mw);
    if (i < 0) return fail(SWF_ERR_UNSUPPORTED, "pack_deep_patch: shape not covered");
    const int K = kShapes[i].K, N = kShapes[i].N;
    bf16* hi = static_cast<bf16*>(dst);
    hipLaunchKernelGGL(dp_pack_kernel, dim3(256), dim3(256), 0, stream, weight, hi, hi + (size_t)N * K, N, K);
    return check_launch("pack_deep_patch");
}

int launch_deep_patch(const PatchFusedDesc& d, const void* const* packed, int nstream, hipStream_t stream, float* const* raw_out,
                      const DeepPatchExtra* extra) {
    const int i = shape_index(d.decoder, d.Cin, d.Cout, d.mh, d.mw);
    if (i < 0) return fail(SWF_ERR_UNSUPPORTED, "deep_patch: shape not covered");
    if (d.M <= 0 || d.M > INT32_MAX / 1024 || (int64_t)d.B * d.H * d.W * d.Cin > INT32_MAX) return fail(SWF_ERR_UNSUPPORTED, "deep_patch: map too large");
    const int K = kShapes[i].K, N = kShapes[i].N;
    DpArgs a{};
    for (int s = 0; s < nstream; ++s) {
        if (!packed || !packed[s] || !d.in[s] || !d.out[s] || !d.gamma[s] || !d.beta[s]) return fail(SWF_ERR_NULL, "deep_patch: NULL operand (stream %d)", s);
        if (kShapes[i].raw && (!raw_out || !raw_out[s])) return fail(SWF_ERR_NULL, "deep_patch: this shape needs the conv output buffer (stream %d)", s);
        a.in[s] = d.in[s]; a.out[s] = kShapes[i].raw ? raw_out[s] : d.out[s]; a.skip[s] = d.skip[s];
        a.w_hi[s] = static_cast<const bf16*>(packed[s]); a.w_lo[s] = a.w_hi[s] + (size_t)N * K;
        a.bias[s] = d.bias[s]; a.gamma[s] = d.gamma[s]; a.beta[s] = d.beta[s];
    }
    a.B = d.B; a.H = d.H; a.W = d.W; a.Hm = d.Hm; a.Wm = d.Wm; a.Ho = d.Ho; a.Wo = d.Wo; a.M = (int)d.M;
    if (extra && !kShapes[i].raw) {
        for (int s = 0; s < nstream; ++s) {
            if (!d.decoder && extra->ln_hi[s]) {
                a.ln_g[s] = extra->ln_gamma[s]; a.ln_b[s] = extra->ln_beta[s];
                a.ln_hi[s] = reinterpret_cast<bf16*>(extra->ln_hi[s]); a.ln_lo[s] = reinterpret_cast<bf16*>(extra->ln_lo[s]);
            }
            if (d.decoder) a.warm[s] = static_cast<const char*>(extra->warm[s]);
        }
        a.warm_bytes = (int)extra->warm_bytes;
    }
    switch (i) {
        case 0: return launch_t<384, 192, 2, 4, 0, 192>(a, nstream, stream);
        case 1: return launch_t<768, 128, 2, 4, 0, 384>(a, nstream, stream);
        case 2: return launch_t<384, 128, 2, 4, 1, 768>(a, nstream, stream);
        default: return launch_t<192, 384, 2, 8, 1, 384>(a, nstream, stream);
    }
}

}  // namespace swf
