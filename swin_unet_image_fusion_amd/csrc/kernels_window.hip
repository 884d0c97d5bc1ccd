// Fast tier (include/swinfuse.h SWF_PREC_FAST), window-level kernels of gfx950 (MI355X).  This file holds
//   * the dispatcher of the fused BasicBlock launch (a005:127-145, both streams): window_block_supported / packed_bytes /
//     pack_window_block / launch_window_block route C = 24, 48, 96 to the register-resident kernels of kernels_win24.hip,
//     kernels_win48.hip, kernels_win96.hip (8x8, 7x7 and 16x16 windows);
//     (the round-1 LDS-image block kernel those replaced is gone; DESIGN.md Appendix A keeps its measurements);
//   * the stand-alone MFMA attention cores on projection buffers: attn_core_mfma_kernel<D, WS> (8x8 / 7x7 windows: the deep
//     levels' core) and attn_core_mfma16_kernel<D> (16x16 windows, online softmax over key tiles), both with optional 16-bit
//     operand inputs and split-plane output for the deep-level GEMM path (kernels_deep.hip);
//   * l2_warm_kernel.
//
// Arithmetic (error budget measured in DESIGN.md):
//   linear layers : split-bf16 "bf16x3" MFMA — a = a_hi + a_lo, w = w_hi + w_lo, a.w ~= a_lo.w_hi + a_hi.w_lo + a_hi.w_hi,
//                   fp32 accumulate (~2^-17 relative: fp32-grade; plain bf16 linears miss the 1e-3 parity gate by 4-10x,
//                   gfx950 has no xf32/tf32)
//   Q.K^T, P.V    : f16 on v_mfma_f32_32x32x16_f16, computed swapped (S^T = K.Q^T) so a lane owns one query column: softmax
//                   max / sum are in-lane + one cross-half exchange; the S^T accumulator tile converts in registers to the B
//                   operand of O^T = V^T.P^T (V^T stored in the matching k order)
//   LayerNorm statistics, bias, mask, softmax, ELU, residual stream, accumulators: fp32.
//   exp() runs as v_exp_f32 (exp2): Wq/bq carry d^-0.5*log2(e), the bias matrices carry log2(e).
//
#include "kernels_window.h"
#include "kernels_win24.h"
#include "kernels_win48.h"
#include "kernels_win96.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>

namespace swf {

using bf16 = __bf16;
using f16 = _Float16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float kLog2e = 1.4426950408889634f;
constexpr int cceil(int a, int b) { return (a + b - 1) / b; }
constexpr int cround(int a, int b) { return cceil(a, b) * b; }
constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ELU(alpha=1) for the fast tier: exp(v)-1 through v_exp_f32.  Near 0 the subtraction cancels, leaving an
// ABSOLUTE error of ~1e-7 on activations of order 1 — four orders below the tier's error budget; the exact
// tier keeps expm1f.
__device__ __forceinline__ float elu_fast(float v) { return v > 0.f ? v : __builtin_amdgcn_exp2f(v * kLog2e) - 1.0f; }

// max of three; with -fno-honor-nans hipcc folds this into one v_max3_f32 (and drops the canonicalising
// v_max it would otherwise put in front of fmaxf on MFMA outputs).  NOT inline asm: an asm statement that
// reads an MFMA result gets none of the MFMA->VALU wait states and reads stale registers.
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

__device__ __forceinline__ void split4_bf16(const float v[4], bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (bf16)v[i];
        lo[i] = (bf16)(v[i] - (float)hi[i]);
    }
}

// position of key `tok` (0..63) inside a V^T row so that the 8 halves a lane needs for k-step s of
// key tile T sit contiguously: the S^T accumulator register 8s+e of lane half h is key row
// 32T + 16s + 8(e>>2) + 4h + (e&3)  (C/D map of the 32x32 MFMA), so pos = 32T + 16s + 8h + e.
// For tok = 4a .. 4a+3 the positions are consecutive (only e&3 changes): one 8-byte store.
__device__ __forceinline__ int vt_pos(int tok) {
    const int k16 = tok & 15;
    const int e = ((k16 >> 3) << 2) | (k16 & 3);
    const int h = (k16 >> 2) & 1;
    return (tok & 48) | (h << 3) | e;
}


// ------------------------------------------------------------------------------------------
// Stand-alone MFMA attention core for 8x8 windows (levels whose linears run as separate GEMMs):
// one workgroup = (window, group of 4 heads, stream), one wave per head.  Same arithmetic as phase 3 of
// the block kernel; Q/K/V come from the fp32 projection buffers in global memory (token-major, image
// order; the cyclic shift is index arithmetic), the bias (+mask) matrix is built per window in LDS.
// ------------------------------------------------------------------------------------------
struct AttnMfmaArgs {
    const float* Q[2]; const float* K[2]; const float* V[2]; float* O[2]; const float* table[2];
    unsigned short* Ohi[2]; unsigned short* Olo[2];   // non-null: O is written as split-bf16 planes (row stride ldo) instead
    const unsigned short* Q16[2]; const unsigned short* K16[2]; const unsigned short* V16[2];   // non-null: Q (pre-scaled bf16), K (bf16), V (fp16)
    int ldq, ldk, ldv, ldo, B, H, W, heads, shift;
};

template <int D, int WS>
__global__ __launch_bounds__(128) void attn_core_mfma_kernel(AttnMfmaArgs a) {
    // one workgroup = (window, head, stream); wave = 32-query block.  WS = 7 (the reference's default window, A000_CONFIG.py:55)
    // runs on the same 8x8 token grid: the 15 padding tokens are staged as zeros, carry probability 0 as keys and are not stored
    static_assert(WS == 7 || WS == 8, "window side");
    constexpr int T = 64, WH = WS, WW = WS, QS = cround(D, 8), QKS = cceil(D, 16), MT = cceil(D, 32), TW = 2 * WW - 1, VRS = T + 8;
    constexpr int VEC = (D % 4 == 0) ? 4 : ((D % 2 == 0) ? 2 : 1);   // floats per global load of a head's channel run
    constexpr int CPT = D / VEC, NCHUNK = T * CPT, NIT = cceil(NCHUNK, 128);
    __shared__ __attribute__((aligned(16))) f16 qimg[T * QS];
    __shared__ __attribute__((aligned(16))) f16 kimg[T * QS];
    __shared__ __attribute__((aligned(16))) f16 vt[D * VRS];
    __shared__ float tab[(2 * WH - 1) * TW];

    const int p = blockIdx.z, head = blockIdx.y, tid = threadIdx.x, lane = tid & 63, qb = tid >> 6;
    const int H = a.H, W = a.W, nwx = W / WW, nwy = H / WH;
    const int win = blockIdx.x;
    const int b = win / (nwx * nwy), wrem = win % (nwx * nwy), wy = wrem / nwx, wx = wrem % nwx;
    const int sh = a.shift ? WH / 2 : 0, sw = a.shift ? WW / 2 : 0;
    const float qscale = kLog2e / sqrtf((float)D);

    for (int i = tid; i < (2 * WH - 1) * TW; i += 128) tab[i] = a.table[p][i] * kLog2e;
    if constexpr (QS != D) {   // K padding of the Q / K rows
        for (int i = tid; i < T * (QS - D); i += 128) {
            const int tok = i / (QS - D), c = D + i % (QS - D);
            qimg[tok * QS + c] = (f16)0.f;
            kimg[tok * QS + c] = (f16)0.f;
        }
    }
    if constexpr (VEC == 4) {
        if (a.Q16[p]) {   // operands already in their final 16-bit formats (deep-level Q/K/V GEMM epilogue): plain copies
            uint2 q2[NIT], k2[NIT], v2[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + it * 128;
                if (e < NCHUNK) {
                    const int tok = e / CPT, c0 = (e % CPT) * VEC;
                    q2[it] = k2[it] = v2[it] = make_uint2(0u, 0u);
                    if ((tok >> 3) < WS && (tok & 7) < WS) {
                        const int oy = (wy * WH + (tok >> 3) + sh) % H, ox = (wx * WW + (tok & 7) + sw) % W;
                        const int64_t t = ((int64_t)b * H + oy) * W + ox;
                        q2[it] = *reinterpret_cast<const uint2*>(a.Q16[p] + t * a.ldq + head * D + c0);
                        k2[it] = *reinterpret_cast<const uint2*>(a.K16[p] + t * a.ldk + head * D + c0);
                        v2[it] = *reinterpret_cast<const uint2*>(a.V16[p] + t * a.ldv + head * D + c0);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + it * 128;
                if (e < NCHUNK) {
                    const int tok = e / CPT, c0 = (e % CPT) * VEC;
                    const int vp = vt_pos(tok);
                    *reinterpret_cast<uint2*>(qimg + tok * QS + c0) = q2[it];
                    *reinterpret_cast<uint2*>(kimg + tok * QS + c0) = k2[it];
                    const f16x4 v4 = __builtin_bit_cast(f16x4, v2[it]);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) vt[(c0 + j) * VRS + vp] = v4[j];
                }
            }
            goto staged;
        }
    }
    {
    // Q / K / V of this head -> LDS images; all loads of a thread are issued before the first conversion
    float qv[NIT][VEC], kv[NIT][VEC], vv[NIT][VEC];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 128;
        if (e < NCHUNK) {
            const int tok = e / CPT, c0 = (e % CPT) * VEC;
#pragma unroll
            for (int j = 0; j < VEC; ++j) qv[it][j] = kv[it][j] = vv[it][j] = 0.f;
            if ((tok >> 3) >= WS || (tok & 7) >= WS) continue;   // padding token of a 7x7 window
            const int oy = (wy * WH + (tok >> 3) + sh) % H, ox = (wx * WW + (tok & 7) + sw) % W;
            const int64_t t = ((int64_t)b * H + oy) * W + ox;
            const float* qp = a.Q[p] + t * a.ldq + head * D + c0;
            const float* kp = a.K[p] + t * a.ldk + head * D + c0;
            const float* vp = a.V[p] + t * a.ldv + head * D + c0;
            if constexpr (VEC == 4) {
                const float4 x = *reinterpret_cast<const float4*>(qp), y = *reinterpret_cast<const float4*>(kp), z = *reinterpret_cast<const float4*>(vp);
                qv[it][0] = x.x; qv[it][1] = x.y; qv[it][2] = x.z; qv[it][3] = x.w;
                kv[it][0] = y.x; kv[it][1] = y.y; kv[it][2] = y.z; kv[it][3] = y.w;
                vv[it][0] = z.x; vv[it][1] = z.y; vv[it][2] = z.z; vv[it][3] = z.w;
            } else if constexpr (VEC == 2) {
                const float2 x = *reinterpret_cast<const float2*>(qp), y = *reinterpret_cast<const float2*>(kp), z = *reinterpret_cast<const float2*>(vp);
                qv[it][0] = x.x; qv[it][1] = x.y; kv[it][0] = y.x; kv[it][1] = y.y; vv[it][0] = z.x; vv[it][1] = z.y;
            } else {
                qv[it][0] = *qp; kv[it][0] = *kp; vv[it][0] = *vp;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 128;
        if (e < NCHUNK) {
            const int tok = e / CPT, c0 = (e % CPT) * VEC;
            const int vp = vt_pos(tok);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                qimg[tok * QS + c0 + j] = (f16)(qv[it][j] * qscale);
                kimg[tok * QS + c0 + j] = (f16)kv[it][j];
                vt[(c0 + j) * VRS + vp] = (f16)vv[it][j];
            }
        }
    }
    }
staged:
    __syncthreads();

    const int r = lane & 31, hf = lane >> 5;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int q = 32 * qb + r, qy = q >> 3, qx = q & 7;
    const bool last_row = a.shift && wy == nwy - 1, last_col = a.shift && wx == nwx - 1;
    const f16* qrow = qimg + q * QS;
    f32x16 acc[2];
    // accumulator init = relative-position bias (+ shift mask), exp2 units.  Register i of key tile kt is key row
    // ky = 4*kt + (i>>2), column kx = (i&3) + 4*hf  (C/D map of the 32x32 MFMA with 8 keys per window row).
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ky = 4 * kt + (i >> 2), kx = (i & 3) + 4 * hf;
            const bool pad_k = ky >= WS || kx >= WS, pad_q = qy >= WS || qx >= WS;
            float v = (pad_k || pad_q) ? 0.f : tab[(ky - qy + WH - 1) * TW + (kx - qx + WW - 1)];
            const bool my = last_row && ((ky >= WH - WH / 2) != (qy >= WH - WH / 2));
            const bool mx = last_col && ((kx >= WW - WW / 2) != (qx >= WW - WW / 2));
            acc[kt][i] = (my || mx || pad_k) ? -1e10f * kLog2e : v;
        }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const f16* krow = kimg + (32 * kt + r) * QS;
#pragma unroll
        for (int ks = 0; ks < QKS; ++ks) {
            f16x8 ka = zero8, qf = zero8;
            if (ks * 16 + 8 * hf + 8 <= QS) {
                ka = *reinterpret_cast<const f16x8*>(krow + ks * 16 + 8 * hf);
                qf = *reinterpret_cast<const f16x8*>(qrow + ks * 16 + 8 * hf);
            }
            acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qf, acc[kt], 0, 0, 0);
        }
    }
    float mx = max3f(acc[0][0], acc[0][1], acc[1][0]);
    mx = max3f(mx, acc[1][1], acc[0][2]);
#pragma unroll
    for (int i = 3; i < 16; i += 2) mx = max3f(mx, acc[0][i], acc[0][i + 1 < 16 ? i + 1 : i]);
#pragma unroll
    for (int i = 2; i < 16; i += 2) mx = max3f(mx, acc[1][i], acc[1][i + 1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pv = __builtin_amdgcn_exp2f(acc[kt][i] - mx);
            acc[kt][i] = pv;
            l += pv;
        }
    l += __shfl_xor(l, 32);
    f32x16 o[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[mt][i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f16x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (f16)acc[kt][8 * s2 + e];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int c = mt * 32 + r;
                c = c < D ? c : D - 1;
                const f16x8 va = *reinterpret_cast<const f16x8*>(vt + c * VRS + kt * 32 + s2 * 16 + 8 * hf);
                o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pf, o[mt], 0, 0, 0);
            }
        }
    const float inv = 1.0f / l;
    if (qy >= WS || qx >= WS) return;   // padding token: nothing to store (no barrier follows)
    const int oy = (wy * WH + qy + sh) % H, ox = (wx * WW + qx + sw) % W;
    const int64_t ooff = (((int64_t)b * H + oy) * W + ox) * a.ldo + head * D;
    if constexpr (D % 4 == 0) {
        if (a.Ohi[p]) {   // split-bf16 planes for the deep-level projection GEMM (kernels_deep.h)
            bf16* hrow = reinterpret_cast<bf16*>(a.Ohi[p]) + ooff;
            bf16* lrow = reinterpret_cast<bf16*>(a.Olo[p]) + ooff;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    const int c = mt * 32 + 8 * (i >> 2) + 4 * hf;
                    if (c < D) {
                        bf16x4 hi, lo;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = o[mt][i + j] * inv;
                            hi[j] = (bf16)v;
                            lo[j] = (bf16)(v - (float)hi[j]);
                        }
                        *reinterpret_cast<bf16x4*>(hrow + c) = hi;
                        *reinterpret_cast<bf16x4*>(lrow + c) = lo;
                    }
                }
            return;
        }
    }
    float* orow = a.O[p] + ooff;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            // registers i..i+3 of a lane are 4 consecutive channels of its query's token: one 16-byte store when aligned
            const int c = mt * 32 + 8 * (i >> 2) + 4 * hf;
            if constexpr (D % 4 == 0) {
                if (c < D) *reinterpret_cast<float4*>(orow + c) = make_float4(o[mt][i] * inv, o[mt][i + 1] * inv, o[mt][i + 2] * inv, o[mt][i + 3] * inv);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < D) orow[c + j] = o[mt][i + j] * inv;
            }
        }
}

template <int D>
static int launch_attn_mfma_t(const AttnMfmaArgs& a, int ws, int nprob, hipStream_t stream) {
    const int nwin = a.B * (a.H / ws) * (a.W / ws);
    if (ws == 8) hipLaunchKernelGGL((attn_core_mfma_kernel<D, 8>), dim3(nwin, a.heads, nprob), dim3(128), 0, stream, a);
    else hipLaunchKernelGGL((attn_core_mfma_kernel<D, 7>), dim3(nwin, a.heads, nprob), dim3(128), 0, stream, a);
    return check_launch("attn_core_mfma");
}

// ------------------------------------------------------------------------------------------
// MFMA attention core for 16x16 windows (T = 256 tokens; BASELINE config 5).  One workgroup = (window, head,
// stream), 8 waves = the 8 blocks of 32 queries.  A 256x256 fp32 score tile per head would be 256 KB — more than
// the CU's LDS — so each wave walks the 8 key tiles of 32 keys with an online softmax: running max m, the output
// accumulator rescaled by exp2(m_old - m_new) per tile, and the denominator carried by the all-ones row of V^T so it
// is rescaled together with the numerator.  The relative-position bias comes from the 31x31 table staged in LDS and the
// shift mask from index arithmetic; both form the C operand of each tile's first MFMA.
// ------------------------------------------------------------------------------------------
struct Attn16Args {
    const float* Q[2]; const float* K[2]; const float* V[2]; float* O[2];
    const float* table[2];    // relative-position bias table [(2*16-1)^2] per stream
    unsigned short* Ohi[2]; unsigned short* Olo[2];   // non-null: O is written as split-bf16 planes (row stride ldo) instead
    const unsigned short* Q16[2]; const unsigned short* K16[2]; const unsigned short* V16[2];   // non-null: Q (pre-scaled), K, V as f16 (deep-level GEMM epilogue)
    int ldq, ldk, ldv, ldo, B, H, W, heads, shift;
};

template <int D>
__global__ __launch_bounds__(512) void attn_core_mfma16_kernel(Attn16Args a) {
    constexpr int T = 256, WH = 16, WW = 16, QS = cround(D, 8), QKS = cceil(D, 16), MT = cceil(D, 32), VRS = T + 8;
    constexpr int VEC = (D % 4 == 0) ? 4 : ((D % 2 == 0) ? 2 : 1);
    constexpr int CPT = D / VEC, NCHUNK = T * CPT, NIT = cceil(NCHUNK, 512);
    extern __shared__ __attribute__((aligned(16))) char sm16[];
    f16* qimg = reinterpret_cast<f16*>(sm16);                    // [256][QS]
    f16* kimg = qimg + T * QS;
    f16* vt = reinterpret_cast<f16*>(kimg + T * QS);               // [D + 1][VRS]; row D = 1.0
    constexpr int TW = 2 * WW - 1, NTAB = (2 * WH - 1) * TW;
    float* tab = reinterpret_cast<float*>(sm16 + (size_t(2) * T * QS * 2 + size_t(D + 1) * VRS * 2 + 15) / 16 * 16);   // [31][31], exp2 units

    const int p = blockIdx.z, head = blockIdx.y, tid = threadIdx.x, lane = tid & 63, qb = tid >> 6;
    const int H = a.H, W = a.W, nwx = W / WW, nwy = H / WH;
    const int win = blockIdx.x;
    const int b = win / (nwx * nwy), wrem = win % (nwx * nwy), wy = wrem / nwx, wx = wrem % nwx;
    const int sh = a.shift ? WH / 2 : 0, sw = a.shift ? WW / 2 : 0;
    const float qscale = kLog2e / sqrtf((float)D);

    for (int i = tid; i < VRS; i += 512) vt[D * VRS + i] = (f16)1.0f;
    for (int i = tid; i < NTAB; i += 512) tab[i] = a.table[p][i] * kLog2e;
    if constexpr (QS != D) {
        for (int i = tid; i < T * (QS - D); i += 512) {
            const int tok = i / (QS - D), c = D + i % (QS - D);
            qimg[tok * QS + c] = (f16)0.f;
            kimg[tok * QS + c] = (f16)0.f;
        }
    }
    bool staged16 = false;
    if constexpr (VEC == 4) {
        if (a.Q16[p]) {   // operands already in their 16-bit formats (deep-level Q/K/V GEMM epilogue): plain copies
            staged16 = true;
            uint2 q2[NIT], k2[NIT], v2[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + it * 512;
                if (e < NCHUNK) {
                    const int tok = e / CPT, c0 = (e % CPT) * VEC;
                    const int oy = (wy * WH + tok / WW + sh) % H, ox = (wx * WW + tok % WW + sw) % W;
                    const int64_t t = ((int64_t)b * H + oy) * W + ox;
                    q2[it] = *reinterpret_cast<const uint2*>(a.Q16[p] + t * a.ldq + head * D + c0);
                    k2[it] = *reinterpret_cast<const uint2*>(a.K16[p] + t * a.ldk + head * D + c0);
                    v2[it] = *reinterpret_cast<const uint2*>(a.V16[p] + t * a.ldv + head * D + c0);
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + it * 512;
                if (e < NCHUNK) {
                    const int tok = e / CPT, c0 = (e % CPT) * VEC;
                    const int k16 = tok & 15;
                    const int vpos = (tok & ~15) | (((k16 >> 2) & 1) << 3) | (((k16 >> 3) << 2) | (k16 & 3));
                    *reinterpret_cast<uint2*>(qimg + tok * QS + c0) = q2[it];
                    *reinterpret_cast<uint2*>(kimg + tok * QS + c0) = k2[it];
                    const f16x4 v4 = __builtin_bit_cast(f16x4, v2[it]);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) vt[(c0 + j) * VRS + vpos] = v4[j];
                }
            }
        }
    }
    if (!staged16) {
    float qv[NIT][VEC], kv[NIT][VEC], vv[NIT][VEC];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 512;
        if (e < NCHUNK) {
            const int tok = e / CPT, c0 = (e % CPT) * VEC;
            const int oy = (wy * WH + tok / WW + sh) % H, ox = (wx * WW + tok % WW + sw) % W;
            const int64_t t = ((int64_t)b * H + oy) * W + ox;
            const float* qp = a.Q[p] + t * a.ldq + head * D + c0;
            const float* kp = a.K[p] + t * a.ldk + head * D + c0;
            const float* vp = a.V[p] + t * a.ldv + head * D + c0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) { qv[it][j] = qp[j]; kv[it][j] = kp[j]; vv[it][j] = vp[j]; }   // adjacent: hipcc merges into one vector load
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * 512;
        if (e < NCHUNK) {
            const int tok = e / CPT, c0 = (e % CPT) * VEC;
            const int k16 = tok & 15;
            const int vpos = (tok & ~15) | (((k16 >> 2) & 1) << 3) | (((k16 >> 3) << 2) | (k16 & 3));   // vt_pos for any number of key tiles
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                qimg[tok * QS + c0 + j] = (f16)(qv[it][j] * qscale);
                kimg[tok * QS + c0 + j] = (f16)kv[it][j];
                vt[(c0 + j) * VRS + vpos] = (f16)vv[it][j];
            }
        }
    }
    }
    __syncthreads();

    const int r = lane & 31, hf = lane >> 5;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int q = 32 * qb + r;
    // relative-position bias (a001:113-144) from the 31x31 table in LDS, shift mask (a001:217-315) by index arithmetic: only
    // windows in the last window row / column of a shifted block hold two region labels, split at wh/2 (ww/2).  (A
    // precomputed [4][256][256] matrix per stream cost 256 KB of L2 reads per (window, head): 34 GB per launch at 1024^2.)
    const int qy = q / WW, qx = q % WW;
    const bool vrow = a.shift && wy == nwy - 1, vcol = a.shift && wx == nwx - 1;
    const bool colmask0 = vcol && (qx >= WW - WW / 2), colmask1 = vcol && !(qx >= WW - WW / 2);   // key column half 0 / 1 masked for this query
    const float* tq = tab + (WH - 1 - qy) * TW + (WW - 1 - qx) + 4 * hf;
    constexpr float NEG = -1e10f * kLog2e;
    f16x8 qf[QKS];
#pragma unroll
    for (int ks = 0; ks < QKS; ++ks)
        qf[ks] = (ks * 16 + 8 * hf + 8 <= QS) ? *reinterpret_cast<const f16x8*>(qimg + q * QS + ks * 16 + 8 * hf) : zero8;
    f32x16 o[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[mt][i] = 0.f;
    float m = -INFINITY;
    for (int kt = 0; kt < T / 32; ++kt) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // register i of key tile kt is key row ky = 2kt + (i >> 3), column kx = 8((i >> 2) & 1) + 4hf + (i & 3)
            const float bv = tq[(2 * kt + (i >> 3)) * TW + 8 * ((i >> 2) & 1) + (i & 3)];
            const bool rowmask = vrow && ((2 * kt + (i >> 3) >= WH - WH / 2) != (qy >= WH - WH / 2));
            acc[i] = (rowmask || (((i >> 2) & 1) ? colmask1 : colmask0)) ? NEG : bv;
        }
        const f16* krow = kimg + (32 * kt + r) * QS;
#pragma unroll
        for (int ks = 0; ks < QKS; ++ks) {
            const f16x8 ka = (ks * 16 + 8 * hf + 8 <= QS) ? *reinterpret_cast<const f16x8*>(krow + ks * 16 + 8 * hf) : zero8;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qf[ks], acc, 0, 0, 0);
        }
        float mt_ = max3f(acc[0], acc[1], acc[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mt_ = max3f(mt_, acc[i], acc[i + 1]);
        mt_ = fmaxf(mt_, acc[15]);
        mt_ = fmaxf(mt_, __shfl_xor(mt_, 32));
        const float m_new = fmaxf(m, mt_);
        const float rescale = __builtin_amdgcn_exp2f(m - m_new);   // first tile: exp2(-inf) = 0 on a zero accumulator
        m = m_new;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[mt][i] *= rescale;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_exp2f(acc[i] - m_new);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f16x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (f16)acc[8 * s2 + e];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int c = mt * 32 + r;
                const f16x8 va = *reinterpret_cast<const f16x8*>(vt + (c < D ? c : D) * VRS + kt * 32 + s2 * 16 + 8 * hf);
                o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pf, o[mt], 0, 0, 0);
            }
        }
    }
    constexpr int LM = D / 32, LI = (D & 3) + 4 * ((D & 31) >> 3), LH = (D >> 2) & 1;
    float l = o[LM][LI];
    const float l_other = __shfl_xor(l, 32);
    l = (hf == LH) ? l : l_other;
    const float inv = 1.0f / l;
    const int oy = (wy * WH + q / WW + sh) % H, ox = (wx * WW + q % WW + sw) % W;
    if constexpr (D % 4 == 0) {
        if (a.Ohi[p]) {   // split-bf16 planes for the deep-level projection GEMM (kernels_deep.h)
            const int64_t ooff = (((int64_t)b * H + oy) * W + ox) * a.ldo + head * D;
            bf16* hrow = reinterpret_cast<bf16*>(a.Ohi[p]) + ooff;
            bf16* lrow = reinterpret_cast<bf16*>(a.Olo[p]) + ooff;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    const int c = mt * 32 + 8 * (i >> 2) + 4 * hf;
                    if (c < D) {
                        bf16x4 hi, lo;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = o[mt][i + j] * inv;
                            hi[j] = (bf16)v;
                            lo[j] = (bf16)(v - (float)hi[j]);
                        }
                        *reinterpret_cast<bf16x4*>(hrow + c) = hi;
                        *reinterpret_cast<bf16x4*>(lrow + c) = lo;
                    }
                }
            return;
        }
    }
    float* orow = a.O[p] + (((int64_t)b * H + oy) * W + ox) * a.ldo + head * D;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const int c = mt * 32 + 8 * (i >> 2) + 4 * hf;
            if constexpr (D % 4 == 0) {
                if (c < D) *reinterpret_cast<float4*>(orow + c) = make_float4(o[mt][i] * inv, o[mt][i + 1] * inv, o[mt][i + 2] * inv, o[mt][i + 3] * inv);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < D) orow[c + j] = o[mt][i + j] * inv;
            }
        }
}

template <int D>
static int launch_attn16_t(const Attn16Args& a, int nprob, hipStream_t stream) {
    constexpr int QS = cround(D, 8);
    constexpr size_t lds = (size_t(2) * 256 * QS * 2 + size_t(D + 1) * (256 + 8) * 2 + 15) / 16 * 16 + size_t(31) * 31 * 4;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    if (lds > 64 * 1024)
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_core_mfma16_kernel<D>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
    if (attr_err != hipSuccess) return fail(SWF_ERR_HIP, "hipFuncSetAttribute(attn16): %s", hipGetErrorString(attr_err));
    const int nwin = a.B * (a.H / 16) * (a.W / 16);
    hipLaunchKernelGGL((attn_core_mfma16_kernel<D>), dim3(nwin, a.heads, nprob), dim3(512), lds, stream, a);
    return check_launch("attn_core_mfma16");
}

size_t attn_core_mfma16_scratch_floats(int nprob) { (void)nprob; return 0; }

int launch_attn_core_mfma16(const float* const* Q, const float* const* K, const float* const* V, float* const* O,
                            const float* const* table, int nprob, int ldq, int ldk, int ldv, int ldo, int B, int H, int W,
                            int heads, int head_dim, int shift, float* bias_scratch, hipStream_t stream, unsigned short* const* O_hi,
                            unsigned short* const* O_lo, const unsigned short* const* Q16, const unsigned short* const* K16,
                            const unsigned short* const* V16) {
    (void)bias_scratch;   // no longer used: the bias comes from the 31x31 table staged in LDS
    Attn16Args a{};
    if ((Q16 || O_hi) && (head_dim % 4 || ldq % 4 || ldk % 4 || ldv % 4)) return fail(SWF_ERR_UNSUPPORTED, "attn_core_mfma16: 16-bit operands need head_dim and strides %% 4 == 0");
    for (int i = 0; i < nprob; ++i) {
        a.Q[i] = Q ? Q[i] : nullptr; a.K[i] = K ? K[i] : nullptr; a.V[i] = V ? V[i] : nullptr; a.O[i] = O ? O[i] : nullptr; a.table[i] = table[i];
        a.Ohi[i] = O_hi ? O_hi[i] : nullptr; a.Olo[i] = O_hi ? O_lo[i] : nullptr;
        a.Q16[i] = Q16 ? Q16[i] : nullptr; a.K16[i] = Q16 ? K16[i] : nullptr; a.V16[i] = Q16 ? V16[i] : nullptr;
    }
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.B = B; a.H = H; a.W = W; a.heads = heads; a.shift = shift;
    switch (head_dim) {
        case 3: return launch_attn16_t<3>(a, nprob, stream);
        case 6: return launch_attn16_t<6>(a, nprob, stream);
        case 12: return launch_attn16_t<12>(a, nprob, stream);
        case 24: return launch_attn16_t<24>(a, nprob, stream);
        case 48: return launch_attn16_t<48>(a, nprob, stream);
    }
    return fail(SWF_ERR_UNSUPPORTED, "attn_core_mfma16: head_dim %d", head_dim);
}

bool attn_core_mfma16_supported(int wh, int ww, int head_dim) {
    return wh == 16 && ww == 16 && (head_dim == 3 || head_dim == 6 || head_dim == 12 || head_dim == 24 || head_dim == 48);
}

bool attn_core_mfma_supported(int wh, int ww, int head_dim) {
    return wh == ww && (wh == 8 || wh == 7) && (head_dim == 3 || head_dim == 6 || head_dim == 12 || head_dim == 24 || head_dim == 48);
}

int launch_attn_core_mfma(const float* const* Q, const float* const* K, const float* const* V, float* const* O,
                          const float* const* table, int nprob, int ldq, int ldk, int ldv, int ldo, int B, int H, int W,
                          int heads, int head_dim, int shift, hipStream_t stream, unsigned short* const* O_hi,
                          unsigned short* const* O_lo, const unsigned short* const* Q16, const unsigned short* const* K16,
                          const unsigned short* const* V16, int win) {
    AttnMfmaArgs a{};
    if ((win != 7 && win != 8) || H % win || W % win) return fail(SWF_ERR_UNSUPPORTED, "attn_core_mfma: window %d on a %d x %d map", win, H, W);
    if (Q16 && (head_dim % 4 || ldq % 4 || ldk % 4 || ldv % 4)) return fail(SWF_ERR_UNSUPPORTED, "attn_core_mfma: 16-bit operands need head_dim and strides %% 4 == 0");
    if (O_hi && head_dim % 4) return fail(SWF_ERR_UNSUPPORTED, "attn_core_mfma: split-plane output needs head_dim %% 4 == 0");
    for (int i = 0; i < nprob; ++i) {
        a.Q[i] = Q ? Q[i] : nullptr; a.K[i] = K ? K[i] : nullptr; a.V[i] = V ? V[i] : nullptr; a.O[i] = O ? O[i] : nullptr; a.table[i] = table[i];
        a.Q16[i] = Q16 ? Q16[i] : nullptr; a.K16[i] = Q16 ? K16[i] : nullptr; a.V16[i] = Q16 ? V16[i] : nullptr;
        a.Ohi[i] = O_hi ? O_hi[i] : nullptr; a.Olo[i] = O_hi ? O_lo[i] : nullptr;
    }
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.B = B; a.H = H; a.W = W; a.heads = heads; a.shift = shift;
    switch (head_dim) {
        case 3: return launch_attn_mfma_t<3>(a, win, nprob, stream);
        case 6: return launch_attn_mfma_t<6>(a, win, nprob, stream);
        case 12: return launch_attn_mfma_t<12>(a, win, nprob, stream);
        case 24: return launch_attn_mfma_t<24>(a, win, nprob, stream);
        case 48: return launch_attn_mfma_t<48>(a, win, nprob, stream);
    }
    return fail(SWF_ERR_UNSUPPORTED, "attn_core_mfma: head_dim %d", head_dim);
}
// ------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------
// Levels 0 / 1 / 2 (C = 24 / 48 / 96) run on the register-resident kernels of kernels_win24.hip / kernels_win48.hip /
// kernels_win96.hip.  Their token rows travel through 32-bit buffer descriptors, so ONE launch covers at most 2^31 - 1 bytes
// of a stream's map; images are independent (LayerNorm per token, attention per window), so a larger batch is launched in
// batch slices (launch_window_block) and only a single image beyond that size is left to the unfused tier.
static bool use_win24(const swf_block_desc& d) { return win24_supported(d); }
static bool use_win48(const swf_block_desc& d) { return win48_supported(d); }
static bool use_win96(const swf_block_desc& d) { return win96_supported(d); }

static int64_t image_bytes(const swf_block_desc& d, int H, int W) { return (int64_t)H * W * d.attn.channels * 4; }
constexpr int64_t kMaxLaunchBytes = (int64_t(1) << 31) - 1;

bool window_block_supported(const swf_block_desc& d, int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return false;
    if (!(use_win24(d) || use_win48(d) || use_win96(d))) return false;
    return H % d.attn.win_h == 0 && W % d.attn.win_w == 0 && image_bytes(d, H, W) <= kMaxLaunchBytes;
}

size_t window_block_packed_bytes(const swf_block_desc& d) {
    if (use_win24(d)) return win24_packed_bytes(d);
    if (use_win48(d)) return win48_packed_bytes(d);
    if (use_win96(d)) return win96_packed_bytes(d);
    return 0;
}

// The 16x16-window kernel at C = 48 runs one workgroup per (window, stream): in a cross block the workgroup of one stream reads the
// other stream's tokens while that stream's workgroup writes its results, so the outputs must not alias the inputs.
bool window_block_out_of_place(const swf_block_desc& d) {
    return (use_win48(d) || use_win96(d)) && d.attn.win_h == 16;
}

size_t window_block_workspace_bytes(const swf_block_desc& d, int B, int H, int W) {
    if (!window_block_supported(d, B, H, W)) return 0;
    // packed weights of both streams (callers without a pre-packed image) + two temporary output maps for in-place callers
    return 2 * window_block_packed_bytes(d) + (window_block_out_of_place(d) ? (size_t)2 * B * H * W * d.attn.channels * 4 + 512 : 0);
}

int pack_window_block(const swf_block_desc& d, const swf_block_stream_params& px, const swf_block_stream_params& py,
                      void* packed_x, void* packed_y, hipStream_t stream) {
    if (use_win24(d)) return pack_win24(d, px, py, packed_x, packed_y, stream);
    if (use_win48(d)) return pack_win48(d, px, py, packed_x, packed_y, stream);
    if (use_win96(d)) return pack_win96(d, px, py, packed_x, packed_y, stream);
    return fail(SWF_ERR_UNSUPPORTED, "pack_window_block: C=%d hidden=%d not covered", d.attn.channels, d.hidden);
}

// Touch `bytes` at p from every XCD (blocks b, b+8, ... share an XCD under round-robin dealing: speed only), so that the lines
// are resident in all eight L2s when the next launch reads them.
__global__ __launch_bounds__(256) void l2_warm_kernel(const char* __restrict__ p, int bytes, int* sink) {
    const int nsl = max(1, (int)gridDim.x / 8), sl = ((int)blockIdx.x / 8) % nsl;
    const int lines = (bytes + 127) / 128, per = (lines + nsl - 1) / nsl, l0 = sl * per, l1 = min(lines, l0 + per);
    unsigned acc = 0;
    for (int l = l0 + (int)threadIdx.x; l < l1; l += 256) acc ^= *reinterpret_cast<const unsigned*>(p + (size_t)l * 128);
    if (acc == 0x9e3779b9u && bytes < 0) *sink = 0;   // never true: keeps the loads alive
}

int launch_l2_warm(const void* p, size_t bytes, hipStream_t stream) {
    if (!p || bytes == 0 || bytes > (1u << 30)) return SWF_OK;
    hipLaunchKernelGGL(l2_warm_kernel, dim3(256), dim3(256), 0, stream, static_cast<const char*>(p), (int)bytes, static_cast<int*>(nullptr));
    return check_launch("l2_warm");
}

int launch_window_block(const swf_block_desc& d, const void* packed_x, const void* packed_y, const float* x_in,
                        const float* y_in, float* x_out, float* y_out, int B, int H, int W, hipStream_t stream,
                        const void* next_packed_x, const void* next_packed_y, size_t next_bytes) {
    if (!window_block_supported(d, B, H, W))
        return fail(SWF_ERR_UNSUPPORTED, "window_block: C=%d hidden=%d, %d x %d map, window %d not covered", d.attn.channels, d.hidden, H, W, d.attn.win_h);
    // batch slices of at most 2^31 - 1 bytes per stream map (32-bit buffer offsets inside the kernels); the slice size depends on
    // the map size only, and a slice runs exactly the kernel the whole batch would: rows are bit-identical either way
    const int64_t img = image_bytes(d, H, W);
    const int per = (int)std::min<int64_t>(B, kMaxLaunchBytes / img);
    for (int b0 = 0; b0 < B; b0 += per) {
        const int nb = std::min(per, B - b0);
        const size_t off = (size_t)b0 * (size_t)(img / 4);
        const bool last = b0 + nb >= B;   // only the last slice touches the next block's packed weights (L2 warm-up)
        const void* nx = last ? next_packed_x : nullptr;
        const void* ny = last ? next_packed_y : nullptr;
        int st;
        if (use_win24(d)) st = launch_win24(d, packed_x, packed_y, x_in + off, y_in + off, x_out + off, y_out + off, nb, H, W, stream, nx, ny, next_bytes);
        else if (use_win48(d)) st = launch_win48(d, packed_x, packed_y, x_in + off, y_in + off, x_out + off, y_out + off, nb, H, W, stream, nx, ny, next_bytes);
        else st = launch_win96(d, packed_x, packed_y, x_in + off, y_in + off, x_out + off, y_out + off, nb, H, W, stream, nx, ny, next_bytes);
        if (st != SWF_OK) return st;
    }
    return SWF_OK;
}

}  // namespace swf
