// Level-0 fused BasicBlock (a005:127-145) for gfx950: C = 24, 8 heads x 3 channels, 8x8 windows, hidden 96 (encoder) or 4
// (decoder) — the full-resolution maps, where a block is 2 x 262 144 tokens of 96 bytes and the kernel must look like a
// streaming kernel.  ONE launch = one BasicBlock for both modality streams.
//
// Design: activations never leave registers.  A 256-thread workgroup owns one window; wave w owns the 32 tokens
// [32*(w&1), +32) of stream w>>1 for the WHOLE block (LayerNorms, Q/K/V, attention for its 32 queries, projection, MLP).
// Every contraction runs on the 32x32x16 MFMAs, and every operand that is produced by one MFMA and consumed by the next is
// used where it lands: the C/D map of a 32x32 tile (lane = column, register i of lane half hf = row rho(i, hf)) is also a
// legal B-operand map (lane = column, element j of k-step s = k index), because a contraction may enumerate its k index in
// any order as long as both operands agree — so the weight images are packed with their k columns in rho order
// (pack24_kernel) and LayerNorm output, attention output and hidden activations go from accumulator registers to the next
// MFMA with only the split into bf16 hi/lo parts in between.  No activation tile, no LayerNorm image, no hidden-chunk image
// in LDS; the only LDS traffic is the K / V^T operand images that the two waves of a stream exchange (1 KB fragments,
// lane-linear, conflict-free) and one workgroup barrier per window (images are double-buffered across windows).
//
//  * Q, K: rows are "virtual channels" 4*head + c (c < 3; row 4*head+3 is a zero row), so that the packed f16 pairs of the
//    accumulator are directly the operand fragments of S^T = K.Q^T: k-step s covers heads 4s..4s+3, a head is selected by
//    zeroing the other heads' slots of the Q fragment (2 selects per head).
//  * V is computed in the non-transposed form (tokens in rows): its accumulator registers are the key slots of the A operand
//    of O^T = V^T.P^T.  Virtual channel 4*head+3 of V is the constant 1 (zero weights, bias 1): row 4*head+3 of O^T is the
//    softmax denominator, summed by the MFMA from the same f16-rounded P as the numerator.
//  * The 8 heads of a query block share ONE O^T accumulator: head h's product lands in rows 4h..4h+3, i.e. registers
//    4(h>>1).. of lane half h&1; 4 selects per head copy them out of the per-head product tile.
//  * Biases ride on a constant-one k slot (the padding of the 24 -> 32 channel tile), the residual is the C operand of the
//    projection and fc2 MFMAs: no epilogue arithmetic for either.
//  * The relative-position bias matrix (variant without shift mask) of the wave's (stream, query block) lives in 32
//    registers for the whole launch and is the C operand of the S^T MFMAs.  The shift mask (a001:217-315) of the edge windows
//    is structural in this layout: "last window row" masks whole key tiles, "last window column" masks whole lanes.
//
// Arithmetic (SWF_PREC_FAST): linear layers split-bf16 x3 (fp32-grade) on v_mfma_f32_32x32x16_bf16; Q.K^T and P.V on
// v_mfma_f32_32x32x16_f16 (f16 operands are 8x closer to fp32 than bf16 at the same rate); LayerNorm statistics,
// softmax, ELU, residual stream, accumulators fp32; exp via v_exp_f32 (Wq, bq and the bias matrix carry log2(e)).
#include "kernels_win24.h"

#include <algorithm>

namespace swf {
namespace {

using bf16 = __bf16;
using f16 = _Float16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

#ifndef W24_FOLD_MAX
#define W24_FOLD_MAX 1   // S - max on the matrix pipe (second S^T MFMA with -max on the head's spare k slot)
#endif
#ifndef W24_MED3
#define W24_MED3 1       // ELU in exp2 units through one v_med3 (fc1 packed with log2 e, fc2 with ln 2)
#endif
#ifndef W24_ACT_F16
#define W24_ACT_F16 0    // A/B: activations of the linear layers as ONE f16 (weights split into f16 hi + lo, 2 MFMAs per k-step) instead of split-bf16 x3
#endif
#ifndef W24_WAVES
#define W24_WAVES 3   // resident workgroups per CU = waves per SIMD (register budget 512 / W24_WAVES)
#endif

// row of accumulator register i in lane half hf (C/D map of the 32x32 MFMAs) == k index of element i & 7 of k-step i >> 3
__host__ __device__ constexpr int rho(int i, int hf) { return (i & 3) + 8 * (i >> 2) + 4 * hf; }

template <int HID_>
struct G24 {
    static constexpr int C = 24, HID = HID_, HEADS = 8, D = 3;
    static constexpr bool ONES_H = HID % 32 != 0;                 // a spare hidden row carries the constant 1 (fc2 bias slot)
    static constexpr int NT1 = (HID + (ONES_H ? 1 : 0) + 31) / 32;   // 32-row tiles of fc1
    static constexpr int KU = (HID + (ONES_H ? 1 : 0) + 15) / 16;    // 16-deep k-steps of fc2
    // fragment table: every fragment is 64 lanes x 16 bytes (8 bf16), lane-linear
    static constexpr int F_QKV = 0;                 // [tile q,k,v][k-step 2][hi,lo]
    static constexpr int F_P = 12;                  // [k-step 2][hi,lo]
    static constexpr int F_W1 = 16;                 // [tile NT1][k-step 2][hi,lo]
    static constexpr int F_W2 = F_W1 + 4 * NT1;     // [k-step KU][hi,lo]
    static constexpr int NFRAG = F_W2 + 2 * KU;
    static constexpr size_t p_vec = size_t(NFRAG) * 1024;        // fp32 [lane half 2][64]: ln1 g/b, ln2 g/b, b2 (12 each)
    static constexpr int V_LN1G = 0, V_LN1B = 12, V_LN2G = 24, V_LN2B = 36, V_B2 = 48;
    static constexpr size_t p_bias = p_vec + 2 * 64 * 4;         // fp32 [query block 2][lane 64][key tile 2][reg 16]
    static constexpr size_t p_total = p_bias + size_t(2) * 2 * 16 * 64 * 4;
    // LDS (bytes): K images [buf 2][stream 2][key tile 2][k-step 2] x 1 KB, V^T images [buf 2][stream 2][pv-step 4] x 1 KB, vectors
    static constexpr size_t l_k = 0, l_v = l_k + 16 * 1024, l_vec = l_v + 16 * 1024, l_total = l_vec + 2 * 2 * 64 * 4;
    // 16x16 windows (window24w16_kernel): the bias section holds one S^T tile per key-tile / query-tile distance kt - qb + 7
    // (a tile = two window rows of 16 tokens, so the relative positions of a tile pair depend on that distance only):
    // fp32 [distance 15][lane 64][reg 16]
    static constexpr size_t p_total16 = p_bias + size_t(15) * 64 * 16 * 4;
    // LDS: K images [stream 2][key tile 8][k-step 2] x 1 KB, V^T images [stream 2][pv-step 16] x 1 KB (single-buffered), vectors
    static constexpr size_t l_k16 = 0, l_v16 = 32 * 1024, l_vec16 = 64 * 1024, l_total16 = l_vec16 + 2 * 2 * 64 * 4;
};

struct Win24Args {
    const float* in[2];
    float* out[2];       // half-block modes: a NULL out[s] drops that stream's stores (its waves only feed K / V to the other stream)
    const char* packed[2];
    const char* warm[2];
    int B, H, W, shift, cross, warm_bytes;
    int ntok[2];         // MLP half (W24_MLP): token count of each stream's flat token list
};

// What one launch computes (template parameter MODE of window24_kernel):
//   W24_BLOCK  the whole BasicBlock (a005:127-145)
//   W24_ATTN   x + proj(attention(LN1 ...)) — AddAndLayerNormWithOtherModule around AutoPathWinAtt (a004:29-38, a002:58-82); with
//              RAW: proj(attention(q, k, v)) on un-normalised inputs and no residual — WindowAttention.forward (a001:448-474),
//              stream 0 = the query tensor and the output, stream 1 = the key / value tensor (its waves stop after K / V)
//   W24_MLP    x + fc2(ELU(fc1(LN2 x))) — AddAndLayerNormWithOtherModule around AutoPathMLP (a004:29-38, a003:46-50); with RAW:
//              fc2(ELU(fc1 x)) — AutoPathMLP.forward.  Tokens are a flat list (no windows): 64 per workgroup step and stream
constexpr int W24_BLOCK = 0, W24_ATTN = 1, W24_MLP = 2;

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// acc += a . b over one 16-deep k-step with split-bf16 operands (a = a_hi + a_lo, b = b_hi + b_lo): three MFMAs, small cross
// terms first so they are not absorbed by the large hi.hi partial sums
template <bool WEIGHT_IS_A = true>
__device__ __forceinline__ f32x16 mma3(u32x4 ahi, u32x4 alo, u32x4 bhi, u32x4 blo, f32x16 acc) {
    if constexpr (W24_ACT_F16) {   // the activation operand is a single f16 fragment (its "lo" is unused), the weight f16 hi + lo
        if constexpr (WEIGHT_IS_A) {
            acc = mfma_f16(alo, bhi, acc);
            acc = mfma_f16(ahi, bhi, acc);
        } else {
            acc = mfma_f16(ahi, blo, acc);
            acc = mfma_f16(ahi, bhi, acc);
        }
        return acc;
    }
    acc = mfma_bf16(alo, bhi, acc);
    acc = mfma_bf16(ahi, blo, acc);
    acc = mfma_bf16(ahi, bhi, acc);
    return acc;
}

__device__ __forceinline__ u32x4 pack8_f16(const float* v);
// 8 fp32 values -> one k-step fragment in split-bf16 (hi = bf16(v), lo = bf16(v - hi))
__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& lo) {
    if constexpr (W24_ACT_F16) {
        hi = pack8_f16(v);
        lo = hi;   // unused
        return;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const bf16x2 h = {(bf16)v[2 * p], (bf16)v[2 * p + 1]};
        const unsigned hu = __builtin_bit_cast(unsigned, h);
        const float h0 = __builtin_bit_cast(float, hu << 16), h1 = __builtin_bit_cast(float, hu & 0xffff0000u);
        const bf16x2 l = {(bf16)(v[2 * p] - h0), (bf16)(v[2 * p + 1] - h1)};
        hi[p] = hu;
        lo[p] = __builtin_bit_cast(unsigned, l);
    }
}
__device__ __forceinline__ u32x4 pack8_f16(const float* v) {
    u32x4 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f16x2 h = {(f16)v[2 * p], (f16)v[2 * p + 1]};   // v_cvt_pk_f16_f32 (round to nearest even)
        o[p] = __builtin_bit_cast(unsigned, h);
    }
    return o;
}

// value of the same register in lane l ^ 32, combined with the own value.  v_permlane32_swap exchanges the upper half of its
// first operand with the lower half of its second: with both = v, a = [v_lo, v_lo] and b = [v_hi, v_hi] afterwards.
// (inline asm: hipcc 7.2 folds the builtin's second result into the first; the s_nop covers the VALU-write -> permlane hazard)
__device__ __forceinline__ void halves(float v, float& a, float& b) {
    a = v;
    b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float sum_halves(float v) { float a, b; halves(v, a, b); return a + b; }
__device__ __forceinline__ float max_halves(float v) { float a, b; halves(v, a, b); return __builtin_fmaxf(a, b); }

// The weight fragments are loop-invariant loads: without a fence hipcc hoists them out of the window loop (or to the top of
// an iteration) and spills.  A compiler-only barrier, no instruction.
#define W24_FENCE() asm volatile("" ::: "memory")

// A pointer that is the same in every lane of the wave but derived from the wave index: made provably uniform so that hipcc
// keeps it in SGPRs and addresses fragments as (scalar base + lane offset + immediate) instead of holding a 64-bit per-lane
// address per fragment group in VGPRs across the window loop.
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// LayerNorm (eps 1e-5, biased variance) of the lane's token — 12 of its 24 channels sit in this lane (registers 0..11 of
// `res`), the other 12 in lane l ^ 32 — straight into the split-bf16 B / A operand fragments of the next linear layer.
// Slot 12 of lane half 0 (k index rho(12, 0) = 24) is the constant 1 the packed weights keep their bias on.
__device__ __forceinline__ void layernorm_frags(const f32x16& res, const float* vec, int goff, int boff, u32x4 (&xh)[2], u32x4 (&xl)[2]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += res[i];
    const float mean = sum_halves(s) * (1.0f / 24.0f);
    float d[12], q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        d[i] = res[i] - mean;
        q += d[i] * d[i];
    }
    const float rstd = __builtin_amdgcn_rsqf(sum_halves(q) * (1.0f / 24.0f) + 1e-5f);
    float n[16];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float4 g = *reinterpret_cast<const float4*>(vec + goff + 4 * a);
        const float4 b = *reinterpret_cast<const float4*>(vec + boff + 4 * a);
        n[4 * a + 0] = d[4 * a + 0] * rstd * g.x + b.x;
        n[4 * a + 1] = d[4 * a + 1] * rstd * g.y + b.y;
        n[4 * a + 2] = d[4 * a + 2] * rstd * g.z + b.z;
        n[4 * a + 3] = d[4 * a + 3] * rstd * g.w + b.w;
    }
    n[12] = 1.0f;   // the bias slot (its weight column is zero in lane half 1)
    n[13] = n[14] = n[15] = 0.f;
    split8(n, xh[0], xl[0]);
    split8(n + 8, xh[1], xl[1]);
}

// RAW modes: the un-normalised row as the operand fragments (slot 12 of lane half 0 = the constant 1 of the bias column)
__device__ __forceinline__ void raw_frags(const f32x16& res, u32x4 (&xh)[2], u32x4 (&xl)[2]) {
    float n[16];
#pragma unroll
    for (int i = 0; i < 12; ++i) n[i] = res[i];
    n[12] = 1.0f;
    n[13] = n[14] = n[15] = 0.f;
    split8(n, xh[0], xl[0]);
    split8(n + 8, xh[1], xl[1]);
}

// Attention of one wave: 32 queries x 64 keys x 8 heads.  ksrc / vsrc: the stream's K and V^T operand images in LDS (+ lane);
// qf: the wave's own Q fragments; bias: relative-position bias of (stream, query block), C operand of the S^T MFMAs;
// (with -inf where the shift mask applies).  Returns the O^T accumulator: registers 4a..4a+3 of
// lane half p = channels 0..2 and softmax denominator of head 2a + p.
__device__ __forceinline__ f32x16 attention24(const u32x4* ksrc, const u32x4* vsrc, const u32x4 (&qf)[2], const f32x16 (&bias)[2], bool half1) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 o = zero16;
    // Register budget (three, ideally four waves per SIMD): the K / V^T fragments are read from LDS where they are used
    // instead of being held across the head loop, and a head's P tile is exponentiated, packed and multiplied eight keys at a
    // time, so that at most the bias (32), the score tile (32, shrinking), the product tile (16) and the output tile (16)
    // are live together.
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const int s = h >> 2, sub = (h >> 1) & 1;
        const bool keep = half1 == ((h & 1) != 0);
        f32x16 s0, s1;
        {
            const u32x4 ka0 = ksrc[(0 * 2 + s) * 64], ka1 = ksrc[(1 * 2 + s) * 64];
            // the head's three channels are elements 4*sub .. 4*sub+2 of lane half h & 1; every other slot is zeroed
            u32x4 qm = {0u, 0u, 0u, 0u};
            qm[2 * sub] = keep ? qf[s][2 * sub] : 0u;
            qm[2 * sub + 1] = keep ? qf[s][2 * sub + 1] : 0u;
            s0 = mfma_f16(ka0, qm, bias[0]);   // S^T[key][query] + bias, exp2 units
            s1 = mfma_f16(ka1, qm, bias[1]);
        }
        float mx = max3f(s0[0], s0[1], s1[0]);
        mx = max3f(mx, s1[1], s0[2]);
#pragma unroll
        for (int i = 3; i < 16; i += 2) mx = max3f(mx, s0[i], s0[i + 1 < 16 ? i + 1 : i]);
#pragma unroll
        for (int i = 2; i < 16; i += 2) mx = max3f(mx, s1[i], s1[i + 1]);
        mx = max_halves(mx);   // the other 32 keys of the query sit in lane l ^ 32 (never all masked: a query's own region is not)
        // S - max on the matrix pipe instead of 32 subtractions on the (saturated) vector pipe: the head's spare k slot (virtual
        // channel 4h+3) is 1 in every K row and -max (rounded to f16: softmax is shift-invariant, any per-query constant
        // near the maximum serves) in the Q fragment, and the scores are computed a second time
        if constexpr (W24_FOLD_MAX) {
            const u32x4 ka0 = ksrc[(0 * 2 + s) * 64], ka1 = ksrc[(1 * 2 + s) * 64];
            const f16 nm = (f16)(-mx);
            const unsigned nmb = keep ? (unsigned)__builtin_bit_cast(unsigned short, nm) : 0u;
            u32x4 qm = {0u, 0u, 0u, 0u};
            qm[2 * sub] = keep ? qf[s][2 * sub] : 0u;
            qm[2 * sub + 1] = (keep ? qf[s][2 * sub + 1] : 0u) | (nmb << 16);
            s0 = mfma_f16(ka0, qm, bias[0]);
            s1 = mfma_f16(ka1, qm, bias[1]);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s0[i] -= mx; s1[i] -= mx; }
        }
        // O^T tile of this head = V^T . P^T over the 64 keys; P = exp2(S - max) in f16, one pv-step (16 keys = registers
        // 8s'.. of key tile kt) at a time: the exponentials of step ps+1 issue under the MFMA of step ps
        f32x16 t;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = __builtin_amdgcn_exp2f((ps >> 1) ? s1[8 * (ps & 1) + j] : s0[8 * (ps & 1) + j]);
            const u32x4 pf = pack8_f16(p);
            const u32x4 va = vsrc[ps * 64];
            t = mfma_f16(va, pf, ps == 0 ? zero16 : t);
            __builtin_amdgcn_sched_barrier(0);
        }
        // rows 4h .. 4h+3 (3 channels + denominator) = registers 4(h>>1) .. +3 of lane half h & 1
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * (h >> 1) + j] = keep ? t[4 * (h >> 1) + j] : o[4 * (h >> 1) + j];
        __builtin_amdgcn_sched_barrier(0);   // one head at a time: interleaving heads doubles the live score tiles
    }
    return o;
}

// ---------------------------------------------------------------------------------------------------------------
// WS = window side, 8 or 7 (the reference's default, A000_CONFIG.py:55).  A 7x7 window runs on the same 8x8 token grid: the
// 15 padding tokens (row 7 / column 7) load zeros and store nothing (buffer addressing: an offset beyond the descriptor's
// range reads 0 and drops the store), and as keys they carry -inf in the packed bias matrix, so their probabilities are 0.
// The shift seam of the last window row / column sits at WS - WS/2 = 4 for both sizes: the structural masks are unchanged.
template <int HID, int WS, int MODE = W24_BLOCK, bool RAW = false>
__global__ __launch_bounds__(256, W24_WAVES) void window24_kernel(Win24Args args) {
    using G = G24<HID>;
    static_assert(WS == 7 || WS == 8, "window side");
    static_assert(MODE == W24_BLOCK || MODE == W24_ATTN || MODE == W24_MLP, "mode");
    static_assert(!RAW || MODE != W24_BLOCK, "RAW belongs to the half-block modes");
    constexpr bool ATT = MODE != W24_MLP, MLP = MODE != W24_ATTN;
    __shared__ __attribute__((aligned(16))) char smem[G::l_total];
    u32x4* kimg = reinterpret_cast<u32x4*>(smem + G::l_k);   // [buf][stream][key tile][k-step][lane]
    u32x4* vimg = reinterpret_cast<u32x4*>(smem + G::l_v);   // [buf][stream][pv-step][lane]
    float* lvec = reinterpret_cast<float*>(smem + G::l_vec);   // [stream][lane half][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ws = wave >> 1, qb = wave & 1, r = lane & 31, hf = lane >> 5;
    const int H = args.H, W = args.W, nwx = MLP && !ATT ? 1 : W / WS, nwy = MLP && !ATT ? 1 : H / WS, npi = nwx * nwy;
    // MLP half: a "window" is 64 consecutive tokens of the flat list; the streams may differ in length (single-stream callers split one list)
    const int nwin = ATT ? args.B * npi : (max(args.ntok[0], args.ntok[1]) + 63) / 64;
    const int sh = args.shift ? WS / 2 : 0;
    // (RAW attention: stream 1 is the key / value tensor of stream 0's queries; its waves stop after K / V and stream 0's skip K / V)
    const int kvs = args.cross ? 1 - ws : ws;   // the stream whose attention reads this wave's tokens as keys (a002:67-82)

    for (int i = tid; i < 2 * 2 * 64; i += 256) lvec[i] = reinterpret_cast<const float*>(args.packed[i >> 7] + G::p_vec)[i & 127];
    // Weight fragment f of a stream = 64 lanes x 16 bytes at f * 1024, read through a buffer descriptor: scalar base (made
    // provably wave-uniform), the lane offset as the 32-bit voffset, the fragment offset as soffset — no address arithmetic on
    // the VALU and no per-fragment 64-bit address registers (with plain pointers hipcc precomputed one per fragment ahead of
    // the window loop and spilled 46 registers).  The token rows go the same way.
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(uniform_ptr(args.packed[ws])), 0, (int)G::p_total, 0x00020000);
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(uniform_ptr(args.packed[kvs])), 0, (int)G::p_total, 0x00020000);
    const int act_bytes = ATT ? args.B * H * W * 24 * 4 : args.ntok[ws] * 24 * 4;   // < 2^31 (launch_win24)
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(args.in[ws])), 0, act_bytes, 0x00020000);
    // a NULL output (half-block modes) becomes an empty descriptor: every store is out of range and dropped
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(args.out[ws]), 0, (MODE == W24_BLOCK || args.out[ws]) ? act_bytes : 0, 0x00020000);
    const unsigned loff = (unsigned)lane * 16u;
    auto WF = [&](int f) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, f * 1024, 0)); };   // own stream: Q, proj, MLP
    auto WK = [&](int f) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, loff, f * 1024, 0)); };   // K / V weights
    const float* vec = lvec + (ws * 2 + hf) * 64;
    // relative-position bias of (stream, query block), both key tiles: C operand of the S^T MFMAs for the whole launch
    f32x16 bias[2];
    // resident bias tile of this wave's token half: 128 contiguous bytes per lane, so every offset beyond the lane's own is an
    // instruction immediate (no scalar registers held across the window loop for the reload after an edge window)
    auto load_bias = [&]() {
        const int vo = lane * 128 + qb * (2 * 16 * 64 * 4);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, vo + (kt * 16 + q4 * 4) * 4, (int)G::p_bias, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) bias[kt][q4 * 4 + e] = t[e];
            }
    };
    if constexpr (ATT) load_bias();
    const bool half1 = hf != 0;
    const bool col_masked = half1 != (((r >> 2) & 1) != 0);   // last-window-column variant: this lane's keys lie across the seam
    __syncthreads();

    int it = 0;
    for (int win = blockIdx.x; win < nwin; win += gridDim.x, ++it) {
        W24_FENCE();
        const int b = win / npi, wrem = win - b * npi;
        const int wy = wrem / nwx, wx = wrem - wy * nwx;
        const int buf = it & 1;
        // ---- the lane's token: window row 4qb + (r >> 3), column r & 7; cyclic shift = index arithmetic (a001:442-445) ----
        // (derived from an opaque copy of the lane id inside the loop: kept across the loop these few values were what hipcc
        // chose to spill; the wrap-around as an unsigned min needs no H / W splat registers)
        int lane_w = lane;
        asm volatile("" : "+v"(lane_w));
        const int ty = 4 * qb + ((lane_w >> 3) & 3), tx = lane_w & 7;
        unsigned oy = wy * WS + ty + sh, ox = wx * WS + tx + sh;
        oy = oy < oy - (unsigned)H ? oy : oy - (unsigned)H;
        ox = ox < ox - (unsigned)W ? ox : ox - (unsigned)W;
        // byte offset of the lane's first float4; padding tokens of a 7x7 window point beyond the buffer (reads 0, stores dropped)
        // (MLP half: flat token list, token 64 win + 32 qb + (lane & 31); past the stream's end -> out of range: reads 0, stores dropped)
        const unsigned tokoff = [&]() -> unsigned {
            if constexpr (ATT) {
                return (WS == 8 || (ty < WS && tx < WS)) ? (unsigned)((((b * H + (int)oy) * W + (int)ox) * 24 + 4 * (lane_w >> 5)) * 4) : 0x80000000u;
            } else {
                const int tok = 64 * win + 32 * qb + (lane_w & 31);
                return tok < args.ntok[ws] ? (unsigned)((tok * 24 + 4 * (lane_w >> 5)) * 4) : 0x80000000u;
            }
        }();
        // rows 24..31 of every output tile have zero weights: registers 12..15 stay zero
        auto load_rows = [&](f32x16& dstv) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                // (one bit_cast of the whole vector: hipcc 7.2 narrows the load to ONE dword and splats it when the four lanes of
                // the b128 result are bit_cast element by element)
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, tokoff, 32 * a, 0));
                dstv[4 * a] = v.x; dstv[4 * a + 1] = v.y; dstv[4 * a + 2] = v.z; dstv[4 * a + 3] = v.w;
            }
            dstv[12] = dstv[13] = dstv[14] = dstv[15] = 0.f;
        };
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

        f32x16 res;
        if constexpr (ATT) {
        // ---- LN1, then Q (own stream's weights), K and V (weights of the stream that attends to these tokens) ----
        u32x4 qf[2];
        {
            // the residual rows are NOT kept in registers across the attention phase (16 registers of a 168 budget): they are
            // read again — an L2 hit, the lines were fetched microseconds ago — as the C operand of the projection
            f32x16 x0;
            load_rows(x0);
            u32x4 xh[2], xl[2];
            if constexpr (RAW) raw_frags(x0, xh, xl);
            else layernorm_frags(x0, vec, G::V_LN1G, G::V_LN1B, xh, xl);
            f32x16 acc = zero16;
            if (!(RAW && ws == 1)) {   // (wave-uniform) RAW: the key / value stream has no queries
#pragma unroll
                for (int s = 0; s < 2; ++s) acc = mma3(WF(G::F_QKV + 2 * s), WF(G::F_QKV + 2 * s + 1), xh[s], xl[s], acc);
                float t[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = acc[i];
                qf[0] = pack8_f16(t);
                qf[1] = pack8_f16(t + 8);
            }
            W24_FENCE();
            if (!(RAW && ws == 0)) {   // RAW: the query stream's tokens are nobody's keys
                float t[16];
                acc = zero16;
#pragma unroll
                for (int s = 0; s < 2; ++s) acc = mma3(WK(G::F_QKV + 4 + 2 * s), WK(G::F_QKV + 5 + 2 * s), xh[s], xl[s], acc);
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = acc[i];
                u32x4* kdst = kimg + (((buf * 2 + kvs) * 2 + qb) * 2) * 64 + lane;
                kdst[0] = pack8_f16(t);
                kdst[64] = pack8_f16(t + 8);
                W24_FENCE();
                // V: tokens in rows (A = x fragments, B = weight fragments): register i of lane (channel r, hf) is token rho(i, hf)
                acc = zero16;
#pragma unroll
                for (int s = 0; s < 2; ++s) acc = mma3<false>(xh[s], xl[s], WK(G::F_QKV + 8 + 2 * s), WK(G::F_QKV + 9 + 2 * s), acc);
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = acc[i];
                u32x4* vdst = vimg + ((buf * 2 + kvs) * 4 + 2 * qb) * 64 + lane;
                vdst[0] = pack8_f16(t);
                vdst[64] = pack8_f16(t + 8);
            }
        }
        __syncthreads();   // K / V^T images of both streams complete (the buffers of the window before stay readable)
        if (RAW && ws == 1) continue;   // (after the barrier of this window; the next window's images use the other buffer)

        // ---- attention of the wave's 32 queries, 8 heads ----
        // Shift mask (a001:217-315; the reference ASSIGNS -1e10 to the masked scores, so their probabilities are exactly 0):
        // only the windows of the last window row / column of a shifted block hold two region labels, split at row /
        // column 4 of the window.  In the S^T layout "query and key on different sides of the row seam" is a whole key tile
        // (key tile kt holds window rows 4kt..4kt+3, the wave's queries rows 4qb..4qb+3) and "different sides of the column
        // seam" is a whole lane (lane half hf holds key columns 4hf..4hf+3, the lane's query column is r & 7).  So the mask
        // is -inf added to a whole bias tile (the C operand of the S^T MFMAs) once per edge window — a wave-uniform branch, no
        // work inside the head loop — and the resident bias registers are read again (L2 hit) after such a window.
        f32x16 o;
        {
            const bool rowv = args.shift && wy == nwy - 1, colv = args.shift && wx == nwx - 1;
            const u32x4* ksrc = kimg + ((buf * 2 + ws) * 4) * 64 + lane;
            const u32x4* vsrc = vimg + ((buf * 2 + ws) * 4) * 64 + lane;
            if (rowv || colv) {
                const float pen0 = ((rowv && qb == 1) || (colv && col_masked)) ? -INFINITY : 0.f;
                const float pen1 = ((rowv && qb == 0) || (colv && col_masked)) ? -INFINITY : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) { bias[0][i] += pen0; bias[1][i] += pen1; }
            }
            o = attention24(ksrc, vsrc, qf, bias, half1);
            if (rowv || colv) load_bias();
        }

        // ---- normalise, output projection + bias + residual: res is the C operand ----
        W24_FENCE();
        if constexpr (RAW) res = zero16;
        else load_rows(res);
        {
            float t[16];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float inv = __builtin_amdgcn_rcpf(o[4 * a + 3]);
                t[4 * a] = o[4 * a] * inv; t[4 * a + 1] = o[4 * a + 1] * inv; t[4 * a + 2] = o[4 * a + 2] * inv;
                t[4 * a + 3] = 1.0f;   // slot rho = 4*head + 3: constant one (the projection bias sits on head 0's)
            }
            u32x4 oh[2], ol[2];
            split8(t, oh[0], ol[0]);
            split8(t + 8, oh[1], ol[1]);
            W24_FENCE();
#pragma unroll
            for (int s = 0; s < 2; ++s) res = mma3(WF(G::F_P + 2 * s), WF(G::F_P + 2 * s + 1), oh[s], ol[s], res);
        }
        } else {   // MLP half: the rows as they are
            load_rows(res);
        }

        // ---- LN2, MLP: fc1 tile -> ELU -> split -> two k-steps of fc2 accumulating onto the residual ----
        if constexpr (MLP) {
            u32x4 xh[2], xl[2];
            if constexpr (RAW) { raw_frags(res, xh, xl); res = zero16; }   // AutoPathMLP.forward: no norm, no residual
            else layernorm_frags(res, vec, G::V_LN2G, G::V_LN2B, xh, xl);
#pragma unroll
            for (int tI = 0; tI < G::NT1; ++tI) {
                W24_FENCE();
                f32x16 acc = zero16;
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    acc = mma3(WF(G::F_W1 + 4 * tI + 2 * s), WF(G::F_W1 + 4 * tI + 2 * s + 1), xh[s], xl[s], acc);
                // ELU(alpha = 1) in exp2 units: the packed fc1 weights carry log2(e) (acc = u = v log2 e) and the packed fc2
                // weights ln 2, so the kernel needs h' = ELU(v) log2(e) = u for u > 0, L = log2(e) (2^u - 1) otherwise.  u <= L
                // everywhere (convexity) and L <= 0 exactly when u <= 0, so h' is the median of (u, L, 0): 3 instructions per
                // hidden activation (exp, fma, med3) instead of multiply, exp, add, compare, select.
                float e[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float u = acc[i];
                    if constexpr (W24_MED3) {
                        const float L = __builtin_fmaf(__builtin_amdgcn_exp2f(u), kLog2e, -kLog2e);
                        e[i] = __builtin_amdgcn_fmed3f(u, L, 0.f);
                    } else {
                        e[i] = u > 0.f ? u : __builtin_amdgcn_exp2f(u * kLog2e) - 1.0f;
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int u = 2 * tI + s2;
                    if (u < G::KU) {
                        u32x4 hh, hl;
                        split8(e + 8 * s2, hh, hl);
                        W24_FENCE();
                        res = mma3(WF(G::F_W2 + 2 * u), WF(G::F_W2 + 2 * u + 1), hh, hl, res);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // one hidden tile at a time
            }
            if constexpr (!G::ONES_H) {   // no spare hidden row to carry the fc2 bias
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float4 b2 = *reinterpret_cast<const float4*>(vec + G::V_B2 + 4 * a);
                    res[4 * a] += b2.x; res[4 * a + 1] += b2.y; res[4 * a + 2] += b2.z; res[4 * a + 3] += b2.w;
                }
            }
        }

        // ---- store the own rows (un-shift = the same index map) ----
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const f32x4 v = {res[4 * a], res[4 * a + 1], res[4 * a + 2], res[4 * a + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, tokoff, 32 * a, 0);
        }
    }

    // ---- L2 warm-up of the next block's packed weights (cold since the previous forward; see kernels_window.hip) ----
    if (args.warm[0]) {
        const int nsl = max(1, (int)gridDim.x / 8), sl = ((int)blockIdx.x / 8) % nsl;
        const int lines = (args.warm_bytes + 127) / 128;
        const int per = (lines + nsl - 1) / nsl, l0 = sl * per, l1 = min(lines, l0 + per);
        unsigned acc = 0;
        for (int s2 = 0; s2 < 2; ++s2)
            for (int l = l0 + tid; l < l1; l += 256) acc ^= *reinterpret_cast<const unsigned*>(args.warm[s2] + (size_t)l * 128);
        if (acc == 0x9e3779b9u && args.B < 0) args.out[0][0] = 0.f;   // never true: keeps the loads alive
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 16x16 windows (BASELINE config 5): the same register-resident pipeline per 32-token tile, attention over 256 keys.
//
// A window is 8 token tiles of 32 (tile j = window rows 2j, 2j+1).  Wave (stream w >> 1, half w & 1) owns tiles 4*half .. +3 of
// its stream.  Phase A: LN1 + Q/K/V of its four tiles — the 12 weight fragments are fetched ONCE per window and serve all
// four —, Q fragments stay in registers, K / V^T images of all 256 keys go to LDS (64 KB for both streams; two workgroups
// per CU).  One barrier.  Phase B, per tile: attention with an online softmax over four chunks of 64 keys, then projection,
// LN2, MLP and the store exactly as in window24_kernel.  A second barrier frees the images for the next window.
//
// Online softmax in this layout: a chunk is the 8x8 kernel's whole attention (S^T with the bias tile as C operand, row maximum,
// S^T again with -max on the spare k slot, exp2, V^T.P^T with the constant-one channel), followed by o = o * 2^(m_old - m_new)
// + t on the four accumulator rows of the head.  The shift that enters the second S^T pass is -max ROUNDED TO f16, so the
// running maximum is kept as that rounded value: the rescale factor then matches what earlier chunks were shifted by.
// Shift masks stay structural: the row seam (window row 8) separates key tiles 0..3 from 4..7 — a masked chunk is skipped —,
// the column seam (column 8) is bit 2 of the accumulator register index against bit 3 of the lane's query column: -inf added
// to those bias registers of the chunk's two tiles.
template <int HID>
__global__ __launch_bounds__(256, 2) void window24w16_kernel(Win24Args args) {
    using G = G24<HID>;
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    u32x4* kimg = reinterpret_cast<u32x4*>(smem16 + G::l_k16);   // [stream][key tile 8][k-step 2][lane]
    u32x4* vimg = reinterpret_cast<u32x4*>(smem16 + G::l_v16);   // [stream][pv-step 16][lane]
    float* lvec = reinterpret_cast<float*>(smem16 + G::l_vec16);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ws = wave >> 1, pw = wave & 1, r = lane & 31, hf = lane >> 5;
    const int H = args.H, W = args.W, nwx = W / 16, nwy = H / 16, npi = nwx * nwy;
    const int nwin = args.B * npi;
    const int sh = args.shift ? 8 : 0;
    const int kvs = args.cross ? 1 - ws : ws;

    for (int i = tid; i < 2 * 2 * 64; i += 256) lvec[i] = reinterpret_cast<const float*>(args.packed[i >> 7] + G::p_vec)[i & 127];
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(uniform_ptr(args.packed[ws])), 0, (int)G::p_total16, 0x00020000);
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(uniform_ptr(args.packed[kvs])), 0, (int)G::p_total16, 0x00020000);
    const int act_bytes = args.B * H * W * 24 * 4;   // < 2^31 (launch_win24)
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(args.in[ws])), 0, act_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(args.out[ws]), 0, act_bytes, 0x00020000);
    const unsigned loff = (unsigned)lane * 16u;
    auto WF = [&](int f) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, f * 1024, 0)); };
    auto WK = [&](int f) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, loff, f * 1024, 0)); };
    const float* vec = lvec + (ws * 2 + hf) * 64;
    const bool half1 = hf != 0;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int win = blockIdx.x; win < nwin; win += gridDim.x) {
        W24_FENCE();
        const int b = win / npi, wrem = win - b * npi;
        const int wy = wrem / nwx, wx = wrem - wy * nwx;
        // byte offset of the lane's first float4 of its token in tile j: window row 2j + (r >> 4), column r & 15; the cyclic
        // shift is index arithmetic (a001:442-445)
        auto tokoff_of = [&](int j) {
            int lane_w = lane;
            asm volatile("" : "+v"(lane_w));
            unsigned oy = wy * 16 + 2 * j + ((lane_w >> 4) & 1) + sh, ox = wx * 16 + (lane_w & 15) + sh;
            oy = oy < oy - (unsigned)H ? oy : oy - (unsigned)H;
            ox = ox < ox - (unsigned)W ? ox : ox - (unsigned)W;
            return (unsigned)((((b * H + (int)oy) * W + (int)ox) * 24 + 4 * (lane_w >> 5)) * 4);
        };
        auto load_rows = [&](f32x16& dstv, unsigned tokoff) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, tokoff, 32 * a, 0));
                dstv[4 * a] = v.x; dstv[4 * a + 1] = v.y; dstv[4 * a + 2] = v.z; dstv[4 * a + 3] = v.w;
            }
            dstv[12] = dstv[13] = dstv[14] = dstv[15] = 0.f;
        };

        // ---- phase A: LN1 + Q/K/V of the wave's four tiles; the weight fragments are fetched once ----
        u32x4 qf[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) { qf[q][0] = u32x4{0u, 0u, 0u, 0u}; qf[q][1] = u32x4{0u, 0u, 0u, 0u}; }
        {
            u32x4 wq[4], wk[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { wq[i] = WF(G::F_QKV + i); wk[i] = WK(G::F_QKV + 4 + i); wv[i] = WK(G::F_QKV + 8 + i); }
#pragma unroll 1
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * pw + jj;
                f32x16 x0;
                load_rows(x0, tokoff_of(j));
                u32x4 xh[2], xl[2];
                layernorm_frags(x0, vec, G::V_LN1G, G::V_LN1B, xh, xl);
                f32x16 acc = zero16;
#pragma unroll
                for (int s = 0; s < 2; ++s) acc = mma3(wq[2 * s], wq[2 * s + 1], xh[s], xl[s], acc);
                float t[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = acc[i];
                // Q fragments rotate through the register array (a runtime tile loop cannot index registers): after four
                // rounds qf[0] is tile 4*pw, the order phase B consumes them in
#pragma unroll
                for (int q = 0; q < 3; ++q) { qf[q][0] = qf[q + 1][0]; qf[q][1] = qf[q + 1][1]; }
                qf[3][0] = pack8_f16(t);
                qf[3][1] = pack8_f16(t + 8);
                acc = zero16;
#pragma unroll
                for (int s = 0; s < 2; ++s) acc = mma3(wk[2 * s], wk[2 * s + 1], xh[s], xl[s], acc);
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = acc[i];
                u32x4* kdst = kimg + ((kvs * 8 + j) * 2) * 64 + lane;
                kdst[0] = pack8_f16(t);
                kdst[64] = pack8_f16(t + 8);
                acc = zero16;
#pragma unroll
                for (int s = 0; s < 2; ++s) acc = mma3<false>(xh[s], xl[s], wv[2 * s], wv[2 * s + 1], acc);
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i] = acc[i];
                u32x4* vdst = vimg + (kvs * 16 + 2 * j) * 64 + lane;
                vdst[0] = pack8_f16(t);
                vdst[64] = pack8_f16(t + 8);
            }
        }
        __syncthreads();   // K / V^T images of all 256 keys of both streams are complete

        // ---- phase B: per tile attention over 256 keys, projection, LN2, MLP, store ----
        const bool rowv = args.shift && wy == nwy - 1, colv = args.shift && wx == nwx - 1;
        const u32x4* ksrc = kimg + (ws * 16) * 64 + lane;
        const u32x4* vsrc = vimg + (ws * 16) * 64 + lane;
#pragma unroll 1
        for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * pw + jj;   // query tile
            const unsigned tokoff = tokoff_of(j);
            f32x16 o = zero16;
            float mrun[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // running (f16-rounded) maximum: register a, lane half p = head 2a + p
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
                if (rowv && ((j < 4) != (c < 2))) continue;   // the chunk's keys lie across the row seam: probabilities exactly 0
                // bias tiles of key tiles 2c, 2c+1 against query tile j: table entry kt - j + 7
                f32x16 bias[2];
                {
                    const int d0 = __builtin_amdgcn_readfirstlane(2 * c - j + 7);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const f32x4 tb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 64 + q4 * 16, (int)G::p_bias + (d0 + kt) * 4096, 0));
#pragma unroll
                            for (int e = 0; e < 4; ++e) bias[kt][q4 * 4 + e] = tb[e];
                        }
                    if (colv) {   // key column (register bit 2) and query column (lane bit 3) on different sides of column 8
                        const bool qhi = (r & 8) != 0;
                        const float pen_lo = qhi ? -INFINITY : 0.f, pen_hi = qhi ? 0.f : -INFINITY;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float pen = ((i >> 2) & 1) ? pen_hi : pen_lo;
                            bias[0][i] += pen; bias[1][i] += pen;
                        }
                    }
                }
                const u32x4* kc = ksrc + (2 * c) * 2 * 64;
                const u32x4* vc = vsrc + (4 * c) * 64;
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const int s = h >> 2, sub = (h >> 1) & 1;
                    const bool keep = half1 == ((h & 1) != 0);
                    const u32x4 ka0 = kc[(0 * 2 + s) * 64], ka1 = kc[(1 * 2 + s) * 64];
                    u32x4 qm = {0u, 0u, 0u, 0u};
                    qm[2 * sub] = keep ? qf[0][s][2 * sub] : 0u;
                    qm[2 * sub + 1] = keep ? qf[0][s][2 * sub + 1] : 0u;
                    f32x16 s0 = mfma_f16(ka0, qm, bias[0]);
                    f32x16 s1 = mfma_f16(ka1, qm, bias[1]);
                    float mx = max3f(s0[0], s0[1], s1[0]);
                    mx = max3f(mx, s1[1], s0[2]);
#pragma unroll
                    for (int i = 3; i < 16; i += 2) mx = max3f(mx, s0[i], s0[i + 1 < 16 ? i + 1 : i]);
#pragma unroll
                    for (int i = 2; i < 16; i += 2) mx = max3f(mx, s1[i], s1[i + 1]);
                    mx = max_halves(mx);
                    const float mold = mrun[h >> 1];
                    const f16 nm = (f16)(-__builtin_fmaxf(mold, mx));
                    const float mnew = -(float)nm;                      // the shift the second pass really applies
                    const float alpha = __builtin_amdgcn_exp2f(mold - mnew);   // first chunk: 2^-inf = 0
                    mrun[h >> 1] = keep ? mnew : mold;
                    const unsigned nmb = keep ? (unsigned)__builtin_bit_cast(unsigned short, nm) : 0u;
                    qm[2 * sub + 1] |= nmb << 16;
                    s0 = mfma_f16(ka0, qm, bias[0]);
                    s1 = mfma_f16(ka1, qm, bias[1]);
                    f32x16 t;
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        float pe[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) pe[e] = __builtin_amdgcn_exp2f((ps >> 1) ? s1[8 * (ps & 1) + e] : s0[8 * (ps & 1) + e]);
                        const u32x4 pf = pack8_f16(pe);
                        const u32x4 va = vc[ps * 64];
                        t = mfma_f16(va, pf, ps == 0 ? zero16 : t);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * (h >> 1) + e;
                        o[i] = keep ? __builtin_fmaf(o[i], alpha, t[i]) : o[i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the next tile's Q fragments move up
#pragma unroll
            for (int q = 0; q < 3; ++q) { qf[q][0] = qf[q + 1][0]; qf[q][1] = qf[q + 1][1]; }

            // ---- normalise, output projection + bias + residual ----
            f32x16 res;
            W24_FENCE();
            load_rows(res, tokoff);
            {
                float t[16];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float inv = __builtin_amdgcn_rcpf(o[4 * a + 3]);
                    t[4 * a] = o[4 * a] * inv; t[4 * a + 1] = o[4 * a + 1] * inv; t[4 * a + 2] = o[4 * a + 2] * inv;
                    t[4 * a + 3] = 1.0f;
                }
                u32x4 oh[2], ol[2];
                split8(t, oh[0], ol[0]);
                split8(t + 8, oh[1], ol[1]);
                W24_FENCE();
#pragma unroll
                for (int s = 0; s < 2; ++s) res = mma3(WF(G::F_P + 2 * s), WF(G::F_P + 2 * s + 1), oh[s], ol[s], res);
            }
            // ---- LN2, MLP ----
            {
                u32x4 xh[2], xl[2];
                layernorm_frags(res, vec, G::V_LN2G, G::V_LN2B, xh, xl);
#pragma unroll
                for (int tI = 0; tI < G::NT1; ++tI) {
                    W24_FENCE();
                    f32x16 acc = zero16;
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        acc = mma3(WF(G::F_W1 + 4 * tI + 2 * s), WF(G::F_W1 + 4 * tI + 2 * s + 1), xh[s], xl[s], acc);
                    float e[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float u = acc[i];
                        if constexpr (W24_MED3) {
                            const float L = __builtin_fmaf(__builtin_amdgcn_exp2f(u), kLog2e, -kLog2e);
                            e[i] = __builtin_amdgcn_fmed3f(u, L, 0.f);
                        } else {
                            e[i] = u > 0.f ? u : __builtin_amdgcn_exp2f(u * kLog2e) - 1.0f;
                        }
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int u = 2 * tI + s2;
                        if (u < G::KU) {
                            u32x4 hh, hl;
                            split8(e + 8 * s2, hh, hl);
                            W24_FENCE();
                            res = mma3(WF(G::F_W2 + 2 * u), WF(G::F_W2 + 2 * u + 1), hh, hl, res);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!G::ONES_H) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const float4 b2 = *reinterpret_cast<const float4*>(vec + G::V_B2 + 4 * a);
                        res[4 * a] += b2.x; res[4 * a + 1] += b2.y; res[4 * a + 2] += b2.z; res[4 * a + 3] += b2.w;
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const f32x4 v = {res[4 * a], res[4 * a + 1], res[4 * a + 2], res[4 * a + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, tokoff, 32 * a, 0);
            }
        }
        __syncthreads();   // every wave is done with the K / V^T images: the next window may overwrite them
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight packing: fp32 nn.Parameter tensors -> fragment-major split-bf16 images with their k columns in rho order
// ---------------------------------------------------------------------------------------------------------------
struct Pack24Args {
    swf_block_stream_params p[2];
    char* dst[2];
    int ws;   // window side (7 or 8)
};

template <int HID>
__global__ __launch_bounds__(256) void pack24_kernel(Pack24Args a) {
    using G = G24<HID>;
    const int s = blockIdx.y;
    const swf_block_stream_params& p = a.p[s];
    char* dst = a.dst[s];
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    const float qscale = kLog2e / sqrtf(3.0f);   // d^-0.5 (a001:32-34) and exp -> exp2
    // (the half-block entries pack only the half they run: a missing layer packs as zeros)
    auto lin = [](const swf_linear& l, int n, int k, int ld) { return l.weight ? l.weight[n * ld + k] : 0.f; };
    auto bia = [](const swf_linear& l, int n) { return (l.weight && l.bias) ? l.bias[n] : 0.f; };

    for (int idx = gtid; idx < G::NFRAG * 512; idx += gsz) {
        const int f = idx >> 9, lane = (idx >> 3) & 63, e = idx & 7;
        const int r = lane & 31, hf = lane >> 5;
        float val = 0.f;
        int hl;
        if (f < G::F_P) {   // Q / K / V: row (A) or column (B) r = virtual channel 4*head + c; k = input channel in rho order, slot 24 = bias
            const int t = f >> 2, st = (f >> 1) & 1;
            hl = f & 1;
            const int k = rho(8 * st + e, hf), head = r >> 2, c = r & 3;
            const swf_linear& l = t == 0 ? p.attn.q : t == 1 ? p.attn.k : p.attn.v;
            if (c < 3) {
                const int row = 3 * head + c;
                val = k < 24 ? lin(l, row, k, 24) : (k == 24 ? bia(l, row) : 0.f);
                if (t == 0) val *= qscale;
            } else if (t != 0 && k == 24) {
                // virtual channel 4*head+3: the constant 1 (zero weights, bias 1).  V: its O^T row is the softmax denominator.
                // K: the k slot on which the Q fragment carries -max (attention24).  Q: stays zero.
                val = 1.0f;
            }
        } else if (f < G::F_W1) {   // projection: row r = output channel; k = virtual channel 4*head + c of O, slot 3 = bias
            const int st = ((f - G::F_P) >> 1) & 1;
            hl = f & 1;
            const int k = rho(8 * st + e, hf), head = k >> 2, c = k & 3;
            if (r < 24) val = c < 3 ? lin(p.attn.proj, r, 3 * head + c, 24) : (k == 3 ? bia(p.attn.proj, r) : 0.f);
        } else if (f < G::F_W2) {   // fc1: row = hidden unit; k = input channel in rho order, slot 24 = bias
            const int g = f - G::F_W1, t = g >> 2, st = (g >> 1) & 1;
            hl = g & 1;
            const int k = rho(8 * st + e, hf), hid = 32 * t + r;
            if (hid < HID) val = (k < 24 ? lin(p.fc1, hid, k, 24) : (k == 24 ? bia(p.fc1, hid) : 0.f)) * (W24_MED3 ? kLog2e : 1.0f);   // exp2 units (ELU in the kernel)
            else if (G::ONES_H && hid == HID && k == 24) val = 1.0f;   // u = 1 -> h' = 1: the constant the fc2 bias rides on
        } else {   // fc2: row r = output channel; k-step u covers hidden units 32(u>>1) + rho(8(u&1) + e, hf)
            const int g = f - G::F_W2, u = g >> 1;
            hl = g & 1;
            const int hid = 32 * (u >> 1) + rho(8 * (u & 1) + e, hf);
            if (r < 24) val = hid < HID ? lin(p.fc2, r, hid, HID) * (W24_MED3 ? kLn2 : 1.0f) : ((G::ONES_H && hid == HID) ? bia(p.fc2, r) : 0.f);   // h' = ELU log2(e)
        }
        if constexpr (W24_ACT_F16) {
            const f16 hi = (f16)val;
            reinterpret_cast<f16*>(dst)[idx] = hl ? (f16)(val - (float)hi) : hi;
        } else {
            const bf16 hi = (bf16)val;
            reinterpret_cast<bf16*>(dst)[idx] = hl ? (bf16)(val - (float)hi) : hi;
        }
    }
    float* vec = reinterpret_cast<float*>(dst + G::p_vec);
    for (int i = gtid; i < 128; i += gsz) {
        const int hf = i >> 6, j = i & 63, which = j / 12, k = j % 12;
        const int c = 8 * (k >> 2) + 4 * hf + (k & 3);
        float v = 0.f;
        if (which == 0) v = p.ln1.gamma ? p.ln1.gamma[c] : 1.f;
        else if (which == 1) v = p.ln1.beta ? p.ln1.beta[c] : 0.f;
        else if (which == 2) v = p.ln2.gamma ? p.ln2.gamma[c] : 1.f;
        else if (which == 3) v = p.ln2.beta ? p.ln2.beta[c] : 0.f;
        else if (which == 4) v = (p.fc2.weight && p.fc2.bias) ? p.fc2.bias[c] : 0.f;
        vec[i] = v;
    }
    // relative-position bias (a001:113-144), exp2 units, the S^T accumulator registers of each lane: [query block][lane][key tile][register]
    float* bm = reinterpret_cast<float*>(dst + G::p_bias);
    if (!p.attn.bias_table) return;   // MLP half: no attention, the bias section is never read
    if (a.ws == 16) {
        // 16x16 windows: one S^T tile per distance d = kt - qb + 7 between the key tile and the query tile (a tile = two window
        // rows): [d][lane][reg], key = rho(reg, lane half), query = lane & 31, row = index >> 4, column = index & 15
        for (int i = gtid; i < 15 * 64 * 16; i += gsz) {
            const int reg = i & 15, lane = (i >> 4) & 63, d = i >> 10;
            const int key = rho(reg, lane >> 5), q = lane & 31;
            const int dy = 2 * (d - 7) + (key >> 4) - (q >> 4), dx = (key & 15) - (q & 15);
            bm[i] = (dy >= -15 && dy <= 15) ? p.attn.bias_table[(dy + 15) * 31 + (dx + 15)] * kLog2e : 0.f;
        }
        return;
    }
    for (int i = gtid; i < 2 * 2 * 16 * 64; i += gsz) {
        const int reg = i & 15, kt = (i >> 4) & 1, lane = (i >> 5) & 63, qb = i >> 11;
        const int key = 32 * kt + rho(reg, lane >> 5), q = 32 * qb + (lane & 31);
        const int ky = key >> 3, kx = key & 7, qy = q >> 3, qx = q & 7, ws = a.ws, tw = 2 * ws - 1;
        float v = 0.f;
        if (ky >= ws || kx >= ws) v = -INFINITY;   // padding token of a 7x7 window as key: probability 0
        else if (qy < ws && qx < ws) v = p.attn.bias_table[(ky - qy + ws - 1) * tw + (kx - qx + ws - 1)] * kLog2e;
        bm[i] = v;
    }
}

int num_cus24() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

}  // namespace

bool win24_supported(const swf_block_desc& d) {
    return d.attn.channels == 24 && d.attn.heads == 8 && d.attn.head_dim == 3 && d.attn.win_h == d.attn.win_w &&
           (d.attn.win_h == 8 || d.attn.win_h == 7 || d.attn.win_h == 16) && (d.hidden == 96 || d.hidden == 4);
}

size_t win24_packed_bytes(const swf_block_desc& d) {
    if (!win24_supported(d)) return 0;
    if (d.attn.win_h == 16) return align_up(d.hidden == 96 ? G24<96>::p_total16 : G24<4>::p_total16, 256);
    return align_up(d.hidden == 96 ? G24<96>::p_total : G24<4>::p_total, 256);
}

int pack_win24(const swf_block_desc& d, const swf_block_stream_params& px, const swf_block_stream_params& py, void* packed_x,
               void* packed_y, hipStream_t stream) {
    if (!win24_supported(d)) return fail(SWF_ERR_UNSUPPORTED, "pack_win24: shape not covered");
    Pack24Args a;
    a.p[0] = px; a.p[1] = py;
    a.dst[0] = static_cast<char*>(packed_x); a.dst[1] = static_cast<char*>(packed_y);
    a.ws = d.attn.win_h;
    if (d.hidden == 96) hipLaunchKernelGGL((pack24_kernel<96>), dim3(32, 2), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((pack24_kernel<4>), dim3(32, 2), dim3(256), 0, stream, a);
    return check_launch("pack_win24");
}

size_t win24_half_packed_bytes(int channels, int hidden) {
    if (channels != 24 || (hidden != 96 && hidden != 4)) return 0;
    return align_up(hidden == 96 ? G24<96>::p_total : G24<4>::p_total, 256);
}

// Half-block launches (8x8 / 7x7 windows).  mode W24_ATTN: x_out = x + proj(attention(LN1 ...)) for both streams (raw = 0), or
// out = proj(attention(q, kv, kv)) with q = x_in, kv = y_in, y_out = NULL (raw = 1; packed_y = packed_x).  mode W24_MLP: tokens as
// flat lists of ntok_x / ntok_y rows (H, W ignored).  A NULL output drops that stream's stores.
int launch_win24_half(const swf_block_desc& d, int mode, int raw, const void* packed_x, const void* packed_y, const float* x_in,
                      const float* y_in, float* x_out, float* y_out, int B, int H, int W, int ntok_x, int ntok_y, hipStream_t stream) {
    const int wsd = d.attn.win_h;
    if (mode != W24_ATTN && mode != W24_MLP) return fail(SWF_ERR_UNSUPPORTED, "win24_half: mode %d", mode);
    if (d.attn.channels != 24 || (d.hidden != 96 && d.hidden != 4)) return fail(SWF_ERR_UNSUPPORTED, "win24_half: shape not covered");
    Win24Args a{};
    a.in[0] = x_in; a.in[1] = y_in; a.out[0] = x_out; a.out[1] = y_out;
    a.packed[0] = static_cast<const char*>(packed_x); a.packed[1] = static_cast<const char*>(packed_y);
    a.B = B; a.H = H; a.W = W; a.shift = d.attn.shift; a.cross = d.cross; a.ntok[0] = ntok_x; a.ntok[1] = ntok_y;
    int nwin;
    if (mode == W24_ATTN) {
        if (!win24_supported(d) || wsd == 16 || H % wsd || W % wsd) return fail(SWF_ERR_UNSUPPORTED, "win24_half: shape not covered");
        if ((int64_t)B * H * W * 24 * 4 >= (int64_t(1) << 31)) return fail(SWF_ERR_UNSUPPORTED, "win24_half: map exceeds the 2 GB buffer window");
        nwin = B * (H / wsd) * (W / wsd);
    } else {
        if ((int64_t)std::max(ntok_x, ntok_y) * 24 * 4 >= (int64_t(1) << 31) || ntok_x <= 0) return fail(SWF_ERR_UNSUPPORTED, "win24_half: token count");
        nwin = (std::max(ntok_x, ntok_y) + 63) / 64;
    }
    const dim3 grid(std::min(nwin, W24_WAVES * num_cus24())), blk(256);
#define W24_LAUNCH(HID_, WS_, MODE_, RAW_) hipLaunchKernelGGL((window24_kernel<HID_, WS_, MODE_, RAW_>), grid, blk, 0, stream, a)
    if (mode == W24_ATTN) {   // the MLP geometry is irrelevant: the hidden-96 image layout serves
        if (wsd == 8) { if (raw) W24_LAUNCH(96, 8, W24_ATTN, true); else W24_LAUNCH(96, 8, W24_ATTN, false); }
        else { if (raw) W24_LAUNCH(96, 7, W24_ATTN, true); else W24_LAUNCH(96, 7, W24_ATTN, false); }
    } else if (d.hidden == 96) {
        if (raw) W24_LAUNCH(96, 8, W24_MLP, true); else W24_LAUNCH(96, 8, W24_MLP, false);
    } else {
        if (raw) W24_LAUNCH(4, 8, W24_MLP, true); else W24_LAUNCH(4, 8, W24_MLP, false);
    }
#undef W24_LAUNCH
    return check_launch("window24 (half block)");
}

int launch_win24(const swf_block_desc& d, const void* packed_x, const void* packed_y, const float* x_in, const float* y_in,
                 float* x_out, float* y_out, int B, int H, int W, hipStream_t stream, const void* next_packed_x,
                 const void* next_packed_y, size_t next_bytes) {
    const int wsd = d.attn.win_h;
    if (!win24_supported(d) || H % wsd || W % wsd) return fail(SWF_ERR_UNSUPPORTED, "win24: shape not covered");
    if ((int64_t)B * H * W * 24 * 4 >= (int64_t(1) << 31)) return fail(SWF_ERR_UNSUPPORTED, "win24: a stream of %d x %d x %d tokens exceeds the 2 GB buffer window", B, H, W);
    Win24Args a;
    a.in[0] = x_in; a.in[1] = y_in; a.out[0] = x_out; a.out[1] = y_out;
    a.packed[0] = static_cast<const char*>(packed_x); a.packed[1] = static_cast<const char*>(packed_y);
    a.warm[0] = static_cast<const char*>(next_packed_x); a.warm[1] = static_cast<const char*>(next_packed_y);
    if (!a.warm[1]) a.warm[0] = nullptr;
    a.warm_bytes = (int)(next_bytes ? next_bytes : win24_packed_bytes(d));
    a.B = B; a.H = H; a.W = W; a.shift = d.attn.shift; a.cross = d.cross;
    const int nwin = B * (H / wsd) * (W / wsd);
    if (wsd == 16) {   // 65 KB of LDS per workgroup: dynamic allocation, two workgroups per CU
        constexpr int lds = (int)G24<96>::l_total16;
        static hipError_t attr_err = [] {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&window24w16_kernel<96>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            return e != hipSuccess ? e : hipFuncSetAttribute(reinterpret_cast<const void*>(&window24w16_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        }();
        if (attr_err != hipSuccess) return fail(SWF_ERR_HIP, "hipFuncSetAttribute(window24w16): %s", hipGetErrorString(attr_err));
        const int grid16 = std::min(nwin, 2 * num_cus24());
        if (d.hidden == 96) hipLaunchKernelGGL((window24w16_kernel<96>), dim3(grid16), dim3(256), lds, stream, a);
        else hipLaunchKernelGGL((window24w16_kernel<4>), dim3(grid16), dim3(256), lds, stream, a);
        return check_launch("window24w16");
    }
    const int grid = std::min(nwin, W24_WAVES * num_cus24());   // resident workgroups per CU (register-limited)
    if (wsd == 8) {
        if (d.hidden == 96) hipLaunchKernelGGL((window24_kernel<96, 8>), dim3(grid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((window24_kernel<4, 8>), dim3(grid), dim3(256), 0, stream, a);
    } else {
        if (d.hidden == 96) hipLaunchKernelGGL((window24_kernel<96, 7>), dim3(grid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((window24_kernel<4, 7>), dim3(grid), dim3(256), 0, stream, a);
    }
    return check_launch("window24");
}

}  // namespace swf
