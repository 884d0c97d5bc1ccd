// Deep-level attention half in one launch (levels with C = 192, 8 heads of 24, 8x8 windows): the Q/K/V projections of a
// window and its attention run in the workgroup that owns (window, stream, group of four heads) — 512 threads, wave
// (head, 32-token half).  Replaces the Q/K/V GEMM launch + the attention-core launch and the 16-bit Q/K/V round trip between them.
//
//  * The window's LayerNorm rows (split-bf16 planes written by the LayerNorm / MLP-reduce kernels) are staged once into LDS,
//    row-major with a 400-byte stride (25 16-byte slots: every ds_read_b128 lane group of an operand fragment hits 16 different
//    slots).  In a cross-attention block a second tile holds the other stream's rows (K and V read those, a002:67-82).
//  * A wave computes the Q, K and V tiles of ITS head for ITS 32 tokens: 3 accumulator tiles, K = C in 16-deep steps, split-bf16
//    x3 on v_mfma_f32_32x32x16_bf16.  Its weight fragments come straight from L2 (packed fragment-major in per-head virtual
//    channel order, rows 24..31 of a head's 32-row tile are padding), prefetched one k-step ahead in registers; nothing is
//    shared with other waves, so the k loop has no barrier.
//  * From there on it is the level-0 kernel's attention (kernels_win24.hip): accumulator registers are packed to f16 and used
//    where they land — Q as the B operand of S^T = K.Q^T, K / V^T through 1-KB lane-linear LDS images exchanged between the two
//    waves of a head (one workgroup barrier), the relative-position bias as the C operand, S - max on the matrix pipe through
//    the spare k slot (virtual channel 24: K carries 1 there, Q carries -max), the softmax denominator from V's constant-one
//    channel 24, shift masks as whole key tiles / whole lanes.
//  * O^T leaves as split-bf16 planes [token][8 * 24], the projection GEMM's input format.
#include "kernels_qkvattn.h"

#include <algorithm>

namespace swf {
namespace {

using bf16 = __bf16;
using f16 = _Float16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr float kLog2e = 1.4426950408889634f;

__host__ __device__ constexpr int rho(int i, int hf) { return (i & 3) + 8 * (i >> 2) + 4 * hf; }

template <int C_>
struct QA {
    static constexpr int C = C_, D = 24, HEADS = 8, HG = 4, KS = C / 16;
    static constexpr int NFRAG = HEADS * 3 * KS * 2;                    // [head][q,k,v][k-step][hi,lo] x 1 KB
    static constexpr size_t p_bvec = size_t(NFRAG) * 1024;              // fp32 [head][q,k,v][32]: bias in virtual-channel order
    static constexpr size_t p_bias = p_bvec + size_t(HEADS) * 3 * 32 * 4;   // fp32 [query tile 2][key tile 2][reg 16][lane 64]
    static constexpr size_t p_proj = p_bias + size_t(2) * 2 * 16 * 64 * 4;   // output projection as B fragments: [head group][32-channel tile][head in group][k-step 2][hi,lo] x 1 KB
    static constexpr int NT = C / 32, NPF = (HEADS / HG) * NT * HG * 2 * 2;
    static constexpr size_t p_total = p_proj + size_t(NPF) * 1024;
    static constexpr int XS = C * 2 + 16;                               // LDS row stride (bytes) of a staged plane
    static constexpr size_t l_plane = size_t(64) * XS, l_tile = 2 * l_plane;
    static constexpr size_t l_xq = 0, l_xkv = l_tile, l_k = 2 * l_tile, l_v = l_k + HG * 4 * 1024, l_total = l_v + HG * 4 * 1024;
    static_assert(XS % 16 == 0 && (XS / 16) % 2 == 1, "row stride must be an odd number of 16-byte slots");
    static_assert(l_total <= 160 * 1024, "LDS");
};

struct QaDev {
    const char* packed[2];
    const bf16* xn_hi[2]; const bf16* xn_lo[2];
    bf16* o_hi[2]; bf16* o_lo[2];
    float* part[2][2];   // [head group][stream]: projection partial sums [token][C] fp32 (nullptr: write O planes instead)
    int B, H, W, shift, cross, nstream;
};

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// split-bf16 x3: small cross terms first
__device__ __forceinline__ f32x16 mma3(u32x4 ahi, u32x4 alo, u32x4 bhi, u32x4 blo, f32x16 acc) {
    acc = mfma_bf16(alo, bhi, acc);
    acc = mfma_bf16(ahi, blo, acc);
    acc = mfma_bf16(ahi, bhi, acc);
    return acc;
}
__device__ __forceinline__ u32x4 pack8_f16(const float* v) {
    u32x4 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f16x2 h = {(f16)v[2 * p], (f16)v[2 * p + 1]};
        o[p] = __builtin_bit_cast(unsigned, h);
    }
    return o;
}
// 8 fp32 values -> one k-step fragment in split-bf16 (hi = bf16(v), lo = bf16(v - hi))
__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& lo) {
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
// a = the value of lanes 0..31 (in both halves), b = the value of lanes 32..63 (kernels_win24.hip)
__device__ __forceinline__ void halves(float v, float& a, float& b) {
    a = v;
    b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
#define QA_FENCE() asm volatile("" ::: "memory")

#ifdef QA_PROBE   // diagnostic build: 10-ns wall-clock stamps of workgroup QA_PROBE, wave 0, into a buffer nothing else reads
__device__ unsigned long long qa_probe[8];
#define QA_STAMP(i) do { if (blockIdx.x == QA_PROBE && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) qa_probe[i] = wall_clock64(); } while (0)
#else
#define QA_STAMP(i) do { } while (0)
#endif

// WS = window side, 8 or 7 (the reference's default) on the same 8x8 token grid: a padding token stages a neighbouring token's
// row (any finite row serves), carries -inf in the packed bias matrix as key, and is not stored.
template <int C, int WS>
__global__ __launch_bounds__(512, 2) void qkv_attn_kernel(QaDev a) {
    using G = QA<C>;
    static_assert(WS == 7 || WS == 8, "window side");
    constexpr int KS = G::KS, XS = G::XS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* kimg = reinterpret_cast<u32x4*>(smem + G::l_k);   // [head in group][key tile][k-step][lane]
    u32x4* vimg = reinterpret_cast<u32x4*>(smem + G::l_v);   // [head in group][pv-step][lane]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hl = wave >> 1, tt = wave & 1, r = lane & 31, hf = lane >> 5;
    const int H = a.H, W = a.W, nwx = W / WS, nwy = H / WS, npi = nwx * nwy;
    const int win = blockIdx.x, st = blockIdx.y, hg = blockIdx.z, head = hg * G::HG + hl;
    const int b = win / npi, wrem = win - b * npi, wy = wrem / nwx, wx = wrem - wy * nwx;
    const int sh = a.shift ? WS / 2 : 0;
    const bool cross = a.cross && a.nstream == 2;
    const int kvs = cross ? 1 - st : st;

    // image token index of window token t (cyclic shift = index arithmetic, a001:442-445)
    auto tok_index = [&](int t) {
        int oy = wy * WS + (t >> 3) + sh, ox = wx * WS + (t & 7) + sh;
        oy = oy >= H ? oy - H : oy;
        ox = ox >= W ? ox - W : ox;
        return (b * H + oy) * W + ox;
    };

    QA_STAMP(0);
    // ---- weight fragments: a register ring PD k-steps deep, filled before anything else so the first L2 round trips run
    //      under the staging of the token rows ----
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(uniform_ptr(a.packed[st])), 0, (int)G::p_total, 0x00020000);
    const unsigned loff = (unsigned)lane * 16u;
    const int fbase = head * 3 * KS * 2;   // fragment index of (head, q, step 0, hi)
    auto WFRAG = [&](int m, int s, int hl2) {
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, ((fbase + (m * KS + s) * 2 + hl2)) * 1024, 0));
    };
    constexpr int PD = 4;
    u32x4 w[PD][6];
#pragma unroll
    for (int p = 0; p < PD && p < KS; ++p)
#pragma unroll
        for (int m = 0; m < 3; ++m) { w[p][2 * m] = WFRAG(m, p, 0); w[p][2 * m + 1] = WFRAG(m, p, 1); }

    // ---- stage the window's LayerNorm rows: 2 planes x 64 rows x C bf16, row-major, stride XS.  All of a thread's loads are
    //      issued before its first LDS store (one L2 round trip, not one per chunk) ----
    {
        constexpr int CPR = C / 8, NCH = 2 * 64 * CPR, PER = NCH / 512;   // 16-byte chunks per row / per tile / per thread
        static_assert(NCH % 512 == 0, "chunks per tile must divide evenly over the workgroup");
        const int ntile = cross ? 2 : 1;
        for (int tile = 0; tile < ntile; ++tile) {
            const int s_src = tile ? kvs : st;
            u32x4 v[PER];
            int dst[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int rem = tid + 512 * i, plane = rem / (64 * CPR), rem2 = rem - plane * 64 * CPR;
                const int row = rem2 / CPR, j = rem2 - row * CPR;
                const bf16* src = (plane ? a.xn_lo[s_src] : a.xn_hi[s_src]) + (int64_t)tok_index(row) * C + j * 8;
                v[i] = *reinterpret_cast<const u32x4*>(src);
                dst[i] = tile * (int)G::l_tile + plane * (int)G::l_plane + row * XS + j * 16;
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) *reinterpret_cast<u32x4*>(smem + dst[i]) = v[i];
        }
    }
    __syncthreads();
    QA_STAMP(1);

    // ---- Q, K, V tiles of (head, token half): K = C in 16-deep steps; weight fragments from the register ring ----
    const char* xq = smem + G::l_xq + (32 * tt + r) * XS + hf * 16;
    const char* xk = smem + (cross ? G::l_xkv : G::l_xq) + (32 * tt + r) * XS + hf * 16;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 aq = zero16, ak = zero16, av = zero16;
    {
        // Register ring of weight fragments, PD k-steps deep: with one step of run-ahead every step waited a whole L2 round trip
        // (12 steps x ~0.7 us); the ring keeps 6 * PD loads (1 KB each) in flight per wave.  Fences pin the issue points.
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            QA_FENCE();
            const u32x4 xh = *reinterpret_cast<const u32x4*>(xq + s * 32), xl = *reinterpret_cast<const u32x4*>(xq + G::l_plane + s * 32);
            u32x4 kh = xh, kl = xl;
            if (cross) {
                kh = *reinterpret_cast<const u32x4*>(xk + s * 32);
                kl = *reinterpret_cast<const u32x4*>(xk + G::l_plane + s * 32);
            }
            u32x4 (&ws_)[6] = w[s % PD];
            aq = mma3(ws_[0], ws_[1], xh, xl, aq);   // [virtual channel][token]
            ak = mma3(ws_[2], ws_[3], kh, kl, ak);
            av = mma3(kh, kl, ws_[4], ws_[5], av);   // [token][virtual channel]: the accumulator registers are V^T's key slots
            QA_FENCE();
            if (s + PD < KS) {
#pragma unroll
                for (int m = 0; m < 3; ++m) { ws_[2 * m] = WFRAG(m, s + PD, 0); ws_[2 * m + 1] = WFRAG(m, s + PD, 1); }
            }
        }
    }
    QA_STAMP(2);
    // ---- biases (virtual-channel order; K's and V's channel 24 carry the constant 1), f16 operand fragments ----
    u32x4 qf[2];
    {
        const float* bvec = reinterpret_cast<const float*>(a.packed[st] + G::p_bvec) + head * 3 * 32;
        float t[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bq = *reinterpret_cast<const float4*>(bvec + 8 * g + 4 * hf);
            t[4 * g] = aq[4 * g] + bq.x; t[4 * g + 1] = aq[4 * g + 1] + bq.y; t[4 * g + 2] = aq[4 * g + 2] + bq.z; t[4 * g + 3] = aq[4 * g + 3] + bq.w;
        }
        qf[0] = pack8_f16(t);
        qf[1] = pack8_f16(t + 8);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bk = *reinterpret_cast<const float4*>(bvec + 32 + 8 * g + 4 * hf);
            t[4 * g] = ak[4 * g] + bk.x; t[4 * g + 1] = ak[4 * g + 1] + bk.y; t[4 * g + 2] = ak[4 * g + 2] + bk.z; t[4 * g + 3] = ak[4 * g + 3] + bk.w;
        }
        u32x4* kdst = kimg + ((hl * 2 + tt) * 2) * 64 + lane;
        kdst[0] = pack8_f16(t);
        kdst[64] = pack8_f16(t + 8);
        const float bv = bvec[64 + r];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = av[i] + bv;
        u32x4* vdst = vimg + (hl * 4 + 2 * tt) * 64 + lane;
        vdst[0] = pack8_f16(t);
        vdst[64] = pack8_f16(t + 8);
    }
    __syncthreads();
    QA_STAMP(3);

    // ---- attention of (head, query tile tt) ----
    const int qt = tt;
    f32x16 bias[2];
    {
        const float* bm = reinterpret_cast<const float*>(a.packed[st] + G::p_bias) + qt * 2 * 16 * 64 + lane;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) bias[kt][i] = bm[(kt * 16 + i) * 64];
    }
    const bool rowv = a.shift && wy == nwy - 1, colv = a.shift && wx == nwx - 1;
    const bool col_masked = (hf != 0) != (((r >> 2) & 1) != 0);
    const bool m0 = (rowv && qt == 1) || (colv && col_masked), m1 = (rowv && qt == 0) || (colv && col_masked);
    const u32x4* ksrc = kimg + (hl * 4) * 64 + lane;
    const u32x4* vsrc = vimg + (hl * 4) * 64 + lane;
    const u32x4 k00 = ksrc[0], k01 = ksrc[64], k10 = ksrc[128], k11 = ksrc[192];
    f32x16 s0 = mfma_f16(k00, qf[0], bias[0]);
    s0 = mfma_f16(k01, qf[1], s0);
    f32x16 s1 = mfma_f16(k10, qf[0], bias[1]);
    s1 = mfma_f16(k11, qf[1], s1);
    if (rowv || colv) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s0[i] = m0 ? -INFINITY : s0[i];
            s1[i] = m1 ? -INFINITY : s1[i];
        }
    }
    float mx = max3f(s0[0], s0[1], s1[0]);
    mx = max3f(mx, s1[1], s0[2]);
#pragma unroll
    for (int i = 3; i < 16; i += 2) mx = max3f(mx, s0[i], s0[i + 1 < 16 ? i + 1 : i]);
#pragma unroll
    for (int i = 2; i < 16; i += 2) mx = max3f(mx, s1[i], s1[i + 1]);
    {
        float lo, hi;
        halves(mx, lo, hi);
        mx = __builtin_fmaxf(lo, hi);
    }
    // S - max on the matrix pipe: virtual channel 24 (k-step 1, lane half 0, element 4) is 1 in K and -max in Q
    {
        const f16 nm = (f16)(-mx);
        u32x4 q1 = qf[1];
        q1[2] |= hf == 0 ? (unsigned)__builtin_bit_cast(unsigned short, nm) : 0u;
        s0 = mfma_f16(k00, qf[0], bias[0]);
        s0 = mfma_f16(k01, q1, s0);
        s1 = mfma_f16(k10, qf[0], bias[1]);
        s1 = mfma_f16(k11, q1, s1);
    }
    if (rowv || colv) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s0[i] = m0 ? -INFINITY : s0[i];
            s1[i] = m1 ? -INFINITY : s1[i];
        }
    }
    f32x16 t;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        float p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = __builtin_amdgcn_exp2f((ps >> 1) ? s1[8 * (ps & 1) + j] : s0[8 * (ps & 1) + j]);
        t = mfma_f16(vsrc[ps * 64], pack8_f16(p), ps == 0 ? zero16 : t);
    }
    QA_STAMP(4);
    // ---- normalise (row 24 = register 12 of lane half 0 holds the denominator) ----
    float den, unused;
    halves(t[12], den, unused);
    const float inv = __builtin_amdgcn_rcpf(den);
    auto padding = [](int t) { return WS != 8 && ((t >> 3) >= WS || (t & 7) >= WS); };
    if (a.part[0][st] == nullptr) {   // O leaves as split-bf16 planes (the projection GEMM's input)
        if (padding(32 * qt + r)) return;   // (no barrier follows on this path)
        const int64_t orow = (int64_t)tok_index(32 * qt + r) * (G::HEADS * G::D) + head * G::D + 4 * hf;
        bf16* oh = a.o_hi[st] + orow;
        bf16* ol = a.o_lo[st] + orow;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            bf16x4 h4, l4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = t[4 * g + j] * inv;
                h4[j] = (bf16)v;
                l4[j] = (bf16)(v - (float)h4[j]);
            }
            *reinterpret_cast<bf16x4*>(oh + 8 * g) = h4;
            *reinterpret_cast<bf16x4*>(ol + 8 * g) = l4;
        }
    } else {
        // ---- output projection of this head group (a001:470-472), partial over the group's 4 x 24 channels:
        //      P[token][c] = sum_{head, d} O[token][head][d] . Wp[c][head*24 + d].  The O^T accumulator registers of a lane,
        //      split to bf16 hi / lo, ARE the A fragments (rows = tokens, k = the head's virtual channels in accumulator order;
        //      the pack stores Wp's k in that order, zero for the padding channels 24..31 — channel 24 holds the denominator).
        //      The 8 waves exchange them through LDS (the dead token tile) and share the C / 32 output tiles: wave (hl, tt)
        //      owns tiles hl, hl + 4, ... of token half tt.  The other head group's partial is added by the consumer (the fused
        //      MLP kernel's prologue), in fixed order.
        constexpr int NT = G::NT;
        const int pbase = (int)(G::p_proj / 1024) + hg * NT * G::HG * 4;
        auto PFRAG = [&](int tile, int f) {   // f = (head in group * 2 + k-step) * 2 + hi/lo
            return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, (pbase + tile * G::HG * 4 + f) * 1024, 0));
        };
        u32x4 wp[G::HG * 4];
#pragma unroll
        for (int f = 0; f < G::HG * 4; ++f) wp[f] = PFRAG(hl, f);   // in flight during the exchange
        float ov[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) ov[i] = t[i] * inv;
        u32x4* ofr = reinterpret_cast<u32x4*>(smem + G::l_xq);   // [head in group][token half][k-step][hi,lo][lane]
        {
            u32x4 h0, l0, h1, l1;
            split8(ov, h0, l0);
            split8(ov + 8, h1, l1);
            u32x4* dst = ofr + ((hl * 2 + tt) * 4) * 64 + lane;
            dst[0] = h0; dst[64] = l0; dst[128] = h1; dst[192] = l1;
        }
        __syncthreads();
        int tokrow[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) tokrow[i] = tok_index(32 * tt + rho(i, hf)) * C;
        float* pout = a.part[hg][st];
        for (int tile = hl; tile < NT; tile += G::HG) {
            f32x16 acc = zero16;
#pragma unroll
            for (int h2 = 0; h2 < G::HG; ++h2)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const u32x4* src = ofr + ((h2 * 2 + tt) * 4 + 2 * ks) * 64 + lane;
                    acc = mma3(src[0], src[64], wp[(h2 * 2 + ks) * 2], wp[(h2 * 2 + ks) * 2 + 1], acc);
                }
            if (tile + G::HG < NT) {
#pragma unroll
                for (int f = 0; f < G::HG * 4; ++f) wp[f] = PFRAG(tile + G::HG, f);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (!padding(32 * tt + rho(i, hf))) pout[tokrow[i] + 32 * tile + r] = acc[i];
        }
    }
    QA_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------
struct QaPackArgs { swf_attn_params p; char* dst; int ws; };

template <int C>
__global__ __launch_bounds__(256) void qa_pack_kernel(QaPackArgs a) {
    using G = QA<C>;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    const float qscale = kLog2e / sqrtf((float)G::D);
    for (int idx = gtid; idx < G::NFRAG * 512; idx += gsz) {
        const int f = idx >> 9, lane = (idx >> 3) & 63, e = idx & 7, r = lane & 31, hf = lane >> 5;
        const int hl = f & 1, s = (f >> 1) % G::KS, m = (f / (2 * G::KS)) % 3, head = f / (6 * G::KS);
        const swf_linear& l = m == 0 ? a.p.q : m == 1 ? a.p.k : a.p.v;
        float val = 0.f;
        if (r < G::D) {
            val = l.weight[(int64_t)(head * G::D + r) * C + s * 16 + 8 * hf + e];   // row = virtual channel r of the head, k in natural order
            if (m == 0) val *= qscale;
        }
        const bf16 hi = (bf16)val;
        reinterpret_cast<bf16*>(a.dst)[idx] = hl ? (bf16)(val - (float)hi) : hi;
    }
    // output projection, B fragments: lane (r, hf) element e of (group, tile, head in group, k-step) = Wp[32 tile + r][head*24 + vc],
    // vc = the row of accumulator register 8 ks + e in lane half hf
    bf16* pf = reinterpret_cast<bf16*>(a.dst + G::p_proj);
    for (int idx = gtid; idx < G::NPF * 512; idx += gsz) {
        const int f = idx >> 9, lane = (idx >> 3) & 63, e = idx & 7, r = lane & 31, hf = lane >> 5;
        const int hl = f & 1, ks = (f >> 1) & 1, h2 = (f >> 2) % G::HG, tile = (f / (4 * G::HG)) % G::NT, hg = f / (4 * G::HG * G::NT);
        const int vc = rho(8 * ks + e, hf);
        float val = 0.f;
        if (vc < G::D) val = a.p.proj.weight[(int64_t)(32 * tile + r) * (G::HEADS * G::D) + (hg * G::HG + h2) * G::D + vc];
        const bf16 hi = (bf16)val;
        pf[idx] = hl ? (bf16)(val - (float)hi) : hi;
    }
    float* bvec = reinterpret_cast<float*>(a.dst + G::p_bvec);
    for (int i = gtid; i < G::HEADS * 3 * 32; i += gsz) {
        const int c = i & 31, m = (i >> 5) % 3, head = i / 96;
        const swf_linear& l = m == 0 ? a.p.q : m == 1 ? a.p.k : a.p.v;
        float v = 0.f;
        if (c < G::D) v = (l.bias ? l.bias[head * G::D + c] : 0.f) * (m == 0 ? qscale : 1.0f);
        else if (c == G::D && m != 0) v = 1.0f;   // K: the slot Q carries -max on; V: the softmax denominator's channel
        bvec[i] = v;
    }
    float* bm = reinterpret_cast<float*>(a.dst + G::p_bias);
    for (int i = gtid; i < 2 * 2 * 16 * 64; i += gsz) {
        const int lane = i & 63, reg = (i >> 6) & 15, kt = (i >> 10) & 1, qt = i >> 11;
        const int key = 32 * kt + rho(reg, lane >> 5), q = 32 * qt + (lane & 31);
        const int ky = key >> 3, kx = key & 7, qy = q >> 3, qx = q & 7, ws = a.ws, tw = 2 * ws - 1;
        float v = 0.f;
        if (ky >= ws || kx >= ws) v = -INFINITY;   // padding token of a 7x7 window as key: probability 0
        else if (qy < ws && qx < ws) v = a.p.bias_table[(ky - qy + ws - 1) * tw + (kx - qx + ws - 1)] * kLog2e;
        bm[i] = v;
    }
}

}  // namespace

bool qkvattn_supported(const swf_block_desc& d) {
    return d.precision == SWF_PREC_FAST && d.attn.channels == 192 && d.attn.heads == 8 && d.attn.head_dim == 24 && d.attn.win_h == d.attn.win_w &&
           (d.attn.win_h == 8 || d.attn.win_h == 7);
}

size_t qkvattn_packed_bytes(const swf_block_desc& d) { return qkvattn_supported(d) ? align_up(QA<192>::p_total, 256) : 0; }

int pack_qkvattn(const swf_block_desc& d, const swf_block_stream_params& p, void* dst, hipStream_t stream) {
    if (!qkvattn_supported(d)) return fail(SWF_ERR_UNSUPPORTED, "pack_qkvattn: shape not covered");
    QaPackArgs a{p.attn, static_cast<char*>(dst), d.attn.win_h};
    hipLaunchKernelGGL((qa_pack_kernel<192>), dim3(128), dim3(256), 0, stream, a);
    return check_launch("pack_qkvattn");
}

int launch_qkvattn(const swf_block_desc& d, const QkvAttnArgs& a, int nstream, hipStream_t stream) {
    const int wsd = d.attn.win_h;
    if (!qkvattn_supported(d) || a.H % wsd || a.W % wsd) return fail(SWF_ERR_UNSUPPORTED, "qkvattn: shape not covered");
    using G = QA<192>;
    static hipError_t attr_err = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qkv_attn_kernel<192, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::l_total);
        return e != hipSuccess ? e : hipFuncSetAttribute(reinterpret_cast<const void*>(&qkv_attn_kernel<192, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::l_total);
    }();
    if (attr_err != hipSuccess) return fail(SWF_ERR_HIP, "hipFuncSetAttribute(qkv_attn): %s", hipGetErrorString(attr_err));
    QaDev dv{};
    for (int s = 0; s < nstream; ++s) {
        dv.packed[s] = static_cast<const char*>(a.packed[s]);
        dv.xn_hi[s] = reinterpret_cast<const bf16*>(a.xn_hi[s]); dv.xn_lo[s] = reinterpret_cast<const bf16*>(a.xn_lo[s]);
        dv.o_hi[s] = reinterpret_cast<bf16*>(a.o_hi[s]); dv.o_lo[s] = reinterpret_cast<bf16*>(a.o_lo[s]);
        dv.part[0][s] = a.part[0][s]; dv.part[1][s] = a.part[1][s];
    }
    if ((dv.part[0][0] != nullptr) != (dv.part[1][0] != nullptr) || (nstream == 2 && (dv.part[0][1] != nullptr) != (dv.part[0][0] != nullptr)))
        return fail(SWF_ERR_NULL, "qkvattn: projection partial buffers must be given for both head groups and streams or not at all");
    dv.B = a.B; dv.H = a.H; dv.W = a.W; dv.shift = a.shift; dv.cross = a.cross; dv.nstream = nstream;
    const int nwin = a.B * (a.H / wsd) * (a.W / wsd);
    if (wsd == 8) hipLaunchKernelGGL((qkv_attn_kernel<192, 8>), dim3(nwin, nstream, 2), dim3(512), G::l_total, stream, dv);
    else hipLaunchKernelGGL((qkv_attn_kernel<192, 7>), dim3(nwin, nstream, 2), dim3(512), G::l_total, stream, dv);
    return check_launch("qkv_attn");
}

}  // namespace swf
