// Fused fast tier for gfx950 (MI355X): ONE launch = one BasicBlock (a005:127-145) for both modality
// streams.  A 512-thread workgroup (8 waves, 2 per SIMD) owns one 8x8 window: it loads the two
// 64xC fp32 token tiles once (cyclic shift = index arithmetic on the load), keeps the residual
// stream in LDS, runs LN -> QKV -> per-head attention -> proj -> LN -> MLP entirely on chip and
// stores the two tiles once.  HBM traffic per block = read + write of each stream, nothing else.
//
// Arithmetic (include/swinfuse.h SWF_PREC_FAST; error budget measured in DESIGN.md):
//   linear layers : split-bf16 "bf16x3" on v_mfma_f32_16x16x32_bf16 — a = a_hi + a_lo, w = w_hi + w_lo,
//                   a.w ~= a_lo.w_hi + a_hi.w_lo + a_hi.w_hi, fp32 accumulate (~2^-17 relative: fp32-grade;
//                   plain bf16 linears miss the 1e-3 parity gate by 4-10x, gfx950 has no xf32/tf32)
//   Q.K^T         : bf16 on v_mfma_f32_32x32x16_bf16, computed swapped (S^T = K.Q^T) so a lane owns one
//                   query column: softmax max/sum are in-lane + one cross-half exchange
//   P.V           : fp16 on v_mfma_f32_32x32x16_f16; the S^T accumulator tile converts in registers to the
//                   B operand of O^T = V^T.P^T (no LDS round trip; V^T is stored in the matching k order)
//   LayerNorm statistics, bias, mask, softmax, ELU, residual stream, accumulators: fp32.
//   exp() runs as v_exp_f32 (exp2): Wq/bq carry d^-0.5*log2(e), the bias matrices carry log2(e).
#include "kernels_window.h"

#include <algorithm>
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
// geometry shared by the pack kernel, the block kernel and the host
// ------------------------------------------------------------------------------------------
template <int C_, int HID_>
struct Geo {
    static constexpr int C = C_, HID = HID_, T = 64, WH = 8, WW = 8, HEADS = 8, D = C / 8;
    static_assert(C % 8 == 0, "8 heads of C/8 channels");
    static constexpr int KC = cround(C, 32), KH = cround(HID, 32);      // K extents padded to the MFMA k-step
    static constexpr int LDC = KC + 8, LDH = KH + 8;                    // LDS row strides (bf16): odd multiples of 16 B -> conflict-free b128 reads
    static constexpr int NTC = cceil(C, 16), NTH = cceil(HID, 16);      // 16-wide output tiles
    static constexpr int NH = NTH * 16;
    static constexpr int QS = cround(D, 8);                             // bf16 slots per (head, token) row of the Q / K images
    static constexpr int QKS = cceil(D, 16);                            // 16-deep k-steps of Q.K^T
    static constexpr int MT = cceil(D, 32);                             // 32-row M tiles of O^T per head

    // ---- packed weights of one stream.  The first `wsec` bytes are staged verbatim into LDS. ----
    static constexpr size_t p_wqkv_hi = 0, p_wqkv_lo = p_wqkv_hi + size_t(3) * C * KC * 2;     // [3C][KC] bf16 (Wq pre-scaled)
    static constexpr size_t p_wp_hi = p_wqkv_lo + size_t(3) * C * KC * 2, p_wp_lo = p_wp_hi + size_t(C) * KC * 2;
    static constexpr size_t p_w1_hi = p_wp_lo + size_t(C) * KC * 2, p_w1_lo = p_w1_hi + size_t(HID) * KC * 2;
    static constexpr size_t p_w2_hi = p_w1_lo + size_t(HID) * KC * 2, p_w2_lo = p_w2_hi + size_t(C) * KH * 2;
    static constexpr size_t p_vec = (p_w2_lo + size_t(C) * KH * 2 + 15) / 16 * 16;             // fp32 vectors
    static constexpr int v_ln1g = 0, v_ln1b = C, v_ln2g = 2 * C, v_ln2b = 3 * C, v_bqkv = 4 * C, v_bp = 7 * C, v_b2 = 8 * C,
                         v_b1 = 9 * C, v_end = 9 * C + NH;
    static constexpr size_t wsec = (p_vec + size_t(v_end) * 4 + 15) / 16 * 16;
    static constexpr size_t p_bias4 = wsec;                                                   // [4 variants][64 keys][64 queries] fp32, global only
    static constexpr size_t p_total = p_bias4 + size_t(4) * T * T * 4;

    // ---- LDS carve (bytes) ----
    static constexpr size_t l_resid = 0;                                  // fp32 [2][64][C]: the residual stream
    static constexpr size_t l_ahi = l_resid + size_t(2) * T * C * 4;      // bf16 [2][64][LDC]: A-operand image (xn / O / xn2), hi part
    static constexpr size_t l_alo = l_ahi + size_t(2) * T * LDC * 2;
    static constexpr size_t l_u = l_alo + size_t(2) * T * LDC * 2;        // union: attention images | MLP hidden images
    static constexpr size_t l_q = l_u;                                    // bf16 [2][8][64][QS]
    static constexpr size_t l_k = l_q + size_t(2) * HEADS * T * QS * 2;
    static constexpr size_t l_vt = l_k + size_t(2) * HEADS * T * QS * 2;  // fp16 [2][8][D][64], keys in MFMA k order
    static constexpr size_t attn_bytes = size_t(4) * HEADS * T * QS * 2 + size_t(2) * C * T * 2;
    static constexpr size_t l_hhi = l_u;                                  // bf16 [2][64][LDH]
    static constexpr size_t l_hlo = l_hhi + size_t(2) * T * LDH * 2;
    static constexpr size_t mlp_bytes = size_t(4) * T * LDH * 2;
    static constexpr size_t l_w = (l_u + cmax(attn_bytes, mlp_bytes) + 15) / 16 * 16;   // the two streams' weight sections
    static constexpr size_t l_total = l_w + 2 * wsec;
    static_assert(l_total <= 160 * 1024, "window tile + weights exceed the 160 KiB LDS of a CU");
};

struct WinArgs {
    const float* in[2];
    float* out[2];
    const char* packed[2];
    int B, H, W, shift, cross;
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float elu_f(float v) { return v > 0.f ? v : expm1f(v); }

__device__ __forceinline__ void split_bf16(float v, bf16& hi, bf16& lo) {
    hi = (bf16)v;
    lo = (bf16)(v - (float)hi);
}

// position of key `tok` (0..63) inside a V^T row so that the 8 halves a lane needs for k-step s of
// key tile T sit contiguously: the S^T accumulator register 8s+e of lane half h is key row
// 32T + 16s + 8(e>>2) + 4h + (e&3)  (C/D map of the 32x32 MFMA), so pos = 32T + 16s + 8h + e.
__device__ __forceinline__ int vt_pos(int tok) {
    const int k16 = tok & 15;
    const int e = ((k16 >> 3) << 2) | (k16 & 3);
    const int h = (k16 >> 2) & 1;
    return (tok & 48) | (h << 3) | e;
}

// B operand (weights) of one 16-wide output tile, all k-steps, split-bf16: lives in registers while the
// wave walks the M tiles that share it.
template <int KSTEPS>
struct BFrag {
    bf16x8 hi[KSTEPS], lo[KSTEPS];
};

template <int KSTEPS, int LDW>
__device__ __forceinline__ void load_bfrag(BFrag<KSTEPS>& f, const bf16* w_hi, const bf16* w_lo, int row, int g) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        f.hi[ks] = *reinterpret_cast<const bf16x8*>(w_hi + row * LDW + ks * 32 + 8 * g);
        f.lo[ks] = *reinterpret_cast<const bf16x8*>(w_lo + row * LDW + ks * 32 + 8 * g);
    }
}

// one 16x16 output tile of A[16 x K] . W[16 x K]^T with split-bf16 operands (three MFMAs per k-step;
// small cross terms first so they are not absorbed by the large hi.hi partial sums)
template <int KSTEPS, int LDA>
__device__ __forceinline__ f32x4 tile_bf16x3(const bf16* a_hi, const bf16* a_lo, const BFrag<KSTEPS>& b, int r, int g) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(a_hi + r * LDA + ks * 32 + 8 * g);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(a_lo + r * LDA + ks * 32 + 8 * g);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, b.hi[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b.lo[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, b.hi[ks], acc, 0, 0, 0);
    }
    return acc;
}

// LayerNorm of the 128 residual rows (2 streams x 64 tokens) into the split-bf16 A image; 4 lanes per row
template <typename G>
__device__ __forceinline__ void layernorm_to_image(const float* resid, bf16* ahi, bf16* alo, const float* vec0,
                                                   const float* vec1, int goff, int boff, int tid) {
    constexpr int C = G::C, PER = C / 4;
    const int row = tid >> 2, part = tid & 3;
    const float* vec = (row >> 6) ? vec1 : vec0;
    const float* x = resid + row * C;
    float v[PER];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { v[i] = x[part + 4 * i]; sum += v[i]; }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    const float mean = sum * (1.0f / C);
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = v[i] - mean; var = fmaf(d, d, var); }
    var += __shfl_xor(var, 1);
    var += __shfl_xor(var, 2);
    const float rstd = 1.0f / sqrtf(var * (1.0f / C) + 1e-5f);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = part + 4 * i;
        const float n = (v[i] - mean) * rstd * vec[goff + c] + vec[boff + c];
        bf16 hi, lo;
        split_bf16(n, hi, lo);
        ahi[row * G::LDC + c] = hi;
        alo[row * G::LDC + c] = lo;
    }
}

// ------------------------------------------------------------------------------------------
// the block kernel: persistent, one workgroup per CU walks the windows
// ------------------------------------------------------------------------------------------
template <int C_, int HID_>
__global__ __launch_bounds__(512, 2) void window_block_kernel(WinArgs args) {
    using G = Geo<C_, HID_>;
    constexpr int C = G::C, D = G::D, T = G::T, QS = G::QS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* resid = reinterpret_cast<float*>(smem + G::l_resid);
    bf16* ahi = reinterpret_cast<bf16*>(smem + G::l_ahi);
    bf16* alo = reinterpret_cast<bf16*>(smem + G::l_alo);
    bf16* qimg = reinterpret_cast<bf16*>(smem + G::l_q);
    bf16* kimg = reinterpret_cast<bf16*>(smem + G::l_k);
    f16* vt = reinterpret_cast<f16*>(smem + G::l_vt);
    bf16* hhi = reinterpret_cast<bf16*>(smem + G::l_hhi);
    bf16* hlo = reinterpret_cast<bf16*>(smem + G::l_hlo);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = args.H, W = args.W;
    const int nwx = W / G::WW, nwy = H / G::WH;
    const int nwin = args.B * nwx * nwy;
    const int sh = args.shift ? G::WH / 2 : 0, sw = args.shift ? G::WW / 2 : 0;

    auto wsec = [&](int s) { return smem + G::l_w + s * G::wsec; };
    auto wmat = [&](int s, size_t off) { return reinterpret_cast<const bf16*>(wsec(s) + off); };
    auto wvec = [&](int s) { return reinterpret_cast<const float*>(wsec(s) + G::p_vec); };

    // ---- once per workgroup: stage both streams' weights into LDS, clear the A / Q / K images
    //      (their K padding must stay exact zeros: 0 * stale-NaN would poison a dot product) ----
    {
        constexpr int W16 = G::wsec / 16;
        for (int i = tid; i < 2 * W16; i += 512) {
            const int s = i / W16, e = i % W16;
            reinterpret_cast<uint4*>(wsec(s))[e] = reinterpret_cast<const uint4*>(args.packed[s])[e];
        }
        constexpr int Z16 = (G::l_vt - G::l_ahi) / 16;
        uint4* z = reinterpret_cast<uint4*>(smem + G::l_ahi);
        for (int i = tid; i < Z16; i += 512) z[i] = make_uint4(0, 0, 0, 0);
    }

    // window tile <-> registers: 2*64*C/4 float4 over 512 threads
    constexpr int V4 = T * C / 4, NV = cceil(2 * V4, 512);
    float4 pre[NV];
    auto tile_addr = [&](int win, int i) -> int64_t {
        const int b = win / (nwx * nwy), wrem = win % (nwx * nwy);
        const int wy = wrem / nwx, wx = wrem % nwx;
        const int e = i % V4;
        const int tok = e / (C / 4), c4 = e % (C / 4);
        const int oy = (wy * G::WH + tok / G::WW + sh) % H, ox = (wx * G::WW + tok % G::WW + sw) % W;   // roll(-s): read at (y+s)%H
        return (((int64_t)b * H + oy) * W + ox) * C + c4 * 4;
    };
    auto prefetch = [&](int win) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + k * 512;
            if (i < 2 * V4) pre[k] = *reinterpret_cast<const float4*>(args.in[i / V4] + tile_addr(win, i));
        }
    };

    int win = blockIdx.x;
    if (win < nwin) prefetch(win);
    int cur_variant = -1;
    f32x16 bfr[2];   // relative-position bias (+mask) of this wave's (stream, query block), S^T layout, exp2 units
    __syncthreads();

    for (; win < nwin; win += gridDim.x) {
        const int wrem = win % (nwx * nwy);
        const int wy = wrem / nwx, wx = wrem % nwx;
        // ---- phase 0: registers -> residual tile; start fetching the next window ----
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + k * 512;
            if (i < 2 * V4) {
                const int e = i % V4;
                *reinterpret_cast<float4*>(resid + ((i / V4) * T + e / (C / 4)) * C + (e % (C / 4)) * 4) = pre[k];
            }
        }
        __syncthreads();
        if (win + (int)gridDim.x < nwin) prefetch(win + gridDim.x);

        // ---- phase 1: LN1 -> split-bf16 A image ----
        layernorm_to_image<G>(resid, ahi, alo, wvec(0), wvec(1), G::v_ln1g, G::v_ln1b, tid);
        __syncthreads();

        // ---- phase 2: Q, K, V projections.  flat job = ((stream, n-tile of [Q|K|V]), m-tile), m fastest,
        //      each wave takes a contiguous run so a weight fragment is fetched once per n-tile ----
        {
            constexpr int NJ = 2 * 3 * G::NTC * 4, JPW = cceil(NJ, 8);
            const int r = lane & 15, g = lane >> 4;
            BFrag<G::KC / 32> bf;
            int cur = -1;
#pragma unroll 1
            for (int jj = 0; jj < JPW; ++jj) {
                const int job = wave * JPW + jj;
                if (job >= NJ) break;
                const int m = job & 3, sn = job >> 2;
                const int s = sn / (3 * G::NTC), n = sn % (3 * G::NTC);
                const int which = n / G::NTC, nt = n % G::NTC;
                const int ch = nt * 16 + r;
                if (sn != cur) {
                    cur = sn;
                    load_bfrag<G::KC / 32, G::KC>(bf, wmat(s, G::p_wqkv_hi), wmat(s, G::p_wqkv_lo), which * C + (ch < C ? ch : 0), g);
                }
                const int src = (which != 0 && args.cross) ? 1 - s : s;   // cross: K,V come from the other stream (a002:67-82)
                const f32x4 acc = tile_bf16x3<G::KC / 32, G::LDC>(ahi + (src * T + m * 16) * G::LDC, alo + (src * T + m * 16) * G::LDC, bf, r, g);
                if (ch < C) {
                    const float bias = wvec(s)[G::v_bqkv + which * C + ch];
                    const int head = ch / D, c = ch % D;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int tok = m * 16 + 4 * g + j;
                        const float v = acc[j] + bias;
                        if (which == 0) qimg[((s * G::HEADS + head) * T + tok) * QS + c] = (bf16)v;
                        else if (which == 1) kimg[((s * G::HEADS + head) * T + tok) * QS + c] = (bf16)v;
                        else vt[((s * G::HEADS + head) * D + c) * T + vt_pos(tok)] = (f16)v;
                    }
                }
            }
        }
        __syncthreads();

        // ---- phase 3: attention.  wave -> (stream, 32-query block, 4 heads) ----
        {
            const int s = wave >> 2, qb = (wave >> 1) & 1, h0 = (wave & 1) * 4;
            const int r = lane & 31, hf = lane >> 5;
            const int variant = args.shift ? ((wy == nwy - 1) * 2 + (wx == nwx - 1)) : 0;
            if (variant != cur_variant) {   // wave-uniform; only edge windows of shifted blocks differ
                cur_variant = variant;
                const float* bias4 = reinterpret_cast<const float*>(args.packed[s] + G::p_bias4) + variant * T * T;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) bfr[kt][i] = bias4[(32 * kt + (i & 3) + 8 * (i >> 2) + 4 * hf) * T + 32 * qb + r];
            }
            const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int hh = 0; hh < 4; ++hh) {
                const int head = h0 + hh;
                const bf16* qrow = qimg + ((s * G::HEADS + head) * T + 32 * qb + r) * QS;
                f32x16 acc[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    acc[kt] = bfr[kt];
                    const bf16* krow = kimg + ((s * G::HEADS + head) * T + 32 * kt + r) * QS;
#pragma unroll
                    for (int ks = 0; ks < G::QKS; ++ks) {
                        // lane half hf supplies k = 16*ks + 8*hf .. +7; slots beyond the stored row width are zeros
                        bf16x8 ka = zero8, qv = zero8;
                        if (ks * 16 + 8 * hf + 8 <= QS) {
                            ka = *reinterpret_cast<const bf16x8*>(krow + ks * 16 + 8 * hf);
                            qv = *reinterpret_cast<const bf16x8*>(qrow + ks * 16 + 8 * hf);
                        }
                        acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qv, acc[kt], 0, 0, 0);
                    }
                }
                float mx = acc[0][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) mx = fmaxf(mx, acc[0][i]);
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, acc[1][i]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                float l = 0.f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float p = __builtin_amdgcn_exp2f(acc[kt][i] - mx);
                        acc[kt][i] = p;
                        l += p;
                    }
                l += __shfl_xor(l, 32);
                f32x16 o[G::MT];
#pragma unroll
                for (int mt = 0; mt < G::MT; ++mt)
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
                        for (int mt = 0; mt < G::MT; ++mt) {
                            int c = mt * 32 + r;
                            c = c < D ? c : D - 1;   // rows beyond the head width are never read back
                            const f16x8 va = *reinterpret_cast<const f16x8*>(vt + ((s * G::HEADS + head) * D + c) * T + kt * 32 + s2 * 16 + 8 * hf);
                            o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, pf, o[mt], 0, 0, 0);
                        }
                    }
                const float inv = 1.0f / l;
                const int tok = 32 * qb + r;
#pragma unroll
                for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int c = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hf;
                        if (c < D) {
                            bf16 hi, lo;
                            split_bf16(o[mt][i] * inv, hi, lo);
                            ahi[(s * T + tok) * G::LDC + head * D + c] = hi;   // xn is dead: the A image now carries O
                            alo[(s * T + tok) * G::LDC + head * D + c] = lo;
                        }
                    }
            }
        }
        __syncthreads();

        // ---- phase 4: output projection + residual ----
        {
            constexpr int NJ = 2 * G::NTC * 4, JPW = cceil(NJ, 8);
            const int r = lane & 15, g = lane >> 4;
            BFrag<G::KC / 32> bf;
            int cur = -1;
#pragma unroll 1
            for (int jj = 0; jj < JPW; ++jj) {
                const int job = wave * JPW + jj;
                if (job >= NJ) break;
                const int m = job & 3, sn = job >> 2;
                const int s = sn / G::NTC, nt = sn % G::NTC;
                const int ch = nt * 16 + r;
                if (sn != cur) {
                    cur = sn;
                    load_bfrag<G::KC / 32, G::KC>(bf, wmat(s, G::p_wp_hi), wmat(s, G::p_wp_lo), ch < C ? ch : 0, g);
                }
                const f32x4 acc = tile_bf16x3<G::KC / 32, G::LDC>(ahi + (s * T + m * 16) * G::LDC, alo + (s * T + m * 16) * G::LDC, bf, r, g);
                if (ch < C) {
                    const float bias = wvec(s)[G::v_bp + ch];
#pragma unroll
                    for (int j = 0; j < 4; ++j) resid[(s * T + m * 16 + 4 * g + j) * C + ch] += acc[j] + bias;
                }
            }
        }
        __syncthreads();

        // ---- phase 5: LN2 (the hidden image's K padding, which overlays the attention images, is re-zeroed) ----
        layernorm_to_image<G>(resid, ahi, alo, wvec(0), wvec(1), G::v_ln2g, G::v_ln2b, tid);
        if constexpr (G::KH > G::NH) {
            constexpr int PADW = G::KH - G::NH;
            for (int i = tid; i < 2 * T * PADW; i += 512) {
                const int row = i / PADW, c = G::NH + i % PADW;
                hhi[row * G::LDH + c] = (bf16)0.f;
                hlo[row * G::LDH + c] = (bf16)0.f;
            }
        }
        __syncthreads();

        // ---- phase 6: MLP fc1 + ELU -> split-bf16 hidden image ----
        {
            constexpr int NJ = 2 * G::NTH * 4, JPW = cceil(NJ, 8);
            const int r = lane & 15, g = lane >> 4;
            BFrag<G::KC / 32> bf;
            int cur = -1;
#pragma unroll 1
            for (int jj = 0; jj < JPW; ++jj) {
                const int job = wave * JPW + jj;
                if (job >= NJ) break;
                const int m = job & 3, sn = job >> 2;
                const int s = sn / G::NTH, nt = sn % G::NTH;
                const int ch = nt * 16 + r;
                if (sn != cur) {
                    cur = sn;
                    load_bfrag<G::KC / 32, G::KC>(bf, wmat(s, G::p_w1_hi), wmat(s, G::p_w1_lo), ch < G::HID ? ch : 0, g);
                }
                const f32x4 acc = tile_bf16x3<G::KC / 32, G::LDC>(ahi + (s * T + m * 16) * G::LDC, alo + (s * T + m * 16) * G::LDC, bf, r, g);
                const float bias = wvec(s)[G::v_b1 + ch];   // zero-padded to NH
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = ch < G::HID ? elu_f(acc[j] + bias) : 0.f;
                    bf16 hi, lo;
                    split_bf16(v, hi, lo);
                    hhi[(s * T + m * 16 + 4 * g + j) * G::LDH + ch] = hi;
                    hlo[(s * T + m * 16 + 4 * g + j) * G::LDH + ch] = lo;
                }
            }
        }
        __syncthreads();

        // ---- phase 7: MLP fc2 + residual ----
        {
            constexpr int NJ = 2 * G::NTC * 4, JPW = cceil(NJ, 8);
            const int r = lane & 15, g = lane >> 4;
            BFrag<G::KH / 32> bf;
            int cur = -1;
#pragma unroll 1
            for (int jj = 0; jj < JPW; ++jj) {
                const int job = wave * JPW + jj;
                if (job >= NJ) break;
                const int m = job & 3, sn = job >> 2;
                const int s = sn / G::NTC, nt = sn % G::NTC;
                const int ch = nt * 16 + r;
                if (sn != cur) {
                    cur = sn;
                    load_bfrag<G::KH / 32, G::KH>(bf, wmat(s, G::p_w2_hi), wmat(s, G::p_w2_lo), ch < C ? ch : 0, g);
                }
                const f32x4 acc = tile_bf16x3<G::KH / 32, G::LDH>(hhi + (s * T + m * 16) * G::LDH, hlo + (s * T + m * 16) * G::LDH, bf, r, g);
                if (ch < C) {
                    const float bias = wvec(s)[G::v_b2 + ch];
#pragma unroll
                    for (int j = 0; j < 4; ++j) resid[(s * T + m * 16 + 4 * g + j) * C + ch] += acc[j] + bias;
                }
            }
        }
        __syncthreads();

        // ---- phase 8: store both tiles (un-shift = same index map as the load); clear the Q/K image
        //      region again: the hidden image overlaid it and its K padding must read as zeros ----
        {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = tid + k * 512;
                if (i < 2 * V4) {
                    const int e = i % V4;
                    *reinterpret_cast<float4*>(args.out[i / V4] + tile_addr(win, i)) =
                        *reinterpret_cast<const float4*>(resid + ((i / V4) * T + e / (C / 4)) * C + (e % (C / 4)) * 4);
                }
            }
            if constexpr (QS != D) {
                constexpr int Z16 = (G::l_vt - G::l_q) / 16;
                uint4* z = reinterpret_cast<uint4*>(smem + G::l_q);
                for (int i = tid; i < Z16; i += 512) z[i] = make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// weight packing: fp32 nn.Parameter tensors -> the kernel layout above
// ------------------------------------------------------------------------------------------
struct PackArgs {
    swf_block_stream_params p[2];
    char* dst[2];
    int head_dim;
};

template <int C_, int HID_>
__global__ __launch_bounds__(256) void pack_block_kernel(PackArgs a) {
    using G = Geo<C_, HID_>;
    constexpr int C = G::C, HID = G::HID, T = G::T;
    const int s = blockIdx.y;
    const swf_block_stream_params& p = a.p[s];
    char* dst = a.dst[s];
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    const float qscale = kLog2e / sqrtf((float)G::D);   // d^-0.5 (a001:32-34) and the exp -> exp2 change of base

    float* vec = reinterpret_cast<float*>(dst + G::p_vec);
    const swf_linear* qkv[3] = {&p.attn.q, &p.attn.k, &p.attn.v};
    for (int i = gtid; i < G::v_end; i += gsz) {
        float v = 0.f;
        if (i < G::v_ln1b) v = p.ln1.gamma[i - G::v_ln1g];
        else if (i < G::v_ln2g) v = p.ln1.beta[i - G::v_ln1b];
        else if (i < G::v_ln2b) v = p.ln2.gamma[i - G::v_ln2g];
        else if (i < G::v_bqkv) v = p.ln2.beta[i - G::v_ln2b];
        else if (i < G::v_bp) {
            const int which = (i - G::v_bqkv) / C, n = (i - G::v_bqkv) % C;
            v = qkv[which]->bias ? qkv[which]->bias[n] : 0.f;
            if (which == 0) v *= qscale;
        } else if (i < G::v_b2) v = p.attn.proj.bias ? p.attn.proj.bias[i - G::v_bp] : 0.f;
        else if (i < G::v_b1) v = p.fc2.bias ? p.fc2.bias[i - G::v_b2] : 0.f;
        else v = (i - G::v_b1 < HID && p.fc1.bias) ? p.fc1.bias[i - G::v_b1] : 0.f;
        vec[i] = v;
    }
    auto put = [&](size_t off_hi, size_t off_lo, int idx, float v) {
        bf16 hi = (bf16)v;
        bf16 lo = (bf16)(v - (float)hi);
        reinterpret_cast<bf16*>(dst + off_hi)[idx] = hi;
        reinterpret_cast<bf16*>(dst + off_lo)[idx] = lo;
    };
    for (int i = gtid; i < 3 * C * G::KC; i += gsz) {
        const int which = i / (C * G::KC), n = (i / G::KC) % C, k = i % G::KC;
        float v = k < C ? qkv[which]->weight[n * C + k] : 0.f;
        if (which == 0) v *= qscale;
        put(G::p_wqkv_hi, G::p_wqkv_lo, i, v);
    }
    for (int i = gtid; i < C * G::KC; i += gsz) {
        const int n = i / G::KC, k = i % G::KC;
        put(G::p_wp_hi, G::p_wp_lo, i, k < C ? p.attn.proj.weight[n * C + k] : 0.f);
    }
    for (int i = gtid; i < HID * G::KC; i += gsz) {
        const int n = i / G::KC, k = i % G::KC;
        put(G::p_w1_hi, G::p_w1_lo, i, k < C ? p.fc1.weight[n * C + k] : 0.f);
    }
    for (int i = gtid; i < C * G::KH; i += gsz) {
        const int n = i / G::KH, k = i % G::KH;
        put(G::p_w2_hi, G::p_w2_lo, i, k < HID ? p.fc2.weight[n * HID + k] : 0.f);
    }
    // relative-position bias (a001:113-144) with the shift mask (a001:217-315) folded in, four variants:
    // bit1 = window in the last window row, bit0 = window in the last window column.  Only those windows
    // contain more than one region label; inside them the label bands split at wh - wh/2 (ww - ww/2).
    constexpr int WH = G::WH, WW = G::WW, TW = 2 * WW - 1;
    for (int i = gtid; i < 4 * T * T; i += gsz) {
        const int variant = i / (T * T), key = (i / T) % T, q = i % T;
        const int ky = key / WW, kx = key % WW, qy = q / WW, qx = q % WW;
        float v = p.attn.bias_table[(ky - qy + WH - 1) * TW + (kx - qx + WW - 1)];
        const bool my = (variant & 2) && ((ky >= WH - WH / 2) != (qy >= WH - WH / 2));
        const bool mx = (variant & 1) && ((kx >= WW - WW / 2) != (qx >= WW - WW / 2));
        if (my || mx) v = -1e10f;
        reinterpret_cast<float*>(dst + G::p_bias4)[i] = v * kLog2e;
    }
}

// ------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------
#define SWF_WINDOW_SHAPES(X) X(24, 96) X(24, 4)

// one persistent workgroup per CU (its LDS footprint admits exactly one)
static int num_cus() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <int C, int HID>
static int launch_t(const WinArgs& a, int nwin, hipStream_t stream) {
    using G = Geo<C, HID>;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&window_block_kernel<C, HID>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::l_total);
    });
    if (attr_err != hipSuccess) return fail(SWF_ERR_HIP, "hipFuncSetAttribute(window_block): %s", hipGetErrorString(attr_err));
    hipLaunchKernelGGL((window_block_kernel<C, HID>), dim3(std::min(nwin, num_cus())), dim3(512), G::l_total, stream, a);
    return check_launch("window_block");
}

template <int C, int HID>
static int pack_t(const PackArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL((pack_block_kernel<C, HID>), dim3(32, 2), dim3(256), 0, stream, a);
    return check_launch("pack_window_block");
}

static bool dims_match(const swf_block_desc& d, int C, int HID) {
    return d.attn.channels == C && d.hidden == HID && d.attn.heads == 8 && d.attn.head_dim * 8 == C && d.attn.win_h == 8 &&
           d.attn.win_w == 8;
}

bool window_block_supported(const swf_block_desc& d, int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || H % 8 || W % 8) return false;
#define X(C, HID) if (dims_match(d, C, HID)) return true;
    SWF_WINDOW_SHAPES(X)
#undef X
    return false;
}

size_t window_block_packed_bytes(const swf_block_desc& d) {
#define X(C, HID) if (dims_match(d, C, HID)) return align_up(Geo<C, HID>::p_total, 256);
    SWF_WINDOW_SHAPES(X)
#undef X
    return 0;
}

size_t window_block_workspace_bytes(const swf_block_desc& d, int B, int H, int W) {
    return window_block_supported(d, B, H, W) ? 2 * window_block_packed_bytes(d) : 0;
}

int pack_window_block(const swf_block_desc& d, const swf_block_stream_params& px, const swf_block_stream_params& py,
                      void* packed_x, void* packed_y, hipStream_t stream) {
    PackArgs a;
    a.p[0] = px; a.p[1] = py;
    a.dst[0] = static_cast<char*>(packed_x); a.dst[1] = static_cast<char*>(packed_y);
    a.head_dim = d.attn.head_dim;
#define X(C, HID) if (dims_match(d, C, HID)) return pack_t<C, HID>(a, stream);
    SWF_WINDOW_SHAPES(X)
#undef X
    return fail(SWF_ERR_UNSUPPORTED, "pack_window_block: C=%d hidden=%d not covered", d.attn.channels, d.hidden);
}

int launch_window_block(const swf_block_desc& d, const void* packed_x, const void* packed_y, const float* x_in,
                        const float* y_in, float* x_out, float* y_out, int B, int H, int W, hipStream_t stream) {
    WinArgs a;
    a.in[0] = x_in; a.in[1] = y_in; a.out[0] = x_out; a.out[1] = y_out;
    a.packed[0] = static_cast<const char*>(packed_x); a.packed[1] = static_cast<const char*>(packed_y);
    a.B = B; a.H = H; a.W = W; a.shift = d.attn.shift; a.cross = d.cross;
    const int nwin = B * (H / 8) * (W / 8);
#define X(C, HID) if (dims_match(d, C, HID)) return launch_t<C, HID>(a, nwin, stream);
    SWF_WINDOW_SHAPES(X)
#undef X
    return fail(SWF_ERR_UNSUPPORTED, "window_block: C=%d hidden=%d not covered", d.attn.channels, d.hidden);
}

}  // namespace swf
