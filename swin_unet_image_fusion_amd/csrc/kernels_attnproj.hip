// Level-4 attention tail in one launch (C = 384 = 8 heads of 48; 8x8 windows, or 7x7 on the same 8x8 token grid):
//   out = res + bias + Attention(Q, K, V) . Wp^T          (a001:459-472 behind the Q/K/V GEMM; a004:29-33)
// One workgroup = (window, third of the output channels, stream), 512 threads: wave h is head h during the attention and
// (column tile h & 3, k half h >> 2) during the projection.
//
//  * Q and K fragments are 16-byte loads straight from the GEMM's fp16 rows (token = lane, 8 channels per lane half): they are the
//    B / A operands of S^T = K . Q^T as they lie, no LDS image.  V goes through a per-head transposed LDS image (the A operand of
//    O^T = V^T . P wants 8 consecutive keys per lane), written with the key permutation that makes the score accumulators P's
//    fragments (vt_pos, as attn_core_mfma_kernel).
//  * The relative-position bias (+ shift mask, + padding-key penalty) is the same for all heads: the workgroup builds the
//    [query tile][key tile][register][lane] matrix once in LDS, each wave reads its accumulator initialisers from there.
//  * O (normalised, split to bf16 hi / lo) is exchanged through a row-major LDS image laid over the V images; each wave produces
//    a 32-column x 64-token tile of O . Wp^T over half of k on the bf16x3 MFMA path (Wp's rows as A fragments from L2, all
//    requested before the exchange), and the k halves are added through LDS in fixed order.
//  * The three column thirds recompute the window's attention (6 + 8 MFMAs per head and query tile): cheaper than a second launch
//    and the O planes' round trip (attention core 8.2 us + projection GEMM 12.5 us -> one launch).
#include "kernels_attnproj.h"

#include <cstdlib>

#include "win_frag.h"

namespace swf {
namespace {

using namespace wf;
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kC = 384, kD = 48, kHeads = 8, kNS = 3, kVRS = 72, kLDO = kC + 8;
constexpr size_t kVtBytes = size_t(kHeads) * kD * kVRS * 2, kBiasBytes = 2 * 2 * 16 * 64 * 4;
constexpr size_t kOBytes = size_t(2) * 64 * kLDO * 2;
constexpr size_t kXchBytes = size_t(8) * 16 * 64 * 4;   // projection partials of the two k halves
static_assert(kVtBytes + kBiasBytes <= kOBytes, "the O image covers the attention-phase images");
constexpr size_t kLds = kOBytes + kXchBytes;

struct ApDev {
    const f16* q[2]; const f16* k[2]; const f16* v[2];
    const bf16* wp_hi[2]; const bf16* wp_lo[2];
    const float* pbias[2]; const float* table[2];
    const float* res[2]; float* out[2];
    int B, H, W, shift;
};

// position of key `tok` in a V^T image row: the order in which the score accumulators hold the keys (kernels_window.hip)
__device__ __forceinline__ int vt_pos(int tok) {
    const int k16 = tok & 15;
    const int e = ((k16 >> 3) << 2) | (k16 & 3);
    const int h = (k16 >> 2) & 1;
    return (tok & 48) | (h << 3) | e;
}

#ifdef AP_PROBE   // tools/ap_probe.hip: wall-clock stamps (10 ns) of workgroup (AP_PROBE, 0, 0), wave 0
__device__ unsigned long long ap_probe[16];
#define AP_STAMP(i) do { if (blockIdx.x == AP_PROBE && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) ap_probe[i] = wall_clock64(); } while (0)
#else
#define AP_STAMP(i) do { } while (0)
#endif

template <int WS>
__global__ __launch_bounds__(512) void attn_proj_kernel(ApDev a) {
    static_assert(WS == 7 || WS == 8, "window side");
    constexpr int C = kC, D = kD, VRS = kVRS, LDO = kLDO, TW = 2 * WS - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* vt = reinterpret_cast<f16*>(smem);                            // [head][channel 48][VRS]
    float* btile = reinterpret_cast<float*>(smem + kVtBytes);          // [query tile][key tile][register][lane]
    bf16* o_hi = reinterpret_cast<bf16*>(smem);                        // [token 64][LDO], laid over vt / btile after the attention
    bf16* o_lo = o_hi + 64 * LDO;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hf = lane >> 5;
    const int H = a.H, W = a.W, nwx = W / WS, nwy = H / WS, npi = nwx * nwy;
    const int win = blockIdx.x, ns = blockIdx.y, st = blockIdx.z;
    const int b = win / npi, wrem = win - b * npi, wy = wrem / nwx, wx = wrem - wy * nwx;
    const int sh = a.shift ? WS / 2 : 0;
    auto padding = [](int t) { return WS != 8 && ((t >> 3) >= WS || (t & 7) >= WS); };
    // image token of window token t; a padding token borrows a neighbour's row (finite values; masked as key, not stored as query)
    auto tok_index = [&](int t) {
        const int ty = WS == 8 ? (t >> 3) : min(t >> 3, WS - 1), tx = WS == 8 ? (t & 7) : min(t & 7, WS - 1);
        int oy = wy * WS + ty + sh, ox = wx * WS + tx + sh;
        oy = oy >= H ? oy - H : oy;
        ox = ox >= W ? ox - W : ox;
        return (size_t)((b * H + oy) * W + ox);
    };

    AP_STAMP(0);
    // ---- operand loads of head `wave`, all requested before anything is consumed ----
    const int head = wave;
    const f16* qg = a.q[st] + head * D + 8 * hf;
    const f16* kg = a.k[st] + head * D + 8 * hf;
    const size_t t0 = tok_index(r) * C, t1 = tok_index(32 + r) * C;
    u32x4 kf[2][3], qf[2][3], vrow[6];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        kf[0][ks] = *reinterpret_cast<const u32x4*>(kg + t0 + 16 * ks);
        kf[1][ks] = *reinterpret_cast<const u32x4*>(kg + t1 + 16 * ks);
        qf[0][ks] = *reinterpret_cast<const u32x4*>(qg + t0 + 16 * ks);
        qf[1][ks] = *reinterpret_cast<const u32x4*>(qg + t1 + 16 * ks);
    }
    {
        const f16* vg = a.v[st] + tok_index(lane) * C + head * D;
#pragma unroll
        for (int j = 0; j < 6; ++j) vrow[j] = *reinterpret_cast<const u32x4*>(vg + 8 * j);
    }
    // ---- bias (+ mask) matrix of this window, exp2 units: register i of key tile kt is key 32 kt + rho(i, lane half) ----
    {
        const bool last_row = a.shift && wy == nwy - 1, last_col = a.shift && wx == nwx - 1;
        const float* tab = a.table[st];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = tid + 512 * j, ln = idx & 63, reg = (idx >> 6) & 15, kt = (idx >> 10) & 1, qt = idx >> 11;
            const int key = 32 * kt + rho(reg, ln >> 5), q = 32 * qt + (ln & 31);
            const int ky = key >> 3, kx = key & 7, qy = q >> 3, qx = q & 7;
            const bool pad_k = ky >= WS || kx >= WS, pad_q = qy >= WS || qx >= WS;
            float v = (pad_k || pad_q) ? 0.f : tab[(ky - qy + WS - 1) * TW + (kx - qx + WS - 1)] * kLog2e;
            const bool my = last_row && ((ky >= WS - WS / 2) != (qy >= WS - WS / 2));
            const bool mx = last_col && ((kx >= WS - WS / 2) != (qx >= WS - WS / 2));
            btile[idx] = (my || mx || pad_k) ? -1e10f * kLog2e : v;
        }
    }
    AP_STAMP(1);
    // ---- V^T image of this head: lane = key ----
    {
        f16* vth = vt + head * D * VRS + vt_pos(lane);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const f16x8 v8 = __builtin_bit_cast(f16x8, vrow[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) vth[(8 * j + e) * VRS] = v8[e];
        }
    }
    AP_STAMP(2);
    __syncthreads();
    AP_STAMP(3);

    // ---- attention of head `wave`, both query tiles ----
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 o[2][2];
    float inv[2];
    const f16* vth = vt + head * D * VRS + 8 * hf;
    const int c1 = 32 + r < D ? 32 + r : D - 1;   // rows 48..63 of the second channel tile repeat channel 47 (never stored)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        f32x16 acc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[kt][i] = btile[((qb * 2 + kt) * 16 + i) * 64 + lane];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) acc[kt] = mfma_f16(kf[kt][ks], qf[qb][ks], acc[kt]);
        float mx = max3f(acc[0][0], acc[0][1], acc[1][0]);
        mx = max3f(mx, acc[1][1], acc[0][2]);
#pragma unroll
        for (int i = 3; i < 16; i += 2) mx = max3f(mx, acc[0][i], acc[0][i + 1 < 16 ? i + 1 : i]);
#pragma unroll
        for (int i = 2; i < 16; i += 2) mx = max3f(mx, acc[1][i], acc[1][i + 1]);
        mx = max_halves(mx);
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[kt][i] = __builtin_amdgcn_exp2f(acc[kt][i] - mx);
                l += acc[kt][i];
            }
        l = sum_halves(l);
        o[qb][0] = zero16;
        o[qb][1] = zero16;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float p[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) p[e] = acc[kt][8 * s2 + e];
                const u32x4 pf = pack8_f16(p);
                const u32x4 va0 = *reinterpret_cast<const u32x4*>(vth + r * VRS + kt * 32 + s2 * 16);
                const u32x4 va1 = *reinterpret_cast<const u32x4*>(vth + c1 * VRS + kt * 32 + s2 * 16);
                o[qb][0] = mfma_f16(va0, pf, o[qb][0]);
                o[qb][1] = mfma_f16(va1, pf, o[qb][1]);
            }
        inv[qb] = 1.0f / l;
    }

    AP_STAMP(4);
    // ---- projection operands go out now and fly under the O exchange: wave (ct, kh) owns column tile ct for BOTH token halves over
    //      k half kh (each Wp fragment crosses the memory pipe once per workgroup; all 24 of a wave are requested at once — with a
    //      4-deep ring the 24-step k loop waited six L2 round trips, 5.2 us), and finishes token half kh in the epilogue ----
    const int ct = wave & 3, kh = wave >> 2, col0 = ns * (C / kNS) + 32 * ct;
    constexpr int KSH = C / 32;   // k-steps of one k half
    u32x4 wfh[KSH], wfl[KSH];
    {
        // fragment-major image: block (32-row tile, k16 step) = one contiguous 1-KB load, lane l at bytes [16 l, 16 l + 16)
        // (row-strided 16-byte loads from the nn.Linear layout ran at 32 GB/s per CU here: 6 us for the 24 fragments of a wave)
        const size_t blk0 = (size_t)(col0 / 32) * (C / 16) + KSH * kh;
        const bf16* wh = a.wp_hi[st] + blk0 * 512 + lane * 8;
        const bf16* wl = a.wp_lo[st] + blk0 * 512 + lane * 8;
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
            wfh[ks] = *reinterpret_cast<const u32x4*>(wh + 512 * ks);
            wfl[ks] = *reinterpret_cast<const u32x4*>(wl + 512 * ks);
        }
    }
    const int t = 32 * kh + r;   // the token this lane finishes
    const size_t orow = tok_index(t) * C + col0 + 4 * hf;
    f32x4 rv[4], bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        rv[g] = *reinterpret_cast<const f32x4*>(a.res[st] + orow + 8 * g);
        bv[g] = a.pbias[st] ? *reinterpret_cast<const f32x4*>(a.pbias[st] + col0 + 8 * g + 4 * hf) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();   // every wave has left the V images and the bias tile
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = mt * 32 + 8 * g + 4 * hf;   // registers 4g..4g+3 of a lane: 4 consecutive channels of its query
                if (c < D) {
                    bf16x4 h4, l4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = o[qb][mt][4 * g + j] * inv[qb];
                        h4[j] = (bf16)v;
                        l4[j] = (bf16)(v - (float)h4[j]);
                    }
                    *reinterpret_cast<bf16x4*>(o_hi + (32 * qb + r) * LDO + head * D + c) = h4;
                    *reinterpret_cast<bf16x4*>(o_lo + (32 * qb + r) * LDO + head * D + c) = l4;
                }
            }
    __syncthreads();
    AP_STAMP(5);

    // ---- out^T[column][token] tiles (ct, both halves), partial over k half kh = Wp[col0.., k half] . O^T ----
    f32x16 acc[2] = {zero16, zero16};
    {
        const bf16* oh = o_hi + r * LDO + (C / 2) * kh + 8 * hf;
        const bf16* ol = o_lo + r * LDO + (C / 2) * kh + 8 * hf;
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
            const u32x4 bh0 = *reinterpret_cast<const u32x4*>(oh + 16 * ks), bl0 = *reinterpret_cast<const u32x4*>(ol + 16 * ks);
            const u32x4 bh1 = *reinterpret_cast<const u32x4*>(oh + 32 * LDO + 16 * ks), bl1 = *reinterpret_cast<const u32x4*>(ol + 32 * LDO + 16 * ks);
            acc[0] = mma3(wfh[ks], wfl[ks], bh0, bl0, acc[0]);
            acc[1] = mma3(wfh[ks], wfl[ks], bh1, bl1, acc[1]);
        }
    }
    // ---- the two k halves meet through LDS: each wave hands over the token half it does not finish.  k half 0 + k half 1 in both
    //      finishing waves (a + b == b + a bit for bit) ----
    float* xch = reinterpret_cast<float*>(smem + kOBytes);   // [wave][register][lane]
#pragma unroll
    for (int i = 0; i < 16; ++i) xch[(wave * 16 + i) * 64 + lane] = kh ? acc[0][i] : acc[1][i];
    __syncthreads();
    AP_STAMP(6);
    f32x16 mine = kh ? acc[1] : acc[0];
#pragma unroll
    for (int i = 0; i < 16; ++i) mine[i] += xch[((wave ^ 4) * 16 + i) * 64 + lane];
    // ---- + bias + residual; registers 4g..4g+3 = 4 consecutive columns of token t ----
    if (padding(t)) return;
    float* out = a.out[st] + orow;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v = {mine[4 * g], mine[4 * g + 1], mine[4 * g + 2], mine[4 * g + 3]};
        v += bv[g];
        v += rv[g];
        *reinterpret_cast<f32x4*>(out + 8 * g) = v;
    }
    AP_STAMP(7);
}

}  // namespace

bool attnproj_supported(const swf_block_desc& d) {
    static const bool off = debug_env("SWF_NO_ATTNPROJ") != nullptr;   // A/B switch (tools)
    return !off && d.precision == SWF_PREC_FAST && d.attn.channels == kC && d.attn.heads == kHeads && d.attn.head_dim == kD &&
           d.attn.win_h == d.attn.win_w && (d.attn.win_h == 8 || d.attn.win_h == 7);
}

int launch_attnproj(const swf_block_desc& d, const AttnProjArgs& a, int nstream, hipStream_t stream) {
    const int ws = d.attn.win_h;
    if (!attnproj_supported(d) || a.H % ws || a.W % ws) return fail(SWF_ERR_UNSUPPORTED, "attnproj: shape not covered");
    if ((int64_t)a.B * a.H * a.W > INT32_MAX / kC) return fail(SWF_ERR_UNSUPPORTED, "attnproj: token count");
    static hipError_t attr_err = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_proj_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
        return e != hipSuccess ? e : hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_proj_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
    }();
    if (attr_err != hipSuccess) return fail(SWF_ERR_HIP, "hipFuncSetAttribute(attn_proj): %s", hipGetErrorString(attr_err));
    ApDev dv{};
    for (int s = 0; s < nstream; ++s) {
        if (!a.q[s] || !a.k[s] || !a.v[s] || !a.wp_hi[s] || !a.wp_lo[s] || !a.table[s] || !a.res[s] || !a.out[s])
            return fail(SWF_ERR_NULL, "attnproj: NULL operand (stream %d)", s);
        dv.q[s] = reinterpret_cast<const f16*>(a.q[s]); dv.k[s] = reinterpret_cast<const f16*>(a.k[s]); dv.v[s] = reinterpret_cast<const f16*>(a.v[s]);
        dv.wp_hi[s] = reinterpret_cast<const bf16*>(a.wp_hi[s]); dv.wp_lo[s] = reinterpret_cast<const bf16*>(a.wp_lo[s]);
        dv.pbias[s] = a.pbias[s]; dv.table[s] = a.table[s]; dv.res[s] = a.res[s]; dv.out[s] = a.out[s];
    }
    dv.B = a.B; dv.H = a.H; dv.W = a.W; dv.shift = a.shift;
    const int nwin = a.B * (a.H / ws) * (a.W / ws);
    if (ws == 8) hipLaunchKernelGGL((attn_proj_kernel<8>), dim3(nwin, kNS, nstream), dim3(512), kLds, stream, dv);
    else hipLaunchKernelGGL((attn_proj_kernel<7>), dim3(nwin, kNS, nstream), dim3(512), kLds, stream, dv);
    return check_launch("attn_proj");
}

}  // namespace swf
