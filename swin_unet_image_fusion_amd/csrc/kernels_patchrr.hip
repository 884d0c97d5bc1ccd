// Register-resident patch layers (PatchMergingAndLinearLayer.forward a011:244-264 with both MyPadding steps a006:167-187):
//   encoder: 2x2 space-to-depth gather (both reflect pads folded into the index map) -> 1x1 conv -> LN -> ELU
//   decoder: crop -> 1x1 conv -> LN -> depth-to-space scatter -> ELU (+ U-Net skip add), cropped to the recorded size
// One wave owns 32 (or 64) tokens from the gather to the store: its token rows are split to bf16 hi / lo in registers and used
// as the B operand of v_mfma_f32_32x32x16_bf16 (a lane = one token, 8 consecutive input features per k-step: two 16-byte loads),
// the conv weights arrive as pre-packed A fragments straight from L2 (one 1-KB lane-linear load per fragment), and the output
// tile D[channel][token] leaves a token's channels in the registers of its two lanes: LayerNorm is an in-lane sum plus one
// cross-half exchange, ELU and the stores follow from the same registers.  No LDS tile, no workgroup barrier (the kernel in
// kernels_patch.hip staged weights per workgroup and 64-token tiles through LDS: 2 barriers per tile, 17-46 % of the HBM
// roofline).  Bound: HBM (input read + output write).
#include "kernels_patchrr.h"

#include <algorithm>

#include "win_frag.h"

namespace swf {
namespace {

using namespace wf;

struct PrrArgs {
    const float* in[2]; float* out[2]; const float* skip[2]; const char* packed[2];
    int B, H, W, Cin;        // input map (decoder: the padded map Hp x Wp)
    int mh, mw, Hm, Wm;      // merge size; merged map (decoder: the cropped map the conv runs on)
    int Ho, Wo;              // encoder: window-padded merged map; decoder: output extent Hout x Wout
    int K, N, Cout, M;
};

template <int KS, int NT>
struct PRR {
    static constexpr int NFRAG = NT * KS * 2;                       // [tile][k-step][hi,lo] x 1 KB
    static constexpr int NV = NT * 16;                              // accumulator registers of a lane half
    static constexpr size_t p_vec = size_t(NFRAG) * 1024;           // fp32 [conv bias, LN gamma, LN beta][lane half][NV], accumulator order
    static constexpr size_t p_total = p_vec + size_t(3) * 2 * NV * 4;
};

__device__ __forceinline__ int reflect_br(int i, int n) { return i < n ? i : 2 * n - 2 - i; }   // bottom / right pad only

// KS: 16-deep k-steps covering K; NT: 32-row tiles covering N; DEC: 0 merge, 1 unmerge; CIN1: encoder with one input channel
// (K = 4 scalar gathers); TPW: 32-token tiles a wave carries through one pass over the weight fragments
template <int KS, int NT, int DEC, int CIN1, int TPW>
__global__ __launch_bounds__(256) void patch_rr_kernel(PrrArgs a) {
    using P = PRR<KS, NT>;
    constexpr int NV = P::NV;
    __shared__ float lvec[3 * 2 * NV];
    const int s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    // (all loads before the first store: win_frag.h fill_vectors; the two "streams" here are the two halves of the one section)
    fill_vectors<3 * NV / 4, 256>(lvec, a.packed[s] + P::p_vec, a.packed[s] + P::p_vec + (3 * NV / 4) * 16, tid);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.packed[s]), 0, (int)P::p_total, 0x00020000);
    const unsigned loff = (unsigned)lane * 16u;
    auto WA = [&](int nt, int ks, int hl) {
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, loff, ((nt * KS + ks) * 2 + hl) * 1024, 0));
    };
    __syncthreads();
    const float* vbias = lvec + hf * NV;
    const float* vgam = lvec + 2 * NV + hf * NV;
    const float* vbet = lvec + 4 * NV + hf * NV;
    const float* in = a.in[s];
    const int K = a.K, N = a.N, Cin = a.Cin;
    const float invN = 1.0f / (float)N;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const int ngroup = (a.M + 32 * TPW - 1) / (32 * TPW);
    for (int grp = blockIdx.x * 4 + wave; grp < ngroup; grp += gridDim.x * 4) {
        // ---- the lane's token(s) and where their input features live ----
        int tok[TPW], base[TPW][4];   // encoder: element offsets of the 2x2 source pixels; decoder: base[][0] = the token row
        bool live[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const int t = (grp * TPW + u) * 32 + r;
            live[u] = t < a.M;
            const int tc = live[u] ? t : a.M - 1;   // rows past M work on a copy of the last row and store nothing
            tok[u] = tc;
            if constexpr (DEC) {
                const int mx = tc % a.Wm, t2 = tc / a.Wm, my = t2 % a.Hm, b = t2 / a.Hm;
                base[u][0] = ((b * a.H + my) * a.W + mx) * Cin;
                base[u][1] = base[u][2] = base[u][3] = 0;
            } else {
                const int ox = tc % a.Wo, t2 = tc / a.Wo, oy = t2 % a.Ho, b = t2 / a.Ho;
                const int my = reflect_br(oy, a.Hm), mx = reflect_br(ox, a.Wm);   // window pad of the merged map
                const int iy0 = reflect_br(2 * my, a.H), iy1 = reflect_br(2 * my + 1, a.H);   // merge pad of the input
                const int ix0 = reflect_br(2 * mx, a.W), ix1 = reflect_br(2 * mx + 1, a.W);
                base[u][0] = ((b * a.H + iy0) * a.W + ix0) * Cin;
                base[u][1] = ((b * a.H + iy0) * a.W + ix1) * Cin;
                base[u][2] = ((b * a.H + iy1) * a.W + ix0) * Cin;
                base[u][3] = ((b * a.H + iy1) * a.W + ix1) * Cin;
            }
        }
        // ---- Z^T tiles: D[channel][token] = W . X^T, split-bf16 x3; a weight fragment serves the wave's TPW token tiles ----
        f32x16 acc[TPW][NT];
#pragma unroll
        for (int u = 0; u < TPW; ++u)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[u][nt] = zero16;
        // All input features of the group are requested up front (2 x 16 bytes per lane and k-step), then the k loop runs on
        // registers.  Encoder: feature k0 = 16 ks + 8 hf sits in source pixel pqk at channel ck; both advance with ks (no divisions).
        float xin[TPW][KS][8];
        {
            int pqk = 0, ck = 8 * hf;
            if constexpr (!DEC && !CIN1) {
                while (ck >= Cin) { ck -= Cin; ++pqk; }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xin[u][ks][j] = 0.f;
                    if constexpr (CIN1) {   // K = 4: features = the 2x2 pixels, all in lane half 0 of the only k-step
                        if (hf == 0) {
                            xin[u][ks][0] = in[base[u][0]]; xin[u][ks][1] = in[base[u][1]];
                            xin[u][ks][2] = in[base[u][2]]; xin[u][ks][3] = in[base[u][3]];
                        }
                    } else {
                        const int k0 = 16 * ks + 8 * hf;   // 8 consecutive features: inside one source pixel (Cin % 8 == 0)
                        if (k0 < K) {
                            int off;
                            if constexpr (DEC) off = base[u][0] + k0;
                            else off = (pqk == 0 ? base[u][0] : pqk == 1 ? base[u][1] : pqk == 2 ? base[u][2] : base[u][3]) + ck;
                            const float4 x0 = *reinterpret_cast<const float4*>(in + off), x1 = *reinterpret_cast<const float4*>(in + off + 4);
                            xin[u][ks][0] = x0.x; xin[u][ks][1] = x0.y; xin[u][ks][2] = x0.z; xin[u][ks][3] = x0.w;
                            xin[u][ks][4] = x1.x; xin[u][ks][5] = x1.y; xin[u][ks][6] = x1.z; xin[u][ks][7] = x1.w;
                        }
                    }
                }
                if constexpr (!DEC && !CIN1) {
                    ck += 16;
                    while (ck >= Cin) { ck -= Cin; ++pqk; }
                }
            }
        }
        // decoder with several channels per pixel: output offsets of the lane's channel groups (-1 = nothing to store) and the U-Net
        // skip values, requested before the k loop where the registers allow it (their HBM round trip then overlaps the conv; loaded
        // one by one between the stores they would also serialise behind them: out may alias skip)
        constexpr bool DEC_VEC = DEC != 0;
        constexpr bool SKIP_EARLY = DEC_VEC && NT <= 3;
        int oidx[DEC_VEC ? TPW : 1][DEC_VEC ? NT : 1][4];
        float4 skv[DEC_VEC ? TPW : 1][DEC_VEC ? NT : 1][4];
        auto dec_offsets = [&](int u) {
            const int t = tok[u];
            const int mx = t % a.Wm, t2 = t / a.Wm, my = t2 % a.Hm, b = t2 / a.Hm;
            int pq = 0, c = 4 * hf;
            while (c >= a.Cout) { c -= a.Cout; ++pq; }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int yy = my * 2 + (pq >> 1), xx = mx * 2 + (pq & 1);
                    const bool ok = live[u] && 32 * nt + 8 * g + 4 * hf < N && yy < a.Ho && xx < a.Wo;
                    oidx[u][nt][g] = ok ? ((b * a.Ho + yy) * a.Wo + xx) * a.Cout + c : -1;
                    c += 8;
                    while (c >= a.Cout) { c -= a.Cout; ++pq; }
                }
        };
        auto load_skips = [&](int u) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    skv[u][nt][g] = (a.skip[s] && oidx[u][nt][g] >= 0) ? *reinterpret_cast<const float4*>(a.skip[s] + oidx[u][nt][g])
                                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        if constexpr (DEC_VEC) {
            if (a.Cout > 1) {
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    dec_offsets(u);
                    if constexpr (SKIP_EARLY) load_skips(u);
                }
            }
        }
        // weight fragments: a two-deep register ring, k-step ks + 1 requested before the MFMAs of k-step ks (the loads are
        // loop-invariant: the fences keep hipcc from hoisting them out of the token loop, or sinking them next to their use)
        u32x4 wr[2][NT][2];
        SWF_WF_FENCE();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { wr[0][nt][0] = WA(nt, 0, 0); wr[0][nt][1] = WA(nt, 0, 1); }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { wr[(ks + 1) & 1][nt][0] = WA(nt, ks + 1, 0); wr[(ks + 1) & 1][nt][1] = WA(nt, ks + 1, 1); }
            }
            SWF_WF_FENCE();
            u32x4 bh[TPW], bl[TPW];
#pragma unroll
            for (int u = 0; u < TPW; ++u) split8(xin[u][ks], bh[u], bl[u]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int u = 0; u < TPW; ++u) acc[u][nt] = mma3(wr[ks & 1][nt][0], wr[ks & 1][nt][1], bh[u], bl[u], acc[u][nt]);
        }
        // ---- per token: conv bias, LayerNorm over the N channels (registers x the two lanes of the token), ELU, store ----
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            float sum = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    acc[u][nt][i] += vbias[nt * 16 + i];   // padded channels: zero weights, zero bias
                    sum += acc[u][nt][i];
                }
            const float mean = sum_halves(sum) * invN;
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float d = acc[u][nt][i] - mean;
                    q += (32 * nt + rho(i, hf) < N) ? d * d : 0.f;
                }
            const float rstd = __builtin_amdgcn_rsqf(sum_halves(q) * invN + 1e-5f);
            const int t = tok[u];
            if constexpr (DEC_VEC) {
                if (a.Cout > 1) {
                    if constexpr (!SKIP_EARLY) load_skips(u);   // all of them before the first store
                }
            }
            // channel group n0 = 32 nt + 8 g + 4 hf .. +3 sits in registers 4g .. 4g+3
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n0 = 32 * nt + 8 * g + 4 * hf;
                    if (n0 < N) {
                        float y[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int i = 4 * g + j;
                            const float z = (acc[u][nt][i] - mean) * rstd * vgam[nt * 16 + i] + vbet[nt * 16 + i];
                            y[j] = z > 0.f ? z : __builtin_amdgcn_exp2f(z * kLog2e) - 1.0f;   // ELU(alpha = 1)
                        }
                        if constexpr (!DEC) {
                            if (live[u]) *reinterpret_cast<float4*>(a.out[s] + (int64_t)t * N + n0) = make_float4(y[0], y[1], y[2], y[3]);
                        } else if (a.Cout == 1) {   // N = 4 output pixels of one channel each (n0 = 0)
                            if (live[u]) {
                                const int mx = t % a.Wm, t2 = t / a.Wm, my = t2 % a.Hm, b = t2 / a.Hm;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const int yy = my * 2 + (j >> 1), xx = mx * 2 + (j & 1);
                                    if (yy < a.Ho && xx < a.Wo) {
                                        const int64_t o = ((int64_t)b * a.Ho + yy) * a.Wo + xx;
                                        a.out[s][o] = a.skip[s] ? y[j] + a.skip[s][o] : y[j];
                                    }
                                }
                            }
                        } else {
                            const int o = oidx[u][nt][g];
                            if (o >= 0) {
                                const float4 k4 = skv[u][nt][g];
                                *reinterpret_cast<float4*>(a.out[s] + o) = make_float4(y[0] + k4.x, y[1] + k4.y, y[2] + k4.z, y[3] + k4.w);
                            }
                        }
                    }
                }
        }
    }
}

struct PrrPackArgs { const float* w; const float* bias; const float* gamma; const float* beta; char* dst; int K, N; };

template <int KS, int NT>
__global__ __launch_bounds__(256) void patch_rr_pack_kernel(PrrPackArgs a) {
    using P = PRR<KS, NT>;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    // A fragment (tile nt, k-step ks): lane (r, hf) element e = W[32 nt + r][16 ks + 8 hf + e], zero beyond N / K
    for (int idx = gtid; idx < P::NFRAG * 512; idx += gsz) {
        const int f = idx >> 9, lane = (idx >> 3) & 63, e = idx & 7, r = lane & 31, hf = lane >> 5;
        const int hl = f & 1, ks = (f >> 1) % KS, nt = f / (2 * KS);
        const int n = 32 * nt + r, k = 16 * ks + 8 * hf + e;
        const float val = (n < a.N && k < a.K) ? a.w[(int64_t)n * a.K + k] : 0.f;
        const bf16 hi = (bf16)val;
        reinterpret_cast<bf16*>(a.dst)[idx] = hl ? (bf16)(val - (float)hi) : hi;
    }
    float* vec = reinterpret_cast<float*>(a.dst + P::p_vec);
    for (int i = gtid; i < 3 * 2 * P::NV; i += gsz) {
        const int which = i / (2 * P::NV), hf = (i / P::NV) & 1, j = i % P::NV, nt = j >> 4, reg = j & 15;
        const int n = 32 * nt + rho(reg, hf);
        const float* src = which == 0 ? a.bias : which == 1 ? a.gamma : a.beta;
        vec[i] = (n < a.N && src) ? src[n] : 0.f;
    }
}

int num_cus_prr() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

// the (k-steps, tiles) class of a layer, or false
bool shape_class(int decoder, int Cin, int Cout, int mh, int mw, int* ks, int* nt) {
    if (mh != 2 || mw != 2 || Cin <= 0 || Cout <= 0) return false;
    const int K = decoder ? Cin : 4 * Cin, N = decoder ? 4 * Cout : Cout;
    if (decoder) {
        if (Cin % 8 || !(Cout == 1 || Cout % 4 == 0)) return false;
    } else {
        if (!(Cin == 1 || Cin % 8 == 0) || Cout % 4) return false;
    }
    *ks = (K + 15) / 16;
    *nt = (N + 31) / 32;
    const int k = *ks, n = *nt;
    // the instantiated classes (launch_patch_rr): the model's levels 0-2 and whatever smaller shapes they cover
    if (decoder) return (k == 2 && n == 1) || (k == 3 && n == 3) || (k == 6 && n == 6);
    return (k == 1 && n == 1 && Cin == 1) || (k == 6 && n == 2) || (k == 12 && n == 3);
}

template <int KS, int NT, int DEC, int CIN1, int TPW>
int launch_t(const PrrArgs& a, int nstream, hipStream_t stream) {
    const int ngroup = (a.M + 32 * TPW - 1) / (32 * TPW);
    const int gx = std::max(1, std::min((ngroup + 3) / 4, 8 * num_cus_prr() / nstream));
    hipLaunchKernelGGL((patch_rr_kernel<KS, NT, DEC, CIN1, TPW>), dim3(gx, nstream), dim3(256), 0, stream, a);
    return check_launch("patch_rr");
}

}  // namespace

bool patch_rr_supported(int decoder, int Cin, int Cout, int mh, int mw) {
    int ks, nt;
    return shape_class(decoder, Cin, Cout, mh, mw, &ks, &nt);
}

size_t patch_rr_packed_bytes(int decoder, int Cin, int Cout, int mh, int mw) {
    int ks, nt;
    if (!shape_class(decoder, Cin, Cout, mh, mw, &ks, &nt)) return 0;
    return align_up(size_t(nt) * ks * 2 * 1024 + size_t(3) * 2 * nt * 16 * 4, 256);
}

int pack_patch_rr(int decoder, int Cin, int Cout, int mh, int mw, const float* weight, const float* bias, const float* gamma,
                  const float* beta, void* dst, hipStream_t stream) {
    int ks, nt;
    if (!shape_class(decoder, Cin, Cout, mh, mw, &ks, &nt)) return fail(SWF_ERR_UNSUPPORTED, "pack_patch_rr: shape not covered");
    PrrPackArgs a{weight, bias, gamma, beta, static_cast<char*>(dst), decoder ? Cin : 4 * Cin, decoder ? 4 * Cout : Cout};
#define SWF_PRR_PACK(KS_, NT_) if (ks == KS_ && nt == NT_) { hipLaunchKernelGGL((patch_rr_pack_kernel<KS_, NT_>), dim3(32), dim3(256), 0, stream, a); return check_launch("pack_patch_rr"); }
    SWF_PRR_PACK(1, 1) SWF_PRR_PACK(2, 1) SWF_PRR_PACK(3, 3) SWF_PRR_PACK(6, 2) SWF_PRR_PACK(6, 6) SWF_PRR_PACK(12, 3)
#undef SWF_PRR_PACK
    return fail(SWF_ERR_UNSUPPORTED, "pack_patch_rr: shape class");
}

int launch_patch_rr(const PatchFusedDesc& d, const void* const* packed, int nstream, hipStream_t stream) {
    int ks, nt;
    if (!shape_class(d.decoder, d.Cin, d.Cout, d.mh, d.mw, &ks, &nt)) return fail(SWF_ERR_UNSUPPORTED, "patch_rr: shape not covered");
    if (d.M <= 0 || d.M > INT32_MAX - 64) return fail(SWF_ERR_UNSUPPORTED, "patch_rr: token count");
    // element offsets are 32-bit inside the kernel
    const int64_t in_elems = (int64_t)d.B * d.H * d.W * d.Cin;
    const int64_t out_elems = d.decoder ? (int64_t)d.B * d.Ho * d.Wo * d.Cout : d.M * (int64_t)d.N;
    if (in_elems >= (int64_t(1) << 31) || out_elems >= (int64_t(1) << 31)) return fail(SWF_ERR_UNSUPPORTED, "patch_rr: tensor too large");
    PrrArgs a{};
    uintptr_t bits = 0;
    for (int s = 0; s < nstream; ++s) {
        if (!packed || !packed[s]) return fail(SWF_ERR_NULL, "patch_rr: packed image missing");
        a.in[s] = d.in[s]; a.out[s] = d.out[s]; a.skip[s] = d.skip[s]; a.packed[s] = static_cast<const char*>(packed[s]);
        bits |= reinterpret_cast<uintptr_t>(d.in[s]) | reinterpret_cast<uintptr_t>(d.out[s]) | reinterpret_cast<uintptr_t>(d.skip[s]) |
                reinterpret_cast<uintptr_t>(packed[s]);
    }
    if (bits % 16) return fail(SWF_ERR_UNSUPPORTED, "patch_rr: tensors must be 16-byte aligned");
    a.B = d.B; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.mh = d.mh; a.mw = d.mw; a.Hm = d.Hm; a.Wm = d.Wm; a.Ho = d.Ho; a.Wo = d.Wo;
    a.K = d.K; a.N = d.N; a.Cout = d.Cout; a.M = (int)d.M;
    if (!d.decoder) {
        if (ks == 1 && nt == 1 && d.Cin == 1) return launch_t<1, 1, 0, 1, 2>(a, nstream, stream);   // (4 tiles per wave: no faster; the decoder twin 20.6 -> 31.8 us)
        if (ks == 6 && nt == 2) return launch_t<6, 2, 0, 0, 2>(a, nstream, stream);
        if (ks == 12 && nt == 3) return launch_t<12, 3, 0, 0, 1>(a, nstream, stream);
    } else {
        if (ks == 2 && nt == 1) return launch_t<2, 1, 1, 0, 2>(a, nstream, stream);
        if (ks == 3 && nt == 3) return launch_t<3, 3, 1, 0, 1>(a, nstream, stream);
        if (ks == 6 && nt == 6) return launch_t<6, 6, 1, 0, 1>(a, nstream, stream);
    }
    return fail(SWF_ERR_UNSUPPORTED, "patch_rr: no instantiation for k-steps %d, tiles %d (decoder %d)", ks, nt, d.decoder);
}

}  // namespace swf
