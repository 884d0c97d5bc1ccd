// Fused patch layers of the fast tier (PatchMergingAndLinearLayer.forward a011:244-264 with both MyPadding steps,
// a006:167-187): ONE launch per layer and stream pair instead of gather + GEMM + LayerNorm (+ scatter).
//   encoder: 2x2 space-to-depth gather (both reflect pads folded into the index map) -> 1x1 conv -> LN -> ELU
//   decoder: crop -> 1x1 conv -> LN -> depth-to-space scatter -> ELU (+ U-Net skip add), cropped to the recorded size
// A 256-thread workgroup walks tiles of 64 tokens; a wave owns 16 tokens for the whole pipeline, so LayerNorm is a
// two-shuffle reduction over the MFMA output registers (transposed tiles: a lane holds 4 consecutive output channels
// of one token) and nothing but the input rows and the output rows touches HBM.  The weights (split-bf16 hi / lo,
// zero padded to the MFMA shape) are staged once per workgroup into LDS; the linear runs as bf16x3 on
// v_mfma_f32_16x16x32_bf16 (fp32-grade, see kernels_window.hip).  Bound: HBM (input read + output write).
#include "kernels_patch.h"

#include <algorithm>
#include <mutex>
#include <cstdlib>

namespace swf {

namespace {

using bf16 = __bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float elu_fast(float v) { return v > 0.f ? v : __builtin_amdgcn_exp2f(v * kLog2e) - 1.0f; }
__device__ __forceinline__ int reflect_br(int i, int n) { return i < n ? i : 2 * n - 2 - i; }   // bottom / right pad only

struct PatchArgs {
    const float* in[2]; float* out[2]; const float* skip[2];
    const float* w[2]; const float* bias[2]; const float* gamma[2]; const float* beta[2];
    int B, H, W, Cin;        // input map (decoder: the padded map Hp x Wp)
    int mh, mw, Hm, Wm;      // merge size; merged map (decoder: the cropped map the conv runs on)
    int Ho, Wo;              // encoder: window-padded merged map; decoder: output extent Hout x Wout
    int K, N, Cout;
    int M;                   // tokens per stream
    int svec;                // outputs (and the skip tensor) allow 16-byte accesses: 4 consecutive channels stay inside one pixel
};

template <int KS>
struct Frag { bf16x8 hi[KS], lo[KS]; };

template <int KS>
__device__ __forceinline__ void load_frag(Frag<KS>& f, const bf16* hi, const bf16* lo, int off) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        f.hi[ks] = *reinterpret_cast<const bf16x8*>(hi + off + ks * 32);
        f.lo[ks] = *reinterpret_cast<const bf16x8*>(lo + off + ks * 32);
    }
}

// KS: 32-deep k-steps covering K; NT: 16-wide tiles covering N; DEC: 0 encoder (merge), 1 decoder (unmerge);
// VEC: floats per gather load (4 needs Cin % 4 == 0 and 16-byte aligned tensors)
template <int KS, int NT, int DEC, int VEC>
__global__ __launch_bounds__(256) void patch_fused_kernel(PatchArgs a) {
    constexpr int KP = 32 * KS + 8;   // row stride (bf16): odd multiple of 16 B -> conflict-free ds_read_b128
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* w_hi = reinterpret_cast<bf16*>(smem);
    bf16* w_lo = w_hi + 16 * NT * KP;
    bf16* a_hi = w_lo + 16 * NT * KP;
    bf16* a_lo = a_hi + 64 * KP;
    float* vb = reinterpret_cast<float*>(a_lo + 64 * KP);   // [3][16*NT]: conv bias, LN gamma, LN beta (zero padded)

    const int s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int K = a.K, N = a.N;

    // ---- once per workgroup: weights -> LDS (zero padded to the MFMA shape), vectors, zeroed A image (its K padding
    //      columns are never written again) ----
    if ((K & 3) == 0 && (reinterpret_cast<uintptr_t>(a.w[s]) & 15) == 0) {
        const int K4 = K >> 2;
        for (int idx = tid; idx < N * K4; idx += 256) {
            const int n = idx / K4, k = (idx % K4) * 4;
            const float4 q = *reinterpret_cast<const float4*>(a.w[s] + (int64_t)n * K + k);
            const float v[4] = {q.x, q.y, q.z, q.w};
            bf16x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) { h[j] = (bf16)v[j]; l[j] = (bf16)(v[j] - (float)h[j]); }
            *reinterpret_cast<bf16x4*>(w_hi + n * KP + k) = h;
            *reinterpret_cast<bf16x4*>(w_lo + n * KP + k) = l;
        }
        // padding: columns K..32KS-1 of the live rows, and whole rows N..16NT-1
        const int padc = 32 * KS - K;
        for (int idx = tid; idx < N * padc; idx += 256) {
            const int n = idx / padc, k = K + idx % padc;
            w_hi[n * KP + k] = (bf16)0.f; w_lo[n * KP + k] = (bf16)0.f;
        }
        for (int idx = tid; idx < (16 * NT - N) * 32 * KS; idx += 256) {
            const int n = N + idx / (32 * KS), k = idx % (32 * KS);
            w_hi[n * KP + k] = (bf16)0.f; w_lo[n * KP + k] = (bf16)0.f;
        }
    } else {
        for (int idx = tid; idx < 16 * NT * 32 * KS; idx += 256) {
            const int n = idx / (32 * KS), k = idx % (32 * KS);
            const float v = (n < N && k < K) ? a.w[s][(int64_t)n * K + k] : 0.f;
            const bf16 h = (bf16)v;
            w_hi[n * KP + k] = h;
            w_lo[n * KP + k] = (bf16)(v - (float)h);
        }
    }
    for (int idx = tid; idx < 3 * 16 * NT; idx += 256) {
        const int which = idx / (16 * NT), n = idx % (16 * NT);
        const float* src = which == 0 ? a.bias[s] : (which == 1 ? a.gamma[s] : a.beta[s]);
        vb[idx] = n < N ? src[n] : 0.f;
    }
    for (int idx = tid; idx < 64 * KP / 8; idx += 256) {
        reinterpret_cast<uint4*>(a_hi)[idx] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4*>(a_lo)[idx] = make_uint4(0, 0, 0, 0);
    }

    const int ntiles = (a.M + 63) / 64;
    const float* in = a.in[s];
    const int grow = tid >> 2, gsub = tid & 3;   // gather: 4 threads per token row
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();   // the previous tile's fragment reads are done (first pass: the staging above is visible)
        // ---- gather the 64 token rows of A (K floats each) -> split-bf16 image; rows past M keep stale data (never stored) ----
        {
            const int t = tile * 64 + grow;
            bf16* rh = a_hi + grow * KP;
            bf16* rl = a_lo + grow * KP;
            auto put = [&](int64_t src, int kk) {
                if constexpr (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(in + src);
                    const float v[4] = {q.x, q.y, q.z, q.w};
                    bf16x4 h, l;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { h[j] = (bf16)v[j]; l[j] = (bf16)(v[j] - (float)h[j]); }
                    *reinterpret_cast<bf16x4*>(rh + kk) = h;
                    *reinterpret_cast<bf16x4*>(rl + kk) = l;
                } else {
                    const float v = in[src];
                    const bf16 h = (bf16)v;
                    rh[kk] = h;
                    rl[kk] = (bf16)(v - (float)h);
                }
            };
            if (t < a.M) {
                if constexpr (DEC) {
                    const int mx = t % a.Wm, t2 = t / a.Wm, my = t2 % a.Hm, b = t2 / a.Hm;
                    const int64_t base = (((int64_t)b * a.H + my) * a.W + mx) * a.Cin;
                    for (int c = gsub * VEC; c < a.Cin; c += 4 * VEC) put(base + c, c);
                } else {
                    const int ox = t % a.Wo, t2 = t / a.Wo, oy = t2 % a.Ho, b = t2 / a.Ho;
                    const int my = reflect_br(oy, a.Hm), mx = reflect_br(ox, a.Wm);   // window pad of the merged map
                    int koff = 0;
                    for (int ph = 0; ph < a.mh; ++ph) {
                        const int iy = reflect_br(my * a.mh + ph, a.H);               // merge pad of the input
                        for (int pw = 0; pw < a.mw; ++pw, koff += a.Cin) {
                            const int ix = reflect_br(mx * a.mw + pw, a.W);
                            const int64_t base = (((int64_t)b * a.H + iy) * a.W + ix) * a.Cin;
                            for (int c = gsub * VEC; c < a.Cin; c += 4 * VEC) put(base + c, koff + c);
                        }
                    }
                }
            }
        }
        // decoder: the U-Net skip values this lane will add (addresses depend on the token only) are requested now, so their
        // HBM round trip overlaps the conv and the LayerNorm instead of serialising the scatter
        float4 skv[DEC ? NT : 1];
        if constexpr (DEC) {
            const int tq = tile * 64 + wave * 16 + r16;
            if (a.svec && a.skip[s] && tq < a.M) {
                const int mx = tq % a.Wm, t2 = tq / a.Wm, my = t2 % a.Hm, b = t2 / a.Hm;
                int pq = (4 * g) / a.Cout, c = (4 * g) % a.Cout;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    skv[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (nt * 16 + 4 * g < N) {
                        int ph = 0, pw = pq;
                        while (pw >= a.mw) { pw -= a.mw; ++ph; }
                        const int y = my * a.mh + ph, xx = mx * a.mw + pw;
                        if (y < a.Ho && xx < a.Wo)
                            skv[nt] = *reinterpret_cast<const float4*>(a.skip[s] + (((int64_t)b * a.Ho + y) * a.Wo + xx) * a.Cout + c);
                    }
                    c += 16;
                    while (c >= a.Cout) { c -= a.Cout; ++pq; }
                }
            }
        }
        __syncthreads();

        // ---- Z^T tiles: D[channel 4g+j][token r16] = W . A^T, bf16x3 ----
        Frag<KS> x;
        load_frag<KS>(x, a_hi, a_lo, (wave * 16 + r16) * KP + 8 * g);
        f32x4 z[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            Frag<KS> wf;
            load_frag<KS>(wf, w_hi, w_lo, (nt * 16 + r16) * KP + 8 * g);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.lo[ks], x.hi[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.hi[ks], x.lo[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.hi[ks], x.hi[ks], acc, 0, 0, 0);
            }
            const float4 b4 = *reinterpret_cast<const float4*>(vb + nt * 16 + 4 * g);   // padded channels: W rows and bias are 0
            acc[0] += b4.x; acc[1] += b4.y; acc[2] += b4.z; acc[3] += b4.w;
            z[nt] = acc;
            if constexpr (NT > 3) asm volatile("" ::: "memory");   // keep the next tile's fragment loads from piling up in registers
        }
        // ---- LayerNorm over the N channels of a token (its 4 lanes x NT x 4 registers), eps 1e-5, biased variance ----
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sum += (z[nt][0] + z[nt][1]) + (z[nt][2] + z[nt][3]);
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum / (float)N;
        float var = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = z[nt][j] - mean;
                var += (nt * 16 + 4 * g + j < N) ? d * d : 0.f;
            }
        var += __shfl_xor(var, 16);
        var += __shfl_xor(var, 32);
        const float rstd = 1.0f / sqrtf(var / (float)N + 1e-5f);

        const int t = tile * 64 + wave * 16 + r16;
        if (t < a.M) {
            if constexpr (DEC) {
                const int mx = t % a.Wm, t2 = t / a.Wm, my = t2 % a.Hm, b = t2 / a.Hm;
                const float* skip = a.skip[s];
                // channel n = pq * Cout + c of the conv output is channel c of output pixel (my*mh + pq/mw, mx*mw + pq%mw);
                // (pq, c) advance incrementally with n0 = nt*16 + 4g: no per-tile integer divisions
                int pq = (4 * g) / a.Cout, c = (4 * g) % a.Cout;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n0 = nt * 16 + 4 * g;
                    if (n0 < N) {
                        const float4 gm = *reinterpret_cast<const float4*>(vb + 16 * NT + n0), bt = *reinterpret_cast<const float4*>(vb + 32 * NT + n0);
                        const float v[4] = {elu_fast((z[nt][0] - mean) * rstd * gm.x + bt.x), elu_fast((z[nt][1] - mean) * rstd * gm.y + bt.y),
                                            elu_fast((z[nt][2] - mean) * rstd * gm.z + bt.z), elu_fast((z[nt][3] - mean) * rstd * gm.w + bt.w)};
                        if (a.svec) {   // Cout % 4 == 0: the 4 channels belong to one output pixel
                            int ph = 0, pw = pq;
                            while (pw >= a.mw) { pw -= a.mw; ++ph; }
                            const int y = my * a.mh + ph, xx = mx * a.mw + pw;
                            if (y < a.Ho && xx < a.Wo) {
                                const int64_t o = (((int64_t)b * a.Ho + y) * a.Wo + xx) * a.Cout + c;
                                float4 q = make_float4(v[0], v[1], v[2], v[3]);
                                if (skip) { q.x += skv[nt].x; q.y += skv[nt].y; q.z += skv[nt].z; q.w += skv[nt].w; }
                                *reinterpret_cast<float4*>(a.out[s] + o) = q;
                            }
                        } else {
                            int pqj = pq, cj = c;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (n0 + j < N) {
                                    int ph = 0, pw = pqj;
                                    while (pw >= a.mw) { pw -= a.mw; ++ph; }
                                    const int y = my * a.mh + ph, xx = mx * a.mw + pw;
                                    if (y < a.Ho && xx < a.Wo) {
                                        const int64_t o = (((int64_t)b * a.Ho + y) * a.Wo + xx) * a.Cout + cj;
                                        a.out[s][o] = skip ? v[j] + skip[o] : v[j];
                                    }
                                }
                                if (++cj == a.Cout) { cj = 0; ++pqj; }
                            }
                        }
                    }
                    c += 16;
                    while (c >= a.Cout) { c -= a.Cout; ++pq; }
                    if constexpr (NT > 3) asm volatile("" ::: "memory");
                }
            } else {
                float* orow = a.out[s] + (int64_t)t * N;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n0 = nt * 16 + 4 * g;
                    if (n0 >= N) continue;
                    const float4 gm = *reinterpret_cast<const float4*>(vb + 16 * NT + n0), bt = *reinterpret_cast<const float4*>(vb + 32 * NT + n0);
                    const float v[4] = {elu_fast((z[nt][0] - mean) * rstd * gm.x + bt.x), elu_fast((z[nt][1] - mean) * rstd * gm.y + bt.y),
                                        elu_fast((z[nt][2] - mean) * rstd * gm.z + bt.z), elu_fast((z[nt][3] - mean) * rstd * gm.w + bt.w)};
                    if (a.svec) {   // N % 4 == 0
                        *reinterpret_cast<float4*>(orow + n0) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (n0 + j < N) orow[n0 + j] = v[j];
                    }
                }
            }
        }
    }
}

template <int KS, int NT, int DEC, int VEC>
int launch_cfg(const PatchArgs& a, int nstream, hipStream_t stream) {
    constexpr int KP = 32 * KS + 8;
    constexpr int lds = (16 * NT + 64) * KP * 2 * 2 + 3 * 16 * NT * 4;
    static std::once_flag once;   // > 64 KB of dynamic LDS needs the attribute once per kernel (thread-safe)
    static hipError_t attr_err = hipSuccess;
    if (lds > 65536) {
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_fused_kernel<KS, NT, DEC, VEC>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        });
        if (attr_err != hipSuccess) return fail(SWF_ERR_HIP, "patch_fused: cannot raise the dynamic LDS limit to %d B", lds);
    }
    const int ntiles = (a.M + 63) / 64;
    // a few tiles per workgroup amortise the weight staging; at most ~8 resident workgroups per CU
    const int per_cu = std::max(1, std::min(8, 160 * 1024 / lds));
    const int grid_x = std::max(1, std::min(ntiles, 256 * per_cu / nstream));   // one resident round over both streams
    hipLaunchKernelGGL((patch_fused_kernel<KS, NT, DEC, VEC>), dim3(grid_x, nstream), dim3(256), lds, stream, a);
    return check_launch("patch_fused");
}

// instantiated (k-steps, n-tiles) shapes: the model's levels (K = 4, 96, 192 -> N = 24, 48, 96 merging;
// K = 24, 48, 96 -> N = 4, 96, 192 unmerging) and whatever smaller shapes they cover
template <int DEC, int VEC>
int dispatch(const PatchArgs& a, int nstream, hipStream_t stream) {
    const int ks = (a.K + 31) / 32, nt = (a.N + 15) / 16;
    if (ks <= 1 && nt <= 1) return launch_cfg<1, 1, DEC, VEC>(a, nstream, stream);
    if (ks <= 1 && nt <= 2) return launch_cfg<1, 2, DEC, VEC>(a, nstream, stream);
    if (ks <= 2 && nt <= 6) return launch_cfg<2, 6, DEC, VEC>(a, nstream, stream);
    if (ks <= 3 && nt <= 3) return launch_cfg<3, 3, DEC, VEC>(a, nstream, stream);
    if (ks <= 3 && nt <= 12) return launch_cfg<3, 12, DEC, VEC>(a, nstream, stream);
    if (ks <= 6 && nt <= 6) return launch_cfg<6, 6, DEC, VEC>(a, nstream, stream);
    return fail(SWF_ERR_UNSUPPORTED, "patch_fused: K=%d N=%d", a.K, a.N);
}

}  // namespace

bool patch_fused_supported(int K, int N) {
    const int ks = (K + 31) / 32, nt = (N + 15) / 16;
    return (ks <= 3 && nt <= 12) || (ks <= 6 && nt <= 6);
}

int launch_patch_fused(const PatchFusedDesc& d, int nstream, hipStream_t stream) {
    if (!patch_fused_supported(d.K, d.N)) return fail(SWF_ERR_UNSUPPORTED, "patch_fused: K=%d N=%d", d.K, d.N);
    if (d.M <= 0 || d.M > INT32_MAX - 64) return fail(SWF_ERR_UNSUPPORTED, "patch_fused: token count");
    PatchArgs a{};
    uintptr_t bits = 0;
    for (int s = 0; s < nstream; ++s) {
        a.in[s] = d.in[s]; a.out[s] = d.out[s]; a.skip[s] = d.skip[s];
        a.w[s] = d.w[s]; a.bias[s] = d.bias[s]; a.gamma[s] = d.gamma[s]; a.beta[s] = d.beta[s];
        bits |= reinterpret_cast<uintptr_t>(d.in[s]) | reinterpret_cast<uintptr_t>(d.out[s]) | reinterpret_cast<uintptr_t>(d.skip[s]);
    }
    a.B = d.B; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.mh = d.mh; a.mw = d.mw; a.Hm = d.Hm; a.Wm = d.Wm; a.Ho = d.Ho; a.Wo = d.Wo;
    a.K = d.K; a.N = d.N; a.Cout = d.Cout; a.M = (int)d.M;
    // 16-byte gathers need runs of 4 channels inside one pixel (Cin % 4 == 0); 16-byte stores need 4 consecutive output
    // channels inside one pixel / row (decoder: Cout % 4 == 0, encoder: N % 4 == 0)
    const bool gvec = bits % 16 == 0 && d.Cin % 4 == 0;
    a.svec = bits % 16 == 0 && (d.decoder ? d.Cout % 4 == 0 : d.N % 4 == 0);
    if (d.decoder) return gvec ? dispatch<1, 4>(a, nstream, stream) : dispatch<1, 1>(a, nstream, stream);
    return gvec ? dispatch<0, 4>(a, nstream, stream) : dispatch<0, 1>(a, nstream, stream);
}

}  // namespace swf
