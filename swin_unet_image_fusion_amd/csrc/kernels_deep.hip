// Deep-level fast tier (see kernels_deep.h): bf16x3 GEMM on pre-split operand planes.
//
// gemm_sp_kernel<WM, WN>: 256 threads = 4 waves as 2 (M) x 2 (N); a wave owns (32*WM) x (32*WN) outputs as
// WM x WN tiles of v_mfma_f32_32x32x16_bf16.  Tiles are computed TRANSPOSED (weight rows are the MFMA A
// operand, tokens the B operand), so registers 4g..4g+3 of a lane are four consecutive output channels of
// one token: bias / residual / output move as 16-byte vectors (8-byte for the bf16 plane epilogue).
// K advances 32 per step.  A stage holds the hi and lo planes of the A and W tiles as 64-byte rows; the four
// 16-byte chunks of a row are XOR-swizzled with (row >> 2) & 3, which makes both the staging stores and the
// ds_read_b128 fragment reads bank-conflict free without padding (64 KB for the 128x128 tile, two stages).
// Global loads of step k+1 are issued before the MFMAs of step k and stored to the other stage after them:
// one barrier per step.
#include "kernels_deep.h"
#include "kernels_attnproj.h"
#include "kernels_deeppatch.h"
#include "kernels_window.h"
#include "kernels_mlp.h"
#include "kernels_qkvattn.h"

#include <algorithm>
#include <mutex>
#include <cstdlib>

#ifndef SWF_GEMM_ABL
#define SWF_GEMM_ABL 0   // tools/gemm_bench.hip: 1 no epilogue stores, 2 no loads in the K loop, 3 no MFMAs
#endif

namespace swf {

using bf16 = __bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

// ELU(alpha = 1) with the hardware exp2 (absolute error ~1e-7: fast-tier grade)
__device__ __forceinline__ float elu_1(float v) { return v > 0.f ? v : __builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.0f; }

__device__ __forceinline__ void split_f4(const float4 v, bf16x4& hi, bf16x4& lo) {
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (bf16)f[i];
        lo[i] = (bf16)(f[i] - (float)hi[i]);
    }
}

struct SpProbDev {
    const bf16* a_hi; const bf16* a_lo; const bf16* w_hi; const bf16* w_lo;
    const float* bias; const float* res; float* out; bf16* o_hi; bf16* o_lo;
};
struct SpBatchDev { SpProbDev p[kMaxProb]; float* scratch; float qscale; };

}  // namespace

template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_sp_kernel(SpBatchDev batch, int M, int N, int K, int ldo, int epi, int splitk,
                                                      int kchunk) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int PA = BM * 64, PW = BN * 64;   // bytes of one plane of one stage (32 bf16 per row)
    constexpr int STAGE = 2 * PA + 2 * PW;
    constexpr int NA = 2 * WM, NW = 2 * WN;     // 16-byte chunks a thread stages per step (hi and lo planes)
    constexpr int ORS = BN * 4 + 16;            // row stride of the epilogue staging tile (bytes)
    static_assert(64 * ORS <= 2 * STAGE, "epilogue staging tile does not fit");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 * STAGE bytes (set by the launcher)

    const int prob = blockIdx.z / splitk, slice = blockIdx.z % splitk;
    const SpProbDev pr = batch.p[prob];
    const int kbeg = slice * kchunk, kend = min(K, kbeg + kchunk);
    const int nsteps = (kend - kbeg) >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int brow = blockIdx.x * BM, bcol = blockIdx.y * BN;

    // ---- staging map: chunk id = tid + i*256 -> (plane, row, 16-byte column) ----
    const bf16* ga[NA];
    const bf16* gw[NW];
    int la[NA], lw[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int plane = i / WM, rem = tid + (i % WM) * 256, row = rem >> 2, c = rem & 3;
        const int grow = min(brow + row, M - 1);
        ga[i] = (plane ? pr.a_lo : pr.a_hi) + (int64_t)grow * K + kbeg + c * 8;
        la[i] = plane * PA + row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int plane = i / WN, rem = tid + (i % WN) * 256, row = rem >> 2, c = rem & 3;
        const int grow = min(bcol + row, N - 1);
        gw[i] = (plane ? pr.w_lo : pr.w_hi) + (int64_t)grow * K + kbeg + c * 8;
        lw[i] = 2 * PA + plane * PW + row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
    }

    f32x16 acc[WN][WM];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

    u32x4 ra[NA], rw[NW];
#define SWF_GLOAD(step)                                                                                    \
    {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const u32x4*>(ga[i] + (step) * 32); \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) rw[i] = *reinterpret_cast<const u32x4*>(gw[i] + (step) * 32); \
    }
#define SWF_LSTORE(stage)                                                                                  \
    {                                                                                                      \
        unsigned char* sb = lds + (stage) * STAGE;                                                         \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(sb + la[i]) = ra[i];      \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) *reinterpret_cast<u32x4*>(sb + lw[i]) = rw[i];      \
    }

    const int r = lane & 31, hf = lane >> 5, swz = (r >> 2) & 3;
    const int arow0 = (wm * 32 * WM + r) * 64, wrow0 = 2 * PA + (wn * 32 * WN + r) * 64;

    SWF_GLOAD(0);
    SWF_LSTORE(0);
    __syncthreads();
    for (int it = 0; it < nsteps; ++it) {
        const int cur = it & 1;
        const bool more = it + 1 < nsteps;
        if (more && SWF_GEMM_ABL != 2) SWF_GLOAD(it + 1);
        const unsigned char* base = lds + cur * STAGE;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int coff = ((2 * kh + hf) ^ swz) << 4;
            bf16x8 wh[WN], wl[WN], ah[WM], al[WM];
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
                wh[nt] = *reinterpret_cast<const bf16x8*>(base + wrow0 + nt * 32 * 64 + coff);
                wl[nt] = *reinterpret_cast<const bf16x8*>(base + wrow0 + PW + nt * 32 * 64 + coff);
            }
#pragma unroll
            for (int mt = 0; mt < WM; ++mt) {
                ah[mt] = *reinterpret_cast<const bf16x8*>(base + arow0 + mt * 32 * 64 + coff);
                al[mt] = *reinterpret_cast<const bf16x8*>(base + arow0 + PA + mt * 32 * 64 + coff);
            }
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int mt = 0; mt < WM; ++mt) {
                    if (SWF_GEMM_ABL == 3) { asm volatile("" :: "v"(wh[nt]), "v"(wl[nt]), "v"(ah[mt]), "v"(al[mt])); continue; }
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[nt], ah[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nt], al[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[nt], ah[mt], acc[nt][mt], 0, 0, 0);
                }
        }
        if (more) SWF_LSTORE(cur ^ 1);
        __syncthreads();
    }
#undef SWF_GLOAD
#undef SWF_LSTORE

    if (SWF_GEMM_ABL == 1) {
        float t = 0.f;
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) t += acc[nt][mt][i];
        if (t != 123456.f) return;
    }

    // ---- epilogue.  D[n][m]: register 4g+j of tile (nt, mt) is channel n = 32 nt + 8g + 4hf + j of token m = 32 mt + r.
    // Per mt, the workgroup's 64 token rows x BN channels go through an LDS tile (row stride ORS: conflict-free
    // 16-byte writes) and leave as whole rows: consecutive lanes = consecutive channels of one token. ----
    float* part = splitk > 1 ? batch.scratch + (int64_t)blockIdx.z * M * N : nullptr;
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        __syncthreads();   // all fragment reads of the last stage / the previous pass's tile reads are done
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wn * 32 * WN + nt * 32 + 8 * g + 4 * hf;
                *reinterpret_cast<float4*>(lds + (wm * 32 + r) * ORS + col * 4) =
                    make_float4(acc[nt][mt][4 * g], acc[nt][mt][4 * g + 1], acc[nt][mt][4 * g + 2], acc[nt][mt][4 * g + 3]);
            }
        __syncthreads();
        constexpr int C4 = BN / 4;   // float4 columns per row
#pragma unroll
        for (int i = 0; i < 64 * C4 / 256; ++i) {
            const int idx = tid + i * 256, row = idx / C4, c4 = idx % C4;
            const int m = brow + (row >> 5) * 32 * WM + mt * 32 + (row & 31), n = bcol + c4 * 4;
            if (m >= M || n >= N) continue;   // N % 4 == 0 (checked by the launcher)
            float4 v = *reinterpret_cast<const float4*>(lds + row * ORS + c4 * 16);
            if (part) {
                *reinterpret_cast<float4*>(part + (int64_t)m * N + n) = v;
                continue;
            }
            if (pr.bias) {
                const float4 b = *reinterpret_cast<const float4*>(pr.bias + n);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            if (epi == SP_EPI_QKV16) {
                const int which = prob % 3;
                if (which == 2) {
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    const f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                    *reinterpret_cast<f16x4*>(pr.o_hi + (int64_t)m * N + n) = h;
                } else {
                    // Q (pre-scaled) and K in f16 too: the Q.K^T operands are 8x closer to fp32 than in bf16 at the same MFMA rate
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    const float sc = which == 0 ? batch.qscale : 1.0f;
                    const f16x4 h = {(_Float16)(v.x * sc), (_Float16)(v.y * sc), (_Float16)(v.z * sc), (_Float16)(v.w * sc)};
                    *reinterpret_cast<f16x4*>(pr.o_hi + (int64_t)m * N + n) = h;
                }
            } else if (epi == SP_EPI_ELU_SPLIT) {
                v.x = elu_1(v.x); v.y = elu_1(v.y); v.z = elu_1(v.z); v.w = elu_1(v.w);
                bf16x4 hi, lo;
                split_f4(v, hi, lo);
                *reinterpret_cast<bf16x4*>(pr.o_hi + (int64_t)m * N + n) = hi;
                *reinterpret_cast<bf16x4*>(pr.o_lo + (int64_t)m * N + n) = lo;
            } else {
                const int64_t o = (int64_t)m * ldo + n;
                if (pr.res) {
                    const float4 rr = *reinterpret_cast<const float4*>(pr.res + o);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                *reinterpret_cast<float4*>(pr.out + o) = v;
            }
        }
    }
}

// fixed-order sum of the K slices + bias (+ residual) -> out
__global__ __launch_bounds__(256) void gemm_sp_reduce_kernel(SpBatchDev batch, int M, int N, int ldo, int splitk) {
    const SpProbDev pr = batch.p[blockIdx.y];
    const int64_t total = (int64_t)M * N, total4 = total >> 2;
    const float* part = batch.scratch + (int64_t)blockIdx.y * splitk * total;
    for (int64_t e4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e4 < total4; e4 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = e4 << 2;
        const int col = (int)(e % N);
        const int64_t row = e / N;
        float4 v = *reinterpret_cast<const float4*>(part + e);
        for (int s = 1; s < splitk; ++s) {
            const float4 t = *reinterpret_cast<const float4*>(part + s * total + e);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (pr.bias) {
            const float4 b = *reinterpret_cast<const float4*>(pr.bias + col);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        const int64_t o = row * ldo + col;
        if (pr.res) {
            const float4 rr = *reinterpret_cast<const float4*>(pr.res + o);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        *reinterpret_cast<float4*>(pr.out + o) = v;
    }
}

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, bf16* __restrict__ hi,
                                                           bf16* __restrict__ lo, int64_t n4) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        bf16x4 h, l;
        split_f4(reinterpret_cast<const float4*>(src)[e], h, l);
        reinterpret_cast<bf16x4*>(hi)[e] = h;
        reinterpret_cast<bf16x4*>(lo)[e] = l;
    }
}

// fp32 [R][K] (R % 32 == 0, K % 16 == 0) -> fragment-major split planes (see DeepWeights)
__global__ __launch_bounds__(256) void pack_fragmajor_kernel(const float* __restrict__ src, bf16* __restrict__ hi, bf16* __restrict__ lo,
                                                             int R, int K) {
    const int64_t total = (int64_t)R * K;
    const int ksteps = K / 16;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t block = e >> 9;
        const int within = (int)(e & 511), lane = within >> 3, j = within & 7, hf = lane >> 5, r = lane & 31;
        const int ks = (int)(block % ksteps), rt = (int)(block / ksteps);
        const float v = src[(int64_t)(32 * rt + r) * K + 16 * ks + 8 * hf + j];
        const bf16 h = (bf16)v;
        hi[e] = h;
        lo[e] = (bf16)(v - (float)h);
    }
}

bool gemm_sp_supported(int N, int K) { return N > 0 && K > 0 && N % 4 == 0 && K % 32 == 0; }

int gemm_sp_splitk_for(int K, int epi) { return (epi == SP_EPI_F32 && K >= 1024 && K % 128 == 0) ? 4 : 1; }

template <int WM, int WN>
static int launch_sp_cfg(dim3 grid, hipStream_t stream, const SpBatchDev& dev, int M, int N, int K, int ldo, int epi, int splitk,
                         int kchunk) {
    constexpr int lds_bytes = 2 * (2 * 64 * WM * 64 + 2 * 64 * WN * 64);   // two stages
    static std::once_flag once;   // > 64 KB of dynamic LDS needs the attribute once per kernel (thread-safe)
    static hipError_t attr_err = hipSuccess;
    if (lds_bytes > 65536) {
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sp_kernel<WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        });
        if (attr_err != hipSuccess) return fail(SWF_ERR_HIP, "gemm_sp: cannot raise the dynamic LDS limit to %d B", lds_bytes);
    }
    hipLaunchKernelGGL((gemm_sp_kernel<WM, WN>), grid, dim3(256), lds_bytes, stream, dev, M, N, K, ldo, epi, splitk, kchunk);
    return SWF_OK;
}

int launch_gemm_sp(const SpGemmBatch& batch, int nprob, int M, int N, int K, int ldo, int epi, hipStream_t stream) {
    if (M <= 0 || nprob <= 0 || nprob > kMaxProb) return fail(SWF_ERR_BAD_SHAPE, "gemm_sp: bad problem count / M");
    if (!gemm_sp_supported(N, K)) return fail(SWF_ERR_UNSUPPORTED, "gemm_sp: N=%d K=%d (need N %% 4 == 0, K %% 32 == 0)", N, K);
    if (epi == SP_EPI_F32 && ldo % 4) return fail(SWF_ERR_UNSUPPORTED, "gemm_sp: ldo=%d not a multiple of 4", ldo);
    SpBatchDev dev{};
    for (int i = 0; i < nprob; ++i) {
        const SpGemmProb& s = batch.p[i];
        dev.p[i] = SpProbDev{reinterpret_cast<const bf16*>(s.a_hi), reinterpret_cast<const bf16*>(s.a_lo),
                             reinterpret_cast<const bf16*>(s.w_hi), reinterpret_cast<const bf16*>(s.w_lo),
                             s.bias, s.res, s.out, reinterpret_cast<bf16*>(s.o_hi), reinterpret_cast<bf16*>(s.o_lo)};
    }
    dev.scratch = batch.scratch;
    dev.qscale = batch.qscale;
    const int splitk = gemm_sp_splitk_for(K, epi);
    if (splitk > 1 && (!batch.scratch || (int64_t)splitk * nprob * M * N > batch.scratch_floats))
        return fail(SWF_ERR_WORKSPACE, "gemm_sp: split-K scratch too small (%d slices of %d x %d x %d)", splitk, nprob, M, N);
    const int kchunk = K / splitk;   // multiple of 32 by the split rule
    // tile shape by workgroup count (does not change any element's summation order)
    auto count = [&](int bm, int bn) { return (int64_t)cdiv(M, bm) * cdiv(N, bn) * nprob * splitk; };
    const int64_t fill = 224;
    int cfg;
    if (count(128, 128) >= fill && N % 128 == 0) cfg = 0;
    else if (count(128, 64) >= fill) cfg = 1;
    else cfg = 2;
    static const int forced_cfg = [] { const char* e = debug_env("SWF_SP_CFG"); return e ? e[0] - '0' : -1; }();   // tools/gemm_bench.hip
    if (forced_cfg >= 0 && forced_cfg <= 2) cfg = forced_cfg;
    if (cfg == 0) {
        dim3 grid(cdiv(M, 128), cdiv(N, 128), nprob * splitk);
        SWF_TRY((launch_sp_cfg<2, 2>(grid, stream, dev, M, N, K, ldo, epi, splitk, kchunk)));
    } else if (cfg == 1) {
        dim3 grid(cdiv(M, 128), cdiv(N, 64), nprob * splitk);
        SWF_TRY((launch_sp_cfg<2, 1>(grid, stream, dev, M, N, K, ldo, epi, splitk, kchunk)));
    } else {
        dim3 grid(cdiv(M, 64), cdiv(N, 64), nprob * splitk);
        SWF_TRY((launch_sp_cfg<1, 1>(grid, stream, dev, M, N, K, ldo, epi, splitk, kchunk)));
    }
    SWF_TRY(check_launch("gemm_sp"));
    if (splitk > 1) {
        dim3 rgrid((unsigned)std::min<int64_t>(cdiv64((int64_t)M * N, 1024), 2048), nprob);
        hipLaunchKernelGGL(gemm_sp_reduce_kernel, rgrid, dim3(256), 0, stream, dev, M, N, ldo, splitk);
        return check_launch("gemm_sp_reduce");
    }
    return SWF_OK;
}

int launch_split_planes(const float* src, bf16_raw* hi, bf16_raw* lo, int64_t n, hipStream_t stream) {
    if (n % 4) return fail(SWF_ERR_UNSUPPORTED, "split_planes: %lld elements (need a multiple of 4)", (long long)n);
    const int64_t n4 = n / 4;
    dim3 grid((unsigned)std::min<int64_t>(cdiv64(n4, 256), 4096));
    hipLaunchKernelGGL(split_planes_kernel, grid, dim3(256), 0, stream, src, reinterpret_cast<bf16*>(hi), reinterpret_cast<bf16*>(lo), n4);
    return check_launch("split_planes");
}

// ---- packed weight image of one stream of one block ----------------------------------------------------
namespace {
struct DeepSizes { int64_t qkv, proj, w1, w2, fm, pfm, qfm, total; };
DeepSizes deep_sizes(const swf_block_desc& d) {
    const int64_t C = d.attn.channels, HD = (int64_t)d.attn.heads * d.attn.head_dim, hid = d.hidden;
    DeepSizes s;
    s.qkv = HD * C; s.proj = C * HD; s.w1 = hid * C; s.w2 = C * hid;
    s.fm = mlp_fused_supported((int)C, (int)hid) ? s.w1 + s.w2 : 0;   // fragment-major copies of fc1 | fc2
    s.pfm = ((attnproj_supported(d) && HD == C) || deep_proj_supported(d)) ? s.proj : 0;   // fragment-major copy of Wproj (kernels_attnproj.hip, launch_deep_proj)
    s.qfm = deep_qkv_supported(d) ? 3 * s.qkv : 0;                        // fragment-major Wq | Wk | Wv (kernels_deeppatch.hip)
    s.total = 3 * s.qkv + s.proj + s.w1 + s.w2 + s.fm + s.pfm + s.qfm;
    return s;
}
}  // namespace

bool deep_block_supported(const swf_block_desc& d) {
    const int C = d.attn.channels, HD = d.attn.heads * d.attn.head_dim, hid = d.hidden;
    return d.precision == SWF_PREC_FAST && C >= 128 && C % 32 == 0 && HD % 32 == 0 && hid % 32 == 0 && C <= 1024 &&
           (attn_core_mfma_supported(d.attn.win_h, d.attn.win_w, d.attn.head_dim) ||
            (attn_core_mfma16_supported(d.attn.win_h, d.attn.win_w, d.attn.head_dim) && d.attn.head_dim % 4 == 0));
}

static size_t deep_planes_bytes(const swf_block_desc& d) { return align_up((size_t)deep_sizes(d).total * 4, 256); }   // hi + lo planes, 2 bytes each

size_t deep_block_packed_bytes(const swf_block_desc& d) {
    if (!deep_block_supported(d)) return 0;
    return deep_planes_bytes(d) + qkvattn_packed_bytes(d);   // + the fused Q/K/V + attention kernel's section, where it applies
}

DeepWeights deep_block_views(const swf_block_desc& d, const void* packed) {
    const DeepSizes s = deep_sizes(d);
    const bf16_raw* hi = static_cast<const bf16_raw*>(packed);
    const bf16_raw* lo = hi + s.total;
    DeepWeights w;
    int64_t o = 0;
    w.q_hi = hi + o; w.q_lo = lo + o; o += s.qkv;
    w.k_hi = hi + o; w.k_lo = lo + o; o += s.qkv;
    w.v_hi = hi + o; w.v_lo = lo + o; o += s.qkv;
    w.p_hi = hi + o; w.p_lo = lo + o; o += s.proj;
    w.w1_hi = hi + o; w.w1_lo = lo + o; o += s.w1;
    w.w2_hi = hi + o; w.w2_lo = lo + o; o += s.w2;
    w.w1f_hi = w.w1f_lo = w.w2f_hi = w.w2f_lo = nullptr;
    if (s.fm) {
        w.w1f_hi = hi + o; w.w1f_lo = lo + o; o += s.w1;
        w.w2f_hi = hi + o; w.w2f_lo = lo + o; o += s.w2;
    }
    w.pf_hi = w.pf_lo = nullptr;
    if (s.pfm) { w.pf_hi = hi + o; w.pf_lo = lo + o; o += s.pfm; }
    w.qkvf_hi = w.qkvf_lo = nullptr;
    if (s.qfm) { w.qkvf_hi = hi + o; w.qkvf_lo = lo + o; o += s.qfm; }
    w.qa = qkvattn_packed_bytes(d) ? static_cast<const char*>(packed) + deep_planes_bytes(d) : nullptr;
    return w;
}

int pack_deep_block(const swf_block_desc& d, const swf_block_stream_params& p, void* packed, hipStream_t stream) {
    if (!deep_block_supported(d)) return fail(SWF_ERR_UNSUPPORTED, "pack_deep_block: unsupported block shape");
    const DeepSizes s = deep_sizes(d);
    bf16_raw* hi = static_cast<bf16_raw*>(packed);
    bf16_raw* lo = hi + s.total;
    const float* src[6] = {p.attn.q.weight, p.attn.k.weight, p.attn.v.weight, p.attn.proj.weight, p.fc1.weight, p.fc2.weight};
    const int64_t n[6] = {s.qkv, s.qkv, s.qkv, s.proj, s.w1, s.w2};
    int64_t o = 0;
    for (int i = 0; i < 6; ++i) {
        SWF_TRY(launch_split_planes(src[i], hi + o, lo + o, n[i], stream));
        o += n[i];
    }
    if (s.fm) {
        const int C = d.attn.channels, hid = d.hidden;
        const float* fsrc[2] = {p.fc1.weight, p.fc2.weight};
        const int R[2] = {hid, C}, K[2] = {C, hid};
        for (int i = 0; i < 2; ++i) {
            const int64_t n2 = (int64_t)R[i] * K[i];
            dim3 grid((unsigned)std::min<int64_t>(cdiv64(n2, 256), 4096));
            hipLaunchKernelGGL(pack_fragmajor_kernel, grid, dim3(256), 0, stream, fsrc[i], reinterpret_cast<bf16*>(hi + o),
                               reinterpret_cast<bf16*>(lo + o), R[i], K[i]);
            SWF_TRY(check_launch("pack_fragmajor"));
            o += n2;
        }
    }
    if (s.pfm) {
        const int C = d.attn.channels;
        dim3 grid((unsigned)std::min<int64_t>(cdiv64(s.pfm, 256), 4096));
        hipLaunchKernelGGL(pack_fragmajor_kernel, grid, dim3(256), 0, stream, p.attn.proj.weight, reinterpret_cast<bf16*>(hi + o),
                           reinterpret_cast<bf16*>(lo + o), C, C);
        SWF_TRY(check_launch("pack_fragmajor(proj)"));
        o += s.pfm;
    }
    if (s.qfm) {   // the three images one after the other = the fragment-major image of the stacked matrix
        const int C = d.attn.channels, HD = d.attn.heads * d.attn.head_dim;
        const float* qsrc[3] = {p.attn.q.weight, p.attn.k.weight, p.attn.v.weight};
        for (int i = 0; i < 3; ++i) {
            dim3 grid((unsigned)std::min<int64_t>(cdiv64(s.qkv, 256), 4096));
            hipLaunchKernelGGL(pack_fragmajor_kernel, grid, dim3(256), 0, stream, qsrc[i], reinterpret_cast<bf16*>(hi + o),
                               reinterpret_cast<bf16*>(lo + o), HD, C);
            SWF_TRY(check_launch("pack_fragmajor(qkv)"));
            o += s.qkv;
        }
    }
    if (qkvattn_packed_bytes(d)) SWF_TRY(pack_qkvattn(d, p, static_cast<char*>(packed) + deep_planes_bytes(d), stream));
    return SWF_OK;
}

}  // namespace swf
