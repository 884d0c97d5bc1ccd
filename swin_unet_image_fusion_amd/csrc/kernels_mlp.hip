// Deep-level fast tier: the MLP half of a BasicBlock (a004:29-38 around a003:46-50) in ONE launch:
//   out = x + fc2(ELU(fc1(LN2(x))))            per stream, C in {128, 192, 256, 384}, hidden % 128 == 0.
// Replaces LayerNorm + fc1 GEMM + fc2 GEMM (+ split-K reduce) and the round trip of the hidden activations.
//
// grid = (64-token tiles, hidden splits, streams), 256 threads = 4 waves (C = 384: 512 threads = 8 waves on a 256-wide chunk).  A workgroup normalises its 64 token rows
// once (LN2, fp32, two shuffles per row) into a split-bf16 LDS image, then walks its hidden range in chunks of 128:
//   fc1   wave w owns hidden rows [32w, 32w+32) of the chunk for all 64 tokens: H^T = W1 . xn^T on v_mfma_f32_32x32x16_bf16
//         (bf16x3: lo.hi + hi.lo + hi.hi, fp32 accumulate), + bias, ELU, split -> LDS image H [64][128] hi / lo
//   fc2   out^T[C][64] += W2[:, chunk] . H^T; the C/32 output tiles are dealt to the waves (whole tiles = both 32-token
//         halves; when C/32 % 4 == 2 the last two tiles are dealt as halves) and stay in accumulators across chunks.
// Weights never touch LDS: the packed image holds fc1 / fc2 a second time in MFMA-fragment-major order (DeepWeights),
// so a wave's A fragment is ONE contiguous 1-KB load (row-strided fragment loads from the nn.Linear layout cost 2x in
// this kernel: 32-byte pieces of 128-byte lines); fragments stream through a register ring D = 14 / 16 deep, issued
// ahead of their MFMAs across phase and chunk boundaries and pinned there with scheduling fences (hipcc otherwise
// sinks every prefetch next to its use).  The token-side fragments are read from LDS one k16 step ahead.
// Measured bound: LDS read bandwidth (every wave re-reads all 64 token rows per k16 step).  With S > 1 hidden splits a workgroup writes its partial out tile to
// scratch and mlp_reduce_kernel adds the S partials in fixed order + bias + residual (bit-reproducible; S depends on the
// layer shape only, never on the batch).
#include "kernels_mlp.h"

#include <algorithm>
#include <mutex>
#include <cstdlib>

#ifdef SWF_MLP_PROBE   // tools/mlp_probe.hip: wall-clock stamps of workgroup (0,0,0) wave 0 at the phase boundaries
__device__ unsigned long long swf_mlp_probe[64];
__device__ unsigned long long swf_mlp_wg[2 * 4096];   // entry / exit stamp of every workgroup (stamps 0 and 41)
#define SWF_PROBE(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); \
        if (blockIdx.x == SWF_MLP_PROBE && blockIdx.y == 0 && blockIdx.z == 0) swf_mlp_probe[i] = t_; \
        if ((i) == 0 || (i) == 41) swf_mlp_wg[2 * (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & 4095) + ((i) == 41)] = t_; } } while (0)
#else
#define SWF_PROBE(i) do { } while (0)
#endif

namespace swf {

namespace {

using bf16 = __bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float elu_fast(float v) { return v > 0.f ? v : __builtin_amdgcn_exp2f(v * kLog2e) - 1.0f; }

struct MlpArgs {
    const float* x[2]; float* out[2];
    const float* gamma[2]; const float* beta[2];
    const bf16* w1_hi[2]; const bf16* w1_lo[2]; const bf16* w2_hi[2]; const bf16* w2_lo[2];
    const float* b1[2]; const float* b2[2];
    float* scratch;          // [stream][split][M][C] partial sums when splits > 1
    int M, HID, splits, nchunks;
    // optional: LayerNorm of the finished rows (the NEXT block's LN1) as split planes, written by the reduce kernel
    const float* ln_gamma[2]; const float* ln_beta[2]; bf16* ln_hi[2]; bf16* ln_lo[2];
    // optional: the attention half's tail.  The rows entering LN2 are x + pbias + part0 + part1 (the output projection's bias and its
    // two head-group partial sums, qkv_attn_kernel); they are written to x1 (by the split-0 workgroups) and serve as the residual
    const float* part0[2]; const float* part1[2]; const float* pbias[2]; float* x1[2];
};

__device__ __forceinline__ void mma3(f32x16& acc, const bf16x8 wh, const bf16x8 wl, const bf16x8 bh, const bf16x8 bl) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bh, acc, 0, 0, 0);
}

// NW waves per workgroup = a hidden chunk of CW = 32 NW.  NW = 8 (C = 384): two waves per SIMD, each streaming its own fragments
// (the per-wave vector-memory rate, not L2, bounds the 4-wave kernel: DESIGN A.3), ONE 256-wide chunk per workgroup with the H image
// laid over the A image once fc1 has read it (A + H side by side would need 168 KB).
// TOK = token rows per workgroup (64, or 32 with NW = 6 at C = 192: twice the workgroups for the same weight bytes each — the level-3
// launches fill 128-256 of the 256 CUs with 64-token tiles, and a workgroup's prologue / epilogue rows are private HBM / MALL round
// trips that only more workgroups hide).
template <int C, int NW, int TOK = 64>
__global__ __launch_bounds__(64 * NW) void mlp_fused_kernel(MlpArgs a) {
    constexpr int NT = 64 * NW, CW = 32 * NW;    // threads, hidden chunk width
    constexpr int NTT = TOK / 32;                // 32-token tiles per workgroup
    static_assert(TOK == 32 || TOK == 64, "token tile");
    // threads per token row in the LayerNorm prologue / the row epilogue: a power of two, rows never straddle a wave
    constexpr int TPR = (NT / TOK == 4 || NT / TOK == 8) ? NT / TOK : 8;
    constexpr int LNT = TOK * TPR;               // threads that own a piece of a row (the rest only stream weights meanwhile)
    static_assert(LNT <= NT, "row threads");
    constexpr int KS1 = C / 16;                  // k16 steps of fc1
    constexpr int T = C / 32;                    // 32-channel output tiles of fc2
    constexpr int NF = T / NW, R = T % NW;       // whole tiles per wave; the R left-over tiles are dealt as 2R halves
    static_assert(R == 0 || 2 * R == NW, "the left-over output tiles must deal out as one half tile per wave");
    static_assert(R == 0 || NTT == 2, "half tiles need two token halves");
    constexpr int NH = R ? 1 : 0;                // + one half tile (one 32-token half)
    constexpr int NFR = NF + NH;                 // W2 fragments per k16 step
    constexpr int KS2 = CW / 16;                 // k16 steps per hidden chunk
    constexpr int NFRAG = KS1 + KS2 * NFR;       // weight fragments a wave streams per chunk
    constexpr int D = NW == 8 ? 8 : (NFRAG % 16 == 0) ? 16 : (NFRAG % 14 == 0) ? 14 : 12;   // ring depth: ~1 us of MFMA work ahead (L2 / MALL latency under load)
    static_assert(NFRAG % D == 0, "ring depth must divide the fragment count");
    constexpr bool ALIAS = NW == 8;              // H over A: one chunk per workgroup (launch_c checks)
    constexpr int LDA = C + 8, LDH = CW + 8;     // row strides (bf16): odd multiples of 16 B
    static_assert(!ALIAS || LDH <= LDA, "the H image must fit the A image");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* a_hi = reinterpret_cast<bf16*>(smem);
    bf16* a_lo = a_hi + TOK * LDA;
    bf16* h_hi = ALIAS ? a_hi : a_lo + TOK * LDA;
    bf16* h_lo = h_hi + TOK * LDH;

    const int tile = blockIdx.x, split = blockIdx.y, s = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hf = lane >> 5;
    const int HID = a.HID;

    // ---- the wave's weight fragment stream: per chunk KS1 fragments of W1 (its 32 hidden rows), then per k16 step of
    //      fc2 one fragment per output tile it owns.  The planes are fragment-major (DeepWeights): a fragment is one
    //      contiguous 1-KB block, lane l reads bytes [16l, 16l+16) ----
    // (buffer loads: lane offset in one VGPR, the fragment's byte offset in an SGPR — per-fragment 64-bit addresses cost the
    //  8-wave kernel 70 spilled registers and the 4-wave ones 60-230 AGPRs)
    const unsigned wbytes = (unsigned)C * (unsigned)HID * 2u;
    const __amdgpu_buffer_rsrc_t r1h = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w1_hi[s]), 0, (int)wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1l = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w1_lo[s]), 0, (int)wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2h = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w2_hi[s]), 0, (int)wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2l = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w2_lo[s]), 0, (int)wbytes, 0x00020000);
    const int loff = lane * 16;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int half_nt = T - R + (wave >> 1), half_tok = wave & 1;
    const int KSH = HID / 16;   // k16 steps of a whole W2 row
    int w2blk[NFR];             // first block of the W2 row tile
#pragma unroll
    for (int j = 0; j < NFR; ++j) w2blk[j] = (j < NF ? wave_u + NW * j : T - R + (wave_u >> 1)) * KSH;

    bf16x8 rh[D], rl[D];
    // fragment f (0 <= f < NFRAG after unrolling: a constant) of the chunk with hidden base hb -> ring slot f % D
    auto frag_load = [&](int f, int hb) {
        if (f < KS1) {
            const int off = (((hb >> 5) + wave_u) * KS1 + f) * 1024;
            rh[f % D] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r1h, loff, off, 0));
            rl[f % D] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r1l, loff, off, 0));
        } else {
            const int q = f - KS1, ks = q / NFR, j = q % NFR;
            const int off = (w2blk[j] + (hb >> 4) + ks) * 1024;
            rh[f % D] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r2h, loff, off, 0));
            rl[f % D] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r2l, loff, off, 0));
        }
    };

    SWF_PROBE(0);
    const int hb0 = split * a.nchunks * CW;
    // ---- LN2 of the 64 token rows -> split-bf16 image (4 threads per row).  Vector memory returns in order, so the token
    //      rows (and gamma / beta, which travel through the idle H buffer) are requested BEFORE the weight ring's first D
    //      fragments: the ring then fills during the LayerNorm arithmetic instead of delaying it (3.1 -> 1.x us at C=192). ----
    {
        const bool rowt = tid < LNT;   // this thread owns a piece of a token row
        const int row = rowt ? tid / TPR : 0, sub = tid % TPR;
        const int m = min(tile * TOK + row, a.M - 1);   // rows past M are computed on a clamped copy and never stored
        const float* xr = a.x[s] + (int64_t)m * C;
        constexpr int NV = C / (4 * TPR), CS = 4 * TPR;   // float4s per thread, column step
        float4 v[NV];
        if (rowt) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(xr + CS * i + 4 * sub);
            if (a.part0[s]) {   // attention residual: x + proj bias + the two head-group partials of the projection, fixed order
                const float* p0 = a.part0[s] + (int64_t)m * C;
                const float* p1 = a.part1[s] + (int64_t)m * C;
                float4 u0[NV], u1[NV], pb[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    u0[i] = *reinterpret_cast<const float4*>(p0 + CS * i + 4 * sub);
                    u1[i] = *reinterpret_cast<const float4*>(p1 + CS * i + 4 * sub);
                    pb[i] = *reinterpret_cast<const float4*>(a.pbias[s] + CS * i + 4 * sub);
                }
                const bool wr = split == 0 && tile * TOK + row < a.M;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    v[i].x = ((v[i].x + pb[i].x) + u0[i].x) + u1[i].x; v[i].y = ((v[i].y + pb[i].y) + u0[i].y) + u1[i].y;
                    v[i].z = ((v[i].z + pb[i].z) + u0[i].z) + u1[i].z; v[i].w = ((v[i].w + pb[i].w) + u0[i].w) + u1[i].w;
                    if (wr) *reinterpret_cast<float4*>(a.x1[s] + (int64_t)m * C + CS * i + 4 * sub) = v[i];
                }
            }
        }
        SWF_PROBE(50);
        float* gb = reinterpret_cast<float*>(ALIAS ? a_lo + TOK * LDA : h_hi);   // [2][C] fp32: gamma, beta
        float4 gbv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < C / 2) gbv = *reinterpret_cast<const float4*>((tid < C / 4 ? a.gamma[s] : a.beta[s] - C) + 4 * tid);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < D; ++f) frag_load(f, hb0);   // the ring's first fragments: in flight during the LayerNorm arithmetic
        __builtin_amdgcn_sched_barrier(0);
        SWF_PROBE(51);
        if (tid < C / 2) *reinterpret_cast<float4*>(gb + 4 * tid) = gbv;
        SWF_PROBE(52);
        float mean = 0.f, rstd = 0.f;
        if (rowt) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            if constexpr (TPR == 8) sum += __shfl_xor(sum, 4);
            mean = sum * (1.0f / C);
            float var = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
                var += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            var += __shfl_xor(var, 1);
            var += __shfl_xor(var, 2);
            if constexpr (TPR == 8) var += __shfl_xor(var, 4);
            rstd = 1.0f / sqrtf(var * (1.0f / C) + 1e-5f);
        }
        SWF_PROBE(53);
        __syncthreads();   // gamma / beta are in LDS
        SWF_PROBE(54);
        if (rowt) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = CS * i + 4 * sub;
                const float4 gm = *reinterpret_cast<const float4*>(gb + c), bt = *reinterpret_cast<const float4*>(gb + C + c);
                const float n[4] = {(v[i].x - mean) * rstd * gm.x + bt.x, (v[i].y - mean) * rstd * gm.y + bt.y,
                                    (v[i].z - mean) * rstd * gm.z + bt.z, (v[i].w - mean) * rstd * gm.w + bt.w};
                bf16x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) { h[j] = (bf16)n[j]; l[j] = (bf16)(n[j] - (float)h[j]); }
                *reinterpret_cast<bf16x4*>(a_hi + row * LDA + c) = h;
                *reinterpret_cast<bf16x4*>(a_lo + row * LDA + c) = l;
            }
        }
    }
    __syncthreads();
    SWF_PROBE(1);

    f32x16 acc2[NTT * NF + NH];
#pragma unroll
    for (int i = 0; i < NTT * NF + NH; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;

    for (int ch = 0; ch < a.nchunks; ++ch) {
        const int hb = hb0 + ch * CW;
        // next chunk's hidden base for the ring's run-ahead; past the last chunk the ring re-reads this chunk's first
        // fragments (dead loads) instead of branching: a conditional prefetch costs the counted waits their count
        const int hbn = ch + 1 < a.nchunks ? hb + CW : hb;
        // ---- fc1: H^T[32 hidden of this wave][64 tokens] ----
        f32x16 acc1[NTT];
#pragma unroll
        for (int t = 0; t < NTT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[t][e] = 0.f;
        // the chunk's fc1 bias goes out FIRST: vmcnt retires in order, so a bias load issued after the weight prefetches
        // would drain the whole ring at the phase boundary
        float4 b1v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b1v[g] = *reinterpret_cast<const float4*>(a.b1[s] + hb + 32 * wave + 8 * g + 4 * hf);
        __builtin_amdgcn_sched_barrier(0);
        // B fragments (tokens) are read one k16 step ahead of their MFMAs: the scheduling fences below pin the order
        // [ring prefetch, next step's LDS reads] -> [this step's MFMAs], so LDS latency hides behind the matrix work
        bf16x8 bh[2], bl[2];   // [token tile] (NTT of them live)
        {
            const int ko = 8 * hf;
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                bh[t] = *reinterpret_cast<const bf16x8*>(a_hi + (32 * t + r) * LDA + ko);
                bl[t] = *reinterpret_cast<const bf16x8*>(a_lo + (32 * t + r) * LDA + ko);
            }
        }
#pragma unroll
        for (int f = 0; f < KS1; ++f) {
            const bf16x8 wh = rh[f % D], wl = rl[f % D];
            if (f + D < NFRAG) frag_load(f + D, hb);
            else frag_load(f + D - NFRAG, hbn);
            bf16x8 ch_[2], cl_[2];
#pragma unroll
            for (int t = 0; t < NTT; ++t) { ch_[t] = bh[t]; cl_[t] = bl[t]; }
            if (f + 1 < KS1) {
                const int ko = 16 * (f + 1) + 8 * hf;
#pragma unroll
                for (int t = 0; t < NTT; ++t) {
                    bh[t] = *reinterpret_cast<const bf16x8*>(a_hi + (32 * t + r) * LDA + ko);
                    bl[t] = *reinterpret_cast<const bf16x8*>(a_lo + (32 * t + r) * LDA + ko);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch D fragments ahead: hipcc otherwise sinks it next to its use
#pragma unroll
            for (int t = 0; t < NTT; ++t) mma3(acc1[t], wh, wl, ch_[t], cl_[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        SWF_PROBE(2 + 4 * (ch & 7));
        if constexpr (ALIAS) __syncthreads();   // every wave has read its last A fragments: the H image may overwrite them
        // bias, ELU, split: register 4g+j of token half t is hidden 32w + 8g + 4hf + j of token 32t + r
#pragma unroll
        for (int t = 0; t < NTT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int hl = 32 * wave + 8 * g + 4 * hf;
                const float4 b = b1v[g];
                const float v[4] = {elu_fast(acc1[t][4 * g] + b.x), elu_fast(acc1[t][4 * g + 1] + b.y), elu_fast(acc1[t][4 * g + 2] + b.z),
                                    elu_fast(acc1[t][4 * g + 3] + b.w)};
                bf16x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) { h[j] = (bf16)v[j]; l[j] = (bf16)(v[j] - (float)h[j]); }
                *reinterpret_cast<bf16x4*>(h_hi + (32 * t + r) * LDH + hl) = h;
                *reinterpret_cast<bf16x4*>(h_lo + (32 * t + r) * LDH + hl) = l;
            }
        SWF_PROBE(3 + 4 * (ch & 7));
        __syncthreads();   // the whole H chunk is in place
        SWF_PROBE(4 + 4 * (ch & 7));
        // ---- fc2: out^T tiles of this wave += W2[:, chunk] . H^T ----
        bf16x8 xh, xl;   // the half tile's token half
        {
            const int ko = 8 * hf;
            if constexpr (NF > 0) {
#pragma unroll
                for (int t = 0; t < NTT; ++t) {
                    bh[t] = *reinterpret_cast<const bf16x8*>(h_hi + (32 * t + r) * LDH + ko);
                    bl[t] = *reinterpret_cast<const bf16x8*>(h_lo + (32 * t + r) * LDH + ko);
                }
            }
            if constexpr (NH) {
                xh = *reinterpret_cast<const bf16x8*>(h_hi + (32 * half_tok + r) * LDH + ko);
                xl = *reinterpret_cast<const bf16x8*>(h_lo + (32 * half_tok + r) * LDH + ko);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            bf16x8 wh[NFR], wl[NFR];
#pragma unroll
            for (int j = 0; j < NFR; ++j) {
                const int f = KS1 + ks * NFR + j;
                wh[j] = rh[f % D]; wl[j] = rl[f % D];
                if (f + D < NFRAG) frag_load(f + D, hb);
                else frag_load(f + D - NFRAG, hbn);
            }
            bf16x8 ch_[2], cl_[2];
#pragma unroll
            for (int t = 0; t < NTT; ++t) { ch_[t] = bh[t]; cl_[t] = bl[t]; }
            const bf16x8 cxh = xh, cxl = xl;
            if (ks + 1 < KS2) {
                const int ko = 16 * (ks + 1) + 8 * hf;
                if constexpr (NF > 0) {
#pragma unroll
                    for (int t = 0; t < NTT; ++t) {
                        bh[t] = *reinterpret_cast<const bf16x8*>(h_hi + (32 * t + r) * LDH + ko);
                        bl[t] = *reinterpret_cast<const bf16x8*>(h_lo + (32 * t + r) * LDH + ko);
                    }
                }
                if constexpr (NH) {
                    xh = *reinterpret_cast<const bf16x8*>(h_hi + (32 * half_tok + r) * LDH + ko);
                    xl = *reinterpret_cast<const bf16x8*>(h_lo + (32 * half_tok + r) * LDH + ko);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NFR; ++j) {
                if (j < NF) {
#pragma unroll
                    for (int t = 0; t < NTT; ++t) mma3(acc2[NTT * j + t], wh[j], wl[j], ch_[t], cl_[t]);
                } else {
                    mma3(acc2[NTT * NF], wh[j], wl[j], cxh, cxl);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        SWF_PROBE(5 + 4 * (ch & 7));
        __syncthreads();   // H is free for the next chunk
    }

    SWF_PROBE(40);
    // ---- epilogue: register 4g+j of (tile nt, token half t) is channel 32nt + 8g + 4hf + j of token 32t + r.  The out tile goes
    //      through the (now free) A image as fp32 rows and leaves as whole rows: consecutive lanes = consecutive channels
    //      (per-lane 16-byte stores at a row stride took 3-5 us here). ----
    constexpr int ORS = C + 4;   // row stride (floats): odd multiple of 16 B
    static_assert(TOK * ORS * 4 <= 2 * TOK * LDA * 2, "out tile must fit the A image");
    float* otile = reinterpret_cast<float*>(smem);
    float* part = a.splits > 1 ? a.scratch + ((int64_t)(s * a.splits + split) * a.M) * C : nullptr;
    // unsplit: the rows are final here.  TPR threads per token row, so the NEXT block's LN1 (optional) is a shuffle reduction over the
    // finished row and leaves as split planes from the same registers: no LayerNorm launch.  Everything the row epilogue reads from
    // global memory (residual row, fc2 bias, the next LayerNorm's gamma / beta) is requested HERE, ahead of the out-tile exchange:
    // cold parameter vectors cost a full HBM round trip each when they are requested where they are used (tools/mlp_probe.hip).
    const int erow = tid < LNT ? tid / TPR : 0, esub = tid % TPR;
    const int em = tile * TOK + erow;
    const bool live = !part && em < a.M && tid < LNT;
    constexpr int ENV = C / (4 * TPR), ECS = 4 * TPR;
    float4 exr[ENV], eb2[ENV], eg[ENV], ebt[ENV];
    const bool ln_out = !part && a.ln_hi[s] != nullptr;
    if (live) {
        const float* xres = (a.part0[s] ? a.x1[s] : a.x[s]) + (int64_t)em * C;   // (x1 holds this very thread's prologue stores: same mapping)
#pragma unroll
        for (int i = 0; i < ENV; ++i) {
            const int c = ECS * i + 4 * esub;
            exr[i] = *reinterpret_cast<const float4*>(xres + c);
            eb2[i] = *reinterpret_cast<const float4*>(a.b2[s] + c);
        }
        if (ln_out) {
#pragma unroll
            for (int i = 0; i < ENV; ++i) {
                const int c = ECS * i + 4 * esub;
                eg[i] = *reinterpret_cast<const float4*>(a.ln_gamma[s] + c);
                ebt[i] = *reinterpret_cast<const float4*>(a.ln_beta[s] + c);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NTT * NF + NH; ++i) {
        const int nt = i < NTT * NF ? wave + NW * (i / NTT) : half_nt;
        const int t = i < NTT * NF ? (i % NTT) : half_tok;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(otile + (32 * t + r) * ORS + 32 * nt + 8 * g + 4 * hf) =
                make_float4(acc2[i][4 * g], acc2[i][4 * g + 1], acc2[i][4 * g + 2], acc2[i][4 * g + 3]);
    }
    __syncthreads();
    constexpr int C4 = C / 4;
    if (part) {
#pragma unroll 4
        for (int idx = tid; idx < TOK * C4; idx += NT) {
            const int row = idx / C4, c = (idx % C4) * 4;
            const int m = tile * TOK + row;
            if (m >= a.M) continue;
            *reinterpret_cast<float4*>(part + (int64_t)m * C + c) = *reinterpret_cast<const float4*>(otile + row * ORS + c);
        }
    } else {
        float4 v[ENV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < ENV; ++i) {
            const int c = ECS * i + 4 * esub;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                float4 t = *reinterpret_cast<const float4*>(otile + erow * ORS + c);
                const float4 b = eb2[i], x = exr[i];
                t.x += b.x + x.x; t.y += b.y + x.y; t.z += b.z + x.z; t.w += b.w + x.w;
                *reinterpret_cast<float4*>(a.out[s] + (int64_t)em * C + c) = t;
                v[i] = t;
            }
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        if (ln_out) {
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            if constexpr (TPR == 8) sum += __shfl_xor(sum, 4);
            const float mean = sum * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < ENV; ++i) {
                const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            q += __shfl_xor(q, 1);
            q += __shfl_xor(q, 2);
            if constexpr (TPR == 8) q += __shfl_xor(q, 4);
            const float rstd = 1.0f / sqrtf(q * (1.0f / C) + 1e-5f);
            if (live) {
#pragma unroll
                for (int i = 0; i < ENV; ++i) {
                    const int c = ECS * i + 4 * esub;
                    const float4 g = eg[i], b = ebt[i];
                    const float n[4] = {(v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                        (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w};
                    bf16x4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { hi[j] = (bf16)n[j]; lo[j] = (bf16)(n[j] - (float)hi[j]); }
                    *reinterpret_cast<bf16x4*>(a.ln_hi[s] + (int64_t)em * C + c) = hi;
                    *reinterpret_cast<bf16x4*>(a.ln_lo[s] + (int64_t)em * C + c) = lo;
                }
            }
        }
    }
    SWF_PROBE(41);
}

// out = x + b2 + sum of the hidden-split partials, fixed order
__global__ __launch_bounds__(256) void mlp_reduce_kernel(MlpArgs a, int C) {
    const int s = blockIdx.y;
    const int64_t total = (int64_t)a.M * C, total4 = total >> 2;
    const float* part = a.scratch + (int64_t)s * a.splits * total;
    for (int64_t e4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e4 < total4; e4 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = e4 << 2;
        const int c = (int)(e % C);
        float4 v = *reinterpret_cast<const float4*>(part + e);
        for (int k = 1; k < a.splits; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(part + k * total + e);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const float4 b = *reinterpret_cast<const float4*>(a.b2[s] + c);
        const float4 x = *reinterpret_cast<const float4*>(a.x[s] + e);
        v.x += b.x + x.x; v.y += b.y + x.y; v.z += b.z + x.z; v.w += b.w + x.w;
        *reinterpret_cast<float4*>(a.out[s] + e) = v;
    }
}

// The same reduce, one token row per L lanes, followed by the next block's LN1 on the finished row (out_hi / out_lo planes, the
// operand format of its Q/K/V GEMMs): saves that block's LayerNorm launch.  Same arithmetic, in the same order, as
// layernorm_vec_kernel (kernels_generic.hip).
template <int NCH>
__global__ __launch_bounds__(256) void mlp_reduce_ln_kernel(MlpArgs a, int C, int L) {
    const int s = blockIdx.y;
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tok = gt / L;
    const int sub = (int)(gt % L);
    const bool live = tok < a.M;
    const int chunks = C >> 2;
    const int64_t total = (int64_t)a.M * C, row = (live ? tok : 0) * C;
    const float* part = a.scratch + (int64_t)s * a.splits * total + row;
    float4 v[NCH], gm[NCH], bt[NCH];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = sub + i * L;
        v[i] = gm[i] = bt[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && ch < chunks) {
            // all loads of the row go out together (the first six partials predicated, not looped), summed in fixed order
            float4 u[6];
#pragma unroll
            for (int k = 0; k < 6; ++k)
                u[k] = k < a.splits ? *reinterpret_cast<const float4*>(part + k * total + 4 * ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 b = *reinterpret_cast<const float4*>(a.b2[s] + 4 * ch);
            const float4 x = *reinterpret_cast<const float4*>(a.x[s] + row + 4 * ch);
            gm[i] = *reinterpret_cast<const float4*>(a.ln_gamma[s] + 4 * ch);
            bt[i] = *reinterpret_cast<const float4*>(a.ln_beta[s] + 4 * ch);
            float4 t = u[0];
#pragma unroll
            for (int k = 1; k < 6; ++k)
                if (k < a.splits) { t.x += u[k].x; t.y += u[k].y; t.z += u[k].z; t.w += u[k].w; }
            for (int k = 6; k < a.splits; ++k) {
                const float4 w = *reinterpret_cast<const float4*>(part + k * total + 4 * ch);
                t.x += w.x; t.y += w.y; t.z += w.z; t.w += w.w;
            }
            t.x += b.x + x.x; t.y += b.y + x.y; t.z += b.z + x.z; t.w += b.w + x.w;
            *reinterpret_cast<float4*>(a.out[s] + row + 4 * ch) = t;
            v[i] = t;
        }
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    for (int o = L >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (sub + i * L < chunks) {
            const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    for (int o = L >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = sub + i * L;
        if (ch < chunks) {
            const float4 g = gm[i], b = bt[i];
            const float r[4] = {(v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w};
            bf16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { hi[j] = (bf16)r[j]; lo[j] = (bf16)(r[j] - (float)hi[j]); }
            *reinterpret_cast<bf16x4*>(a.ln_hi[s] + row + 4 * ch) = hi;
            *reinterpret_cast<bf16x4*>(a.ln_lo[s] + row + 4 * ch) = lo;
        }
    }
}

template <int C, int NW, int TOK = 64>
int launch_c(const MlpArgs& a, int nstream, hipStream_t stream) {
    // A image + H image (NW = 4, 6) or A image with H laid over it + gamma / beta (NW = 8)
    constexpr int lds = NW == 8 ? TOK * (C + 8) * 2 * 2 + 2 * C * 4 : (TOK * (C + 8) + TOK * (32 * NW + 8)) * 2 * 2;
    if (NW == 8 && a.nchunks != 1) return fail(SWF_ERR_UNSUPPORTED, "mlp_fused: the 8-wave kernel takes one hidden chunk per workgroup");
    static std::once_flag once;   // > 64 KB of dynamic LDS needs the attribute once per kernel (thread-safe)
    static hipError_t attr_err = hipSuccess;
    if (lds > 65536) {
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<C, NW, TOK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        });
        if (attr_err != hipSuccess) return fail(SWF_ERR_HIP, "mlp_fused: cannot raise the dynamic LDS limit to %d B", lds);
    }
    dim3 grid((a.M + TOK - 1) / TOK, a.splits, nstream);
    hipLaunchKernelGGL((mlp_fused_kernel<C, NW, TOK>), grid, dim3(64 * NW), lds, stream, a);
    SWF_TRY(check_launch("mlp_fused"));
    MlpArgs ra = a;   // the reduce kernels' residual is the row that entered LN2
    for (int s = 0; s < nstream; ++s)
        if (a.part0[s]) ra.x[s] = a.x1[s];
    if (a.splits > 1 && a.ln_hi[0]) {
        constexpr int chunks = C / 4, L = chunks > 32 ? 64 : 32, NCH = (chunks + L - 1) / L;   // as launch_layernorm picks them
        dim3 rgrid((unsigned)cdiv64((int64_t)a.M * L, 256), nstream);
        hipLaunchKernelGGL((mlp_reduce_ln_kernel<NCH>), rgrid, dim3(256), 0, stream, ra, C, L);
        return check_launch("mlp_reduce_ln");
    }
    if (a.splits > 1) {
        dim3 rgrid((unsigned)std::min<int64_t>(cdiv64((int64_t)a.M * C, 1024), 2048), nstream);
        hipLaunchKernelGGL(mlp_reduce_kernel, rgrid, dim3(256), 0, stream, ra, C);
        return check_launch("mlp_reduce");
    }
    return SWF_OK;
}

}  // namespace

// 8 waves on one 256-wide hidden chunk per workgroup (C = 384; SWF_MLP8=0: the 4-wave kernel, tools only)
static bool mlp_wide(int C, int HID) {
    static const bool off = [] { const char* e = debug_env("SWF_MLP8"); return e && e[0] == '0'; }();
    return !off && C == 384 && HID % 256 == 0;
}

// C = 192 with a hidden width that is a multiple of 192: six waves on 192-wide hidden chunks, 32-token tiles, no hidden split
// (SWF_MLP_TOK32=0 under SWF_DEBUG_SWITCHES: the 64-token kernel, tools only).  The rule looks at the layer shape only.
static bool mlp_tok32(int C, int HID) {
    static const bool off = [] { const char* e = debug_env("SWF_MLP_TOK32"); return e && e[0] == '0'; }();
    return !off && C == 192 && HID % 192 == 0;
}

bool mlp_fused_supported(int C, int HID) {
    return (C == 128 || C == 192 || C == 256 || C == 384) && HID > 0 && HID % 128 == 0 && (int64_t)C * HID < (1 << 30);
}

// hidden splits as a function of the layer shape alone: about two 128-wide chunks per workgroup
int mlp_fused_splits(int C, int HID) {
    static const int forced = [] { const char* e = debug_env("SWF_MLP_SPLITS"); return e ? atoi(e) : 0; }();   // tools: tuning override
    if (mlp_wide(C, HID)) return HID / 256;
    if (mlp_tok32(C, HID)) return 1;
    if (forced > 0 && (HID / 128) % forced == 0) return forced;
    const int chunks = HID / 128;
    // measured at B=16 256x256 (us, kernel + reduce): C=192 hid 768: S=1 33, S=2 23+6, S=3 32+7; C=384 hid 1536: S=2 56, S=4 38+5,
    // S=6 32+5, S=12 50+8; hid 768: S=2 35+5, S=3 29+5.  More workgroups streaming the same weights contend for L2 delivery.
    if (C <= 256) return chunks % 2 == 0 ? 2 : 1;
    return chunks % 2 == 0 ? chunks / 2 : chunks;
}

int launch_mlp_fused(const MlpFusedDesc& d, int nstream, hipStream_t stream) {
    if (!mlp_fused_supported(d.C, d.HID)) return fail(SWF_ERR_UNSUPPORTED, "mlp_fused: C=%d hidden=%d", d.C, d.HID);
    if (d.M <= 0 || d.M > (1 << 30) / d.C) return fail(SWF_ERR_UNSUPPORTED, "mlp_fused: token count %d", d.M);
    MlpArgs a{};
    for (int s = 0; s < nstream; ++s) {
        a.x[s] = d.x[s]; a.out[s] = d.out[s]; a.gamma[s] = d.gamma[s]; a.beta[s] = d.beta[s];
        a.w1_hi[s] = reinterpret_cast<const bf16*>(d.w1_hi[s]); a.w1_lo[s] = reinterpret_cast<const bf16*>(d.w1_lo[s]);
        a.w2_hi[s] = reinterpret_cast<const bf16*>(d.w2_hi[s]); a.w2_lo[s] = reinterpret_cast<const bf16*>(d.w2_lo[s]);
        a.b1[s] = d.b1[s]; a.b2[s] = d.b2[s];
    }
    if (d.part0[0]) {
        for (int s = 0; s < nstream; ++s) {
            if (!d.part0[s] || !d.part1[s] || !d.pbias[s] || !d.x1[s]) return fail(SWF_ERR_NULL, "mlp_fused: attention-tail inputs must be given for every stream");
            if (d.x1[s] == d.x[s] || d.x1[s] == d.out[s]) return fail(SWF_ERR_UNSUPPORTED, "mlp_fused: x1 must not alias x / out");
            a.part0[s] = d.part0[s]; a.part1[s] = d.part1[s]; a.pbias[s] = d.pbias[s]; a.x1[s] = d.x1[s];
        }
    }
    a.M = d.M; a.HID = d.HID;
    a.splits = mlp_fused_splits(d.C, d.HID);
    // (mlp_fused_splits is 1 for the C = 192 shapes either way: the 64-token kernel then walks all six 128-wide chunks itself —
    //  half the workgroups, each streaming the weights once: less CU-time, longer launch)
    const bool wide = mlp_wide(d.C, d.HID), tok32 = mlp_tok32(d.C, d.HID) && d.schedule != SWF_SCHED_THROUGHPUT;
    a.nchunks = d.HID / (wide ? 256 : tok32 ? 192 : 128) / a.splits;
    a.scratch = d.scratch;
    if (d.ln_hi[0])
        for (int s = 0; s < nstream; ++s) {
            a.ln_gamma[s] = d.ln_gamma[s]; a.ln_beta[s] = d.ln_beta[s];
            a.ln_hi[s] = reinterpret_cast<bf16*>(d.ln_hi[s]); a.ln_lo[s] = reinterpret_cast<bf16*>(d.ln_lo[s]);
        }
    if (a.splits > 1 && (!d.scratch || (int64_t)nstream * a.splits * d.M * d.C > d.scratch_floats))
        return fail(SWF_ERR_WORKSPACE, "mlp_fused: scratch too small for %d hidden splits", a.splits);
    switch (d.C) {
        case 128: return launch_c<128, 4>(a, nstream, stream);
        case 192: return tok32 ? launch_c<192, 6, 32>(a, nstream, stream) : launch_c<192, 4>(a, nstream, stream);
        case 256: return launch_c<256, 4>(a, nstream, stream);
        case 384: return wide ? launch_c<384, 8>(a, nstream, stream) : launch_c<384, 4>(a, nstream, stream);
    }
    return fail(SWF_ERR_UNSUPPORTED, "mlp_fused: C=%d", d.C);
}

}  // namespace swf
