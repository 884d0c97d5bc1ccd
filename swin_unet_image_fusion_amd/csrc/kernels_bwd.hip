// Backward of one BasicBlock (a005_BasicBlock.py:127-145 under torch.autograd, as the reference's training loop runs it:
// a016_train.py:150-196) in exact fp32 — SURVEY.md section 8(f) rank 4, first stage: the block, not yet the patch layers / head / loss.
//
//   x1 = x + proj(attention(q, k, v)),  q = Wq LN1(x) + bq,  k / v from LN1 of the own (self) or the other (cross) stream
//   x2 = x1 + W2 ELU(W1 LN2(x1) + b1) + b2
//
// The backward RECOMPUTES the forward intermediates from the block's inputs with the exact-tier forward kernels (kernels_generic.hip)
// and then walks the graph in reverse with the kernels of this file:
//   bwd_dx_kernel        dX (+)= dY . W                      (the input gradient of a linear layer; LDS-tiled fp32 FMA)
//   bwd_dw_kernel        dW  = dY^T . X, token chunks -> partial sums, reduced in fixed order (bit-reproducible, no atomics)
//   bwd_colsum_kernel    db  = column sums of dY, same chunking
//   ln_bwd_kernel        LayerNorm backward per token (+ the residual branch's gradient) and per-wave partial d gamma / d beta
//   elu_bwd_kernel       du = dh . (h > 0 ? 1 : h + 1)     (ELU'(u) from the saved output: exp(u) = h + 1 for u <= 0)
//   attn_bwd_kernel      per (window, head): recomputes the scores (bias, shift mask assigned as -1e10: a001:310), softmax statistics and
//                        dS = P . (dP - rowsum(dP . P)); thread = query for dQ, thread = key for dK / dV, thread = table entry for the
//                        relative-position bias gradient — every sum in a fixed order
//   reduce_rows_kernel   out[i] = sum over partial rows in index order
// Everything here is test-covered against torch.autograd of the CPU oracle (tests/test_gpu_backward.py).  It is the plain,
// correct-first tier: no MFMA, no fusion.
#include "kernels_bwd.h"

#include <algorithm>

#include "kernels_generic.h"

namespace swf {
namespace {

constexpr int kChunk = 256;    // tokens per partial sum of bwd_dw / bwd_colsum (level 0 at B=16: 1 024 chunks fill the chip)
constexpr int kGroup = 32;     // partial rows summed per thread and tree level (reduce_rows)

// out[M][K] (+)= dY[M][N] . W[N][K]
__global__ __launch_bounds__(256) void bwd_dx_kernel(const float* __restrict__ dY, const float* __restrict__ W, float* __restrict__ out,
                                                      int M, int N, int K, int accumulate) {
    __shared__ float As[64][17];
    __shared__ float Bs[16][65];
    const int m0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    float acc[4][4] = {};
    for (int n0 = 0; n0 < N; n0 += 16) {
        for (int e = tid; e < 64 * 16; e += 256) {
            const int r = e >> 4, c = e & 15;
            As[r][c] = (m0 + r < M && n0 + c < N) ? dY[(int64_t)(m0 + r) * N + n0 + c] : 0.f;
        }
        for (int e = tid; e < 16 * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            Bs[r][c] = (n0 + r < N && k0 + c < K) ? W[(int64_t)(n0 + r) * K + k0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[4 * ty + i][n]; b[i] = Bs[n][4 * tx + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + 4 * ty + i, k = k0 + 4 * tx + j;
            if (m < M && k < K) {
                float* o = out + (int64_t)m * K + k;
                *o = accumulate ? *o + acc[i][j] : acc[i][j];
            }
        }
}

// partial[chunk] = { [N][K]: sum over the chunk's tokens of dY[m][n] X[m][k];  [N]: column sums of dY (the bias gradient) }
__global__ __launch_bounds__(256) void bwd_dw_kernel(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ partial,
                                                      int M, int N, int K) {
    __shared__ float As[16][65];
    __shared__ float Bs[16][65];
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64, chunk = blockIdx.z;
    const int mlo = chunk * kChunk, mhi = min(M, mlo + kChunk);
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    float acc[4][4] = {};
    float bsum = 0.f;   // threads 0..63 of the k-tile-0 workgroups: column n0 + tid of dY
    for (int m0 = mlo; m0 < mhi; m0 += 16) {
        for (int e = tid; e < 16 * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            const bool live = m0 + r < mhi;
            As[r][c] = (live && n0 + c < N) ? dY[(int64_t)(m0 + r) * N + n0 + c] : 0.f;
            Bs[r][c] = (live && k0 + c < K) ? X[(int64_t)(m0 + r) * K + k0 + c] : 0.f;
        }
        __syncthreads();
        if (blockIdx.x == 0 && tid < 64) {
#pragma unroll
            for (int m = 0; m < 16; ++m) bsum += As[m][tid];
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[m][4 * ty + i]; b[i] = Bs[m][4 * tx + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* p = partial + (int64_t)chunk * N * (K + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 4 * ty + i, k = k0 + 4 * tx + j;
            if (n < N && k < K) p[(int64_t)n * K + k] = acc[i][j];
        }
    if (blockIdx.x == 0 && tid < 64 && n0 + tid < N) p[(int64_t)N * K + n0 + tid] = bsum;
}

// partial[chunk][N] = column sums of dY over the chunk's tokens.  The block covers NP = min(256, pow2 >= N) columns with 256 / NP row
// lanes: lane j sums rows mlo + j, mlo + j + RL, ... and the lanes are added in index order (fixed order: bit-reproducible).
__global__ __launch_bounds__(256) void bwd_colsum_kernel(const float* __restrict__ dY, float* __restrict__ partial, int M, int N, int NP) {
    __shared__ float red[256];
    const int chunk = blockIdx.y, RL = 256 / NP;
    const int col = threadIdx.x % NP, lane = threadIdx.x / NP, n = blockIdx.x * NP + col;
    const int mlo = chunk * kChunk, mhi = min(M, mlo + kChunk);
    float s = 0.f;
    if (n < N)
        for (int m = mlo + lane; m < mhi; m += RL) s += dY[(int64_t)m * N + n];
    red[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && n < N) {
        float t = red[col];
        for (int j = 1; j < RL; ++j) t += red[j * NP + col];
        partial[(int64_t)chunk * N + n] = t;
    }
}

// out[g][i] = sum of partial[r][i] over the rows r of group g (kGroup consecutive rows; blockIdx.y = g), r in index order
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ partial, float* __restrict__ out, int64_t count, int rows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int r0 = blockIdx.y * kGroup, r1 = min(rows, r0 + kGroup);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += partial[(int64_t)r * count + i];
    out[(int64_t)blockIdx.y * count + i] = s;
}
// the last level with two destinations: elements [0, count_a) -> out_a, the rest -> out_b (either may be NULL)
__global__ __launch_bounds__(256) void reduce_rows_split_kernel(const float* __restrict__ partial, float* __restrict__ out_a, int64_t count_a,
                                                                 float* __restrict__ out_b, int64_t count, int rows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += partial[(int64_t)r * count + i];
    if (i < count_a) { if (out_a) out_a[i] = s; }
    else if (out_b) out_b[i - count_a] = s;
}

__global__ __launch_bounds__(256) void elu_bwd_kernel(float* __restrict__ dh, const float* __restrict__ h, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float hv = h[i];
    dh[i] *= hv > 0.f ? 1.0f : hv + 1.0f;
}

// LayerNorm backward (eps 1e-5, biased variance; a004:54-72 / nn.LayerNorm):  with xhat = (x - mean) rstd, g = dy gamma:
//   dx = rstd (g - mean(g) - xhat mean(g xhat)) (+ dres);   d gamma = sum_tokens dy xhat;   d beta = sum_tokens dy.
// One wave per token row, eight rows per wave; lane l owns channels l, l + 64, ...  The wave's partial d gamma / d beta rows go to
// pgb[(block * 4 + wave)][2][C]; reduce_rows_kernel sums them in index order.
constexpr int kLnRows = 8, kLnMaxCh = 16;   // C <= 1024
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dy,
                                                      const float* __restrict__ dres, float* __restrict__ dx, float* __restrict__ pgb,
                                                      int64_t M, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * kLnRows;
    float dg[kLnMaxCh], db[kLnMaxCh];
#pragma unroll
    for (int i = 0; i < kLnMaxCh; ++i) { dg[i] = 0.f; db[i] = 0.f; }
    for (int rr = 0; rr < kLnRows; ++rr) {
        const int64_t m = row0 + rr;
        if (m >= M) break;   // wave-uniform
        const float* xr = x + m * C;
        const float* dyr = dy + m * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += xr[c];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < kLnMaxCh; ++i) {
            const int c = lane + 64 * i;
            if (c < C) {
                const float xh = (xr[c] - mean) * rstd, d = dyr[c], g = d * gamma[c];
                sg += g; sgx += g * xh;
                dg[i] += d * xh; db[i] += d;
            }
        }
        for (int o = 32; o > 0; o >>= 1) { sg += __shfl_xor(sg, o); sgx += __shfl_xor(sgx, o); }
        const float mg = sg / (float)C, mgx = sgx / (float)C;
        for (int c = lane; c < C; c += 64) {
            const float xh = (xr[c] - mean) * rstd, g = dyr[c] * gamma[c];
            float v = rstd * (g - mg - xh * mgx);
            if (dres) v += dres[m * C + c];
            dx[m * C + c] = v;
        }
    }
    float* p = pgb + ((int64_t)blockIdx.x * 4 + wave) * 2 * C;
#pragma unroll
    for (int i = 0; i < kLnMaxCh; ++i) {
        const int c = lane + 64 * i;
        if (c < C) { p[c] = dg[i]; p[C + c] = db[i]; }
    }
}

// ---- attention backward ------------------------------------------------------------------------------------------------------------
// One workgroup = one (window, head); Q, K, V, dO rows of the head in LDS (zero-padded to DMAX).  Scores exactly as attn_core_kernel.
struct AttnBwdProb {
    const float* Q; const float* K; const float* V; const float* dO;   // [tokens][heads*d], row stride ld
    float* dQ; float* dK; float* dV;
    const float* bias_table;
    float* dtable_partial;   // [windows * heads][table entries]
};
struct AttnBwdBatch { AttnBwdProb p[2]; };

// (the recompute form: nothing of size t x t is kept, every pass rebuilds the scores it needs — the fallback for windows whose
//  probability tile does not fit in LDS, e.g. 16 x 16)
template <int DMAX>
__global__ void attn_bwd_recompute_kernel(AttnBwdBatch batch, int ld, int B, int H, int W, int wh, int ww, int heads, int d, int shift, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AttnBwdProb pr = batch.p[blockIdx.z];
    const int t = wh * ww;
    float* Qs = smem;
    float* Ks = Qs + t * DMAX;
    float* Vs = Ks + t * DMAX;
    float* Gs = Vs + t * DMAX;           // dO
    float* tab = Gs + t * DMAX;
    const int tw = 2 * ww - 1, tsz = (2 * wh - 1) * tw;
    float* mrow = tab + tsz;             // per query: max, 1 / sum, D = sum_j p_ij dP_ij
    float* linv = mrow + t;
    float* Drow = linv + t;
    const int nwx = W / ww, nwy = H / wh;
    const int win = blockIdx.x, head = blockIdx.y;
    const int b = win / (nwx * nwy), wrem = win % (nwx * nwy);
    const int wy = wrem / nwx, wx = wrem % nwx;
    const int sh = shift ? wh / 2 : 0, sw = shift ? ww / 2 : 0;
    const int tid = threadIdx.x;
    auto token_of = [&](int j) -> int64_t {
        const int sy = wy * wh + j / ww, sx = wx * ww + j % ww;
        const int oy = (sy + sh) % H, ox = (sx + sw) % W;
        return ((int64_t)b * H + oy) * W + ox;
    };
    auto region_of = [&](int j) {
        const int sy = wy * wh + j / ww, sx = wx * ww + j % ww;
        return ((sy >= H - wh) + (sy >= H - wh / 2)) * 3 + ((sx >= W - ww) + (sx >= W - ww / 2));
    };
    for (int i = tid; i < tsz; i += blockDim.x) tab[i] = pr.bias_table[i];
    for (int e = tid; e < t * DMAX; e += blockDim.x) {
        const int j = e / DMAX, c = e % DMAX;
        float qv = 0.f, kv = 0.f, vv = 0.f, gv = 0.f;
        if (c < d) {
            const int64_t o = token_of(j) * ld + head * d + c;
            qv = pr.Q[o]; kv = pr.K[o]; vv = pr.V[o]; gv = pr.dO[o];
        }
        Qs[e] = qv; Ks[e] = kv; Vs[e] = vv; Gs[e] = gv;
    }
    __syncthreads();
    auto score = [&](int i, int j) -> float {   // exactly attn_core_kernel's score(i, j)
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < DMAX; ++c) dot = fmaf(Qs[i * DMAX + c], Ks[j * DMAX + c], dot);
        float s = dot * scale + tab[(j / ww - i / ww + wh - 1) * tw + (j % ww - i % ww + ww - 1)];
        if (shift && region_of(j) != region_of(i)) s = -1e10f;
        return s;
    };
    auto dP = [&](int i, int j) -> float {
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < DMAX; ++c) dot = fmaf(Gs[i * DMAX + c], Vs[j * DMAX + c], dot);
        return dot;
    };
    // ---- pass 1, thread = query i: softmax statistics, D_i, dQ_i ----
    if (tid < t) {
        const int i = tid;
        float mx = -INFINITY;
        for (int j = 0; j < t; ++j) mx = fmaxf(mx, score(i, j));
        float l = 0.f;
        for (int j = 0; j < t; ++j) l += expf(score(i, j) - mx);
        const float inv = 1.0f / l;
        float D = 0.f;
        for (int j = 0; j < t; ++j) D = fmaf(expf(score(i, j) - mx) * inv, dP(i, j), D);
        mrow[i] = mx; linv[i] = inv; Drow[i] = D;
        float dq[DMAX];
#pragma unroll
        for (int c = 0; c < DMAX; ++c) dq[c] = 0.f;
        for (int j = 0; j < t; ++j) {
            const float s = score(i, j);
            // the masked scores were ASSIGNED the constant -1e10 (a001:310): no gradient flows through them (their p is 0 anyway)
            const float p = expf(s - mx) * inv, ds = p * (dP(i, j) - D);
#pragma unroll
            for (int c = 0; c < DMAX; ++c) dq[c] = fmaf(ds, Ks[j * DMAX + c], dq[c]);
        }
        const int64_t o = token_of(i) * ld + head * d;
#pragma unroll
        for (int c = 0; c < DMAX; ++c)
            if (c < d) pr.dQ[o + c] = dq[c] * scale;
    }
    __syncthreads();
    // ---- pass 2, thread = key j: dK_j, dV_j ----
    if (tid < t) {
        const int j = tid;
        float dk[DMAX], dv[DMAX];
#pragma unroll
        for (int c = 0; c < DMAX; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        const int rj = region_of(j);
        for (int i = 0; i < t; ++i) {
            const float s = score(i, j);
            const float p = expf(s - mrow[i]) * linv[i];
            const bool masked = shift && rj != region_of(i);
            const float ds = masked ? 0.f : p * (dP(i, j) - Drow[i]);
#pragma unroll
            for (int c = 0; c < DMAX; ++c) {
                dk[c] = fmaf(ds, Qs[i * DMAX + c], dk[c]);
                dv[c] = fmaf(p, Gs[i * DMAX + c], dv[c]);
            }
        }
        const int64_t o = token_of(j) * ld + head * d;
#pragma unroll
        for (int c = 0; c < DMAX; ++c)
            if (c < d) { pr.dK[o + c] = dk[c] * scale; pr.dV[o + c] = dv[c]; }
    }
    // ---- pass 3, thread = table entry: d table[(dy, dx)] = sum over the (query, key) pairs at that offset of dS (unmasked pairs only) ----
    float* dt = pr.dtable_partial + ((int64_t)win * heads + head) * tsz;
    for (int e = tid; e < tsz; e += blockDim.x) {
        const int dy = e / tw - (wh - 1), dx = e % tw - (ww - 1);   // key - query
        float acc = 0.f;
        for (int i = 0; i < t; ++i) {
            const int jy = i / ww + dy, jx = i % ww + dx;
            if (jy < 0 || jy >= wh || jx < 0 || jx >= ww) continue;
            const int j = jy * ww + jx;
            if (shift && region_of(j) != region_of(i)) continue;
            const float p = expf(score(i, j) - mrow[i]) * linv[i];
            acc += p * (dP(i, j) - Drow[i]);
        }
        dt[e] = acc;
    }
}

// The resident form: the probability tile P [t][t + 1] of the (window, head) lives in LDS (row stride t + 1: rows by query thread and
// columns by key thread are both conflict-free), scores and exponentials are computed ONCE:
//   A1 thread = query i : scores -> LDS, row maximum, exponentials, P_ij, D_i = sum_j P_ij dP_ij
//   B1 thread = key j   : dV_j = sum_i P_ij dO_i
//   A2 thread = query i : dS_ij = P_ij (dP_ij - D_i) over P in place, dQ_i = scale sum_j dS_ij K_j
//   B2 thread = key j   : dK_j = scale sum_i dS_ij Q_i
//   C  thread = table entry: sum of dS over the (query, key) pairs at that offset
// Masked scores are ASSIGNED -1e10 (a001:310): their P is exactly 0, so is their dS, and no gradient reaches the table through them.
template <int DMAX>
__global__ void attn_bwd_kernel(AttnBwdBatch batch, int ld, int B, int H, int W, int wh, int ww, int heads, int d, int shift, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const AttnBwdProb pr = batch.p[blockIdx.z];
    const int t = wh * ww, tp = t + 1;
    float* Qs = smem;
    float* Ks = Qs + t * DMAX;
    float* Vs = Ks + t * DMAX;
    float* Gs = Vs + t * DMAX;           // dO
    float* tab = Gs + t * DMAX;
    const int tw = 2 * ww - 1, tsz = (2 * wh - 1) * tw;
    float* P = tab + tsz;                // [t][t + 1]
    float* Drow = P + t * tp;
    const int nwx = W / ww, nwy = H / wh;
    const int win = blockIdx.x, head = blockIdx.y;
    const int b = win / (nwx * nwy), wrem = win % (nwx * nwy);
    const int wy = wrem / nwx, wx = wrem % nwx;
    const int sh = shift ? wh / 2 : 0, sw = shift ? ww / 2 : 0;
    const int tid = threadIdx.x;
    auto token_of = [&](int j) -> int64_t {
        const int sy = wy * wh + j / ww, sx = wx * ww + j % ww;
        const int oy = (sy + sh) % H, ox = (sx + sw) % W;
        return ((int64_t)b * H + oy) * W + ox;
    };
    auto region_of = [&](int j) {
        const int sy = wy * wh + j / ww, sx = wx * ww + j % ww;
        return ((sy >= H - wh) + (sy >= H - wh / 2)) * 3 + ((sx >= W - ww) + (sx >= W - ww / 2));
    };
    for (int i = tid; i < tsz; i += blockDim.x) tab[i] = pr.bias_table[i];
    for (int e = tid; e < t * DMAX; e += blockDim.x) {
        const int j = e / DMAX, c = e % DMAX;
        float qv = 0.f, kv = 0.f, vv = 0.f, gv = 0.f;
        if (c < d) {
            const int64_t o = token_of(j) * ld + head * d + c;
            qv = pr.Q[o]; kv = pr.K[o]; vv = pr.V[o]; gv = pr.dO[o];
        }
        Qs[e] = qv; Ks[e] = kv; Vs[e] = vv; Gs[e] = gv;
    }
    __syncthreads();
    auto dP = [&](int i, int j) -> float {
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < DMAX; ++c) dot = fmaf(Gs[i * DMAX + c], Vs[j * DMAX + c], dot);
        return dot;
    };
    // ---- A1 ----
    if (tid < t) {
        const int i = tid, ri = region_of(i), iy = i / ww, ix = i % ww;
        float* row = P + i * tp;
        float q[DMAX];
#pragma unroll
        for (int c = 0; c < DMAX; ++c) q[c] = Qs[i * DMAX + c];
        float mx = -INFINITY;
        for (int j = 0; j < t; ++j) {   // exactly attn_core_kernel's score(i, j)
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < DMAX; ++c) dot = fmaf(q[c], Ks[j * DMAX + c], dot);
            float sc = dot * scale + tab[(j / ww - iy + wh - 1) * tw + (j % ww - ix + ww - 1)];
            if (shift && region_of(j) != ri) sc = -1e10f;
            row[j] = sc;
            mx = fmaxf(mx, sc);
        }
        float l = 0.f;
        for (int j = 0; j < t; ++j) { const float e = expf(row[j] - mx); row[j] = e; l += e; }
        const float inv = 1.0f / l;
        float D = 0.f;
        for (int j = 0; j < t; ++j) { const float pj = row[j] * inv; row[j] = pj; D = fmaf(pj, dP(i, j), D); }
        Drow[i] = D;
    }
    __syncthreads();
    // ---- B1 ----
    if (tid < t) {
        const int j = tid;
        float dv[DMAX];
#pragma unroll
        for (int c = 0; c < DMAX; ++c) dv[c] = 0.f;
        for (int i = 0; i < t; ++i) {
            const float pij = P[i * tp + j];
#pragma unroll
            for (int c = 0; c < DMAX; ++c) dv[c] = fmaf(pij, Gs[i * DMAX + c], dv[c]);
        }
        const int64_t o = token_of(j) * ld + head * d;
#pragma unroll
        for (int c = 0; c < DMAX; ++c)
            if (c < d) pr.dV[o + c] = dv[c];
    }
    __syncthreads();
    // ---- A2 ----
    if (tid < t) {
        const int i = tid;
        float* row = P + i * tp;
        const float D = Drow[i];
        float dq[DMAX];
#pragma unroll
        for (int c = 0; c < DMAX; ++c) dq[c] = 0.f;
        for (int j = 0; j < t; ++j) {
            const float ds = row[j] * (dP(i, j) - D);
            row[j] = ds;
#pragma unroll
            for (int c = 0; c < DMAX; ++c) dq[c] = fmaf(ds, Ks[j * DMAX + c], dq[c]);
        }
        const int64_t o = token_of(i) * ld + head * d;
#pragma unroll
        for (int c = 0; c < DMAX; ++c)
            if (c < d) pr.dQ[o + c] = dq[c] * scale;
    }
    __syncthreads();
    // ---- B2 ----
    if (tid < t) {
        const int j = tid;
        float dk[DMAX];
#pragma unroll
        for (int c = 0; c < DMAX; ++c) dk[c] = 0.f;
        for (int i = 0; i < t; ++i) {
            const float ds = P[i * tp + j];
#pragma unroll
            for (int c = 0; c < DMAX; ++c) dk[c] = fmaf(ds, Qs[i * DMAX + c], dk[c]);
        }
        const int64_t o = token_of(j) * ld + head * d;
#pragma unroll
        for (int c = 0; c < DMAX; ++c)
            if (c < d) pr.dK[o + c] = dk[c] * scale;
    }
    // ---- C ----
    float* dt = pr.dtable_partial + ((int64_t)win * heads + head) * tsz;
    for (int e = tid; e < tsz; e += blockDim.x) {
        const int dy = e / tw - (wh - 1), dx = e % tw - (ww - 1);   // key - query
        float acc = 0.f;
        for (int i = 0; i < t; ++i) {
            const int jy = i / ww + dy, jx = i % ww + dx;
            if (jy < 0 || jy >= wh || jx < 0 || jx >= ww) continue;
            acc += P[i * tp + jy * ww + jx];
        }
        dt[e] = acc;
    }
}

template <int DMAX>
int launch_attn_bwd_t(const AttnBwdBatch& batch, int nprob, int ld, int B, int H, int W, int wh, int ww, int heads, int d, int shift, hipStream_t stream) {
    const int t = wh * ww, tsz = (2 * wh - 1) * (2 * ww - 1);
    const int threads = cdiv(t, 64) * 64;
    if (threads > 1024) return fail(SWF_ERR_UNSUPPORTED, "attention backward: window of %d tokens > 1024", t);
    dim3 grid(B * (H / wh) * (W / ww), heads, nprob);
    const float scale = 1.0f / sqrtf((float)d);
    const size_t lds_res = ((size_t)4 * t * DMAX + tsz + (size_t)t * (t + 1) + t) * sizeof(float);
    if (lds_res <= 96 * 1024) {   // the probability tile stays in LDS (8 x 8 and 7 x 7 windows at every head width)
        if (lds_res > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<DMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_res);
            if (e != hipSuccess) return fail(SWF_ERR_HIP, "hipFuncSetAttribute(attn_bwd): %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(attn_bwd_kernel<DMAX>, grid, dim3(threads), lds_res, stream, batch, ld, B, H, W, wh, ww, heads, d, shift, scale);
        return check_launch("attn_bwd");
    }
    const size_t lds = ((size_t)4 * t * DMAX + tsz + 3 * t) * sizeof(float);
    if (lds > 160 * 1024) return fail(SWF_ERR_UNSUPPORTED, "attention backward tile (t=%d, d=%d) needs %zu B of LDS", t, d, lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_recompute_kernel<DMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(SWF_ERR_HIP, "hipFuncSetAttribute(attn_bwd): %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(attn_bwd_recompute_kernel<DMAX>, grid, dim3(threads), lds, stream, batch, ld, B, H, W, wh, ww, heads, d, shift, scale);
    return check_launch("attn_bwd (recompute)");
}

int launch_attn_bwd(const AttnBwdBatch& batch, int nprob, int ld, int B, int H, int W, int wh, int ww, int heads, int d, int shift, hipStream_t stream) {
#define SWF_AB(D) return launch_attn_bwd_t<D>(batch, nprob, ld, B, H, W, wh, ww, heads, d, shift, stream)
    if (d <= 4) SWF_AB(4);
    if (d <= 8) SWF_AB(8);
    if (d <= 16) SWF_AB(16);
    if (d <= 32) SWF_AB(32);
    if (d <= 64) SWF_AB(64);
#undef SWF_AB
    return fail(SWF_ERR_UNSUPPORTED, "attention backward: head_dim %d > 64", d);
}

// ---- patch layers and padding ---------------------------------------------------------------------------------------------------------
// merged-token rows Z[(b, i, j)][(ph * mw + pw) * C + c]  <->  image [b][i * mh + ph][j * mw + pw][c] (a011:73-117): a permutation, so
// each direction is the other's adjoint.  to_image = 1: Z -> image (depth-to-space), 0: image -> Z (space-to-depth).
__global__ __launch_bounds__(256) void patch_permute_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int Hm, int Wm, int C,
                                                             int mh, int mw, int to_image) {
    const int64_t total = (int64_t)B * Hm * Wm * mh * mw * C;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    // e indexes the image [b][y][x][c]
    const int c = (int)(e % C);
    int64_t r = e / C;
    const int W = Wm * mw, H = Hm * mh;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    const int64_t z = (((int64_t)b * Hm + y / mh) * Wm + x / mw) * ((int64_t)mh * mw * C) + ((y % mh) * mw + x % mw) * C + c;
    if (to_image) dst[e] = src[z];
    else dst[z] = src[e];
}

// adjoint of the bottom / right reflect pad (a006:122-131, F.pad mode="reflect"): padded[H + i] = x[H - 2 - i], so input row y also
// receives the gradient of padded row 2H - 2 - y when that row exists; columns alike.  [B][H][W][C] <- [B][H + ph][W + pw][C]
__global__ __launch_bounds__(256) void reflect_pad_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int B, int H, int W, int C,
                                                               int ph, int pw) {
    const int64_t total = (int64_t)B * H * W * C;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C);
    int64_t r = e / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    const int Hp = H + ph, Wp = W + pw;
    const int y2 = 2 * H - 2 - y, x2 = 2 * W - 2 - x;
    const bool my = y2 >= H && y2 < Hp, mx = x2 >= W && x2 < Wp;
    auto at = [&](int yy, int xx) { return g[(((int64_t)b * Hp + yy) * Wp + xx) * C + c]; };
    float v = at(y, x);
    if (my) v += at(y2, x);
    if (mx) v += at(y, x2);
    if (my && mx) v += at(y2, x2);
    dx[e] = v;
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// ---- host helpers ----------------------------------------------------------------------------------------------------------------------
int dx(const float* dY, const float* W, float* out, int64_t M, int N, int K, int accumulate, hipStream_t st) {
    dim3 grid(cdiv(K, 64), (unsigned)cdiv64(M, 64));
    hipLaunchKernelGGL(bwd_dx_kernel, grid, dim3(256), 0, st, dY, W, out, (int)M, N, K, accumulate);
    return check_launch("bwd_dx");
}
int chunks_of(int64_t M) { return (int)cdiv64(M, kChunk); }
// Sum of `rows` partial rows of `count` floats in a fixed tree: kGroup consecutive rows per thread and level, levels in sequence
// (sequential sums of 131 072 per-window rows of the bias-table gradient took 13 ms a call).  The intermediate levels live BEHIND the
// partial rows: the buffer holds tree_rows(rows) rows.
int64_t tree_rows(int64_t rows) {
    int64_t t = rows;
    while (rows > kGroup) { rows = cdiv64(rows, kGroup); t += rows; }
    return t;
}
// out_b != NULL or count_a < count: the summed vector is split, [0, count_a) -> out, the rest -> out_b
int reduce_rows(float* partial, float* out, int64_t count, int64_t rows, hipStream_t st, int64_t count_a = -1, float* out_b = nullptr) {
    float* src = partial;
    while (rows > kGroup) {
        const int64_t g = cdiv64(rows, kGroup);
        float* dst = src + rows * count;
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)cdiv64(count, 256), (unsigned)g), dim3(256), 0, st, src, dst, count, (int)rows);
        SWF_TRY(check_launch("reduce_rows level"));
        src = dst; rows = g;
    }
    if (count_a >= 0)
        hipLaunchKernelGGL(reduce_rows_split_kernel, dim3((unsigned)cdiv64(count, 256)), dim3(256), 0, st, src, out, count_a, out_b, count, (int)rows);
    else
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)cdiv64(count, 256), 1), dim3(256), 0, st, src, out, count, (int)rows);
    return check_launch("reduce_rows");
}
int colsum_np(int N) { int np = 1; while (np < N && np < 256) np *= 2; return np; }
int colsum(const float* dY, float* partial, int64_t M, int N, hipStream_t st) {
    const int np = colsum_np(N);
    hipLaunchKernelGGL(bwd_colsum_kernel, dim3(cdiv(N, np), chunks_of(M)), dim3(256), 0, st, dY, partial, (int)M, N, np);
    return check_launch("bwd_colsum");
}
// dW [N][K] (and db [N] when asked) of a linear layer y = x W^T + b from dY [M][N] and X [M][K]; scratch: tree_rows(chunks) * N * (K + 1) floats
int dw(const float* dY, const float* X, float* dW, float* db, int64_t M, int N, int K, float* scratch, hipStream_t st) {
    if (!dW && !db) return SWF_OK;
    const int ch = chunks_of(M);
    // one pass over dY and X for both: the weight-gradient workgroups of k tile 0 also sum their dY columns (the bias gradient)
    hipLaunchKernelGGL(bwd_dw_kernel, dim3(cdiv(K, 64), cdiv(N, 64), ch), dim3(256), 0, st, dY, X, scratch, (int)M, N, K);
    SWF_TRY(check_launch("bwd_dw"));
    return reduce_rows(scratch, dW, (int64_t)N * (K + 1), ch, st, (int64_t)N * K, db);
}
int ln_blocks(int64_t M) { return (int)cdiv64(M, 4 * kLnRows); }
// dx = dres + LayerNorm backward of dy; d gamma / d beta (either may be NULL); scratch: tree_rows(ln_blocks * 4) * 2 * C + 2 * C floats
int ln_bwd(const float* x, const float* gamma, const float* dy, const float* dres, float* dxo, float* dgamma, float* dbeta, int64_t M, int C,
           float* scratch, hipStream_t st) {
    if (C > 64 * kLnMaxCh) return fail(SWF_ERR_UNSUPPORTED, "LayerNorm backward: C = %d > %d", C, 64 * kLnMaxCh);
    const int nb = ln_blocks(M);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb), dim3(256), 0, st, x, gamma, dy, dres, dxo, scratch, M, C);
    SWF_TRY(check_launch("ln_bwd"));
    if (dgamma || dbeta) {
        float* red = scratch + tree_rows((int64_t)nb * 4) * 2 * C;
        SWF_TRY(reduce_rows(scratch, red, (int64_t)2 * C, (int64_t)nb * 4, st));
        if (dgamma && hipMemcpyAsync(dgamma, red, (size_t)C * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(SWF_ERR_HIP, "ln_bwd: copy failed");
        if (dbeta && hipMemcpyAsync(dbeta, red + C, (size_t)C * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(SWF_ERR_HIP, "ln_bwd: copy failed");
    }
    return SWF_OK;
}

// scratch shared by dw() and ln_bwd() of a layer with M tokens, widest matrix dimension mx and C normalised channels
int64_t bwd_scratch_floats(int64_t M, int64_t mx, int64_t C) {
    return std::max(tree_rows(chunks_of(M)) * mx * (mx + 1), tree_rows((int64_t)ln_blocks(M) * 4) * 2 * C + 2 * C) + 64;
}

}  // namespace

size_t basic_block_bwd_ws(const swf_block_desc& d, int nstream, int B, int H, int W) {
    const int64_t N = (int64_t)B * H * W, C = d.attn.channels, HD = (int64_t)d.attn.heads * d.attn.head_dim, hid = d.hidden;
    const int64_t nwin = (int64_t)B * (H / d.attn.win_h) * (W / d.attn.win_w), tsz = (int64_t)(2 * d.attn.win_h - 1) * (2 * d.attn.win_w - 1);
    const int64_t mx = std::max(std::max(C, HD), hid);
    size_t t = 0;
    for (int s = 0; s < nstream; ++s)
        t += carve_bytes({N * C, N * HD, N * HD, N * HD, N * HD, N * C, N * C, N * hid,      // xn, q, k, v, o, x1, xn2, h
                          N * hid, N * C, N * C, N * HD, N * HD, N * HD, N * HD, N * C,      // dh, dxn2, gx1, do, dq, dk, dv, dxn
                          tree_rows(nwin * d.attn.heads) * tsz});
    t += carve_bytes({bwd_scratch_floats(N, mx, C)});
    return t;
}

int basic_block_bwd(const swf_block_desc& d, const swf_block_stream_params* px, const swf_block_stream_params* py, const float* x_in,
                    const float* y_in, const float* gx_out, const float* gy_out, float* gx_in, float* gy_in, const swf_block_stream_grads* gx,
                    const swf_block_stream_grads* gy, int B, int H, int W, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int nstream = py ? 2 : 1;
    const int64_t N = (int64_t)B * H * W;
    const int C = d.attn.channels, HD = d.attn.heads * d.attn.head_dim, hid = d.hidden;
    const bool cross = d.cross && nstream == 2;
    const int wh = d.attn.win_h, ww = d.attn.win_w, tsz = (2 * wh - 1) * (2 * ww - 1);
    const int64_t nwin = (int64_t)B * (H / wh) * (W / ww);
    if (N > INT32_MAX / std::max(std::max(C, HD), hid)) return fail(SWF_ERR_UNSUPPORTED, "basic_block_bwd: token count");
    Carver ws(workspace, workspace_bytes);
    struct S { float *xn, *q, *k, *v, *o, *x1, *xn2, *h, *dh, *dxn2, *gx1, *dO, *dq, *dk, *dv, *dxn, *dtab; } b[2];
    for (int s = 0; s < nstream; ++s) {
        b[s].xn = ws.floats(N * C); b[s].q = ws.floats(N * HD); b[s].k = ws.floats(N * HD); b[s].v = ws.floats(N * HD); b[s].o = ws.floats(N * HD);
        b[s].x1 = ws.floats(N * C); b[s].xn2 = ws.floats(N * C); b[s].h = ws.floats(N * hid);
        b[s].dh = ws.floats(N * hid); b[s].dxn2 = ws.floats(N * C); b[s].gx1 = ws.floats(N * C); b[s].dO = ws.floats(N * HD);
        b[s].dq = ws.floats(N * HD); b[s].dk = ws.floats(N * HD); b[s].dv = ws.floats(N * HD); b[s].dxn = ws.floats(N * C);
        b[s].dtab = ws.floats(tree_rows(nwin * d.attn.heads) * tsz);
    }
    const int64_t mx = std::max(std::max(C, HD), hid);
    float* scratch = ws.floats(bwd_scratch_floats(N, mx, C));
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "basic_block_bwd workspace too small (need %zu B)", ws.used);
    const swf_block_stream_params* pp[2] = {px, py};
    const swf_block_stream_grads* gp[2] = {gx, gy};
    const float* xin[2] = {x_in, y_in};
    const float* gout[2] = {gx_out, gy_out};
    float* gin[2] = {gx_in, gy_in};

    // ---- recompute the forward intermediates (exact tier) ----
    {
        LnBatch l1{};
        for (int s = 0; s < nstream; ++s) l1.p[s] = LnProb{xin[s], b[s].xn, pp[s]->ln1.gamma, pp[s]->ln1.beta};
        SWF_TRY(launch_layernorm(l1, nstream, N, C, 0, st));
        GemmBatch gq{};
        for (int s = 0; s < nstream; ++s) {
            const int kvs = cross ? 1 - s : s;
            gq.p[3 * s] = GemmProb{b[s].xn, pp[s]->attn.q.weight, pp[s]->attn.q.bias, nullptr, b[s].q};
            gq.p[3 * s + 1] = GemmProb{b[kvs].xn, pp[s]->attn.k.weight, pp[s]->attn.k.bias, nullptr, b[s].k};
            gq.p[3 * s + 2] = GemmProb{b[kvs].xn, pp[s]->attn.v.weight, pp[s]->attn.v.bias, nullptr, b[s].v};
        }
        SWF_TRY(launch_gemm_f32(gq, 3 * nstream, (int)N, HD, C, C, HD, 0, st));
        AttnCoreBatch ab{};
        for (int s = 0; s < nstream; ++s) ab.p[s] = AttnCoreProb{b[s].q, b[s].k, b[s].v, b[s].o, pp[s]->attn.bias_table};
        SWF_TRY(launch_attn_core(ab, nstream, HD, HD, HD, HD, B, H, W, wh, ww, d.attn.heads, d.attn.head_dim, d.attn.shift, st));
        GemmBatch gpj{};
        for (int s = 0; s < nstream; ++s) gpj.p[s] = GemmProb{b[s].o, pp[s]->attn.proj.weight, pp[s]->attn.proj.bias, xin[s], b[s].x1};
        SWF_TRY(launch_gemm_f32(gpj, nstream, (int)N, C, HD, HD, C, 0, st));
        LnBatch l2{};
        for (int s = 0; s < nstream; ++s) l2.p[s] = LnProb{b[s].x1, b[s].xn2, pp[s]->ln2.gamma, pp[s]->ln2.beta};
        SWF_TRY(launch_layernorm(l2, nstream, N, C, 0, st));
        GemmBatch g1{};
        for (int s = 0; s < nstream; ++s) g1.p[s] = GemmProb{b[s].xn2, pp[s]->fc1.weight, pp[s]->fc1.bias, nullptr, b[s].h};
        SWF_TRY(launch_gemm_f32(g1, nstream, (int)N, hid, C, C, hid, 1, st));
    }
    auto G = [&](int s) -> const swf_block_stream_grads& { static const swf_block_stream_grads none{}; return gp[s] ? *gp[s] : none; };
    // ---- MLP half, reverse ----
    for (int s = 0; s < nstream; ++s) {
        SWF_TRY(dx(gout[s], pp[s]->fc2.weight, b[s].dh, N, C, hid, 0, st));                                   // dh = g2 . W2
        SWF_TRY(dw(gout[s], b[s].h, G(s).fc2.weight, G(s).fc2.bias, N, C, hid, scratch, st));                 // dW2, db2
        hipLaunchKernelGGL(elu_bwd_kernel, dim3((unsigned)cdiv64(N * hid, 256)), dim3(256), 0, st, b[s].dh, b[s].h, N * hid);
        SWF_TRY(check_launch("elu_bwd"));
        SWF_TRY(dx(b[s].dh, pp[s]->fc1.weight, b[s].dxn2, N, hid, C, 0, st));                                 // dxn2 = du . W1
        SWF_TRY(dw(b[s].dh, b[s].xn2, G(s).fc1.weight, G(s).fc1.bias, N, hid, C, scratch, st));               // dW1, db1
        SWF_TRY(ln_bwd(b[s].x1, pp[s]->ln2.gamma, b[s].dxn2, gout[s], b[s].gx1, G(s).ln2.gamma, G(s).ln2.beta, N, C, scratch, st));
        SWF_TRY(dx(b[s].gx1, pp[s]->attn.proj.weight, b[s].dO, N, C, HD, 0, st));                              // dO = gx1 . Wp
        SWF_TRY(dw(b[s].gx1, b[s].o, G(s).attn.proj.weight, G(s).attn.proj.bias, N, C, HD, scratch, st));
    }
    // ---- attention core, reverse ----
    {
        AttnBwdBatch ab{};
        for (int s = 0; s < nstream; ++s)
            ab.p[s] = AttnBwdProb{b[s].q, b[s].k, b[s].v, b[s].dO, b[s].dq, b[s].dk, b[s].dv, pp[s]->attn.bias_table, b[s].dtab};
        SWF_TRY(launch_attn_bwd(ab, nstream, HD, B, H, W, wh, ww, d.attn.heads, d.attn.head_dim, d.attn.shift, st));
        for (int s = 0; s < nstream; ++s)
            if (G(s).attn.bias_table) {
                SWF_TRY(reduce_rows(b[s].dtab, G(s).attn.bias_table, (int64_t)tsz, nwin * d.attn.heads, st));
            }
    }
    // ---- Q / K / V projections, reverse: stream s's K and V read the normalised tokens of stream kvs (a002:67-82) ----
    for (int s = 0; s < nstream; ++s) SWF_TRY(dx(b[s].dq, pp[s]->attn.q.weight, b[s].dxn, N, HD, C, 0, st));   // initialises dxn[s]
    for (int s = 0; s < nstream; ++s) {
        const int kvs = cross ? 1 - s : s;
        SWF_TRY(dx(b[s].dk, pp[s]->attn.k.weight, b[kvs].dxn, N, HD, C, 1, st));
        SWF_TRY(dx(b[s].dv, pp[s]->attn.v.weight, b[kvs].dxn, N, HD, C, 1, st));
        SWF_TRY(dw(b[s].dq, b[s].xn, G(s).attn.q.weight, G(s).attn.q.bias, N, HD, C, scratch, st));
        SWF_TRY(dw(b[s].dk, b[kvs].xn, G(s).attn.k.weight, G(s).attn.k.bias, N, HD, C, scratch, st));
        SWF_TRY(dw(b[s].dv, b[kvs].xn, G(s).attn.v.weight, G(s).attn.v.bias, N, HD, C, scratch, st));
    }
    // ---- LN1, reverse: the input gradient = gx1 (residual branch) + LayerNorm backward of dxn ----
    for (int s = 0; s < nstream; ++s)
        SWF_TRY(ln_bwd(xin[s], pp[s]->ln1.gamma, b[s].dxn, b[s].gx1, gin[s], G(s).ln1.gamma, G(s).ln1.beta, N, C, scratch, st));
    return SWF_OK;
}

// ---- the inner modules on their own (a001 / a003 / a004 under autograd): WindowAttention, one MLP stream, one LayerNorm ---------------
size_t window_attention_bwd_ws(const swf_attn_desc& d, int B, int H, int W) {
    const int64_t N = (int64_t)B * H * W, C = d.channels, HD = (int64_t)d.heads * d.head_dim;
    const int64_t nwin = (int64_t)B * (H / d.win_h) * (W / d.win_w), tsz = (int64_t)(2 * d.win_h - 1) * (2 * d.win_w - 1);
    return carve_bytes({N * HD, N * HD, N * HD, N * HD, N * HD, N * HD, N * HD, N * HD, tree_rows(nwin * d.heads) * tsz}) +
           carve_bytes({bwd_scratch_floats(N, std::max(C, HD), C)});
}

// WindowAttention.forward (a001:448-474) under autograd: out = proj(attention(q_in Wq, k_in Wk, v_in Wv)).  Q / K / V and the attention
// output are recomputed in exact fp32; gq / gk / gv are the gradients of the three inputs (separate buffers: the caller adds them where
// one tensor was passed more than once).
int window_attention_bwd(const swf_attn_desc& d, const swf_attn_params& p, const float* q_in, const float* k_in, const float* v_in, const float* gout,
                         float* gq, float* gk, float* gv, const swf_attn_grads* gp, int B, int H, int W, void* workspace, size_t workspace_bytes,
                         hipStream_t st) {
    const int64_t N = (int64_t)B * H * W;
    const int C = d.channels, HD = d.heads * d.head_dim, wh = d.win_h, ww = d.win_w, tsz = (2 * wh - 1) * (2 * ww - 1);
    const int64_t nwin = (int64_t)B * (H / wh) * (W / ww);
    if (N > INT32_MAX / std::max(C, HD)) return fail(SWF_ERR_UNSUPPORTED, "window_attention_bwd: token count");
    Carver ws(workspace, workspace_bytes);
    float* Q = ws.floats(N * HD); float* K = ws.floats(N * HD); float* V = ws.floats(N * HD); float* O = ws.floats(N * HD);
    float* dO = ws.floats(N * HD); float* dQ = ws.floats(N * HD); float* dK = ws.floats(N * HD); float* dV = ws.floats(N * HD);
    float* dtab = ws.floats(tree_rows(nwin * d.heads) * tsz);
    float* scratch = ws.floats(bwd_scratch_floats(N, std::max(C, HD), C));
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "window_attention_bwd workspace too small (need %zu B)", ws.used);
    static const swf_attn_grads none{};
    const swf_attn_grads& g = gp ? *gp : none;
    GemmBatch gq3{};
    gq3.p[0] = GemmProb{q_in, p.q.weight, p.q.bias, nullptr, Q};
    gq3.p[1] = GemmProb{k_in, p.k.weight, p.k.bias, nullptr, K};
    gq3.p[2] = GemmProb{v_in, p.v.weight, p.v.bias, nullptr, V};
    SWF_TRY(launch_gemm_f32(gq3, 3, (int)N, HD, C, C, HD, 0, st));
    AttnCoreBatch ac{};
    ac.p[0] = AttnCoreProb{Q, K, V, O, p.bias_table};
    SWF_TRY(launch_attn_core(ac, 1, HD, HD, HD, HD, B, H, W, wh, ww, d.heads, d.head_dim, d.shift, st));
    SWF_TRY(dx(gout, p.proj.weight, dO, N, C, HD, 0, st));                                  // dO = gout . Wp
    SWF_TRY(dw(gout, O, g.proj.weight, g.proj.bias, N, C, HD, scratch, st));
    AttnBwdBatch ab{};
    ab.p[0] = AttnBwdProb{Q, K, V, dO, dQ, dK, dV, p.bias_table, dtab};
    SWF_TRY(launch_attn_bwd(ab, 1, HD, B, H, W, wh, ww, d.heads, d.head_dim, d.shift, st));
    if (g.bias_table) SWF_TRY(reduce_rows(dtab, g.bias_table, (int64_t)tsz, nwin * d.heads, st));
    SWF_TRY(dx(dQ, p.q.weight, gq, N, HD, C, 0, st));
    SWF_TRY(dx(dK, p.k.weight, gk, N, HD, C, 0, st));
    SWF_TRY(dx(dV, p.v.weight, gv, N, HD, C, 0, st));
    SWF_TRY(dw(dQ, q_in, g.q.weight, g.q.bias, N, HD, C, scratch, st));
    SWF_TRY(dw(dK, k_in, g.k.weight, g.k.bias, N, HD, C, scratch, st));
    SWF_TRY(dw(dV, v_in, g.v.weight, g.v.bias, N, HD, C, scratch, st));
    return SWF_OK;
}

size_t mlp_bwd_ws(int64_t N, int C, int hid) {
    return carve_bytes({N * hid, N * hid}) + carve_bytes({bwd_scratch_floats(N, std::max(C, hid), C)});
}

// one stream of AutoPathMLP.forward (a003:46-50) under autograd: out = fc2(ELU(fc1(x)))
int mlp_bwd(const swf_linear& fc1, const swf_linear& fc2, const float* x, const float* gout, float* gx, const swf_linear_grad* g1, const swf_linear_grad* g2,
            int64_t N, int C, int hid, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (N > INT32_MAX / std::max(C, hid)) return fail(SWF_ERR_UNSUPPORTED, "mlp_bwd: token count");
    Carver ws(workspace, workspace_bytes);
    float* h = ws.floats(N * hid); float* dh = ws.floats(N * hid);
    float* scratch = ws.floats(bwd_scratch_floats(N, std::max(C, hid), C));
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "mlp_bwd workspace too small (need %zu B)", ws.used);
    static const swf_linear_grad none{};
    GemmBatch gb{};
    gb.p[0] = GemmProb{x, fc1.weight, fc1.bias, nullptr, h};
    SWF_TRY(launch_gemm_f32(gb, 1, (int)N, hid, C, C, hid, 1, st));                          // h = ELU(fc1 x)
    SWF_TRY(dx(gout, fc2.weight, dh, N, C, hid, 0, st));
    SWF_TRY(dw(gout, h, (g2 ? *g2 : none).weight, (g2 ? *g2 : none).bias, N, C, hid, scratch, st));
    hipLaunchKernelGGL(elu_bwd_kernel, dim3((unsigned)cdiv64(N * hid, 256)), dim3(256), 0, st, dh, h, N * hid);
    SWF_TRY(check_launch("elu_bwd"));
    SWF_TRY(dx(dh, fc1.weight, gx, N, hid, C, 0, st));
    SWF_TRY(dw(dh, x, (g1 ? *g1 : none).weight, (g1 ? *g1 : none).bias, N, hid, C, scratch, st));
    return SWF_OK;
}

size_t layernorm_bwd_ws(int64_t N, int C) { return carve_bytes({bwd_scratch_floats(N, 1, C)}); }

// my_layer_norm (a004:54-72) under autograd
int layernorm_bwd(const swf_norm& ln, const float* x, const float* gout, float* gx, const swf_norm_grad* gp, int64_t N, int C, void* workspace,
                  size_t workspace_bytes, hipStream_t st) {
    Carver ws(workspace, workspace_bytes);
    float* scratch = ws.floats(bwd_scratch_floats(N, 1, C));
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "layernorm_bwd workspace too small (need %zu B)", ws.used);
    return ln_bwd(x, ln.gamma, gout, nullptr, gx, gp ? gp->gamma : nullptr, gp ? gp->beta : nullptr, N, C, scratch, st);
}

// ---- PatchMergingAndLinearLayer backward (a011:244-264 under autograd), one stream, no window padding (the module-level layer) ----------
size_t patch_bwd_ws(int B, int H, int W, int Cin, int Cout, int mh, int mw, int encoder) {
    const int64_t n = encoder ? (int64_t)B * (H / mh) * (W / mw) : (int64_t)B * H * W;   // merged tokens
    const int64_t K = encoder ? (int64_t)Cin * mh * mw : Cin, Nn = encoder ? Cout : (int64_t)Cout * mh * mw;
    const int64_t mx = std::max(K, Nn);
    return carve_bytes({n * K, n * Nn, n * Nn, n * Nn, n * Nn, n * K}) +
           carve_bytes({bwd_scratch_floats(n, mx, Nn)});
}

int patch_bwd(const swf_patch_params& p, const float* in, const float* gout, float* gin, const swf_patch_grads* gp, int B, int H, int W, int Cin,
              int Cout, int mh, int mw, int encoder, void* workspace, size_t workspace_bytes, hipStream_t st) {
    // H x W: the layer's INPUT map (encoder: full map, divisible by the merging size; decoder: the merged map)
    if (encoder && (H % mh || W % mw)) return fail(SWF_ERR_BAD_SHAPE, "patch backward: map %dx%d not divisible by the merging size", H, W);
    const int Hm = encoder ? H / mh : H, Wm = encoder ? W / mw : W;
    const int64_t n = (int64_t)B * Hm * Wm;
    const int K = encoder ? Cin * mh * mw : Cin, Nn = encoder ? Cout : Cout * mh * mw;
    if (n > INT32_MAX / std::max(K, Nn)) return fail(SWF_ERR_UNSUPPORTED, "patch backward: token count");
    Carver ws(workspace, workspace_bytes);
    float* Z = ws.floats(n * K);      // encoder: gathered patches; decoder: unused (the input rows are Z)
    float* U = ws.floats(n * Nn);     // conv output
    float* V = ws.floats(n * Nn);     // LayerNorm output (decoder: in merged-token order)
    float* A = ws.floats(n * Nn);     // ELU output in merged-token order
    float* dV = ws.floats(n * Nn);
    float* dZ = ws.floats(n * K);
    const int64_t mx = std::max(K, Nn);
    float* scratch = ws.floats(bwd_scratch_floats(n, mx, Nn));
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "patch backward workspace too small (need %zu B)", ws.used);
    const swf_patch_grads none{};
    const swf_patch_grads& g = gp ? *gp : none;
    const unsigned blocks_img = (unsigned)cdiv64(n * (int64_t)mh * mw * (encoder ? Cin : Cout), 256);
    const float* rows = in;
    // ---- recompute: rows -> conv -> LayerNorm -> ELU (in merged-token order; ELU commutes with the depth-to-space permutation) ----
    if (encoder) {
        hipLaunchKernelGGL(patch_permute_kernel, dim3(blocks_img), dim3(256), 0, st, in, Z, B, Hm, Wm, Cin, mh, mw, 0);
        SWF_TRY(check_launch("patch space-to-depth"));
        rows = Z;
    }
    GemmBatch gb{};
    gb.p[0] = GemmProb{rows, p.conv.weight, p.conv.bias, nullptr, U};
    SWF_TRY(launch_gemm_f32(gb, 1, (int)n, Nn, K, K, Nn, 0, st));
    LnBatch lb{};
    lb.p[0] = LnProb{U, A, p.ln.gamma, p.ln.beta};
    SWF_TRY(launch_layernorm(lb, 1, n, Nn, 1, st));   // A = ELU(LN(U))
    // ---- reverse ----
    const float* gA = gout;   // gradient w.r.t. A in merged-token order
    if (!encoder) {           // the output image -> merged-token order
        hipLaunchKernelGGL(patch_permute_kernel, dim3(blocks_img), dim3(256), 0, st, gout, V, B, Hm, Wm, Cout, mh, mw, 0);
        SWF_TRY(check_launch("patch space-to-depth (gradient)"));
        gA = V;
    }
    if (hipMemcpyAsync(dV, gA, (size_t)n * Nn * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(SWF_ERR_HIP, "patch backward: copy failed");
    hipLaunchKernelGGL(elu_bwd_kernel, dim3((unsigned)cdiv64(n * Nn, 256)), dim3(256), 0, st, dV, A, n * Nn);   // dV = gA . ELU'(.)
    SWF_TRY(check_launch("elu_bwd"));
    float* dU = V;            // (V is free again)
    SWF_TRY(ln_bwd(U, p.ln.gamma, dV, nullptr, dU, g.ln.gamma, g.ln.beta, n, Nn, scratch, st));
    SWF_TRY(dw(dU, rows, g.conv.weight, g.conv.bias, n, Nn, K, scratch, st));
    if (encoder) {
        SWF_TRY(dx(dU, p.conv.weight, dZ, n, Nn, K, 0, st));
        hipLaunchKernelGGL(patch_permute_kernel, dim3(blocks_img), dim3(256), 0, st, dZ, gin, B, Hm, Wm, Cin, mh, mw, 1);
        return check_launch("patch depth-to-space (gradient)");
    }
    return dx(dU, p.conv.weight, gin, n, Nn, K, 0, st);
}

int reflect_pad_bwd(const float* g, float* dxo, int B, int H, int W, int C, int ph, int pw, hipStream_t st) {
    if (ph >= H || pw >= W || ph < 0 || pw < 0) return fail(SWF_ERR_PAD, "reflect pad backward: pad (%d,%d) must be smaller than the map (%d,%d)", ph, pw, H, W);
    hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3((unsigned)cdiv64((int64_t)B * H * W * C, 256)), dim3(256), 0, st, g, dxo, B, H, W, C, ph, pw);
    return check_launch("reflect_pad_bwd");
}

int add_tensors(const float* a, const float* b, float* out, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, a, b, out, n);
    return check_launch("add");
}

// ---- final head backward (a013:126-152 under autograd; BatchNorm in eval mode = a per-channel affine map of its running statistics) ------
// forward:  t1 = conv1(cat(x, y)) [2 ch];  t2 = a t1 + c,  a = gamma / sqrt(var + eps),  c = beta - mean a;  t3 = ELU(t2);  out = conv2(t3)
// Both convolutions read reflect-padded maps ('same', padding_mode='reflect'): the adjoint folds the padded gradient back, here in gather
// form — a pixel sums over every (output pixel, tap) pair whose reflected source it is — so every sum has a fixed order.
namespace {

__device__ __forceinline__ int reflect2b(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ __launch_bounds__(256) void head_t1_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ t1,
                                                       swf_head_params p, int B, int H, int W, int ks) {
    const int64_t total = (int64_t)B * H * W, e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int r = ks / 2, px = (int)(e % W), py = (int)((e / W) % H);
    const int64_t b = e / ((int64_t)W * H);
    float a0 = p.conv1_b ? p.conv1_b[0] : 0.f, a1 = p.conv1_b ? p.conv1_b[1] : 0.f;
    for (int ky = 0; ky < ks; ++ky) {
        const int yy = reflect2b(py + ky - r, H);
        for (int kx = 0; kx < ks; ++kx) {
            const int xx = reflect2b(px + kx - r, W);
            const int64_t sidx = (b * H + yy) * W + xx;
            const float vx = x[sidx], vy = y[sidx];
            a0 = fmaf(p.conv1_w[((0 * 2 + 0) * ks + ky) * ks + kx], vx, a0);
            a0 = fmaf(p.conv1_w[((0 * 2 + 1) * ks + ky) * ks + kx], vy, a0);
            a1 = fmaf(p.conv1_w[((1 * 2 + 0) * ks + ky) * ks + kx], vx, a1);
            a1 = fmaf(p.conv1_w[((1 * 2 + 1) * ks + ky) * ks + kx], vy, a1);
        }
    }
    t1[2 * e] = a0; t1[2 * e + 1] = a1;
}

// out[pixel][ci] = sum over (output pixel o, tap k) with reflect(o + k - r) == pixel of g[o][co] w[co][ci][k]: the adjoint of a reflect-'same'
// convolution with CO output and 2 input channels.  post != 0: multiply channel ci by ELU'(a[ci] t1[ci] + c[ci]) (the head's dt2).
__global__ __launch_bounds__(256) void head_conv_adjoint_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ out,
                                                                 int CO, const float* __restrict__ t1, float a0, float c0, float a1, float c1,
                                                                 int post, int B, int H, int W, int ks) {
    const int64_t total = (int64_t)B * H * W, e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int r = ks / 2, px = (int)(e % W), py = (int)((e / W) % H);
    const int64_t b = e / ((int64_t)W * H);
    float acc[2] = {0.f, 0.f};
    for (int oy = max(0, py - 2 * r); oy <= min(H - 1, py + 2 * r); ++oy)
        for (int ky = 0; ky < ks; ++ky) {
            if (reflect2b(oy + ky - r, H) != py) continue;
            for (int ox = max(0, px - 2 * r); ox <= min(W - 1, px + 2 * r); ++ox)
                for (int kx = 0; kx < ks; ++kx) {
                    if (reflect2b(ox + kx - r, W) != px) continue;
                    const int64_t o = (b * H + oy) * W + ox;
                    for (int co = 0; co < CO; ++co) {
                        const float gv = g[o * CO + co];
                        acc[0] = fmaf(gv, w[((co * 2 + 0) * ks + ky) * ks + kx], acc[0]);
                        acc[1] = fmaf(gv, w[((co * 2 + 1) * ks + ky) * ks + kx], acc[1]);
                    }
                }
        }
    if (post) {
        const float u0 = a0 * t1[2 * e] + c0, u1 = a1 * t1[2 * e + 1] + c1;
        acc[0] *= u0 > 0.f ? 1.0f : expf(u0);
        acc[1] *= u1 > 0.f ? 1.0f : expf(u1);
    }
    out[2 * e] = acc[0]; out[2 * e + 1] = acc[1];
}

// per-pixel rows whose column sums are the conv2 gradients: [ci][ky][kx] gout t3(reflected source), then gout (bias)
__global__ __launch_bounds__(256) void head_rows2_kernel(const float* __restrict__ gout, const float* __restrict__ t1, float* __restrict__ rows,
                                                          float a0, float c0, float a1, float c1, int B, int H, int W, int ks) {
    const int64_t total = (int64_t)B * H * W, e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int r = ks / 2, px = (int)(e % W), py = (int)((e / W) % H), NC = 2 * ks * ks + 1;
    const int64_t b = e / ((int64_t)W * H);
    const float g = gout[e];
    float* row = rows + e * NC;
    for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx) {
            const int64_t sidx = (b * H + reflect2b(py + ky - r, H)) * W + reflect2b(px + kx - r, W);
            const float u0 = a0 * t1[2 * sidx] + c0, u1 = a1 * t1[2 * sidx + 1] + c1;
            row[(0 * ks + ky) * ks + kx] = g * (u0 > 0.f ? u0 : expm1f(u0));
            row[(1 * ks + ky) * ks + kx] = g * (u1 > 0.f ? u1 : expm1f(u1));
        }
    row[NC - 1] = g;
}

// per-pixel rows for conv1 / BatchNorm gradients: [co][ci][ky][kx] dt1[co] in[ci](reflected source), dt1[0..1] (conv1 bias),
// dt2[c] xhat[c] (gamma), dt2[c] (beta)
// (BatchNorm in training mode normalises with the batch statistics, which depend on every t1: dt1 = a (dt2 - k1 - xhat k2) with
//  k1 = mean(dt2), k2 = mean(dt2 xhat) per channel; eval mode: k1 = k2 = 0)
__global__ __launch_bounds__(256) void head_rows1_kernel(const float* __restrict__ dt2, const float* __restrict__ t1, const float* __restrict__ x,
                                                          const float* __restrict__ y, float* __restrict__ rows, float a0, float a1, float m0,
                                                          float m1, float is0, float is1, float k10, float k11, float k20, float k21,
                                                          int B, int H, int W, int ks) {
    const int64_t total = (int64_t)B * H * W, e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int r = ks / 2, px = (int)(e % W), py = (int)((e / W) % H), NC = 4 * ks * ks + 6;
    const int64_t b = e / ((int64_t)W * H);
    const float d0 = dt2[2 * e], d1 = dt2[2 * e + 1];
    const float xh0 = (t1[2 * e] - m0) * is0, xh1 = (t1[2 * e + 1] - m1) * is1;
    const float g0 = a0 * (d0 - k10 - xh0 * k20), g1 = a1 * (d1 - k11 - xh1 * k21);   // dt1
    float* row = rows + e * NC;
    for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx) {
            const int64_t sidx = (b * H + reflect2b(py + ky - r, H)) * W + reflect2b(px + kx - r, W);
            const float vx = x[sidx], vy = y[sidx];
            row[((0 * 2 + 0) * ks + ky) * ks + kx] = g0 * vx;
            row[((0 * 2 + 1) * ks + ky) * ks + kx] = g0 * vy;
            row[((1 * 2 + 0) * ks + ky) * ks + kx] = g1 * vx;
            row[((1 * 2 + 1) * ks + ky) * ks + kx] = g1 * vy;
        }
    row[4 * ks * ks] = g0; row[4 * ks * ks + 1] = g1;
    row[4 * ks * ks + 2] = d0 * xh0; row[4 * ks * ks + 3] = d1 * xh1;
    row[4 * ks * ks + 4] = d0; row[4 * ks * ks + 5] = d1;
}

// dt1 = a (dt2 - k1 - xhat k2) per channel
__global__ __launch_bounds__(256) void head_dt1_kernel(const float* __restrict__ dt2, const float* __restrict__ t1, float* __restrict__ out, float a0,
                                                        float a1, float m0, float m1, float is0, float is1, float k10, float k11, float k20,
                                                        float k21, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    out[2 * e] = a0 * (dt2[2 * e] - k10 - (t1[2 * e] - m0) * is0 * k20);
    out[2 * e + 1] = a1 * (dt2[2 * e + 1] - k11 - (t1[2 * e + 1] - m1) * is1 * k21);
}

// rows [dt2_0 xhat_0, dt2_1 xhat_1, dt2_0, dt2_1]: their column means are k2, k1 of the training-mode BatchNorm backward
__global__ __launch_bounds__(256) void head_rowsk_kernel(const float* __restrict__ dt2, const float* __restrict__ t1, float* __restrict__ rows, float m0,
                                                          float m1, float is0, float is1, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float d0 = dt2[2 * e], d1 = dt2[2 * e + 1];
    rows[4 * e] = d0 * (t1[2 * e] - m0) * is0; rows[4 * e + 1] = d1 * (t1[2 * e + 1] - m1) * is1; rows[4 * e + 2] = d0; rows[4 * e + 3] = d1;
}

// batch statistics of t1: pass 0 rows = t1, pass 1 rows = (t1 - mean)^2 with the mean read from `stat`
__global__ __launch_bounds__(256) void head_stat_rows_kernel(const float* __restrict__ t1, const float* __restrict__ stat, float* __restrict__ rows,
                                                              int pass, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    if (pass == 0) { rows[2 * e] = t1[2 * e]; rows[2 * e + 1] = t1[2 * e + 1]; }
    else { const float d0 = t1[2 * e] - stat[0], d1 = t1[2 * e + 1] - stat[1]; rows[2 * e] = d0 * d0; rows[2 * e + 1] = d1 * d1; }
}
// sums -> mean (pass 0) or biased variance (pass 1); pass 1 also updates the running statistics (nn.BatchNorm2d: momentum, unbiased variance)
__global__ void head_stat_finish_kernel(const float* __restrict__ sums, float* __restrict__ mean, float* __restrict__ var, float* __restrict__ rmean,
                                        float* __restrict__ rvar, float momentum, int pass, float n) {
    const int c = threadIdx.x;
    if (c >= 2) return;
    if (pass == 0) { mean[c] = sums[c] / n; return; }
    var[c] = sums[c] / n;
    if (rmean) rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean[c];
    if (rvar) rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (n > 1.0f ? sums[c] / (n - 1.0f) : var[c]);
}

__global__ __launch_bounds__(256) void head_split_kernel(const float* __restrict__ g2, float* __restrict__ gx, float* __restrict__ gy, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) { gx[e] = g2[2 * e]; gy[e] = g2[2 * e + 1]; }
}

}  // namespace

size_t head_bwd_ws(int B, int H, int W, int ks) {
    const int64_t n = (int64_t)B * H * W, nc = 4 * ks * ks + 6;
    return carve_bytes({2 * n, 2 * n, 2 * n, n * nc, tree_rows(chunks_of(n)) * nc + nc + 64});
}

int head_batch_stats(const swf_head_params& p, const float* x, const float* y, float* mean, float* var, float* running_mean, float* running_var,
                     float momentum, int B, int H, int W, int ks, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (ks < 1 || ks % 2 == 0 || ks / 2 >= H || ks / 2 >= W) return fail(SWF_ERR_PAD, "head statistics: kernel %d on a %dx%d map", ks, H, W);
    const int64_t n = (int64_t)B * H * W;
    Carver ws(workspace, workspace_bytes);
    float* t1 = ws.floats(2 * n);
    float* rows = ws.floats(2 * n);
    float* part = ws.floats(tree_rows(chunks_of(n)) * 2 + 64);
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "head statistics workspace too small (need %zu B)", ws.used);
    const unsigned blocks = (unsigned)cdiv64(n, 256);
    const int ch = chunks_of(n);
    float* sums = part + tree_rows(ch) * 2;
    hipLaunchKernelGGL(head_t1_kernel, dim3(blocks), dim3(256), 0, st, x, y, t1, p, B, H, W, ks);
    SWF_TRY(check_launch("head_t1"));
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(head_stat_rows_kernel, dim3(blocks), dim3(256), 0, st, t1, mean, rows, pass, n);
        SWF_TRY(check_launch("head stat rows"));
        SWF_TRY(colsum(rows, part, n, 2, st));
        SWF_TRY(reduce_rows(part, sums, 2, ch, st));
        hipLaunchKernelGGL(head_stat_finish_kernel, dim3(1), dim3(64), 0, st, sums, mean, var, running_mean, running_var, momentum, pass, (float)n);
        SWF_TRY(check_launch("head stat finish"));
    }
    return SWF_OK;
}

int head_bwd(const swf_head_params& p, const float* x, const float* y, const float* gout, float* gx, float* gy, const swf_head_grads* gp, int B,
             int H, int W, int ks, int batch_stats, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (ks < 1 || ks % 2 == 0 || ks / 2 >= H || ks / 2 >= W) return fail(SWF_ERR_PAD, "head backward: kernel %d on a %dx%d map", ks, H, W);
    const int64_t n = (int64_t)B * H * W;
    const int nc1 = 4 * ks * ks + 6, nc2 = 2 * ks * ks + 1;
    Carver ws(workspace, workspace_bytes);
    float* t1 = ws.floats(2 * n);
    float* dt2 = ws.floats(2 * n);
    float* din = ws.floats(2 * n);
    float* rows = ws.floats(n * nc1);
    float* part = ws.floats(tree_rows(chunks_of(n)) * nc1 + nc1 + 64);
    if (!ws.ok()) return fail(SWF_ERR_WORKSPACE, "head backward workspace too small (need %zu B)", ws.used);
    // the BatchNorm constants are four scalars per channel: read them on the host once (a 32-byte synchronous copy per call)
    float hg[2], hb[2], hm[2], hv[2];
    if (hipMemcpyAsync(hg, p.bn_gamma, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(hb, p.bn_beta, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(hm, p.bn_mean, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(hv, p.bn_var, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(SWF_ERR_HIP, "head backward: reading the BatchNorm constants failed");
    float a[2], c[2], is[2];
    for (int i = 0; i < 2; ++i) { is[i] = 1.0f / sqrtf(hv[i] + 1e-5f); a[i] = hg[i] * is[i]; c[i] = hb[i] - hm[i] * a[i]; }
    const unsigned blocks = (unsigned)cdiv64(n, 256);
    const swf_head_grads none{};
    const swf_head_grads& g = gp ? *gp : none;
    hipLaunchKernelGGL(head_t1_kernel, dim3(blocks), dim3(256), 0, st, x, y, t1, p, B, H, W, ks);
    SWF_TRY(check_launch("head_t1"));
    // dt2 = adjoint(conv2)(gout) . ELU'(t2)
    hipLaunchKernelGGL(head_conv_adjoint_kernel, dim3(blocks), dim3(256), 0, st, gout, p.conv2_w, dt2, 1, t1, a[0], c[0], a[1], c[1], 1, B, H, W, ks);
    SWF_TRY(check_launch("head adjoint conv2"));
    auto colsums = [&](int nc, float* out_host_layout) -> int {   // column sums of rows [n][nc] -> part tail (device)
        SWF_TRY(colsum(rows, part, n, nc, st));
        return reduce_rows(part, out_host_layout, nc, chunks_of(n), st);
    };
    float* sums = part + tree_rows(chunks_of(n)) * nc1;
    auto copy = [&](float* dst, const float* src, int count) -> int {
        if (!dst) return SWF_OK;
        return hipMemcpyAsync(dst, src, (size_t)count * 4, hipMemcpyDeviceToDevice, st) == hipSuccess ? SWF_OK : fail(SWF_ERR_HIP, "head backward: copy failed");
    };
    if (g.conv2_w || g.conv2_b) {
        hipLaunchKernelGGL(head_rows2_kernel, dim3(blocks), dim3(256), 0, st, gout, t1, rows, a[0], c[0], a[1], c[1], B, H, W, ks);
        SWF_TRY(check_launch("head rows2"));
        SWF_TRY(colsums(nc2, sums));
        SWF_TRY(copy(g.conv2_w, sums, 2 * ks * ks));
        SWF_TRY(copy(g.conv2_b, sums + 2 * ks * ks, 1));
    }
    float k1[2] = {0.f, 0.f}, k2[2] = {0.f, 0.f};
    if (batch_stats) {   // p.bn_mean / p.bn_var hold the BATCH statistics of this forward: the normalisation itself has a gradient
        hipLaunchKernelGGL(head_rowsk_kernel, dim3(blocks), dim3(256), 0, st, dt2, t1, rows, hm[0], hm[1], is[0], is[1], n);
        SWF_TRY(check_launch("head rows k"));
        SWF_TRY(colsums(4, sums));
        float hs[4];
        if (hipMemcpyAsync(hs, sums, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail(SWF_ERR_HIP, "head backward: reading the batch sums failed");
        for (int i = 0; i < 2; ++i) { k2[i] = hs[i] / (float)n; k1[i] = hs[2 + i] / (float)n; }
    }
    hipLaunchKernelGGL(head_rows1_kernel, dim3(blocks), dim3(256), 0, st, dt2, t1, x, y, rows, a[0], a[1], hm[0], hm[1], is[0], is[1], k1[0], k1[1],
                       k2[0], k2[1], B, H, W, ks);
    SWF_TRY(check_launch("head rows1"));
    SWF_TRY(colsums(nc1, sums));
    SWF_TRY(copy(g.conv1_w, sums, 4 * ks * ks));
    SWF_TRY(copy(g.conv1_b, sums + 4 * ks * ks, 2));
    SWF_TRY(copy(g.bn_gamma, sums + 4 * ks * ks + 2, 2));
    SWF_TRY(copy(g.bn_beta, sums + 4 * ks * ks + 4, 2));
    // d(cat(x, y)) = adjoint(conv1)(dt1), dt1 = dt2 a: fold a into the weights' side by scaling dt2 in place
    // (dt2 is not needed afterwards)
    {
        // scale channel c of dt2 by a[c] with the split kernel's twin: reuse head_conv_adjoint's generic form on pre-scaled gradients
        // through rows as scratch: rows[e*2+c] = dt2[e*2+c] * a[c]
        float* dt1 = rows;
        hipLaunchKernelGGL(head_dt1_kernel, dim3(blocks), dim3(256), 0, st, dt2, t1, dt1, a[0], a[1], hm[0], hm[1], is[0], is[1], k1[0], k1[1], k2[0],
                           k2[1], n);
        SWF_TRY(check_launch("head dt1"));
        hipLaunchKernelGGL(head_conv_adjoint_kernel, dim3(blocks), dim3(256), 0, st, dt1, p.conv1_w, din, 2, t1, 0.f, 0.f, 0.f, 0.f, 0, B, H, W, ks);
        SWF_TRY(check_launch("head adjoint conv1"));
    }
    hipLaunchKernelGGL(head_split_kernel, dim3(blocks), dim3(256), 0, st, din, gx, gy, n);
    return check_launch("head split");
}

}  // namespace swf
