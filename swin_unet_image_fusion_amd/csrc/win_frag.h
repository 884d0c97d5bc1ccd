// Device helpers shared by the register-resident window kernels (kernels_win48.hip; kernels_win24.hip keeps its own copies):
// 32x32x16 MFMA wrappers on 16-byte operand fragments, split-bf16 / f16 packing of accumulator registers, the lane-half
// exchange, wave-uniform pointers.  See kernels_win24.hip for the layout conventions (rho order, lane (column, half)).
#pragma once
#include <hip/hip_runtime.h>

namespace swf {
namespace wf {

using bf16 = __bf16;
using f16 = _Float16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

// row of accumulator register i in lane half hf (C/D map of the 32x32 MFMAs) == k index of element i & 7 of k-step i >> 3
__host__ __device__ constexpr int rho(int i, int hf) { return (i & 3) + 8 * (i >> 2) + 4 * hf; }

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// acc += a . b over one 16-deep k-step with split-bf16 operands: three MFMAs, small cross terms first
__device__ __forceinline__ f32x16 mma3(u32x4 ahi, u32x4 alo, u32x4 bhi, u32x4 blo, f32x16 acc) {
    acc = mfma_bf16(alo, bhi, acc);
    acc = mfma_bf16(ahi, blo, acc);
    acc = mfma_bf16(ahi, bhi, acc);
    return acc;
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
__device__ __forceinline__ u32x4 pack8_f16(const float* v) {
    u32x4 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f16x2 h = {(f16)v[2 * p], (f16)v[2 * p + 1]};   // v_cvt_pk_f16_f32, round to nearest even
        o[p] = __builtin_bit_cast(unsigned, h);
    }
    return o;
}
// a = the value of lanes 0..31 (in every lane), b = the value of lanes 32..63.  v_permlane32_swap exchanges the upper half of
// its first operand with the lower half of its second (inline asm: hipcc 7.2 folds the builtin's second result into the
// first; the s_nop covers the VALU-write -> permlane hazard)
__device__ __forceinline__ void halves(float v, float& a, float& b) {
    a = v;
    b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float sum_halves(float v) { float a, b; halves(v, a, b); return a + b; }
__device__ __forceinline__ float max_halves(float v) { float a, b; halves(v, a, b); return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}


// Copy the fp32 vector sections of the two streams' packed images (n4 16-byte groups each) into LDS with every load of a thread
// issued before its first store (the obvious strided loop "lvec[i] = src[i]" runs one dependent global round trip per iteration;
// worth 0.7 us of a 48-us level-1 launch).
template <int N4, int NTHREADS>
__device__ __forceinline__ void fill_vectors(float* lvec, const char* vec0, const char* vec1, int tid) {
    constexpr int PER = (2 * N4 + NTHREADS - 1) / NTHREADS;
    f32x4 tmp[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int idx = tid + NTHREADS * k;
        if (idx < 2 * N4) tmp[k] = reinterpret_cast<const f32x4*>(idx < N4 ? vec0 : vec1)[idx < N4 ? idx : idx - N4];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int idx = tid + NTHREADS * k;
        if (idx < 2 * N4) reinterpret_cast<f32x4*>(lvec)[idx] = tmp[k];
    }
}

}  // namespace wf
}  // namespace swf

// compiler-only barrier: the loop-invariant weight fragment loads must not be hoisted (they would spill)
#define SWF_WF_FENCE() asm volatile("" ::: "memory")
