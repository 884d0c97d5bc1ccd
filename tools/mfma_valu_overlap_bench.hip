// Do MFMA and VALU work overlap on a gfx950 SIMD, within a wave and across the waves of a SIMD?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap_bench.hip -o tools/mfma_valu_overlap_bench0 && tools/mfma_valu_overlap_bench0
// One iteration = 4 x v_mfma_f32_32x32x16_f16 (128 matrix cycles) and / or 32 x v_exp_f32 + 16 x v_cvt_pk_f16_f32 (about 320 VALU
// cycles), the shape of one head's P.V loop in the level-0 kernel.  Prints wall ns per iteration and SIMD for 1-4 waves per SIMD:
//   mfma        matrix work only            valu        vector work only
//   indep       both, no data dependence    chain       exp of the accumulator -> packed -> B operand of the next MFMA
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void bench(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 a = {0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, b = a;
    f32x16 acc;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; v[i] = -0.001f * (lane + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            if constexpr (MODE == 1 || MODE == 2) {   // vector work on its own registers
                u32x4 p;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e0 = __builtin_amdgcn_exp2f(v[4 * (ps & 1) + j]), e1 = __builtin_amdgcn_exp2f(v[8 + 4 * (ps & 1) + j]);
                    const f16x2 h = {(_Float16)e0, (_Float16)e1};
                    p[j] = __builtin_bit_cast(unsigned, h);
                    v[4 * (ps & 1) + j] = e0 - 1.0009765625f; v[8 + 4 * (ps & 1) + j] = e1 - 1.0009765625f;
                }
                asm volatile("" :: "v"(p));
            }
            if constexpr (MODE == 3) {   // exp of the previous accumulator -> packed -> this MFMA's B operand
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e0 = __builtin_amdgcn_exp2f(acc[4 * ps + j] * -1e-6f), e1 = __builtin_amdgcn_exp2f(acc[(4 * ps + j + 8) & 15] * -1e-6f);
                    const f16x2 h = {(_Float16)e0, (_Float16)e1};
                    b[j] = __builtin_bit_cast(unsigned, h);
                }
            }
            if constexpr (MODE != 1)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int iters = 20000;
    float* out;
    hipMalloc(&out, 1024 * 1024 * sizeof(float));
    const char* names[] = {"mfma", "valu", "indep", "chain"};
    for (int mode = 0; mode < 4; ++mode) {
        printf("%-6s", names[mode]);
        for (int w = 1; w <= 4; ++w) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(bench<0>, dim3(256), dim3(256 * w), 0, 0, out, iters); break;
                    case 1: hipLaunchKernelGGL(bench<1>, dim3(256), dim3(256 * w), 0, 0, out, iters); break;
                    case 2: hipLaunchKernelGGL(bench<2>, dim3(256), dim3(256 * w), 0, 0, out, iters); break;
                    case 3: hipLaunchKernelGGL(bench<3>, dim3(256), dim3(256 * w), 0, 0, out, iters); break;
                }
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("  %dw: %7.1f ns/iter/wave (%6.1f per SIMD-iter)", w, ms * 1e6 / iters, ms * 1e6 / iters / w);
        }
        printf("\n");
    }
    return 0;
}
