// Issue rate of single VALU instructions on gfx950 with 1 / 4 / 8 waves per SIMD (every CU busy):
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_bench.hip -o tools/valu_rate_bench0 && tools/valu_rate_bench0
// Prints wall nanoseconds per wave-instruction per SIMD (whole launch / instructions a SIMD issued).  s_memtime of one wave is
// not used: the oldest wave of a workgroup wins the issue arbitration and finishes early, so its own duration under-reports.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(1024) void bench(float* out, int iters, unsigned long long m) {
    float v[8], w[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pv[8], pw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i; w[i] = 1.0f + i * 0.25f; pv[i] = f2{v[i], 1.f}; pw[i] = f2{w[i], 2.f}; }
    const unsigned long long mask = m | (threadIdx.x & 1);   // wave-uniform enough for "s": launched with m only
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define ONE(i)                                                                                                      \
    if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));                      \
    else if constexpr (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));                                     \
    else if constexpr (OP == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));              \
    else if constexpr (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));                \
    else if constexpr (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(w[i]));            \
    else if constexpr (OP == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "s"(m)); \
    else if constexpr (OP == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));             \
    else if constexpr (OP == 7) asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(v[i]) : "v"(w[i]));                 \
    else if constexpr (OP == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));                     \
    else if constexpr (OP == 9) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[i]));                             \
    else if constexpr (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[i]) : "v"(pw[i]));           \
    else if constexpr (OP == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pv[i]) : "v"(pw[i]));               \
    else if constexpr (OP == 12) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));                    \
    else if constexpr (OP == 13) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));                    \
    else if constexpr (OP == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));                    \
    else if constexpr (OP == 15) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(w[i]));                        \
    else if constexpr (OP == 16) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));               \
    else if constexpr (OP == 17) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(w[i]) : "vcc");        \
    else if constexpr (OP == 18) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));                    \
    else if constexpr (OP == 19) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));                   \
    else if constexpr (OP == 20) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i]));                                \
    else if constexpr (OP == 21) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pv[i]) : "v"(pw[i]));               \
    else if constexpr (OP == 22) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(v[i]) : "v"(w[i]));            \
    else if constexpr (OP == 23) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(v[i]) : "v"(w[i]));                \
    else if constexpr (OP == 24) asm volatile("v_rsq_f32 %0, %0" : "+v"(v[i]));                                    \
    else if constexpr (OP == 25) asm volatile("v_mul_f32 %0, %0, %1 row_shr:1" : "+v"(v[i]) : "v"(w[i]));          \
    else if constexpr (OP == 26) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));                 \
    else if constexpr (OP == 27) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
            REP8(ONE)
#undef ONE
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + pv[i][0] + pv[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)mask;
}

int main() {
    const int iters = 10000;
    float* out;
    hipMalloc(&out, 512 * 1024 * sizeof(float));
    const char* names[] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_f16_f32", "v_max3_f32", "v_cndmask_b32 vcc", "v_cndmask_b32_e64 sgpr", "v_cvt_pk_bf16_f32",
                           "v_med3_f32", "v_add_f32", "v_lshlrev_b32", "v_pk_fma_f32", "v_pk_add_f32", "v_mul_f32", "v_max_f32", "v_and_b32", "v_mov_b32",
                           "v_perm_b32", "v_cmp_lt_f32 vcc", "v_sub_f32", "v_fmac_f32", "v_cvt_f16_f32", "v_pk_mul_f32", "v_lshl_or_b32", "v_bfi_b32",
                           "v_rsq_f32", "v_mul_f32 dpp row_shr", "v_pk_max_f16", "v_pk_fma_f16"};
    for (int op = 0; op <= 27; ++op) {
        printf("%-24s", names[op]);
        for (int waves_per_simd : {1, 4, 8}) {
            const int threads = waves_per_simd == 8 ? 1024 : 256 * waves_per_simd;
            const int grid = waves_per_simd == 8 ? 512 : 256;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                switch (op) {
#define C(N) case N: hipLaunchKernelGGL(bench<N>, dim3(grid), dim3(threads), 0, 0, out, iters, 0x5555555555555555ull); break;
                    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25) C(26) C(27)
#undef C
                }
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("  %dw: %5.2f ns", waves_per_simd, ms * 1e6 / ((double)waves_per_simd * iters * 32));
        }
        printf("\n");
    }
    return 0;
}
