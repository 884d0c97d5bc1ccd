// Micro-benchmark of the deep-level split-plane GEMM (kernels_deep.hip) on the shapes of levels 3 / 4 at B=16, 256x256.
// Build (from the repo root; libswinfuse.so must exist):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DSWF_GEMM_ABL=n] tools/gemm_bench.hip \
//         -Lswin_unet_image_fusion_amd -lswinfuse -Wl,-rpath,'$ORIGIN/../swin_unet_image_fusion_amd' -o tools/gemm_bench
// SWF_GEMM_ABL: 0 full kernel, 1 no epilogue stores, 2 no global loads inside the K loop, 3 no MFMAs (ablations for
// finding the bound; results of 1-3 are wrong by construction).
#include "../swin_unet_image_fusion_amd/csrc/kernels_deep.hip"

#include <cstdio>
#include <vector>

using namespace swf;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Shape { const char* name; int M, N, K, nprob, epi, res; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const Shape shapes[] = {
        {"L3 qkv ", 4096, 192, 192, 6, SP_EPI_F32, 0},       {"L3 proj", 4096, 192, 192, 2, SP_EPI_F32, 1},
        {"L3 fc1 ", 4096, 768, 192, 2, SP_EPI_ELU_SPLIT, 0}, {"L3 fc2 ", 4096, 192, 768, 2, SP_EPI_F32, 1},
        {"L4 qkv ", 1024, 384, 384, 6, SP_EPI_F32, 0},       {"L4 proj", 1024, 384, 384, 2, SP_EPI_F32, 1},
        {"L4 fc1 ", 1024, 1536, 384, 2, SP_EPI_ELU_SPLIT, 0}, {"L4 fc2 ", 1024, 384, 1536, 2, SP_EPI_F32, 1},
        {"c5 fc1 ", 32768, 768, 192, 2, SP_EPI_ELU_SPLIT, 0}, {"c5 fc2 ", 32768, 192, 768, 2, SP_EPI_F32, 1},
    };
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        const size_t a_el = (size_t)s.M * s.K, w_el = (size_t)s.N * s.K, o_el = (size_t)s.M * s.N;
        bf16_raw *a_hi, *a_lo, *w_hi, *w_lo, *o_hi, *o_lo;
        float *bias, *out, *scratch;
        CK(hipMalloc(&a_hi, a_el * 2 * s.nprob)); CK(hipMalloc(&a_lo, a_el * 2 * s.nprob));
        CK(hipMalloc(&w_hi, w_el * 2 * s.nprob)); CK(hipMalloc(&w_lo, w_el * 2 * s.nprob));
        CK(hipMalloc(&o_hi, o_el * 2 * s.nprob)); CK(hipMalloc(&o_lo, o_el * 2 * s.nprob));
        CK(hipMalloc(&bias, s.N * 4)); CK(hipMalloc(&out, o_el * 4 * s.nprob));
        const int sk = gemm_sp_splitk_for(s.K, s.epi);
        CK(hipMalloc(&scratch, o_el * 4 * s.nprob * sk));
        CK(hipMemset(a_hi, 0, a_el * 2 * s.nprob)); CK(hipMemset(a_lo, 0, a_el * 2 * s.nprob));
        CK(hipMemset(w_hi, 0, w_el * 2 * s.nprob)); CK(hipMemset(w_lo, 0, w_el * 2 * s.nprob));
        CK(hipMemset(bias, 0, s.N * 4)); CK(hipMemset(out, 0, o_el * 4 * s.nprob));
        SpGemmBatch b{};
        b.scratch = scratch; b.scratch_floats = (int64_t)o_el * s.nprob * sk;
        for (int i = 0; i < s.nprob; ++i)
            b.p[i] = SpGemmProb{a_hi + i * a_el, a_lo + i * a_el, w_hi + i * w_el, w_lo + i * w_el, bias,
                                s.res ? out + i * o_el : nullptr, out + i * o_el, o_hi + i * o_el, o_lo + i * o_el};
        for (int i = 0; i < 5; ++i)
            if (launch_gemm_sp(b, s.nprob, s.M, s.N, s.K, s.N, s.epi, st) != SWF_OK) { printf("launch failed: %s\n", swf_last_error_string()); return 1; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) launch_gemm_sp(b, s.nprob, s.M, s.N, s.K, s.N, s.epi, st);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        const double gf = 2.0 * s.M * s.N * (double)s.K * s.nprob * 3 / 1e9;
        printf("abl=%d %s M=%5d N=%4d K=%4d x%d  %7.2f us  %6.0f TF(bf16x3)\n", SWF_GEMM_ABL, s.name, s.M, s.N, s.K, s.nprob, us, gf / us / 1e3);
        hipFree(a_hi); hipFree(a_lo); hipFree(w_hi); hipFree(w_lo); hipFree(o_hi); hipFree(o_lo); hipFree(bias); hipFree(out); hipFree(scratch);
    }
    return 0;
}
