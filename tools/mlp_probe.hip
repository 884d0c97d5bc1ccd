// Phase timeline of one workgroup of the fused MLP kernel (kernels_mlp.hip) at the level-3 / level-4 shapes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSWF_MLP_PROBE=<workgroup x index> tools/mlp_probe.hip \
//         -Lswin_unet_image_fusion_amd -lswinfuse -Wl,-rpath,'$ORIGIN/../swin_unet_image_fusion_amd' -o tools/mlp_probe
// wall_clock64() ticks at 100 MHz (10 ns).
#include "../swin_unet_image_fusion_amd/csrc/kernels_mlp.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace swf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MLP_PROBE_COLD=1: a 768 MB memset between launches evicts L2 / Infinity Cache (weights and rows then come from HBM, as they do
// inside a forward).  MLP_PROBE_FOLD=1: the attention tail (x + proj bias + two partials -> x1) in the prologue and the next
// block's LN1 planes from the epilogue / reduce, as the level-3 blocks run it inside the model.
int main() {
    const bool cold = getenv("MLP_PROBE_COLD") != nullptr, fold = getenv("MLP_PROBE_FOLD") != nullptr;
    char* evict = nullptr;
    const size_t evict_bytes = size_t(768) << 20;
    if (cold) CK(hipMalloc(&evict, evict_bytes));
    printf("== cold=%d fold=%d\n", (int)cold, (int)fold);
    struct Shape { const char* name; int M, C, HID; } shapes[] = {{"L3 enc", 4096, 192, 768}, {"L3 dec", 4096, 192, 384}, {"L4 enc", 1024, 384, 1536}, {"L4 dec", 1024, 384, 768}};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (const Shape& sh : shapes) {
        const size_t xe = (size_t)sh.M * sh.C, we = (size_t)sh.HID * sh.C;
        float *x[2], *out[2], *g, *b, *b1, *b2, *scratch;
        bf16_raw* w[4];
        for (int s = 0; s < 2; ++s) { CK(hipMalloc(&x[s], xe * 4)); CK(hipMalloc(&out[s], xe * 4)); CK(hipMemset(x[s], 0, xe * 4)); }
        CK(hipMalloc(&g, 4096)); CK(hipMalloc(&b, 4096)); CK(hipMalloc(&b1, 8192)); CK(hipMalloc(&b2, 4096));
        CK(hipMemset(g, 0, 4096)); CK(hipMemset(b, 0, 4096)); CK(hipMemset(b1, 0, 8192)); CK(hipMemset(b2, 0, 4096));
        for (int i = 0; i < 4; ++i) { CK(hipMalloc(&w[i], we * 2)); CK(hipMemset(w[i], 0, we * 2)); }
        const int S = mlp_fused_splits(sh.C, sh.HID);
        CK(hipMalloc(&scratch, xe * 4 * 2 * S));
        MlpFusedDesc d{};
        for (int s = 0; s < 2; ++s) {
            d.x[s] = x[s]; d.out[s] = out[s]; d.gamma[s] = g; d.beta[s] = b; d.w1_hi[s] = w[0]; d.w1_lo[s] = w[1]; d.w2_hi[s] = w[2]; d.w2_lo[s] = w[3];
            d.b1[s] = b1; d.b2[s] = b2;
        }
        d.scratch = scratch; d.scratch_floats = (int64_t)xe * 2 * S; d.M = sh.M; d.C = sh.C; d.HID = sh.HID;
        float* extra[2][3]; bf16_raw* lnp[2][2];
        if (fold)
            for (int s = 0; s < 2; ++s) {
                for (int i = 0; i < 3; ++i) { CK(hipMalloc(&extra[s][i], xe * 4)); CK(hipMemset(extra[s][i], 0, xe * 4)); }
                for (int i = 0; i < 2; ++i) CK(hipMalloc(&lnp[s][i], xe * 2));
                d.part0[s] = extra[s][0]; d.part1[s] = extra[s][1]; d.x1[s] = extra[s][2]; d.pbias[s] = b2;
                d.ln_gamma[s] = g; d.ln_beta[s] = b; d.ln_hi[s] = lnp[s][0]; d.ln_lo[s] = lnp[s][1];
            }
        for (int it = 0; it < 3; ++it)
            if (launch_mlp_fused(d, 2, st) != SWF_OK) { printf("launch failed: %s\n", swf_last_error_string()); return 1; }
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0.f;
        for (int it = 0; it < 10; ++it) {
            if (cold) CK(hipMemsetAsync(evict, it, evict_bytes, st));
            CK(hipEventRecord(e0, st));
            launch_mlp_fused(d, 2, st);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float t = 0.f;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms += t;
        }
        unsigned long long h[64];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(swf_mlp_probe), sizeof(h)));
        {   // every workgroup's entry / exit, relative to the first entry (last launch)
            static unsigned long long wg[2 * 4096];
            CK(hipMemcpyFromSymbol(wg, HIP_SYMBOL(swf_mlp_wg), sizeof(wg)));
            const char* e32 = getenv("SWF_MLP_TOK32");
            const bool t32 = sh.C == 192 && sh.HID % 192 == 0 && !(getenv("SWF_DEBUG_SWITCHES") && e32 && e32[0] == '0');   // as mlp_tok32() in kernels_mlp.hip
            const int tok = t32 ? 32 : 64;
            const int nwg = ((sh.M + tok - 1) / tok) * S * 2;
            std::vector<double> en, ex;
            unsigned long long t0 = ~0ull;
            for (int i = 0; i < nwg; ++i) t0 = std::min(t0, wg[2 * i]);
            for (int i = 0; i < nwg; ++i) { en.push_back((wg[2 * i] - t0) * 0.01); ex.push_back((wg[2 * i + 1] - t0) * 0.01); }
            std::sort(en.begin(), en.end()); std::sort(ex.begin(), ex.end());
            printf("%s: %d workgroups, launch+reduce pair %.1f us by events; entry median %.2f max %.2f | exit min %.2f median %.2f max %.2f\n", sh.name, nwg,
                   ms * 100.f, en[nwg / 2], en[nwg - 1], ex[0], ex[nwg / 2], ex[nwg - 1]);
        }
        const char* e8 = getenv("SWF_MLP8");
        const bool wide = sh.C == 384 && sh.HID % 256 == 0 && !(e8 && e8[0] == '0');   // as mlp_wide() in kernels_mlp.hip
        const char* e32b = getenv("SWF_MLP_TOK32");
        const bool t32b = sh.C == 192 && sh.HID % 192 == 0 && !(getenv("SWF_DEBUG_SWITCHES") && e32b && e32b[0] == '0');
        const int nch = sh.HID / (wide ? 256 : t32b ? 192 : 128) / S;
        printf("%s C=%d hid=%d S=%d chunks/WG=%d (us from kernel entry of WG %d): LN done %.2f", sh.name, sh.C, sh.HID, S, nch, SWF_MLP_PROBE, (h[1] - h[0]) * 0.01);
        printf(" [prologue: rows+fold done %.2f | ring issued %.2f | gamma/beta in LDS %.2f | stats %.2f | barrier %.2f]", (h[50] - h[0]) * 0.01, (h[51] - h[0]) * 0.01,
               (h[52] - h[0]) * 0.01, (h[53] - h[0]) * 0.01, (h[54] - h[0]) * 0.01);
        for (int c = 0; c < nch && c < 8; ++c)
            printf(" | ch%d fc1 %.2f H %.2f bar %.2f fc2 %.2f", c, (h[2 + 4 * c] - h[0]) * 0.01, (h[3 + 4 * c] - h[0]) * 0.01, (h[4 + 4 * c] - h[0]) * 0.01, (h[5 + 4 * c] - h[0]) * 0.01);
        printf(" | loop end %.2f | done %.2f\n", (h[40] - h[0]) * 0.01, (h[41] - h[0]) * 0.01);
        for (int s = 0; s < 2; ++s) { hipFree(x[s]); hipFree(out[s]); }
        hipFree(g); hipFree(b); hipFree(b1); hipFree(b2); hipFree(scratch);
        for (int i = 0; i < 4; ++i) hipFree(w[i]);
    }
    return 0;
}
