// Phase timeline of one workgroup of the deep-level patch kernels (kernels_deeppatch.hip) at the B=16 256x256 shapes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDP_PROBE=<tile> tools/dp_probe.hip -Lswin_unet_image_fusion_amd -lswinfuse \
//         -Wl,-rpath,'$ORIGIN/../swin_unet_image_fusion_amd' -o tools/dp_probe0
#include "../swin_unet_image_fusion_amd/csrc/kernels_deeppatch.hip"

#include <cstdio>

using namespace swf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    struct Sh { const char* name; int dec, Cin, Cout, Hin, Win; } shapes[] = {
        {"enc 96->192 (32x32 -> 16x16)", 0, 96, 192, 32, 32}, {"enc 192->384 (16x16 -> 8x8)", 0, 192, 384, 16, 16},
        {"dec 384->192 (8x8 -> 16x16)", 1, 384, 192, 8, 8}, {"dec 192->96 (16x16 -> 32x32)", 1, 192, 96, 16, 16}};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int B = 16;
    for (const Sh& sh : shapes) {
        const int K = sh.dec ? sh.Cin : 4 * sh.Cin, N = sh.dec ? 4 * sh.Cout : sh.Cout;
        const int Hm = sh.dec ? sh.Hin : sh.Hin / 2, Wm = sh.dec ? sh.Win : sh.Win / 2;
        const int Ho = sh.dec ? 2 * sh.Hin : Hm, Wo = sh.dec ? 2 * sh.Win : Wm;
        const size_t in_e = (size_t)B * sh.Hin * sh.Win * sh.Cin, out_e = (size_t)B * Ho * Wo * sh.Cout;
        float *in[2], *out[2], *vec, *z[2];
        void* pk[2];
        for (int s = 0; s < 2; ++s) {
            CK(hipMalloc(&in[s], in_e * 4)); CK(hipMemset(in[s], 0, in_e * 4)); CK(hipMalloc(&out[s], out_e * 4)); CK(hipMemset(out[s], 0, out_e * 4));
            CK(hipMalloc(&pk[s], (size_t)K * N * 4)); CK(hipMemset(pk[s], 0, (size_t)K * N * 4));
            CK(hipMalloc(&z[s], (size_t)B * Ho * Wo * N * 4));   // conv output rows of the column-sliced shapes (upper bound)
        }
        CK(hipMalloc(&vec, 8192)); CK(hipMemset(vec, 0, 8192));
        PatchFusedDesc d{};
        for (int s = 0; s < 2; ++s) { d.in[s] = in[s]; d.out[s] = out[s]; d.skip[s] = sh.dec ? out[s] : nullptr; d.bias[s] = vec; d.gamma[s] = vec; d.beta[s] = vec; }
        d.decoder = sh.dec; d.B = B; d.H = sh.Hin; d.W = sh.Win; d.Cin = sh.Cin; d.mh = d.mw = 2; d.Hm = Hm; d.Wm = Wm; d.Ho = Ho; d.Wo = Wo;
        d.K = K; d.N = N; d.Cout = sh.Cout; d.M = (int64_t)B * (sh.dec ? Hm * Wm : Ho * Wo);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int it = 0; it < 3; ++it)
            if (launch_deep_patch(d, pk, 2, st, z) != SWF_OK) { printf("launch failed: %s\n", swf_last_error_string()); return 1; }
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < 10; ++it) launch_deep_patch(d, pk, 2, st, z);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[16];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(dp_probe), sizeof(h)));
        auto us = [&](int i) { return (h[i] - h[0]) * 0.01; };
        printf("%s: K=%d N=%d  %.1f us per launch (events); staged %.2f | barrier %.2f | k loop %.2f", sh.name, K, N, ms * 100.f, us(1), us(2), us(3));
        if (K > 384) printf(" | staged(2) %.2f | barrier %.2f | k loop(2) %.2f", us(4), us(5), us(6));
        printf(" | barrier %.2f | out tile in LDS %.2f | done %.2f\n", us(8), us(9), us(10));
    }
    return 0;
}
