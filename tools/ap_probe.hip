// Phase timeline of one workgroup of the level-4 attention + projection kernel (kernels_attnproj.hip).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DAP_PROBE=<window> tools/ap_probe.hip -Lswin_unet_image_fusion_amd -lswinfuse \
//         -Wl,-rpath,'$ORIGIN/../swin_unet_image_fusion_amd' -o tools/ap_probe0
#include "../swin_unet_image_fusion_amd/csrc/kernels_attnproj.hip"

#include <cstdio>
#include <vector>

using namespace swf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    const int B = 16, H = 8, W = 8, M = B * H * W, C = 384;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    bf16_raw *q[2], *k[2], *v[2], *wh, *wl;
    float *res[2], *out[2], *bias, *tab;
    for (int s = 0; s < 2; ++s) {
        CK(hipMalloc(&q[s], M * C * 2)); CK(hipMalloc(&k[s], M * C * 2)); CK(hipMalloc(&v[s], M * C * 2));
        CK(hipMemset(q[s], 0, M * C * 2)); CK(hipMemset(k[s], 0, M * C * 2)); CK(hipMemset(v[s], 0, M * C * 2));
        CK(hipMalloc(&res[s], M * C * 4)); CK(hipMalloc(&out[s], M * C * 4)); CK(hipMemset(res[s], 0, M * C * 4));
    }
    CK(hipMalloc(&wh, C * C * 2)); CK(hipMalloc(&wl, C * C * 2)); CK(hipMemset(wh, 0, C * C * 2)); CK(hipMemset(wl, 0, C * C * 2));
    CK(hipMalloc(&bias, 4096)); CK(hipMalloc(&tab, 4096)); CK(hipMemset(bias, 0, 4096)); CK(hipMemset(tab, 0, 4096));
    swf_block_desc d{};
    d.precision = SWF_PREC_FAST; d.attn.channels = C; d.attn.heads = 8; d.attn.head_dim = 48; d.attn.win_h = d.attn.win_w = 8; d.attn.shift = 1;
    AttnProjArgs a{};
    for (int s = 0; s < 2; ++s) {
        a.q[s] = q[s]; a.k[s] = k[s]; a.v[s] = v[s]; a.wp_hi[s] = wh; a.wp_lo[s] = wl; a.pbias[s] = bias; a.table[s] = tab; a.res[s] = res[s]; a.out[s] = out[s];
    }
    a.B = B; a.H = H; a.W = W; a.shift = 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it)
        if (launch_attnproj(d, a, 2, st) != SWF_OK) { printf("launch failed: %s\n", swf_last_error_string()); return 1; }
    CK(hipEventRecord(e0, st));
    for (int it = 0; it < 10; ++it) launch_attnproj(d, a, 2, st);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[16];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(ap_probe), sizeof(h)));
    printf("attn_proj: %.1f us per launch (events, back to back); workgroup %d wave 0, us from entry: loads issued+bias tile %.2f | V image %.2f | barrier %.2f | attention %.2f | O stored %.2f | k loop + partial handed over %.2f | done %.2f\n",
           ms * 100.f, AP_PROBE, (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01, (h[3] - h[0]) * 0.01, (h[4] - h[0]) * 0.01, (h[5] - h[0]) * 0.01, (h[6] - h[0]) * 0.01, (h[7] - h[0]) * 0.01);
    return 0;
}
