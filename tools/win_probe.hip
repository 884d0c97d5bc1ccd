// Phase timeline of one workgroup of the fused block kernel (kernels_window.hip) on its first window.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSWF_WIN_PROBE=<workgroup> tools/win_probe.hip \
//         -o tools/win_probe        (stand-alone: NOT linked against libswinfuse.so, whose kernels of the same name would
//         be the ones registered with the HIP runtime)
#include "../swin_unet_image_fusion_amd/csrc/kernels_window.hip"

#include <cstdio>
#include <vector>

namespace swf {
char* err_buf() { static char b[512]; return b; }
int fail(int status, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(err_buf(), 512, fmt, ap); va_end(ap); return status; }
}
using namespace swf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    struct Shape { const char* name; int C, HID, H; } shapes[] = {{"L0 enc", 24, 96, 128}, {"L1 enc", 48, 192, 64}, {"L1 dec", 48, 96, 64}, {"L2 enc", 96, 384, 32}, {"L2 dec", 96, 192, 32}};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int B = 16;
    for (const Shape& sh : shapes) {
        swf_block_desc d{};
        d.attn.channels = sh.C; d.attn.heads = 8; d.attn.head_dim = sh.C / 8; d.attn.win_h = 8; d.attn.win_w = 8; d.attn.shift = 1;
        d.hidden = sh.HID; d.cross = 1; d.precision = SWF_PREC_FAST;
        const size_t pb = window_block_packed_bytes(d);
        const size_t ne = (size_t)B * sh.H * sh.H * sh.C;
        char* packed; float *x, *y, *ox, *oy;
        CK(hipMalloc(&packed, 2 * pb));
        {   // random but finite contents: every 16-bit word a bf16 in +-[0.008, 0.12) (two of them also form a tame fp32)
            std::vector<unsigned short> hp(pb);
            unsigned rs = 12345u;
            for (auto& v : hp) { rs = rs * 1664525u + 1013904223u; v = (unsigned short)(0x3C00u + ((rs >> 16) & 0x1FFu) + ((rs >> 30) & 1u) * 0x8000u); }
            CK(hipMemcpy(packed, hp.data(), 2 * pb, hipMemcpyHostToDevice));
        }
        CK(hipMalloc(&x, ne * 4)); CK(hipMalloc(&y, ne * 4)); CK(hipMalloc(&ox, ne * 4)); CK(hipMalloc(&oy, ne * 4));
        {
            std::vector<float> hx(ne);
            unsigned rs = 777u;
            for (auto& v : hx) { rs = rs * 1664525u + 1013904223u; v = ((rs >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
            CK(hipMemcpy(x, hx.data(), ne * 4, hipMemcpyHostToDevice));
            for (auto& v : hx) { rs = rs * 1664525u + 1013904223u; v = ((rs >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
            CK(hipMemcpy(y, hx.data(), ne * 4, hipMemcpyHostToDevice));
        }
        for (int it = 0; it < 3; ++it)
            if (launch_window_block(d, packed, packed + pb, x, y, ox, oy, B, sh.H, sh.H, st) != SWF_OK) { printf("launch failed: %s\n", err_buf()); return 1; }
        CK(hipStreamSynchronize(st));
        {   // kernel time by events: same block back to back, out-of-place and in-place
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int mode = 0; mode < 2; ++mode) {
                CK(hipEventRecord(e0, st));
                for (int it = 0; it < 20; ++it)
                    launch_window_block(d, packed, packed + pb, x, y, mode ? x : ox, mode ? y : oy, B, sh.H, sh.H, st);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("      %s: %.1f us per launch (20 back-to-back)\n", mode ? "in place    " : "out of place", ms * 1e3 / 20);
            }
        }
        unsigned long long h[16];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(swf_win_probe), sizeof(h)));
        auto us = [&](int i) { return (h[i] - h[0]) * 0.01; };
        printf("%s C=%d hid=%d (us, first window of WG %d): LN1 %.2f | QKV %.2f | bar %.2f | attention %.2f | bar %.2f | proj %.2f | LN2 %.2f | MLP %.2f | store %.2f\n", sh.name,
               sh.C, sh.HID, SWF_WIN_PROBE, us(1), us(2), us(3), us(4), us(5), us(6), us(7), us(8), us(9));
        printf("      setup (kernel entry -> first window) %.2f us; last window %.2f -> %.2f us after entry\n", (h[0] - h[10]) * 0.01, (h[11] - h[10]) * 0.01, (h[12] - h[10]) * 0.01);
        hipFree(packed); hipFree(x); hipFree(y); hipFree(ox); hipFree(oy);
    }
    return 0;
}
