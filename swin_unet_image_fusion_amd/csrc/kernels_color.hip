// Pre/post-processing either side of MyModel.forward in the reference's inference script:
//   in : IR uint8 gray and visible uint8 BGR (cv2.imread, a015_dataset.py:73-74) -> cv2.cvtColor(BGR2YCrCb) on
//        uint8 (a015:89) -> float32 / 255 (a015:57-60) -> Y plane to the model, Cr/Cb planes kept (a017:68)
//   out: fused Y -> clamp [0,1] (a017:83) -> cat with Cr/Cb -> cv2.cvtColor(YCrCb2RGB) on float32 (a017:87)
//        (-> save_image's x255+0.5 uint8 quantisation, a017:90)
// cv2 is not available to this build, so the arithmetic restates OpenCV's published 8-bit fixed-point forward
// transform (yuv_shift 14: B2Y 1868, G2Y 9617, R2Y 4899, YCRI 11682, YCBI 9241, CV_DESCALE rounding, saturate) and
// its float inverse (1.403, -0.714, -0.344, 1.773, delta 0.5): parity is pinned by formula only ("parity unpinned"
// against cv2 itself; tests/test_color_gpu.py checks against the same restatement in numpy).
// Pure HBM-bound byte work: one thread per pixel, planar outputs, no LDS, no MFMA.
#include "swf_common.h"

namespace swf {

__device__ __forceinline__ int descale14(int x) { return (x + (1 << 13)) >> 14; }
__device__ __forceinline__ int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// bgr: [B][H][W][3] uint8 (HWC, cv2 layout).  y: [B][1][H][W], crcb: [B][2][H][W] float32 in [0,1].
__global__ __launch_bounds__(256) void bgr8_to_ycrcb_kernel(const uint8_t* __restrict__ bgr, float* __restrict__ y,
                                                            float* __restrict__ crcb, int64_t pixels_per_image, int64_t total) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int b = bgr[e * 3 + 0], g = bgr[e * 3 + 1], r = bgr[e * 3 + 2];
        const int Y = descale14(b * 1868 + g * 9617 + r * 4899);
        const int Cr = descale14((r - Y) * 11682 + (128 << 14));
        const int Cb = descale14((b - Y) * 9241 + (128 << 14));
        const int64_t img = e / pixels_per_image, p = e % pixels_per_image;
        y[e] = (float)sat8(Y) / 255.0f;
        crcb[(img * 2 + 0) * pixels_per_image + p] = (float)sat8(Cr) / 255.0f;
        crcb[(img * 2 + 1) * pixels_per_image + p] = (float)sat8(Cb) / 255.0f;
    }
}

__global__ __launch_bounds__(256) void gray8_to_unit_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t total) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
        out[e] = (float)in[e] / 255.0f;
}

// fused_y: [B][1][H][W] (unclamped), crcb: [B][2][H][W] -> rgb_f: [B][3][H][W] float32 and/or rgb8: [B][H][W][3] uint8
__global__ __launch_bounds__(256) void ycrcb_to_rgb_kernel(const float* __restrict__ fy, const float* __restrict__ crcb,
                                                           float* __restrict__ rgb_f, uint8_t* __restrict__ rgb8,
                                                           int64_t pixels_per_image, int64_t total) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t img = e / pixels_per_image, p = e % pixels_per_image;
        const float Y = fminf(fmaxf(fy[e], 0.f), 1.f);                       // torch.clamp_(0, 1), a017:83
        const float Cr = crcb[(img * 2 + 0) * pixels_per_image + p], Cb = crcb[(img * 2 + 1) * pixels_per_image + p];
        const float bb = Y + (Cb - 0.5f) * 1.773f;
        const float gg = Y + (Cb - 0.5f) * -0.344f + (Cr - 0.5f) * -0.714f;
        const float rr = Y + (Cr - 0.5f) * 1.403f;
        if (rgb_f) {
            rgb_f[(img * 3 + 0) * pixels_per_image + p] = rr;
            rgb_f[(img * 3 + 1) * pixels_per_image + p] = gg;
            rgb_f[(img * 3 + 2) * pixels_per_image + p] = bb;
        }
        if (rgb8) {   // torchvision save_image: mul(255).add_(0.5).clamp_(0, 255).to(uint8)
            const float q[3] = {rr, gg, bb};
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb8[e * 3 + c] = (uint8_t)fminf(fmaxf(q[c] * 255.0f + 0.5f, 0.f), 255.f);
        }
    }
}

static unsigned grid_for(int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 16384); }

}  // namespace swf

using namespace swf;

extern "C" {

int swf_bgr8_to_ycrcb_fwd(const uint8_t* bgr, float* y, float* crcb, int32_t B, int32_t H, int32_t W, swf_stream_t stream) {
    if (!bgr || !y || !crcb) return fail(SWF_ERR_NULL, "bgr8_to_ycrcb: NULL tensor");
    if (B <= 0 || H <= 0 || W <= 0) return fail(SWF_ERR_BAD_SHAPE, "bgr8_to_ycrcb: empty image");
    const int64_t ppi = (int64_t)H * W, total = ppi * B;
    hipLaunchKernelGGL(bgr8_to_ycrcb_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), bgr, y, crcb, ppi, total);
    return check_launch("bgr8_to_ycrcb");
}

int swf_gray8_to_unit_fwd(const uint8_t* gray, float* out, int64_t count, swf_stream_t stream) {
    if (!gray || !out) return fail(SWF_ERR_NULL, "gray8_to_unit: NULL tensor");
    if (count <= 0) return fail(SWF_ERR_BAD_SHAPE, "gray8_to_unit: empty image");
    hipLaunchKernelGGL(gray8_to_unit_kernel, dim3(grid_for(count)), dim3(256), 0, as_stream(stream), gray, out, count);
    return check_launch("gray8_to_unit");
}

int swf_ycrcb_to_rgb_fwd(const float* fused_y, const float* crcb, float* rgb_f, uint8_t* rgb8, int32_t B, int32_t H, int32_t W,
                         swf_stream_t stream) {
    if (!fused_y || !crcb || (!rgb_f && !rgb8)) return fail(SWF_ERR_NULL, "ycrcb_to_rgb: NULL tensor");
    if (B <= 0 || H <= 0 || W <= 0) return fail(SWF_ERR_BAD_SHAPE, "ycrcb_to_rgb: empty image");
    const int64_t ppi = (int64_t)H * W, total = ppi * B;
    hipLaunchKernelGGL(ycrcb_to_rgb_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), fused_y, crcb, rgb_f, rgb8, ppi, total);
    return check_launch("ycrcb_to_rgb");
}

}  // extern "C"
