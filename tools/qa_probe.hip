// Phase stamps of one workgroup of the fused Q/K/V + attention kernel (diagnostic):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DQA_PROBE=37 -I swin_unet_image_fusion_amd/csrc tools/qa_probe.hip -o tools/qa_probe0 && tools/qa_probe0
#include "../swin_unet_image_fusion_amd/csrc/kernels_qkvattn.hip"
#include <vector>
namespace swf { char* err_buf() { static char b[512]; return b; } int fail(int st, const char* fmt, ...) { (void)fmt; return st; } }
int main() {
    using namespace swf;
    const int B = 16, H = 16, W = 16, C = 192;
    const int64_t N = (int64_t)B * H * W;
    swf_block_desc d{}; d.attn = {C, 8, 24, 8, 8, 1}; d.hidden = 768; d.cross = 1; d.precision = SWF_PREC_FAST;
    const size_t pb = qkvattn_packed_bytes(d);
    char* packed[2]; bf16_raw *xh[2], *xl[2], *oh[2], *ol[2];
    for (int s = 0; s < 2; ++s) {
        hipMalloc(&packed[s], pb); hipMemset(packed[s], 0, pb);
        hipMalloc(&xh[s], N * C * 2); hipMalloc(&xl[s], N * C * 2); hipMalloc(&oh[s], N * C * 2); hipMalloc(&ol[s], N * C * 2);
        hipMemset(xh[s], 0, N * C * 2); hipMemset(xl[s], 0, N * C * 2);
    }
    QkvAttnArgs a{};
    for (int s = 0; s < 2; ++s) { a.packed[s] = packed[s]; a.xn_hi[s] = xh[s]; a.xn_lo[s] = xl[s]; a.o_hi[s] = oh[s]; a.o_lo[s] = ol[s]; }
    a.B = B; a.H = H; a.W = W; a.shift = 1; a.cross = 1;
    for (int it = 0; it < 5; ++it) launch_qkvattn(d, a, 2, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; ++it) launch_qkvattn(d, a, 2, 0);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(qa_probe), sizeof(h));
    printf("launch %.1f us; stamps (us since kernel-entry stamp): ring issued+staged %.2f  k-loop %.2f  epilogue+barrier %.2f  attention %.2f  store %.2f\n",
           ms * 1000 / 20, (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01, (h[3] - h[0]) * 0.01, (h[4] - h[0]) * 0.01, (h[5] - h[0]) * 0.01);
    return 0;
}
