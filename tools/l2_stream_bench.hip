// How fast can the waves of a launch stream the SAME small region (a block's weight fragments) out of L2?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/l2_stream_bench.hip -o tools/l2_stream_bench0
// The fused block kernels at C=48 / 96 and the deep-level fused MLP all deliver ~35-40 GB/s of weight fragments per CU
// (8.6-10 TB/s aggregate, profiles/r01v_pmc_window_block_c48_tt2.json).  This probe separates the candidates: 256 workgroups
// (one per CU), W waves each, every wave reads 1-KB fragments (16 B per lane, as the kernels do) of a region of R bytes
// `iters` times with D loads in flight; `shared` = every workgroup reads the same region (the kernels' pattern) or its own.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(512) void stream_kernel(const u32x4* __restrict__ base, int nfrag, int iters, int wg_stride_frags,
                                                     int split, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const u32x4* p = base + (size_t)blockIdx.x * wg_stride_frags * 64 + lane;
    // split = 1: the waves of a workgroup share the region (each fragment read by one wave); 0: every wave reads all of it
    const int f0 = split ? wave : 0, fs = split ? nw : 1;
    const int per = (nfrag - f0 + fs - 1) / fs;       // fragments this wave reads per pass
    const int total = per * iters;
    u32x4 r[D];
    u32x4 acc = {0u, 0u, 0u, 0u};
    int issued = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int f = f0 + (issued % per) * fs;
        r[i] = p[(size_t)f * 64];
        ++issued;
    }
    for (int done = 0; done < total; done += D) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            acc ^= r[i];
            __builtin_amdgcn_sched_barrier(0);
            const int f = f0 + (issued % per) * fs;   // past the end: harmless re-reads
            r[i] = p[(size_t)f * 64];
            ++issued;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;   // keeps the loads alive
}

template <int D>
float run(const u32x4* buf, int nfrag, int iters, int waves, int stride, int split, unsigned* sink, hipStream_t st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_kernel<D>), dim3(256), dim3(64 * waves), 0, st, buf, nfrag, iters, stride, split, sink);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    hipLaunchKernelGGL((stream_kernel<D>), dim3(256), dim3(64 * waves), 0, st, buf, nfrag, iters, stride, split, sink);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const size_t maxbytes = (size_t)256 * 2 * 1024 * 1024;   // room for a private 2-MB region per workgroup
    u32x4* buf; unsigned* sink;
    CK(hipMalloc(&buf, maxbytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, maxbytes));
    const int regions_kb[] = {110, 440, 1200};
    printf("256 workgroups, 1-KB fragments; GB/s per CU (aggregate TB/s)\n");
    for (int shared = 1; shared >= 0; --shared)
        for (int rk : regions_kb)
            for (int waves : {4, 8})
                for (int split = 0; split < 2; ++split) {
                    const int nfrag = rk;                                      // 1 KB per fragment
                    const int iters = std::max(1, 8 * 1024 / rk) * (split ? waves : 1);   // ~8 MB per wave
                    const int stride = shared ? 0 : nfrag;
                    if (!shared && (size_t)256 * nfrag * 1024 > maxbytes) continue;
                    float ms[4];
                    ms[0] = run<4>(buf, nfrag, iters, waves, stride, split, sink, st);
                    ms[1] = run<8>(buf, nfrag, iters, waves, stride, split, sink, st);
                    ms[2] = run<16>(buf, nfrag, iters, waves, stride, split, sink, st);
                    ms[3] = run<32>(buf, nfrag, iters, waves, stride, split, sink, st);
                    const int per = split ? (nfrag + waves - 1) / waves : nfrag;
                    const double bytes_per_cu = (double)per * iters * 1024.0 * waves;
                    printf("%s region %4d KB, %d waves/WG, %s:", shared ? "shared " : "private", rk, waves, split ? "waves split it " : "each wave reads all");
                    const int depth[4] = {4, 8, 16, 32};
                    for (int i = 0; i < 4; ++i)
                        printf("  D=%-2d %6.1f (%5.2f)", depth[i], bytes_per_cu / (ms[i] * 1e-3) / 1e9, bytes_per_cu * 256 / (ms[i] * 1e-3) / 1e12);
                    printf("\n");
                }
    return 0;
}
