// What does the core clock do under load?  s_memtime counts shader-core cycles (tools/ubench/mfma_valu.hip: 14 MFMAs of 32 cycles
// = 449 ticks), wall_clock64() is the constant 100 MHz counter: their ratio over a busy interval is the clock the CU actually ran
// at.  The GEMM main loop reaches 91 % matrix-pipe duty with 8 workgroups and 45 - 53 % with 256 IN CYCLES, so
// a lower clock under chip-wide MFMA load would come ON TOP of that in wall time (rocprof durations) — this probe measures it:
//   mode 0  dependent v_fma chain (light)      mode 1  back-to-back 32x32x16 MFMAs, 2 waves per SIMD (the GEMM's matrix load)
//   mode 2  MFMAs + 16-byte global loads from a private 1-MiB panel per workgroup (adds fabric / HBM power)
// for 8, 64, 256 and 1024 workgroups.   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(const uint4* __restrict__ panel, float* __restrict__ sink,
                                             long* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
    f16v acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float v = 0.01f * lane;
    uint4 ld = make_uint4(0, 0, 0, 0);
    const uint4* mine = panel + (size_t)blockIdx.x * 65536 + threadIdx.x;       // 1 MiB per workgroup
    __syncthreads();
    const long c0 = (long)__builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 64; ++k) v = __builtin_fmaf(v, 0.999f, 0.001f);
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            if (MODE == 2) {
                const uint4 x = mine[(it & 127) * 512];
                ld.x ^= x.x; ld.y ^= x.y; ld.z ^= x.z; ld.w ^= x.w;
            }
        }
    }
    const long c1 = (long)__builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = v + (float)(ld.x ^ ld.y ^ ld.z ^ ld.w);
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
static void run(const char* name, int grid, int iters, const uint4* panel, float* sink, long* out_d) {
    std::vector<long> out(2 * grid);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<grid, 512>>>(panel, sink, out_d, iters / 8);          // warm-up
    hipEventRecord(e0);
    probe<MODE><<<grid, 512>>>(panel, sink, out_d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(out.data(), out_d, out.size() * sizeof(long), hipMemcpyDeviceToHost);
    std::vector<double> mhz(grid);
    for (int i = 0; i < grid; ++i) mhz[i] = 100.0 * (double)out[2 * i] / (double)out[2 * i + 1];
    std::sort(mhz.begin(), mhz.end());
    printf("%-28s %5d workgroups: %8.2f ms  core clock min %7.1f  median %7.1f  max %7.1f MHz   (cycles per iteration %.1f)\n", name, grid, ms,
           mhz.front(), mhz[grid / 2], mhz.back(), (double)out[0] / iters);
}

int main() {
    uint4* panel; float* sink; long* out_d;
    hipMalloc(&panel, (size_t)1024 << 20);
    hipMemset(panel, 1, (size_t)1024 << 20);
    hipMalloc(&sink, 64);
    hipMalloc(&out_d, 2 * 1024 * sizeof(long));
    const int iters = 40000;                    // 40000 x 16 MFMAs x 32 cycles x 2 waves per SIMD ~ 41 M cycles ~ 20 ms
    for (int grid : {8, 64, 256, 1024}) {
        run<0>("v_fma chain", grid, iters * 4, panel, sink, out_d);
        run<1>("MFMA 32x32x16, 2 waves/SIMD", grid, iters, panel, sink, out_d);
        run<2>("MFMA + 16-byte loads", grid, iters, panel, sink, out_d);
    }
    return 0;
}
