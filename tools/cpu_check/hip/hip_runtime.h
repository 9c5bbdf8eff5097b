// Host-side stand-in for <hip/hip_runtime.h>: just enough of the HIP execution model to run the SIMPLE kernels of this
// repository (element-wise passes and wave / workgroup reductions; no MFMA, no LDS-DMA) on the CPU, from their real
// source, so that index arithmetic and reduction logic can be checked in the build container, which has no GPU.
// A workgroup runs as blockDim.x OS threads (blocks one after the other); __syncthreads is a std::barrier over the
// workgroup, __shfl_xor an exchange through a per-wave scratch array between two wave barriers (all 64 lanes of a wave
// must call it, as on the hardware), __shared__ is static storage (one workgroup is alive at a time).
// Used by tools/cpu_check/check_*.cpp only (with hip_gemm.h for the MFMA / LDS-DMA kernels); never part of a library.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <barrier>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#define __HIPCC__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#ifdef CPUHIP_DYNAMIC_LDS_ONLY        // the kernel only has `extern __shared__ smem[]`: see hip_gemm.h
#define __shared__
#else
#define __shared__ static
#endif
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
typedef int hipError_t;
typedef void* hipStream_t;
constexpr int hipSuccess = 0;

namespace cpuhip {
struct Ctx {
    dim3 tid, bid, bdim, gdim;
    std::barrier<>* block_bar = nullptr;
    std::barrier<>* wave_bar = nullptr;
    float* wave_slots = nullptr;
    void* wave_scratch = nullptr;      // 64 KiB per wave for collective emulations (MFMA)
    const void* kernarg = nullptr;     // the first kernel argument (what the kernarg segment starts with)
};
inline thread_local Ctx ctx;
}  // namespace cpuhip
#define threadIdx (cpuhip::ctx.tid)
#define blockIdx (cpuhip::ctx.bid)
#define blockDim (cpuhip::ctx.bdim)
#define gridDim (cpuhip::ctx.gdim)

static inline void __syncthreads() { cpuhip::ctx.block_bar->arrive_and_wait(); }
static inline float __shfl_xor(float v, int mask, int width = 64) {
    (void)width;
    const int lane = (int)(cpuhip::ctx.tid.x & 63);
    cpuhip::ctx.wave_slots[lane] = v;
    cpuhip::ctx.wave_bar->arrive_and_wait();
    const float r = cpuhip::ctx.wave_slots[lane ^ mask];
    cpuhip::ctx.wave_bar->arrive_and_wait();
    return r;
}
static inline float __shfl(float v, int src, int width = 64) {
    (void)width;
    const int lane = (int)(cpuhip::ctx.tid.x & 63);
    cpuhip::ctx.wave_slots[lane] = v;
    cpuhip::ctx.wave_bar->arrive_and_wait();
    const float r = cpuhip::ctx.wave_slots[src & 63];
    cpuhip::ctx.wave_bar->arrive_and_wait();
    return r;
}
#define __expf(x) expf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
using std::max;
using std::min;

template <typename A0, typename... Rest>
const void* cpuhip_first(const A0& a0, const Rest&...) { return &a0; }

template <typename K, typename... Args>
void cpuhip_launch(K kernel, dim3 grid, dim3 block, Args... args) {
    const void* kernarg = cpuhip_first(args...);
    const unsigned nthreads = block.x;
    const unsigned nwaves = (nthreads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                std::barrier<> block_bar((std::ptrdiff_t)nthreads);
                std::vector<std::unique_ptr<std::barrier<>>> wave_bars;
                std::vector<std::vector<float>> slots(nwaves, std::vector<float>(64, 0.f));
                std::vector<std::vector<unsigned char>> scratch(nwaves, std::vector<unsigned char>(65536));
                for (unsigned w = 0; w < nwaves; ++w) {
                    const unsigned lanes = std::min(64u, nthreads - w * 64);
                    wave_bars.emplace_back(new std::barrier<>((std::ptrdiff_t)lanes));
                }
                std::vector<std::thread> threads;
                for (unsigned t = 0; t < nthreads; ++t)
                    threads.emplace_back([&, t]() {
                        auto& c = cpuhip::ctx;
                        c.tid = dim3(t, 0, 0);
                        c.bid = dim3(bx, by, bz);
                        c.bdim = block;
                        c.gdim = grid;
                        c.block_bar = &block_bar;
                        c.wave_bar = wave_bars[t / 64].get();
                        c.wave_slots = slots[t / 64].data();
                        c.wave_scratch = scratch[t / 64].data();
                        c.kernarg = kernarg;
                        kernel(args...);
                        // a thread that returns early must not leave the others waiting at a later barrier
                        block_bar.arrive_and_drop();
                        wave_bars[t / 64]->arrive_and_drop();
                    });
                for (auto& th : threads) th.join();
            }
}
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) cpuhip_launch(kernel, grid, block, __VA_ARGS__)
static inline const char* hipGetErrorString(hipError_t) { return "cpu"; }
