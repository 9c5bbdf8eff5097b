// Microbenchmark: how many bytes per cycle can one CU pull from L2 — (a) global_load_dwordx4 -> VGPR, (b) the same
// plus ds_write_b128 into LDS, (c) LDS-DMA (buffer_load ... lds).  Each workgroup (8 waves) sweeps its own 256 KiB
// window (L1 32 KiB -> misses, L2 4 MiB per XCD -> hits after the first pass).
//   hipcc --offload-arch=gfx950 -O3 -o l2_delivery l2_delivery.hip && ./l2_delivery
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lptr_t;
#ifndef WIN_KB
#define WIN_KB 256
#endif
constexpr int WIN = WIN_KB * 1024;     // bytes per workgroup window
constexpr int THREADS = 512;

template <int MODE>
__global__ __launch_bounds__(THREADS) void pull(const uint4* __restrict__ src, uint4* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const uint4* base = src + (size_t)blockIdx.x * (WIN / 16);
    uint4 acc = {0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(base), 0, WIN, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        // one sweep = WIN bytes = 32 steps of 8 KiB (512 threads x 16 B)
#pragma unroll 4
        for (int s = 0; s < WIN / (THREADS * 16); ++s) {
            const int idx = s * THREADS + tid;
            if (MODE == 0) {
                const uint4 v = base[idx];
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            } else if (MODE == 1) {
                const uint4 v = base[idx];
                *reinterpret_cast<uint4*>(smem + ((s & 7) * THREADS + tid) * 16) = v;
            } else {
                // lane-linear destination: wave's 1 KiB piece at ring slot (s & 7)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + ((s & 7) * THREADS + wave * 64) * 16), 16,
                                                         idx * 16, 0, 0, 0);
            }
        }
    }
    if (MODE != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc = *reinterpret_cast<const uint4*>(smem + tid * 16);
    }
    if (acc.x == 0x12345678u) sink[blockIdx.x * THREADS + tid] = acc;
}

// GEMM-like staging: a [ROWS x PITCH] fp16 operand tile read in 64-byte K-slabs (4 lanes per row, 16 rows per
// 1-KiB DMA piece), i.e. half of every 128-byte line per slab, the other half one slab later.
template <int ROWS, int PITCH>
__global__ __launch_bounds__(THREADS) void pull_slabs(const uint4* __restrict__ src, uint4* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(src) + (size_t)blockIdx.x * ROWS * PITCH;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, ROWS * PITCH, 0x00020000);
    constexpr int PIECES = ROWS / 16;            // per slab
    int ring = 0;
    for (int it = 0; it < iters; ++it)
        for (int ks = 0; ks < PITCH / 64; ++ks) {
#pragma unroll
            for (int pc = wave; pc < PIECES; pc += THREADS / 64) {
                const int row = pc * 16 + (lane >> 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (ring * PIECES + pc) * 1024), 16,
                                                         row * PITCH + ks * 64 + (lane & 3) * 16, 0, 0, 0);
            }
            ring = (ring + 1) & 1;
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint4 acc = *reinterpret_cast<const uint4*>(smem + tid * 16);
    if (acc.x == 0x12345678u) sink[blockIdx.x * THREADS + tid] = acc;
}

template <int ROWS, int PITCH>
static void run_slabs(const uint4* src, uint4* sink, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((pull_slabs<ROWS, PITCH>), dim3(blocks), dim3(THREADS), 64 * 1024, 0, src, sink, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL((pull_slabs<ROWS, PITCH>), dim3(blocks), dim3(THREADS), 64 * 1024, 0, src, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * ROWS * PITCH * iters;
    printf("LDS-DMA slabs %4d x %5d B      blocks %4d: %8.1f us  %8.1f GB/s total  %6.1f GB/s per CU  (%5.1f B/cyc/CU)\n", ROWS,
           PITCH, blocks, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / blocks, bytes / (ms * 1e-3) / blocks / 2.4e9);
}

// The GEMM's staging loop without the MFMAs: ring of NST slabs of ROWS x 64 B, per iteration
// s_waitcnt vmcnt((NST-2)*G) ; s_barrier ; issue the slab NST-1 ahead.  L2-resident window (ROWS x PITCH per block).
template <int ROWS, int PITCH, int NST, int SEG>
__global__ __launch_bounds__(THREADS) void stage_loop(const uint4* __restrict__ src, uint4* __restrict__ sink, int nslab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(src) + (size_t)blockIdx.x * ROWS * PITCH;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, ROWS * PITCH, 0x00020000);
    constexpr int RPP = 1024 / SEG;              // rows per 1-KiB piece
    constexpr int LPR = SEG / 16;                // lanes per row
    constexpr int PIECES = ROWS / RPP;
    constexpr int G = (PIECES + THREADS / 64 - 1) / (THREADS / 64);     // pieces per wave per slab (uniform: padded)
    constexpr int KS = PITCH / SEG;
    auto issue = [&](int kt) {
        unsigned char* sb = smem + (kt % NST) * ROWS * SEG;
        const int ks = kt % KS;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int pc = min(wave * G + g, PIECES - 1);
            const int row = pc * RPP + lane / LPR;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(sb + pc * 1024), 16, row * PITCH + ks * SEG + (lane % LPR) * 16,
                                                     0, 0, 0);
        }
    };
    for (int s = 0; s < NST - 1; ++s) issue(s);
    uint4 acc = {0, 0, 0, 0};
    for (int kt = 0; kt < nslab; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * G) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(kt + NST - 1);
        const uint4 v = *reinterpret_cast<const uint4*>(smem + (kt % NST) * ROWS * SEG + tid * 16);
        acc.x ^= v.x; acc.y ^= v.y;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x == 0x12345678u) sink[blockIdx.x * THREADS + tid] = acc;
}

template <int ROWS, int PITCH, int NST, int SEG>
static void run_stage(const uint4* src, uint4* sink, int blocks, int nslab) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int lds = NST * ROWS * SEG;
    hipFuncSetAttribute((const void*)stage_loop<ROWS, PITCH, NST, SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((stage_loop<ROWS, PITCH, NST, SEG>), dim3(blocks), dim3(THREADS), lds, 0, src, sink, 64);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stage_loop<ROWS, PITCH, NST, SEG>), dim3(blocks), dim3(THREADS), lds, 0, src, sink, nslab);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * ROWS * SEG * nslab;
    printf("stage loop %3d rows x%3dB ring %d  blocks %4d: %8.1f us  %6.3f us/slab  %8.1f GB/s total  %6.1f GB/s per CU\n", ROWS, SEG, NST,
           blocks, ms * 1e3, ms * 1e3 / nslab, bytes / ms / 1e6, bytes / ms / 1e6 / blocks);
}

template <int MODE>
static void run(const char* name, const uint4* src, uint4* sink, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(pull<MODE>, dim3(blocks), dim3(THREADS), 64 * 1024, 0, src, sink, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(pull<MODE>, dim3(blocks), dim3(THREADS), 64 * 1024, 0, src, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * WIN * iters;
    printf("%-28s blocks %4d: %8.1f us  %8.1f GB/s total  %6.1f GB/s per CU  (%5.1f B/cyc/CU at 2.4 GHz)\n", name, blocks,
           ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / blocks, bytes / (ms * 1e-3) / blocks / 2.4e9);
}

int main() {
    uint4 *src, *sink;
    const size_t SRC_BYTES = (size_t)1 << 30;
    hipMalloc(&src, SRC_BYTES);
    hipMalloc(&sink, (size_t)1024 * THREADS * 16);
    hipMemset(src, 1, SRC_BYTES);
    for (int blocks : {8, 64, 256}) {
        run<0>("global_load -> VGPR", src, sink, blocks, 200);
        run<1>("global_load -> ds_write_b128", src, sink, blocks, 200);
        run<2>("LDS-DMA (buffer_load lds)", src, sink, blocks, 200);
    }
    for (int blocks : {8, 256}) {
        run_stage<448, 256, 4, 64>(src, sink, blocks, 2000);
        run_stage<448, 256, 2, 128>(src, sink, blocks, 1000);
        run_stage<448, 256, 2, 256>(src, sink, blocks, 500);
        run_stage<448, 640, 4, 64>(src, sink, blocks, 2000);
        run_stage<448, 640, 2, 128>(src, sink, blocks, 1000);
        run_stage<224, 1280, 4, 64>(src, sink, blocks, 2000);
        run_stage<224, 1280, 2, 128>(src, sink, blocks, 1000);
    }
    for (int blocks : {8, 64, 256}) {
        run_slabs<448, 640>(src, sink, blocks, 400);        // 128x320 tile of a K = 320 projection (A + B rows)
        run_slabs<448, 5760>(src, sink, blocks, 40);        // K = 2880 conv rows (window 2.5 MB per block: beyond L2 at 256)
        run_slabs<128, 2560>(src, sink, blocks, 200);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return 0;
}
