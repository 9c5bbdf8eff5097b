// Staging-loop model of the GEMM (no MFMAs): per slab, every workgroup stages AROWS private activation rows (pitch
// APITCH, window re-swept) and BROWS weight rows SHARED by all workgroups, in SEG-byte row segments; BTILED = the
// weight slab is one contiguous block (pre-tiled layout) instead of BROWS strided segments.
//   hipcc --offload-arch=gfx950 -O3 -o stage2 stage2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int THREADS = 512, NW = 8;

template <int AROWS, int BROWS, int SEG, int NST, bool BTILED, int APITCH, int BPITCH>
__global__ __launch_bounds__(THREADS) void stage(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b,
                                                 uint4* __restrict__ sink, int nslab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int RPP = 1024 / SEG, LPR = SEG / 16;
    constexpr int AP = AROWS / RPP, BP = BROWS / RPP;
    constexpr int GA = (AP + NW - 1) / NW, GB = (BP + NW - 1) / NW;
    constexpr int SLAB = (AROWS + BROWS) * SEG;
    constexpr int KSA = APITCH / SEG, KSB = BPITCH / SEG;
    const unsigned char* abase = a + (size_t)blockIdx.x * AROWS * APITCH;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(abase), 0, AROWS * APITCH, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(b), 0, BROWS * BPITCH, 0x00020000);
    auto issue = [&](int kt) {
        unsigned char* sb = smem + (kt % NST) * SLAB;
#pragma unroll
        for (int g = 0; g < GA; ++g) {
            const int pc = min(wave * GA + g, AP - 1);
            const int row = pc * RPP + lane / LPR;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(sb + pc * 1024), 16, row * APITCH + (kt % KSA) * SEG + (lane % LPR) * 16, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const int pc = min(wave * GB + g, BP - 1);
            int off;
            if (BTILED) off = ((kt % KSB) * BP + pc) * 1024 + lane * 16;
            else off = (pc * RPP + lane / LPR) * BPITCH + (kt % KSB) * SEG + (lane % LPR) * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(sb + AROWS * SEG + pc * 1024), 16, off, 0, 0, 0);
        }
    };
    for (int s = 0; s < NST - 1; ++s) issue(s);
    uint4 acc = {0, 0, 0, 0};
    for (int kt = 0; kt < nslab; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (GA + GB)) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(kt + NST - 1);
        const uint4 v = *reinterpret_cast<const uint4*>(smem + (kt % NST) * SLAB + tid * 16);
        acc.x ^= v.x; acc.y ^= v.y;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x == 0x12345678u) sink[blockIdx.x * THREADS + tid] = acc;
}

template <int AROWS, int BROWS, int SEG, int NST, bool BTILED, int APITCH, int BPITCH>
static void run(const unsigned char* a, const unsigned char* b, uint4* sink, int blocks, int nslab) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int lds = NST * (AROWS + BROWS) * SEG;
    auto k = stage<AROWS, BROWS, SEG, NST, BTILED, APITCH, BPITCH>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(THREADS), lds, 0, a, b, sink, 64);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(THREADS), lds, 0, a, b, sink, nslab);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * (AROWS + BROWS) * SEG * nslab;
    printf("A %3d rows pitch %5d + B %3d rows pitch %5d %-7s seg %3d ring %d blocks %3d: %6.3f us per %5.1f KB  %8.1f GB/s total %6.1f per CU  (%s)\n",
           AROWS, APITCH, BROWS, BPITCH, BTILED ? "tiled" : "strided", SEG, NST, blocks, ms * 1e3 / nslab * (32.0 * 1024 / ((AROWS + BROWS) * SEG)) ,
           32.0, bytes / ms / 1e6, bytes / ms / 1e6 / blocks, hipGetErrorString(hipGetLastError()));
}

int main() {
    unsigned char *a, *b;
    uint4* sink;
    (void)hipMalloc(&a, (size_t)1 << 30);
    (void)hipMalloc(&b, (size_t)64 << 20);
    (void)hipMalloc(&sink, (size_t)1024 * THREADS * 16);
    (void)hipMemset(a, 1, (size_t)1 << 30);
    (void)hipMemset(b, 1, (size_t)64 << 20);
    for (int blocks : {256}) {
        // K = 320 projection (A private 80 KB window, L2-resident; B 200 KB shared)
        run<128, 320, 64, 4, false, 640, 640>(a, b, sink, blocks, 2000);
        run<128, 320, 64, 4, true, 640, 640>(a, b, sink, blocks, 2000);
        run<128, 320, 128, 2, false, 640, 640>(a, b, sink, blocks, 1000);
        run<128, 320, 128, 2, true, 640, 640>(a, b, sink, blocks, 1000);
        // conv-like: A rows 640 B pitch (C = 320), B K = 2880 (pitch 5760)
        run<256, 320, 64, 4, false, 640, 5760>(a, b, sink, blocks, 2000);
        run<256, 320, 64, 4, true, 640, 5760>(a, b, sink, blocks, 2000);
        run<256, 320, 128, 2, false, 640, 5760>(a, b, sink, blocks, 1000);
        run<256, 320, 128, 2, true, 640, 5760>(a, b, sink, blocks, 1000);
        // wide K rows (pitch 2560: K = 1280), A window 128 x 2560 = 320 KB per block (not L2-resident at 256 blocks)
        run<128, 320, 64, 4, false, 2560, 2560>(a, b, sink, blocks, 2000);
        run<128, 320, 64, 4, true, 2560, 2560>(a, b, sink, blocks, 2000);
        run<128, 320, 128, 2, true, 2560, 2560>(a, b, sink, blocks, 1000);
        // B only (A tiny) to see the shared-weight stream alone
        run<16, 320, 64, 4, false, 640, 5760>(a, b, sink, blocks, 2000);
        run<16, 320, 64, 4, true, 640, 5760>(a, b, sink, blocks, 2000);
    }
    printf("status %s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
