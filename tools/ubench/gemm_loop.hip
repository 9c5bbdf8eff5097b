// Main-loop study for vsx_gemm_f16: what matrix-pipe duty can a 256x320x64-slab tile reach on gfx950 when the operand
// stream (LDS-DMA, 128-byte row segments, 2-slot ring), the fragment reads (swizzled ds_read_b128) and the MFMAs are
// scheduled
//   variant 8 : as the shipped persistent kernel does — 8 waves (2 per SIMD) of 64x160, ping-pong phases, a barrier per
//               phase (reference point measured inside this harness)
//   variant 4 : ONE wave per SIMD (4 waves of 128x160: 320 accumulators, needs the unified 512-register file), every
//               wave software-pipelined: the fragment reads of k-step s+1 and the wave's share of the next slab's DMA
//               pieces are issued BETWEEN the 20 MFMAs of k-step s; one barrier per slab.  Fragment traffic per MFMA
//               drops from 0.70 to 0.45 KiB, barriers from 8 to 1 per slab.
// No epilogue, no conv loader: A rows are private to the workgroup (M = 256 x #workgroups), B is shared, K is swept
// `nslab` slabs.  Prints TF/s and the matrix-pipe duty against 2.5 PF/s (and against the clock-true peak from s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o gemm_loop gemm_loop.hip && ./gemm_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 256, BN = 320, BKH = 64;               // tile, K slab in halfs (128-byte rows)
constexpr int STAGE = (BM + BN) * 128;                    // bytes per ring slot (73 728)
constexpr int NPIECES = (BM + BN) / 8;                    // 1-KiB pieces per slab (72)

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }
__device__ __forceinline__ void bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------------------
// variant 4: one wave per SIMD, 128 x 160 per wave
// ---------------------------------------------------------------------------------------------------------------------
// TM: 32-row blocks per wave (4: 128x160 per wave, tile 256x320, 320 accumulators; 3: 96x160, tile 192x320, 240
// accumulators = fits the 256 AGPRs).  NDMA0 / NDMA1: DMA pieces (of the wave's 2*TM + 10) issued in k-step 3 of slab
// t-1 / k-step 0 of slab t (the rest in k-step 1)
template <int TM, int NDMA0, int NDMA1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void loop4(const half_t* __restrict__ A, const half_t* __restrict__ B, float* __restrict__ sink, int K, int nslab) {
    constexpr int TN = 5, NPW = 2 * TM + 10, BMT = 64 * TM, STG = (BMT + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned ld2 = (unsigned)K * 2u;                   // row pitch in bytes
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(A + (size_t)blockIdx.x * BMT * K), 0, BMT * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(B), 0, BN * K * 2, 0x00020000);
    // DMA lane coordinates: lane -> (row of an 8-row piece, 16-byte slot), slot XOR-swizzled on the source side
    const int lrow = lane >> 3, pslot = lane & 7;
    const int kofs_e = (pslot ^ (lrow >> 1)) * 16, kofs_o = (pslot ^ (4 | (lrow >> 1))) * 16;
    const int v_e = (int)((unsigned)lrow * ld2) + kofs_e, v_o = (int)((unsigned)lrow * ld2) + kofs_o;
    // this wave's 18 pieces: 8 of A (rows wave*64 ..), 10 of B (rows wave*80 ..)
    int i_slab = 0;
    auto issue = [&](const int q, const int slab) {          // q compile-time
        const int slot = (slab & 1) * STG;
        const unsigned koff = (unsigned)(slab % (K / BKH)) * 128u;
        if (q < 2 * TM) {
            const int pc = wave * 2 * TM + q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(smem + slot + pc * 1024), 16, (pc & 1) ? v_o : v_e,
                                                     (int)(koff + (unsigned)(pc * 8) * ld2), 0, 0);
        } else {
            const int pc = wave * 10 + (q - 2 * TM);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lptr_t)(smem + slot + BMT * 128 + pc * 1024), 16,
                                                     (pc & 1) ? v_o : v_e, (int)(koff + (unsigned)(pc * 8) * ld2), 0, 0);
        }
    };
    const int fr = ((hi ^ ((l31 >> 1) & 7)) * 16);
    const int a_addr = (wr * 32 * TM + l31) * 128 + fr;
    const int b_addr = BMT * 128 + (wc * 160 + l31) * 128 + fr;
    f16v acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    h8 af[2][TM], bf[2][TN];
    auto ldfrag = [&](const int buf, const int slot_off, const int ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[buf][i] = *reinterpret_cast<const h8*>(smem + slot_off + ((a_addr ^ (ks * 32)) + i * 4096));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bf[buf][j] = *reinterpret_cast<const h8*>(smem + slot_off + ((b_addr ^ (ks * 32)) + j * 4096));
    };
    auto mma = [&](const int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[buf][j], af[buf][i], acc[j][i], 0, 0, 0);
    };
    // prologue: slab 0 (all 18 pieces), landed; fragments of (slab 0, k-step 0)
#pragma unroll
    for (int q = 0; q < NPW; ++q) issue(q, 0);
    wait_vmcnt<0>();
    bar();
    ldfrag(0, 0, 0);
    // steady state per slab t (slot so): k-steps 0..2 read the fragments of the next k-step between their MFMAs; k-step 0
    // also issues the rest of slab t+1's pieces; before k-step 3 the wave waits for its pieces of slab t+1 and everyone
    // meets (RAW for slab t+1, WAR for the slot slab t+2 will use = slab t's, whose last fragment reads were issued in
    // k-step 2 and are waited for here); k-step 3 prefetches (t+1, 0) and issues the first pieces of slab t+2.
    // The first NDMA0 pieces of slab 1 are issued here, as k-step 3 of a virtual slab -1 would
#pragma unroll
    for (int q = 0; q < NDMA0; ++q) issue(q, 1);
    for (int t = 0; t < nslab; ++t) {
        const int so = (t & 1) * STG, sn = ((t + 1) & 1) * STG;
        // ---- k-step 0 ----
        ldfrag(1, so, 1);
#pragma unroll
        for (int q = NDMA0; q < NDMA0 + NDMA1; ++q) issue(q, t + 1);
        lgkm0();                                              // fragments of k-step 0 (buffer 0) were read long ago
        __builtin_amdgcn_sched_barrier(0);
        mma(0);
        // ---- k-step 1 ----
#pragma unroll
        for (int q = NDMA0 + NDMA1; q < NPW; ++q) issue(q, t + 1);
        lgkm0();
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(0, so, 2);
        mma(1);
        // ---- k-step 2 ----
        lgkm0();
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(1, so, 3);
        mma(0);
        // ---- k-step 3 ----
        wait_vmcnt<0>();
        lgkm0();
        bar();
        ldfrag(0, sn, 0);
#pragma unroll
        for (int q = 0; q < NDMA0; ++q) issue(q, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(1);
        lgkm0();
    }
    wait_vmcnt<0>();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[j][i][e];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------------------------------
// variant 8: the shipped ping-pong structure (8 waves of 64x160, a barrier per phase), same harness
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void loop8(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                             float* __restrict__ sink, int K, int nslab, int a_private, int b_private, int stagger, int a_every) {
    constexpr int TM = 2, TN = 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1, G = wave >> 2;
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned ld2 = (unsigned)K * 2u;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(A + (a_private ? (size_t)blockIdx.x * BM * K : 0)), 0, BM * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(B + (b_private ? (size_t)blockIdx.x * BN * K : 0)), 0, BN * K * 2, 0x00020000);
    const int k_start = stagger * (int)(blockIdx.x >> 3);
    const int lrow = lane >> 3, pslot = lane & 7;
    const int kofs_e = (pslot ^ (lrow >> 1)) * 16, kofs_o = (pslot ^ (4 | (lrow >> 1))) * 16;
    const int v_e = (int)((unsigned)lrow * ld2) + kofs_e, v_o = (int)((unsigned)lrow * ld2) + kofs_o;
    auto issue = [&](const int q, const int slab) {          // 9 pieces per wave: 4 of A, 5 of B
        const int slot = (slab & 1) * STAGE;
        // stagger: workgroup w of an XCD starts its K sweep `stagger * w` slabs further (the CUs of an XCD then read
        // different lines of a shared panel at any moment instead of all hitting the same L2 lines together)
        const unsigned koff = (unsigned)((slab + k_start) % (K / BKH)) * 128u;
        if (q < 4) {
            if (slab % a_every) return;          // a_every = 3: the A panel is refreshed every third slab (conv with kw reuse)
            const int pc = wave * 4 + q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(smem + slot + pc * 1024), 16, (pc & 1) ? v_o : v_e,
                                                     (int)(koff + (unsigned)(pc * 8) * ld2), 0, 0);
        } else {
            const int pc = wave * 5 + (q - 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lptr_t)(smem + slot + BM * 128 + pc * 1024), 16,
                                                     (pc & 1) ? v_o : v_e, (int)(koff + (unsigned)(pc * 8) * ld2), 0, 0);
        }
    };
    const int fr = ((hi ^ ((l31 >> 1) & 7)) * 16);
    const int a_addr = (wr * 64 + l31) * 128 + fr;
    const int b_addr = BM * 128 + (wc * 160 + l31) * 128 + fr;
    f16v acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    h8 af[TM], bf[TN];
    auto ldfrag = [&](const int slot_off, const int ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[i] = *reinterpret_cast<const h8*>(smem + slot_off + ((a_addr ^ (ks * 32)) + i * 4096));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bf[j] = *reinterpret_cast<const h8*>(smem + slot_off + ((b_addr ^ (ks * 32)) + j * 4096));
    };
    auto mma = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[j][i], 0, 0, 0);
    };
#pragma unroll
    for (int q = 0; q < 9; ++q) issue(q, 0);
    wait_vmcnt<0>();
    bar();
    if (G == 1) bar();
    for (int t = 0; t < nslab; ++t) {
        const int so = (t & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            ldfrag(so, ks);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) { issue(0, t + 1); issue(1, t + 1); issue(2, t + 1); }
            if (ks == 1) { issue(3, t + 1); issue(4, t + 1); issue(5, t + 1); }
            if (ks == 2) { issue(6, t + 1); issue(7, t + 1); issue(8, t + 1); }
            if (ks == 3 && G == 1) wait_vmcnt<0>();
            lgkm0();
            bar();
            __builtin_amdgcn_s_setprio(1);
            mma();
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 3 && G == 0) wait_vmcnt<0>();
            bar();
        }
    }
    if (G == 0) bar();
    wait_vmcnt<0>();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[j][i][e];
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename F>
static double time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return 1e3 * ms / reps;
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256;
    const int K = argc > 2 ? atoi(argv[2]) : 2560;            // K elements of the operand panels (re-swept)
    const int nslab = argc > 3 ? atoi(argv[3]) : 160;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;            // 0: all variants; 1: variant 8 only, A/B sharing sweep
    half_t *A, *B;
    float* sink;
    const size_t na = (size_t)nwg * BM * K, nb = (size_t)(mode ? nwg : 1) * BN * K;
    (void)hipMalloc(&A, na * 2);
    (void)hipMalloc(&B, nb * 2);
    (void)hipMalloc(&sink, (size_t)nwg * 512 * 4);
    std::vector<half_t> h(na > nb ? na : nb);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (half_t)(((int)((i * 2654435761u) >> 20) % 255 - 127) / 128.0f);
    (void)hipMemcpy(A, h.data(), na * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, h.data(), nb * 2, hipMemcpyHostToDevice);
    const size_t smem = 2 * STAGE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&loop8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const double flop = 2.0 * BM * BN * BKH * (double)nslab * nwg;
    printf("# %d workgroups, K panel %d (A %.1f MB if private, B %.2f MB per panel), %d slabs per tile: %.1f GFLOP\n", nwg, K,
           na * 2 / 1e6, (double)BN * K * 2 / 1e6, nslab, flop / 1e9);
    auto report = [&](const char* name, double us) {
        printf("%-58s %9.1f us  %7.1f TF/s  %.3f of 2.5 PF/s\n", name, us, flop / us / 1e6, flop / us / 1e6 / 2500.0);
    };
    if (mode) {
        for (int st : {0, 1, 3, 7})
            for (int ap = 1; ap >= 0; --ap)
                for (int bp = 0; bp <= (st ? 0 : 1); ++bp) {
                    char name[96];
                    snprintf(name, sizeof(name), "8 waves ping-pong, A %s, B %s, K start + %d x wg", ap ? "private" : "shared ",
                             bp ? "private" : "shared ", st);
                    report(name, time_us([&] {
                        hipLaunchKernelGGL(loop8, dim3(nwg), dim3(512), smem, 0, A, B, sink, K, nslab, ap, bp, st, 1); }, 5));
                }
        for (int ae : {1, 3, 9})
            for (int ap = 1; ap >= 0; --ap) {
                char name[96];
                snprintf(name, sizeof(name), "8 waves ping-pong, A %s refreshed every %d slab(s), B shared", ap ? "private" : "shared ", ae);
                report(name, time_us([&] {
                    hipLaunchKernelGGL(loop8, dim3(nwg), dim3(512), smem, 0, A, B, sink, K, nslab, ap, 0, 0, ae); }, 5));
            }
        printf("status %s\n", hipGetErrorString(hipGetLastError()));
        return 0;
    }
    report("8 waves, ping-pong, barrier per phase", time_us([&] {
        hipLaunchKernelGGL(loop8, dim3(nwg), dim3(512), smem, 0, A, B, sink, K, nslab, 1, 0, 0, 1); }, 5));
#define RUN4(TM, N0, N1)                                                                                                \
    {                                                                                                                   \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&loop4<TM, N0, N1>),                                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                               \
        const double fl = flop * (64 * TM) / 256.0;                                                                     \
        const double us = time_us([&] {                                                                                 \
            hipLaunchKernelGGL((loop4<TM, N0, N1>), dim3(nwg), dim3(256), smem, 0, A, B, sink, K, nslab); }, 5);         \
        printf("%-46s %9.1f us  %7.1f TF/s  %.3f of 2.5 PF/s\n", "4 waves, " #TM "x32 rows per wave, pieces " #N0 "|" #N1 "|rest", \
               us, fl / us / 1e6, fl / us / 1e6 / 2500.0);                                                               \
    }
    RUN4(3, 5, 6)
    RUN4(3, 8, 8)
    RUN4(3, 0, 8)
    RUN4(4, 6, 6)
    printf("status %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
