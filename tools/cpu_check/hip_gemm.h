// Additions to hip/hip_runtime.h for the MFMA / LDS-DMA kernels (gemm_pp.hip): the gfx950 builtins they use, mapped to
// host code with the hardware's data layout.
//   * v_mfma_f32_32x32x16_f16: a wave-collective; lane l holds A[row l%32][k = 8*(l/32) .. +7], B[k = 8*(l/32) .. +7][col
//     l%32] and D[row 8*(r/4) + 4*(l/32) + r%4][col l%32] in register r.
//   * buffer_load_dwordx4 ... lds: lane l moves 16 bytes from base + voffset + soffset to LDS at (M0 base) + 16*l; the
//     descriptor's range check looks at voffset only (+16 <= num_records), out of range reads as zero.  The harness ALSO
//     checks that an in-range lane never reads beyond the allocation (on the GPU that would be a silent over-read).
//   * s_barrier = workgroup barrier; sched_barrier = wave rendezvous (see below); waitcnt / setprio are no-ops (the emulated DMA is synchronous, so the
//     ASYNCHRONOUS hazards of a schedule are not visible here — only whether every piece is issued exactly once, to the
//     right place, from the right address).
// Define CPUHIP_DYNAMIC_LDS_ONLY before including hip/hip_runtime.h.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

// `extern __shared__ unsigned char smem[];` inside a kernel becomes a block-scope redeclaration of `cpuhip_smem`: the
// harness defines that array (160 KiB, 16-byte aligned) in the namespace that encloses the kernel.
#define smem cpuhip_smem
#define CPUHIP_DEFINE_LDS alignas(16) unsigned char cpuhip_smem[160 * 1024];

struct cpuhip_rsrc { const unsigned char* base; unsigned bytes; };
inline long cpuhip_oob_reads = 0;
#define __amdgpu_buffer_rsrc_t cpuhip_rsrc
#define __builtin_amdgcn_make_buffer_rsrc(ptr, stride, records, flags) \
    cpuhip_rsrc{reinterpret_cast<const unsigned char*>(ptr), (unsigned)(records)}

// When does an LDS-DMA piece land?  Anywhere between its issue and the issuing wave's next `s_waitcnt vmcnt(0)`.  The
// two extremes are emulated: CPUHIP_DMA=early (default) writes LDS at issue — a piece aimed at a ring slot that some wave
// is still reading (write-after-read hazard) corrupts the result; CPUHIP_DMA=late keeps the 16 bytes in a per-lane queue
// and writes them at the wave's wait — a piece issued after the wait that was supposed to cover it (read-before-landing)
// leaves stale data in LDS.  A schedule that is correct under both has no ordering hazard in this model.
struct cpuhip_pending { unsigned char* dst; unsigned char data[16]; };
inline thread_local std::vector<cpuhip_pending> cpuhip_queue;
static inline bool cpuhip_dma_late() {
    static const bool late = getenv("CPUHIP_DMA") && !strcmp(getenv("CPUHIP_DMA"), "late");
    return late;
}
static inline void cpuhip_land(unsigned char* dst, const unsigned char* src16) {
    if (cpuhip_dma_late()) {
        cpuhip_pending p;
        p.dst = dst;
        if (src16) memcpy(p.data, src16, 16); else memset(p.data, 0, 16);
        cpuhip_queue.push_back(p);
    } else if (src16) {
        memcpy(dst, src16, 16);
    } else {
        memset(dst, 0, 16);
    }
}
static inline void cpuhip_wait_vmcnt(int n) {       // all but the n most recent pieces of this lane have landed
    const size_t keep = (size_t)n < cpuhip_queue.size() ? (size_t)n : cpuhip_queue.size();
    const size_t done = cpuhip_queue.size() - keep;
    for (size_t i = 0; i < done; ++i)
        if (cpuhip_queue[i].dst) memcpy(cpuhip_queue[i].dst, cpuhip_queue[i].data, 16);
    cpuhip_queue.erase(cpuhip_queue.begin(), cpuhip_queue.begin() + (long)done);
}

// Ordinary VGPR-destination loads share the vmcnt queue with the DMA pieces: a kernel that keeps such loads in flight across
// counted slab waits (gemm.hip's residual prefetch behind the last slab) says so with VSX_VMEM_NOTE(n), and the late-landing
// model queues n placeholders so that its counts match the hardware's.
static inline void cpuhip_note_vmem(int n) {
    if (!cpuhip_dma_late()) return;
    for (int i = 0; i < n; ++i) {
        cpuhip_pending p;
        p.dst = nullptr;
        cpuhip_queue.push_back(p);
    }
}
#define VSX_VMEM_NOTE(n) cpuhip_note_vmem(n)

template <typename LdsPtr>
static inline void cpuhip_buffer_load_lds(cpuhip_rsrc r, LdsPtr lds, int size, int voffset, int soffset, int ioffset, int) {
    unsigned char* dst = (unsigned char*)(uintptr_t)lds + 16 * (cpuhip::ctx.tid.x & 63);
    const unsigned long off = (unsigned long)(unsigned)voffset + (unsigned)ioffset;
    if (size != 16) abort();
    if (off + 16 > r.bytes) {
        cpuhip_land(dst, nullptr);
        return;
    }
    const unsigned long addr = off + (unsigned long)(unsigned)soffset;
    if (addr + 16 > r.bytes) {                       // in range for the descriptor, but past the tensor
        __atomic_add_fetch(&cpuhip_oob_reads, 1, __ATOMIC_RELAXED);
        cpuhip_land(dst, nullptr);
        return;
    }
    cpuhip_land(dst, r.base + addr);
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, size, voff, soff, ioff, aux) \
    cpuhip_buffer_load_lds(rsrc, lds, size, voff, soff, ioff, aux)

typedef _Float16 cpuhip_h8 __attribute__((ext_vector_type(8)));
typedef float cpuhip_f16v __attribute__((ext_vector_type(16)));
static inline cpuhip_f16v cpuhip_mfma_32x32x16(cpuhip_h8 a, cpuhip_h8 b, cpuhip_f16v c, int, int, int) {
    struct Slot { cpuhip_h8 a, b; };
    Slot* s = static_cast<Slot*>(cpuhip::ctx.wave_scratch);
    const int l = (int)(cpuhip::ctx.tid.x & 63);
    s[l].a = a;
    s[l].b = b;
    cpuhip::ctx.wave_bar->arrive_and_wait();
    const int col = l & 31, hi = l >> 5;
    cpuhip_f16v d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r >> 2) + 4 * hi + (r & 3);
        float acc = 0.f;
        for (int k = 0; k < 16; ++k) acc += (float)s[row + 32 * (k >> 3)].a[k & 7] * (float)s[col + 32 * (k >> 3)].b[k & 7];
        d[r] += acc;
    }
    cpuhip::ctx.wave_bar->arrive_and_wait();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) cpuhip_mfma_32x32x16(a, b, c, x, y, z)
// v_mfma_f32_16x16x32_f16 with the hardware's layout (tools/ubench/mfma16_probe.hip confirmed it on an MI355X): lane l supplies
// A[l % 16][8 (l / 16) + e] and B[8 (l / 16) + e][l % 16], and owns D[4 (l / 16) + r][l % 16]
typedef float cpuhip_f4v __attribute__((ext_vector_type(4)));
static inline cpuhip_f4v cpuhip_mfma_16x16x32(cpuhip_h8 a, cpuhip_h8 b, cpuhip_f4v c, int, int, int) {
    struct Slot { cpuhip_h8 a, b; };
    Slot* s = static_cast<Slot*>(cpuhip::ctx.wave_scratch);
    const int l = (int)(cpuhip::ctx.tid.x & 63);
    s[l].a = a;
    s[l].b = b;
    cpuhip::ctx.wave_bar->arrive_and_wait();
    const int col = l & 15, g = l >> 4;
    cpuhip_f4v d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k) acc += (float)s[row + 16 * (k >> 3)].a[k & 7] * (float)s[col + 16 * (k >> 3)].b[k & 7];
        d[r] += acc;
    }
    cpuhip::ctx.wave_bar->arrive_and_wait();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) cpuhip_mfma_16x16x32(a, b, c, x, y, z)
// v_permlane16_swap vdst, src (probed on an MI355X, profiles/r06_gemm_data_power_same_box.txt): the odd 16-lane rows of vdst trade
// places with the even rows of src — vdst' = {vdst.row0, src.row0, vdst.row2, src.row2}, src' = {vdst.row1, src.row1, vdst.row3,
// src.row3}; returns (vdst', src')
typedef unsigned cpuhip_u2v __attribute__((ext_vector_type(2)));
static inline cpuhip_u2v cpuhip_permlane16_swap(unsigned vdst, unsigned src) {
    struct Slot { unsigned x, y; };
    Slot* s = static_cast<Slot*>(cpuhip::ctx.wave_scratch);
    const int l = (int)(cpuhip::ctx.tid.x & 63);
    s[l].x = vdst;
    s[l].y = src;
    cpuhip::ctx.wave_bar->arrive_and_wait();
    cpuhip_u2v r;
    if (((l >> 4) & 1) == 0) { r[0] = s[l].x; r[1] = s[l + 16].x; }
    else { r[0] = s[l - 16].y; r[1] = s[l].y; }
    cpuhip::ctx.wave_bar->arrive_and_wait();
    return r;
}
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) cpuhip_permlane16_swap(a, b)
#define __builtin_amdgcn_s_barrier() (cpuhip::ctx.block_bar->arrive_and_wait())
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
// The 64 lanes of a wave run in lockstep on the hardware, so a wave may write LDS and read it back (other lanes' data)
// without any barrier — the epilogue's staging area does.  Here the lanes are independent threads: every
// scheduling barrier in the source (they sit exactly between such phases) doubles as a wave-wide rendezvous.
#define __builtin_amdgcn_sched_barrier(x) (cpuhip::ctx.wave_bar->arrive_and_wait())
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_kernarg_segment_ptr() (const_cast<void*>(cpuhip::ctx.kernarg))

// device queries of the launcher
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
inline int cpuhip_num_cus = 8;
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = cpuhip_num_cus; return hipSuccess; }

// `asm volatile("s_waitcnt ...")` / `asm volatile("" : "+s"(x))`: AMDGPU text, meaningless here.  (Every system header this
// translation unit needs is already included above.)  gemm_common.h's wait_vmcnt<N>() is renamed away and replaced by
// one that lands the queued DMA pieces (CPUHIP_GEMM_COMMON_INCLUDED: include gemm_common.h through this header).
#define asm
#define volatile(...)
#define wait_vmcnt cpuhip_unused_wait_vmcnt
#include "gemm_common.h"
#undef wait_vmcnt
namespace vsxg {
template <int N>
inline void wait_vmcnt() { cpuhip_wait_vmcnt(N); }
}

// v_dot2_f32_f16: acc + a0 * b0 + a1 * b1 in fp32 (the products of two fp16 values are exact in fp32)
typedef _Float16 cpuhip_h2 __attribute__((ext_vector_type(2)));
static inline float cpuhip_fdot2(cpuhip_h2 a, cpuhip_h2 b, float acc) { return acc + (float)a[0] * (float)b[0] + (float)a[1] * (float)b[1]; }
#define __builtin_amdgcn_fdot2(a, b, acc, clamp) cpuhip_fdot2(a, b, acc)
