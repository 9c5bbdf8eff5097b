// Issue cost (core cycles per wave64 instruction, one wave per SIMD) of the VALU instructions the softmax / GELU epilogues are made
// of: which transcendental forms are quarter rate, and whether the packed-fp16 forms are cheaper than packed fp32.  Dependent chains
// of NCHAIN independent registers so that latency is hidden and the number is issue throughput.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NCHAIN = 8, REP = 16;

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* __restrict__ sink, long* __restrict__ cycles, int iters) {
    const int lane = threadIdx.x & 63;
    float v[NCHAIN];
    h2 hv[NCHAIN];
    f2 fv[NCHAIN];
    _Float16 sv[NCHAIN];
    for (int k = 0; k < NCHAIN; ++k) {
        v[k] = 0.5f + 0.001f * (lane + k);
        hv[k] = h2{(_Float16)(0.5f + 0.001f * lane), (_Float16)(0.25f + 0.002f * k)};
        fv[k] = f2{0.5f + 0.001f * lane, 0.25f + 0.002f * k};
        sv[k] = (_Float16)(0.5f + 0.001f * (lane + k));
    }
    __syncthreads();
    const long t0 = (long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r)
#pragma unroll
            for (int k = 0; k < NCHAIN; ++k) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k]) : "v"(v[(k + 1) % NCHAIN]));     // v_fma_f32
                if (MODE == 1) v[k] = __builtin_amdgcn_exp2f(v[k]);                                  // v_exp_f32
                if (MODE == 2) v[k] = __builtin_amdgcn_rcpf(v[k]);                                   // v_rcp_f32
                if (MODE == 3) fv[k] = fv[k] * 0.999f + 0.001f;                                     // v_pk_fma_f32
                if (MODE == 4) hv[k] = hv[k] * (_Float16)0.999f + (_Float16)0.001f;                 // v_pk_fma_f16
                if (MODE == 5) asm volatile("v_exp_f16 %0, %0" : "+v"(sv[k]));                      // v_exp_f16
                if (MODE == 6) asm volatile("v_rcp_f16 %0, %0" : "+v"(sv[k]));                      // v_rcp_f16
                if (MODE == 7) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(hv[k]) : "v"(v[k]));   // f32 pair -> packed f16
                if (MODE == 8) v[k] = __builtin_amdgcn_sqrtf(v[k]);                                  // v_sqrt_f32
                if (MODE == 9) asm volatile("v_log_f32 %0, %0" : "+v"(v[k]));                       // v_log_f32
                if (MODE == 10) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v[k]));             // v_max3_f32
                if (MODE == 11) asm volatile("v_pk_mul_f16 %0, %0, %0" : "+v"(hv[k]));              // v_pk_mul_f16
                if (MODE == 12) asm volatile("v_pk_max_f16 %0, %0, %0" : "+v"(hv[k]));              // v_pk_max_f16
                if (MODE == 13) asm volatile("v_pk_add_u16 %0, %0, %0" : "+v"(hv[k]));              // v_pk_add_u16 (exponent arithmetic)
                if (MODE == 14) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(v[k]));                 // v_ldexp_f32
                if (MODE == 15) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %1" : "=v"(hv[k]) : "v"(v[k]));  // round toward zero
                if (MODE == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(v[(k + 1) % NCHAIN]));
                if (MODE == 17) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[k]) : "v"(v[(k + 1) % NCHAIN]));
            }
    }
    const long t1 = (long)__builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < NCHAIN; ++k) s += v[k] + (float)hv[k][0] + (float)hv[k][1] + fv[k][0] + fv[k][1] + (float)sv[k];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// round 4: the same chains with 1, 2 and 4 waves per SIMD (workgroups of 256 / 512 / 1024 threads, one per CU).  One wave
// alone cannot issue faster than one VALU instruction per ~5 cycles, whatever the instruction; what a kernel with several
// waves per SIMD (flash attention: 4) pays per instruction is the SIMD's issue time = cycles / waves in the last column.
template <int MODE>
static void run(const char* name, float* sink, long* out_d) {
    const int grid = 256, iters = 2000;
    printf("%-36s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {
        probe<MODE><<<grid, 256 * wps>>>(sink, out_d, 100);
        probe<MODE><<<grid, 256 * wps>>>(sink, out_d, iters);
        std::vector<long> out(grid);
        hipMemcpy(out.data(), out_d, grid * sizeof(long), hipMemcpyDeviceToHost);
        long mx = 0;
        for (long c : out) mx = c > mx ? c : mx;
        const double per = (double)mx / ((double)iters * REP * NCHAIN);
        printf("  %d wave%s/SIMD: %5.2f per wave instr = %5.2f SIMD cycles each", wps, wps > 1 ? "s" : " ", per, per / wps);
    }
    printf("\n");
}

int main() {
    float* sink; long* out_d;
    hipMalloc(&sink, 64);
    hipMalloc(&out_d, 256 * sizeof(long));
    run<0>("v_fma_f32", sink, out_d);
    run<3>("v_pk_fma_f32 (2 values per lane)", sink, out_d);
    run<4>("v_pk_fma_f16 (2 values per lane)", sink, out_d);
    run<11>("v_pk_mul_f16", sink, out_d);
    run<12>("v_pk_max_f16", sink, out_d);
    run<13>("v_pk_add_u16", sink, out_d);
    run<10>("v_max3_f32", sink, out_d);
    run<14>("v_ldexp_f32", sink, out_d);
    run<17>("v_mul_f32", sink, out_d);
    run<16>("v_cndmask_b32", sink, out_d);
    run<7>("v_cvt_pk_f16_f32", sink, out_d);
    run<15>("v_cvt_pkrtz_f16_f32", sink, out_d);
    run<1>("v_exp_f32", sink, out_d);
    run<5>("v_exp_f16", sink, out_d);
    run<2>("v_rcp_f32", sink, out_d);
    run<6>("v_rcp_f16", sink, out_d);
    run<8>("v_sqrt_f32", sink, out_d);
    run<9>("v_log_f32", sink, out_d);
    return 0;
}
