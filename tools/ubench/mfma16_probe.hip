// Probe of the two gfx950 instructions the d = 40 flash-attention O tile is built on, run before the kernel is
// written against them ("measure, don't guess"):
//   1. v_permlane16_swap_b32 vdst, src : which 16-lane rows trade places (printed as the lane each output lane received);
//   2. v_mfma_f32_16x16x32_f16         : operand / result layout — hypothesis A: row = l % 16, k = 8 (l / 16) + e;
//      B: col = l % 16, k = 8 (l / 16) + e; D: col = l % 16, row = 4 (l / 16) + r — checked against a host product of asymmetric
//      random matrices; and its issue cost beside v_mfma_f32_32x32x16_f16 (cycles per instruction in a dependent-free chain).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma16_probe tools/ubench/mfma16_probe.hip && /tmp/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void swap_probe(unsigned* out) {
    const unsigned lane = threadIdx.x;
    const auto r = __builtin_amdgcn_permlane16_swap(lane, 100u + lane, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}

__global__ void mfma_probe(const _Float16* A, const _Float16* B, float* D) {      // A [16][32], B [32][16] row-major, D [16][16]
    const int l = threadIdx.x;
    h8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = A[(l % 16) * 32 + 8 * (l / 16) + e];
        b[e] = B[(8 * (l / 16) + e) * 16 + (l % 16)];
    }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + (l % 16)] = c[r];
}

template <int KIND>
__global__ void rate_probe(long* cycles, float* sink, int n) {
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (e + 1)); }
    f4 c4[8] = {};
    f16v c16[4] = {};
    const long t0 = (long)__builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (KIND == 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[k], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) c16[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16[k], 0, 0, 0);
        }
    }
    const long t1 = (long)__builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += c4[k][0];
    for (int k = 0; k < 4; ++k) s += c16[k][0];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = s;
}

int main() {
    unsigned* d_out;
    hipMalloc(&d_out, 128 * sizeof(unsigned));
    swap_probe<<<1, 64>>>(d_out);
    unsigned h[128];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("v_permlane16_swap(x = lane, y = 100 + lane): per 16-lane row, what the first lane of the row holds afterwards\n");
    for (int r = 0; r < 4; ++r) printf("  row %d: x' = %3u .. %3u   y' = %3u .. %3u\n", r, h[16 * r], h[16 * r + 15], h[64 + 16 * r], h[64 + 16 * r + 15]);

    _Float16 A[16 * 32], B[32 * 16];
    float ref[256] = {}, D[256];
    srand(7);
    for (int i = 0; i < 512; ++i) { A[i] = (_Float16)((rand() % 17 - 8) / 8.f); B[i] = (_Float16)((rand() % 13 - 6) / 4.f); }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            for (int k = 0; k < 32; ++k) ref[i * 16 + j] += (float)A[i * 32 + k] * (float)B[k * 16 + j];
    _Float16 *dA, *dB;
    float* dD;
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dD, sizeof(D));
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice);
    hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    mfma_probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
    float err = 0.f;
    for (int i = 0; i < 256; ++i) err = fmaxf(err, fabsf(D[i] - ref[i]));
    printf("v_mfma_f32_16x16x32_f16 layout hypothesis (A row = l%%16, k = 8(l/16)+e; B col = l%%16, same k; D col = l%%16, row = 4(l/16)+r): max |err| = %g -> %s\n",
           err, err < 1e-3f ? "CONFIRMED" : "WRONG");

    long* d_cyc; float* d_sink;
    hipMalloc(&d_cyc, 8 * sizeof(long)); hipMalloc(&d_sink, 8 * 64 * sizeof(float));
    long cyc[8];
    const int n = 2000;
    rate_probe<16><<<1, 64>>>(d_cyc, d_sink, n);
    hipMemcpy(cyc, d_cyc, sizeof(long), hipMemcpyDeviceToHost);
    const double c16x16 = (double)cyc[0] / (8.0 * n);
    rate_probe<32><<<1, 64>>>(d_cyc, d_sink, n);
    hipMemcpy(cyc, d_cyc, sizeof(long), hipMemcpyDeviceToHost);
    const double c32x32 = (double)cyc[0] / (4.0 * n);
    printf("issue cost, one wave, independent accumulators (s_memtime ticks per instruction; ratio is what matters): 16x16x32 %.2f, 32x32x16 %.2f, ratio %.3f (FLOP ratio 0.5)\n",
           c16x16, c32x32, c16x16 / c32x32);
    return 0;
}
