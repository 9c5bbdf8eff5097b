// What does a matrix instruction cost in ENERGY on this board?  Register-resident MFMA streams on all 256 CUs (no LDS, no memory in the
// loop), random fp16 operands against zeros, 32x32x16 against 16x16x32, 1 / 2 / 4 waves per SIMD, while the host samples the board's
// power and shader clock from sysfs (this device, by PCI address).  Under the 1 400 W cap the rate a kernel reaches on real data is
// energy per FLOP, so this is the roofline the GEMM / attention kernels are priced against in DESIGN.md §3.6: the rate of a loop that
// does NOTHING but issue MFMAs on live data.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/ubench/mfma_power.hip && /tmp/mfma_power [seconds per row]
#include <hip/hip_runtime.h>
#include <dirent.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// operands: 8 A and 8 B fragments per lane from memory (random or zero), rotated through the independent accumulator chains
template <int KIND>
__global__ __launch_bounds__(256) void mfma_stream(const h8* __restrict__ ops, float* sink, int iters, int prio_from) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    // prio_from > 0: the workgroups dispatched in the second round (they share SIMDs with the first round's) run at s_setprio 3, so one
    // wave of a SIMD always wins the arbitration: does the 2-waves-per-SIMD rate recover when the waves stop alternating?
    if (prio_from > 0 && (int)blockIdx.x >= prio_from) __builtin_amdgcn_s_setprio(3);
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = ops[(size_t)t * 8 + i]; b[i] = ops[(size_t)t * 8 + 4 + i]; }
    f4 c4[16] = {};
    f16v c16[8] = {};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k & 3], b[(k >> 2) & 3], c4[k], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) c16[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k & 3], b[(k >> 1) & 3], c16[k], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += c4[k][0];
    for (int k = 0; k < 8; ++k) s += c16[k][0];
    sink[t] = s;
}

// Two waves per SIMD that TAKE TURNS: an 8-wave workgroup, waves 0-3 and 4-7 alternate phases of NB back-to-back MFMAs with one s_barrier
// per phase (the persistent GEMM's ping-pong without its loads).  If the 2 / 3 of the freely interleaved streams is an arbitration effect, this
// form is not subject to it.
template <int NB>
__global__ __launch_bounds__(512) void mfma_pingpong(const h8* __restrict__ ops, float* sink, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = ops[(size_t)t * 8 + i]; b[i] = ops[(size_t)t * 8 + 4 + i]; }
    f16v c16[8] = {};
    for (int it = 0; it < iters; ++it) {
        if ((it & 1) == grp) {
#pragma unroll
            for (int k = 0; k < NB; ++k) c16[k & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k & 3], b[(k >> 1) & 3], c16[k & 7], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += c16[k][0];
    sink[t] = s;
}

static std::string read_file(const std::string& p) {
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return "";
    char buf[128] = {0};
    size_t n = fread(buf, 1, 127, f);
    fclose(f);
    while (n && (buf[n - 1] == '\n' || buf[n - 1] == ' ')) buf[--n] = 0;
    return buf;
}

static std::string find_hwmon(const std::string& pci) {
    DIR* d = opendir("/sys/class/drm");
    if (!d) return "";
    std::string found;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
        const std::string dev = std::string("/sys/class/drm/") + e->d_name + "/device";
        char real[512];
        if (!realpath(dev.c_str(), real)) continue;
        const char* base = strrchr(real, '/');
        if (!base || strcasecmp(base + 1, pci.c_str()) != 0) continue;
        DIR* h = opendir((dev + "/hwmon").c_str());
        if (!h) continue;
        while (dirent* he = readdir(h))
            if (strncmp(he->d_name, "hwmon", 5) == 0) found = dev + "/hwmon/" + he->d_name;
        closedir(h);
    }
    closedir(d);
    return found;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    char pci[32] = {0};
    hipDeviceGetPCIBusId(pci, 32, 0);
    const std::string hw = find_hwmon(pci);
    const std::string pfile = hw.empty() ? "" : (read_file(hw + "/power1_average").empty() ? hw + "/power1_input" : hw + "/power1_average");
    printf("# device %s hwmon %s cap %s uW idle %s uW\n", pci, hw.c_str(), read_file(hw + "/power1_cap").c_str(), read_file(pfile).c_str());
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%-74s %8s %8s %8s %9s %9s\n", "MFMA stream (registers only)", "TF/s", "W mean", "W max", "MHz mean", "pJ/FLOP");
    for (int data = 0; data < 2; ++data)
        for (int kind : {32, 16})
            for (int wcfg : {1, 2, 4, -2, -8, -16}) {      // -2: two waves per SIMD, the second one at s_setprio 3; -8 / -16: ping-pong phases of 8 / 16 MFMAs
                if (wcfg <= -8 && kind != 32) continue;
                const bool pp = wcfg <= -8;
                const int wps = pp ? 2 : (wcfg < 0 ? -wcfg : wcfg), prio_from = (wcfg == -2) ? cus : 0;
                const int blocks = pp ? cus : cus * wps, threads = pp ? 512 : 256;          // 4 waves per block = 1 per SIMD; wps blocks per CU
                const size_t n = (size_t)blocks * threads;
                std::vector<h8> host(n * 8);
                srand(7);
                for (auto& v : host)
                    for (int e = 0; e < 8; ++e) {       // normal(0, 1) by Box-Muller; zeros for data == 0
                        const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
                        v[e] = data ? (_Float16)(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2)) : (_Float16)0.f;
                    }
                h8* dops;
                float* dsink;
                hipMalloc(&dops, n * 8 * sizeof(h8));
                hipMalloc(&dsink, n * sizeof(float));
                hipMemcpy(dops, host.data(), n * 8 * sizeof(h8), hipMemcpyHostToDevice);
                const int iters = 20000;
                auto launch = [&] {
                    if (wcfg == -8) hipLaunchKernelGGL(mfma_pingpong<8>, dim3(blocks), dim3(threads), 0, 0, dops, dsink, 2 * iters);
                    else if (wcfg == -16) hipLaunchKernelGGL(mfma_pingpong<16>, dim3(blocks), dim3(threads), 0, 0, dops, dsink, iters);
                    else if (kind == 16) hipLaunchKernelGGL(mfma_stream<16>, dim3(blocks), dim3(threads), 0, 0, dops, dsink, iters, prio_from);
                    else hipLaunchKernelGGL(mfma_stream<32>, dim3(blocks), dim3(threads), 0, 0, dops, dsink, iters, prio_from);
                };
                launch();
                hipDeviceSynchronize();
                std::atomic<bool> stop{false};
                std::vector<double> pw, ck;
                std::thread sampler([&] {
                    while (!stop) {
                        const std::string p = read_file(pfile), c = read_file(hw + "/freq1_input");
                        if (!p.empty()) pw.push_back(atof(p.c_str()) / 1e6);
                        if (!c.empty()) ck.push_back(atof(c.c_str()) / 1e6);
                        usleep(50000);
                    }
                });
                const auto t0 = std::chrono::steady_clock::now();
                long launches = 0;
                double dt = 0;
                while (dt < seconds) {
                    launch();
                    hipDeviceSynchronize();
                    ++launches;
                    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                }
                stop = true;
                sampler.join();
                const double flop_per_launch = pp ? (double)n / 64 * iters * 8 * 32768.0      // NB = 8: 2 * iters / 2 phases of 8; NB = 16: iters / 2 phases of 16
                                                 : (double)n / 64 * iters * (kind == 16 ? 16 : 8) * 32768.0 / (kind == 16 ? 2 : 1);
                const double tf = flop_per_launch * launches / dt / 1e12;
                double pm = 0, px = 0, cm = 0;
                const size_t skip = pw.size() / 4;
                for (size_t i = skip; i < pw.size(); ++i) { pm += pw[i]; px = pw[i] > px ? pw[i] : px; }
                for (size_t i = skip; i < ck.size(); ++i) cm += ck[i];
                pm /= (pw.size() - skip ? pw.size() - skip : 1);
                cm /= (ck.size() - skip ? ck.size() - skip : 1);
                char name[128];
                snprintf(name, 128, "v_mfma_f32_%s_f16, %s, %d wave(s) per SIMD%s", kind == 16 ? "16x16x32" : "32x32x16", data ? "normal(0,1)" : "zeros", wps, prio_from ? ", 2nd at prio 3" : (wcfg == -8 ? ", taking turns: 8 MFMAs per phase" : (wcfg == -16 ? ", taking turns: 16 MFMAs per phase" : "")));
                printf("%-74s %8.0f %8.0f %8.0f %9.0f %9.3f\n", name, tf, pm, px, cm, pm / tf);       // W / (TFLOP/s) = pJ per FLOP
                fflush(stdout);
                hipFree(dops);
                hipFree(dsink);
            }
    return 0;
}
