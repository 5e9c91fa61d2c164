// anyorder_probe.hip -- do two kernels queued on ONE stream overlap when the second is launched
// with hipExtAnyOrderLaunch (its AQL packet carries no barrier bit)?  And what does a step of the
// pattern [A: ordered] [B: any order] [A: ordered] [B: any order] ... cost, against the same pairs
// on two streams joined by events (the two-stream form of the overlapped pivot loop)?
// Every kernel stamps s_memrealtime (100 MHz) at its start and end.  Prints JSON lines.
//   hipcc -O3 --offload-arch=gfx950 tools/anyorder_probe.hip -o tools/_bin/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));          \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

__global__ void k_spin(unsigned long long ticks, unsigned long long* stamps, int slot) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[2 * slot] = t0;
        stamps[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 50;                        // pairs
    const unsigned long long ticks = argc > 2 ? std::atoi(argv[2]) : 10000;  // 100 us
    hipStream_t S, H;
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&H, hipStreamNonBlocking));
    unsigned long long* d = nullptr;
    CK(hipMalloc(&d, sizeof(unsigned long long) * 4 * n));
    std::vector<unsigned long long> h(4 * n);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    auto report = [&](const char* mode, double wall_us) {
        // overlap of the two kernels of a pair, and the gap from the end of a pair to the start of
        // the next (both averaged over the pairs)
        double ov = 0, gap = 0;
        for (int i = 0; i < n; ++i) {
            const double a0 = h[4 * i] * 0.01, a1 = h[4 * i + 1] * 0.01;
            const double b0 = h[4 * i + 2] * 0.01, b1 = h[4 * i + 3] * 0.01;
            const double lo = a0 > b0 ? a0 : b0, hi = a1 < b1 ? a1 : b1;
            ov += hi > lo ? hi - lo : 0.0;
            if (i + 1 < n) {
                const double end = a1 > b1 ? a1 : b1;
                const double n0 = h[4 * i + 4] * 0.01, n1 = h[4 * i + 6] * 0.01;
                gap += (n0 < n1 ? n1 : n0) - end;   // when BOTH kernels of the next pair have started
            }
        }
        std::printf("{\"mode\": \"%s\", \"pairs\": %d, \"kernel_us\": %.1f, \"us_per_pair\": %.2f, "
                    "\"overlap_us\": %.2f, \"gap_to_next_pair_us\": %.2f}\n",
                    mode, n, ticks * 0.01, wall_us / n, ov / n, gap / (n - 1));
    };
    for (int rep = 0; rep < 2; ++rep) {  // (1) one stream, everything ordered
        CK(hipStreamSynchronize(S));
        auto t0 = now();
        for (int i = 0; i < n; ++i) {
            hipExtLaunchKernelGGL(k_spin, dim3(8), dim3(64), 0, S, nullptr, nullptr, 0, ticks, d, 2 * i);
            hipExtLaunchKernelGGL(k_spin, dim3(8), dim3(64), 0, S, nullptr, nullptr, 0, ticks, d, 2 * i + 1);
        }
        CK(hipStreamSynchronize(S));
        const double w = us(t0, now());
        CK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * 4 * n, hipMemcpyDeviceToHost));
        if (rep) report("one stream, ordered", w);
    }
    for (int rep = 0; rep < 2; ++rep) {  // (2) one stream, the second kernel of a pair in any order
        CK(hipStreamSynchronize(S));
        auto t0 = now();
        for (int i = 0; i < n; ++i) {
            hipExtLaunchKernelGGL(k_spin, dim3(8), dim3(64), 0, S, nullptr, nullptr, 0, ticks, d, 2 * i);
            hipExtLaunchKernelGGL(k_spin, dim3(8), dim3(64), 0, S, nullptr, nullptr,
                                  hipExtAnyOrderLaunch, ticks, d, 2 * i + 1);
        }
        CK(hipStreamSynchronize(S));
        const double w = us(t0, now());
        CK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * 4 * n, hipMemcpyDeviceToHost));
        if (rep) report("one stream, second kernel hipExtAnyOrderLaunch", w);
    }
    {  // (3) two streams, each kernel waits for BOTH kernels of the pair before (completion events)
        hipEvent_t ea[2], eb[2];
        for (int k = 0; k < 2; ++k) {
            CK(hipEventCreateWithFlags(&ea[k], hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&eb[k], hipEventDisableTiming));
        }
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(S));
            CK(hipStreamSynchronize(H));
            auto t0 = now();
            for (int i = 0; i < n; ++i) {
                const int cur = i & 1, prev = cur ^ 1;
                if (i > 0) {
                    CK(hipStreamWaitEvent(H, ea[prev], 0));
                    CK(hipStreamWaitEvent(S, eb[prev], 0));
                }
                hipExtLaunchKernelGGL(k_spin, dim3(8), dim3(64), 0, S, nullptr, ea[cur], 0, ticks, d, 2 * i);
                hipExtLaunchKernelGGL(k_spin, dim3(8), dim3(64), 0, H, nullptr, eb[cur], 0, ticks, d, 2 * i + 1);
            }
            CK(hipStreamSynchronize(S));
            CK(hipStreamSynchronize(H));
            const double w = us(t0, now());
            CK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * 4 * n, hipMemcpyDeviceToHost));
            if (rep) report("two streams, completion events both ways", w);
        }
    }
    return 0;
}
