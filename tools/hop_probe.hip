// hop_probe.hip -- what does one cross-stream dependency cost on this device?  A strict ping-pong
// of short kernels between two streams (kernel on A -> B may start -> kernel on B -> A may start
// ...), with the dependency carried by (1) hipEventRecord / hipStreamWaitEvent, (2)
// hipStreamWriteValue32 / hipStreamWaitValue32 on device memory, (3) the same on signal memory
// (hipMallocSignalMemory), (4) a flag stored by the kernel itself + hipStreamWaitValue32; and the
// same kernels back to back on ONE stream for reference.  Prints JSON lines (us per kernel).
//   hipcc -O3 --offload-arch=gfx950 tools/hop_probe.hip -o tools/_bin/hop_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));          \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

__global__ void k_spin(unsigned long long ticks, unsigned* flag, unsigned value) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    }
    if (flag && threadIdx.x == 0 && blockIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 200;        // kernels per stream
    const unsigned long long ticks = argc > 2 ? std::atoi(argv[2]) : 500;  // 5 us
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    std::printf("{\"can_use_stream_wait_value\": %d}\n", can);
    hipStream_t A, B;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };

    {  // reference: one stream
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(A));
            auto t0 = now();
            for (int i = 0; i < 2 * n; ++i) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, A, ticks, nullptr, 0u);
            CK(hipStreamSynchronize(A));
            if (rep) std::printf("{\"mode\": \"one stream\", \"us_per_kernel\": %.2f, \"kernel_us\": %.2f}\n",
                                 us(t0, now()) / (2 * n), ticks * 0.01);
        }
    }
    {  // (1) events
        hipEvent_t ea[2], eb[2];
        for (int k = 0; k < 2; ++k) {
            CK(hipEventCreateWithFlags(&ea[k], hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&eb[k], hipEventDisableTiming));
        }
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = now();
            for (int i = 0; i < n; ++i) {
                if (i) CK(hipStreamWaitEvent(A, eb[(i - 1) & 1], 0));
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, A, ticks, nullptr, 0u);
                CK(hipEventRecord(ea[i & 1], A));
                CK(hipStreamWaitEvent(B, ea[i & 1], 0));
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, B, ticks, nullptr, 0u);
                CK(hipEventRecord(eb[i & 1], B));
            }
            CK(hipDeviceSynchronize());
            if (rep) std::printf("{\"mode\": \"events\", \"us_per_kernel\": %.2f}\n", us(t0, now()) / (2 * n));
        }
    }
    for (int kind = 0; kind < 3 && can; ++kind) {
        // kind 0: write/wait value on device memory, 1: on signal memory, 2: kernel-stored flag
        unsigned* f = nullptr;  // f[0]: A's counter, f[16]: B's counter
        if (kind == 1) {
            if (hipExtMallocWithFlags((void**)&f, 8, hipMallocSignalMemory) != hipSuccess) {
                std::printf("{\"mode\": \"signal memory\", \"error\": \"allocation refused\"}\n");
                (void)hipGetLastError();
                continue;
            }
        } else {
            CK(hipMalloc(&f, 256));
        }
        unsigned* fa = f;
        unsigned* fb = (kind == 1) ? f : f + 16;  // signal memory is ONE 8-byte signal
        if (kind == 1) {
            // one signal only: A's and B's counters interleave on it (odd / even values)
            CK(hipMemset(f, 0, 8));
        } else {
            CK(hipMemset(f, 0, 256));
        }
        bool ok = true;
        unsigned va = 0, vb = 0, v1 = 0;
        for (int rep = 0; rep < 2 && ok; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = now();
            for (int i = 0; i < n && ok; ++i) {
                if (kind == 1) {
                    if (i) ok = ok && hipStreamWaitValue32(A, f, v1, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
                    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, A, ticks, nullptr, 0u);
                    ok = ok && hipStreamWriteValue32(A, f, ++v1, 0) == hipSuccess;
                    ok = ok && hipStreamWaitValue32(B, f, v1, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
                    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, B, ticks, nullptr, 0u);
                    ok = ok && hipStreamWriteValue32(B, f, ++v1, 0) == hipSuccess;
                    continue;
                }
                if (i) ok = ok && hipStreamWaitValue32(A, fb, vb, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
                ++va;
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, A, ticks, kind == 2 ? fa : nullptr, va);
                if (kind == 0) ok = ok && hipStreamWriteValue32(A, fa, va, 0) == hipSuccess;
                ok = ok && hipStreamWaitValue32(B, fa, va, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
                ++vb;
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, B, ticks, kind == 2 ? fb : nullptr, vb);
                if (kind == 0) ok = ok && hipStreamWriteValue32(B, fb, vb, 0) == hipSuccess;
            }
            if (!ok) {
                std::printf("{\"mode\": \"value kind %d\", \"error\": \"%s\"}\n", kind,
                            hipGetErrorString(hipGetLastError()));
                // release anything queued behind a wait: store large values from the host side
                const unsigned big[17] = {0x7fffffffu, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                          0x7fffffffu};
                (void)hipMemcpy(f, big, kind == 1 ? 4 : sizeof(big), hipMemcpyHostToDevice);
                (void)hipDeviceSynchronize();
                break;
            }
            CK(hipDeviceSynchronize());
            if (rep) std::printf("{\"mode\": \"%s\", \"us_per_kernel\": %.2f}\n",
                                 kind == 0 ? "write/wait value, device memory"
                                           : kind == 1 ? "write/wait value, signal memory"
                                                       : "kernel-stored flag + wait value",
                                 us(t0, now()) / (2 * n));
        }
        std::fflush(stdout);
    }
    return 0;
}
