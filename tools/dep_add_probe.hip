// dep_add_probe.hip -- latency of a DEPENDENT chain of fp64 adds on one wave (the floor of the
// order-faithful sums of the revised simplex: s = s + p_i, 4096 of them per output).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/dep_add_probe.hip -o tools/_bin/dep_add_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k_chain(double* out, const double* in, int n, unsigned long long* ticks) {
    double s = in[threadIdx.x];
    const double p = in[64 + threadIdx.x];
    double q = in[128 + threadIdx.x];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(s) : "v"(p));
            } else if (MODE == 1) {  // independent multiply between the dependent adds
                double t;
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(q), "v"(p));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(s) : "v"(t));
            } else {  // two independent chains interleaved
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(s) : "v"(p));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(q) : "v"(p));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = s + q;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
    double *in, *out;
    unsigned long long* ticks;
    hipMalloc(&in, 4096);
    hipMalloc(&out, 4096);
    hipMalloc(&ticks, 8);
    hipMemset(in, 0, 4096);
    const int n = 1 << 16;
    for (int mode = 0; mode < 3; ++mode) {
        for (int lanes : {16, 64}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(lanes), 0, 0, out, in, n, ticks);
                if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(lanes), 0, 0, out, in, n, ticks);
                if (mode == 2) hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(lanes), 0, 0, out, in, n, ticks);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t = 0;
            hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            std::printf("{\"mode\": \"%s\", \"lanes\": %d, \"steps\": %d, \"ns_per_step_events\": %.2f, "
                        "\"memtime_ticks_per_step\": %.3f}\n",
                        mode == 0 ? "dependent v_add_f64" : mode == 1 ? "mul + dependent add"
                                                                     : "two interleaved chains",
                        lanes, n, 1e6 * ms / n, (double)t / n);
        }
    }
    return 0;
}
