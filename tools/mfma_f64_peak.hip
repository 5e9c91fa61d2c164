// mfma_f64_peak.hip -- measures the back-to-back issue rate of v_mfma_f64_16x16x4_f64 on the
// device (the local guide documents the instruction but no fp64 MFMA peak, so the
// utilisation of k_rev_gemm is quoted against THIS measured number).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_f64_peak && /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_peak(double* out, int iters, double a0, double b0) {
    double4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;  // random-ish, not zero
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static double run(int blocks, int iters) {
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_peak<NACC>, dim3(blocks), dim3(256), 0, 0, out, 100, 0.37, 1.91);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_peak<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.37, 1.91);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4.0 * blocks;  // 4 waves/block
    hipFree(out);
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double best = 0;
    for (int wpc : {1, 2, 4, 8}) {  // workgroups per CU (4 .. 32 waves per CU)
        const double t4 = run<4>(cus * wpc, 20000);
        const double t8 = run<8>(cus * wpc, 10000);
        const double t16 = run<16>(cus * wpc, 5000);
        printf("{\"cus\": %d, \"wg_per_cu\": %d, \"tflops_4acc\": %.2f, \"tflops_8acc\": %.2f, "
               "\"tflops_16acc\": %.2f}\n", cus, wpc, t4, t8, t16);
        if (t4 > best) best = t4;
        if (t8 > best) best = t8;
        if (t16 > best) best = t16;
    }
    printf("{\"mfma_f64_16x16x4_peak_tflops\": %.2f}\n", best);
    return 0;
}
