// sweep_bench.hip -- the out-of-place 16-pivot sweep (k_ov2_sweep) alone on a 4097 x 12289 tableau,
// with parts of it switched off at compile time (LPR_OV_DIAG, see overlap_kernels.hip), to find out
// what bounds it.  Not part of the library; built by tools/sweep_bench.sh:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DLPR_OV_KERNELS_ONLY \
//         -DLPR_OV_DIAG=<bits> -I lpr_381_group_v22_amd/csrc -I include tools/sweep_bench.hip -o ...
// usage: sweep_bench [reps] [wgs_per_cu] [leave_xcc (-1: none)] [tile code 4|8|16|0x24|0x28]
// prints one JSON line: average / min launch time by HIP events, GB/s of 2*8*R*ld.
#include "overlap_kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void k_fill(double* p, size_t n, double scale, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z ^= z >> 29;
        z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 32;
        p[i] = scale * (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
}

// reference points: a straight copy with the same loads in flight per lane (8 x 16 B), and the
// same with every workgroup on its own contiguous 32 KB pieces taken from a counter
__global__ __launch_bounds__(256) void k_copy_linear(const lpr::ov_v2d* __restrict__ src,
                                                     lpr::ov_v2d* __restrict__ dst, size_t n2,
                                                     unsigned* q) {
    __shared__ unsigned s_p;
    const size_t piece = 256 * 8;  // double2 per workgroup piece = 32 KB
    const size_t npieces = (n2 + piece - 1) / piece;
    for (;;) {
        if (threadIdx.x == 0) s_p = atomicAdd(q, 1u);
        __syncthreads();
        const size_t pc = s_p;
        __syncthreads();
        if (pc >= npieces) return;
        const size_t base = pc * piece + threadIdx.x;
        lpr::ov_v2d x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (base + k * 256 < n2) x[k] = __builtin_nontemporal_load(&src[base + k * 256]);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (base + k * 256 < n2) __builtin_nontemporal_store(x[k], &dst[base + k * 256]);
    }
}

int main(int argc, char** argv) {
    using namespace lpr;
    const int reps = argc > 1 ? std::atoi(argv[1]) : 40;
    const int wgs_per_cu = argc > 2 ? std::atoi(argv[2]) : 4;
    const int leave = argc > 3 ? std::atoi(argv[3]) : -1;
    const int tile = argc > 4 ? (int)std::strtol(argv[4], nullptr, 0) : 8;
    const int R = argc > 5 ? std::atoi(argv[5]) : 4097;
    const int C = argc > 6 ? std::atoi(argv[6]) : 12289;
    const int mode = argc > 7 ? std::atoi(argv[7]) : 0;  // 1: linear copy kernel, 2: hipMemcpyDtoD
    const int ld = (C + 15) / 16 * 16, Rp = (R + 15) / 16 * 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;

    OvBuffers b{};
    const size_t tn = (size_t)R * ld;
    CK(hipMalloc(&b.Tb[0], tn * 8));
    CK(hipMalloc(&b.Tb[1], tn * 8));
    CK(hipMalloc(&b.prow, (size_t)2 * kOvMax * ld * 8));
    CK(hipMalloc(&b.fcol, (size_t)2 * kOvMax * Rp * 8));
    CK(hipMalloc(&b.hx, 16));
    CK(hipMalloc(&b.tileq, 16));
    CK(hipMalloc(&b.sflag, 16));
    CK(hipMalloc(&b.ctl, 2 * sizeof(OvCtl)));
    CK(hipMalloc(&b.bar, 16));
    CK(hipMemset(b.hx, 0, 16));
    CK(hipMemset(b.tileq, 0, 16));
    CK(hipMemset(b.sflag, 0, 16));
    CK(hipMemset(b.bar, 0, 16));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, b.Tb[0], tn, 1.0, 1u);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, b.Tb[1], tn, 1.0, 2u);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, b.prow, (size_t)2 * kOvMax * ld, 1.0, 3u);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, b.fcol, (size_t)2 * kOvMax * Rp, 1e-3, 4u);
    OvCtl h[2] = {};
    for (int k = 0; k < 2; ++k) {
        h[k].status = kRunning;
        h[k].pending = kRunning;
        h[k].kdone = kOvMax;
        h[k].slot = 0;
        h[k].cur = k;  // launch parity k reads buffer k (the sweep flips `cur` itself)
        h[k].sweep = k;
        for (int s = 0; s < kOvMax; ++s) h[k].r[s] = 1 + 251 * s;
        h[k].head_xcc = leave;
    }
    CK(hipMemcpy(b.ctl, h, sizeof(h), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());

    const int nct = (ld / 2 + kOvNT - 1) / kOvNT, nrt = (R + kOvTileRows - 1) / kOvTileRows;
    const int ntiles = nct * nrt;
    const int cap = wgs_per_cu * cus;
    const dim3 grid(wgs_per_cu > 0 && ntiles > cap ? cap : ntiles), blk(kOvNT);
    const int avoid = leave >= 0 ? 2 : 0;
    std::vector<hipEvent_t> ev(2 * (reps + 4));
    for (auto& e : ev) CK(hipEventCreate(&e));
    hipStream_t S;
    CK(hipStreamCreate(&S));
    auto launch = [&](int lp) {
        if (mode == 1) {
            hipMemsetAsync(b.bar, 0, 4, S);
            hipLaunchKernelGGL(k_copy_linear, grid, blk, 0, S,
                               reinterpret_cast<const ov_v2d*>(b.Tb[lp]),
                               reinterpret_cast<ov_v2d*>(b.Tb[lp ^ 1]), tn / 2, b.bar);
            return;
        }
        if (mode == 2) {
            hipMemcpyAsync(b.Tb[lp ^ 1], b.Tb[lp], tn * 8, hipMemcpyDeviceToDevice, S);
            return;
        }
#define SW(TR, DB)                                                                               \
    hipLaunchKernelGGL((k_ov2_sweep<TR, DB>), grid, blk, 0, S, b, b.fcol, b.prow, ld, R, Rp, lp, \
                       avoid, 1, -1)
        switch (tile) {
            case 0x04: SW(4, false); break;
            case 0x10: SW(16, false); break;
            case 0x24: SW(4, true); break;
            case 0x28: SW(8, true); break;
            default: SW(8, false); break;
        }
#undef SW
    };
    for (int k = 0; k < 4; ++k) launch(k & 1);
    CK(hipStreamSynchronize(S));
    for (int k = 0; k < reps; ++k) {
        CK(hipEventRecord(ev[2 * k], S));
        launch(k & 1);
        CK(hipEventRecord(ev[2 * k + 1], S));
    }
    CK(hipStreamSynchronize(S));
    CK(hipGetLastError());
    double sum = 0, mn = 1e9;
    for (int k = 0; k < reps; ++k) {
        float ms = 0;
        CK(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
        sum += ms;
        if (ms < mn) mn = ms;
    }
    const double bytes = 2.0 * 8.0 * R * (double)ld;
    std::printf("{\"mode\": %d, \"diag\": %d, \"tile\": \"0x%x\", \"wgs_per_cu\": %d, \"grid\": %d, \"leave_xcc\": %d, "
                "\"R\": %d, \"C\": %d, \"avg_us\": %.2f, \"min_us\": %.2f, \"avg_gbps\": %.1f, "
                "\"min_gbps\": %.1f}\n",
                mode, (int)LPR_OV_DIAG, tile, wgs_per_cu, (int)grid.x, leave, R, C, 1e3 * sum / reps,
                1e3 * mn, bytes / (sum / reps) * 1e-6, bytes / mn * 1e-6);
    return 0;
}
