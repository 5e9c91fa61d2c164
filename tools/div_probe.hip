// div_probe.hip -- is the device's fp64 division correctly rounded?  Compares `a / b` as hipcc
// compiles it for gfx950 with lpr::ieee_div (engine_common.hpp: exact-residual repair) on the host's
// own IEEE division: (1) the pair tools/fuzz_side_gpu.py found, (2) "decimal" operands k1 * 0.1^i
// perturbed by a few ulps -- the kind a tableau of small rationals is made of --, (3) random
// mantissas.  Prints JSON lines.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Ilpr_381_group_v22_amd/csrc -Iinclude \
//         tools/div_probe.hip -o tools/_bin/div_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "engine_common.hpp"

__global__ void k_div(const double* a, const double* b, double* plain, double* fixed, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    plain[i] = a[i] / b[i];
    fixed[i] = lpr::ieee_div(a[i], b[i]);
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}
static double perturb(double x, int ulps) {
    int64_t bits; std::memcpy(&bits, &x, 8); bits += ulps; std::memcpy(&x, &bits, 8); return x;
}

static int run(const char* name, const std::vector<double>& a, const std::vector<double>& b) {
    const int n = (int)a.size();
    double *da, *db, *dp, *df;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dp, n * 8); hipMalloc(&df, n * 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_div, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dp, df, n);
    std::vector<double> p(n), f(n);
    hipMemcpy(p.data(), dp, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(f.data(), df, n * 8, hipMemcpyDeviceToHost);
    long bad_plain = 0, bad_fixed = 0; int shown = 0;
    for (int i = 0; i < n; ++i) {
        const volatile double va = a[i], vb = b[i];
        const double h = va / vb;  // the host's IEEE division
        if (std::memcmp(&h, &p[i], 8) != 0) {
            ++bad_plain;
            if (shown++ < 3)
                std::printf("{\"case\": \"%s\", \"a\": \"%a\", \"b\": \"%a\", \"host\": \"%a\", \"device\": \"%a\", "
                            "\"ieee_div\": \"%a\"}\n", name, a[i], b[i], h, p[i], f[i]);
        }
        if (std::memcmp(&h, &f[i], 8) != 0) ++bad_fixed;
    }
    std::printf("{\"case\": \"%s\", \"pairs\": %d, \"device_division_differs_from_ieee\": %ld, "
                "\"ieee_div_differs_from_ieee\": %ld}\n", name, n, bad_plain, bad_fixed);
    hipFree(da); hipFree(db); hipFree(dp); hipFree(df);
    return bad_fixed ? 1 : 0;
}

int main() {
    int rc = 0;
    rc |= run("found by the fuzzer", {-0x1.6666666666663p+0}, {-0x1.ffffffffffffbp+1});
    {
        std::vector<double> a, b;
        const int N = 1 << 24;
        for (int i = 0; i < N; ++i) {
            const double x = (double)(1 + rnd() % 9999) * 0.01, y = (double)(1 + rnd() % 999) * 0.1;
            a.push_back(perturb(x, (int)(rnd() % 17) - 8));
            b.push_back(perturb(y, (int)(rnd() % 17) - 8));
        }
        rc |= run("small decimals, a few ulps off", a, b);
    }
    {
        std::vector<double> a, b;
        const int N = 1 << 24;
        for (int i = 0; i < N; ++i) {
            uint64_t ma = (rnd() >> 12) | 0x3ff0000000000000ull, mb = (rnd() >> 12) | 0x3ff0000000000000ull;
            double x, y; std::memcpy(&x, &ma, 8); std::memcpy(&y, &mb, 8);
            a.push_back((rnd() & 1) ? x : -x);
            b.push_back(y * (double)(1 + rnd() % 7));
        }
        rc |= run("random mantissas", a, b);
    }
    return rc;
}
