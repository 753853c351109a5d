// Micro-benchmark (not part of the product): can v_mfma_f64_16x16x4 run beside a saturated
// FP64 VALU stream for free?  (Would the sweep kernel gain from forming q = (x - x0)^2 + 1 as a
// rank-3 product on the matrix core?)  Per loop trip: VALU_PER_TRIP independent FMAs and
// MFMA_PER_TRIP matrix instructions; long runs, throughput and sustained clock reported.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench_mfma_mix.hip -o tools/microbench_mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double dvec4 __attribute__((ext_vector_type(4)));

template <int NMFMA>
__global__ __launch_bounds__(256) void mix(double* out, double seed, int iters, long long* clk) {
    double a0 = seed + threadIdx.x * 1e-9, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3, a4 = a0 + 0.4, a5 = a0 + 0.5,
           a6 = a0 + 0.6, a7 = a0 + 0.7;
    const double c = 1.0000001, d = 1e-9;
    dvec4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double ma = seed, mb = 1e-12;
    const long long t0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (NMFMA >= 1) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc0, 0, 0, 0);
        a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = fma(a2, c, d); a3 = fma(a3, c, d);
        a4 = fma(a4, c, d); a5 = fma(a5, c, d); a6 = fma(a6, c, d); a7 = fma(a7, c, d);
        a0 = a0 * c; a1 = a1 + d; a2 = a2 * c; a3 = a3 + d;
        if (NMFMA >= 2) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mb, ma, acc1, 0, 0, 0);
        a4 = fma(a4, c, d); a5 = fma(a5, c, d); a6 = fma(a6, c, d); a7 = fma(a7, c, d);
        a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = a2 * c; a3 = a3 + d;
        a4 = a4 * c; a5 = a5 + d;
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc0[3] + acc1[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = w1 - w0;
    }
}

template <int NMFMA>
void run(double* out, long long* clk, int blocks, const char* name) {
    const int iters = 1 << 17;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mix<NMFMA><<<blocks, 256>>>(out, 1.5, iters, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mix<NMFMA><<<blocks, 256>>>(out, 1.5, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double valu = (double)blocks * 256 * iters * 22;
    printf("%-34s %7.2f ms  VALU %6.2f T lane-ops/s, %d MFMA per 22 VALU; %.0f MHz sustained\n", name, ms,
           valu / (ms * 1e-3) / 1e12, NMFMA, h[0] / (h[1] / 1e2));
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 3;   // 3 blocks x 4 waves per CU = 3 waves per SIMD, like the sweep kernel
    double* out; (void)hipMalloc(&out, blocks * 256 * sizeof(double));
    long long* clk; (void)hipMalloc(&clk, 16);
    run<0>(out, clk, blocks, "K1-like VALU mix alone");
    run<1>(out, clk, blocks, "+ 1 v_mfma_f64_16x16x4 per trip");
    run<2>(out, clk, blocks, "+ 2 v_mfma_f64_16x16x4 per trip");
    return 0;
}
