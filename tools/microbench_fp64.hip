// Micro-benchmark (not part of the product): FP64 VALU issue rates on gfx950 and the
// accuracy of v_rcp_f64, used to budget the K1 sweep kernel.  The long runs at the end
// (~20 ms each, like a K1 launch) also read the shader clock counter against the 100 MHz
// constant counter, i.e. the clock the chip actually sustains under that instruction mix.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench_fp64.hip -o tools/microbench_fp64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096;

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double seed) {
    double a0 = seed + threadIdx.x * 1e-9, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3;
    double a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;
    const double c = 1.0000001, d = 1e-9;
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
#define STEP(x)                                                   \
        if (OP == 0) x = fma(x, c, d);                            \
        else if (OP == 1) x = x + d;                              \
        else if (OP == 2) x = x * c;                              \
        else if (OP == 3) x = __builtin_amdgcn_rcp(x);            \
        else if (OP == 4) x = (double)__builtin_amdgcn_rcpf((float)x); \
        else if (OP == 5) x = sqrt(x);                            \
        else if (OP == 6) x = __builtin_amdgcn_rsq(x);
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// Long-running streams: OP 0 = all FMA, 1 = the K1 pair-loop mix (per 15 ops: 9 FMA, 4 mul, 2 add).
// clk[0..1] = shader-clock cycles and 100 MHz ticks spent by block 0.
template <int OP>
__global__ __launch_bounds__(256) void long_k(double* out, double seed, int iters, long long* clk) {
    double a0 = seed + threadIdx.x * 1e-9, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3, a4 = a0 + 0.4;
    const double c = 1.0000001, d = 1e-9;
    const long long t0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = fma(a2, c, d); a3 = fma(a3, c, d); a4 = fma(a4, c, d);
            }
        } else {
            a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = fma(a2, c, d); a3 = a3 * c;        a4 = a4 + d;
            a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = fma(a2, c, d); a3 = a3 * c;        a4 = a4 * c;
            a0 = fma(a0, c, d); a1 = fma(a1, c, d); a2 = fma(a2, c, d); a3 = a3 * c;        a4 = a4 + d;
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = w1 - w0;
    }
}

template <int OP>
void run_long(double* out, int blocks, const char* name) {
    long long* clk;
    hipMalloc(&clk, 16);
    const int iters = 1 << 18;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    long_k<OP><<<blocks, 256>>>(out, 1.5, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    long_k<OP><<<blocks, 256>>>(out, 1.5, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ops = (double)blocks * 256 * iters * 15;
    printf("%-28s %7.2f ms  %6.2f T lane-ops/s; block 0: %lld shader cycles in %.3f ms -> %.0f MHz sustained\n", name, ms,
           ops / (ms * 1e-3) / 1e12, h[0], h[1] / 1e5, h[0] / (h[1] / 1e2));
    hipFree(clk);
}

__global__ void rcp_err(const double* q, double* r, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = __builtin_amdgcn_rcp(q[i]);
}

template <int OP>
double run(double* out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const int blocks = cus * 8;   // 8 blocks x 4 waves = 8 waves per SIMD
    double* out; CHECK(hipMalloc(&out, blocks * 256 * sizeof(double)));
    const char* names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "cvt+v_rcp_f32+cvt", "sqrt(f64) ocml", "v_rsq_f64"};
    double ms[7] = {run<0>(out, blocks), run<1>(out, blocks), run<2>(out, blocks), run<3>(out, blocks),
                    run<4>(out, blocks), run<5>(out, blocks), run<6>(out, blocks)};
    printf("device %s, %d CUs, clock %.0f MHz\n", p.gcnArchName, cus, p.clockRate / 1e3);
    for (int i = 0; i < 7; ++i) {
        // wave-instructions per SIMD: 8 waves/SIMD * ITER * 8 per launch
        const double winst = 8.0 * ITER * 8;
        const double ns_per = ms[i] * 1e6 / winst;
        printf("%-20s %8.3f ms  -> %6.2f ns per wave-instruction per SIMD (= %5.2f cycles @2.4GHz), %7.2f T lane-ops/s\n",
               names[i], ms[i], ns_per, ns_per * 2.4, (double)blocks * 256 * ITER * 8 / (ms[i] * 1e-3) / 1e12);
    }
    run_long<0>(out, blocks, "long run, all v_fma_f64");
    run_long<1>(out, blocks, "long run, K1 mix 9:4:2");
    // accuracy of v_rcp_f64 on q in [1, 1000]
    const int n = 1 << 20;
    std::vector<double> hq(n), hr(n);
    for (int i = 0; i < n; ++i) hq[i] = 1.0 + 999.0 * (double)rand() / RAND_MAX * ((i & 1) ? 1.0 : 0.001);
    double *dq, *dr; CHECK(hipMalloc(&dq, n * 8)); CHECK(hipMalloc(&dr, n * 8));
    CHECK(hipMemcpy(dq, hq.data(), n * 8, hipMemcpyHostToDevice));
    rcp_err<<<n / 256, 256>>>(dq, dr, n);
    CHECK(hipMemcpy(hr.data(), dr, n * 8, hipMemcpyDeviceToHost));
    double worst = 0, mean = 0;
    for (int i = 0; i < n; ++i) { double e = fabs(hr[i] * hq[i] - 1.0); worst = fmax(worst, e); mean += e; }
    printf("v_rcp_f64 relative error: max %.3e (2^%.1f), mean %.3e\n", worst, log2(worst), mean / n);
    return 0;
}
