// Micro-benchmark (not part of the product): calibrates rocprofv3's FETCH_SIZE for the access
// pattern of the K1 sweep kernel — wave-uniform s_load_dwordx16 streaming through the scalar
// cache.  Every wave reads its own contiguous slice of a 1 GiB buffer exactly once (1 GiB: four
// times the Infinity Cache, so the reads reach the memory-side counters whatever the cache does);
// a second kernel reads the same bytes with 8-byte-per-lane vector loads for comparison.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_sload.hip -o tools/microbench_sload
//   rocprofv3 --pmc FETCH_SIZE -d out -o sload -- tools/microbench_sload
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void scalar_stream(const double* __restrict__ in, long long per_wave, double* out) {
    const double* __restrict__ p = in + (long long)blockIdx.x * per_wave;   // uniform: scalar loads
    double acc = 0.0;
    for (long long i = 0; i < per_wave; i += 16) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += p[i + k];
        acc += s;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void vector_stream(const double* __restrict__ in, long long n, double* out) {
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += in[i];
    out[(long long)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const long long n = 1LL << 27;            // 1 GiB of doubles
    const int waves = 16384;
    double *in, *out;
    CHECK(hipMalloc(&in, n * sizeof(double)));
    CHECK(hipMalloc(&out, 1 << 22));
    CHECK(hipMemset(in, 0, n * sizeof(double)));
    for (int rep = 0; rep < 3; ++rep) {
        scalar_stream<<<waves, 64>>>(in, n / waves, out);
        vector_stream<<<2048, 256>>>(in, n, out);
    }
    CHECK(hipDeviceSynchronize());
    printf("each launch reads %lld bytes once\n", n * (long long)sizeof(double));
    return 0;
}
