// Host round trip of a tiny kernel: hipStreamSynchronize against spinning on a value the kernel itself
// writes into page-locked host memory (developer aid).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_sync.hip -o tools/microbench_sync && tools/microbench_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

__global__ void tiny(volatile uint64_t* host_flag, uint64_t seq, double* dev) {
    if (threadIdx.x == 0) {
        dev[0] = (double)seq;                       // some device work
        __threadfence_system();
        *host_flag = seq;                           // the result lands in host memory
    }
}

int main() {
    uint64_t* flag;
    double* dev;
    hipHostMalloc(&flag, 64, hipHostMallocDefault);
    hipMalloc(&dev, 64);
    uint64_t* dflag;
    hipHostGetDevicePointer((void**)&dflag, flag, 0);
    hipStream_t st;
    hipStreamCreate(&st);
    *flag = 0;
    const int reps = 3000;
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<double> us;
        for (int i = 1; i <= reps; ++i) {
            const uint64_t seq = (uint64_t)mode * 1000000 + i;
            auto t0 = std::chrono::steady_clock::now();
            tiny<<<1, 64, 0, st>>>(dflag, seq, dev);
            if (mode == 0) {
                hipStreamSynchronize(st);
            } else {
                while (*(volatile uint64_t*)flag != seq) {
                }
            }
            auto t1 = std::chrono::steady_clock::now();
            us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            if (mode == 1) hipStreamSynchronize(st);        // (outside the timed region: keep the queue drained)
        }
        std::sort(us.begin(), us.end());
        printf("%s: median %.2f us, p10 %.2f, p90 %.2f\n", mode == 0 ? "launch + hipStreamSynchronize" : "launch + spin on the kernel's own host write",
               us[reps / 2], us[reps / 10], us[reps * 9 / 10]);
    }
    return 0;
}
