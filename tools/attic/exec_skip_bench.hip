// Micro-benchmark (developer probe): does a gfx950 SIMD skip the 16-lane passes of a wave64 vector instruction whose lanes are all
// masked off?  The same chain of dependent fp32 / fp64 FMAs with 64, 32, 16 and 1 active lanes.
//   hipcc --offload-arch=gfx950 -O3 -o exec_skip_bench tools/attic/exec_skip_bench.hip && ./exec_skip_bench
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ void chain(T* out, int iters, int active) {
    const int lane = threadIdx.x & 63;
    T a0 = (T)lane, a1 = (T)(lane + 1), a2 = (T)(lane + 2), a3 = (T)(lane + 3), a4 = (T)(lane + 4), a5 = (T)(lane + 5), a6 = (T)(lane + 6), a7 = (T)(lane + 7);
    const T m = (T)1.0000001, c = (T)0.5;
    if (lane < active) {
        for (int i = 0; i < iters; i++) {
            a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
            a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename T>
static void run(const char* name) {
    T* d; hipMalloc(&d, sizeof(T) * 256 * 4 * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int active : {64, 32, 16, 1}) {
        for (int waves : {1, 3}) {                       // wavefronts per SIMD
            const int blocks = 256 * 4 * waves;          // one wave per block
            hipLaunchKernelGGL(chain<T>, dim3(blocks), dim3(64), 0, 0, d, 1000, active);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain<T>, dim3(blocks), dim3(64), 0, 0, d, 200000, active);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s, %2d active lanes, %d wave(s) per SIMD: %.2f ms = %.2f clocks per FMA and wave (at 2.4 GHz)\n", name, active, waves, ms,
                   ms * 1e-3 * 2.4e9 / (200000.0 * 8));
        }
    }
    hipFree(d);
}

int main() { run<float>("fp32"); run<double>("fp64"); return 0; }
