// microbench_mfma.hip -- achievable v_mfma_f64_16x16x4_f64 rate (dense loop, independent accumulators, 1..4 workgroups per CU).
// Measured on the MI355X box: 22.9 / 32.1 / 41.9 / 44.7 / 35.5 TFLOP/s for (NACC, WG/CU) = (1,1) (4,1) (4,2) (4,4) (8,1): the practical
// ceiling of this instruction is ~45 TFLOP/s, 57 % of the 78.6 TFLOP/s data-sheet figure.
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    d4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = d4{0, 0, 0, 0};
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    double s = 0; for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wgs_per_cu) {
    double* d; hipMalloc(&d, sizeof(double) * 256 * 256 * 16);
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters); hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * 4 /*waves*/ * iters * NACC * 2048.0;
    printf("NACC %d, %d WG/CU: %.1f TFLOP/s\n", NACC, wgs_per_cu, flop / (ms * 1e-3) / 1e12);
}
int main() { run<1>(1); run<4>(1); run<4>(2); run<4>(4); run<8>(1); return 0; }
