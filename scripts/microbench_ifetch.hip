// microbench_ifetch.hip -- does straight-line code run slower than the same instructions in a loop?  One wave executes
// NINST dependent-free v_fma_f64 (8 accumulators) either as a rolled loop whose body fits the instruction cache or fully
// unrolled (NINST * 8 bytes of code, executed once).  Clocks by s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define FMA8 "v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t" \
             "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7\n\t"
#define FMA64 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8
#define FMA512 FMA64 FMA64 FMA64 FMA64 FMA64 FMA64 FMA64 FMA64
#define FMA4096 FMA512 FMA512 FMA512 FMA512 FMA512 FMA512 FMA512 FMA512
#define BODY(S) asm volatile(S : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y))
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* clk, int mode) {
    double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, x = 1.0 + threadIdx.x * 1e-9, y = 0.999;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (mode == 0) { for (int i = 0; i < 256; ++i) BODY(FMA64); }                 // 16384 FMAs, 512-byte loop body
    else if (mode == 1) { for (int i = 0; i < 4; ++i) BODY(FMA4096); }            // 16384 FMAs, 32 KB loop body (fits the 64 KB cache)
    else { BODY(FMA4096); BODY(FMA4096); BODY(FMA4096); BODY(FMA4096); }         // 16384 FMAs, 128 KB straight line
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    double* d; unsigned long long* c; hipMalloc(&d, 8 * 64); hipMalloc(&c, 8);
    const char* names[3] = {"loop, 512 B body", "loop, 32 KB body", "straight line, 128 KB"};
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, mode); hipDeviceSynchronize();
            unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            printf("%-24s run %d: %llu memtime ticks for 16384 FMAs = %.2f ticks per instruction\n", names[mode], rep, h, h / 16384.0);
        }
    return 0;
}
