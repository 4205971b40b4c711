// microbench2.hip -- calibrations for the register-resident walk of update.hip (gfx950):
// s_memtime stamp overhead, DPP-broadcast fp64 FMA vs plain FMA vs readlane+FMA, LDS-only barrier, dynamic register insert.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../dqmc_amd/csrc/wave.h"
using namespace dq;
typedef double d16v __attribute__((ext_vector_type(16)));
#define STAMP(v) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); v = _t; }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
__device__ __forceinline__ d16v vec_set(d16v v, int q, double x) { v[q] = x; return v; }

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* cyc, int iters) {
    __shared__ double buf[1024];
    const int t = threadIdx.x;
    double x = 1.0 + t * 1e-3, y = 0.5 + t * 1e-4;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    d16v v; for (int m = 0; m < 16; ++m) v[m] = x + m;
    buf[t] = x;
    __syncthreads();
    unsigned long long t0, t1, tsum = 0;
    STAMP(t0)
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { unsigned long long a, b; STAMP(a) STAMP(b) tsum += b - a; }
        if (MODE == 1) {   // 32 DPP FMAs on 4 chains
#pragma unroll
            for (int c = 0; c < 8; ++c)
                asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %1, %4, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %3, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        }
        if (MODE == 2) {   // 32 plain FMAs on 4 chains
#pragma unroll
            for (int c = 0; c < 8; ++c)
                asm volatile("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\tv_fmac_f64 %2, %4, %5\n\tv_fmac_f64 %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        }
        if (MODE == 3) {   // 32 x (2 readlane + FMA)
#pragma unroll
            for (int c = 0; c < 32; ++c) { const double s = readlane_f64(x, c); a0 += s * y; }
        }
        if (MODE == 4) { lds_barrier(); }
        if (MODE == 5) { __syncthreads(); }
        if (MODE == 6) { v = vec_set(v, i & 15, x); x += 1.0; }
        if (MODE == 7) {   // 2-deep dependent LDS read (site -> diag) + ratio math
            const int s = ((int)buf[(i + t) & 1023]) & 1023;
            const double r = 1.0 + (1.0 - buf[s]) * y;
            a0 += fmin(1.0, fabs(x * r * r));
        }
        if (MODE == 8) {   // ballot + ffs + 5 readlanes
            const unsigned long long bal = __ballot(x + i > y * 3.0);
            const int f = __ffsll((long long)bal) - 1;
            a0 += readlane_f64(x, f & 63) + readlane_f64(y, f & 63) + __builtin_amdgcn_readlane(t, f & 63);
        }
        if (MODE == 9) { a0 += y / (x + i); }   // fp64 division
    }
    STAMP(t1)
    if (t == 0) { cyc[0] = t1 - t0; cyc[1] = tsum; }
    out[blockIdx.x * blockDim.x + t] = a0 + a1 + a2 + a3 + v[3] + x;
}

template <int MODE>
void run(const char* name, int iters) {
    double* d; unsigned long long* c; hipMalloc(&d, sizeof(double) * 1024); hipMalloc(&c, 16);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, d, c, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, d, c, iters);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    printf("%-44s %8.1f ns/iter  %8.1f stamp-ticks/iter (inner %0.1f)\n", name, ms * 1e6 / iters, (double)h[0] / iters, (double)h[1] / iters);
    hipFree(d); hipFree(c);
}
int main() {
    const int it = 20000;
    run<0>("back-to-back stamps", it);
    run<1>("32 DPP-bcast fp64 FMAs (4 chains)", it);
    run<2>("32 plain fp64 FMAs (4 chains)", it);
    run<3>("32 x (readlane_f64 + FMA)", it);
    run<4>("lds_barrier, 4 waves", it);
    run<5>("__syncthreads, 4 waves", it);
    run<6>("dynamic d16v insert (s_set_gpr_idx)", it);
    run<7>("site->diag dependent LDS + ratio", it);
    run<8>("ballot + ffs + 5 readlanes", it);
    run<9>("fp64 division", it);
    return 0;
}
