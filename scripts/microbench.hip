// microbench.hip -- calibrates the per-step cost of the serial single-workgroup kernels
// (barriers, LDS hops, DPP reductions, dependent fp64 chains) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../dqmc_amd/csrc/wave.h"
using namespace dq;

template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, int iters) {
    __shared__ double buf[1024];
    __shared__ unsigned long long keys[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double x = 1.0 + t * 1e-3, acc = 0.0;
    buf[t] = x;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { __syncthreads(); }
        if (MODE == 1) { __syncthreads(); __syncthreads(); }
        if (MODE == 2) {   // key max -> LDS -> barrier -> read 16 -> barrier
            unsigned long long key = wave_max_u64((unsigned long long)__double_as_longlong(x) + i);
            if (lane == 0) keys[wave] = key;
            __syncthreads();
            unsigned long long b = keys[0];
            for (int q = 1; q < 16; ++q) { unsigned long long o = keys[q]; b = o > b ? o : b; }
            x += (double)(b & 7);
            __syncthreads();
        }
        if (MODE == 3) {   // one thread publishes 32 doubles, barrier, all read 32 + 32 FMAs, barrier
            if (t == (i & 1023)) for (int c = 0; c < 32; ++c) buf[c] = x + c;
            __syncthreads();
            for (int c = 0; c < 32; ++c) acc += buf[c] * x;
            __syncthreads();
        }
        if (MODE == 4) {   // wave_sum chain (DPP)
            x = wave_sum(x) * 1e-2;
        }
        if (MODE == 5) {   // wave_sum via shuffles
            x = wave_sum_shfl(x) * 1e-2;
        }
        if (MODE == 6) {   // dependent LDS read chain
            x = buf[((int)x) & 1023] + 1.0;
        }
        if (MODE == 7) {   // fp64 division + sqrt dependent
            x = sqrt(x) / (x + 1.0) + 2.0;
        }
        if (MODE == 8) {   // 64 independent FMAs per thread
            double s0 = x, s1 = x + 1, s2 = x + 2, s3 = x + 3;
            for (int c = 0; c < 16; ++c) { s0 = s0 * 1.0000001 + 0.5; s1 = s1 * 1.0000001 + 0.5; s2 = s2 * 1.0000001 + 0.5; s3 = s3 * 1.0000001 + 0.5; }
            x = s0 + s1 + s2 + s3;
        }
    }
    out[blockIdx.x * blockDim.x + t] = x + acc;
}

template <int MODE>
void run(const char* name, int threads, int iters) {
    double* d; hipMalloc(&d, sizeof(double) * 1024 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s threads=%4d  %8.1f ns/iter\n", name, threads, ms * 1e6 / iters);
    hipFree(d);
}

int main() {
    const int it = 20000;
    for (int th : {256, 512, 1024}) {
        run<0>("1 barrier", th, it);
        run<1>("2 barriers", th, it);
        run<2>("wave key-max + LDS + 2 barriers", th, it);
        run<3>("publish 32 dbl + 2 barriers + 32 FMA", th, it);
        run<4>("wave_sum DPP (dependent)", th, it);
        run<5>("wave_sum shuffle (dependent)", th, it);
        run<6>("dependent LDS read", th, it);
        run<7>("sqrt + div (dependent)", th, it);
        run<8>("64 fp64 FMA / thread", th, it);
    }
    return 0;
}
