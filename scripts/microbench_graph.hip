// Dependent-launch floor: a chain of K tiny kernels (each reads what the previous wrote) enqueued one by one on a stream, against the same
// chain captured once into a hipGraph and replayed.  Prints microseconds per kernel (HIP events around the whole chain, median of 9).
// build: hipcc -O3 --offload-arch=gfx950 -o scripts/mb_build/microbench_graph scripts/microbench_graph.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void step(double* p, int blocks_work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = p[i];
    for (int r = 0; r < blocks_work; ++r) v = v * 1.0000001 + 1e-9;
    p[i] = v;
}
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 1000;
    double* d; CK(hipMalloc(&d, sizeof(double) * 256 * 256)); CK(hipMemset(d, 0, sizeof(double) * 256 * 256));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {64}) {
        for (int work : {0, 50, 100, 200, 400, 2000}) {
            auto chain = [&]() { for (int k = 0; k < K; ++k) hipLaunchKernelGGL(step, dim3(grid), dim3(256), 0, s, d, work); };
            std::vector<float> ts, tg;
            for (int rep = 0; rep < 9; ++rep) {
                CK(hipEventRecord(e0, s)); chain(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
            }
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); chain(); CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 9; ++rep) {
                CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tg.push_back(ms);
            }
            std::sort(ts.begin(), ts.end()); std::sort(tg.begin(), tg.end());
            printf("grid %3d x 256, %4d fma per thread: stream %.2f us per kernel, graph replay %.2f us per kernel (chain of %d)\n", grid, work, 1e3 * ts[4] / K, 1e3 * tg[4] / K, K);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
