// microbench_gj.hip -- time of one Gauss-Jordan panel launch (n = 256): the panel kernels of lu_gj.hip against the number of
// elimination steps (k0 = 256 - steps: the last, partial panel), and the multi-wave kernel with pieces of its step removed
// (GJ_EXP bitmask, see lu_gj.hip: 1 no FMAs, 2 no barrier, 4 no DPP argmax, 8 no row publish, 16 no key read); timing only, the
// GJ_EXP variants are not correct factorisations.
// Build: hipcc -O3 --offload-arch=gfx950 [-DGJ_EXP=<mask>] -Idqmc_amd/csrc scripts/microbench_gj.hip
#include "../dqmc_amd/csrc/lu_gj.hip"
#include <cstdio>
#include <vector>
namespace dq { void set_error(const std::string&) {} const char* get_error() { return ""; } }
int main() {
    const int n = 256;
    std::vector<double> h((size_t)n * n);
    unsigned s = 12345u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
    double *A, *tinv, *ld; int *rowpos, *perm, *info;
    hipMalloc(&A, sizeof(double) * n * n); hipMalloc(&tinv, sizeof(double) * 2048); hipMalloc(&ld, 8);
    hipMalloc(&rowpos, 4 * n); hipMalloc(&perm, 4 * n); hipMalloc(&info, 4);
    hipMemcpy(A, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice); hipMemset(info, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 200;
#ifdef GJ_EXP
    const int mask = GJ_EXP;
#else
    const int mask = 0;
#endif
    for (int kern = 0; kern < 2; ++kern)
        for (int steps : {32, 16, 8, 4}) {
            const int k0 = n - steps;
            float ms = 0.f;
            for (int pass = 0; pass < 2; ++pass) {
                hipEventRecord(a);
                for (int r = 0; r < reps; ++r) {
                    hipMemsetAsync(rowpos, 0xFF, 4 * n, 0);       // every row live (the memset is part of every variant's time)
                    if (kern == 0) hipLaunchKernelGGL(dq::gj_panel_mw_kernel, dim3(1, 1), dim3(256), 0, 0, dq::CMat(A, 0), rowpos, (long)n, perm, (long)n, tinv, ld, 0, info, n, k0);
                    else hipLaunchKernelGGL((dq::gj_panel_kernel<4>), dim3(1, 1), dim3(64), 0, 0, dq::CMat(A, 0), rowpos, (long)n, perm, (long)n, tinv, ld, 0, info, n, k0);
                }
                hipEventRecord(b); hipDeviceSynchronize();
                hipEventElapsedTime(&ms, a, b);
            }
            printf("GJ_EXP=%d %s steps=%2d: %.2f us per (memset + launch)\n", mask, kern == 0 ? "multi-wave " : "single-wave", steps, ms * 1e3 / reps);
        }
    // measured on the MI355X box (us per memset + launch; steps = 32 / 16 / 8 / 4):
    //   multi-wave   32.8 / 23.0 / 16.2 / 12.4      single-wave  35.6 / 27.3 / 21.3 / 17.5
    //   a two-wave column-split variant (wave A pivots on columns 0..15 and streams multipliers to wave B through an LDS queue, no
    //   pivot-row traffic; tried and dropped): 32.3 / 22.2 / 16.6 / 13.4 -- every design lands on 0.5 .. 0.9 us per step
    return 0;
}
