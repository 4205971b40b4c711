// microbench_gj.hip -- time of one gj_panel_mw_kernel launch (n = 256, first panel) with pieces of the step removed
// (GJ_EXP bitmask, see lu_gj.hip: 1 no FMAs, 2 no barrier, 4 no DPP argmax, 8 no row publish, 16 no key read); timing only,
// the variants are not correct factorisations.  Build: hipcc -O3 --offload-arch=gfx950 -DGJ_EXP=<mask> -Idqmc_amd/csrc scripts/microbench_gj.hip
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
    for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(a);
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(dq::gj_panel_mw_kernel, dim3(1, 1), dim3(256), 0, 0, dq::CMat(A, 0), rowpos, (long)n, perm, (long)n, tinv, ld, 0, info, n, 0);
        hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        if (pass) printf("GJ_EXP=%d: %.2f us per launch (back-to-back, incl. launch gap)\n",
#ifdef GJ_EXP
                         GJ_EXP,
#else
                         0,
#endif
                         ms * 1e3 / reps);
    }
    return 0;
}
