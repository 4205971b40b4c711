// checkerboard.hip -- checkerboard break-up of the kinetic propagator, applied pair by pair.
//
// The reference multiplies by the dense exp(-+dtau K) (source/dqmc.cpp:78-132, two N^3 GEMMs per wrap) and lists the
// checkerboard break-up as future work (README.md:40).  With the bonds of K split into groups of disjoint site pairs,
//     exp(-dtau K) ~= E = f * E_{G-1} ... E_1 E_0,      E_g = prod over pairs (i, j) of the 2x2 block [c s; s c],
// with c = cosh(dtau t), s = sinh(dtau t), f = exp(dtau mu), and E^-1 = (1/f) E_0^-1 ... E_{G-1}^-1 with s -> -s.  A product
// E * M touches every column of M independently: a workgroup stages a strip of columns in LDS, runs the G pair passes on
// it (ping-pong buffers, one barrier per group) and writes the strip back -- 2 N^2 doubles of HBM traffic, O(G N^2) flops.
// A product from the right is the same kernel on the transposed operand (every E_g is symmetric), so the kernel can take
// row / column scalings on the way in and out and write its result transposed as well: the two launches of a wrap hand
// the intermediate over transposed and finish with both G and G^T (the local-update walk reads rows of G from G^T).
#include "common.h"
#include <cstdlib>

namespace dq {

template <int CB>
__global__ __launch_bounds__(1024) void cb_apply_kernel(CbDesc d, int partner_in_lds) {
    extern __shared__ double cb_lds[];                       // 2 strips of CB columns, then the pair tables
    const int c = blockIdx.y, n = d.n, t = threadIdx.x, nthr = blockDim.x, G = d.n_groups;
    const int col0 = blockIdx.x * CB;
    const int ncol = (n - col0 < CB) ? n - col0 : CB;
    double* a = cb_lds; double* b = cb_lds + (size_t)CB * n;
    int* pl = reinterpret_cast<int*>(b + (size_t)CB * n);
    const double* par = d.par + (long)c * d.par_stride;
    const double ch = par[0];
    const double sh = d.inverse ? -par[1] : par[1];
    const double f = d.inverse ? par[3] : par[2];
    const double* in = d.in.at(c);
    const double* rsi = d.rs_in.p ? d.rs_in.at(c) : nullptr;
    const double* csi = d.cs_in.p ? d.cs_in.at(c) : nullptr;
    // every global load of the strip and of the pair tables is issued before the first use
    double csv[CB];
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) csv[cc] = (csi && cc < ncol) ? csi[col0 + cc] : 1.0;
    for (int r = t; r < n; r += nthr) {
        double v[CB];
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) v[cc] = (cc < ncol) ? in[r + (size_t)n * (col0 + cc)] : 0.0;
        const double sr = rsi ? rsi[r] : 1.0;
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) a[cc * n + r] = v[cc] * sr * csv[cc];
    }
    if (partner_in_lds) for (int k = t; k < G * n; k += nthr) pl[k] = d.partner[k];
    __syncthreads();
    for (int gi = 0; gi < G; ++gi) {
        const int g = d.reverse ? G - 1 - gi : gi;
        for (int r = t; r < n; r += nthr) {
            const int p = partner_in_lds ? pl[g * n + r] : d.partner[(size_t)g * n + r];
            const double s = (p == r) ? 0.0 : sh;            // a site outside every pair of the group passes through
            const double cdiag = (p == r) ? 1.0 : ch;
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) b[cc * n + r] = cdiag * a[cc * n + r] + s * a[cc * n + p];
        }
        __syncthreads();
        double* tmp = a; a = b; b = tmp;
    }
    const double* rso = d.rs_out.p ? d.rs_out.at(c) : nullptr;
    const double* cso = d.cs_out.p ? d.cs_out.at(c) : nullptr;
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) csv[cc] = (cso && cc < ncol) ? f * cso[col0 + cc] : f;
    if (d.out.p) {
        double* out = d.out.at(c);
        for (int r = t; r < n; r += nthr) {
            const double sr = rso ? rso[r] : 1.0;
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) if (cc < ncol) out[r + (size_t)n * (col0 + cc)] = a[cc * n + r] * sr * csv[cc];
        }
    }
    if (d.outT.p) {                                          // outT[col, r]: CB consecutive doubles per row r
        double* outT = d.outT.at(c);
        for (int e = t; e < CB * n; e += nthr) {
            const int r = e / CB, cc = e % CB;
            if (cc < ncol) {
                double v = a[cc * n + r] * (cso ? f * cso[col0 + cc] : f);
                if (rso) v *= rso[r];
                outT[(col0 + cc) + (size_t)n * r] = v;
            }
        }
    }
}

int launch_cb_apply(const CbDesc& d, int n_chains, hipStream_t s) {
    if (d.n <= 0 || d.n > 4096 || d.n_groups <= 0 || !d.partner || !d.par || !d.in.p || (!d.out.p && !d.outT.p)) { set_error("launch_cb_apply: bad argument"); return -1; }
    if (d.outT.p && d.outT.p == d.in.p) { set_error("launch_cb_apply: the transposed output may not alias the input"); return -1; }
    const int n = d.n;
    // columns per workgroup: 64 KiB of LDS hold the two strips and (when they fit beside them) the pair tables
    int cb = n <= 320 ? 2 : n <= 1024 ? 4 : n <= 2048 ? 2 : 1;     // measured (scripts/cb_time.py): 4.6 us at N = 256 with 2, 7.7 us at N = 576 with 2 or 4
    while (cb > 1 && (size_t)2 * cb * n * sizeof(double) > 65536) cb >>= 1;
    size_t lds = (size_t)2 * cb * n * sizeof(double);
    const size_t tables = (size_t)d.n_groups * n * sizeof(int);
    const int in_lds = lds + tables <= 65536 ? 1 : 0;
    if (in_lds) lds += tables;
    const dim3 grid((unsigned)((n + cb - 1) / cb), n_chains);
    const dim3 block((unsigned)(n >= 1024 ? 1024 : ((n + 63) / 64) * 64));      // one row per thread up to N = 1024
    switch (cb) {
        case 8: hipLaunchKernelGGL(cb_apply_kernel<8>, grid, block, lds, s, d, in_lds); break;
        case 4: hipLaunchKernelGGL(cb_apply_kernel<4>, grid, block, lds, s, d, in_lds); break;
        case 2: hipLaunchKernelGGL(cb_apply_kernel<2>, grid, block, lds, s, d, in_lds); break;
        default: hipLaunchKernelGGL(cb_apply_kernel<1>, grid, block, lds, s, d, in_lds); break;
    }
    DQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace dq
