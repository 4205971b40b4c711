// handoff_litmus.hip -- word-by-word check of the persistent slice kernel's hand-off (dqmc_amd/csrc/update.hip, slice_kernel)
// under load.  Same grid (1 walk + 64 flush workgroups, 256 threads, ~140 KB of dynamic LDS so that a workgroup owns its CU),
// same instruction forms (8-byte sc1 stores / loads for every shared byte, s_waitcnt vmcnt(0) + barrier + one flag store), the
// same exit-ticket re-arm across launches, 4 windows per launch -- but the payload is a function of the window number, so EVERY
// word a consumer reads is checked:
//   flush side: the 2 x 8 panel words a lane reads must carry this window's value, the 2 x 4 tile words its own last write;
//   walk side:  after the arrivals, 24 columns of G and of GT (23 of them re-read from the previous window: warm in the reader's
//               L2 with the old content) must carry this window's value.
// Variants (argv[1]):
//   0  product form of round 2: per-workgroup arrival words (sc1 stores), one polling wave per workgroup, no fences
//   1  the form the wrong-G events were seen with: every flush wave polls, arrival = atomic add on ONE counter, no fences
//   2  variant 0 + agent-scope release fence before every flag store and agent-scope acquire fence after every poll
//   3  variant 1 + the same fences
// Load (argv[3], bit mask): 1 = three litmus instances at once (three engines on the persistent kernel), 2 = HBM streaming
// kernels on two more streams, 4 = a stream of tiny kernels (kernel-boundary cache maintenance), 8 = a host thread doing small
// synchronous device-to-host copies (system-scope release / acquire on another queue).
// usage: handoff_litmus <variant> <launches> <load mask>
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int N = 256, KD = 32, F = 64, WINDOWS = 4;
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct Sync { unsigned long long seq; unsigned int arrived; unsigned int exits; unsigned int error; unsigned int pad[11]; unsigned int arrive[240]; };
struct Counts { unsigned long long panel_bad, panel_stale, tile_bad, walk_bad, walk_stale, timeouts, handoffs; };

__device__ __forceinline__ double ld_coh(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_coh(double* p, double x) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double pval(unsigned epoch, int m, int j, int which) { return (double)epoch * 65536.0 + (double)(which * 32768 + m * 256 + j); }
__device__ __forceinline__ double gval(unsigned epoch, int a, int b) { return (double)epoch * 131072.0 + (double)(a * 256 + b); }

template <int VARIANT>
__global__ __launch_bounds__(256) void litmus_kernel(double* G, double* GT, double* Up, double* Wp, Sync* sy, unsigned launch, Counts* cnt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool ATOMIC_ARRIVE = (VARIANT & 1) != 0, FENCES = VARIANT >= 2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned long long bad0 = 0, bad1 = 0, stale0 = 0, stale1 = 0, tmo = 0;
    const unsigned e0 = launch * WINDOWS;                               // epoch of window w of this launch: e0 + w
    if (blockIdx.x > 0) {
        const int tile = blockIdx.x - 1;
        const int a0 = (tile % 8) * 32 + (wave & 1) * 16, b0 = (tile / 8) * 32 + (wave >> 1) * 16;
        const int r = lane & 15, kk = lane >> 4, a = a0 + r, b = b0 + r;
        unsigned long long* bcast = reinterpret_cast<unsigned long long*>(smem);
        for (unsigned win = 1;; ++win) {
            unsigned long long word = 0;
            if (ATOMIC_ARRIVE || wave == 0) {
                unsigned spins = 0; bool give_up = false;
                for (;;) {
                    word = __hip_atomic_load(&sy->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(word >> 32) == win) break;
                    if (++spins > SPIN_LIMIT) { give_up = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (give_up) word = ~0ULL;
                if (FENCES) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                if (!ATOMIC_ARRIVE && lane == 0) bcast[win & 1] = word;
            }
            if (!ATOMIC_ARRIVE) { __syncthreads(); word = bcast[win & 1]; }
            if (word == ~0ULL) { tmo++; break; }
            const bool final = (word >> 31) & 1ULL;
            const unsigned ep = e0 + win;
            double uv[8], wv[8], gv[4], gt[4];
#pragma unroll
            for (int s = 0; s < 8; ++s) { const int m = 4 * s + kk; wv[s] = ld_coh(Wp + m * N + b); uv[s] = ld_coh(Up + m * N + a); }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { const int bb = b0 + kk + 4 * reg, aa = a0 + kk + 4 * reg; gv[reg] = ld_coh(G + a + N * bb); gt[reg] = ld_coh(GT + b + N * aa); }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int m = 4 * s + kk;
                if (wv[s] != pval(ep, m, b, 1)) { bad0++; if (wv[s] == pval(ep - 1, m, b, 1)) stale0++; }
                if (uv[s] != pval(ep, m, a, 0)) { bad0++; if (uv[s] == pval(ep - 1, m, a, 0)) stale0++; }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int bb = b0 + kk + 4 * reg, aa = a0 + kk + 4 * reg;
                if (gv[reg] != gval(ep - 1, a, bb)) bad1++;
                if (gt[reg] != gval(ep - 1, b, aa)) bad1++;
                st_coh(G + a + N * bb, gval(ep, a, bb)); st_coh(GT + b + N * aa, gval(ep, b, aa));
            }
            if (final) break;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                if (FENCES) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                if (ATOMIC_ARRIVE) __hip_atomic_fetch_add(&sy->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(&sy->arrive[tile], win, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (bad0) atomicAdd(&cnt->panel_bad, bad0);
        if (stale0) atomicAdd(&cnt->panel_stale, stale0);
        if (bad1) atomicAdd(&cnt->tile_bad, bad1);
    } else {
        const int j = t;
        double* scratch = reinterpret_cast<double*>(smem);
        for (unsigned win = 1;; ++win) {
            const unsigned ep = e0 + win;
            const bool final = win == WINDOWS;
#pragma unroll
            for (int m = 0; m < KD; ++m) { st_coh(Up + m * N + j, pval(ep, m, j, 0)); st_coh(Wp + m * N + j, pval(ep, m, j, 1)); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                if (FENCES) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __hip_atomic_store(&sy->seq, ((unsigned long long)win << 32) | ((unsigned long long)final << 31) | (unsigned long long)KD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (final) break;
            bool broken = false;
            if (wave == 0) {
                unsigned spins = 0;
                for (;;) {
                    bool ok;
                    if (ATOMIC_ARRIVE) ok = __hip_atomic_load(&sy->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)F * win;
                    else { const unsigned av = __hip_atomic_load(&sy->arrive[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = __all(av == win); }
                    if (ok) break;
                    if (++spins > SPIN_LIMIT) { broken = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (FENCES) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                if (lane == 0) scratch[0] = broken ? 1.0 : 0.0;
            }
            __syncthreads();
            if (scratch[0] != 0.0) { tmo++; break; }
            __syncthreads();
            // what the walk's prefetch does after a hand-off: columns of G and of GT, most of them read before the flush as well
#pragma unroll 8
            for (int q = 0; q < 24; ++q) {
                const int c = (int)((ep + q) & 255u);
                const double g = ld_coh(G + j + N * c), h = ld_coh(GT + j + N * c);
                if (g != gval(ep, j, c)) { bad0++; if (g == gval(ep - 1, j, c)) stale0++; }
                if (h != gval(ep, j, c)) { bad0++; if (h == gval(ep - 1, j, c)) stale0++; }
            }
        }
        if (bad0) atomicAdd(&cnt->walk_bad, bad0);
        if (stale0) atomicAdd(&cnt->walk_stale, stale0);
        if (t == 0) atomicAdd(&cnt->handoffs, (unsigned long long)(WINDOWS - 1));
    }
    if (tmo && lane == 0) atomicAdd(&cnt->timeouts, 1ULL);
    (void)bad1; (void)stale1;
    // exit ticket of slice_kernel: the last workgroup to leave re-arms the words for the next launch
    __syncthreads();
    if (t == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(&sy->exits, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (unsigned)F) {
            __hip_atomic_store(&sy->seq, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int f = 0; f < 64; ++f) __hip_atomic_store(&sy->arrive[f], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->exits, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// between two launches the product runs GEMMs that rewrite G / GT with PLAIN stores from every XCD; here: the values the next launch expects
__global__ void rewrite_kernel(double* G, double* GT, unsigned epoch) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = idx & 255, b = idx >> 8;
    G[a + N * b] = gval(epoch, a, b); GT[a + N * b] = gval(epoch, a, b);
}
__global__ void stream_kernel(const double4* __restrict__ src, double4* __restrict__ dst, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { double4 v = src[i]; v.x += 1.0; dst[i] = v; }
}
__global__ void tiny_kernel(double* p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0; }

static std::atomic<bool> g_stop{false};

template <int V>
static void instance(int launches, Counts* out, double* ms) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double *G, *GT, *Up, *Wp; Sync* sy; Counts* cnt;
    CK(hipMalloc(&G, sizeof(double) * N * N)); CK(hipMalloc(&GT, sizeof(double) * N * N));
    CK(hipMalloc(&Up, sizeof(double) * KD * N)); CK(hipMalloc(&Wp, sizeof(double) * KD * N));
    CK(hipMalloc(&sy, sizeof(Sync))); CK(hipMalloc(&cnt, sizeof(Counts)));
    CK(hipMemsetAsync(sy, 0, sizeof(Sync), s)); CK(hipMemsetAsync(cnt, 0, sizeof(Counts), s));
    CK(hipMemsetAsync(Up, 0, sizeof(double) * KD * N, s)); CK(hipMemsetAsync(Wp, 0, sizeof(double) * KD * N, s));
    const size_t lds = 142976;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(litmus_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, s));
    for (int l = 1; l <= launches; ++l) {
        hipLaunchKernelGGL(rewrite_kernel, dim3(256), dim3(256), 0, s, G, GT, (unsigned)l * WINDOWS);     // epoch e0 = state before window 1
        hipLaunchKernelGGL(litmus_kernel<V>, dim3(1 + F), dim3(256), lds, s, G, GT, Up, Wp, sy, (unsigned)l, cnt);
    }
    CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float t = 0.f; CK(hipEventElapsedTime(&t, a, b)); *ms = t;
    CK(hipMemcpy(out, cnt, sizeof(Counts), hipMemcpyDeviceToHost));
    hipFree(G); hipFree(GT); hipFree(Up); hipFree(Wp); hipFree(sy); hipFree(cnt); hipStreamDestroy(s);
}

static void hbm_load() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t n4 = (size_t)64 << 20 >> 5; double4 *x, *y; CK(hipMalloc(&x, n4 * 32)); CK(hipMalloc(&y, n4 * 32)); CK(hipMemsetAsync(x, 0, n4 * 32, s));
    while (!g_stop.load()) { for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(stream_kernel, dim3(1024), dim3(256), 0, s, x, y, n4); CK(hipStreamSynchronize(s)); }
    hipFree(x); hipFree(y); hipStreamDestroy(s);
}
static void tiny_load() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double* p; CK(hipMalloc(&p, 8 * 256 * 256)); CK(hipMemsetAsync(p, 0, 8 * 256 * 256, s));
    while (!g_stop.load()) { for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, s, p); CK(hipStreamSynchronize(s)); }
    hipFree(p); hipStreamDestroy(s);
}
static void d2h_load() {
    int* p; CK(hipMalloc(&p, 64)); CK(hipMemset(p, 0, 64)); int h = 0;
    while (!g_stop.load()) CK(hipMemcpy(&h, p, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(p);
}

template <int V>
static void run(int launches, int load) {
    const int inst = (load & 1) ? 3 : 1;
    std::vector<std::thread> bg;
    g_stop = false;
    if (load & 2) { bg.emplace_back(hbm_load); bg.emplace_back(hbm_load); }
    if (load & 4) bg.emplace_back(tiny_load);
    if (load & 8) bg.emplace_back(d2h_load);
    std::vector<Counts> c(inst); std::vector<double> ms(inst); std::vector<std::thread> th;
    for (int i = 0; i < inst; ++i) th.emplace_back(instance<V>, launches, &c[i], &ms[i]);
    for (auto& x : th) x.join();
    g_stop = true; for (auto& x : bg) x.join();
    Counts s{}; double tmax = 0;
    for (int i = 0; i < inst; ++i) { s.panel_bad += c[i].panel_bad; s.panel_stale += c[i].panel_stale; s.tile_bad += c[i].tile_bad; s.walk_bad += c[i].walk_bad; s.walk_stale += c[i].walk_stale; s.timeouts += c[i].timeouts; s.handoffs += c[i].handoffs; if (ms[i] > tmax) tmax = ms[i]; }
    printf("variant %d load %d: %d instance(s) x %d launches, %llu hand-off rounds, %.1f us per launch | flush side: %llu bad panel words (%llu = previous window's), %llu bad own-tile words | "
           "walk side: %llu bad G / GT words (%llu = previous window's) | %llu time-outs\n", V, load, inst, launches, s.handoffs, 1e3 * tmax / launches, s.panel_bad, s.panel_stale, s.tile_bad, s.walk_bad, s.walk_stale, s.timeouts);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0, launches = argc > 2 ? atoi(argv[2]) : 20000, load = argc > 3 ? atoi(argv[3]) : 0;
    switch (variant) { case 0: run<0>(launches, load); break; case 1: run<1>(launches, load); break; case 2: run<2>(launches, load); break; default: run<3>(launches, load); break; }
    return 0;
}
