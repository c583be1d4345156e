// Persistent multi-workgroup kernel for the launch-latency-bound multigrid levels: what does one
// "phase" (operator pass over a small level + grid barrier) cost inside ONE launch, against the same pass as its own
// kernel launch (eager and replayed from a hipGraph)?
//   * data path between workgroups: sc1 (write-through) stores + sc1 loads on both sides -- no agent-scope fences
//     (MI355X_MICROARCH.md: fence(release) ~1.7-6.5 us, fence(acquire) ~1.7 us; "sc1 stores and loads both sides" is valid
//     at any workgroup -> XCD placement)
//   * barrier: every wave drains its stores (s_waitcnt vmcnt(0)), __syncthreads, lane 0 arrives on ONE monotonic counter
//     (relaxed agent atomic) and polls it with sc1 loads + s_sleep; bounded spin (never hangs the GPU)
// Work per phase: one damped-Jacobi sweep x_out = x_in + w D^-1 (b - K x_in) with the library's matrix-free operator
// (grid_apply_pairs) on a (n+1)^2-node level.  The result after NPH phases must be BIT-IDENTICAL to NPH launches of the
// library kernel k_mg_smooth<0,1> -- any stale read (L1 or a remote XCD's L2) shows up as a mismatch.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pylabfea_amd/csrc -o tools/probes/persist_probe tools/probes/persist_probe.hip
#include "plfx_mg.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace plfx;

__device__ __forceinline__ double2 ld_sc1(const double2 *p)
{
    const double *q = reinterpret_cast<const double *>(p);
    return make_double2(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                        __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double2 *p, double2 v)
{
    double *q = reinterpret_cast<double *>(p);
    __hip_atomic_store(q, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Bar {
    unsigned count;   // arrivals (monotonic within a launch)
    unsigned exits;   // workgroups that left the kernel; the last one zeroes both words for the next launch
    unsigned timeout; // sticky: a spin ran into its bound
    unsigned pad;
};

__device__ __forceinline__ void grid_barrier(Bar *bar, unsigned &target, unsigned G)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        target += G;
        __hip_atomic_fetch_add(&bar->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&bar->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) {
                __hip_atomic_store(&bar->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void grid_exit(Bar *bar, unsigned G)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned e = __hip_atomic_fetch_add(&bar->exits, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e == G - 1) {
            __hip_atomic_store(&bar->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&bar->exits, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// NPH smoothing phases in one launch; x0 <-> x1 ping-pong.  MODE 0: sc1 loads/stores; MODE 1: plain loads/stores +
// __threadfence() around the barrier (the textbook form, for the price comparison)
template <int T, int MODE>
__global__ void __launch_bounds__(T)
k_persist(KOp op, const double2 *__restrict__ dinv, const double2 *__restrict__ b, double2 *x0, double2 *x1, double omega, int nph,
          Bar *bar, int skew)
{
    const unsigned G = gridDim.x;
    unsigned target = 0;
    const double2 *M2 = reinterpret_cast<const double2 *>(op.M);
    double2 *src = x0, *dst = x1;
    for (int ph = 0; ph < nph; ph++) {
        if (skew && (int)(blockIdx.x % 7) == ph % 7)  // uneven load: some workgroups arrive late
            for (int s = 0; s < skew; s++) __builtin_amdgcn_s_sleep(8);
        for (int i = blockIdx.x * T + threadIdx.x; i < op.nnode; i += G * T) {
            const double2 di = dinv[i], bi = b[i];
            double2 qv, xi;
            if (MODE == 0) {
                qv = grid_apply_pairs(op.nxn, op.nyn, op.nel, op.tab, i, [&](int q) { return M2[q]; }, [&](int j) { return ld_sc1(src + j); });
                xi = ld_sc1(src + i);
            } else {
                qv = grid_apply_pairs(op.nxn, op.nyn, op.nel, op.tab, i, [&](int q) { return M2[q]; }, [&](int j) { return src[j]; });
                xi = src[i];
            }
            const double2 xo = make_double2(fma(omega * di.x, bi.x - qv.x, xi.x), fma(omega * di.y, bi.y - qv.y, xi.y));
            if (MODE == 0)
                st_sc1(dst + i, xo);
            else
                dst[i] = xo;
        }
        if (MODE == 1) __threadfence();
        grid_barrier(bar, target, G);
        if (MODE == 1) __threadfence();
        double2 *t = src;
        src = dst;
        dst = t;
    }
    grid_exit(bar, G);
}


// The verdict's exact proposal: levels resident on ONE XCD, XCD-local hand-off through that XCD's L2 -- plain stores (stay in
// the L2), sc1 loads (bypass the L1, served by the L2), no write-through.  Only the blocks b % 8 == 0 of an 8 G grid work
// (observed placement: block b runs on XCD b % 8); valid ONLY if all of them report the same XCC_ID (recorded in xcc[]).
template <int T>
__global__ void __launch_bounds__(T)
k_persist_xcd(KOp op, const double2 *__restrict__ dinv, const double2 *__restrict__ b, double2 *x0, double2 *x1, double omega, int nph,
              Bar *bar, unsigned *xcc)
{
    if (blockIdx.x & 7) return;
    const unsigned G = gridDim.x >> 3, rank = blockIdx.x >> 3;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[rank] = id & 15;
    }
    unsigned target = 0;
    const double2 *M2 = reinterpret_cast<const double2 *>(op.M);
    double2 *src = x0, *dst = x1;
    for (int ph = 0; ph < nph; ph++) {
        for (int i = rank * T + threadIdx.x; i < op.nnode; i += G * T) {
            const double2 di = dinv[i], bi = b[i];
            const double2 qv = grid_apply_pairs(op.nxn, op.nyn, op.nel, op.tab, i, [&](int q) { return M2[q]; }, [&](int j) { return ld_sc1(src + j); });
            const double2 xi = ld_sc1(src + i);
            dst[i] = make_double2(fma(omega * di.x, bi.x - qv.x, xi.x), fma(omega * di.y, bi.y - qv.y, xi.y));
        }
        grid_barrier(bar, target, G);
        double2 *t = src;
        src = dst;
        dst = t;
    }
    grid_exit(bar, G);
}

// barrier only
template <int T>
__global__ void __launch_bounds__(T) k_bar_only(int nph, Bar *bar)
{
    const unsigned G = gridDim.x;
    unsigned target = 0;
    for (int ph = 0; ph < nph; ph++) grid_barrier(bar, target, G);
    grid_exit(bar, G);
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

template <int T, int MODE>
double run_persist(hipStream_t s, int G, KOp op, double2 *dinv, double2 *b, double2 *x0, double2 *x1, int nph, Bar *bar, int skew,
                   int reps)
{
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_persist<T, MODE>), dim3(G), dim3(T), 0, s, op, dinv, b, x0, x1, 0.65, nph, bar, skew);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_persist<T, MODE>), dim3(G), dim3(T), 0, s, op, dinv, b, x0, x1, 0.65, nph, bar, skew);
    CK(hipStreamSynchronize(s));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main(int argc, char **argv)
{
    const int nph = argc > 1 ? atoi(argv[1]) : 24;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Bar *bar;
    CK(hipMalloc(&bar, sizeof(Bar)));
    CK(hipMemset(bar, 0, sizeof(Bar)));
    // geometry table of a unit square element (any numbers do for timing / bit comparison; use the library's structure)
    double htab[64];
    for (int i = 0; i < 64; i++) htab[i] = 0.05 + 0.01 * ((i * 37) % 11);
    double *tab;
    CK(hipMalloc(&tab, sizeof(htab)));
    CK(hipMemcpy(tab, htab, sizeof(htab), hipMemcpyHostToDevice));
    for (int n : {256, 128, 64, 32}) {
        const int nn = (n + 1) * (n + 1), nel = n * n;
        std::vector<double> hM(6 * (size_t)nel), hd(2 * (size_t)nn), hb(2 * (size_t)nn), hx(2 * (size_t)nn);
        srand(1);
        for (auto &v : hM) v = 1. + 0.1 * (rand() / (double)RAND_MAX);
        for (auto &v : hd) v = 1e-2 * (0.5 + rand() / (double)RAND_MAX);
        for (auto &v : hb) v = rand() / (double)RAND_MAX - 0.5;
        for (auto &v : hx) v = rand() / (double)RAND_MAX - 0.5;
        double *M;
        double2 *dinv, *b, *x0, *x1, *r0, *r1;
        CK(hipMalloc(&M, hM.size() * 8));
        CK(hipMalloc(&dinv, nn * 16));
        CK(hipMalloc(&b, nn * 16));
        CK(hipMalloc(&x0, nn * 16));
        CK(hipMalloc(&x1, nn * 16));
        CK(hipMalloc(&r0, nn * 16));
        CK(hipMalloc(&r1, nn * 16));
        CK(hipMemcpy(M, hM.data(), hM.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dinv, hd.data(), nn * 16, hipMemcpyHostToDevice));
        CK(hipMemcpy(b, hb.data(), nn * 16, hipMemcpyHostToDevice));
        KOp op{};
        op.nnode = nn;
        op.nxn = n + 1;
        op.nyn = n + 1;
        op.nel = nel;
        op.M = M;
        op.tab = tab;
        CgScalars *sc;
        CK(hipMalloc(&sc, sizeof(CgScalars)));
        CK(hipMemset(sc, 0, sizeof(CgScalars)));
        // reference: nph launches of the library kernel
        const int grid = (nn + BLOCK - 1) / BLOCK;
        auto ref_once = [&](hipStream_t st) {
            double2 *src = r0, *dst = r1;
            for (int ph = 0; ph < nph; ph++) {
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_smooth<0, 1>), dim3(grid < 1024 ? grid : 1024), dim3(BLOCK), 0, st, op, (const double2 *)dinv,
                                   (const double2 *)b, (const double2 *)src, dst, 0.65, 0, (const CgScalars *)sc, DotOut{});
                std::swap(src, dst);
            }
        };
        CK(hipMemcpy(r0, hx.data(), nn * 16, hipMemcpyHostToDevice));
        ref_once(s);
        CK(hipStreamSynchronize(s));
        std::vector<double> href(2 * (size_t)nn);
        CK(hipMemcpy(href.data(), (nph & 1) ? r1 : r0, nn * 16, hipMemcpyDeviceToHost));
        // timing of the launches: eager and graph
        const int reps = 200;
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) ref_once(s);
        CK(hipStreamSynchronize(s));
        const double us_eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        hipGraph_t g;
        hipGraphExec_t gx;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        ref_once(s);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&gx, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(gx, s));
        CK(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) CK(hipGraphLaunch(gx, s));
        CK(hipStreamSynchronize(s));
        const double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        {   // the same number of launches, but ROTATING through the five kernel types of a V-cycle level (different code objects)
            auto mixed_once = [&](hipStream_t st) {
                double2 *src = r0, *dst = r1;
                const int g = grid < 1024 ? grid : 1024;
                for (int ph = 0; ph < nph; ph++) {
                    switch (ph % 6) {
                    case 0:
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_smooth2_zero<0, 1>), dim3(g), dim3(BLOCK), 0, st, op, (const double2 *)dinv, (const double2 *)b, dst, 0.65,
                                           (const CgScalars *)sc);
                        break;
                    case 1:
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_residual<0, 1>), dim3(g), dim3(BLOCK), 0, st, op, (const double2 *)dinv, (const double2 *)b,
                                           (const double2 *)src, dst, (const CgScalars *)sc);
                        break;
                    case 2:
                        hipLaunchKernelGGL(k_mg_restrict, dim3(g), dim3(BLOCK), 0, st, n / 2 + 1, n / 2 + 1, n + 1, n + 1, (const double2 *)src, (const double2 *)dinv, dst);
                        break;
                    case 3:
                        hipLaunchKernelGGL(k_mg_prolong_add, dim3(g), dim3(BLOCK), 0, st, n + 1, n + 1, n / 2 + 1, (const double2 *)src, (const double2 *)dinv, dst);
                        break;
                    default:
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mg_smooth<0, 1>), dim3(g), dim3(BLOCK), 0, st, op, (const double2 *)dinv, (const double2 *)b,
                                           (const double2 *)src, dst, 0.65, 0, (const CgScalars *)sc, DotOut{});
                    }
                    std::swap(src, dst);
                }
            };
            mixed_once(s);
            CK(hipStreamSynchronize(s));
            auto tm = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; r++) mixed_once(s);
            CK(hipStreamSynchronize(s));
            const double us_mixed = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tm).count() / reps;
            printf("level %4d^2: %d launches rotating through 5 kernel types (s2z, residual, restrict, prolong, smooth, smooth): %.1f us = %.2f per launch\n", n,
                   nph, us_mixed, us_mixed / nph);
        }
        if (argc > 2) {  // does a TIMED hipEvent recorded on the stream change the cost of the launches that follow?
            hipEvent_t ev0, ev1;
            CK(hipEventCreate(&ev0));
            CK(hipEventCreate(&ev1));
            CK(hipEventRecord(ev0, s));
            auto te = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; r++) ref_once(s);
            CK(hipEventRecord(ev1, s));
            CK(hipStreamSynchronize(s));
            const double us_ev = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - te).count() / reps;
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, ev0, ev1));
            te = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; r++) ref_once(s);
            CK(hipStreamSynchronize(s));
            const double us_after = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - te).count() / reps;
            printf("level %4d^2: eager launches between two timed hipEvents %.2f us per launch (events say %.2f); afterwards %.2f per launch\n", n,
                   us_ev / nph, 1e3 * ms / reps / nph, us_after / nph);
        }
        printf("level %4d^2 elements (%6d nodes), %d phases: launches eager %.1f us (%.2f per phase), hipGraph %.1f us (%.2f per phase)\n", n, nn,
               nph, us_eager, us_eager / nph, us_graph, us_graph / nph);
        auto check = [&](const char *what, int G, int T, double us) {
            std::vector<double> h(2 * (size_t)nn);
            CK(hipMemcpy(h.data(), (nph & 1) ? x1 : x0, nn * 16, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < h.size(); i++) bad += (h[i] != href[i]);
            Bar hb2;
            CK(hipMemcpy(&hb2, bar, sizeof(Bar), hipMemcpyDeviceToHost));
            printf("    %-22s G=%3d T=%4d: %8.1f us = %5.2f us per phase   mismatching words %zu%s\n", what, G, T, us, us / nph, bad,
                   hb2.timeout ? "  BARRIER TIMEOUT" : "");
            if (hb2.timeout) CK(hipMemset(bar, 0, sizeof(Bar)));
        };
        for (int G : {16, 32, 64, 128, 256}) {
            // one correctness run from the same start vector (the timing loop keeps iterating on its own output)
            for (int skew : {0, 40}) {
                CK(hipMemcpy(x0, hx.data(), nn * 16, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_persist<1024, 0>), dim3(G), dim3(1024), 0, s, op, dinv, b, x0, x1, 0.65, nph, bar, skew);
                CK(hipStreamSynchronize(s));
                check(skew ? "sc1 (uneven load)" : "sc1", G, 1024, 0.);
            }
            double us = run_persist<1024, 0>(s, G, op, dinv, b, x0, x1, nph, bar, 0, reps);
            printf("    %-22s G=%3d T=%4d: %8.1f us = %5.2f us per phase\n", "sc1 timing", G, 1024, us, us / nph);
            us = run_persist<256, 0>(s, G, op, dinv, b, x0, x1, nph, bar, 0, reps);
            printf("    %-22s G=%3d T=%4d: %8.1f us = %5.2f us per phase\n", "sc1 timing", G, 256, us, us / nph);
            CK(hipMemcpy(x0, hx.data(), nn * 16, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_persist<1024, 1>), dim3(G), dim3(1024), 0, s, op, dinv, b, x0, x1, 0.65, nph, bar, 0);
            CK(hipStreamSynchronize(s));
            check("plain + __threadfence", G, 1024, 0.);
            us = run_persist<1024, 1>(s, G, op, dinv, b, x0, x1, nph, bar, 0, reps);
            printf("    %-22s G=%3d T=%4d: %8.1f us = %5.2f us per phase\n", "fence timing", G, 1024, us, us / nph);

            if (G <= 32) {
                unsigned *xcc;
                CK(hipMalloc(&xcc, 64 * 4));
                CK(hipMemcpy(x0, hx.data(), nn * 16, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_persist_xcd<256>), dim3(8 * G), dim3(256), 0, s, op, dinv, b, x0, x1, 0.65, nph, bar, xcc);
                CK(hipStreamSynchronize(s));
                unsigned hxcc[64];
                CK(hipMemcpy(hxcc, xcc, G * 4, hipMemcpyDeviceToHost));
                int same = 1;
                for (int q = 1; q < G; q++) same &= (hxcc[q] == hxcc[0]);
                check(same ? "one XCD (all same XCC)" : "one XCD (XCC DIFFER)", G, 256, 0.);
                CK(hipStreamSynchronize(s));
                auto t1 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; r++)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_persist_xcd<256>), dim3(8 * G), dim3(256), 0, s, op, dinv, b, x0, x1, 0.65, nph, bar, xcc);
                CK(hipStreamSynchronize(s));
                us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / reps;
                printf("    %-22s G=%3d T=%4d: %8.1f us = %5.2f us per phase\n", "one-XCD timing", G, 256, us, us / nph);
                CK(hipFree(xcc));
            }
        }
        CK(hipFree(M)); CK(hipFree(dinv)); CK(hipFree(b)); CK(hipFree(x0)); CK(hipFree(x1)); CK(hipFree(r0)); CK(hipFree(r1)); CK(hipFree(sc));
    }
    // barrier alone
    for (int G : {16, 32, 64, 128, 256}) {
        const int nb = 200;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bar_only<1024>), dim3(G), dim3(1024), 0, s, nb, bar);
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 20; r++) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_bar_only<1024>), dim3(G), dim3(1024), 0, s, nb, bar);
        CK(hipStreamSynchronize(s));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20;
        printf("barrier alone G=%3d T=1024: %.2f us per barrier\n", G, us / nb);
    }
    return 0;
}
