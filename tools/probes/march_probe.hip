// Fine-level operator kernels that MARCH along x with the 3 x 3 stencil window in registers (round 3; profiles/r03g: at
// 2048^2 -- beyond the Infinity Cache -- k_mg_smooth<1,1> moves 1.15x and k_spmv<1,1> 1.38x its algorithmic bytes: the
// 9-node / 4-element gathers re-read what the L2 has lost).  A wave owns 64 consecutive rows k and a range of LC columns j:
// stepping j -> j+1 it loads only the NEW column of the vector (3 entries per lane: rows k-1, k, k+1) and the NEW element
// column (2 elements x 3 generator pairs); the other 6 vector entries and 2 elements stay in registers.
//   loads per node: 3 (1 + 2/LC) + 6 (1 + 1/LC) + own streams  instead of  9 + 12 + own streams
// Results are bit-identical to the gather form (same arithmetic in the same order).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pylabfea_amd/csrc -o tools/probes/march_probe tools/probes/march_probe.hip
#include "plfx_mg.hpp"
#include <cstdio>
#include <vector>
using namespace plfx;

// (Gen3 = the generator pairs of one element: plfx_kernels.hpp)

__device__ __forceinline__ double2 stencil(const double2 (&u)[3][3], const Gen3 (&m)[2][2], const double *tab)
{
    double qx = 0., qy = 0.;
#pragma unroll
    for (int pj = 0; pj < 2; pj++)
#pragma unroll
        for (int pk = 0; pk < 2; pk++) {
            const int p = pj * 2 + pk;
            const double *T = tab + p * 16;
            double A1 = 0., A2 = 0., A3 = 0., A4 = 0., A5 = 0., A6 = 0., A7 = 0., A8 = 0.;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const double2 ub = u[pj + (b >> 1)][pk + (b & 1)];
                const double sxx = T[b * 4 + 0], syy = T[b * 4 + 1], sxy = T[b * 4 + 2], syx = T[b * 4 + 3];
                A1 = fma(sxx, ub.x, A1); A2 = fma(sxx, ub.y, A2); A3 = fma(syy, ub.x, A3); A4 = fma(syy, ub.y, A4);
                A5 = fma(sxy, ub.x, A5); A6 = fma(sxy, ub.y, A6); A7 = fma(syx, ub.x, A7); A8 = fma(syx, ub.y, A8);
            }
            const double Mxx = m[pj][pk].a.x, Mxy = m[pj][pk].a.y, Mxs = m[pj][pk].b.x, Myy = m[pj][pk].b.y, Mys = m[pj][pk].c.x, Mss = m[pj][pk].c.y;
            qx = fma(Mxx, A1, fma(Mxs, A5 + A7 + A2, fma(Mss, A3 + A8, fma(Mxy, A6, fma(Mys, A4, qx)))));
            qy = fma(Mxy, A7, fma(Mys, A3 + A8 + A6, fma(Mxs, A1, fma(Mss, A5 + A2, fma(Myy, A4, qy)))));
        }
    return make_double2(qx, qy);
}

// wave task (rk, rj): rows [64 rk, 64 rk + 64), columns [LC rj, LC rj + LC); xf(node) = vector entry, emit(node, q) per node
template <int LC, class XF, class EM>
__device__ __forceinline__ void march(int nxn, int nyn, int nel, const double2 *__restrict__ M2, const double *tab, XF xf, EM emit)
{
    const int nye = nyn - 1, nxe = nxn - 1;
    const int nrk = (nyn + 63) >> 6, nrj = (nxn + LC - 1) / LC, ntask = nrk * nrj;
    const int lane = threadIdx.x & 63;
    const int wpb = BLOCK >> 6;
    const int nwave = gridDim.x * wpb;
    // XCD-aware and balanced: block b runs on XCD b % 8; XCD x takes the contiguous task range [x ntask / 8, (x + 1) ntask / 8)
    // (rows fastest: consecutive tasks share their halo rows / columns in that XCD's L2), its blocks stride through it
    (void)nwave;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nbx = max(1, (int)gridDim.x >> 3);
    const int t0 = (int)((long long)ntask * xcd / 8), t1 = (int)((long long)ntask * (xcd + 1) / 8);
    for (int task = t0 + lb * wpb + (threadIdx.x >> 6); task < t1; task += nbx * wpb) {
        const int rj = task / nrk, rk = task - rj * nrk;
        const int k = (rk << 6) + lane;
        const bool act = k < nyn;
        const int kc = min(k, nye), km = max(kc - 1, 0), kp = min(kc + 1, nye);
        const int j0 = rj * LC, j1 = min(j0 + LC, nxn);
        const int ek0 = min(max(kc - 1, 0), nye - 1), ek1 = min(kc, nye - 1);   // element rows k-1, k (clamped)
        const bool ok0 = kc - 1 >= 0, ok1 = kc < nye;
        double2 u[3][3];
        Gen3 m[2][2];
        auto load_col = [&](int jj, double2 (&col)[3]) {
            const int jc = min(max(jj, 0), nxe);
            col[0] = xf(jc * nyn + km); col[1] = xf(jc * nyn + kc); col[2] = xf(jc * nyn + kp);
        };
        auto load_el = [&](int ej, Gen3 (&g)[2]) {
            const bool okj = ej >= 0 && ej < nxe;
            const int ec = min(max(ej, 0), nxe - 1);
            const size_t e0 = (size_t)ec * nye + ek0, e1 = (size_t)ec * nye + ek1;
            Gen3 g0 = {M2[e0], M2[(size_t)nel + e0], M2[(size_t)2 * nel + e0]};
            Gen3 g1 = {M2[e1], M2[(size_t)nel + e1], M2[(size_t)2 * nel + e1]};
            const double2 z = make_double2(0., 0.);
            if (!(okj && ok0)) g0 = {z, z, z};
            if (!(okj && ok1)) g1 = {z, z, z};
            g[0] = g0; g[1] = g1;
        };
        load_col(j0 - 1, u[0]);
        load_col(j0, u[1]);
        load_el(j0 - 1, m[0]);
        for (int j = j0; j < j1; j++) {
            load_col(j + 1, u[2]);
            load_el(j, m[1]);
            if (act) emit(j * nyn + k, stencil(u, m, tab), u[1][1]);
#pragma unroll
            for (int r = 0; r < 3; r++) { u[0][r] = u[1][r]; u[1][r] = u[2][r]; }
            m[0][0] = m[1][0]; m[0][1] = m[1][1];
        }
    }
}

// the same with the loads of step j + 1 issued BEFORE the arithmetic of step j (one column / element column in flight)
template <int LC, class XF, class EM>
__device__ __forceinline__ void marchp(int nxn, int nyn, int nel, const double2 *__restrict__ M2, const double *tab, XF xf, EM emit)
{
    const int nye = nyn - 1, nxe = nxn - 1;
    const int nrk = (nyn + 63) >> 6, nrj = (nxn + LC - 1) / LC, ntask = nrk * nrj;
    const int lane = threadIdx.x & 63;
    const int wpb = BLOCK >> 6;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nbx = max(1, (int)gridDim.x >> 3);
    const int t0 = (int)((long long)ntask * xcd / 8), t1 = (int)((long long)ntask * (xcd + 1) / 8);
    for (int task = t0 + lb * wpb + (threadIdx.x >> 6); task < t1; task += nbx * wpb) {
        const int rj = task / nrk, rk = task - rj * nrk;
        const int k = (rk << 6) + lane;
        const bool act = k < nyn;
        const int kc = min(k, nye), km = max(kc - 1, 0), kp = min(kc + 1, nye);
        const int j0 = rj * LC, j1 = min(j0 + LC, nxn);
        const int ek0 = min(max(kc - 1, 0), nye - 1), ek1 = min(kc, nye - 1);
        const bool ok0 = kc - 1 >= 0, ok1 = kc < nye;
        double2 u[3][3], un[3];
        Gen3 m[2][2], mn[2];
        auto load_col = [&](int jj, double2 (&col)[3]) {
            const int jc = min(max(jj, 0), nxe);
            col[0] = xf(jc * nyn + km); col[1] = xf(jc * nyn + kc); col[2] = xf(jc * nyn + kp);
        };
        auto load_el = [&](int ej, Gen3 (&g)[2]) {
            const bool okj = ej >= 0 && ej < nxe;
            const int ec = min(max(ej, 0), nxe - 1);
            const size_t e0 = (size_t)ec * nye + ek0, e1 = (size_t)ec * nye + ek1;
            Gen3 g0 = {M2[e0], M2[(size_t)nel + e0], M2[(size_t)2 * nel + e0]};
            Gen3 g1 = {M2[e1], M2[(size_t)nel + e1], M2[(size_t)2 * nel + e1]};
            const double2 z = make_double2(0., 0.);
            if (!(okj && ok0)) g0 = {z, z, z};
            if (!(okj && ok1)) g1 = {z, z, z};
            g[0] = g0; g[1] = g1;
        };
        load_col(j0 - 1, u[0]);
        load_col(j0, u[1]);
        load_el(j0 - 1, m[0]);
        load_col(j0 + 1, u[2]);
        load_el(j0, m[1]);
        for (int j = j0; j < j1; j++) {
            if (j + 1 < j1) { load_col(j + 2, un); load_el(j + 1, mn); }
            if (act) emit(j * nyn + k, stencil(u, m, tab), u[1][1]);
#pragma unroll
            for (int r = 0; r < 3; r++) { u[0][r] = u[1][r]; u[1][r] = u[2][r]; u[2][r] = un[r]; }
            m[0][0] = m[1][0]; m[0][1] = m[1][1]; m[1][0] = mn[0]; m[1][1] = mn[1];
        }
    }
}

template <int LC>
__global__ void __launch_bounds__(BLOCK)
k_smooth_marchp(int nxn, int nyn, int nel, const double2 *__restrict__ M2, const double *__restrict__ tab, const double2 *__restrict__ dinv,
                const double2 *__restrict__ b, const double2 *__restrict__ xin, double2 *__restrict__ xout, double omega)
{
    marchp<LC>(nxn, nyn, nel, M2, tab, [&](int n) { return xin[n]; }, [&](int i, double2 qv, double2 xi) {
        const double2 di = dinv[i], bi = b[i];
        xout[i] = make_double2(fma(omega * di.x, bi.x - qv.x, xi.x), fma(omega * di.y, bi.y - qv.y, xi.y));
    });
}

__global__ void __launch_bounds__(BLOCK)
k_smooth_base(int nxn, int nyn, int nel, const double2 *__restrict__ M2, const double *tab, const double2 *__restrict__ dinv,
              const double2 *__restrict__ b, const double2 *__restrict__ xin, double2 *__restrict__ xout, double omega)
{
    const int nb = gridDim.x, nnode = nxn * nyn;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        const double2 di = dinv[i], bi = b[i];
        const double2 qv = grid_apply_pairs(nxn, nyn, nel, tab, i, [&](int q) { return M2[q]; }, [&](int j) { return xin[j]; });
        const double2 xi = xin[i];
        xout[i] = make_double2(fma(omega * di.x, bi.x - qv.x, xi.x), fma(omega * di.y, bi.y - qv.y, xi.y));
    }
}

template <int LC>
__global__ void __launch_bounds__(BLOCK)
k_smooth_march(int nxn, int nyn, int nel, const double2 *__restrict__ M2, const double *__restrict__ tab, const double2 *__restrict__ dinv,
               const double2 *__restrict__ b, const double2 *__restrict__ xin, double2 *__restrict__ xout, double omega)
{
    march<LC>(nxn, nyn, nel, M2, tab, [&](int n) { return xin[n]; }, [&](int i, double2 qv, double2 xi) {
        const double2 di = dinv[i], bi = b[i];
        xout[i] = make_double2(fma(omega * di.x, bi.x - qv.x, xi.x), fma(omega * di.y, bi.y - qv.y, xi.y));
    });
}

__global__ void __launch_bounds__(BLOCK)
k_spmv_base(int nxn, int nyn, int nel, const double2 *__restrict__ M2, const double *tab, const double2 *__restrict__ p,
            const double2 *__restrict__ z, double2 *__restrict__ pnew, double2 *__restrict__ q, double beta, double *__restrict__ part)
{
    __shared__ double sh[BLOCK / 64];
    const int nb = gridDim.x, nnode = nxn * nyn;
    double acc = 0.;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        const double2 qv = grid_apply_pairs(nxn, nyn, nel, tab, i, [&](int qq) { return M2[qq]; }, [&](int j) {
            const double2 zj = z[j], po = p[j];
            return make_double2(fma(beta, po.x, zj.x), fma(beta, po.y, zj.y));
        });
        q[i] = qv;
        const double2 zi = z[i], po = p[i];
        const double2 pn = make_double2(fma(beta, po.x, zi.x), fma(beta, po.y, zi.y));
        pnew[i] = pn;
        acc = fma(pn.x, qv.x, fma(pn.y, qv.y, acc));
    }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

template <int LC>
__global__ void __launch_bounds__(BLOCK)
k_spmv_march(int nxn, int nyn, int nel, const double2 *__restrict__ M2, const double *__restrict__ tab, const double2 *__restrict__ p,
             const double2 *__restrict__ z, double2 *__restrict__ pnew, double2 *__restrict__ q, double beta, double *__restrict__ part)
{
    __shared__ double sh[BLOCK / 64];
    double acc = 0.;
    march<LC>(nxn, nyn, nel, M2, tab, [&](int n) {
        const double2 zj = z[n], po = p[n];
        return make_double2(fma(beta, po.x, zj.x), fma(beta, po.y, zj.y));
    }, [&](int i, double2 qv, double2 pn) {
        q[i] = qv;
        pnew[i] = pn;
        acc = fma(pn.x, qv.x, fma(pn.y, qv.y, acc));
    });
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

template <class L>
static float best_of(L launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 25; rep++) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 5 && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main(int argc, char **argv)
{
    std::vector<int> sizes = {1024, 2048};
    if (argc > 1) sizes = {atoi(argv[1])};
    for (int nx : sizes) {
        const int ny = nx, nxn = nx + 1, nyn = ny + 1, nel = nx * ny, nn = nxn * nyn;
        std::vector<double> hM2(6 * (size_t)nel), htab(64), hv(2 * (size_t)nn), hw(2 * (size_t)nn);
        for (size_t e = 0; e < (size_t)nel; e++)
            for (int c = 0; c < 6; c++) {
                const double v = 1e5 * (1. + 0.3 * ((e * 7 + c * 13) % 11) / 11.) * (c == 1 || c == 2 || c == 4 ? 0.3 : 1.);
                hM2[((size_t)(c >> 1) * nel + e) * 2 + (c & 1)] = v;
            }
        for (int i = 0; i < 64; i++) htab[i] = 0.1 * ((i * 5) % 7 - 3);
        for (size_t i = 0; i < hv.size(); i++) { hv[i] = 1e-3 * ((i * 31) % 17 - 8); hw[i] = 1e-3 * ((i * 17) % 23 - 11); }
        double *M2, *tab, *dinv, *b, *x0, *x1, *x2, *q1, *q2, *part;
        (void)hipMalloc(&M2, hM2.size() * 8); (void)hipMalloc(&tab, 64 * 8); (void)hipMalloc(&part, 8 * 65536);
        for (double **pp : {&dinv, &b, &x0, &x1, &x2, &q1, &q2}) (void)hipMalloc(pp, hv.size() * 8);
        (void)hipMemcpy(M2, hM2.data(), hM2.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(tab, htab.data(), 64 * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(dinv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(b, hw.data(), hv.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(x0, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
        const double mb = (64. * nn + 48. * nel) / 1e6;
        std::vector<double> r0(hv.size()), r1(hv.size());
        auto diff = [&](double *a, double *bb) {
            (void)hipMemcpy(r0.data(), a, hv.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(r1.data(), bb, hv.size() * 8, hipMemcpyDeviceToHost);
            double d = 0.; for (size_t i = 0; i < r0.size(); i++) d = fmax(d, fabs(r0[i] - r1[i]));
            return d;
        };
#define ARGS_S nxn, nyn, nel, (const double2 *)M2, tab, (const double2 *)dinv, (const double2 *)b, (const double2 *)x0, (double2 *)
#define ARGS_P nxn, nyn, nel, (const double2 *)M2, tab, (const double2 *)x0, (const double2 *)b, (double2 *)
        const float tA = best_of([&] { k_smooth_base<<<1024, BLOCK>>>(ARGS_S x1, 0.65); });
        const float tC = best_of([&] { k_spmv_base<<<1024, BLOCK>>>(ARGS_P x1, (double2 *)q1, 0.37, part); });
        printf("%d^2 (%.1f MB algorithmic): smoother gather %.2f us (%.0f GB/s) | PCG spmv gather %.2f us (%.0f GB/s)\n", nx, mb, tA, mb / tA * 1e3, tC, mb / tC * 1e3);
        for (int grid : {512, 1024, 2048}) {
            k_smooth_base<<<1024, BLOCK>>>(ARGS_S x1, 0.65);
            const float t4 = best_of([&] { k_smooth_march<2><<<grid, BLOCK>>>(ARGS_S x2, 0.65); });
            const double d4 = diff(x1, x2);
            const float t8 = best_of([&] { k_smooth_march<8><<<grid, BLOCK>>>(ARGS_S x2, 0.65); });
            const double d8 = diff(x1, x2);
            const float t16 = best_of([&] { k_smooth_march<16><<<grid, BLOCK>>>(ARGS_S x2, 0.65); });
            const double d16 = diff(x1, x2);
            const float t32 = best_of([&] { k_smooth_march<4><<<grid, BLOCK>>>(ARGS_S x2, 0.65); });
            const float p8 = best_of([&] { k_smooth_marchp<8><<<grid, BLOCK>>>(ARGS_S x2, 0.65); });
            const double dp8 = diff(x1, x2);
            const float p16 = best_of([&] { k_smooth_marchp<16><<<grid, BLOCK>>>(ARGS_S x2, 0.65); });
            printf("   grid %5d smoother march + prefetch LC=8 %.2f us (diff %.1e) | LC=16 %.2f\n", grid, p8, dp8, p16);
            printf("   grid %5d smoother march LC=2 %.2f us (diff %.1e) | LC=8 %.2f (%.1e) | LC=16 %.2f (%.1e) | LC=4 %.2f\n", grid, t4, d4, t8, d8, t16, d16, t32);
            k_spmv_base<<<1024, BLOCK>>>(ARGS_P x1, (double2 *)q1, 0.37, part);
            const float s4 = best_of([&] { k_spmv_march<2><<<grid, BLOCK>>>(ARGS_P x2, (double2 *)q2, 0.37, part); });
            const double e4 = diff(q1, q2), e4p = diff(x1, x2);
            const float s8 = best_of([&] { k_spmv_march<8><<<grid, BLOCK>>>(ARGS_P x2, (double2 *)q2, 0.37, part); });
            const float s16 = best_of([&] { k_spmv_march<16><<<grid, BLOCK>>>(ARGS_P x2, (double2 *)q2, 0.37, part); });
            const double e16 = diff(q1, q2);
            const float s32 = best_of([&] { k_spmv_march<4><<<grid, BLOCK>>>(ARGS_P x2, (double2 *)q2, 0.37, part); });
            printf("   grid %5d PCG spmv march LC=2 %.2f us (diff q %.1e p %.1e) | LC=8 %.2f | LC=16 %.2f (%.1e) | LC=4 %.2f\n", grid, s4, e4, e4p, s8, s16, e16, s32);
        }
        for (double *pp : {M2, tab, part, dinv, b, x0, x1, x2, q1, q2}) (void)hipFree(pp);
    }
    return 0;
}
