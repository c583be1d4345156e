// Fine-level operator kernels with LDS-staged tiles (k_spmv<1,1> 0.50 of the HBM peak, bound by the
// loads a wave keeps in flight -- 18 gathered vector entries + 12 generator pairs per node).  Variants, 1024^2 grid:
//   A  k_mg_smooth<1,1> as built: every thread gathers its 9 vector entries and 12 generator pairs itself
//   B  the same sweep, a TJ x TK node tile per workgroup: the (TJ+2) x (TK+2) vector halo and the (TJ+1) x (TK+1) x 3
//      generator pairs are loaded ONCE into LDS (1.5 + 3.8 loads per node instead of 9 + 12), the stencil reads LDS
//   C  k_spmv<1,1> as built (PCG: p = z + beta p_old fused: two gathered vectors)
//   D  the same with the tile in LDS: p_new of the halo computed once per tile
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pylabfea_amd/csrc -o tools/probes/lds_tile_probe tools/probes/lds_tile_probe.hip
#include "plfx_mg.hpp"
#include <cstdio>
#include <vector>
using namespace plfx;

template <int TJ, int TK, class F>
__device__ __forceinline__ void tile_loop(int nxn, int nyn, F f)
{
    const int tkn = (nyn + TK - 1) / TK, tjn = (nxn + TJ - 1) / TJ;
    const int ntile = tkn * tjn;
    for (int t = xcd_tile(blockIdx.x, gridDim.x); t < ntile; t += gridDim.x) {
        const int tjq = t / tkn, tkq = t - tjq * tkn;
        f(tjq * TJ, tkq * TK);
    }
}

// stencil from LDS: sx[(TJ+2)][(TK+2)] vector halo, sm[3][(TJ+1)][(TK+1)] generator pairs (zero outside the grid)
template <int TJ, int TK>
__device__ __forceinline__ double2 apply_lds(const double2 *sx, const double2 *sm, const double *tab, int tj, int tk)
{
    constexpr int HK = TK + 2, EK = TK + 1, EN = (TJ + 1) * (TK + 1);
    double2 u[3][3];
#pragma unroll
    for (int dj = 0; dj < 3; dj++)
#pragma unroll
        for (int dk = 0; dk < 3; dk++) u[dj][dk] = sx[(tj + dj) * HK + tk + dk];
    double qx = 0., qy = 0.;
#pragma unroll
    for (int pj = 0; pj < 2; pj++)
#pragma unroll
        for (int pk = 0; pk < 2; pk++) {
            const int p = pj * 2 + pk;
            const int le = (tj + pj) * EK + tk + pk;
            const double2 m01 = sm[le], m23 = sm[EN + le], m45 = sm[2 * EN + le];
            const double *T = tab + p * 16;
            double A1 = 0., A2 = 0., A3 = 0., A4 = 0., A5 = 0., A6 = 0., A7 = 0., A8 = 0.;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const double2 ub = u[pj + (b >> 1)][pk + (b & 1)];
                const double sxx = T[b * 4 + 0], syy = T[b * 4 + 1], sxy = T[b * 4 + 2], syx = T[b * 4 + 3];
                A1 = fma(sxx, ub.x, A1); A2 = fma(sxx, ub.y, A2); A3 = fma(syy, ub.x, A3); A4 = fma(syy, ub.y, A4);
                A5 = fma(sxy, ub.x, A5); A6 = fma(sxy, ub.y, A6); A7 = fma(syx, ub.x, A7); A8 = fma(syx, ub.y, A8);
            }
            const double Mxx = m01.x, Mxy = m01.y, Mxs = m23.x, Myy = m23.y, Mys = m45.x, Mss = m45.y;
            qx = fma(Mxx, A1, fma(Mxs, A5 + A7 + A2, fma(Mss, A3 + A8, fma(Mxy, A6, fma(Mys, A4, qx)))));
            qy = fma(Mxy, A7, fma(Mys, A3 + A8 + A6, fma(Mxs, A1, fma(Mss, A5 + A2, fma(Myy, A4, qy)))));
        }
    return make_double2(qx, qy);
}

template <int TJ, int TK, class XF>
__device__ __forceinline__ void stage_tile(int nxn, int nyn, int nel, const double2 *__restrict__ M2, int j0, int k0,
                                           double2 *sx, double2 *sm, XF xf)
{
    constexpr int HK = TK + 2, HN = (TJ + 2) * HK, EK = TK + 1, EN = (TJ + 1) * EK;
    const int nye = nyn - 1, nxe = nxn - 1;
    for (int idx = threadIdx.x; idx < HN; idx += BLOCK) {
        const int hj = idx / HK, hk = idx - hj * HK;
        const int jj = min(max(j0 - 1 + hj, 0), nxe), kk = min(max(k0 - 1 + hk, 0), nye);
        sx[idx] = xf(jj * nyn + kk);
    }
    for (int idx = threadIdx.x; idx < 3 * EN; idx += BLOCK) {
        const int c = idx / EN, r = idx - c * EN;
        const int ej = r / EK, ek = r - ej * EK;
        const int gj = j0 - 1 + ej, gk = k0 - 1 + ek;
        const bool ok = gj >= 0 && gj < nxe && gk >= 0 && gk < nye;
        const int e = min(max(gj, 0), nxe - 1) * nye + min(max(gk, 0), nye - 1);
        const double2 v = M2[(size_t)c * nel + e];
        sm[idx] = ok ? v : make_double2(0., 0.);
    }
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

template <int TJ, int TK>
__global__ void __launch_bounds__(BLOCK)
k_smooth_lds(int nxn, int nyn, int nel, const double2 *__restrict__ M2, const double *tab, const double2 *__restrict__ dinv,
             const double2 *__restrict__ b, const double2 *__restrict__ xin, double2 *__restrict__ xout, double omega)
{
    static_assert(TJ * TK == BLOCK, "one node per thread");
    __shared__ double2 sx[(TJ + 2) * (TK + 2)];
    __shared__ double2 sm[3 * (TJ + 1) * (TK + 1)];
    const int tj = threadIdx.x / TK, tk = threadIdx.x - tj * TK;
    tile_loop<TJ, TK>(nxn, nyn, [&](int j0, int k0) {
        __syncthreads();
        stage_tile<TJ, TK>(nxn, nyn, nel, M2, j0, k0, sx, sm, [&](int j) { return xin[j]; });
        __syncthreads();
        const int j = j0 + tj, k = k0 + tk;
        if (j < nxn && k < nyn) {
            const int i = j * nyn + k;
            const double2 di = dinv[i], bi = b[i];
            const double2 qv = apply_lds<TJ, TK>(sx, sm, tab, tj, tk);
            const double2 xi = sx[(tj + 1) * (TK + 2) + tk + 1];
            xout[i] = make_double2(fma(omega * di.x, bi.x - qv.x, xi.x), fma(omega * di.y, bi.y - qv.y, xi.y));
        }
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

template <int TJ, int TK>
__global__ void __launch_bounds__(BLOCK)
k_spmv_lds(int nxn, int nyn, int nel, const double2 *__restrict__ M2, const double *tab, const double2 *__restrict__ p,
           const double2 *__restrict__ z, double2 *__restrict__ pnew, double2 *__restrict__ q, double beta, double *__restrict__ part)
{
    __shared__ double2 sx[(TJ + 2) * (TK + 2)];
    __shared__ double2 sm[3 * (TJ + 1) * (TK + 1)];
    __shared__ double sh[BLOCK / 64];
    const int tj = threadIdx.x / TK, tk = threadIdx.x - tj * TK;
    double acc = 0.;
    tile_loop<TJ, TK>(nxn, nyn, [&](int j0, int k0) {
        __syncthreads();
        stage_tile<TJ, TK>(nxn, nyn, nel, M2, j0, k0, sx, sm, [&](int j) {
            const double2 zj = z[j], po = p[j];
            return make_double2(fma(beta, po.x, zj.x), fma(beta, po.y, zj.y));
        });
        __syncthreads();
        const int j = j0 + tj, k = k0 + tk;
        if (j < nxn && k < nyn) {
            const int i = j * nyn + k;
            const double2 qv = apply_lds<TJ, TK>(sx, sm, tab, tj, tk);
            const double2 pn = sx[(tj + 1) * (TK + 2) + tk + 1];
            q[i] = qv;
            pnew[i] = pn;
            acc = fma(pn.x, qv.x, fma(pn.y, qv.y, acc));
        }
    });
    __syncthreads();
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

template <class L>
static float best_of(L launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 30; rep++) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 5 && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 1024, ny = nx, nxn = nx + 1, nyn = ny + 1, nel = nx * ny, nn = nxn * nyn;
    std::vector<double> hM2(6 * (size_t)nel), htab(64), hv(2 * (size_t)nn), hw(2 * (size_t)nn);
    for (size_t e = 0; e < (size_t)nel; e++)
        for (int c = 0; c < 6; c++) {
            const double v = 1e5 * (1. + 0.3 * ((e * 7 + c * 13) % 11) / 11.) * (c == 1 || c == 2 || c == 4 ? 0.3 : 1.);
            hM2[((size_t)(c >> 1) * nel + e) * 2 + (c & 1)] = v;
        }
    for (int i = 0; i < 64; i++) htab[i] = 0.1 * ((i * 5) % 7 - 3);
    for (size_t i = 0; i < hv.size(); i++) { hv[i] = 1e-3 * ((i * 31) % 17 - 8); hw[i] = 1e-3 * ((i * 17) % 23 - 11); }
    double *M2, *tab, *dinv, *b, *x0, *x1, *x2, *q1, *q2, *part;
    hipMalloc(&M2, hM2.size() * 8); hipMalloc(&tab, 64 * 8); hipMalloc(&part, 8 * 8192);
    for (double **pp : {&dinv, &b, &x0, &x1, &x2, &q1, &q2}) hipMalloc(pp, hv.size() * 8);
    hipMemcpy(M2, hM2.data(), hM2.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(tab, htab.data(), 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dinv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice); hipMemcpy(b, hw.data(), hv.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(x0, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
    const double mb = (64. * nn + 48. * nel) / 1e6;
    std::vector<double> r0(hv.size()), r1(hv.size());
    auto diff = [&](double *a, double *bb) {
        hipMemcpy(r0.data(), a, hv.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), bb, hv.size() * 8, hipMemcpyDeviceToHost);
        double d = 0.; for (size_t i = 0; i < r0.size(); i++) d = fmax(d, fabs(r0[i] - r1[i]));
        return d;
    };
#define SM(name, ...) name<<<dim3(grid), dim3(BLOCK), 0, 0>>>(nxn, nyn, nel, (const double2 *)M2, tab, __VA_ARGS__)
    for (int grid : {1024, 2048, 4096}) {
        const float tA = best_of([&] { SM(k_smooth_base, (const double2 *)dinv, (const double2 *)b, (const double2 *)x0, (double2 *)x1, 0.65); });
        const float tB1 = best_of([&] { SM((k_smooth_lds<4, 64>), (const double2 *)dinv, (const double2 *)b, (const double2 *)x0, (double2 *)x2, 0.65); });
        const double d1 = diff(x1, x2);
        const float tB2 = best_of([&] { SM((k_smooth_lds<2, 128>), (const double2 *)dinv, (const double2 *)b, (const double2 *)x0, (double2 *)x2, 0.65); });
        const double d2 = diff(x1, x2);
        const float tB3 = best_of([&] { SM((k_smooth_lds<8, 32>), (const double2 *)dinv, (const double2 *)b, (const double2 *)x0, (double2 *)x2, 0.65); });
        const double d3 = diff(x1, x2);
        printf("grid %5d smoother: A gather %.2f us (%.0f GB/s) | B LDS tile 4x64 %.2f us (%.0f GB/s, diff %.1e) | 2x128 %.2f us (diff %.1e) | 8x32 %.2f us (diff %.1e)\n",
               grid, tA, mb / tA * 1e3, tB1, mb / tB1 * 1e3, d1, tB2, d2, tB3, d3);
        const float tC = best_of([&] { SM(k_spmv_base, (const double2 *)x0, (const double2 *)b, (double2 *)x1, (double2 *)q1, 0.37, part); });
        const float tD1 = best_of([&] { SM((k_spmv_lds<4, 64>), (const double2 *)x0, (const double2 *)b, (double2 *)x2, (double2 *)q2, 0.37, part); });
        const double e1 = diff(q1, q2), e1p = diff(x1, x2);
        const float tD2 = best_of([&] { SM((k_spmv_lds<2, 128>), (const double2 *)x0, (const double2 *)b, (double2 *)x2, (double2 *)q2, 0.37, part); });
        const double e2 = diff(q1, q2);
        printf("grid %5d PCG spmv: C gather %.2f us (%.0f GB/s) | D LDS tile 4x64 %.2f us (%.0f GB/s, diff q %.1e p %.1e) | 2x128 %.2f us (diff %.1e)\n",
               grid, tC, mb / tC * 1e3, tD1, mb / tD1 * 1e3, e1, e1p, tD2, e2);
    }
    return 0;
}
