// Fine-level damped-Jacobi sweep of the V-cycle (k_mg_smooth<1,1>) with the element stiffness generators stored as
// SoA [6][nel] doubles (the library's layout: 24 eight-byte loads per node) against [3][nel] double2 pairs (12 sixteen-byte
// loads per node): does halving the number of load instructions help a kernel that is bound by the loads a wave can keep
// in flight?   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pylabfea_amd/csrc -o tools/probes/pair_probe tools/probes/pair_probe.hip
#include "plfx_mg.hpp"
#include <cstdio>
#include <vector>
using namespace plfx;

template <class MF2, class XF>
__device__ __forceinline__ double2 grid_apply_pairs(int nxn, int nyn, int nel, const double *tab, int i, MF2 mf2, XF xf)
{
    const int nye = nyn - 1, nxe = nxn - 1;
    const int j = i / nyn, k = i - j * nyn;
    double2 u[3][3];
#pragma unroll
    for (int dj = 0; dj < 3; dj++) {
        const int jj = min(max(j + dj - 1, 0), nxe);
#pragma unroll
        for (int dk = 0; dk < 3; dk++) {
            const int kk = min(max(k + dk - 1, 0), nye);
            u[dj][dk] = xf(jj * nyn + kk);
        }
    }
    double m[4][6];
#pragma unroll
    for (int pj = 0; pj < 2; pj++)
#pragma unroll
        for (int pk = 0; pk < 2; pk++) {
            const int ej = j - 1 + pj, ek = k - 1 + pk;
            const bool ok = ej >= 0 && ej < nxe && ek >= 0 && ek < nye;
            const int e = min(max(ej, 0), nxe - 1) * nye + min(max(ek, 0), nye - 1);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double2 v = mf2(c * nel + e);
                m[pj * 2 + pk][2 * c] = ok ? v.x : 0.;
                m[pj * 2 + pk][2 * c + 1] = ok ? v.y : 0.;
            }
        }
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
            const double Mxx = m[p][0], Mxy = m[p][1], Mxs = m[p][2], Myy = m[p][3], Mys = m[p][4], Mss = m[p][5];
            qx = fma(Mxx, A1, fma(Mxs, A5 + A7 + A2, fma(Mss, A3 + A8, fma(Mxy, A6, fma(Mys, A4, qx)))));
            qy = fma(Mxy, A7, fma(Mys, A3 + A8 + A6, fma(Mxs, A1, fma(Mss, A5 + A2, fma(Myy, A4, qy)))));
        }
    return make_double2(qx, qy);
}

template <int PAIR>
__global__ void __launch_bounds__(BLOCK)
k_smooth(int nxn, int nyn, int nel, const double *__restrict__ M, const double2 *__restrict__ M2, const double *tab,
         const double2 *__restrict__ dinv, const double2 *__restrict__ b, const double2 *__restrict__ xin,
         double2 *__restrict__ xout, double omega)
{
    const int nb = gridDim.x, nnode = nxn * nyn;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        const double2 di = dinv[i], bi = b[i];
        double2 qv;
        if (PAIR)
            qv = grid_apply_pairs(nxn, nyn, nel, tab, i, [&](int q) { return M2[q]; }, [&](int j) { return xin[j]; });
        else
            qv = grid_apply_g(nxn, nyn, nel, tab, i, [&](int q) { return M[q]; }, [&](int j) { return xin[j]; });
        const double2 xi = xin[i];
        xout[i] = make_double2(fma(omega * di.x, bi.x - qv.x, xi.x), fma(omega * di.y, bi.y - qv.y, xi.y));
    }
}

int main()
{
    const int nx = 1024, ny = 1024, nxn = nx + 1, nyn = ny + 1, nel = nx * ny, nn = nxn * nyn;
    std::vector<double> hM(6 * (size_t)nel), hM2(6 * (size_t)nel), htab(64), hv(2 * (size_t)nn);
    for (size_t e = 0; e < (size_t)nel; e++)
        for (int c = 0; c < 6; c++) {
            const double v = 1e5 * (1. + 0.3 * ((e * 7 + c * 13) % 11) / 11.) * (c == 1 || c == 2 || c == 4 ? 0.3 : 1.);
            hM[(size_t)c * nel + e] = v;
            hM2[((size_t)(c >> 1) * nel + e) * 2 + (c & 1)] = v;
        }
    for (int i = 0; i < 64; i++) htab[i] = 0.1 * ((i * 5) % 7 - 3);
    for (size_t i = 0; i < hv.size(); i++) hv[i] = 1e-3 * ((i * 31) % 17 - 8);
    double *M, *M2, *tab, *dinv, *b, *x0, *x1;
    hipMalloc(&M, hM.size() * 8); hipMalloc(&M2, hM2.size() * 8); hipMalloc(&tab, 64 * 8);
    hipMalloc(&dinv, hv.size() * 8); hipMalloc(&b, hv.size() * 8); hipMalloc(&x0, hv.size() * 8); hipMalloc(&x1, hv.size() * 8);
    hipMemcpy(M, hM.data(), hM.size() * 8, hipMemcpyHostToDevice); hipMemcpy(M2, hM2.data(), hM2.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(tab, htab.data(), 64 * 8, hipMemcpyHostToDevice);
    hipMemcpy(dinv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice); hipMemcpy(b, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(x0, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> r0(hv.size()), r1(hv.size());
    for (int grid : {1024, 2048, 4096}) {
        for (int pair = 0; pair < 2; pair++) {
            float best = 1e9f;
            for (int rep = 0; rep < 30; rep++) {
                hipEventRecord(e0);
                if (pair)
                    hipLaunchKernelGGL(k_smooth<1>, dim3(grid), dim3(BLOCK), 0, 0, nxn, nyn, nel, M, (const double2 *)M2, tab,
                                       (const double2 *)dinv, (const double2 *)b, (const double2 *)x0, (double2 *)x1, 0.65);
                else
                    hipLaunchKernelGGL(k_smooth<0>, dim3(grid), dim3(BLOCK), 0, 0, nxn, nyn, nel, M, (const double2 *)M2, tab,
                                       (const double2 *)dinv, (const double2 *)b, (const double2 *)x0, (double2 *)x1, 0.65);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 5 && ms < best) best = ms;
            }
            hipMemcpy(pair ? r1.data() : r0.data(), x1, hv.size() * 8, hipMemcpyDeviceToHost);
            printf("grid %5d  %s generators: %.2f us  (%.0f GB/s of 117.6 MB)\n", grid, pair ? "double2-pair" : "SoA double  ", best * 1e3,
                   117.57e6 / (best * 1e-3) / 1e9);
        }
        double d = 0.; for (size_t i = 0; i < r0.size(); i++) d = fmax(d, fabs(r0[i] - r1[i]));
        printf("      max |difference| %.3e\n", d);
    }
    return 0;
}
