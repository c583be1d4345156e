// VERDICT r4 item 3, measured before building it: the fine-level damped-Jacobi sweep of the V-cycle (k_mg_smooth<1,1>) with the
// iterate, the Jacobi scaling and the stiffness generators stored in FP32 (right-hand side FP64 as PCG hands it over, arithmetic
// in FP64 after the loads) against the library's all-FP64 form.  The FP64 kernels are bound by the loads a wave keeps in flight
// (the pair layout), not by bytes -- does halving the bytes at the same NUMBER of loads buy anything, and what
// does a layout with fewer loads (generators of an element as two float4) buy?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pylabfea_amd/csrc -o tools/probes/fp32_probe tools/probes/fp32_probe.hip
#include "plfx_mg.hpp"
#include <cstdio>
#include <vector>
using namespace plfx;

// generic stencil: u(node) -> double2 (converted), gen(e, m[6])
template <class GEN, class XF>
__device__ __forceinline__ double2 apply(int nxn, int nyn, const double *tab, int i, GEN gen, XF xf)
{
    const int nye = nyn - 1, nxe = nxn - 1;
    const int j = i / nyn, k = i - j * nyn;
    double2 u[3][3];
#pragma unroll
    for (int dj = 0; dj < 3; dj++) {
        const int jj = min(max(j + dj - 1, 0), nxe);
#pragma unroll
        for (int dk = 0; dk < 3; dk++) u[dj][dk] = xf(jj * nyn + min(max(k + dk - 1, 0), nye));
    }
    double m[4][6];
#pragma unroll
    for (int pj = 0; pj < 2; pj++)
#pragma unroll
        for (int pk = 0; pk < 2; pk++) {
            const int ej = j - 1 + pj, ek = k - 1 + pk;
            const bool ok = ej >= 0 && ej < nxe && ek >= 0 && ek < nye;
            const int e = min(max(ej, 0), nxe - 1) * nye + min(max(ek, 0), nye - 1);
            gen(e, m[pj * 2 + pk]);
            if (!ok)
#pragma unroll
                for (int c = 0; c < 6; c++) m[pj * 2 + pk][c] = 0.;
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

// MODE 0: all FP64, generator pairs (the library).  1: FP32 iterate / scaling / generator pairs (3 float2 per element).
// 2: FP32, generators of an element as two float4 (32 B per element, two loads)
template <int MODE>
__global__ void __launch_bounds__(BLOCK)
k_smooth(int nxn, int nyn, int nel, const double2 *__restrict__ M2, const float2 *__restrict__ F2, const float4 *__restrict__ F4,
         const double *tab, const double2 *__restrict__ dinv, const float2 *__restrict__ dinvf, const double2 *__restrict__ b,
         const double2 *__restrict__ xin, const float2 *__restrict__ xinf, double2 *__restrict__ xout, float2 *__restrict__ xoutf, double omega)
{
    const int nb = gridDim.x, nnode = nxn * nyn;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        const double2 bi = b[i];
        if (MODE == 0) {
            const double2 di = dinv[i];
            const double2 qv = apply(nxn, nyn, tab, i,
                                     [&](int e, double *m) {
#pragma unroll
                                         for (int c = 0; c < 3; c++) { const double2 v = M2[(size_t)c * nel + e]; m[2 * c] = v.x; m[2 * c + 1] = v.y; }
                                     },
                                     [&](int j) { return xin[j]; });
            const double2 xi = xin[i];
            xout[i] = make_double2(fma(omega * di.x, bi.x - qv.x, xi.x), fma(omega * di.y, bi.y - qv.y, xi.y));
        } else {
            const float2 df = dinvf[i];
            const double2 qv = apply(nxn, nyn, tab, i,
                                     [&](int e, double *m) {
                                         if (MODE == 1) {
#pragma unroll
                                             for (int c = 0; c < 3; c++) { const float2 v = F2[(size_t)c * nel + e]; m[2 * c] = v.x; m[2 * c + 1] = v.y; }
                                         } else {
                                             const float4 a = F4[2 * (size_t)e], bq = F4[2 * (size_t)e + 1];
                                             m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = bq.x; m[5] = bq.y;
                                         }
                                     },
                                     [&](int j) { const float2 v = xinf[j]; return make_double2(v.x, v.y); });
            const float2 xi = xinf[i];
            xoutf[i] = make_float2((float)fma(omega * df.x, bi.x - qv.x, (double)xi.x), (float)fma(omega * df.y, bi.y - qv.y, (double)xi.y));
        }
    }
}

int main(int argc, char **argv)
{
    for (int n : {1024, 2048}) {
        const int nx = n, ny = n, nxn = nx + 1, nyn = ny + 1, nel = nx * ny, nn = nxn * nyn;
        std::vector<double> hM2(6 * (size_t)nel), htab(64), hv(2 * (size_t)nn);
        std::vector<float> hF2(6 * (size_t)nel), hF4(8 * (size_t)nel, 0.f), hvf(2 * (size_t)nn);
        for (size_t e = 0; e < (size_t)nel; e++)
            for (int c = 0; c < 6; c++) {
                const double v = 1e5 * (1. + 0.3 * ((e * 7 + c * 13) % 11) / 11.) * (c == 1 || c == 2 || c == 4 ? 0.3 : 1.);
                hM2[((size_t)(c >> 1) * nel + e) * 2 + (c & 1)] = v;
                hF2[((size_t)(c >> 1) * nel + e) * 2 + (c & 1)] = (float)v;
                hF4[8 * e + c] = (float)v;
            }
        for (int i = 0; i < 64; i++) htab[i] = 0.1 * ((i * 5) % 7 - 3);
        for (size_t i = 0; i < hv.size(); i++) hv[i] = 1e-3 * ((i * 31) % 17 - 8), hvf[i] = (float)hv[i];
        double *M2, *tab, *dinv, *b, *x0, *x1;
        float *F2, *F4, *dinvf, *x0f, *x1f;
        hipMalloc(&M2, hM2.size() * 8); hipMalloc(&tab, 64 * 8); hipMalloc(&F2, hF2.size() * 4); hipMalloc(&F4, hF4.size() * 4);
        hipMalloc(&dinv, hv.size() * 8); hipMalloc(&b, hv.size() * 8); hipMalloc(&x0, hv.size() * 8); hipMalloc(&x1, hv.size() * 8);
        hipMalloc(&dinvf, hv.size() * 4); hipMalloc(&x0f, hv.size() * 4); hipMalloc(&x1f, hv.size() * 4);
        hipMemcpy(M2, hM2.data(), hM2.size() * 8, hipMemcpyHostToDevice); hipMemcpy(tab, htab.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(F2, hF2.data(), hF2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(F4, hF4.data(), hF4.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dinv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice); hipMemcpy(b, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(x0, hv.data(), hv.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(dinvf, hvf.data(), hvf.size() * 4, hipMemcpyHostToDevice); hipMemcpy(x0f, hvf.data(), hvf.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const double mb[3] = {(64. * nn + 48. * nel) / 1e6, (40. * nn + 24. * nel) / 1e6, (40. * nn + 32. * nel) / 1e6};
        const char *nm[3] = {"FP64 (library): double2 vectors, 3 double2 generator pairs per element   ",
                             "FP32 storage: float2 iterate / scaling, FP64 rhs, 3 float2 pairs per element",
                             "FP32 storage, generators of an element as 2 float4 (32 B, two loads)        "};
        for (int grid : {1024, 2048}) {
            for (int mode = 0; mode < 3; mode++) {
                float best = 1e9f;
                for (int rep = 0; rep < 30; rep++) {
                    hipEventRecord(e0);
#define ARGS nxn, nyn, nel, (const double2 *)M2, (const float2 *)F2, (const float4 *)F4, tab, (const double2 *)dinv, (const float2 *)dinvf, \
             (const double2 *)b, (const double2 *)x0, (const float2 *)x0f, (double2 *)x1, (float2 *)x1f, 0.65
                    if (mode == 0) hipLaunchKernelGGL(k_smooth<0>, dim3(grid), dim3(BLOCK), 0, 0, ARGS);
                    if (mode == 1) hipLaunchKernelGGL(k_smooth<1>, dim3(grid), dim3(BLOCK), 0, 0, ARGS);
                    if (mode == 2) hipLaunchKernelGGL(k_smooth<2>, dim3(grid), dim3(BLOCK), 0, 0, ARGS);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep >= 5 && ms < best) best = ms;
                }
                printf("%d^2 grid %5d  %s %7.2f us  %6.1f MB  %5.0f GB/s\n", n, grid, nm[mode], best * 1e3, mb[mode], mb[mode] * 1e6 / (best * 1e-3) / 1e9);
            }
        }
        hipFree(M2); hipFree(tab); hipFree(F2); hipFree(F4); hipFree(dinv); hipFree(b); hipFree(x0); hipFree(x1); hipFree(dinvf); hipFree(x0f); hipFree(x1f);
    }
    return 0;
}
