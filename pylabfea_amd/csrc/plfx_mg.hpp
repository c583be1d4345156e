// plfx_mg.hpp — geometric multigrid V-cycle used as the PCG preconditioner on structured Q4 grids.
//
// The reference solves K du = f with dense LU (model.py:1028-1033, 1291); any solver that returns the
// same du to round-off is a legal replacement.  Jacobi-PCG needs O(NX) iterations; a V(2,2) cycle on
// the node grid (node id = j*(NY+1)+k, model.py:893) makes the count mesh-independent.
//   * transfer: bilinear interpolation P per displacement component, restriction R = P^T
//   * coarse operators: re-assembly with the arithmetic mean of the four children's stiffness
//     generators M (Q4 stiffness is invariant under uniform scaling of the element, so the class
//     tables of the fine grid apply on every level)
//   * smoother: damped Jacobi, same number of pre- and post-sweeps  -> symmetric, SPD preconditioner
//   * Dirichlet DOFs: a coarse DOF is prescribed iff its coincident fine DOF is; corrections and
//     residuals are kept zero on prescribed DOFs on every level
//   * coarsest grid: Jacobi-PCG inside one workgroup (vectors in LDS)
#pragma once
#include "plfx_kernels.hpp"

namespace plfx {

// The last post-smoothing launch of the finest level writes z = M^-1 r (x aliases z, b aliases r there): it can deliver the
// per-block partial sums of r.z that PCG needs next -- b[i] and the new x[i] are in registers -- and saves the separate
// k_dot_rz pass over both vectors (one launch and 32 B per node per PCG iteration).  part == nullptr: no sums.
struct DotOut {
    double *part = nullptr;   // [nslots] partial sums, one per block; slots >= gridDim.x are zeroed by block 0
    int nslots = 0;
    int own_lo = 0, own_hi = 0;   // nodes summed (all; the owned columns of a strip)
};

__device__ __forceinline__ void dot_finish(const DotOut &dot, double acc, double *sh)
{
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) dot.part[blockIdx.x] = t;
    if (blockIdx.x == 0)
        for (int k = gridDim.x + threadIdx.x; k < dot.nslots; k += BLOCK) dot.part[k] = 0.;
}

// xout = xin + omega * dinv * (b - K xin)      (first != 0: xin == 0 -> xout = omega * dinv * b)
// FINE = 1 instantiations run the finest grid only, so that profilers list the HBM-bound fine-level
// launches (the dominant kernels of a load step) separately from the latency-bound coarse ones.
template <int FINE, int GRID>
__global__ void __launch_bounds__(BLOCK)
k_mg_smooth(KOp op,
            const double2 *__restrict__ dinv, const double2 *__restrict__ b,
            const double2 *__restrict__ xin, double2 *__restrict__ xout, double omega, int first,
            const CgScalars *sc, DotOut dot = DotOut{})
{
    // PCG already converged: the launches of a speculatively enqueued cycle head are no-ops.  Only the finest level is ever
    // enqueued speculatively (mg_vcycle_head); the coarser levels run after the host has seen "not converged", and there
    // the flag would be one more dependent scalar load in front of a launch-latency-bound kernel
    if (FINE && sc->done) return;
    __shared__ double sh[BLOCK / 64];
    double acc = 0.;
    const int nb = gridDim.x, nnode = op.nnode;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        const double2 di = dinv[i], bi = b[i];
        if (first) {
            xout[i] = make_double2(omega * di.x * bi.x, omega * di.y * bi.y);
            continue;
        }
        const double2 qv = op_apply<GRID>(op, i, [&](int j) { return xin[j]; });
        const double qx = qv.x, qy = qv.y;
        const double2 xi = xin[i];
        const double2 xo = make_double2(fma(omega * di.x, bi.x - qx, xi.x), fma(omega * di.y, bi.y - qy, xi.y));
        xout[i] = xo;
        if (dot.part && i >= dot.own_lo && i < dot.own_hi) acc = fma(bi.x, xo.x, fma(bi.y, xo.y, acc));
    }
    if (dot.part) dot_finish(dot, acc, sh);
}

// two damped-Jacobi sweeps from a zero guess in one pass:
//   x1 = w D^-1 b ;  x2 = x1 + w D^-1 (b - K x1)   with x1 of the neighbours recomputed from (dinv, b)
template <int FINE, int GRID>
__global__ void __launch_bounds__(BLOCK)
k_mg_smooth2_zero(KOp op,
                  const double2 *__restrict__ dinv, const double2 *__restrict__ b,
                  double2 *__restrict__ xout, double omega, const CgScalars *sc)
{
    if (FINE && sc->done) return;
    const int nb = gridDim.x, nnode = op.nnode;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        const double2 qv = op_apply<GRID>(op, i, [&](int j) {
            const double2 dj = dinv[j], bj = b[j];
            return make_double2(omega * dj.x * bj.x, omega * dj.y * bj.y);
        });
        const double qx = qv.x, qy = qv.y;
        const double2 di = dinv[i], bi = b[i];
        const double x1x = omega * di.x * bi.x, x1y = omega * di.y * bi.y;
        xout[i] = make_double2(fma(omega * di.x, bi.x - qx, x1x), fma(omega * di.y, bi.y - qy, x1y));
    }
}

// res = P_free (b - K x)
template <int FINE, int GRID>
__global__ void __launch_bounds__(BLOCK)
k_mg_residual(KOp op,
              const double2 *__restrict__ dinv, const double2 *__restrict__ b,
              const double2 *__restrict__ x, double2 *__restrict__ res, const CgScalars *sc)
{
    if (FINE && sc->done) return;
    const int nb = gridDim.x, nnode = op.nnode;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        const double2 qv = op_apply<GRID>(op, i, [&](int j) { return x[j]; });
        const double qx = qv.x, qy = qv.y;
        const double2 di = dinv[i], bi = b[i];
        res[i] = make_double2(di.x != 0. ? bi.x - qx : 0., di.y != 0. ? bi.y - qy : 0.);
    }
}

// The three fine-level kernels of the V-cycle in marching form (grid_march, plfx_kernels.hpp): bit-identical results, fewer
// and fully coalesced loads; used on the finest grid when one operator pass exceeds the Infinity Cache (plfx.hip: use_march)
__global__ void __launch_bounds__(BLOCK)
k_mg_smooth_march(KOp op, const double2 *__restrict__ dinv, const double2 *__restrict__ b, const double2 *__restrict__ xin,
                  double2 *__restrict__ xout, double omega, const CgScalars *sc, DotOut dot)
{
    if (sc->done) return;
    __shared__ double sh[BLOCK / 64];
    double acc = 0.;
    grid_march<MARCH_LC>(
        op, [&](int n) { return xin[n]; },
        [&](int i, double2 qv, double2 xi) {
            const double2 di = dinv[i], bi = b[i];
            const double2 xo = make_double2(fma(omega * di.x, bi.x - qv.x, xi.x), fma(omega * di.y, bi.y - qv.y, xi.y));
            xout[i] = xo;
            if (dot.part && i >= dot.own_lo && i < dot.own_hi) acc = fma(bi.x, xo.x, fma(bi.y, xo.y, acc));
        });
    if (dot.part) dot_finish(dot, acc, sh);
}

__global__ void __launch_bounds__(BLOCK)
k_mg_smooth2_zero_march(KOp op, const double2 *__restrict__ dinv, const double2 *__restrict__ b, double2 *__restrict__ xout,
                        double omega, const CgScalars *sc)
{
    if (sc->done) return;
    grid_march<MARCH_LC>(
        op,
        [&](int n) {
            const double2 dj = dinv[n], bj = b[n];
            return make_double2(omega * dj.x * bj.x, omega * dj.y * bj.y);
        },
        [&](int i, double2 qv, double2 x1) {
            const double2 di = dinv[i], bi = b[i];
            xout[i] = make_double2(fma(omega * di.x, bi.x - qv.x, x1.x), fma(omega * di.y, bi.y - qv.y, x1.y));
        });
}

__global__ void __launch_bounds__(BLOCK)
k_mg_residual_march(KOp op, const double2 *__restrict__ dinv, const double2 *__restrict__ b, const double2 *__restrict__ x,
                    double2 *__restrict__ res, const CgScalars *sc)
{
    if (sc->done) return;
    grid_march<MARCH_LC>(
        op, [&](int n) { return x[n]; },
        [&](int i, double2 qv, double2) {
            const double2 di = dinv[i], bi = b[i];
            res[i] = make_double2(di.x != 0. ? bi.x - qv.x : 0., di.y != 0. ? bi.y - qv.y : 0.);
        });
}

// One step of the Jacobi-preconditioned Chebyshev iteration for K x = b (coarsest level too large for a dense inverse):
//   d <- c1 d + c2 D^-1 (b - K x),  x <- x + d        (first != 0: x = 0, d = c2 D^-1 b)
// A fixed number of steps with fixed coefficients is a fixed polynomial in D^-1 K: a symmetric positive definite
// coarse solver, so the V-cycle stays a valid PCG preconditioner.
template <int GRID>
__global__ void __launch_bounds__(BLOCK)
k_mg_cheby(KOp op, const double2 *__restrict__ dinv, const double2 *__restrict__ b, const double2 *__restrict__ xin,
           double2 *__restrict__ xout, double2 *__restrict__ d, double c1, double c2, int first, const CgScalars *sc)
{
    if (sc->done) return;
    const int nb = gridDim.x, nnode = op.nnode;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        const double2 di = dinv[i], bi = b[i];
        if (first) {
            const double2 dn = make_double2(c2 * di.x * bi.x, c2 * di.y * bi.y);
            d[i] = dn;
            xout[i] = dn;
            continue;
        }
        const double2 qv = op_apply<GRID>(op, i, [&](int j) { return xin[j]; });
        const double2 xi = xin[i], dold = d[i];
        const double2 dn = make_double2(fma(c1, dold.x, c2 * di.x * (bi.x - qv.x)), fma(c1, dold.y, c2 * di.y * (bi.y - qv.y)));
        d[i] = dn;
        xout[i] = make_double2(xi.x + dn.x, xi.y + dn.y);
    }
}

// transfer weight of fine node (jf, kf) towards coarse node (J, K) -- the 2-d product of mg_tr1d; levels that halve exactly:
// 1, 1/2, 1/4
__device__ __forceinline__ double mg_tr_weight(int jf, int kf, int J, int K, int nfx, int nfy, double rxf, double ryf)
{
    int J0, K0;
    double a0, a1, b0, b1;
    mg_tr1d(jf, nfx, rxf, J0, a0, a1);
    mg_tr1d(kf, nfy, ryf, K0, b0, b1);
    const double wj = (J0 == J) ? a0 : (J0 + 1 == J ? a1 : 0.);
    const double wk = (K0 == K) ? b0 : (K0 + 1 == K ? b1 : 0.);
    return wj * wk;
}

// value of the coarse vector xc (nyc nodes per column) interpolated at fine node (j, k)
template <class XC>
__device__ __forceinline__ double2 mg_interpolate(int j, int k, int nfx, int nfy, double rxf, double ryf, int nyc, XC xc)
{
    int J0, K0;
    double a0, a1, b0, b1;
    mg_tr1d(j, nfx, rxf, J0, a0, a1);
    mg_tr1d(k, nfy, ryf, K0, b0, b1);
    double2 v = xc(J0 * nyc + K0);
    double cx = a0 * b0 * v.x, cy = a0 * b0 * v.y;
    if (a1 != 0.) {
        v = xc((J0 + 1) * nyc + K0);
        cx = fma(a1 * b0, v.x, cx), cy = fma(a1 * b0, v.y, cy);
    }
    if (b1 != 0.) {
        v = xc(J0 * nyc + K0 + 1);
        cx = fma(a0 * b1, v.x, cx), cy = fma(a0 * b1, v.y, cy);
    }
    if (a1 != 0. && b1 != 0.) {
        v = xc((J0 + 1) * nyc + K0 + 1);
        cx = fma(a1 * b1, v.x, cx), cy = fma(a1 * b1, v.y, cy);
    }
    return make_double2(cx, cy);
}

// (P^T res)(J, K): res(jf, kf) -> double2 reads the fine residual.  plain: the level halves exactly (weights 1, 1/2, 1/4 of the
// nine fine nodes around (2J, 2K), summed in the order the kernels always used)
template <class RES>
__device__ __forceinline__ double2 mg_restrict_at(int J, int K, int nfx, int nfy, double rxf, double ryf, bool plain, RES res)
{
    double sx = 0., sy = 0.;
    if (plain) {
#pragma unroll
        for (int dj = -1; dj <= 1; dj++) {
            const int jf = 2 * J + dj;
            if (jf < 0 || jf > nfx) continue;
#pragma unroll
            for (int dk = -1; dk <= 1; dk++) {
                const int kf = 2 * K + dk;
                if (kf < 0 || kf > nfy) continue;
                const double w = (dj == 0 ? 1. : 0.5) * (dk == 0 ? 1. : 0.5);
                const double2 r = res(jf, kf);
                sx = fma(w, r.x, sx);
                sy = fma(w, r.y, sy);
            }
        }
    } else {
        const int jc = mg_fine_node(J, nfx, rxf), kc = mg_fine_node(K, nfy, ryf);   // fine nodes within two lines of the coincident one
        for (int jf = max(jc - 2, 0); jf <= min(jc + 2, nfx); jf++)
            for (int kf = max(kc - 2, 0); kf <= min(kc + 2, nfy); kf++) {
                const double w = mg_tr_weight(jf, kf, J, K, nfx, nfy, rxf, ryf);
                if (w == 0.) continue;
                const double2 r = res(jf, kf);
                sx = fma(w, r.x, sx);
                sy = fma(w, r.y, sy);
            }
    }
    return make_double2(sx, sy);
}

// restriction b_c = P^T res_f (P: bilinear interpolation, mg_tr1d per direction); nyf/nyc = nodes per column; rxf, ryf =
// relative size of the fine level's last cell
__global__ void __launch_bounds__(BLOCK)
k_mg_restrict(int nxc_nodes, int nyc, int nxf_nodes, int nyf, const double2 *__restrict__ res_f,
              const double2 *__restrict__ dinv_c, double2 *__restrict__ b_c, double rxf = 1., double ryf = 1.,
              const CgScalars *__restrict__ sc = nullptr /* finest level: the head of a cycle is enqueued behind the convergence test */)
{
    if (sc && sc->done) return;
    const int nc = nxc_nodes * nyc;
    const int nfx = nxf_nodes - 1, nfy = nyf - 1;   // cells of the fine level
    const bool plain = !(nfx & 1) && !(nfy & 1) && rxf == 1. && ryf == 1.;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nc; i += gridDim.x * BLOCK) {
        const int J = i / nyc, K = i - J * nyc;
        const double2 sr = mg_restrict_at(J, K, nfx, nfy, rxf, ryf, plain, [&](int jf, int kf) { return res_f[(size_t)jf * nyf + kf]; });
        const double sx = sr.x, sy = sr.y;
        const double2 d = dinv_c[i];
        b_c[i] = make_double2(d.x != 0. ? sx : 0., d.y != 0. ? sy : 0.);
    }
}

// x_f += P x_c on free fine DOFs
__global__ void __launch_bounds__(BLOCK)
k_mg_prolong_add(int nxf_nodes, int nyf, int nyc, const double2 *__restrict__ x_c,
                 const double2 *__restrict__ dinv_f, double2 *__restrict__ x_f, double rxf = 1., double ryf = 1.)
{
    const int nf = nxf_nodes * nyf;
    const int nfx = nxf_nodes - 1, nfy = nyf - 1;
    const bool plain = !(nfx & 1) && !(nfy & 1) && rxf == 1. && ryf == 1.;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nf; i += gridDim.x * BLOCK) {
        const int j = i / nyf, k = i - j * nyf;
        double cx, cy, w = 1.;
        if (plain) {
            const int J0 = j >> 1, K0 = k >> 1;
            const int oj = j & 1, ok = k & 1;
            double2 v = x_c[(size_t)J0 * nyc + K0];
            cx = v.x, cy = v.y;
            if (oj) {
                v = x_c[(size_t)(J0 + 1) * nyc + K0];
                cx += v.x;
                cy += v.y;
            }
            if (ok) {
                v = x_c[(size_t)J0 * nyc + K0 + 1];
                cx += v.x;
                cy += v.y;
            }
            if (oj && ok) {
                v = x_c[(size_t)(J0 + 1) * nyc + K0 + 1];
                cx += v.x;
                cy += v.y;
            }
            w = (oj ? 0.5 : 1.) * (ok ? 0.5 : 1.);
        } else {
            int J0, K0;
            double a0, a1, b0, b1;
            mg_tr1d(j, nfx, rxf, J0, a0, a1);
            mg_tr1d(k, nfy, ryf, K0, b0, b1);
            double2 v = x_c[(size_t)J0 * nyc + K0];
            cx = a0 * b0 * v.x, cy = a0 * b0 * v.y;
            if (a1 != 0.) {
                v = x_c[(size_t)(J0 + 1) * nyc + K0];
                cx = fma(a1 * b0, v.x, cx), cy = fma(a1 * b0, v.y, cy);
            }
            if (b1 != 0.) {
                v = x_c[(size_t)J0 * nyc + K0 + 1];
                cx = fma(a0 * b1, v.x, cx), cy = fma(a0 * b1, v.y, cy);
            }
            if (a1 != 0. && b1 != 0.) {
                v = x_c[(size_t)(J0 + 1) * nyc + K0 + 1];
                cx = fma(a1 * b1, v.x, cx), cy = fma(a1 * b1, v.y, cy);
            }
        }
        const double2 d = dinv_f[i];
        double2 xf = x_f[i];
        if (d.x != 0.) xf.x = fma(w, cx, xf.x);
        if (d.y != 0.) xf.y = fma(w, cy, xf.y);
        x_f[i] = xf;
    }
}

// coarse stiffness generator: mean of the children (element id = j*NY + k, model.py:935) -- four equal ones on levels that
// halve exactly; otherwise the children mg_coarse_cells gives the cell (one, two or three per direction), weighted with their areas
__global__ void __launch_bounds__(BLOCK)
k_mg_coarsen_M(int nxc, int nyc, int nyf, int nel_f, const double *__restrict__ M_f, double *__restrict__ M_c,
               int pair_f, int pair_c /* layouts of the two arrays (gen_index) */, double rxf = 1., double ryf = 1.)
{
    const int nel_c = nxc * nyc, nxf = nel_f / nyf;
    const bool plain = !(nxf & 1) && !(nyf & 1) && rxf == 1. && ryf == 1.;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nel_c; i += gridDim.x * BLOCK) {
        const int J = i / nyc, K = i - J * nyc;
        if (plain) {
            const size_t e00 = (size_t)(2 * J) * nyf + 2 * K, e10 = e00 + nyf;
#pragma unroll
            for (int c = 0; c < 6; c++)
                M_c[gen_index(pair_c, c, nel_c, i)] =
                    0.25 * (M_f[gen_index(pair_f, c, nel_f, e00)] + M_f[gen_index(pair_f, c, nel_f, e00 + 1)] +
                            M_f[gen_index(pair_f, c, nel_f, e10)] + M_f[gen_index(pair_f, c, nel_f, e10 + 1)]);
            continue;
        }
        // children columns [2J, jend) and rows [2K, kend) counted from the side of the regular cells: the coarse cell at the
        // other end takes what is left (mg_odd_cell)
        const int Jm = PLFX_MG_RAGGED_FIRST ? nxc - 1 - J : J, Km = PLFX_MG_RAGGED_FIRST ? nyc - 1 - K : K;
        const int jend = (Jm == nxc - 1) ? nxf : 2 * Jm + 2, kend = (Km == nyc - 1) ? nyf : 2 * Km + 2;
        double acc[6] = {0., 0., 0., 0., 0., 0.}, wsum = 0.;
        for (int jm = 2 * Jm; jm < jend; jm++)
            for (int km = 2 * Km; km < kend; km++) {
                const int jf = PLFX_MG_RAGGED_FIRST ? nxf - 1 - jm : jm, kf = PLFX_MG_RAGGED_FIRST ? nyf - 1 - km : km;
                const double w = ((jf == mg_odd_cell(nxf)) ? rxf : 1.) * ((kf == mg_odd_cell(nyf)) ? ryf : 1.);
                const size_t e = (size_t)jf * nyf + kf;
#pragma unroll
                for (int c = 0; c < 6; c++) acc[c] = fma(w, M_f[gen_index(pair_f, c, nel_f, e)], acc[c]);
                wsum += w;
            }
#pragma unroll
        for (int c = 0; c < 6; c++) M_c[gen_index(pair_c, c, nel_c, i)] = acc[c] / wsum;
    }
}

// coarse Dirichlet mask from the coincident fine nodes; dinv_c = free ? 1/|diag_c| : 0
__global__ void __launch_bounds__(BLOCK)
k_mg_coarse_dinv(int nxc_nodes, int nyc, int nyf, const double2 *__restrict__ dinv_f,
                 const double2 *__restrict__ diag_c, double2 *__restrict__ dinv_c, int nxf_nodes, double rxf = 1., double ryf = 1.)
{
    const int nc = nxc_nodes * nyc;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nc; i += gridDim.x * BLOCK) {
        const int J = i / nyc, K = i - J * nyc;
        // (the last coarse node line coincides with the last fine one, however the fine level is coarsened)
        const double2 df = dinv_f[(size_t)mg_fine_node(J, nxf_nodes - 1, rxf) * nyf + mg_fine_node(K, nyf - 1, ryf)];
        const double2 dg = diag_c[i];
        double2 o;
        o.x = (df.x != 0.) ? (fabs(dg.x) > 1e-300 ? 1. / fabs(dg.x) : 1.) : 0.;
        o.y = (df.y != 0.) ? (fabs(dg.y) > 1e-300 ? 1. / fabs(dg.y) : 1.) : 0.;
        dinv_c[i] = o;
    }
}

// Coarsest grid: Jacobi-PCG on (nnode <= MG_COARSE_MAX) nodes inside one workgroup, vectors in LDS.
// z = D^-1 r (the preconditioner of a strip's Jacobi fall-back; the V-cycle's slot in the PCG loop)
__global__ void __launch_bounds__(BLOCK)
k_jacobi_z(int nn, const double2 *__restrict__ dinv, const double2 *__restrict__ r, double2 *__restrict__ z, const CgScalars *sc)
{
    if (sc->done) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += gridDim.x * blockDim.x) {
        const double2 d = dinv[i], ri = r[i];
        z[i] = make_double2(d.x * ri.x, d.y * ri.y);
    }
}

// ---------------------------------------------------------------------------------------------- MINRES
// The tangents of the reference are not always positive semi-definite: the correction step of Material.response
// (material.py:317-338) subtracts a least-squares fit of the excess stress from the tangent, and single elements end up with
// negative diagonal entries.  The reference's dense LU does not care; PCG stops at the first direction of negative curvature.
// Such systems are solved by preconditioned MINRES (Paige & Saunders) with the same SPD preconditioner (V-cycle or D^-1).
// One iteration: k_minres_apply (v = s y, q = P K v, partials of v.q and v.r1), k_minres_update1 (three-term recurrence: new
// residual vector), preconditioner, k_dot_rz, k_minres_update2 (search direction and iterate); scalars on the host.
template <int GRID>
__global__ void __launch_bounds__(BLOCK)
k_minres_apply(KOp op, int nnode, double s, const double2 *__restrict__ y, const double2 *__restrict__ dinv,
               const double2 *__restrict__ r1, double2 *__restrict__ v, double2 *__restrict__ q, double *__restrict__ part,
               double *__restrict__ part_vr1, int own_lo, int own_hi)
{
    __shared__ double sh[BLOCK / 64];
    double acc = 0., acc1 = 0.;
    const int nb = gridDim.x;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        const double2 qv = op_apply<GRID>(op, i, [&](int j) {
            const double2 yj = y[j];
            return make_double2(s * yj.x, s * yj.y);
        });
        const double2 yi = y[i], di = dinv[i];
        const double2 vi = make_double2(s * yi.x, s * yi.y);
        const double2 qi = make_double2(di.x != 0. ? qv.x : 0., di.y != 0. ? qv.y : 0.);
        v[i] = vi;
        q[i] = qi;
        if (i >= own_lo && i < own_hi) {
            const double2 a = r1[i];
            acc = fma(vi.x, qi.x, fma(vi.y, qi.y, acc));
            acc1 = fma(vi.x, a.x, fma(vi.y, a.y, acc1));  // alfa = v.(K v - (beta / oldb) r1), as Paige & Saunders order it
        }
    }
    const double t = block_sum(acc, sh);
    const double t1 = block_sum(acc1, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = t;
        part_vr1[blockIdx.x] = t1;
    }
}

// y = q - c2 r2 - c1 r1;  r1 <- r2;  r2 <- y
__global__ void __launch_bounds__(BLOCK)
k_minres_update1(int nn, double c2, double c1, const double2 *__restrict__ q, double2 *__restrict__ r2, double2 *__restrict__ r1)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nn; i += gridDim.x * BLOCK) {
        const double2 qi = q[i], b = r2[i], a = r1[i];
        r1[i] = b;
        r2[i] = make_double2(qi.x - c2 * b.x - c1 * a.x, qi.y - c2 * b.y - c1 * a.y);
    }
}

// w = (v - oldeps w1 - delta w2) / gamma (written over w1, the oldest direction);  x += phi w
__global__ void __launch_bounds__(BLOCK)
k_minres_update2(int nn, double oldeps, double delta, double inv_gamma, double phi, const double2 *__restrict__ v,
                 double2 *__restrict__ w1, const double2 *__restrict__ w2, double2 *__restrict__ x)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nn; i += gridDim.x * BLOCK) {
        const double2 vi = v[i], a = w1[i], b = w2[i];
        double2 xi = x[i];
        const double2 w = make_double2((vi.x - oldeps * a.x - delta * b.x) * inv_gamma, (vi.y - oldeps * a.y - delta * b.y) * inv_gamma);
        w1[i] = w;
        xi.x = fma(phi, w.x, xi.x);
        xi.y = fma(phi, w.y, xi.y);
        x[i] = xi;
    }
}

// partials of |P (b - K x)|^2 over [own_lo, own_hi) (true residual of an iterate; nothing stored)
template <int GRID>
__global__ void __launch_bounds__(BLOCK)
k_resid_norm(KOp op, int nnode, const double2 *__restrict__ x, const double2 *__restrict__ b, const double2 *__restrict__ dinv,
             double *__restrict__ part, int own_lo, int own_hi)
{
    __shared__ double sh[BLOCK / 64];
    double acc = 0.;
    const int nb = gridDim.x;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode || i < own_lo || i >= own_hi) continue;
        const double2 qv = op_apply<GRID>(op, i, [&](int j) { return x[j]; });
        const double2 bi = b[i], di = dinv[i];
        const double rx = di.x != 0. ? bi.x - qv.x : 0., ry = di.y != 0. ? bi.y - qv.y : 0.;
        acc = fma(rx, rx, fma(ry, ry, acc));
    }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------------------------- SQMR
// Simplified QMR for symmetric indefinite systems with a symmetric, possibly INDEFINITE preconditioner (Freund & Nachtigal 1994,
// algorithm without look-ahead, right preconditioning): the coupled two-term recurrences of preconditioned CG -- which do not
// need positivity, only p.Kp != 0 and r.Br != 0 -- and a quasi-minimal-residual smoothing of the iterates:
//   t = K q;  sigma = q.t;  alpha = rho / sigma;  r -= alpha t;  theta = |r| / tau;  c = 1 / sqrt(1 + theta^2);  tau *= theta c
//   d = (c theta_old)^2 d + c^2 alpha q;  x += d;  z = B r;  rho_new = r.z;  q = z + (rho_new / rho) q
// Short recurrences like MINRES (no Krylov basis, no orthogonalisation: a GMRES iteration at 2048^2 reads its j basis vectors
// four times -- 27 us each -- against ~1 ms for the V-cycle) and, unlike MINRES, the V-cycle built on the indefinite operator
// itself is admissible (it is symmetric: same smoother before and after, restriction = prolongation^T).
// k_sqmr_apply: q_new = z + beta q_old (written to the other buffer), t = P K q_new, partials of q_new . t
template <int GRID>
__global__ void __launch_bounds__(BLOCK)
k_sqmr_apply(KOp op, int nnode, double beta, const double2 *__restrict__ z, const double2 *__restrict__ qold,
             const double2 *__restrict__ dinv, double2 *__restrict__ qnew, double2 *__restrict__ t, double *__restrict__ part,
             int own_lo, int own_hi)
{
    __shared__ double sh[BLOCK / 64];
    double acc = 0.;
    const int nb = gridDim.x;
    for (int tl = xcd_tile(blockIdx.x, nb); tl * BLOCK < nnode; tl += nb) {
        const int i = tl * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        const double2 kv = op_apply<GRID>(op, i, [&](int j) {
            const double2 zj = z[j], qj = qold[j];
            return make_double2(fma(beta, qj.x, zj.x), fma(beta, qj.y, zj.y));
        });
        const double2 zi = z[i], qo = qold[i], di = dinv[i];
        const double2 qi = make_double2(fma(beta, qo.x, zi.x), fma(beta, qo.y, zi.y));
        const double2 ti = make_double2(di.x != 0. ? kv.x : 0., di.y != 0. ? kv.y : 0.);
        qnew[i] = qi;
        t[i] = ti;
        if (i >= own_lo && i < own_hi) acc = fma(qi.x, ti.x, fma(qi.y, ti.y, acc));
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// r -= alpha t; partials of |r|^2 over the owned nodes
__global__ void __launch_bounds__(BLOCK)
k_sqmr_update_r(int nn, double alpha, const double2 *__restrict__ t, double2 *__restrict__ r, double *__restrict__ part, int own_lo, int own_hi)
{
    __shared__ double sh[BLOCK / 64];
    double acc = 0.;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nn; i += gridDim.x * BLOCK) {
        const double2 ti = t[i];
        double2 ri = r[i];
        ri.x = fma(-alpha, ti.x, ri.x);
        ri.y = fma(-alpha, ti.y, ri.y);
        r[i] = ri;
        if (i >= own_lo && i < own_hi) acc = fma(ri.x, ri.x, fma(ri.y, ri.y, acc));
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// d = cd d + cq q;  x += d
__global__ void __launch_bounds__(BLOCK)
k_sqmr_update_x(int nn, double cd, double cq, const double2 *__restrict__ q, double2 *__restrict__ d, double2 *__restrict__ x)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nn; i += gridDim.x * BLOCK) {
        const double2 qi = q[i], di = d[i];
        double2 xi = x[i];
        const double2 dn = make_double2(fma(cd, di.x, cq * qi.x), fma(cd, di.y, cq * qi.y));
        d[i] = dn;
        xi.x += dn.x;
        xi.y += dn.y;
        x[i] = xi;
    }
}

// ---------------------------------------------------------------------------------------------- GMRES
// Restarted GMRES with the V-cycle as RIGHT preconditioner: the solver of last resort for indefinite tangent stiffness when
// the V-cycle built on such an operator is not positive definite either (MINRES needs an SPD preconditioner, GMRES needs
// nothing).  Minimises the true residual |P(b - K x)|_2, the quantity PCG's stopping test uses.  Classical Gram-Schmidt
// with re-orthogonalisation, eight basis vectors per pass (coefficients by value).
struct Coef8 {
    double c[8];
};
struct Ptr8 {
    const double2 *p[8];
};

// partials of w . V_k, k < n (<= 8), over [own_lo, own_hi): part[k * MAXPART + block]
__global__ void __launch_bounds__(BLOCK)
k_gmres_dots(int own_lo, int own_hi, int n, const double2 *__restrict__ w, Ptr8 V, double *__restrict__ part)
{
    __shared__ double sh[BLOCK / 64];
    double acc[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    for (int i = own_lo + blockIdx.x * BLOCK + threadIdx.x; i < own_hi; i += gridDim.x * BLOCK) {
        const double2 wi = w[i];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < n) {
                const double2 v = V.p[k][i];
                acc[k] = fma(wi.x, v.x, fma(wi.y, v.y, acc[k]));
            }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= n) break;
        const double t = block_sum(acc[k], sh);
        if (threadIdx.x == 0) part[(size_t)k * MAXPART + blockIdx.x] = t;
    }
}

// w += sum_k c_k V_k, k < n (<= 8); with norm_part: partials of |w|^2 over [own_lo, own_hi) after the update
__global__ void __launch_bounds__(BLOCK)
k_gmres_axpy(int nn, int n, double2 *__restrict__ w, Ptr8 V, Coef8 C, double *__restrict__ norm_part, int own_lo, int own_hi)
{
    __shared__ double sh[BLOCK / 64];
    double acc = 0.;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nn; i += gridDim.x * BLOCK) {
        double2 wi = w[i];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < n) {
                const double2 v = V.p[k][i];
                wi.x = fma(C.c[k], v.x, wi.x);
                wi.y = fma(C.c[k], v.y, wi.y);
            }
        w[i] = wi;
        if (norm_part && i >= own_lo && i < own_hi) acc = fma(wi.x, wi.x, fma(wi.y, wi.y, acc));
    }
    if (norm_part) {
        const double t = block_sum(acc, sh);
        if (threadIdx.x == 0) norm_part[blockIdx.x] = t;
    }
}

// dst = s * src (and a second copy, if given)
__global__ void __launch_bounds__(BLOCK)
k_scale_copy(int nn, double s, const double2 *__restrict__ src, double2 *__restrict__ dst, double2 *__restrict__ dst2)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nn; i += gridDim.x * BLOCK) {
        const double2 a = src[i];
        const double2 b = make_double2(s * a.x, s * a.y);
        dst[i] = b;
        if (dst2) dst2[i] = b;
    }
}

// ---- GMRES with delayed re-orthogonalisation (round 6): the basis is read TWICE per iteration instead of four
// times.  Classical Gram-Schmidt twice (above) makes two projection passes per new vector, each a dots pass + an update pass
// over the j basis vectors.  Here the second pass of vector j is delayed by one iteration and shares its two sweeps over the
// basis with the first pass of vector j + 1 (Swirydowicz, Langou, Ananthan, Yang, Thomas 2020; Bielich et al. 2022: "DCGS-2"):
// with u = the once-projected candidate for q_j and z = A u, ONE dots pass yields s = Q^T u, t = Q^T z, u.u, u.z, and ONE update
// pass writes q_j = (u - Q s) / alpha and the next candidate (z - Q t) / alpha - e alpha q_j (plfx.hip: gmres_solve has the
// algebra).  Up to 128 basis vectors per dots launch, all of them in one update launch.
struct GmBlocks {
    const double *blk[40];   // 40 allocations of GMRES_BLK = 32 vectors hold the 1201 vectors of the longest cycle
};

// partials of V_k . a (column 2 k) and V_k . b (column 2 k + 1) for the basis vectors k0 <= k < k0 + kn (kn <= GM_KC) over
// [own_lo, own_hi); extra >= 0: also a . a and a . b into columns extra, extra + 1.   part[column * MAXPART + block].
// b == nullptr: a only.  A workgroup walks tiles of 512 nodes: every thread keeps its two nodes of a and b in registers and loops
// over the kn basis vectors (two 16-byte loads per vector), the products are summed across the wave by DPP butterflies + four
// readlanes (no LDS crossbar) and lane 0 adds them to the wave's own accumulator row in LDS; a and b are read once per GM_KC
// vectors (first version: 16 vectors per launch with 32 register accumulators per thread -- 3.3 TB/s and a re-read of a, b per
// launch; profiles/r07d_config5_kernel_summary_first_dcgs2.txt).
constexpr int GM_KC = 128;
__global__ void __launch_bounds__(BLOCK)
k_gmres_dots3(int own_lo, int own_hi, int k0, int kn, GmBlocks B, size_t nd, const double2 *__restrict__ a,
              const double2 *__restrict__ b, int extra, double *__restrict__ part)
{
    __shared__ double acc[BLOCK / 64][2 * GM_KC];
    __shared__ double sh[BLOCK / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = lane; q < 2 * GM_KC; q += 64) acc[wave][q] = 0.;   // (a wave only ever touches its own row: no barrier needed)
    const bool two = b != nullptr;
    double aa = 0., ab = 0.;
    constexpr int TILE = 2 * BLOCK;
    for (int base = own_lo + blockIdx.x * TILE; base < own_hi; base += gridDim.x * TILE) {
        const int i0 = base + threadIdx.x, i1 = i0 + BLOCK;
        const bool ok0 = i0 < own_hi, ok1 = i1 < own_hi;
        const double2 zero = make_double2(0., 0.);
        const double2 a0 = ok0 ? a[i0] : zero, a1 = ok1 ? a[i1] : zero;
        const double2 b0 = (two && ok0) ? b[i0] : zero, b1 = (two && ok1) ? b[i1] : zero;
        if (extra >= 0) {
            aa = fma(a0.x, a0.x, fma(a0.y, a0.y, fma(a1.x, a1.x, fma(a1.y, a1.y, aa))));
            ab = fma(a0.x, b0.x, fma(a0.y, b0.y, fma(a1.x, b1.x, fma(a1.y, b1.y, ab))));
        }
#pragma unroll 4
        for (int k = 0; k < kn; k++) {
            const double2 *vk = (const double2 *)(B.blk[(k0 + k) >> 5] + (size_t)((k0 + k) & 31) * nd);
            const double2 v0 = ok0 ? vk[i0] : zero, v1 = ok1 ? vk[i1] : zero;
            double pa = fma(a0.x, v0.x, fma(a0.y, v0.y, fma(a1.x, v1.x, a1.y * v1.y)));
            pa = wave_allsum(pa);
            double pb = 0.;
            if (two) {
                pb = fma(b0.x, v0.x, fma(b0.y, v0.y, fma(b1.x, v1.x, b1.y * v1.y)));
                pb = wave_allsum(pb);
            }
            if (lane == 0) {
                acc[wave][2 * k] += pa;
                if (two) acc[wave][2 * k + 1] += pb;
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 2 * kn; q += BLOCK) {
        if (!two && (q & 1)) continue;
        double t = acc[0][q];
#pragma unroll
        for (int w = 1; w < BLOCK / 64; w++) t += acc[w][q];
        part[(size_t)(2 * k0 + q) * MAXPART + blockIdx.x] = t;
    }
    if (extra >= 0) {
        const double t0 = block_sum(aa, sh);
        const double t1 = block_sum(ab, sh);
        if (threadIdx.x == 0) {
            part[(size_t)extra * MAXPART + blockIdx.x] = t0;
            part[(size_t)(extra + 1) * MAXPART + blockIdx.x] = t1;
        }
    }
}

// out[col] = sum of the gn partials of column col, in block order (one wave per column; deterministic)
__global__ void __launch_bounds__(BLOCK)
k_gmres_reduce(int ncol, int gn, const double *__restrict__ part, double *__restrict__ out)
{
    const int col = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (col >= ncol) return;
    double t = 0.;
    for (int i = lane; i < gn; i += 64) t += part[(size_t)col * MAXPART + i];
    t = wave_sum(t);
    if (lane == 0) out[col] = t;
}

// One sweep over the j basis vectors for both vectors of an iteration (coefficients cs[0..j) and cf[0..j) in device memory,
// read with scalar loads):   qout = (u - sum_k cs_k Q_k) * inv_alpha ;   uout = z * inv_alpha - e u - sum_k cf_k Q_k.
// z == nullptr: first step of a cycle, uout = u - sum_k cf_k Q_k only (qout untouched).  qout may alias u.
__global__ void __launch_bounds__(BLOCK)
k_gmres_update2(int nn, size_t nd, int j, GmBlocks B, const double *__restrict__ cs, const double *__restrict__ cf, double inv_alpha,
                double e, const double2 *u, const double2 *__restrict__ z, double2 *qout, double2 *__restrict__ uout)
{
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nn; i += gridDim.x * BLOCK) {
        const double2 ui = u[i];
        double2 a = ui, b;
        if (z) {
            const double2 zi = z[i];
            b = make_double2(fma(-e, ui.x, zi.x * inv_alpha), fma(-e, ui.y, zi.y * inv_alpha));
        } else
            b = ui;
        int k = 0;
        for (; k + 8 <= j; k += 8) {
            double2 v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = ((const double2 *)(B.blk[(k + q) >> 5] + (size_t)((k + q) & 31) * nd))[i];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const double s = z ? cs[k + q] : 0., f = cf[k + q];
                a.x = fma(-s, v[q].x, a.x);
                a.y = fma(-s, v[q].y, a.y);
                b.x = fma(-f, v[q].x, b.x);
                b.y = fma(-f, v[q].y, b.y);
            }
        }
        for (; k < j; k++) {
            const double2 v = ((const double2 *)(B.blk[k >> 5] + (size_t)(k & 31) * nd))[i];
            const double s = z ? cs[k] : 0., f = cf[k];
            a.x = fma(-s, v.x, a.x);
            a.y = fma(-s, v.y, a.y);
            b.x = fma(-f, v.x, b.x);
            b.y = fma(-f, v.y, b.y);
        }
        if (z) qout[i] = make_double2(a.x * inv_alpha, a.y * inv_alpha);
        uout[i] = b;
    }
}

constexpr int MG_COARSE_MAX = 1089;  // 33 x 33 nodes
constexpr int MG_TAIL_BLOCK = 1024;  // threads of the single-workgroup tail kernel
constexpr int MG_TAIL_NODES = 1089;  // levels up to 33 x 33 nodes run inside the tail kernel
static_assert(MG_TAIL_NODES <= 2 * MG_TAIL_BLOCK, "k_mg_tail_mf: a thread owns at most two nodes of a level");
constexpr int MG_DENSE_MAX = 128;    // coarsest grids up to this many DOFs are solved with a dense inverse

struct MgLevDev {
    int nx, ny, nnode, nslot;
    const double *ainv;  // dense inverse of the (masked) operator, coarsest level only (or nullptr)
    int tail_off, _pad;  // node offset of this level inside the LDS arena of k_mg_tail_lds
    const int32_t *col;
    const double *val;
    const double2 *dinv;
    double2 *x, *b, *t, *res;
    const double *Mel;   // stiffness generators of the level, SoA [6][nel] (matrix-free tail)
    int nel, elem_off;   // elements of the level, element offset inside the LDS arena of k_mg_tail_mf
    double rx, ry;       // relative size of the level's last element column / row (KOp::rx, ry)
};

// the level halves exactly and all its cells have one size: the transfer weights to the next level are 1, 1/2, 1/4
__device__ __forceinline__ bool mg_level_plain(const MgLevDev &L) { return !(L.nx & 1) && !(L.ny & 1) && L.rx == 1. && L.ry == 1.; }

__device__ inline void coarse_solve_block(int nnode, int nslot, const int32_t *__restrict__ col,
                                          const double *__restrict__ val, const double2 *__restrict__ dinv,
                                          const double2 *__restrict__ b, double2 *__restrict__ x, int maxit,
                                          double rtol, double2 *sh2)
{
    const int nt = blockDim.x, nw = nt >> 6;
    double2 *xs = sh2, *r = sh2 + nnode, *p = sh2 + 2 * nnode, *q = sh2 + 3 * nnode;
    __shared__ double red[MG_TAIL_BLOCK / 64];
    __shared__ double bc;
    auto bsum = [&](double v) {
        v = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.;
            for (int i = 0; i < nw; i++) t += red[i];
            bc = t;
        }
        __syncthreads();
        return bc;
    };
    double a_rz = 0., a_bb = 0.;
    for (int i = threadIdx.x; i < nnode; i += nt) {
        const double2 bi = b[i], di = dinv[i];
        xs[i] = make_double2(0., 0.);
        r[i] = bi;
        const double2 zi = make_double2(di.x * bi.x, di.y * bi.y);
        p[i] = zi;
        a_rz += bi.x * zi.x + bi.y * zi.y;
        a_bb += bi.x * bi.x + bi.y * bi.y;
    }
    double rz = bsum(a_rz);
    const double bb = bsum(a_bb);
    const double thresh = rtol * rtol * bb;
    for (int it = 0; it < maxit && bb > 0.; it++) {
        double a_pq = 0.;
        for (int i = threadIdx.x; i < nnode; i += nt) {
            double qx = 0., qy = 0.;
            for (int s = 0; s < nslot; s++) {
                const int j = col[(size_t)s * nnode + i];
                if (j < 0) continue;
                const double2 pj = p[j];
                qx = fma(val[((size_t)s * 4 + 0) * nnode + i], pj.x, fma(val[((size_t)s * 4 + 1) * nnode + i], pj.y, qx));
                qy = fma(val[((size_t)s * 4 + 2) * nnode + i], pj.x, fma(val[((size_t)s * 4 + 3) * nnode + i], pj.y, qy));
            }
            const double2 di = dinv[i];
            if (di.x == 0.) qx = 0.;
            if (di.y == 0.) qy = 0.;
            q[i] = make_double2(qx, qy);
            a_pq += p[i].x * qx + p[i].y * qy;
        }
        const double pq = bsum(a_pq);
        if (!(pq > 0.)) break;
        const double alpha = rz / pq;
        double a_rzn = 0., a_rr = 0.;
        for (int i = threadIdx.x; i < nnode; i += nt) {
            double2 xi = xs[i], ri = r[i];
            const double2 pi = p[i], qi = q[i], di = dinv[i];
            xi.x += alpha * pi.x;
            xi.y += alpha * pi.y;
            ri.x -= alpha * qi.x;
            ri.y -= alpha * qi.y;
            xs[i] = xi;
            r[i] = ri;
            a_rzn += ri.x * ri.x * di.x + ri.y * ri.y * di.y;
            a_rr += ri.x * ri.x + ri.y * ri.y;
        }
        const double rzn = bsum(a_rzn);
        const double rr = bsum(a_rr);
        if (rr <= thresh) break;
        const double beta = rzn / rz;
        rz = rzn;
        for (int i = threadIdx.x; i < nnode; i += nt) {
            const double2 ri = r[i], di = dinv[i], pi = p[i];
            p[i] = make_double2(fma(beta, pi.x, di.x * ri.x), fma(beta, pi.y, di.y * ri.y));
        }
        __syncthreads();
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nnode; i += nt) x[i] = xs[i];
    __syncthreads();
}

// Dense inverse of the coarsest operator (n = 2*nnode <= MG_DENSE_MAX) by in-place Gauss-Jordan in
// LDS; prescribed DOFs are replaced by identity rows/columns.  Runs once per apply_bc.
__global__ void __launch_bounds__(BLOCK)
k_mg_coarse_invert(int nnode, int nslot, const int32_t *__restrict__ col, const double *__restrict__ val,
                   const double2 *__restrict__ dinv, double *__restrict__ ainv)
{
    extern __shared__ double A[];
    const int n = 2 * nnode;
    const double *dv = reinterpret_cast<const double *>(dinv);
    for (int i = threadIdx.x; i < n * n; i += BLOCK) A[i] = 0.;
    __syncthreads();
    for (int idx = threadIdx.x; idx < nnode * nslot; idx += BLOCK) {
        const int i = idx % nnode, s = idx / nnode;
        const int j = col[(size_t)s * nnode + i];
        if (j < 0) continue;
        for (int r = 0; r < 2; r++)
            for (int cc = 0; cc < 2; cc++) {
                const int row = 2 * i + r, cl = 2 * j + cc;
                const bool freedof = dv[row] != 0. && dv[cl] != 0.;
                A[row * n + cl] = freedof ? val[((size_t)s * 4 + r * 2 + cc) * nnode + i] : (row == cl ? 1. : 0.);
            }
    }
    __syncthreads();
    // in-place Gauss-Jordan, two barriers per pivot: every thread first reads what its entries need from the
    // old matrix (pivot row k, pivot column k), then all write
    constexpr int PER = (MG_DENSE_MAX * MG_DENSE_MAX + BLOCK - 1) / BLOCK;
    for (int k = 0; k < n; k++) {
        double pk = A[k * n + k];
        if (fabs(pk) < 1e-300) pk = 1.;
        const double ip = 1. / pk;
        double nv[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int idx = threadIdx.x + u * BLOCK;
            if (idx >= n * n) break;
            const int i = idx / n, j = idx - i * n;
            const double aik = A[i * n + k], akj = A[k * n + j], aij = A[idx];
            double v;
            if (i == k)
                v = (j == k) ? ip : akj * ip;
            else if (j == k)
                v = -aik * ip;
            else
                v = fma(-aik * ip, akj, aij);
            nv[u] = v;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int idx = threadIdx.x + u * BLOCK;
            if (idx >= n * n) break;
            A[idx] = nv[u];
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n * n; i += BLOCK) ainv[i] = A[i];
}

// x = Ainv b on the coarsest grid (masked DOFs stay zero because b is zero there)
__device__ inline void coarse_dense_block(int nnode, const double *__restrict__ ainv,
                                          const double2 *__restrict__ b, double2 *__restrict__ x)
{
    const int n = 2 * nnode;
    const double *bv = reinterpret_cast<const double *>(b);
    double *xv = reinterpret_cast<double *>(x);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double s = 0.;
        for (int j = 0; j < n; j++) s = fma(ainv[(size_t)i * n + j], bv[j], s);
        xv[i] = s;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(BLOCK)
k_mg_coarse_solve(int nnode, int nslot, const int32_t *__restrict__ col, const double *__restrict__ val,
                  const double2 *__restrict__ dinv, const double2 *__restrict__ b, double2 *__restrict__ x,
                  int maxit, double rtol, const CgScalars *sc)
{
    if (sc->done) return;
    extern __shared__ double2 sh2[];
    coarse_solve_block(nnode, nslot, col, val, dinv, b, x, maxit, rtol, sh2);
}

__global__ void __launch_bounds__(BLOCK)
k_mg_coarse_dense(int nnode, const double *__restrict__ ainv, const double2 *__restrict__ b,
                  double2 *__restrict__ x, const CgScalars *sc)
{
    if (sc->done) return;
    coarse_dense_block(nnode, ainv, b, x);
}

// ---- single-workgroup pieces of the V-cycle (levels with few nodes are launch-latency bound:
//      one workgroup walks them all, separated by workgroup barriers instead of kernel boundaries)
__device__ inline void blk_smooth(const MgLevDev &L, const double2 *xin, double2 *xout, double omega, int first)
{
    for (int i = threadIdx.x; i < L.nnode; i += blockDim.x) {
        const double2 di = L.dinv[i], bi = L.b[i];
        if (first) {
            xout[i] = make_double2(omega * di.x * bi.x, omega * di.y * bi.y);
            continue;
        }
        double qx = 0., qy = 0.;
        for (int s = 0; s < L.nslot; s++) {
            const int j = L.col[(size_t)s * L.nnode + i];
            if (j < 0) continue;
            const double2 pj = xin[j];
            qx = fma(L.val[((size_t)s * 4 + 0) * L.nnode + i], pj.x, fma(L.val[((size_t)s * 4 + 1) * L.nnode + i], pj.y, qx));
            qy = fma(L.val[((size_t)s * 4 + 2) * L.nnode + i], pj.x, fma(L.val[((size_t)s * 4 + 3) * L.nnode + i], pj.y, qy));
        }
        const double2 xi = xin[i];
        xout[i] = make_double2(fma(omega * di.x, bi.x - qx, xi.x), fma(omega * di.y, bi.y - qy, xi.y));
    }
    __syncthreads();
}

__device__ inline void blk_residual(const MgLevDev &L)
{
    for (int i = threadIdx.x; i < L.nnode; i += blockDim.x) {
        double qx = 0., qy = 0.;
        for (int s = 0; s < L.nslot; s++) {
            const int j = L.col[(size_t)s * L.nnode + i];
            if (j < 0) continue;
            const double2 pj = L.x[j];
            qx = fma(L.val[((size_t)s * 4 + 0) * L.nnode + i], pj.x, fma(L.val[((size_t)s * 4 + 1) * L.nnode + i], pj.y, qx));
            qy = fma(L.val[((size_t)s * 4 + 2) * L.nnode + i], pj.x, fma(L.val[((size_t)s * 4 + 3) * L.nnode + i], pj.y, qy));
        }
        const double2 di = L.dinv[i], bi = L.b[i];
        L.res[i] = make_double2(di.x != 0. ? bi.x - qx : 0., di.y != 0. ? bi.y - qy : 0.);
    }
    __syncthreads();
}

__device__ inline void blk_restrict(const MgLevDev &F, const MgLevDev &Cc)
{
    const int nyc = Cc.ny + 1, nyf = F.ny + 1;
    const bool plain = mg_level_plain(F);
    for (int i = threadIdx.x; i < Cc.nnode; i += blockDim.x) {
        const int J = i / nyc, K = i - J * nyc;
        const double2 sr = mg_restrict_at(J, K, F.nx, F.ny, F.rx, F.ry, plain, [&](int jf, int kf) { return F.res[(size_t)jf * nyf + kf]; });
        const double2 d = Cc.dinv[i];
        Cc.b[i] = make_double2(d.x != 0. ? sr.x : 0., d.y != 0. ? sr.y : 0.);
    }
    __syncthreads();
}

__device__ inline void blk_prolong_add(const MgLevDev &F, const MgLevDev &Cc)
{
    const int nyc = Cc.ny + 1, nyf = F.ny + 1;
    for (int i = threadIdx.x; i < F.nnode; i += blockDim.x) {
        const int j = i / nyf, k = i - j * nyf;
        const int J0 = j >> 1, K0 = k >> 1, oj = j & 1, ok = k & 1;
        double cx, cy, w = 1.;
        if (mg_level_plain(F)) {
            double2 v = Cc.x[(size_t)J0 * nyc + K0];
            cx = v.x, cy = v.y;
            if (oj) {
                v = Cc.x[(size_t)(J0 + 1) * nyc + K0];
                cx += v.x;
                cy += v.y;
            }
            if (ok) {
                v = Cc.x[(size_t)J0 * nyc + K0 + 1];
                cx += v.x;
                cy += v.y;
            }
            if (oj && ok) {
                v = Cc.x[(size_t)(J0 + 1) * nyc + K0 + 1];
                cx += v.x;
                cy += v.y;
            }
            w = (oj ? 0.5 : 1.) * (ok ? 0.5 : 1.);
        } else {
            const double2 v = mg_interpolate(j, k, F.nx, F.ny, F.rx, F.ry, nyc, [&](int q) { return Cc.x[q]; });
            cx = v.x, cy = v.y;
        }
        const double2 d = F.dinv[i];
        double2 xf = F.x[i];
        if (d.x != 0.) xf.x = fma(w, cx, xf.x);
        if (d.y != 0.) xf.y = fma(w, cy, xf.y);
        F.x[i] = xf;
    }
    __syncthreads();
}

// V-cycle over the levels [l0, nl) in ONE workgroup: b of level l0 in, x of level l0 out.
__global__ void __launch_bounds__(MG_TAIL_BLOCK)
k_mg_tail(const MgLevDev *__restrict__ lev, int l0, int nl, double omega, int nu, const CgScalars *sc)
{
    if (sc->done) return;
    extern __shared__ double2 sh2[];
    for (int l = l0; l < nl - 1; l++) {
        const MgLevDev L = lev[l];
        double2 *src = nullptr, *dst = (nu & 1) ? L.x : L.t;
        for (int k = 0; k < nu; k++) {
            blk_smooth(L, src, dst, omega, k == 0);
            src = dst;
            dst = (dst == L.x) ? L.t : L.x;
        }
        blk_residual(L);
        blk_restrict(L, lev[l + 1]);
    }
    {
        const MgLevDev L = lev[nl - 1];
        if (L.ainv)
            coarse_dense_block(L.nnode, L.ainv, L.b, L.x);
        else
            coarse_solve_block(L.nnode, L.nslot, L.col, L.val, L.dinv, L.b, L.x, 4 * L.nnode + 20, 1.e-10, sh2);
    }
    for (int l = nl - 2; l >= l0; l--) {
        const MgLevDev L = lev[l];
        blk_prolong_add(L, lev[l + 1]);
        double2 *src = L.x, *dst = L.t;
        for (int k = 0; k < nu; k++) {
            blk_smooth(L, src, dst, omega, 0);
            double2 *tmp = src;
            src = dst;
            dst = tmp;
        }
        if (src != L.x) {
            for (int i = threadIdx.x; i < L.nnode; i += blockDim.x) L.x[i] = L.t[i];
            __syncthreads();
        }
    }
}

// ---- LDS-resident tail: the vectors (x, b, t, res) and the neighbour tables of ALL tail levels live in
//      LDS (<= ~1500 nodes x 100 B), so a smoothing step is [coalesced matrix loads from L2] + [LDS gathers]
//      with no dependent global-memory chain; only the level-l0 right-hand side is read from and the
//      level-l0 correction written to global memory.  Requires the dense coarse inverse.
struct TailView {
    double2 *x, *b, *t, *res;
    const int *col;
};

__device__ __forceinline__ TailView tail_view(double2 *arena, int *cols, int T, const MgLevDev &L)
{
    TailView v;
    v.x = arena + L.tail_off;
    v.b = arena + T + L.tail_off;
    v.t = arena + 2 * T + L.tail_off;
    v.res = arena + 3 * T + L.tail_off;
    v.col = cols + 9 * L.tail_off;
    return v;
}

// y = K x at node i with x in LDS and the neighbour table in LDS ([node][slot] layout)
__device__ __forceinline__ double2 tail_apply(const MgLevDev &L, const TailView &v, const double2 *xin, int i)
{
    double qx = 0., qy = 0.;
    if (L.nslot == 9) {
        // branch-free and unrolled like bell_apply: the 36 matrix values (L2) are all in flight together instead of
        // one dependent global-memory latency per slot; empty slots hold exact zeros and gather the node itself
        int j[9];
        double a[36];
#pragma unroll
        for (int s = 0; s < 9; s++) j[s] = v.col[9 * i + s];
#pragma unroll
        for (int k = 0; k < 36; k++) a[k] = L.val[(size_t)k * L.nnode + i];
#pragma unroll
        for (int s = 0; s < 9; s++) {
            const double2 pj = xin[j[s] < 0 ? i : j[s]];
            qx = fma(a[4 * s + 0], pj.x, fma(a[4 * s + 1], pj.y, qx));
            qy = fma(a[4 * s + 2], pj.x, fma(a[4 * s + 3], pj.y, qy));
        }
        return make_double2(qx, qy);
    }
    for (int s = 0; s < L.nslot; s++) {
        const int j = v.col[9 * i + s];
        if (j < 0) continue;
        const double2 pj = xin[j];
        qx = fma(L.val[((size_t)s * 4 + 0) * L.nnode + i], pj.x, fma(L.val[((size_t)s * 4 + 1) * L.nnode + i], pj.y, qx));
        qy = fma(L.val[((size_t)s * 4 + 2) * L.nnode + i], pj.x, fma(L.val[((size_t)s * 4 + 3) * L.nnode + i], pj.y, qy));
    }
    return make_double2(qx, qy);
}

__global__ void __launch_bounds__(MG_TAIL_BLOCK)
k_mg_tail_lds(const MgLevDev *__restrict__ lev, int l0, int nl, int T, double omega, const CgScalars *sc)
{
    if (sc->done) return;
    extern __shared__ double2 arena[];  // 4 T double2 + 9 T int
    int *cols = reinterpret_cast<int *>(arena + 4 * (size_t)T);
    const int nt = blockDim.x;
    // stage: neighbour tables of every tail level, right-hand side of level l0
    for (int l = l0; l < nl; l++) {
        const MgLevDev L = lev[l];
        for (int idx = threadIdx.x; idx < L.nnode * 9; idx += nt) {
            const int i = idx / 9, s = idx - 9 * i;
            cols[9 * (L.tail_off + i) + s] = (s < L.nslot) ? L.col[(size_t)s * L.nnode + i] : -1;
        }
    }
    {
        const MgLevDev L = lev[l0];
        TailView v = tail_view(arena, cols, T, L);
        for (int i = threadIdx.x; i < L.nnode; i += nt) v.b[i] = L.b[i];
    }
    __syncthreads();
    for (int l = l0; l < nl - 1; l++) {  // down: two Jacobi sweeps from zero (one pass), residual, restriction
        const MgLevDev L = lev[l];
        const MgLevDev Cc = lev[l + 1];
        TailView v = tail_view(arena, cols, T, L), vc = tail_view(arena, cols, T, Cc);
        for (int i = threadIdx.x; i < L.nnode; i += nt) {  // t = x1 = w D^-1 b
            const double2 di = L.dinv[i], bi = v.b[i];
            v.t[i] = make_double2(omega * di.x * bi.x, omega * di.y * bi.y);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < L.nnode; i += nt) {  // x = x1 + w D^-1 (b - K x1)
            const double2 q = tail_apply(L, v, v.t, i), di = L.dinv[i], bi = v.b[i], x1 = v.t[i];
            v.x[i] = make_double2(fma(omega * di.x, bi.x - q.x, x1.x), fma(omega * di.y, bi.y - q.y, x1.y));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < L.nnode; i += nt) {  // res = P (b - K x)
            const double2 q = tail_apply(L, v, v.x, i), di = L.dinv[i], bi = v.b[i];
            v.res[i] = make_double2(di.x != 0. ? bi.x - q.x : 0., di.y != 0. ? bi.y - q.y : 0.);
        }
        __syncthreads();
        const int nyc = Cc.ny + 1, nyf = L.ny + 1;
        const bool plain_l = mg_level_plain(L);
        for (int i = threadIdx.x; i < Cc.nnode; i += nt) {  // b_c = P^T res
            const int J = i / nyc, K = i - J * nyc;
            const double2 sr = mg_restrict_at(J, K, L.nx, L.ny, L.rx, L.ry, plain_l, [&](int jf, int kf) { return v.res[jf * nyf + kf]; });
            const double sx = sr.x, sy = sr.y;
            const double2 d = Cc.dinv[i];
            vc.b[i] = make_double2(d.x != 0. ? sx : 0., d.y != 0. ? sy : 0.);
        }
        __syncthreads();
    }
    {   // coarsest grid: x = Ainv b
        const MgLevDev L = lev[nl - 1];
        TailView v = tail_view(arena, cols, T, L);
        const int n = 2 * L.nnode;
        const double *bv = reinterpret_cast<const double *>(v.b);
        double *xv = reinterpret_cast<double *>(v.x);
        for (int i = threadIdx.x; i < n; i += nt) {
            double acc = 0.;
            for (int j = 0; j < n; j++) acc = fma(L.ainv[(size_t)i * n + j], bv[j], acc);
            xv[i] = acc;
        }
        __syncthreads();
    }
    for (int l = nl - 2; l >= l0; l--) {  // up: prolongation + two post-smoothing sweeps
        const MgLevDev L = lev[l];
        const MgLevDev Cc = lev[l + 1];
        TailView v = tail_view(arena, cols, T, L), vc = tail_view(arena, cols, T, Cc);
        const int nyc = Cc.ny + 1, nyf = L.ny + 1;
        for (int i = threadIdx.x; i < L.nnode; i += nt) {
            const int j = i / nyf, k = i - j * nyf;
            const int J0 = j >> 1, K0 = k >> 1, oj = j & 1, ok = k & 1;
            double cx, cy, w = 1.;
            if (mg_level_plain(L)) {
                double2 c = vc.x[J0 * nyc + K0];
                cx = c.x, cy = c.y;
                if (oj) {
                    c = vc.x[(J0 + 1) * nyc + K0];
                    cx += c.x;
                    cy += c.y;
                }
                if (ok) {
                    c = vc.x[J0 * nyc + K0 + 1];
                    cx += c.x;
                    cy += c.y;
                }
                if (oj && ok) {
                    c = vc.x[(J0 + 1) * nyc + K0 + 1];
                    cx += c.x;
                    cy += c.y;
                }
                w = (oj ? 0.5 : 1.) * (ok ? 0.5 : 1.);
            } else {
                const double2 c = mg_interpolate(j, k, L.nx, L.ny, L.rx, L.ry, nyc, [&](int q) { return vc.x[q]; });
                cx = c.x, cy = c.y;
            }
            const double2 d = L.dinv[i];
            double2 xf = v.x[i];
            if (d.x != 0.) xf.x = fma(w, cx, xf.x);
            if (d.y != 0.) xf.y = fma(w, cy, xf.y);
            v.x[i] = xf;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < L.nnode; i += nt) {  // t = x + w D^-1 (b - K x)
            const double2 q = tail_apply(L, v, v.x, i), di = L.dinv[i], bi = v.b[i], xi = v.x[i];
            v.t[i] = make_double2(fma(omega * di.x, bi.x - q.x, xi.x), fma(omega * di.y, bi.y - q.y, xi.y));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < L.nnode; i += nt) {  // x = t + w D^-1 (b - K t)
            const double2 q = tail_apply(L, v, v.t, i), di = L.dinv[i], bi = v.b[i], ti = v.t[i];
            v.x[i] = make_double2(fma(omega * di.x, bi.x - q.x, ti.x), fma(omega * di.y, bi.y - q.y, ti.y));
        }
        __syncthreads();
    }
    {
        const MgLevDev L = lev[l0];
        TailView v = tail_view(arena, cols, T, L);
        for (int i = threadIdx.x; i < L.nnode; i += nt) L.x[i] = v.x[i];
    }
}

// ---- matrix-free LDS-resident tail: like k_mg_tail_lds, but the operator is applied from the stiffness generators,
//      which are staged into LDS together with the vectors (x, b, w: 48 B/node; M: 48 B/element; <= 1502 nodes and
//      1364 elements = 137.6 KB).  A smoothing step then touches no global memory except the node's own dinv (the
//      block-ELL tail reads 288 B of matrix per node and application through ONE compute unit: 5-7 us per application
//      on the 33 x 33 level, measured with in-kernel timestamps).  The residual is written over w.
// RAGGED: some level of the tail has an odd number of cells or a last cell of another size (hierarchies of odd-sized meshes):
// general transfer weights and shape-scaled stiffness integrals; the plain instantiation is the kernel of rounds 3-4
template <bool RAGGED>
__global__ void __launch_bounds__(MG_TAIL_BLOCK)
k_mg_tail_mf(const MgLevDev *__restrict__ lev, int l0, int nl, int T, int E, const double *__restrict__ tab,
             double omega, const CgScalars *sc)
{
    if (sc->done) return;
    extern __shared__ double2 arena[];  // 3 T double2 + 6 E double
    __shared__ MgLevDev slev[16];       // level descriptors: one global read at the start instead of one per visit
    double2 *X = arena, *Bv = arena + T, *W = arena + 2 * (size_t)T;
    double *Ms = reinterpret_cast<double *>(arena + 3 * (size_t)T);
    const int nt = blockDim.x;
    const int tid = threadIdx.x;
    if (tid >= l0 && tid < nl && tid < 16) slev[tid] = lev[tid];
    __syncthreads();
    const double2 zero2 = make_double2(0., 0.);
    for (int l = l0; l < nl - 1; l++) {  // the coarsest level is solved with the dense inverse: no generators needed
        const MgLevDev &L = slev[l];
        double *dst = Ms + 6 * (size_t)L.elem_off;
        for (int q = tid; q < 6 * L.nel; q += nt) dst[q] = L.Mel[q];
    }
    {
        const MgLevDev &L = slev[l0];
        for (int i = tid; i < L.nnode; i += nt) Bv[L.tail_off + i] = L.b[i];
    }
    __syncthreads();
    for (int l = l0; l < nl - 1; l++) {  // down
        const MgLevDev &L = slev[l];
        const MgLevDev &Cc = slev[l + 1];
        double2 *x = X + L.tail_off, *b = Bv + L.tail_off, *w = W + L.tail_off, *bc = Bv + Cc.tail_off;
        const double *Ml = Ms + 6 * (size_t)L.elem_off;
        const int nxn = L.nx + 1, nyn = L.ny + 1, nn = L.nnode, nel = L.nel;
        // a thread owns nodes tid and tid + nt of a level in every phase (MG_TAIL_NODES <= 2 * MG_TAIL_BLOCK): the Jacobi
        // scaling of its nodes -- the only global data of a phase -- is read once per visit of a level
        // (clamped index + value select: a conditional load would become a pointer select through scratch memory)
        double2 dA = L.dinv[tid < nn ? tid : 0], dB = L.dinv[tid + nt < nn ? tid + nt : 0];
        double2 dC = Cc.dinv[tid < Cc.nnode ? tid : 0];
        if (tid >= nn) dA = zero2;
        if (tid + nt >= nn) dB = zero2;
        if (tid >= Cc.nnode) dC = zero2;
#pragma unroll
        for (int r = 0; r < 2; r++) {  // w = x1 = omega D^-1 b
            const int i = tid + r * nt;
            if (i < nn) {
                const double2 di = r ? dB : dA, bi = b[i];
                w[i] = make_double2(omega * di.x * bi.x, omega * di.y * bi.y);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; r++) {  // x = x1 + omega D^-1 (b - K x1)
            const int i = tid + r * nt;
            if (i < nn) {
                const double2 di = r ? dB : dA;
                const double2 q = grid_apply_pairs<RAGGED>(nxn, nyn, nel, tab, i, [&](int m) { return reinterpret_cast<const double2 *>(Ml)[m]; },
                                               [&](int j) { return w[j]; }, L.rx, L.ry);
                const double2 bi = b[i], x1 = w[i];
                x[i] = make_double2(fma(omega * di.x, bi.x - q.x, x1.x), fma(omega * di.y, bi.y - q.y, x1.y));
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; r++) {  // w = res = P (b - K x)
            const int i = tid + r * nt;
            if (i < nn) {
                const double2 di = r ? dB : dA;
                const double2 q = grid_apply_pairs<RAGGED>(nxn, nyn, nel, tab, i, [&](int m) { return reinterpret_cast<const double2 *>(Ml)[m]; },
                                               [&](int j) { return x[j]; }, L.rx, L.ry);
                const double2 bi = b[i];
                w[i] = make_double2(di.x != 0. ? bi.x - q.x : 0., di.y != 0. ? bi.y - q.y : 0.);
            }
        }
        __syncthreads();
        const int nyc = Cc.ny + 1;
        if (tid < Cc.nnode) {  // b_c = P^T res   (a coarse level of the tail has at most nt nodes)
            const int i = tid;
            const int J = i / nyc, K = i - J * nyc;
            const double2 sr = mg_restrict_at(J, K, L.nx, L.ny, L.rx, L.ry, !RAGGED || mg_level_plain(L), [&](int jf, int kf) { return w[jf * nyn + kf]; });
            const double sx = sr.x, sy = sr.y;
            bc[i] = make_double2(dC.x != 0. ? sx : 0., dC.y != 0. ? sy : 0.);
        }
        __syncthreads();
    }
    {   // coarsest grid: x = Ainv b
        const MgLevDev &L = slev[nl - 1];
        const int n = 2 * L.nnode;
        const double *bv = reinterpret_cast<const double *>(Bv + L.tail_off);
        double *xv = reinterpret_cast<double *>(X + L.tail_off);
        for (int i = tid; i < n; i += nt) {
            double acc = 0.;
            for (int j = 0; j < n; j++) acc = fma(L.ainv[(size_t)i * n + j], bv[j], acc);
            xv[i] = acc;
        }
        __syncthreads();
    }
    for (int l = nl - 2; l >= l0; l--) {  // up: prolongation + two post-smoothing sweeps
        const MgLevDev &L = slev[l];
        const MgLevDev &Cc = slev[l + 1];
        double2 *x = X + L.tail_off, *b = Bv + L.tail_off, *w = W + L.tail_off;
        const double2 *xc = X + Cc.tail_off;
        const double *Ml = Ms + 6 * (size_t)L.elem_off;
        const int nxn = L.nx + 1, nyn = L.ny + 1, nyc = Cc.ny + 1, nn = L.nnode, nel = L.nel;
        double2 dA = L.dinv[tid < nn ? tid : 0], dB = L.dinv[tid + nt < nn ? tid + nt : 0];
        if (tid >= nn) dA = zero2;
        if (tid + nt >= nn) dB = zero2;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int i = tid + r * nt;
            if (i < nn) {
                const double2 d = r ? dB : dA;
                const int j = i / nyn, k = i - j * nyn;
                const int J0 = j >> 1, K0 = k >> 1, oj = j & 1, ok = k & 1;
                double cx, cy, wt = 1.;
                if (!RAGGED || mg_level_plain(L)) {
                    double2 c = xc[J0 * nyc + K0];
                    cx = c.x, cy = c.y;
                    if (oj) {
                        c = xc[(J0 + 1) * nyc + K0];
                        cx += c.x;
                        cy += c.y;
                    }
                    if (ok) {
                        c = xc[J0 * nyc + K0 + 1];
                        cx += c.x;
                        cy += c.y;
                    }
                    if (oj && ok) {
                        c = xc[(J0 + 1) * nyc + K0 + 1];
                        cx += c.x;
                        cy += c.y;
                    }
                    wt = (oj ? 0.5 : 1.) * (ok ? 0.5 : 1.);
                } else {
                    const double2 c = mg_interpolate(j, k, L.nx, L.ny, L.rx, L.ry, nyc, [&](int q) { return xc[q]; });
                    cx = c.x, cy = c.y;
                }
                double2 xf = x[i];
                if (d.x != 0.) xf.x = fma(wt, cx, xf.x);
                if (d.y != 0.) xf.y = fma(wt, cy, xf.y);
                x[i] = xf;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; r++) {  // w = x + omega D^-1 (b - K x)
            const int i = tid + r * nt;
            if (i < nn) {
                const double2 di = r ? dB : dA;
                const double2 q = grid_apply_pairs<RAGGED>(nxn, nyn, nel, tab, i, [&](int m) { return reinterpret_cast<const double2 *>(Ml)[m]; },
                                               [&](int j) { return x[j]; }, L.rx, L.ry);
                const double2 bi = b[i], xi = x[i];
                w[i] = make_double2(fma(omega * di.x, bi.x - q.x, xi.x), fma(omega * di.y, bi.y - q.y, xi.y));
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; r++) {  // x = w + omega D^-1 (b - K w)
            const int i = tid + r * nt;
            if (i < nn) {
                const double2 di = r ? dB : dA;
                const double2 q = grid_apply_pairs<RAGGED>(nxn, nyn, nel, tab, i, [&](int m) { return reinterpret_cast<const double2 *>(Ml)[m]; },
                                               [&](int j) { return w[j]; }, L.rx, L.ry);
                const double2 bi = b[i], wi = w[i];
                x[i] = make_double2(fma(omega * di.x, bi.x - q.x, wi.x), fma(omega * di.y, bi.y - q.y, wi.y));
            }
        }
        __syncthreads();
    }
    {
        const MgLevDev &L = slev[l0];
        for (int i = tid; i < L.nnode; i += nt) L.x[i] = X[L.tail_off + i];
    }
}

// PCG update without the Jacobi z (the V-cycle computes z): x += alpha p; r -= alpha q; partial r.r
__global__ void __launch_bounds__(BLOCK)
k_cg_update_mg(int nnode, const double2 *__restrict__ p, const double2 *__restrict__ q,
               const double2 *__restrict__ dinv, double2 *__restrict__ x, double2 *__restrict__ r, const double *__restrict__ part_pq,
               int npart_pq, const double *__restrict__ part_rz, int npart_prev, double *__restrict__ part_rr_out,
               CgScalars *__restrict__ sc, int own_lo, int own_hi /* nodes whose r.r this rank sums (strip: owned columns) */)
{
    __shared__ double sh[BLOCK / 64];
    if (sc->done) return;
    double pq, rz;
    if (npart_pq == npart_prev) {  // both scalars in one round of loads / barriers
        const double *const arr[2] = {part_pq, part_rz};
        double o[2];
        sum_partials_n<2>(arr, npart_pq, o);
        pq = o[0];
        rz = o[1];
    } else {
        pq = sum_partials(part_pq, npart_pq, sh);
        rz = sum_partials(part_rz, npart_prev, sh);
    }
    if (!(pq > 0.)) {  // breakdown: stop and keep the last iterate (the host falls back to Jacobi-PCG)
        if (blockIdx.x == 0 && threadIdx.x == 0) sc->done = 2;
        return;
    }
    const double alpha = rz / pq;
    double a_rr = 0.;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nnode; i += gridDim.x * BLOCK) {
        const double2 pi = p[i], qi = q[i], di = dinv[i];
        double2 xi = x[i], ri = r[i];
        xi.x = fma(alpha, pi.x, xi.x);
        xi.y = fma(alpha, pi.y, xi.y);
        ri.x = (di.x != 0.) ? fma(-alpha, qi.x, ri.x) : 0.;
        ri.y = (di.y != 0.) ? fma(-alpha, qi.y, ri.y) : 0.;
        x[i] = xi;
        r[i] = ri;
        if (i >= own_lo && i < own_hi) a_rr = fma(ri.x, ri.x, fma(ri.y, ri.y, a_rr));
    }
    const double t2 = block_sum(a_rr, sh);
    if (threadIdx.x == 0) part_rr_out[blockIdx.x] = t2;
}

// partial sums of r.z over the nodes [own_lo, own_hi) (all nodes; the owned columns of a strip)
__global__ void __launch_bounds__(BLOCK)
k_dot_rz(int own_lo, int own_hi, const double2 *__restrict__ r, const double2 *__restrict__ z, double *part_rz_out)
{
    __shared__ double sh[BLOCK / 64];
    double acc = 0.;
    for (int i = own_lo + blockIdx.x * BLOCK + threadIdx.x; i < own_hi; i += gridDim.x * BLOCK) {
        const double2 a = r[i], b = z[i];
        acc = fma(a.x, b.x, fma(a.y, b.y, acc));
    }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) part_rz_out[blockIdx.x] = t;
}

// Strip-local engine: dst[c * dst_n + i] = (off <= i < off + cnt) ? src[c * src_n + i - off + soff] : 0 for c < ncomp -- the owned
// window of a strip's level-Ld array placed into the zero-filled array of the replicated coarse grid; one all-reduce over
// the ranks then completes it (x + 0 + ... + 0 = x exactly).  Node / element columns are contiguous, so a window is a range.
__global__ void __launch_bounds__(BLOCK)
k_strip_pack(int ncomp, size_t dst_n, size_t src_n, size_t off, size_t cnt, size_t soff, const double *__restrict__ src,
             double *__restrict__ dst)
{
    const size_t tot = (size_t)ncomp * dst_n;
    for (size_t t = blockIdx.x * (size_t)BLOCK + threadIdx.x; t < tot; t += (size_t)gridDim.x * BLOCK) {
        const size_t c = t / dst_n, i = t - c * dst_n;
        dst[t] = (i >= off && i < off + cnt) ? src[c * src_n + (i - off) + soff] : 0.;
    }
}

}  // namespace plfx
