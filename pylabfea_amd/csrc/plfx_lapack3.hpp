// plfx_lapack3.hpp -- the order in which LAPACK's dgeev returns the eigenpairs of a symmetric 3 x 3 matrix, replayed.
//
// Why this exists.  basic.sig_princ (reference basic.py:153-175) calls np.linalg.eig -- the GENERAL eigen-solver dgeev, not the
// symmetric one -- and re-orders the principal stresses through the row-wise argmax of |eigenvector components| (:157-170), a
// rule whose result depends on the ORDER of dgeev's output whenever that argmax table is a 3-cycle.  dgeev has no ordering
// rule: the eigenvalues come out in the order the QR sweeps of dlahqr deflate them (an extreme one first in
// 99.4 % of random matrices, otherwise nothing short of replaying it).  Principal-stress materials (sdim = 3) inside
// Material.response (material.py:250-340) see every sub-step's stress through that function, so on states with out-of-plane
// shear the device has to take LAPACK's decisions: this file restates, for n = 3 and the call dgeev('N', 'V'), the chain
//     dgebal('B')  ->  dgehrd (dgehd2: one Householder reflector)  ->  dorghr  ->  dhseqr('S', 'V') = dlahqr (Francis
//     double-shift QR with the Ahues-Tisseur deflation test, dlanv2 standardisation of 2 x 2 blocks)  ->  dtrevc3 (back
//     substitution + back-transformation)  ->  dgebak  ->  unit 2-norm columns
// following the published reference-LAPACK 3.11/3.12 algorithms (what OpenBLAS 0.3.29 -- the library under the numpy that wrote
// tests/golden/princ_general.npz -- ships; not present under /root/reference, which only calls numpy).  Decisions are taken on
// the same quantities in the same order.  Deflation tests of nearly diagonal matrices are decided by the last bits of the
// iterates (8 of 20 000 such matrices came out in another order with plain IEEE arithmetic), so the arithmetic of the BLAS
// kernels underneath is followed too, where it was found to differ from source order -- by bitwise comparison, stage by stage,
// with the very library numpy loads (tools/probes/lapack3_stages.py calls its dgebal / dgehrd / dorghr / dlahqr / dtrevc3 through
// ctypes): the fused multiply-adds of the Haswell dgemv_n / daxpy / dgemm / drot kernels inside dgehd2, dorg2r, dlahqr's 2 x 2 standardisation and dtrevc3 (explicit
// fma() below; everything else with contraction off, as gfortran compiled it), and dnrm2 in x87 extended precision (here: a
// double-double sum of squares + corrected square root, i.e. the correctly rounded norm, which the x87 result equals except for
// double-rounding cases).  With that, eigenvalues are bit-identical to numpy's on every matrix tried (2000 of 2000) and the
// eigenvectors to the last bit or one off; what remains open are exact argmax ties between eigenvector components that are
// equal in exact arithmetic (small-integer matrices with eigenvectors like (1,1,1)/sqrt 3: 1 of 20 000 such matrices; 89 before
// the kernels' arithmetic was followed).
// Not restated: dgeev's rescaling of matrices whose largest entry is outside [1e-292, 1e292] (stresses never are), the
// overflow guards of dlarfg / dlanv2 / dlaln2, complex output (a symmetric matrix whose dlanv2 block is classified complex
// by round-off makes numpy return complex arrays, on which the reference fails too).
//
// Host + device: the same source is the device routine of the PRINC3 / SVC3 kernels (plfx_device.hpp: sig_princ_dev on states
// with s23 or s13 != 0) and a host entry of the C-ABI (plfx_sig_princ_host), which the CPU suite holds against numpy itself.
#pragma once
#include <math.h>

namespace plfx {
namespace lapack3 {

#define PLFX_L3_HD __host__ __device__

PLFX_L3_HD inline double l3_sign(double a, double b) { return copysign(fabs(a), b); }   // Fortran SIGN(a, b)

PLFX_L3_HD inline double l3_lapy2(double x, double y)
{
    const double xa = fabs(x), ya = fabs(y);
    const double w = xa > ya ? xa : ya, z = xa > ya ? ya : xa;
    if (z == 0.) return w;
    const double q = z / w;
    return w * sqrt(1. + q * q);
}

// dlanv2: Schur factorisation of a real 2 x 2 block in standardised form; returns cs, sn and the eigenvalue pair
PLFX_L3_HD inline void l3_lanv2(double &a, double &b, double &c, double &d, double &rt1r, double &rt1i, double &rt2r, double &rt2i,
                                double &cs, double &sn)
{
#pragma clang fp contract(off)
    const double eps = 2.220446049250313e-16;   // dlamch('P') = eps * base
    const double multpl = 4.;
    if (c == 0.) {
        cs = 1.;
        sn = 0.;
    } else if (b == 0.) {   // swap rows and columns
        cs = 0.;
        sn = 1.;
        const double t = d;
        d = a;
        a = t;
        b = -c;
        c = 0.;
    } else if ((a - d) == 0. && l3_sign(1., b) != l3_sign(1., c)) {
        cs = 1.;
        sn = 0.;
    } else {
        double temp = a - d;
        double p = 0.5 * temp;
        const double bcmax = fmax(fabs(b), fabs(c));
        const double bcmis = fmin(fabs(b), fabs(c)) * l3_sign(1., b) * l3_sign(1., c);
        double scale = fmax(fabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= multpl * eps) {   // real eigenvalues
            z = p + l3_sign(sqrt(scale) * sqrt(z), p);
            a = d + z;
            d = d - (bcmax / z) * bcmis;
            const double tau = l3_lapy2(c, z);
            cs = z / tau;
            sn = c / tau;
            b = b - c;
            c = 0.;
        } else {   // complex, or real (almost) equal eigenvalues: make the diagonal elements equal
            const double sigma = b + c;
            p = 0.5 * temp;
            const double tau = l3_lapy2(sigma, temp);
            cs = sqrt(0.5 * (1. + fabs(sigma) / tau));
            sn = -(p / (tau * cs)) * l3_sign(1., sigma);
            const double aa = a * cs + b * sn, bb = -a * sn + b * cs, cc = c * cs + d * sn, dd = -c * sn + d * cs;
            a = aa * cs + cc * sn;
            b = bb * cs + dd * sn;
            c = -aa * sn + cc * cs;
            d = -bb * sn + dd * cs;
            temp = 0.5 * (a + d);
            a = temp;
            d = temp;
            if (c != 0.) {
                if (b != 0.) {
                    if (l3_sign(1., b) == l3_sign(1., c)) {   // real eigenvalues: reduce to upper triangular form
                        const double sab = sqrt(fabs(b)), sac = sqrt(fabs(c));
                        p = l3_sign(sab * sac, c);
                        const double tau2 = 1. / sqrt(fabs(b + c));
                        a = temp + p;
                        d = temp - p;
                        b = b - c;
                        c = 0.;
                        const double cs1 = sab * tau2, sn1 = sac * tau2;
                        temp = cs * cs1 - sn * sn1;
                        sn = cs * sn1 + sn * cs1;
                        cs = temp;
                    }
                } else {
                    b = -c;
                    c = 0.;
                    temp = cs;
                    cs = -sn;
                    sn = temp;
                }
            }
        }
    }
    rt1r = a;
    rt2r = d;
    if (c == 0.) {
        rt1i = 0.;
        rt2i = 0.;
    } else {
        rt1i = sqrt(fabs(b)) * sqrt(fabs(c));
        rt2i = -rt1i;
    }
}

// sqrt(a^2 + b^2 + c^2) correctly rounded (double-double sum of squares, one corrected square root): OpenBLAS's x86-64 dnrm2
// kernel works in x87 extended precision and rounds once at the end
PLFX_L3_HD inline double l3_nrm2(double a, double b, double c)
{
    const double pa = a * a, pb = b * b, pc = c * c;
    const double ea = fma(a, a, -pa), eb = fma(b, b, -pb), ec = fma(c, c, -pc);
    // two-sum of the three leading parts
    double s = pa + pb;
    double bb = s - pa;
    double err = (pa - (s - bb)) + (pb - bb);
    double s2 = s + pc;
    bb = s2 - s;
    err += (s - (s2 - bb)) + (pc - bb);
    const double lo = err + ((ea + eb) + ec);
    const double hi = s2 + lo;
    const double lo2 = lo - (hi - s2);
    if (!(hi > 0.)) return 0.;
    const double r = sqrt(hi);
    // hi + lo2 - r^2, exactly enough
    const double res = fma(-r, r, hi) + lo2;
    return r + res / (2. * r);
}

// dlarfg for the two sizes that occur (n = 2, 3): v(0) = alpha (in/out: beta), v(1..n-1) = x (in/out: the reflector's tail)
PLFX_L3_HD inline double l3_larfg(int n, double *v)
{
#pragma clang fp contract(off)
    if (n <= 1) return 0.;
    double xnorm = (n == 2) ? fabs(v[1]) : l3_nrm2(v[1], v[2], 0.);   // dnrm2 of one / two entries
    if (xnorm == 0.) return 0.;
    const double alpha = v[0];
    const double beta = -l3_sign(l3_lapy2(alpha, xnorm), alpha);
    const double tau = (beta - alpha) / beta;
    const double sc = 1. / (alpha - beta);
    for (int i = 1; i < n; i++) v[i] *= sc;
    v[0] = beta;
    return tau;
}

// Eigenvalues wr[3] in dgeev's order and right eigenvectors V (V[i * 3 + k] = component i of eigenvector k, unit 2-norm) of
// the symmetric matrix [[s0 s5 s4] [s5 s1 s3] [s4 s3 s2]].  Returns 0, or 1 when the iteration limit of dlahqr was reached /
// a 2 x 2 block stayed complex (wr then holds the real parts).
PLFX_L3_HD inline int dgeev3(const double *s, double *wr, double *V)
{
#pragma clang fp contract(off)
    double h[3][3] = {{s[0], s[5], s[4]}, {s[5], s[1], s[3]}, {s[4], s[3], s[2]}};
#define H(i, j) h[(i)-1][(j)-1]
    int info = 0;
    // ---------------------------------------------------------------- dgebal('B'): permutation (1-based k, l, scale)
    int k = 1, l = 3;
    int scale[4] = {0, 1, 2, 3};
    bool all_isolated = false;
    for (bool again = true; again && !all_isolated;) {   // rows isolating an eigenvalue are pushed down
        again = false;
        for (int j = l; j >= 1; j--) {
            bool iso = true;
            for (int i = 1; i <= l; i++)
                if (i != j && H(j, i) != 0.) { iso = false; break; }
            if (!iso) continue;
            scale[l] = j;
            if (j != l) {
                for (int r = 1; r <= l; r++) { const double t = H(r, j); H(r, j) = H(r, l); H(r, l) = t; }
                for (int c = k; c <= 3; c++) { const double t = H(j, c); H(j, c) = H(l, c); H(l, c) = t; }
            }
            if (l == 1) { all_isolated = true; break; }
            l--;
            again = true;
            break;
        }
    }
    if (!all_isolated)
        for (bool again = true; again;) {   // columns isolating an eigenvalue are pushed left
            again = false;
            for (int j = k; j <= l; j++) {
                bool iso = true;
                for (int i = k; i <= l; i++)
                    if (i != j && H(i, j) != 0.) { iso = false; break; }
                if (!iso) continue;
                scale[k] = j;
                if (j != k) {
                    for (int r = 1; r <= l; r++) { const double t = H(r, j); H(r, j) = H(r, k); H(r, k) = t; }
                    for (int c = k; c <= 3; c++) { const double t = H(j, c); H(j, c) = H(k, c); H(k, c) = t; }
                }
                k++;
                again = true;
                break;
            }
        }
    const int ilo = all_isolated ? 1 : k, ihi = all_isolated ? 1 : l;
    // (scaling loop of dgebal: column and row norms of a symmetric block are equal -> f = 1, nothing is scaled)
    // ---------------------------------------------------------------- dgehrd (dgehd2) + dorghr
    double z[3][3] = {{1., 0., 0.}, {0., 1., 0.}, {0., 0., 1.}};
#define Z(i, j) z[(i)-1][(j)-1]
    if (ilo == 1 && ihi == 3) {
        double v[2] = {H(2, 1), H(3, 1)};
        const double tau = l3_larfg(2, v);
        if (tau != 0.) {
            const double v2 = v[1];
            // H := H (I - tau v v^T) on rows 1..3, columns 2..3
            // (fused multiply-adds exactly where OpenBLAS 0.3.29's Haswell dgemv_n / daxpy tails fuse them: found by bitwise
            // comparison with numpy's own library, tools/probes/lapack3_stages.py -- 300 of 300 matrices identical to the last bit;
            // the transposed product below is not fused there)
            for (int r = 1; r <= 3; r++) {
                const double w = fma(H(r, 3), v2, H(r, 2));
                H(r, 2) = fma(w, -tau, H(r, 2));
                H(r, 3) = fma(w, -tau * v2, H(r, 3));
            }
            // H := (I - tau v v^T) H on rows 2..3, columns 2..3
            for (int c = 2; c <= 3; c++) {
                const double w = H(2, c) + H(3, c) * v2;
                const double t = -tau * w;
                H(2, c) = H(2, c) + t;
                H(3, c) = fma(v2, t, H(3, c));
            }
            // dorghr / dorg2r: Q(2:3, 2:3) = I - tau v v^T built column by column
            const double w = v2;            // (0, 1) . (1, v2)
            Z(2, 3) = 0. + (-tau * w);
            Z(3, 3) = fma(v2, -tau * w, 1.);
            Z(3, 2) = -tau * v2;
            Z(2, 2) = 1. - tau;
        }
        H(2, 1) = v[0];
        H(3, 1) = 0.;   // (dlahqr: "clear out the trash")
    }
    // ---------------------------------------------------------------- dhseqr -> dlahqr(wantt, wantz, iloz = ilo, ihiz = ihi)
    double wi[4] = {0., 0., 0., 0.};
    for (int i = 1; i <= 3; i++)
        if (i < ilo || i > ihi) wr[i - 1] = H(i, i);
    if (ilo == ihi)
        wr[ilo - 1] = H(ilo, ilo);
    else {
        const double safmin = 2.2250738585072014e-308, ulp = 2.220446049250313e-16;
        const int nh = ihi - ilo + 1;
        const double smlnum = safmin * ((double)nh / ulp);
        const int i1 = 1, i2 = 3;
        const int itmax = 30 * (nh > 10 ? nh : 10);
        int kdefl = 0;
        int i = ihi;
        while (i >= ilo) {
            int ll = ilo;
            bool converged = false;
            for (int its = 0; its <= itmax; its++) {
                int kk;
                for (kk = i; kk >= ll + 1; kk--) {   // look for a single small subdiagonal element
                    if (fabs(H(kk, kk - 1)) <= smlnum) break;
                    double tst = fabs(H(kk - 1, kk - 1)) + fabs(H(kk, kk));
                    if (tst == 0.) {
                        if (kk - 2 >= ilo) tst += fabs(H(kk - 1, kk - 2));
                        if (kk + 1 <= ihi) tst += fabs(H(kk + 1, kk));
                    }
                    if (fabs(H(kk, kk - 1)) <= ulp * tst) {   // Ahues & Tisseur
                        const double ab = fmax(fabs(H(kk, kk - 1)), fabs(H(kk - 1, kk)));
                        const double ba = fmin(fabs(H(kk, kk - 1)), fabs(H(kk - 1, kk)));
                        const double aa = fmax(fabs(H(kk, kk)), fabs(H(kk - 1, kk - 1) - H(kk, kk)));
                        const double bb = fmin(fabs(H(kk, kk)), fabs(H(kk - 1, kk - 1) - H(kk, kk)));
                        const double ss = aa + ab;
                        if (ba * (ab / ss) <= fmax(smlnum, ulp * (bb * (aa / ss)))) break;
                    }
                }
                ll = kk;
                if (ll > ilo) H(ll, ll - 1) = 0.;
                if (ll >= i - 1) { converged = true; break; }
                kdefl++;
                double h11, h21, h12, h22;
                if (kdefl % 20 == 0) {   // exceptional shifts
                    const double ss = fabs(H(i, i - 1)) + fabs(H(i - 1, i - 2));
                    h11 = 0.75 * ss + H(i, i);
                    h12 = -0.4375 * ss;
                    h21 = ss;
                    h22 = h11;
                } else if (kdefl % 10 == 0) {
                    const double ss = fabs(H(ll + 1, ll)) + fabs(H(ll + 2, ll + 1));
                    h11 = 0.75 * ss + H(ll, ll);
                    h12 = -0.4375 * ss;
                    h21 = ss;
                    h22 = h11;
                } else {
                    h11 = H(i - 1, i - 1);
                    h21 = H(i, i - 1);
                    h12 = H(i - 1, i);
                    h22 = H(i, i);
                }
                double rt1r, rt1i, rt2r, rt2i;
                double ss = fabs(h11) + fabs(h12) + fabs(h21) + fabs(h22);
                if (ss == 0.) {
                    rt1r = rt1i = rt2r = rt2i = 0.;
                } else {
                    h11 /= ss;
                    h21 /= ss;
                    h12 /= ss;
                    h22 /= ss;
                    const double tr = (h11 + h22) / 2.;
                    const double det = (h11 - tr) * (h22 - tr) - h12 * h21;
                    const double rtdisc = sqrt(fabs(det));
                    if (det >= 0.) {   // complex conjugate shifts
                        rt1r = tr * ss;
                        rt2r = rt1r;
                        rt1i = rtdisc * ss;
                        rt2i = -rt1i;
                    } else {           // real shifts: the one closer to h22, twice
                        rt1r = tr + rtdisc;
                        rt2r = tr - rtdisc;
                        if (fabs(rt1r - h22) <= fabs(rt2r - h22)) {
                            rt1r = rt1r * ss;
                            rt2r = rt1r;
                        } else {
                            rt2r = rt2r * ss;
                            rt1r = rt2r;
                        }
                        rt1i = rt2i = 0.;
                    }
                }
                // two consecutive small subdiagonal elements
                int m;
                double v[3];
                for (m = i - 2; m >= ll; m--) {
                    double h21s = fabs(H(m + 1, m));
                    double sv = fabs(H(m, m) - rt2r) + fabs(rt2i) + h21s;
                    h21s = H(m + 1, m) / sv;
                    v[0] = h21s * H(m, m + 1) + (H(m, m) - rt1r) * ((H(m, m) - rt2r) / sv) - rt1i * (rt2i / sv);
                    v[1] = h21s * (H(m, m) + H(m + 1, m + 1) - rt1r - rt2r);
                    v[2] = h21s * H(m + 2, m + 1);
                    sv = fabs(v[0]) + fabs(v[1]) + fabs(v[2]);
                    v[0] /= sv;
                    v[1] /= sv;
                    v[2] /= sv;
                    if (m == ll) break;
                    const double h00 = fabs(H(m - 1, m - 1)), h11a = fabs(H(m, m)), h22a = fabs(H(m + 1, m + 1));
                    if (fabs(H(m, m - 1)) * (fabs(v[1]) + fabs(v[2])) <= ulp * fabs(v[0]) * (h00 + h11a + h22a)) break;
                }
                // double-shift QR step
                for (int kq = m; kq <= i - 1; kq++) {
                    const int nr = (3 < i - kq + 1) ? 3 : i - kq + 1;
                    if (kq > m)
                        for (int q = 0; q < nr; q++) v[q] = H(kq + q, kq - 1);
                    const double t1 = l3_larfg(nr, v);
                    if (kq > m) {
                        H(kq, kq - 1) = v[0];
                        H(kq + 1, kq - 1) = 0.;
                        if (kq < i - 1) H(kq + 2, kq - 1) = 0.;
                    } else if (m > ll) {
                        H(kq, kq - 1) = H(kq, kq - 1) * (1. - t1);
                    }
                    const double v2 = v[1], t2 = t1 * v2;
                    if (nr == 3) {
                        const double v3 = v[2], t3 = t1 * v3;
                        for (int j = kq; j <= i2; j++) {
                            const double sum = H(kq, j) + v2 * H(kq + 1, j) + v3 * H(kq + 2, j);
                            H(kq, j) = H(kq, j) - sum * t1;
                            H(kq + 1, j) = H(kq + 1, j) - sum * t2;
                            H(kq + 2, j) = H(kq + 2, j) - sum * t3;
                        }
                        const int jmax = (kq + 3 < i) ? kq + 3 : i;
                        for (int j = i1; j <= jmax; j++) {
                            const double sum = H(j, kq) + v2 * H(j, kq + 1) + v3 * H(j, kq + 2);
                            H(j, kq) = H(j, kq) - sum * t1;
                            H(j, kq + 1) = H(j, kq + 1) - sum * t2;
                            H(j, kq + 2) = H(j, kq + 2) - sum * t3;
                        }
                        for (int j = ilo; j <= ihi; j++) {
                            const double sum = Z(j, kq) + v2 * Z(j, kq + 1) + v3 * Z(j, kq + 2);
                            Z(j, kq) = Z(j, kq) - sum * t1;
                            Z(j, kq + 1) = Z(j, kq + 1) - sum * t2;
                            Z(j, kq + 2) = Z(j, kq + 2) - sum * t3;
                        }
                    } else if (nr == 2) {
                        for (int j = kq; j <= i2; j++) {
                            const double sum = H(kq, j) + v2 * H(kq + 1, j);
                            H(kq, j) = H(kq, j) - sum * t1;
                            H(kq + 1, j) = H(kq + 1, j) - sum * t2;
                        }
                        for (int j = i1; j <= i; j++) {
                            const double sum = H(j, kq) + v2 * H(j, kq + 1);
                            H(j, kq) = H(j, kq) - sum * t1;
                            H(j, kq + 1) = H(j, kq + 1) - sum * t2;
                        }
                        for (int j = ilo; j <= ihi; j++) {
                            const double sum = Z(j, kq) + v2 * Z(j, kq + 1);
                            Z(j, kq) = Z(j, kq) - sum * t1;
                            Z(j, kq + 1) = Z(j, kq + 1) - sum * t2;
                        }
                    }
                }
            }
            if (!converged) {   // iteration limit: leave with what is there
                info = 1;
                for (int q = ilo; q <= i; q++) wr[q - 1] = H(q, q);
                break;
            }
            if (ll == i) {
                wr[i - 1] = H(i, i);
                wi[i] = 0.;
            } else {   // ll == i - 1: a pair -- standardise the 2 x 2 block
                double cs, sn, r1i, r2i;
                l3_lanv2(H(i - 1, i - 1), H(i - 1, i), H(i, i - 1), H(i, i), wr[i - 2], r1i, wr[i - 1], r2i, cs, sn);
                wi[i - 1] = r1i;
                wi[i] = r2i;
                if (r1i != 0.) info = 1;
                if (i2 > i)   // drot on the rest of the two rows
                    for (int j = i + 1; j <= i2; j++) {
                        const double t = fma(cs, H(i - 1, j), sn * H(i, j));   // (drot kernel: fused like this, see the header)
                        H(i, j) = fma(cs, H(i, j), -(sn * H(i - 1, j)));
                        H(i - 1, j) = t;
                    }
                for (int j = i1; j <= i - 2; j++) {   // ... and of the two columns above the block
                    const double t = fma(cs, H(j, i - 1), sn * H(j, i));
                    H(j, i) = fma(cs, H(j, i), -(sn * H(j, i - 1)));
                    H(j, i - 1) = t;
                }
                for (int j = ilo; j <= ihi; j++) {
                    const double t = fma(cs, Z(j, i - 1), sn * Z(j, i));
                    Z(j, i) = fma(cs, Z(j, i), -(sn * Z(j, i - 1)));
                    Z(j, i - 1) = t;
                }
            }
            kdefl = 0;
            i = ll - 1;
        }
    }
    // ---------------------------------------------------------------- dtrevc3('R', 'B'): eigenvectors of T, times Z
    {
        const double ulp = 2.220446049250313e-16, smlnum = 2.2250738585072014e-308 * (3. / ulp);
        for (int ki = 3; ki >= 1; ki--) {
            double work[4];
            const double wk = H(ki, ki);
            const double smin = fmax(ulp * fabs(wk), smlnum);
            work[ki] = 1.;
            for (int q = 1; q < ki; q++) work[q] = -H(q, ki);
            for (int j = ki - 1; j >= 1; j--) {
                // dlaln2, 1 x 1 real: (T(j,j) - wr) x = work(j)
                double csr = H(j, j) - wk;
                double cn = fabs(csr);
                if (cn < smin) { csr = smin; cn = smin; }
                const double x = work[j] / csr;
                work[j] = x;
                for (int q = 1; q < j; q++) work[q] = work[q] + (-x) * H(q, j);   // daxpy
            }
            // back-transformation as the blocked dtrevc3 does it: one GEMM, i.e. per entry a fused accumulation over j = 1..ki
            // starting from the first product; then the largest |component| is scaled to 1
            double col[4];
            for (int r = 1; r <= 3; r++) {
                double y = Z(r, 1) * work[1];
                for (int j = 2; j <= ki; j++) y = fma(Z(r, j), work[j], y);
                col[r] = y;
            }
            const double amax = fmax(fabs(col[1]), fmax(fabs(col[2]), fabs(col[3])));
            const double remax = 1. / amax;
            for (int r = 1; r <= 3; r++) V[(r - 1) * 3 + (ki - 1)] = col[r] * remax;
        }
    }
    // ---------------------------------------------------------------- dgebak('B', 'R'): undo the permutation (rows of V)
    for (int ii = 1; ii <= 3; ii++) {
        int i = ii;
        if (i >= ilo && i <= ihi) continue;
        if (i < ilo) i = ilo - ii;
        const int kx = scale[i];
        if (kx == i) continue;
        for (int c = 0; c < 3; c++) {
            const double t = V[(i - 1) * 3 + c];
            V[(i - 1) * 3 + c] = V[(kx - 1) * 3 + c];
            V[(kx - 1) * 3 + c] = t;
        }
    }
    // ---------------------------------------------------------------- unit 2-norm (dgeev: scl = 1 / dnrm2, dscal)
    for (int c = 0; c < 3; c++) {
        const double nrm = l3_nrm2(V[c], V[3 + c], V[6 + c]);
        if (!(nrm > 0.)) continue;
        const double scl = 1. / nrm;
        V[c] *= scl;
        V[3 + c] *= scl;
        V[6 + c] *= scl;
    }
#undef H
#undef Z
    return info;
}

// basic.sig_princ (basic.py:107-179) on one Voigt stress: principal stresses in the reference's axis-tracking order --
// dgeev's eigenpairs, then rows grouped by the column of their largest |component| (argmax: first maximum), the eigenvalues
// re-indexed by that ROW list (:157-172)
PLFX_L3_HD inline int sig_princ_lapack3(const double *s, double *sp)
{
    double w[3], V[9];
    const int info = dgeev3(s, w, V);
    int iev[3];
    for (int i = 0; i < 3; i++) {
        int kx = 0;
        for (int c = 1; c < 3; c++)
            if (fabs(V[i * 3 + c]) > fabs(V[i * 3 + kx])) kx = c;
        iev[i] = kx;
    }
    int j[3] = {0, 0, 0}, n = 0;
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 3; i++)
            if (iev[i] == c && n < 3) j[n++] = i;
    sp[0] = w[j[0]];
    sp[1] = w[j[1]];
    sp[2] = w[j[2]];
    return info;
}

}  // namespace lapack3
}  // namespace plfx
