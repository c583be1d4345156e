/*
 * plfx_oracle.c — CPU oracle for the pyLabFEA hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference algorithm; every function cites the reference
 * lines it follows (paths relative to /root/reference/src/pylabfea).  Deliberately
 * un-optimised and written in the order of the Python statements so that it can be
 * audited side by side with the reference.  Pinned by tests/test_oracle_golden.py against
 * vectors dumped from the imported reference (oracle/gen_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include "plfx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define YF_TOLERANCE 5.e-3 /* basic.py:26 */

/* ------------------------------------------------------------------ small linear algebra */
static void matvec6(const double A[36], const double x[6], double y[6])
{
    for (int i = 0; i < 6; i++) {
        double s = 0.;
        for (int j = 0; j < 6; j++) s += A[i * 6 + j] * x[j];
        y[i] = s;
    }
}

static double dot6(const double a[6], const double b[6])
{
    double s = 0.;
    for (int i = 0; i < 6; i++) s += a[i] * b[i];
    return s;
}

/* Gauss-Jordan inverse with partial pivoting of the leading n x n block (n <= 3) */
static void inv_block(const double *A, int lda, int n, double *Ai /* n*n */)
{
    double w[3][6];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            w[i][j] = A[i * lda + j];
            w[i][n + j] = (i == j) ? 1. : 0.;
        }
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++)
            if (fabs(w[r][c]) > fabs(w[p][c])) p = r;
        if (p != c)
            for (int j = 0; j < 2 * n; j++) {
                double t = w[c][j];
                w[c][j] = w[p][j];
                w[p][j] = t;
            }
        double d = w[c][c];
        for (int j = 0; j < 2 * n; j++) w[c][j] /= d;
        for (int r = 0; r < n; r++)
            if (r != c) {
                double f = w[r][c];
                for (int j = 0; j < 2 * n; j++) w[r][j] -= f * w[c][j];
            }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ai[i * n + j] = w[i][n + j];
}

/* cyclic Jacobi eigen-decomposition of a symmetric 3x3 matrix: G = V diag(w) V^T */
static void jacobi3(const double G[9], double w[3], double V[9])
{
    double A[9];
    memcpy(A, G, sizeof(A));
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1. : 0.;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
        if (off == 0.) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = A[p * 3 + q];
                if (apq == 0.) continue;
                double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2. * apq);
                double t = (theta >= 0. ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
                double c = 1. / sqrt(t * t + 1.), s = t * c;
                for (int k = 0; k < 3; k++) { /* A <- A J */
                    double akp = A[k * 3 + p], akq = A[k * 3 + q];
                    A[k * 3 + p] = c * akp - s * akq;
                    A[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) { /* A <- J^T A */
                    double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                    A[p * 3 + k] = c * apk - s * aqk;
                    A[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    w[0] = A[0];
    w[1] = A[4];
    w[2] = A[8];
}

/* minimum-norm least-squares solution of a(3x6) x = b, as numpy.linalg.lstsq(rcond=None):
 * singular values <= eps*max(M,N)*s_max are treated as zero (material.py:331). */
static void lstsq_3x6(const double a[18], const double b[3], double x[6])
{
    double G[9], w[3], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0.;
            for (int k = 0; k < 6; k++) s += a[i * 6 + k] * a[j * 6 + k];
            G[i * 3 + j] = s;
        }
    jacobi3(G, w, V);
    double wmax = fmax(fmax(w[0], w[1]), w[2]);
    double smax = wmax > 0. ? sqrt(wmax) : 0.;
    double cut = 2.220446049250313e-16 * 6. * smax;
    double y[3] = {0., 0., 0.};
    for (int k = 0; k < 3; k++) {
        double sk = w[k] > 0. ? sqrt(w[k]) : 0.;
        if (sk <= cut || sk == 0.) continue;
        double proj = V[0 * 3 + k] * b[0] + V[1 * 3 + k] * b[1] + V[2 * 3 + k] * b[2];
        for (int i = 0; i < 3; i++) y[i] += V[i * 3 + k] * proj / w[k];
    }
    for (int j = 0; j < 6; j++) x[j] = a[0 * 6 + j] * y[0] + a[1 * 6 + j] * y[1] + a[2 * 6 + j] * y[2];
}

/* ------------------------------------------------------------------ basic.py helpers */
void plfo_sig_dev(const double sig[6], double out[6]) /* basic.py:316-324 */
{
    double p = (sig[0] + sig[1] + sig[2]) / 3.;
    out[0] = sig[0] - p;
    out[1] = sig[1] - p;
    out[2] = sig[2] - p;
    out[3] = sig[3];
    out[4] = sig[4];
    out[5] = sig[5];
}

double plfo_eps_eq(const double e[6]) /* basic.py:350-352 */
{
    double n = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
    double s = e[3] * e[3] + e[4] * e[4] + e[5] * e[5];
    return sqrt(2. * (n + 0.5 * s) / 3.);
}

/* basic.py:107-179 */
void plfo_sig_princ(const double s[6], double sp[3])
{
    if (s[3] == 0. && s[4] == 0.) {
        /* plane state: the 2x2 block decouples; the eigenvector closer to axis 0 carries the larger
         * eigenvalue iff s0 >= s1 (argmax of |ev|, first maximum on ties, basic.py:156) */
        double mean = 0.5 * (s[0] + s[1]);
        double R = sqrt(0.25 * (s[0] - s[1]) * (s[0] - s[1]) + s[5] * s[5]);
        if (s[5] == 0.) {
            sp[0] = s[0];
            sp[1] = s[1];
        } else if (s[0] >= s[1]) {
            sp[0] = mean + R;
            sp[1] = mean - R;
        } else {
            sp[0] = mean - R;
            sp[1] = mean + R;
        }
        sp[2] = s[2];
        return;
    }
    double G[9] = {s[0], s[5], s[4], s[5], s[1], s[3], s[4], s[3], s[2]}, w[3], V[9];
    jacobi3(G, w, V);
    for (int i = 0; i < 3; i++) {
        int k = 0;
        for (int c = 1; c < 3; c++)
            if (fabs(V[i * 3 + c]) > fabs(V[i * 3 + k])) k = c;
        sp[i] = w[k];
    }
}

static void eig3_values(const double s[6], double w[3])
{
    double G[9] = {s[0], s[5], s[4], s[5], s[1], s[3], s[4], s[3], s[2]}, V[9];
    jacobi3(G, w, V);
}

static double calc_seqB(const plfo_material *m, const double sv[6]) /* material.py:678-702 */
{
    const double *b = m->barlat;
    double sd[6], st1[6], st2[6], p1[3], p2[3];
    plfo_sig_dev(sv, sd);
    /* Bar_m1 / Bar_m2 (material.py:2578-2591) */
    st1[0] = -b[0] * sd[1] - b[1] * sd[2];
    st1[1] = -b[2] * sd[0] - b[3] * sd[2];
    st1[2] = -b[4] * sd[0] - b[5] * sd[1];
    st1[3] = b[6] * sd[3];
    st1[4] = b[7] * sd[4];
    st1[5] = b[8] * sd[5];
    st2[0] = -b[9] * sd[1] - b[10] * sd[2];
    st2[1] = -b[11] * sd[0] - b[12] * sd[2];
    st2[2] = -b[13] * sd[0] - b[14] * sd[1];
    st2[3] = b[15] * sd[3];
    st2[4] = b[16] * sd[4];
    st2[5] = b[17] * sd[5];
    eig3_values(st1, p1);
    eig3_values(st2, p2);
    double a = m->barlat_exp, acc = 0.;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) acc += pow(fabs(p1[i] - p2[j]), a);
    return pow(0.25 * acc, 1. / a);
}

/* ------------------------------------------------------------------ material.py */
double plfo_calc_seq(const plfo_material *m, const double sig[6]) /* material.py:636-673 */
{
    if (m->kind == PLFO_TRESCA) { /* material.py:630-632 */
        double sp[3];
        eig3_values(sig, sp);
        return fmax(fmax(sp[0], sp[1]), sp[2]) - fmin(fmin(sp[0], sp[1]), sp[2]);
    }
    if (m->kind == PLFO_BARLAT) return calc_seqB(m, sig); /* material.py:633-637 */
    if (m->kind == PLFO_PRINC3 || m->kind == PLFO_SVC3) { /* material.py:662-673, 3-parameter Hill on principal stresses */
        double sp[3];
        plfo_sig_princ(sig, sp);
        double d12 = sp[0] - sp[1], d23 = sp[1] - sp[2], d31 = sp[2] - sp[0];
        double I2 = 0.5 * (m->hill[0] * d12 * d12 + m->hill[1] * d23 * d23 + m->hill[2] * d31 * d31);
        double I1 = (sig[0] * m->dp[0] + sig[1] * m->dp[1] + sig[2] * m->dp[2]) / 3.;
        return sqrt(I2) + I1;
    }
    double hp[6] = {1., 1., 1., 1., 1., 1.};
    double d0[3] = {0., 0., 0.};
    if (m->kind != PLFO_ELASTIC) { /* self.sy is not None */
        for (int i = 0; i < 6; i++) hp[i] = m->hill[i];
        for (int i = 0; i < 3; i++) d0[i] = m->dp[i];
    }
    double I1 = (sig[0] * d0[0] + sig[1] * d0[1] + sig[2] * d0[2]) / 3.;
    double I2 = hp[0] * (sig[0] - sig[1]) * (sig[0] - sig[1]) +
                hp[1] * (sig[1] - sig[2]) * (sig[1] - sig[2]) +
                hp[2] * (sig[2] - sig[0]) * (sig[2] - sig[0]) +
                6. * hp[3] * sig[3] * sig[3] + 6. * hp[4] * sig[4] * sig[4] +
                6. * hp[5] * sig[5] * sig[5];
    I2 *= 0.5;
    return sqrt(I2) + I1;
}

double plfo_get_sflow(const plfo_material *m, const double epl[6]) /* material.py:992-997 */
{
    return m->sy + plfo_eps_eq(epl) * m->khard;
}

/* basic.py:68-104 sig_polar_ang on principal stresses */
static const double A_VEC[3] = {0.816496580927726, -0.408248290463863, -0.408248290463863}; /* [1,-.5,-.5]/sqrt(1.5) */
static const double B_VEC[3] = {0., 0.7071067811865476, -0.7071067811865476};               /* [0,.5,-.5]*sqrt(2) */

static double polar_ang(const double sp[3])
{
    double hyd = (sp[0] + sp[1] + sp[2]) / 3.;
    double dev[3] = {sp[0] - hyd, sp[1] - hyd, sp[2] - hyd};
    double vn = sqrt(dev[0] * dev[0] + dev[1] * dev[1] + dev[2] * dev[2]);
    if (vn < 1.e-4) vn = 1.;
    double dsa = (dev[0] * A_VEC[0] + dev[1] * A_VEC[1] + dev[2] * A_VEC[2]) / vn;
    double dsb = (dev[0] * B_VEC[0] + dev[1] * B_VEC[1] + dev[2] * B_VEC[2]) / vn;
    return atan2(dsb, dsa);
}

/* create_scaled_input, sdim == 3 (material.py:2331-2333) on the principal stresses of sig */
static void svc3_features(const plfo_material *m, const double sp[3], double x[2])
{
    double d12 = sp[0] - sp[1], d23 = sp[1] - sp[2], d31 = sp[2] - sp[0];
    double seq = sqrt(0.5 * (d12 * d12 + d23 * d23 + d31 * d31)); /* sig_eq_j2, basic.py:58-62 */
    x[0] = seq / m->scale_seq - 1.;
    x[1] = polar_ang(sp) / 3.141592653589793;
}

static double svc3_decision(const plfo_material *m, const double sig[6])
{
    double sp[3], x[2], f = 0.;
    plfo_sig_princ(sig, sp);
    svc3_features(m, sp, x);
    for (int k = 0; k < m->nsv; k++) {
        const double *v = m->sv + (size_t)k * m->ndof;
        double hh = (x[0] - v[0]) * (x[0] - v[0]) + (x[1] - v[1]) * (x[1] - v[1]);
        f += m->dual[k] * exp(-m->gamma * hh);
    }
    return f + m->intercept;
}

static double svc_decision(const plfo_material *m, const double sig[6]) /* material.py:398-405, 2330-2340 */
{
    double s[6], x[6];
    if (m->dev_only)
        plfo_sig_dev(sig, s);
    else
        memcpy(s, sig, sizeof(s));
    for (int i = 0; i < 6; i++) x[i] = s[i] / m->scale_seq;
    double f = 0.;
    for (int k = 0; k < m->nsv; k++) {
        const double *v = m->sv + (size_t)k * m->ndof;
        double hh = 0.;
        for (int i = 0; i < 6; i++) hh += (x[i] - v[i]) * (x[i] - v[i]);
        f += m->dual[k] * exp(-m->gamma * hh);
    }
    return f + m->intercept;
}

/* create_scaled_input with work-hardening features (material.py:2334-2346): 6 stress features, epl / scale_wh, then
 * accumulated strain, max. stress / scale_seq and flag, which response / Model.solve leave at their defaults 0 */
static void wh_features(const plfo_material *m, const double sig[6], const double epl[6], double x[15])
{
    double s[6];
    if (m->dev_only)
        plfo_sig_dev(sig, s);
    else
        memcpy(s, sig, sizeof(s));
    for (int i = 0; i < 6; i++) x[i] = s[i] / m->scale_seq;
    for (int i = 0; i < 6; i++) x[6 + i] = (epl ? epl[i] : 0.) / m->scale_wh;
    x[12] = 0.;
    x[13] = 0. / m->scale_seq;
    x[14] = 0.;
}

static double svc_wh_decision(const plfo_material *m, const double sig[6], const double epl[6])
{
    double x[15], f = 0.;
    wh_features(m, sig, epl, x);
    for (int k = 0; k < m->nsv; k++) {
        const double *v = m->sv + (size_t)k * 15;
        double hh = 0.;
        for (int i = 0; i < 15; i++) hh += (x[i] - v[i]) * (x[i] - v[i]);
        f += m->dual[k] * exp(-m->gamma * hh);
    }
    return f + m->intercept;
}

double plfo_calc_yf(const plfo_material *m, const double sig[6], const double epl[6]) /* material.py:378-411 */
{
    if (m->kind == PLFO_SVC_WH) return svc_wh_decision(m, sig, epl);
    if (m->kind == PLFO_SVC6) return svc_decision(m, sig);
    if (m->kind == PLFO_SVC3) return svc3_decision(m, sig);
    return plfo_calc_seq(m, sig) - plfo_get_sflow(m, epl);
}

/* EXTENSION (test infrastructure for the product's opt-in Barlat normal; the reference raises for Barlat, material.py:822):
 * analytic gradient of calc_seqB.  phi = sum_ij |S'_i - S''_j|^a, seq = (phi/4)^(1/a); dS_i = n_i.dT.n_i for the
 * eigenvector n_i of the transformed deviator T; chain rule through Bar_m1 / Bar_m2 (material.py:2578-2591) and sig_dev. */
static void barlat_fgrad(const plfo_material *m, const double sv[6], double a[6])
{
    const double *b = m->barlat;
    double sd[6], T[2][6], w[2][3], V[2][9];
    plfo_sig_dev(sv, sd);
    for (int q = 0; q < 2; q++) {
        const double *c = b + 9 * q;
        T[q][0] = -c[0] * sd[1] - c[1] * sd[2];
        T[q][1] = -c[2] * sd[0] - c[3] * sd[2];
        T[q][2] = -c[4] * sd[0] - c[5] * sd[1];
        T[q][3] = c[6] * sd[3];
        T[q][4] = c[7] * sd[4];
        T[q][5] = c[8] * sd[5];
        double G[9] = {T[q][0], T[q][5], T[q][4], T[q][5], T[q][1], T[q][3], T[q][4], T[q][3], T[q][2]};
        jacobi3(G, w[q], V[q]);
    }
    const double ex = m->barlat_exp;
    double phi = 0., dw[2][3] = {{0., 0., 0.}, {0., 0., 0.}};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double d = w[0][i] - w[1][j];
            phi += pow(fabs(d), ex);
            if (d != 0.) {
                double t = ex * pow(fabs(d), ex - 1.) * (d > 0. ? 1. : -1.);
                dw[0][i] += t;
                dw[1][j] -= t;
            }
        }
    for (int i = 0; i < 6; i++) a[i] = 0.;
    if (!(phi > 0.)) return;
    double h[6] = {0., 0., 0., 0., 0., 0.};
    for (int q = 0; q < 2; q++) {
        const double *c = b + 9 * q;
        double g[6] = {0., 0., 0., 0., 0., 0.};
        for (int i = 0; i < 3; i++) { /* eigenvector i = column i of V */
            double n0 = V[q][0 * 3 + i], n1 = V[q][1 * 3 + i], n2 = V[q][2 * 3 + i];
            g[0] += dw[q][i] * n0 * n0;
            g[1] += dw[q][i] * n1 * n1;
            g[2] += dw[q][i] * n2 * n2;
            g[3] += dw[q][i] * 2. * n1 * n2; /* Voigt 23 */
            g[4] += dw[q][i] * 2. * n0 * n2; /* Voigt 13 */
            g[5] += dw[q][i] * 2. * n0 * n1; /* Voigt 12 */
        }
        h[0] += -c[2] * g[1] - c[4] * g[2];
        h[1] += -c[0] * g[0] - c[5] * g[2];
        h[2] += -c[1] * g[0] - c[3] * g[1];
        h[3] += c[6] * g[3];
        h[4] += c[7] * g[4];
        h[5] += c[8] * g[5];
    }
    const double hm = (h[0] + h[1] + h[2]) / 3.;
    const double seq = pow(0.25 * phi, 1. / ex), sc = seq / (ex * phi);
    for (int i = 0; i < 3; i++) a[i] = sc * (h[i] - hm);
    for (int i = 3; i < 6; i++) a[i] = sc * h[i];
}

void plfo_calc_fgrad_wh(const plfo_material *m, const double sig[6], const double epl[6], double a[6], double *kh_raw)
{   /* material.py:797-814 for whdat materials, N = 1 */
    double x[15], dK[15];
    wh_features(m, sig, epl, x);
    for (int i = 0; i < 15; i++) dK[i] = 0.;
    for (int k = 0; k < m->nsv; k++) { /* grad_rbf, :768-778 */
        const double *v = m->sv + (size_t)k * 15;
        double hv[15], hh = 0.;
        for (int i = 0; i < 15; i++) {
            hv[i] = x[i] - v[i];
            hh += hv[i] * hv[i];
        }
        double kk = exp(-m->gamma * hh);
        for (int i = 0; i < 15; i++) dK[i] += m->dual[k] * (kk * (-2. * m->gamma * hv[i]));
    }
    for (int i = 0; i < 6; i++) a[i] = dK[i] / m->scale_seq; /* :807 */
    double hk = 0.;
    for (int i = 6; i < 12; i++) hk -= dK[i] * m->scale_seq / m->scale_wh; /* :808-809, ind_wh = 6 */
    if (kh_raw) *kh_raw = hk;
    ((plfo_material *)m)->khard = hk < 0. ? 0. : hk; /* :811-814: self.khard = ... (the caller owns a private copy) */
}

/* calc_fgrad(sig, epl=epl) as epl_dot / C_tan call it (material.py:1050, 1082) */
static void fgrad_epl(const plfo_material *m, const double sig[6], const double epl[6], double a[6])
{
    if (m->kind == PLFO_SVC_WH)
        plfo_calc_fgrad_wh(m, sig, epl, a, NULL);
    else
        plfo_calc_fgrad(m, sig, a);
}

void plfo_calc_fgrad(const plfo_material *m, const double sig[6], double a[6]) /* material.py:765-845 */
{
    if (m->kind == PLFO_SVC_WH) { /* epl = None -> zeros (:734-735) */
        plfo_calc_fgrad_wh(m, sig, NULL, a, NULL);
        return;
    }
    if (m->kind == PLFO_BARLAT) {
        barlat_fgrad(m, sig, a);
        return;
    }
    if (m->kind == PLFO_SVC6) {
        double s[6], x[6], dK[6] = {0., 0., 0., 0., 0., 0.};
        if (m->dev_only)
            plfo_sig_dev(sig, s);
        else
            memcpy(s, sig, sizeof(s));
        for (int i = 0; i < 6; i++) x[i] = s[i] / m->scale_seq;
        for (int k = 0; k < m->nsv; k++) { /* grad_rbf, material.py:768-778 */
            const double *v = m->sv + (size_t)k * m->ndof;
            double hv[6], hh = 0.;
            for (int i = 0; i < 6; i++) {
                hv[i] = x[i] - v[i];
                hh += hv[i] * hv[i];
            }
            double kk = exp(-m->gamma * hh);
            for (int i = 0; i < 6; i++) dK[i] += m->dual[k] * (kk * (-2. * m->gamma * hv[i]));
        }
        for (int i = 0; i < 6; i++) a[i] = dK[i] / m->scale_seq; /* material.py:807 */
        return;
    }
    if (m->kind == PLFO_SVC3) {
        /* gradient of the 2-feature SVC through the Jacobian of (seq, theta) (material.py:779-807), written
         * into the normal Voigt components like every sdim == 3 normal (material.py:1044-1047) */
        double sp[3], x[2], dK1 = 0.;
        plfo_sig_princ(sig, sp);
        svc3_features(m, sp, x);
        for (int k = 0; k < m->nsv; k++) {
            const double *v = m->sv + (size_t)k * m->ndof;
            double h0 = x[0] - v[0], h1 = x[1] - v[1];
            double kk = exp(-m->gamma * (h0 * h0 + h1 * h1));
            dK1 += m->dual[k] * (kk * (-2. * m->gamma * h1));
        }
        double hyd = (sp[0] + sp[1] + sp[2]) / 3.;
        double dev[3] = {sp[0] - hyd, sp[1] - hyd, sp[2] - hyd};
        double vn = sqrt(dev[0] * dev[0] + dev[1] * dev[1] + dev[2] * dev[2]) * sqrt(1.5);
        if (vn > 0.1) {
            double cr = sp[0] * A_VEC[0] + sp[1] * A_VEC[1] + sp[2] * A_VEC[2];
            double ci = sp[0] * B_VEC[0] + sp[1] * B_VEC[1] + sp[2] * B_VEC[2];
            double n2 = cr * cr + ci * ci;
            for (int i = 0; i < 3; i++) /* J[:,0] = 3 dev/vn ; J[:,1] = Re(-i((a+ib)/sc - dseqds/vn)) */
                a[i] = 3. * dev[i] / vn + (B_VEC[i] * cr - A_VEC[i] * ci) / n2 * dK1;
        } else {
            for (int i = 0; i < 3; i++) a[i] = 1. + dK1; /* J = ones */
        }
        a[3] = a[4] = a[5] = 0.;
        return;
    }
    if (m->kind == PLFO_PRINC3) {
        /* epl_dot / C_tan with sdim == 3 (material.py:1044-1047, 1079-1081): the gradient w.r.t. the
         * principal stresses is written into the normal Voigt components, shear components stay 0 */
        double sp[3], sq;
        plfo_sig_princ(sig, sp);
        double d12 = sp[0] - sp[1], d23 = sp[1] - sp[2], d31 = sp[2] - sp[0];
        sq = sqrt(0.5 * (m->hill[0] * d12 * d12 + m->hill[1] * d23 * d23 + m->hill[2] * d31 * d31)) +
             (sp[0] * m->dp[0] + sp[1] * m->dp[1] + sp[2] * m->dp[2]) / 3.;
        double pm = (sp[0] + sp[1] + sp[2]) / 3.;
        double s0 = sp[0] - pm, s1 = sp[1] - pm, s2 = sp[2] - pm;
        double g0 = m->hill[0], g1 = m->hill[1], g2 = m->hill[2];
        a[0] = ((g0 + g2) * s0 - g0 * s1 - g2 * s2) / (2. * sq) + m->dp[0] / 3.;
        a[1] = ((g1 + g0) * s1 - g0 * s0 - g1 * s2) / (2. * sq) + m->dp[1] / 3.;
        a[2] = ((g2 + g1) * s2 - g2 * s0 - g1 * s1) / (2. * sq) + m->dp[2] / 3.;
        a[3] = a[4] = a[5] = 0.;
        return;
    }
    double h0 = m->hill[0], h1 = m->hill[1], h2 = m->hill[2];
    double d3[3] = {m->dp[0] / 3., m->dp[1] / 3., m->dp[2] / 3.}; /* ones*drucker/3 (:833); the lhs variant cannot
                                                                    * run in the reference (`if self.lhs:` on an array, :642) */
    double seq = plfo_calc_seq(m, sig);
    double sd[6];
    plfo_sig_dev(sig, sd);
    a[0] = ((h0 + h2) * sd[0] - h0 * sd[1] - h2 * sd[2]) / (2. * seq) + d3[0];
    a[1] = ((h1 + h0) * sd[1] - h0 * sd[0] - h1 * sd[2]) / (2. * seq) + d3[1];
    a[2] = ((h2 + h1) * sd[2] - h2 * sd[0] - h1 * sd[1]) / (2. * seq) + d3[2];
    a[3] = 3. * m->hill[3] * sd[3] / seq;
    a[4] = 3. * m->hill[4] * sd[4] / seq;
    a[5] = 3. * m->hill[5] * sd[5] / seq;
}

/* ---- scipy.optimize.brentq (scipy 1.15.3, scipy/optimize/Zeros/brentq.c; Brent 1973) */
double plfo_brentq(plfo_fn f, void *ctx, double xa, double xb, double xtol, double rtol,
                   int maxiter, int *converged)
{
    double xpre = xa, xcur = xb;
    double xblk = 0., fpre, fcur, fblk = 0., spre = 0., scur = 0., sbis;
    double delta, stry, dpre, dblk;
    *converged = 1;
    fpre = f(xpre, ctx);
    fcur = f(xcur, ctx);
    if (fpre == 0.) return xpre;
    if (fcur == 0.) return xcur;
    if (signbit(fpre) == signbit(fcur)) {
        *converged = 0;
        return 0.;
    }
    for (int i = 0; i < maxiter; i++) {
        if (fpre != 0. && fcur != 0. && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre;
            fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur;
            xcur = xblk;
            xblk = xpre;
            fpre = fcur;
            fcur = fblk;
            fblk = fpre;
        }
        delta = (xtol + rtol * fabs(xcur)) / 2.;
        sbis = (xblk - xcur) / 2.;
        if (fcur == 0. || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                dpre = (fpre - fcur) / (xpre - xcur);
                dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            double lim = fmin(fabs(spre), 3. * fabs(sbis) - delta);
            if (2. * fabs(stry) < lim) {
                spre = scur;
                scur = stry;
            } else {
                spre = sbis;
                scur = sbis;
            }
        } else {
            spre = sbis;
            scur = sbis;
        }
        xpre = xcur;
        fpre = fcur;
        if (fabs(scur) > delta)
            xcur += scur;
        else
            xcur += (sbis > 0. ? delta : -delta);
        fcur = f(xcur, ctx);
    }
    *converged = 0;
    return xcur;
}

typedef struct {
    const plfo_material *m;
    const double *su;
    const double *epl;
} yloc_ctx;

static double yloc_scalar(double x, void *vctx) /* material.py:547-574 find_yloc_scalar */
{
    yloc_ctx *c = (yloc_ctx *)vctx;
    double s[6];
    for (int i = 0; i < 6; i++) s[i] = x * c->su[i];
    return plfo_calc_yf(c->m, s, c->epl);
}

double plfo_ML_full_yf_ld(const plfo_material *m, const double sig[6], const double epl_in[6],
                          const double *ld, int *status) /* material.py:414-516; ld == NULL: ray through sig */
{
    static const double zero6[6] = {0., 0., 0., 0., 0., 0.};
    const double *epl = epl_in ? epl_in : zero6;
    int st = 0;
    double seq = plfo_calc_seq(m, sig);
    double sflow = plfo_get_sflow(m, epl);
    double yf;
    if (seq < 0.01 && !ld) {
        yf = seq - 0.85 * sflow; /* :445-448 */
    } else {
        double su[6];
        if (!ld) {
            for (int i = 0; i < 6; i++) su[i] = sig[i] / seq; /* :452 */
        } else { /* :454-462 loading direction -> unit stress (sdim = 6; for sdim = 3 only ld[0:3] counts) */
            const int sd = (m->kind == PLFO_SVC3) ? 3 : 6;
            double hh = 0.;
            for (int i = 0; i < sd; i++) hh += ld[i] * ld[i];
            hh = sqrt(hh);
            if (hh < 1.e-3) { /* :456-461 inconsistent ld: x direction */
                for (int i = 0; i < 6; i++) su[i] = 0.;
                su[0] = sqrt(1.5);
            } else {
                for (int i = 0; i < 6; i++) su[i] = (i < sd) ? ld[i] * sqrt(1.5) / hh : 0.;
            }
        }
        double x0 = sflow;
        if (su[0] * su[1] < -1.e-5) x0 *= 0.5; /* :468-473 (tresca off) */
        double x1 = x0;
        yloc_ctx c = {m, su, epl};
        while (yloc_scalar(x0, &c) >= 0. && x0 > 0.01) x0 *= 0.98; /* :475-480 */
        while (yloc_scalar(x1, &c) < 0. && x1 < 5. * sflow) x1 *= 1.02; /* :481-486 */
        double f0 = yloc_scalar(x0, &c), f1 = yloc_scalar(x1, &c);
        if (f0 * f1 > 0.) { /* :495-499 */
            if (status) *status = 1;
            return seq - 0.85 * sflow;
        }
        int conv;
        double xs = plfo_brentq(yloc_scalar, &c, x0, x1, 1.e-5, 4. * 2.220446049250313e-16, 100, &conv);
        if (conv && xs < 4. * sflow) {
            yf = seq - xs * plfo_calc_seq(m, su); /* :507 */
        } else {
            yf = seq - 0.85 * sflow; /* :510 */
            st = 2;
        }
    }
    if (status) *status = st;
    return yf;
}

double plfo_ML_full_yf(const plfo_material *m, const double sig[6], const double epl_in[6], int *status)
{
    return plfo_ML_full_yf_ld(m, sig, epl_in, NULL, status); /* material.py:414-516, ld=None */
}

/* batched ld variant (calc_scf, model.py:1049-1053) */
void plfo_full_yf_ld_batch(const plfo_material *m, int n, const double *sig, const double *epl, const double *ld,
                           double *out)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n; i++) out[i] = plfo_ML_full_yf_ld(m, sig + 6 * i, epl ? epl + 6 * i : NULL, ld, NULL);
}

void plfo_epl_dot(const plfo_material *m, const double sig[6], const double epl[6],
                  const double Cel[36], const double deps[6], double pdot[6]) /* material.py:1032-1055 */
{
    double ds[6], st[6];
    matvec6(Cel, deps, ds);
    for (int i = 0; i < 6; i++) st[i] = sig[i] + ds[i];
    double yfun = plfo_calc_yf(m, st, epl);
    if (yfun <= YF_TOLERANCE) { /* ABSOLUTE tolerance, material.py:1041 */
        for (int i = 0; i < 6; i++) pdot[i] = 0.;
        return;
    }
    double a[6], ca[6];
    fgrad_epl(m, sig, epl, a); /* calc_fgrad(sig, epl=epl): a work-hardening SVC overwrites khard here (:1050) */
    matvec6(Cel, a, ca);
    double hh = dot6(a, ca) + m->khard;
    double cd[6];
    matvec6(Cel, deps, cd);
    double lam = dot6(a, cd) / hh;
    for (int i = 0; i < 6; i++) pdot[i] = lam * a[i];
}

static void C_tan_epl(const plfo_material *m, const double sig[6], const double epl[6], const double Cel[36], double Ct[36])
{   /* C_tan(sig, CV, epl=epl) as response calls it (material.py:278, 299, 1076-1086) */
    double a[6], ca[6];
    fgrad_epl(m, sig, epl, a);
    matvec6(Cel, a, ca);
    double hh = dot6(a, ca) + m->khard;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) Ct[i * 6 + j] = Cel[i * 6 + j] - ca[i] * ca[j] / hh;
}

void plfo_C_tan(const plfo_material *m, const double sig[6], const double Cel[36], double Ct[36]) /* material.py:1076-1086 */
{
    double a[6], ca[6];
    plfo_calc_fgrad(m, sig, a);
    matvec6(Cel, a, ca);
    double hh = dot6(a, ca) + m->khard;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) Ct[i * 6 + j] = Cel[i * 6 + j] - ca[i] * ca[j] / hh;
}

static double resp_yf(const plfo_material *m, const double s[6], const double e[6])
{
    if (m->kind == PLFO_SVC6 || m->kind == PLFO_SVC3 || m->kind == PLFO_SVC_WH) return plfo_ML_full_yf(m, s, e, NULL); /* material.py:249-252 */
    return plfo_calc_yf(m, s, e);
}

int plfo_response(const plfo_material *m, const double sig_in[6], const double epl[6],
                  const double deps[6], const double CV[36],
                  double *fy, double sig[6], double depl[6], double Ct[36]) /* material.py:207-346 */
{
    const int maxit = 50;
    double dsig[6], tmp[6], fy1;
    int niter = 0;
    memcpy(sig, sig_in, 6 * sizeof(double)); /* :241 copy */
    for (int i = 0; i < 6; i++) depl[i] = 0.;
    double toler = YF_TOLERANCE * plfo_get_sflow(m, epl); /* :243 */
    matvec6(CV, deps, dsig);                              /* :244 */
    double st_scal = 1.;
    for (int i = 0; i < 6; i++) tmp[i] = sig[i] + dsig[i];
    fy1 = resp_yf(m, tmp, epl); /* :249-252 */
    if (fy1 < toler) {
        for (int i = 0; i < 6; i++) sig[i] += dsig[i];
        memcpy(Ct, CV, 36 * sizeof(double));
    } else {
        double deps_r[6];
        double fy0 = plfo_calc_yf(m, sig, epl); /* :259 */
        if (fy0 < -0.15) {
            if (m->kind == PLFO_SVC6 || m->kind == PLFO_SVC3 || m->kind == PLFO_SVC_WH) fy0 = plfo_ML_full_yf(m, sig, NULL, NULL); /* :265 */
            st_scal += fy0 / plfo_calc_seq(m, dsig);                             /* :266 */
            double deps_el[6], ds_el[6];
            for (int i = 0; i < 6; i++) deps_el[i] = deps[i] * (1. - st_scal);
            matvec6(CV, deps_el, ds_el);
            for (int i = 0; i < 6; i++) sig[i] += ds_el[i];
            for (int i = 0; i < 36; i++) Ct[i] = CV[i] * (1. - st_scal);
            for (int i = 0; i < 6; i++) deps_r[i] = deps[i] - deps_el[i];
        } else {
            memcpy(deps_r, deps, sizeof(deps_r));
            for (int i = 0; i < 36; i++) Ct[i] = 0.;
        }
        double ddepl[6], T[36], eplt[6];
        plfo_epl_dot(m, sig, epl, CV, deps_r, ddepl); /* :277 */
        C_tan_epl(m, sig, epl, CV, T);                    /* :278 */
        for (int i = 0; i < 6; i++) eplt[i] = epl[i] + depl[i] + ddepl[i];
        matvec6(T, deps_r, dsig);
        for (int i = 0; i < 6; i++) tmp[i] = sig[i] + dsig[i];
        fy1 = resp_yf(m, tmp, eplt); /* :282-285 */
        int nsteps;
        if (fy1 > toler) {
            for (int i = 0; i < 6; i++) deps_r[i] /= maxit;
            nsteps = maxit;
        } else {
            nsteps = 1;
        }
        for (niter = 0; niter < nsteps; niter++) { /* :295 */
            plfo_epl_dot(m, sig, epl, CV, deps_r, ddepl);
            C_tan_epl(m, sig, epl, CV, T);
            for (int i = 0; i < 6; i++) eplt[i] = epl[i] + depl[i] + ddepl[i];
            matvec6(T, deps_r, dsig);
            for (int i = 0; i < 6; i++) sig[i] += dsig[i];
            fy1 = resp_yf(m, sig, eplt);
            if (fy1 > toler) { /* :310-342 radial scale-back */
                double SV[36];
                for (int i = 0; i < 36; i++) SV[i] = 0.;
                int nb = (CV[2 * 6 + 2] > 1.) ? 3 : 2;
                double hh[9];
                inv_block(CV, 6, nb, hh);
                for (int i = 0; i < nb; i++)
                    for (int j = 0; j < nb; j++) SV[i * 6 + j] = hh[i * nb + j];
                for (int i = 3; i < 6; i++)
                    if (CV[i * 6 + i] > 1.) SV[i * 6 + i] = 1. / CV[i * 6 + i];
                double sq = plfo_calc_seq(m, sig);
                for (int i = 0; i < 6; i++) dsig[i] = sig[i] * fy1 / sq;
                for (int i = 0; i < 6; i++) sig[i] -= dsig[i];
                double sd[6];
                matvec6(SV, dsig, sd);
                for (int i = 0; i < 6; i++) ddepl[i] += sd[i];
                for (int i = 0; i < 6; i++) eplt[i] = epl[i] + depl[i] + ddepl[i];
                double a[18] = {deps_r[0], 0., 0., 0., deps_r[2], deps_r[1],
                                0., deps_r[1], 0., deps_r[2], 0., deps_r[0],
                                0., 0., deps_r[2], deps_r[1], deps_r[0], 0.};
                double x[6];
                lstsq_3x6(a, dsig, x);
                T[0 * 6 + 0] -= x[0];
                T[0 * 6 + 1] -= x[5];
                T[0 * 6 + 2] -= x[4];
                T[1 * 6 + 0] -= x[5];
                T[1 * 6 + 1] -= x[1];
                T[1 * 6 + 2] -= x[3];
                T[2 * 6 + 0] -= x[4];
                T[2 * 6 + 1] -= x[3];
                T[2 * 6 + 2] -= x[2];
                fy1 = resp_yf(m, sig, eplt);
            }
            for (int i = 0; i < 36; i++) Ct[i] += T[i] * st_scal / nsteps; /* :343 */
            for (int i = 0; i < 6; i++) depl[i] += ddepl[i];                /* :344 */
        }
        niter = nsteps - 1; /* Python leaves niter at the last loop index (:345) */
    }
    *fy = fy1;
    return niter;
}

/* ------------------------------------------------------------------ batched drivers */
void plfo_seq_batch(const plfo_material *m, int n, const double *sig, double *seq)
{
    for (int i = 0; i < n; i++) seq[i] = plfo_calc_seq(m, sig + 6 * (size_t)i);
}

void plfo_fgrad_batch(const plfo_material *m, int n, const double *sig, double *a)
{
    for (int i = 0; i < n; i++) plfo_calc_fgrad(m, sig + 6 * (size_t)i, a + 6 * (size_t)i);
}

void plfo_yf_batch(const plfo_material *m, int n, const double *sig, const double *epl, double *yf)
{
    for (int i = 0; i < n; i++) yf[i] = plfo_calc_yf(m, sig + 6 * (size_t)i, epl + 6 * (size_t)i);
}

void plfo_full_yf_batch(const plfo_material *m, int n, const double *sig, const double *epl,
                        double *yf, int *status)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n; i++)
        yf[i] = plfo_ML_full_yf(m, sig + 6 * (size_t)i, epl ? epl + 6 * (size_t)i : NULL,
                                status ? status + i : NULL);
}

void plfo_response_batch(const plfo_material *mats, int n, const int *mat_id,
                         const double *sig, const double *epl, const double *deps,
                         const double *CV, double *fy, double *sig_out, double *depl,
                         double *ct, int *nsteps, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) {
        int mid = mat_id ? mat_id[i] : 0;
        size_t o = 6 * (size_t)i;
        if (mats[mid].kind == PLFO_ELASTIC) { /* model.py:1341 skips elastic materials */
            fy[i] = 0.;
            nsteps[i] = 0;
            continue;
        }
        nsteps[i] = plfo_response(&mats[mid], sig + o, epl + o, deps + o, CV + 36 * (size_t)mid,
                                  fy + i, sig_out + o, depl + o, ct + 36 * (size_t)i);
    }
}

/* ------------------------------------------------------------------ model.py element */
void plfo_calc_Bmat(double lx, double ly, double x, double y, int planestress,
                    const double CV[36], double E, double nu, double B[48]) /* model.py:475-501 */
{
    for (int i = 0; i < 48; i++) B[i] = 0.;
    double xi1 = 2. * x / lx - 1.;
    double xi2 = 2. * y / ly - 1.;
    double hxm = 0.125 * (1. - xi1) / ly;
    double hym = 0.125 * (1. - xi2) / lx;
    double hxp = 0.125 * (1. + xi1) / ly;
    double hyp = 0.125 * (1. + xi2) / lx;
    B[0 * 8 + 0] = -hym;
    B[0 * 8 + 2] = -hyp;
    B[0 * 8 + 4] = hym;
    B[0 * 8 + 6] = hyp;
    B[1 * 8 + 1] = -hxm;
    B[1 * 8 + 3] = hxm;
    B[1 * 8 + 5] = -hxp;
    B[1 * 8 + 7] = hxp;
    B[5 * 8 + 0] = -hxm;
    B[5 * 8 + 1] = -hym;
    B[5 * 8 + 2] = hxm;
    B[5 * 8 + 3] = -hyp;
    B[5 * 8 + 4] = -hxp;
    B[5 * 8 + 5] = hym;
    B[5 * 8 + 6] = hxp;
    B[5 * 8 + 7] = hyp;
    if (planestress) { /* :498-501 */
        for (int j = 0; j < 8; j++) {
            double h0 = 0., h1 = 0.;
            for (int k = 0; k < 6; k++) {
                h0 += CV[0 * 6 + k] * B[k * 8 + j];
                h1 += CV[1 * 6 + k] * B[k * 8 + j];
            }
            B[2 * 8 + j] = -nu * (h0 + h1) / E;
        }
    }
}

static void gauss_point(double lx, double ly, int i, double *x, double *y) /* model.py:339-346 */
{
    double cpos = sqrt(1. / 3.);
    double sx = ((i / 2) % 2 == 0) ? 1. : -1.;
    double sy = (i % 2 == 0) ? 1. : -1.;
    *x = 0.5 * (1. + sx * cpos) * lx;
    *y = 0.5 * (1. + sy * cpos) * ly;
}

void plfo_calc_Kel(double lx, double ly, double thick, int planestress, const double CV[36],
                   double E, double nu, const double D[36], double Kel[64]) /* model.py:365-370 */
{
    double Jac = lx * ly * thick * 4.; /* model.py:316-322, 340 */
    double sum[64];
    for (int i = 0; i < 64; i++) sum[i] = 0.;
    for (int g = 0; g < 4; g++) {
        double x, y, B[48], DB[48];
        gauss_point(lx, ly, g, &x, &y);
        plfo_calc_Bmat(lx, ly, x, y, planestress, CV, E, nu, B);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 8; j++) {
                double s = 0.;
                for (int k = 0; k < 6; k++) s += D[i * 6 + k] * B[k * 8 + j];
                DB[i * 8 + j] = s;
            }
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++) {
                double s = 0.;
                for (int k = 0; k < 6; k++) s += B[k * 8 + i] * DB[k * 8 + j];
                sum[i * 8 + j] += s;
            }
    }
    for (int i = 0; i < 64; i++) Kel[i] = Jac * 1. * sum[i];
}

void plfo_strain(double lx, double ly, int planestress, const double CV[36], double E, double nu,
                 const double ue[8], double eps[6]) /* model.py:387-411 */
{
    for (int i = 0; i < 6; i++) eps[i] = 0.;
    for (int g = 0; g < 4; g++) {
        double x, y, B[48];
        gauss_point(lx, ly, g, &x, &y);
        plfo_calc_Bmat(lx, ly, x, y, planestress, CV, E, nu, B);
        for (int i = 0; i < 6; i++) {
            double s = 0.;
            for (int j = 0; j < 8; j++) s += B[i * 8 + j] * ue[j];
            eps[i] += 1. * s;
        }
    }
}

void plfo_kel_batch(int nel, const double *lxy, const int *mat_id, double thick, int planestress,
                    const double *CV, const double *E, const double *nu, const double *D, double *Kel)
{
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nel; e++) {
        int m = mat_id[e];
        plfo_calc_Kel(lxy[2 * e], lxy[2 * e + 1], thick, planestress, CV + 36 * (size_t)m, E[m], nu[m],
                      D + 36 * (size_t)e, Kel + 64 * (size_t)e);
    }
}

void plfo_strain_batch(int nel, const int *conn, const double *lxy, const int *mat_id, int planestress,
                       const double *CV, const double *E, const double *nu, const double *u, double *eps)
{
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nel; e++) {
        int m = mat_id[e];
        double ue[8];
        for (int a = 0; a < 4; a++) { /* Element.node_num, model.py:372-385 */
            ue[2 * a] = u[2 * (size_t)conn[4 * e + a]];
            ue[2 * a + 1] = u[2 * (size_t)conn[4 * e + a] + 1];
        }
        plfo_strain(lxy[2 * e], lxy[2 * e + 1], planestress, CV + 36 * (size_t)m, E[m], nu[m], ue,
                    eps + 6 * (size_t)e);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Jacobi-preconditioned CG on a CSR matrix, projected onto the free DOFs (free[i] != 0): the iterative counterpart of
 * the reference's `Kred = K[ind][:, ind]; np.linalg.solve(Kred, df[ind])` (model.py:1028-1033, 1291).  Rows and
 * columns of prescribed DOFs are skipped, x stays 0 there.  OpenMP over rows; the same-host CPU baseline of
 * BASELINE.md section 3 ("CSR assembly, Jacobi-PCG, OpenMP material sweep").  Returns the iteration count. */
int plfo_pcg_csr(int n, const int *indptr, const int *indices, const double *data, const double *b,
                 const unsigned char *free_mask, double *x, double rtol, int maxit, int nthreads, double *relres)
{
    double *r = (double *)malloc(sizeof(double) * n), *z = (double *)malloc(sizeof(double) * n);
    double *p = (double *)malloc(sizeof(double) * n), *q = (double *)malloc(sizeof(double) * n);
    double *dinv = (double *)malloc(sizeof(double) * n);
    int it = 0;
    double bb = 0., rr = 0., rz = 0.;
#ifdef _OPENMP
    const int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    const int nt = 1;
    (void)nthreads;
#endif
#pragma omp parallel for num_threads(nt) schedule(static) reduction(+ : bb, rr, rz)
    for (int i = 0; i < n; i++) {
        double d = 0., s = 0.;
        if (free_mask[i]) {
            for (int k = indptr[i]; k < indptr[i + 1]; k++) {
                const int j = indices[k];
                if (j == i) d = data[k];
                if (free_mask[j]) s += data[k] * x[j];
            }
            dinv[i] = (fabs(d) > 1e-300) ? 1. / fabs(d) : 1.;
            r[i] = b[i] - s;
            bb += b[i] * b[i];
        } else {
            dinv[i] = 0.;
            r[i] = 0.;
            x[i] = 0.;
        }
        z[i] = dinv[i] * r[i];
        p[i] = z[i];
        rr += r[i] * r[i];
        rz += r[i] * z[i];
    }
    const double thresh2 = rtol * rtol * bb;
    while (it < maxit && rr > thresh2) {
        double pq = 0.;
#pragma omp parallel for num_threads(nt) schedule(static) reduction(+ : pq)
        for (int i = 0; i < n; i++) {
            double s = 0.;
            if (free_mask[i])
                for (int k = indptr[i]; k < indptr[i + 1]; k++) s += data[k] * p[indices[k]];  /* p = 0 on prescribed DOFs */
            q[i] = s;
            pq += p[i] * s;
        }
        if (!(pq > 0.)) break;
        const double alpha = rz / pq;
        double rr2 = 0., rz2 = 0.;
#pragma omp parallel for num_threads(nt) schedule(static) reduction(+ : rr2, rz2)
        for (int i = 0; i < n; i++) {
            x[i] += alpha * p[i];
            r[i] -= alpha * q[i];
            z[i] = dinv[i] * r[i];
            rr2 += r[i] * r[i];
            rz2 += r[i] * z[i];
        }
        const double beta = rz2 / rz;
#pragma omp parallel for num_threads(nt) schedule(static)
        for (int i = 0; i < n; i++) p[i] = z[i] + beta * p[i];
        rr = rr2;
        rz = rz2;
        it++;
    }
    if (relres) *relres = bb > 0. ? sqrt(rr / bb) : 0.;
    free(r);
    free(z);
    free(p);
    free(q);
    free(dinv);
    return it;
}

/* ------------------------------------------------------------------------------------------------
 * Work-hardening-aware SVC materials (PLFO_SVC_WH): batched drivers with the hardening modulus as explicit state. */
void plfo_fgrad_wh_batch(const plfo_material *m, int n, const double *sig, const double *epl, double *a, double *kh_raw)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        plfo_material mm = *m;
        plfo_calc_fgrad_wh(&mm, sig + 6 * (size_t)i, epl ? epl + 6 * (size_t)i : NULL, a + 6 * (size_t)i, kh_raw ? kh_raw + i : NULL);
    }
}

void plfo_yf_wh_batch(const plfo_material *m, int n, const double *sig, const double *epl, double *yf)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) yf[i] = svc_wh_decision(m, sig + 6 * (size_t)i, epl ? epl + 6 * (size_t)i : NULL);
}

void plfo_full_yf_wh_batch(const plfo_material *m, int n, const double *sig, const double *epl, const double *khard, double *yf)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n; i++) {
        plfo_material mm = *m;
        if (khard) mm.khard = khard[i];
        yf[i] = plfo_ML_full_yf(&mm, sig + 6 * (size_t)i, epl ? epl + 6 * (size_t)i : NULL, NULL);
    }
}

void plfo_response_wh_batch(const plfo_material *mats, int n, const int *mat_id, const double *sig, const double *epl,
                            const double *deps, const double *CV, const double *khard_in, double *fy, double *sig_out,
                            double *depl, double *ct, int *nsteps, double *khard_out, int sequential, int nthreads,
                            double *kh_point /* [n] or NULL: Material.khard right after each point's call */)
{
    if (sequential) { /* one Material object per material, mutated call after call in index order (model.py:1340-1359) */
        int nmat = 0;
        for (int i = 0; i < n; i++)
            if ((mat_id ? mat_id[i] : 0) + 1 > nmat) nmat = (mat_id ? mat_id[i] : 0) + 1;
        plfo_material *mm = (plfo_material *)malloc(sizeof(plfo_material) * nmat);
        for (int k = 0; k < nmat; k++) mm[k] = mats[k];
        if (khard_in)
            for (int k = 0; k < nmat; k++) mm[k].khard = khard_in[k]; /* [nmat] entry values */
        for (int i = 0; i < n; i++) {
            const int mid = mat_id ? mat_id[i] : 0;
            if (mm[mid].kind == PLFO_ELASTIC) {
                nsteps[i] = -1;
                continue;
            }
            nsteps[i] = plfo_response(&mm[mid], sig + 6 * (size_t)i, epl + 6 * (size_t)i, deps + 6 * (size_t)i,
                                      CV + 36 * (size_t)mid, &fy[i], sig_out + 6 * (size_t)i, depl + 6 * (size_t)i,
                                      ct + 36 * (size_t)i);
            if (kh_point) kh_point[i] = mm[mid].khard;
        }
        if (khard_out)
            for (int k = 0; k < nmat; k++) khard_out[k] = mm[k].khard; /* [nmat] exit values */
        free(mm);
        return;
    }
#ifdef _OPENMP
    const int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    (void)nthreads;
#endif
#pragma omp parallel for num_threads(nt) schedule(dynamic, 16)
    for (int i = 0; i < n; i++) {
        const int mid = mat_id ? mat_id[i] : 0;
        plfo_material mm = mats[mid];
        if (khard_in) mm.khard = khard_in[i];
        if (mm.kind == PLFO_ELASTIC) {
            nsteps[i] = -1;
            if (khard_out) khard_out[i] = mm.khard;
            continue;
        }
        nsteps[i] = plfo_response(&mm, sig + 6 * (size_t)i, epl + 6 * (size_t)i, deps + 6 * (size_t)i, CV + 36 * (size_t)mid,
                                  &fy[i], sig_out + 6 * (size_t)i, depl + 6 * (size_t)i, ct + 36 * (size_t)i);
        if (khard_out) khard_out[i] = mm.khard;
        if (kh_point) kh_point[i] = mm.khard;
    }
}
