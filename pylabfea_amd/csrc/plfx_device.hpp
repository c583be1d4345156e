// plfx_device.hpp — device-side constitutive math for gfx950 (MI355X).
//
// One thread owns one material point: the 6-component stress / plastic strain / strain
// increment and the 21-entry symmetric tangent live in VGPRs; per-material constants
// (elastic matrix, compliance, Hill coefficients, SVC support vectors) are staged in LDS and
// read with wave-uniform (broadcast) addresses when the points of a wave share a material.
//
// What is computed follows pyLabFEA v4.4.2 (paths relative to /root/reference/src/pylabfea):
//   material.py:207-346 response   :348-412 calc_yf    :414-516 ML_full_yf   :576-676 calc_seq
//   material.py:704-858 calc_fgrad :974-1007 get_sflow :1009-1055 epl_dot    :1057-1086 C_tan
//   basic.py:304 sig_dev, :328 eps_eq, :26 yf_tolerance
// How it is computed is not: the tangent T = C - (Ca)(Ca)^T/h is never formed (only its action and
// its weighted sum), the principal-stress eigen-solve that the reference computes and discards on
// the Hill-6p branch is skipped, and the 3x6 min-norm least-squares of the scale-back step is solved
// in closed form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "plfx_raypoly.hpp"
#include "plfx_lapack3.hpp"

namespace plfx {

extern __shared__ double dyn_lds[];

// Region timers of the wave-per-element SVC kernels (probe builds only: hipcc ... -DPLFX_PROF_REGIONS -o libplfx_prof.so, loaded
// through PLFX_LIB): lane 0 of every wave adds the shader-clock ticks it spent in a region; plfx_destroy prints the totals.
// CAVEAT (measured, round 5): s_memtime waits for all outstanding memory operations, which serialises the software-pipelined
// LDS reads -- the instrumented corrector runs 4x slower and the shares are distorted; the scaling probe
// tools/probes/svc_scalar_share.py (time against the number of support vectors) is the measurement DESIGN 10.2 quotes.
#ifdef PLFX_PROF_REGIONS
__device__ unsigned long long g_prof[16];
#define PROF_CNT(r) do { if ((threadIdx.x & 15) == 0) atomicAdd(&g_prof[r], 1ull); } while (0)   /* per row of 16 lanes */
#define PROF_T0(t) const unsigned long long prof_##t = __builtin_readcyclecounter()
#define PROF_ADD(r, t)                                                                                         \
    do {                                                                                                      \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_prof[r], (unsigned long long)__builtin_readcyclecounter() - prof_##t); \
    } while (0)
#else
#define PROF_T0(t) do { } while (0)
#define PROF_ADD(r, t) do { } while (0)
#define PROF_CNT(r) do { } while (0)
#endif

// 2^y for the RBF kernel sums, y <= 0 (tiny positive round-off allowed); the callers fold log2(e) into -gamma.
// n = rint(y), r = y - n exactly (|r| <= 0.5), 2^r by the degree-11 interpolating polynomial on Chebyshev nodes
// (max relative error 2.2e-16 = 1 ulp, checked against 50-digit arithmetic), 2^n added to the exponent field.
// Arguments below -1020 are clamped (2^-1020 ~ 1e-307: zero for any kernel sum); no overflow / NaN paths.
// 17 instructions instead of the ~35 of the library exp().
__device__ __forceinline__ double exp2_neg(double y)
{
    y = fmax(y, -1020.);
    // n = rint(y) by the 1.5 * 2^52 trick: the low word of (y + magic) is n as a two's-complement integer, so neither
    // v_rndne_f64 nor v_cvt_i32_f64 is needed
    const double t = y + 6755399441055744.0;
    const double n = t - 6755399441055744.0;
    const double r = y - n;
    const int ni = __double2loint(t);
    double p = 4.4558179083360645e-10;
    p = fma(p, r, 7.074194297288521e-09);
    p = fma(p, r, 1.0178057087733941e-07);
    p = fma(p, r, 1.3215432535912375e-06);
    p = fma(p, r, 1.5252733841556773e-05);
    p = fma(p, r, 0.00015403530463724353);
    p = fma(p, r, 0.001333355814640647);
    p = fma(p, r, 0.009618129107587256);
    p = fma(p, r, 0.055504108664821625);
    p = fma(p, r, 0.24022650695910158);
    p = fma(p, r, 0.6931471805599453);
    p = fma(p, r, 1.0);
    return __hiloint2double(__double2hiint(p) + (ni << 20), __double2loint(p));
}
// N of them in lock-step (the same operations per value, so the same bits): written coefficient by coefficient so that the
// N Horner chains are issued interleaved.  The row kernels of the SVC evaluate 4 - 8 exponentials per chunk of support vectors; the
// compiler interleaved them pairwise at best, and with a 4-cycle issue slot per wave64 FP64 instruction two dependent chains do
// not cover the latency of v_fma_f64 (round 6: valu issue utilisation of k_sweep_svc_row<1> 0.65 -> see DESIGN 11.5).
template <int N>
__device__ __forceinline__ void exp2_neg_n(const double (&yin)[N], double (&out)[N])
{
    double r[N], p[N];
    int ni[N];
#pragma unroll
    for (int c = 0; c < N; c++) {
        const double y = fmax(yin[c], -1020.);
        const double t = y + 6755399441055744.0;
        const double n = t - 6755399441055744.0;
        r[c] = y - n;
        ni[c] = __double2loint(t);
        p[c] = 4.4558179083360645e-10;
    }
#define PLFX_EXP2_STEP(COEF)                              \
    _Pragma("unroll") for (int c = 0; c < N; c++) p[c] = fma(p[c], r[c], COEF); \
    __builtin_amdgcn_sched_barrier(0);
    PLFX_EXP2_STEP(7.074194297288521e-09)
    PLFX_EXP2_STEP(1.0178057087733941e-07)
    PLFX_EXP2_STEP(1.3215432535912375e-06)
    PLFX_EXP2_STEP(1.5252733841556773e-05)
    PLFX_EXP2_STEP(0.00015403530463724353)
    PLFX_EXP2_STEP(0.001333355814640647)
    PLFX_EXP2_STEP(0.009618129107587256)
    PLFX_EXP2_STEP(0.055504108664821625)
    PLFX_EXP2_STEP(0.24022650695910158)
    PLFX_EXP2_STEP(0.6931471805599453)
    PLFX_EXP2_STEP(1.0)
#undef PLFX_EXP2_STEP
#pragma unroll
    for (int c = 0; c < N; c++) out[c] = __hiloint2double(__double2hiint(p[c]) + (ni[c] << 20), __double2loint(p[c]));
}
constexpr double LOG2E = 1.4426950408889634;
#ifndef PLFX_RAY_BOUND
#define PLFX_RAY_BOUND 1.e-7
#endif
constexpr double RAY_BOUND = PLFX_RAY_BOUND;   // rigorous bound on |f - p| the sampled-ray interval is shrunk to (see ray_sample)

// sum over the 64 lanes of a wave, result in every lane (and wave-uniform for the compiler: scalar branches).
// Four DPP butterfly steps inside each row of 16 lanes (quad xor 1, xor 2, half-row mirror, row mirror: no LDS
// crossbar), then the four row sums are read into scalar registers and added.  Fixed order: deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_allsum(double v)
{
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror
    v += dpp_f64<0x140>(v);  // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

constexpr double YF_TOL = 5.e-3;       // basic.py:26
constexpr double SPLIT_THRESHOLD = -0.15;  // material.py:260
constexpr int MAXIT = 50;              // material.py:207

// symmetric 6x6 stored as 21 entries, upper triangle row-major
__host__ __device__ constexpr int sym_idx(int i, int j)
{
    return (i <= j) ? (i * 6 - i * (i - 1) / 2 + (j - i)) : (j * 6 - j * (j - 1) / 2 + (i - j));
}

// Per-material record in device memory (and staged in LDS by the kernels).
struct MatDev {
    double CV[21];   // element elastic matrix (symmetric)
    double SV[21];   // compliance used by the scale-back step (material.py:315-320)
    double hill[6];
    double sy, khard, d0;  // d0 = drucker (calc_seq adds d0*tr(sig)/3, calc_fgrad adds d0/3)
    double E, nu;
    double gamma, intercept, scale_seq;
    double scale_wh;     // SVC with work-hardening features: scaling of the plastic-strain features (material.py:2343)
    double svc_sabs, svc_vvmax;  // sum |dual_k| and max |v_k|^2 over the support vectors (error bounds of YfSvcT::ray_sample)
    const double *rowtab;        // 6-feature SVC: the tables of the row kernels in device memory, laid out like their LDS copy
                                 // (v[6][rowpad], dual, |v|^2, -, RAYPOLY_MT, 0.98^i, 1.02^i) -- read from here (L2) when the
                                 // tables do not fit the LDS of a CU (k_*_row<..., false>)
    int32_t rowpad, _pad_row;    // support vectors padded to a multiple of 64
    const double *sv;    // device pointer [nsv*nfeat]
    const double *dual;  // device pointer [nsv]
    double barlat[18], barlat_exp;  // Yld2004-18p coefficients
    int32_t kind, sdim, nsv, dev_only, nfeat, barlat_normal;  // barlat_normal: the native Barlat normal is enabled (extension)
};

// y = C x for a symmetric 21-entry matrix
__device__ __forceinline__ void symv(const double *C, const double *x, double *y)
{
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double s = 0.;
#pragma unroll
        for (int j = 0; j < 6; j++) s = fma(C[sym_idx(i, j)], x[j], s);
        y[i] = s;
    }
}

__device__ __forceinline__ double dot6(const double *a, const double *b)
{
    double s = 0.;
#pragma unroll
    for (int i = 0; i < 6; i++) s = fma(a[i], b[i], s);
    return s;
}

// basic.py:350-352
__device__ __forceinline__ double eps_eq(const double *e)
{
    double n = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
    double s = e[3] * e[3] + e[4] * e[4] + e[5] * e[5];
    return sqrt(2. * (n + 0.5 * s) / 3.);
}

// material.py:650-673 (Hill-6p on the Voigt stress; J2 when hill == 1)
__device__ __forceinline__ double hill_seq(const MatDev &m, const double *s)
{
    double d01 = s[0] - s[1], d12 = s[1] - s[2], d20 = s[2] - s[0];
    double I2 = m.hill[0] * d01 * d01 + m.hill[1] * d12 * d12 + m.hill[2] * d20 * d20 +
                6. * m.hill[3] * s[3] * s[3] + 6. * m.hill[4] * s[4] * s[4] +
                6. * m.hill[5] * s[5] * s[5];
    double I1 = (s[0] + s[1] + s[2]) * m.d0 / 3.;
    return sqrt(0.5 * I2) + I1;
}

__device__ __forceinline__ double sflow_of(const MatDev &m, const double *epl)
{
    return m.sy + eps_eq(epl) * m.khard;  // material.py:997
}

// material.py:826-845 analytic normal
__device__ __forceinline__ void hill_fgrad(const MatDev &m, const double *s, double *a)
{
    double seq = hill_seq(m, s);
    double p = (s[0] + s[1] + s[2]) / 3.;
    double s0 = s[0] - p, s1 = s[1] - p, s2 = s[2] - p;
    double h0 = m.hill[0], h1 = m.hill[1], h2 = m.hill[2];
    double d3 = m.d0 / 3.;
    a[0] = ((h0 + h2) * s0 - h0 * s1 - h2 * s2) / (2. * seq) + d3;
    a[1] = ((h1 + h0) * s1 - h0 * s0 - h1 * s2) / (2. * seq) + d3;
    a[2] = ((h2 + h1) * s2 - h2 * s0 - h1 * s1) / (2. * seq) + d3;
    a[3] = 3. * m.hill[3] * s[3] / seq;
    a[4] = 3. * m.hill[4] * s[4] / seq;
    a[5] = 3. * m.hill[5] * s[5] / seq;
}

// ---------------------------------------------------------------------------------------------
// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations (Voigt input), V columns = eigenvectors.
__device__ inline void jacobi3_dev(const double *s, double *w, double *V)
{
    double A[9] = {s[0], s[5], s[4], s[5], s[1], s[3], s[4], s[3], s[2]};
#pragma unroll
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1. : 0.;
    for (int sweep = 0; sweep < 30; sweep++) {
        const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
        if (off == 0.) break;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                const double apq = A[p * 3 + q];
                if (apq == 0.) continue;
                const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2. * apq);
                const double t = (theta >= 0. ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
                const double c = 1. / sqrt(t * t + 1.), sn = t * c;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double akp = A[k * 3 + p], akq = A[k * 3 + q];
                    A[k * 3 + p] = c * akp - sn * akq;
                    A[k * 3 + q] = sn * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                    A[p * 3 + k] = c * apk - sn * aqk;
                    A[q * 3 + k] = sn * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - sn * vkq;
                    V[k * 3 + q] = sn * vkp + c * vkq;
                }
            }
    }
    w[0] = A[0];
    w[1] = A[4];
    w[2] = A[8];
}

// States with out-of-plane shear: the order of the principal stresses is the one np.linalg.eig + the reference's re-ordering
// give, i.e. LAPACK's dgeev replayed (plfx_lapack3.hpp).  Out of line: a cold path (no state of a 2-d model takes it) that must
// not cost the sweep kernels registers.
__device__ __noinline__ void sig_princ_general(const double *s, double *sp)
{
    double q[6] = {s[0], s[1], s[2], s[3], s[4], s[5]}, o[3];
    lapack3::sig_princ_lapack3(q, o);
    sp[0] = o[0];
    sp[1] = o[1];
    sp[2] = o[2];
}

// basic.py:107-179 sig_princ: principal stresses in the reference's axis-tracking order.  Plane
// states (s23 = s13 = 0, every stress of the 2-d FE path) have the closed form below; the larger
// eigenvalue of the in-plane block belongs to axis 0 iff s0 >= s1 (argmax |ev|, first maximum on ties).
// General 3-d states: sig_princ_general above (round 6; until then the natural rule "axis i -> eigenvector with the
// largest |component i|", which is the reference's only when dgeev's order happens to agree).
__device__ inline void sig_princ_dev(const double *s, double *sp)
{
    if (s[3] == 0. && s[4] == 0.) {
        const double mean = 0.5 * (s[0] + s[1]);
        const double hd = 0.5 * (s[0] - s[1]);
        const double R = sqrt(hd * hd + s[5] * s[5]);
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
    sig_princ_general(s, sp);
}

// 3-parameter Hill / J2 on principal stresses (material.py:662-673); I1 from the Voigt normal components
__device__ __forceinline__ double princ_seq_sp(const MatDev &m, const double *sp, double tr)
{
    const double d12 = sp[0] - sp[1], d23 = sp[1] - sp[2], d31 = sp[2] - sp[0];
    const double I2 = 0.5 * (m.hill[0] * d12 * d12 + m.hill[1] * d23 * d23 + m.hill[2] * d31 * d31);
    return sqrt(I2) + tr * m.d0 / 3.;
}

__device__ inline double princ_seq(const MatDev &m, const double *s)
{
    double sp[3];
    sig_princ_dev(s, sp);
    return princ_seq_sp(m, sp, s[0] + s[1] + s[2]);
}

// epl_dot / C_tan with sdim == 3 (material.py:1044-1047, 1079-1081): the gradient w.r.t. the principal
// stresses goes into the normal Voigt components, the shear components stay zero
__device__ inline void princ_fgrad(const MatDev &m, const double *s, double *a)
{
    double sp[3];
    sig_princ_dev(s, sp);
    const double tr = sp[0] + sp[1] + sp[2];
    const double seq = princ_seq_sp(m, sp, tr);
    const double pm = tr / 3.;
    const double s0 = sp[0] - pm, s1 = sp[1] - pm, s2 = sp[2] - pm;
    const double h0 = m.hill[0], h1 = m.hill[1], h2 = m.hill[2], d3 = m.d0 / 3.;
    a[0] = ((h0 + h2) * s0 - h0 * s1 - h2 * s2) / (2. * seq) + d3;
    a[1] = ((h1 + h0) * s1 - h0 * s0 - h1 * s2) / (2. * seq) + d3;
    a[2] = ((h2 + h1) * s2 - h2 * s0 - h1 * s1) / (2. * seq) + d3;
    a[3] = a[4] = a[5] = 0.;
}

// Tresca (material.py:630-632) and Barlat Yld2004-18p (material.py:678-702) equivalent stresses
__device__ inline double tresca_seq(const double *s)
{
    double w[3], V[9];
    jacobi3_dev(s, w, V);
    return fmax(fmax(w[0], w[1]), w[2]) - fmin(fmin(w[0], w[1]), w[2]);
}

__device__ inline double barlat_seq(const MatDev &m, const double *s)
{
    const double *b = m.barlat;
    const double p = (s[0] + s[1] + s[2]) / 3.;
    const double sd[6] = {s[0] - p, s[1] - p, s[2] - p, s[3], s[4], s[5]};
    const double st1[6] = {-b[0] * sd[1] - b[1] * sd[2], -b[2] * sd[0] - b[3] * sd[2], -b[4] * sd[0] - b[5] * sd[1],
                           b[6] * sd[3], b[7] * sd[4], b[8] * sd[5]};
    const double st2[6] = {-b[9] * sd[1] - b[10] * sd[2], -b[11] * sd[0] - b[12] * sd[2],
                           -b[13] * sd[0] - b[14] * sd[1], b[15] * sd[3], b[16] * sd[4], b[17] * sd[5]};
    double p1[3], p2[3], V[9];
    jacobi3_dev(st1, p1, V);
    jacobi3_dev(st2, p2, V);
    double acc = 0.;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) acc += pow(fabs(p1[i] - p2[j]), m.barlat_exp);
    return pow(0.25 * acc, 1. / m.barlat_exp);
}

// Barlat Yld2004-18p equivalent stress AND its gradient d seq / d sigma (Voigt, shear components carrying both symmetric
// entries like every normal of the path: the plastic strain increment lam * a has engineering shear strains).  The
// reference has no flow rule for Barlat (calc_fgrad raises, material.py:822-825); this is the native normal the north star
// asks for (EXTENSION, opt-in per material: plfx_material.barlat_normal).  With phi = sum_ij |S'_i - S''_j|^a,
// seq = (phi/4)^(1/a):  d seq/d phi = seq / (a phi);  d phi/d S'_i = a sum_j |.|^(a-1) sgn(.), d phi/d S''_j = - a sum_i ...;
// d S_i / d s~ = n_i n_i^T of the eigenvector n_i (s~ = M sd the linearly transformed deviator), Voigt form
// [n0^2, n1^2, n2^2, 2 n1 n2, 2 n0 n2, 2 n0 n1]; then through M^T (18 coefficients) and the deviator projection.
// Repeated principal values need no special care: the sum over a degenerate pair is invariant under the choice of basis.
__device__ inline double barlat_seq_grad(const MatDev &m, const double *s, double *a)
{
    const double *b = m.barlat;
    const double ex = m.barlat_exp;
    const double p = (s[0] + s[1] + s[2]) / 3.;
    const double sd[6] = {s[0] - p, s[1] - p, s[2] - p, s[3], s[4], s[5]};
    const double st1[6] = {-b[0] * sd[1] - b[1] * sd[2], -b[2] * sd[0] - b[3] * sd[2], -b[4] * sd[0] - b[5] * sd[1],
                           b[6] * sd[3], b[7] * sd[4], b[8] * sd[5]};
    const double st2[6] = {-b[9] * sd[1] - b[10] * sd[2], -b[11] * sd[0] - b[12] * sd[2],
                           -b[13] * sd[0] - b[14] * sd[1], b[15] * sd[3], b[16] * sd[4], b[17] * sd[5]};
    double p1[3], p2[3], V1[9], V2[9];
    jacobi3_dev(st1, p1, V1);
    jacobi3_dev(st2, p2, V2);
    double phi = 0., d1[3] = {0., 0., 0.}, d2[3] = {0., 0., 0.};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double df = p1[i] - p2[j], ad = fabs(df);
            phi += pow(ad, ex);
            const double w = (ad > 0.) ? ex * pow(ad, ex - 1.) * (df > 0. ? 1. : -1.) : 0.;
            d1[i] += w;
            d2[j] -= w;
        }
    const double seq = pow(0.25 * phi, 1. / ex);
    if (!(phi > 0.)) {
#pragma unroll
        for (int i = 0; i < 6; i++) a[i] = 0.;
        return seq;
    }
    double g1[6] = {0., 0., 0., 0., 0., 0.}, g2[6] = {0., 0., 0., 0., 0., 0.};
#pragma unroll
    for (int i = 0; i < 3; i++) {  // eigenvector i = column i of V
        const double n0 = V1[0 * 3 + i], n1 = V1[1 * 3 + i], n2 = V1[2 * 3 + i];
        g1[0] = fma(d1[i], n0 * n0, g1[0]);
        g1[1] = fma(d1[i], n1 * n1, g1[1]);
        g1[2] = fma(d1[i], n2 * n2, g1[2]);
        g1[3] = fma(d1[i], 2. * n1 * n2, g1[3]);
        g1[4] = fma(d1[i], 2. * n0 * n2, g1[4]);
        g1[5] = fma(d1[i], 2. * n0 * n1, g1[5]);
        const double m0 = V2[0 * 3 + i], m1 = V2[1 * 3 + i], m2 = V2[2 * 3 + i];
        g2[0] = fma(d2[i], m0 * m0, g2[0]);
        g2[1] = fma(d2[i], m1 * m1, g2[1]);
        g2[2] = fma(d2[i], m2 * m2, g2[2]);
        g2[3] = fma(d2[i], 2. * m1 * m2, g2[3]);
        g2[4] = fma(d2[i], 2. * m0 * m2, g2[4]);
        g2[5] = fma(d2[i], 2. * m0 * m1, g2[5]);
    }
    // M^T g: st_0 = -b0 sd1 - b1 sd2, st_1 = -b2 sd0 - b3 sd2, st_2 = -b4 sd0 - b5 sd1 (and b9.. for the second map)
    double h[6];
    h[0] = -b[2] * g1[1] - b[4] * g1[2] - b[11] * g2[1] - b[13] * g2[2];
    h[1] = -b[0] * g1[0] - b[5] * g1[2] - b[9] * g2[0] - b[14] * g2[2];
    h[2] = -b[1] * g1[0] - b[3] * g1[1] - b[10] * g2[0] - b[12] * g2[1];
    h[3] = b[6] * g1[3] + b[15] * g2[3];
    h[4] = b[7] * g1[4] + b[16] * g2[4];
    h[5] = b[8] * g1[5] + b[17] * g2[5];
    const double hm = (h[0] + h[1] + h[2]) / 3.;  // deviator projection (symmetric)
    const double sc = seq / (ex * phi);
    a[0] = sc * (h[0] - hm);
    a[1] = sc * (h[1] - hm);
    a[2] = sc * (h[2] - hm);
    a[3] = sc * h[3];
    a[4] = sc * h[4];
    a[5] = sc * h[5];
    return seq;
}

// ---------------------------------------------------------------------------------------------
// RBF-SVC yield function (material.py:398-405 decision function, :765-807 gradient).
// sv/dual point to LDS when the kernel staged them, to global memory otherwise; every lane reads
// the same address (broadcast), so the loop is FP64-VALU bound (software exp).
__device__ __forceinline__ void svc_features(const MatDev &m, const double *s, double *x)
{
    double p = m.dev_only ? (s[0] + s[1] + s[2]) / 3. : 0.;  // material.py:2336
    // the reference divides (x = sig/scale_seq); keep the division for bit-fidelity of x
    x[0] = (s[0] - p) / m.scale_seq;
    x[1] = (s[1] - p) / m.scale_seq;
    x[2] = (s[2] - p) / m.scale_seq;
    x[3] = s[3] / m.scale_seq;
    x[4] = s[4] / m.scale_seq;
    x[5] = s[5] / m.scale_seq;
}

__device__ inline double svc_decision_x(const MatDev &m, const double *sv, const double *dual,
                                        const double *x)
{
    double f = 0.;
    const int n = m.nsv;
    const double g = -m.gamma * LOG2E;
    for (int k = 0; k < n; k++) {
        const double *v = sv + 6 * k;
        double hh = 0.;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double d = x[i] - v[i];
            hh = fma(d, d, hh);
        }
        f = fma(dual[k], exp2_neg(g * hh), f);
    }
    return f + m.intercept;
}

__device__ inline double svc_decision(const MatDev &m, const double *sv, const double *dual,
                                      const double *s)
{
    double x[6];
    svc_features(m, s, x);
    return svc_decision_x(m, sv, dual, x);
}

__device__ inline void svc_fgrad(const MatDev &m, const double *sv, const double *dual,
                                 const double *s, double *a)
{
    double x[6], acc[6] = {0., 0., 0., 0., 0., 0.};
    svc_features(m, s, x);
    const int n = m.nsv;
    const double g = -m.gamma * LOG2E;
    for (int k = 0; k < n; k++) {
        const double *v = sv + 6 * k;
        double hv[6], hh = 0.;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            hv[i] = x[i] - v[i];
            hh = fma(hv[i], hv[i], hh);
        }
        double w = dual[k] * exp2_neg(g * hh);
#pragma unroll
        for (int i = 0; i < 6; i++) acc[i] = fma(w, hv[i], acc[i]);
    }
    double sc = -2. * m.gamma / m.scale_seq;  // (-2 gamma)(x - sv) / scale_seq, material.py:775,807
#pragma unroll
    for (int i = 0; i < 6; i++) a[i] = acc[i] * sc;
}

// Yield-function policy: analytic Hill-6p / J2.
struct YfHill {
    const MatDev &m;
    __device__ YfHill(const MatDev &mm) : m(mm) {}
    __device__ __forceinline__ double seq(const double *s) const { return hill_seq(m, s); }
    // calc_yf(sig, epl)
    __device__ __forceinline__ double plain(const double *s, const double *epl) const
    {
        return hill_seq(m, s) - sflow_of(m, epl);
    }
    // yield function used for the convergence tests inside response()
    __device__ __forceinline__ double full(const double *s, const double *epl) const { return plain(s, epl); }
    __device__ __forceinline__ double full0(const double *s, double fy0) const { (void)s; return fy0; }
    __device__ __forceinline__ void fgrad(const double *s, double *a) const { hill_fgrad(m, s, a); }
    // hardening modulus / flow stress as the response loop sees them (a work-hardening-aware SVC carries a mutable khard)
    __device__ __forceinline__ void fgrad(const double *s, const double *epl, double *a) const { (void)epl; fgrad(s, a); }
    __device__ __forceinline__ double kh() const { return m.khard; }
    __device__ __forceinline__ int touched() const { return 0; }
    __device__ __forceinline__ double sflow(const double *epl) const { return sflow_of(m, epl); }
    __device__ __forceinline__ double sflow_entry(const double *epl) const { return sflow_of(m, epl); }
};

// Yield-function policy: Barlat Yld2004-18p with the native normal (extension; see barlat_seq_grad).
struct YfBarlat {
    const MatDev &m;
    __device__ YfBarlat(const MatDev &mm) : m(mm) {}
    __device__ __forceinline__ double seq(const double *s) const { return barlat_seq(m, s); }
    __device__ __forceinline__ double plain(const double *s, const double *epl) const
    {
        return barlat_seq(m, s) - sflow_of(m, epl);
    }
    __device__ __forceinline__ double full(const double *s, const double *epl) const { return plain(s, epl); }
    __device__ __forceinline__ double full0(const double *s, double fy0) const { (void)s; return fy0; }
    __device__ __forceinline__ void fgrad(const double *s, double *a) const { barlat_seq_grad(m, s, a); }
    // hardening modulus / flow stress as the response loop sees them (a work-hardening-aware SVC carries a mutable khard)
    __device__ __forceinline__ void fgrad(const double *s, const double *epl, double *a) const { (void)epl; fgrad(s, a); }
    __device__ __forceinline__ double kh() const { return m.khard; }
    __device__ __forceinline__ int touched() const { return 0; }
    __device__ __forceinline__ double sflow(const double *epl) const { return sflow_of(m, epl); }
    __device__ __forceinline__ double sflow_entry(const double *epl) const { return sflow_of(m, epl); }
};

// Yield-function policy: sdim = 3, Hill-3p / J2 on principal stresses.
struct YfPrinc3 {
    const MatDev &m;
    __device__ YfPrinc3(const MatDev &mm) : m(mm) {}
    __device__ __forceinline__ double seq(const double *s) const { return princ_seq(m, s); }
    __device__ __forceinline__ double plain(const double *s, const double *epl) const
    {
        return princ_seq(m, s) - sflow_of(m, epl);
    }
    __device__ __forceinline__ double full(const double *s, const double *epl) const { return plain(s, epl); }
    __device__ __forceinline__ double full0(const double *s, double fy0) const { (void)s; return fy0; }
    __device__ __forceinline__ void fgrad(const double *s, double *a) const { princ_fgrad(m, s, a); }
    // hardening modulus / flow stress as the response loop sees them (a work-hardening-aware SVC carries a mutable khard)
    __device__ __forceinline__ void fgrad(const double *s, const double *epl, double *a) const { (void)epl; fgrad(s, a); }
    __device__ __forceinline__ double kh() const { return m.khard; }
    __device__ __forceinline__ int touched() const { return 0; }
    __device__ __forceinline__ double sflow(const double *epl) const { return sflow_of(m, epl); }
    __device__ __forceinline__ double sflow_entry(const double *epl) const { return sflow_of(m, epl); }
};

// ---------------------------------------------------------------------------------------------
// 2-feature SVC of sdim = 3 ML materials: x = (seq_J2/scale - 1, polar angle/pi) of the principal
// stresses (create_scaled_input, material.py:2331-2333; sig_polar_ang, basic.py:68-104)
__device__ __forceinline__ void svc3_features(const MatDev &m, const double *sp, double *x)
{
    const double d12 = sp[0] - sp[1], d23 = sp[1] - sp[2], d31 = sp[2] - sp[0];
    const double seq = sqrt(0.5 * (d12 * d12 + d23 * d23 + d31 * d31));
    const double hyd = (sp[0] + sp[1] + sp[2]) / 3.;
    const double e0 = sp[0] - hyd, e1 = sp[1] - hyd, e2 = sp[2] - hyd;
    double vn = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
    if (vn < 1.e-4) vn = 1.;
    const double dsa = (e0 * 0.816496580927726 - (e1 + e2) * 0.408248290463863) / vn;
    const double dsb = (e1 - e2) * 0.7071067811865476 / vn;
    x[0] = seq / m.scale_seq - 1.;
    x[1] = atan2(dsb, dsa) / 3.141592653589793;
}

__device__ inline double svc3_decision(const MatDev &m, const double *sv, const double *dual, const double *s)
{
    double sp[3], x[2];
    sig_princ_dev(s, sp);
    svc3_features(m, sp, x);
    double f = 0.;
    const double g = -m.gamma * LOG2E;
    for (int k = 0; k < m.nsv; k++) {
        const double h0 = x[0] - sv[2 * k], h1 = x[1] - sv[2 * k + 1];
        f = fma(dual[k], exp2_neg(g * fma(h0, h0, h1 * h1)), f);
    }
    return f + m.intercept;
}

// gradient through the Jacobian of (seq, theta) (material.py:779-807), in the normal Voigt components
__device__ inline void svc3_fgrad(const MatDev &m, const double *sv, const double *dual, const double *s, double *a)
{
    double sp[3], x[2];
    sig_princ_dev(s, sp);
    svc3_features(m, sp, x);
    double dK1 = 0.;
    const double g = -m.gamma * LOG2E;
    for (int k = 0; k < m.nsv; k++) {
        const double h0 = x[0] - sv[2 * k], h1 = x[1] - sv[2 * k + 1];
        dK1 = fma(dual[k] * exp2_neg(g * fma(h0, h0, h1 * h1)), -2. * m.gamma * h1, dK1);
    }
    const double hyd = (sp[0] + sp[1] + sp[2]) / 3.;
    const double dev[3] = {sp[0] - hyd, sp[1] - hyd, sp[2] - hyd};
    const double vn = sqrt(dev[0] * dev[0] + dev[1] * dev[1] + dev[2] * dev[2]) * sqrt(1.5);
    const double av[3] = {0.816496580927726, -0.408248290463863, -0.408248290463863};
    const double bv[3] = {0., 0.7071067811865476, -0.7071067811865476};
    if (vn > 0.1) {
        const double cr = sp[0] * av[0] + sp[1] * av[1] + sp[2] * av[2];
        const double ci = sp[0] * bv[0] + sp[1] * bv[1] + sp[2] * bv[2];
        const double n2 = cr * cr + ci * ci;
#pragma unroll
        for (int i = 0; i < 3; i++) a[i] = 3. * dev[i] / vn + (bv[i] * cr - av[i] * ci) / n2 * dK1;
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) a[i] = 1. + dK1;
    }
    a[3] = a[4] = a[5] = 0.;
}

// scipy.optimize.brentq (scipy 1.15.3, Brent 1973) specialised to f(x) = decision(x*su)
template <class F>
__device__ inline double brentq_dev(F f, double xa, double xb, double fa, double fb, double xtol,
                                    double rtol, int maxiter, bool &converged)
{
    double xpre = xa, xcur = xb, xblk = 0., fpre = fa, fcur = fb, fblk = 0.;
    double spre = 0., scur = 0., sbis, delta, stry, dpre, dblk;
    converged = true;
    if (fpre == 0.) return xpre;
    if (fcur == 0.) return xcur;
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
        fcur = f(xcur);
    }
    converged = false;
    return xcur;
}

// The same iteration as brentq_dev, cut at its single function evaluation: next() runs the step logic up to the point
// where f(xcur) is needed (returns true) or the search has ended (returns false, `root` / `converged` set); the caller
// evaluates and stores fcur.  Lets ML_full_yf evaluate the yield function from ONE call site.
struct BrentState {
    double xpre, xcur, xblk, fpre, fcur, fblk, spre, scur, root;
    int it;
    bool converged;
    __device__ __forceinline__ bool start(double xa, double xb, double fa, double fb)
    {
        xpre = xa; xcur = xb; xblk = 0.; fpre = fa; fcur = fb; fblk = 0.; spre = scur = 0.; it = 0;
        converged = true;
        if (fpre == 0.) { root = xpre; return false; }
        if (fcur == 0.) { root = xcur; return false; }
        return true;
    }
    __device__ __forceinline__ bool next(double xtol, double rtol, int maxiter)
    {
        if (it >= maxiter) { converged = false; root = xcur; return false; }
        double sbis, delta, stry, dpre, dblk;
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
        if (fcur == 0. || fabs(sbis) < delta) { root = xcur; return false; }
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                dpre = (fpre - fcur) / (xpre - xcur);
                dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            const double lim = fmin(fabs(spre), 3. * fabs(sbis) - delta);
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
        it++;
        return true;  // caller: fcur = f(xcur)
    }
};

// Yield-function policy: RBF-SVC (ML_yf) on NF = 6 stress features (sdim 6) or NF = 2 (sdim 3).
// calc_seq of an ML material is J2 (hill = ones), on Voigt or on principal stresses.
// WAVE = NC > 0 (NF = 6): one wave works on ONE material point; every lane carries the same point (the scalar part of the
// algorithm runs redundantly in lock-step, as cheap as running it once) and the support-vector sums are split over the
// lanes: lane L takes the vectors L, L+64, ... from the SoA tables in dynamic LDS (stride-1 across lanes: conflict-free
// ds_read_b64, no dependent flat loads), followed by one wave reduction.  npad = vectors padded to a multiple of 64 NC
// with dual = 0.  Tables: v[6][npad] at dyn_lds[0], dual[npad] at dyn_lds[6*npad].
template <int NF, int WAVE = 0, bool POLY = false>
struct YfSvcT {
    const MatDev &m;
    const double *sv;
    const double *dual;
    int npad;
    __device__ YfSvcT(const MatDev &mm, const double *s, const double *d, int np = 0) : m(mm), sv(s), dual(d), npad(np) {}
    __device__ __forceinline__ double seq(const double *s) const { return NF == 6 ? hill_seq(m, s) : princ_seq(m, s); }
    // NC = WAVE support vectors per lane and trip (k, k + 64, ...): all 7 NC LDS reads are issued before the first use
    // (the empty asm pins them: otherwise the scheduler re-uses one register pair and waits after every read), and the
    // NC independent distance / exp chains give the one or two resident waves of a SIMD instruction-level parallelism.
    // npad is a multiple of 64 NC.
    static constexpr int NC = WAVE > 0 ? WAVE : 1;
    // LDS reads of one trip: issued as a batch (the sched_barrier keeps the scheduler from sinking them to their uses) ...
    __device__ __forceinline__ static void issue(int npad, int k, double (*v)[7])
    {
#pragma unroll
        for (int i = 0; i < 7; i++)
#pragma unroll
            for (int c = 0; c < NC; c++) v[c][i] = dyn_lds[i * npad + k + 64 * c];
        __builtin_amdgcn_sched_barrier(0);
    }
    // ... and awaited together, one trip later (software pipeline: the reads of trip t+1 fly during the arithmetic of t)
    __device__ __forceinline__ static void pin(double (*v)[7])
    {
#pragma unroll
        for (int c = 0; c < NC; c++)
            asm volatile("" : "+v"(v[c][0]), "+v"(v[c][1]), "+v"(v[c][2]), "+v"(v[c][3]), "+v"(v[c][4]), "+v"(v[c][5]),
                         "+v"(v[c][6]));
    }
    // body(v) consumes one trip of NC vectors per lane.  (Ping-pong prefetching of the next trip's reads was measured:
    // no gain, the loop is bound by FP64 issue, not by LDS latency.)
    template <class BODY>
    __device__ __forceinline__ void for_trips(BODY body) const
    {
        for (int k = threadIdx.x & 63; k < npad; k += 64 * NC) {
            double A[NC][7];
            issue(npad, k, A);
            pin(A);
            body(A);
        }
    }
    // ---- evaluations along a ray x = t su (ML_full_yf evaluates ~12 points on the same ray): the feature vector is
    // linear in t, X(t) = t D with D = features(su), so |X - v_k|^2 = t^2 |D|^2 - 2 t (D.v_k) + |v_k|^2.  D.v_k is
    // computed once per ray and kept in registers (one value per vector of this lane), |v_k|^2 is the 8th LDS table:
    // 2 fused multiply-adds and 2 LDS reads per vector and evaluation instead of 12 operations and 7 reads.
    static constexpr int MAXTRIP = 2048 / (64 * NC);   // 64 NC MAXTRIP >= npad: up to 2048 support vectors
    // FP32 sign screen of the marching bracket (ray_screen): in the corrector kernel (4 vectors per lane and trip, one wave
    // per SIMD, registers to spare); the streaming kernel (2 waves per SIMD, 256 VGPRs) has no room for the FP32 copies
    static constexpr bool SCREEN = WAVE >= 4 && NF == 6 && !POLY;
    struct RaySetup {
        double DD;
        double ck[WAVE > 0 ? MAXTRIP : 1][NC];
        float ck32[SCREEN ? MAXTRIP : 1][NC];    // the same in FP32 for the sign screen of the marching bracket
        float ckmax, gvvmax;                     // bounds of |D.v_k| and gamma log2(e) |v_k|^2 over the vectors (error margin)
    };
    __device__ __forceinline__ void ray_setup(const double *su, RaySetup &r) const
    {
        if (WAVE == 0 || NF != 6) return;
        double D[6];
        svc_features(m, su, D);
        r.DD = 0.;
#pragma unroll
        for (int i = 0; i < 6; i++) r.DD = fma(D[i], D[i], r.DD);
        const int lane = threadIdx.x & 63;
        float ckm = 0.f;
#pragma unroll
        for (int t = 0; t < MAXTRIP; t++) {
            if (t * 64 * NC < npad) {  // wave-uniform
                double v[NC][7];
                issue(npad, lane + t * 64 * NC, v);
                pin(v);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    double a = 0.;
#pragma unroll
                    for (int i = 0; i < 6; i++) a = fma(D[i], v[c][i], a);
                    r.ck[t][c] = a;
                    if (SCREEN) {
                        r.ck32[t][c] = (float)a;
                        ckm = fmaxf(ckm, fabsf((float)a));
                    }
                }
            }
        }
        if (!SCREEN) return;
        // bounds for the error margin of ray_screen (wave maxima; |v_k|^2 table * gamma log2 e)
        float gm = 0.f;
        const float2 *t32 = reinterpret_cast<const float2 *>(dyn_lds + 8 * npad);
        for (int k = lane; k < npad; k += 64) gm = fmaxf(gm, fabsf(t32[k].y));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ckm = fmaxf(ckm, __shfl_xor(ckm, off, 64));
            gm = fmaxf(gm, __shfl_xor(gm, off, 64));
        }
        r.ckmax = ckm;
        r.gvvmax = gm;
    }
    // FP32 estimate of the decision function at x su with a rigorous error margin: the marching bracket of ML_full_yf
    // (material.py:475-486, ten to thirty-five 2 % steps per call) only needs the SIGN of the yield function at the
    // intermediate points -- wherever |estimate| > margin the FP64 evaluation is skipped; the points where the march stops,
    // and every point the estimate cannot decide, are evaluated in FP64 as before, so bracket ends, brentq iterates and the
    // result are unchanged.  Per vector: arg = (g t^2 |D|^2) + (-2 g t)(D.v_k) + g |v_k|^2 in FP32, v_exp_f32, two
    // multiply-adds (sum and sum of magnitudes S): 5 FP32 instructions against 21 FP64 ones.
    // Error of the estimate <= S (ln 2 * d_arg + e_exp + e_sum) with d_arg <= 6 * 2^-24 * M, M = |g t^2 DD| + |2 g t| max|D.v| +
    // max g|v|^2 (three roundings + the FP32 representation of the three inputs), e_exp <= 2^-22 (v_exp_f32: 1 ulp, table
    // entry 0.5 ulp), e_sum <= (npad / 64 + 7) 2^-24 (sequential partial sums per lane + wave reduction).  The margin used is
    // S (1e-6 M + 1e-5): four to five times that bound.
    __device__ __forceinline__ double ray_screen(const RaySetup &r, double x, double &margin) const
    {
        const double g = -m.gamma * LOG2E;
        const float A = (float)(g * x * x * r.DD), B = (float)(-2. * g * x);
        const int lane = threadIdx.x & 63;
        const float2 *t32 = reinterpret_cast<const float2 *>(dyn_lds + 8 * npad);
        float f[NC], sa[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) f[c] = 0.f, sa[c] = 0.f;
#pragma unroll
        for (int t = 0; t < MAXTRIP; t++) {
            if (t * 64 * NC < npad) {
                const int k = lane + t * 64 * NC;
                float2 tv[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) tv[c] = t32[k + 64 * c];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(B, r.ck32[SCREEN ? t : 0][c], A) + tv[c].y);
                    f[c] = fmaf(tv[c].x, e, f[c]);
                    sa[c] = fmaf(fabsf(tv[c].x), e, sa[c]);
                }
            }
        }
        float fs = f[0], ss = sa[0];
#pragma unroll
        for (int c = 1; c < NC; c++) fs += f[c], ss += sa[c];
        const double F = wave_allsum((double)fs), S = wave_allsum((double)ss);
        const double M = fabs((double)A) + fabs((double)B) * (double)r.ckmax + (double)r.gvvmax;
        margin = S * (1.e-6 * M + 1.e-5) + 1.e-300;
        return F + m.intercept;
    }
    __device__ __forceinline__ double ray_eval(const double *su, const RaySetup &r, double x) const
    {
        if (WAVE == 0 || NF != 6) {
            double xs[6];
#pragma unroll
            for (int i = 0; i < 6; i++) xs[i] = x * su[i];
            return decision(xs);
        }
        const double g = -m.gamma * LOG2E;
        const int lane = threadIdx.x & 63;
        const double tt = x * x * r.DD, m2t = -2. * x;
        double f[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) f[c] = 0.;
#pragma unroll
        for (int t = 0; t < MAXTRIP; t++) {
            if (t * 64 * NC < npad) {
                const int k = lane + t * 64 * NC;
                double du[NC], vv[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    du[c] = dyn_lds[6 * npad + k + 64 * c];
                    vv[c] = dyn_lds[7 * npad + k + 64 * c];
                }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const double h = fma(m2t, r.ck[t][c], tt) + vv[c];
                    f[c] = fma(du[c], exp2_neg(g * h), f[c]);
                }
            }
        }
        double tsum = f[0];
#pragma unroll
        for (int c = 1; c < NC; c++) tsum += f[c];
        return wave_allsum(tsum) + m.intercept;
    }
    // ---- sampled-ray form of the ray search (round 5; POLY, wave mode, 6 stress features).  Along x = t su every kernel
    // term is a Gaussian in t, d_k exp(-gamma (DD t^2 - 2 c_k t + |v_k|^2)), so on equally spaced points t_j = lo + j dl
    //   term_k(t_j) = [d_k e^{-gamma h_k(lo)}] rho_k^j e^{-gamma DD dl^2 j^2},   rho_k = e^{2 gamma dl (c_k - DD lo)}:
    // ONE pass over the support vectors (two exp per vector, then one multiply and one add per sample) yields NS samples
    // of the decision function, the fixed matrix RAYPOLY_MT turns them into the Chebyshev coefficients of the
    // interpolating polynomial p on [lo, hi], and the 2 % marching bracket of ML_full_yf (material.py:468-486) and the
    // brentq iterates (:501-503) are evaluated on p -- 32 instructions per point in lock-step instead of a pass over the
    // vectors (~10 FP64 + ~20 FP32 passes per call before).  f along a ray is entire with length scale 1/sqrt(2 gamma DD):
    //   |f - p| <= sum|d_k| K sqrt(NS!) / (4 NS) (sqrt(2 gamma DD) dl)^NS          (K = 1.0865: Cramer's bound on Hermite functions)
    // on the whole interval; the interval is shrunk about the start of the march until that bound is <= 1e-7 (a root shift of
    // <= 2e-6 MPa at the decision function's slope; measured |f - p| ~ 4e-11, the conditioning of the equispaced samples).  March
    // decisions use p only where |p| exceeds the bound + a round-off allowance, and any point outside [lo, hi] or closer to zero
    // than that is evaluated as before (decision_wave on x su): the bracket is the reference's, the root the one brentq finds on p.
    static constexpr int NS = RAYPOLY_N;
    struct RayPoly {
        double c[NS];              // Chebyshev coefficients (wave-uniform)
        double lo, hi, ua, ub;     // u = ua x + ub maps [lo, hi] to [-1, 1]
        double margin;             // |p| <= margin: the sign of f is not decided by p
        bool ok;
        __device__ __forceinline__ bool covers(double x) const { return ok && x >= lo && x <= hi; }
        __device__ __forceinline__ double eval(double x) const   // Clenshaw
        {
            const double u = fma(ua, x, ub), u2 = u + u;
            double b1 = 0., b2 = 0.;
#pragma unroll
            for (int k = NS - 1; k >= 1; k--) {
                const double t = fma(u2, b1, c[k] - b2);
                b2 = b1;
                b1 = t;
            }
            return fma(u, b1, c[0] - b2);
        }
    };
    // tables behind the support-vector tables in dynamic LDS (stage_svc_wave): RAYPOLY_MT, 0.98^i, 1.02^i (i < 64)
    __device__ __forceinline__ const double *poly_tab() const { return dyn_lds + 9 * npad; }
    __device__ __forceinline__ void ray_sample(const double *su, double x0, bool halved, RayPoly &P) const
    {
        P.ok = false;
        if (!(WAVE > 0 && NF == 6 && POLY)) return;
        double D[6];
        svc_features(m, su, D);
        double DD = 0.;
#pragma unroll
        for (int i = 0; i < 6; i++) DD = fma(D[i], D[i], DD);
        // the march starts at x0 = sflow and goes down or up, or at x0 = sflow / 2 (material.py:468-473) and goes up
        double lo = halved ? 0.94 * x0 : 0.72 * x0, hi = halved ? 2.7 * x0 : 1.30 * x0;
        double dl = (hi - lo) * (1. / (NS - 1));
        const double q = sqrt(2. * m.gamma * DD);
        constexpr double KN = 1.0865 * 4574143.623 / (4. * NS);   // K sqrt(16!) / (4 N)
        static_assert(NS == 16, "KN and the 16th power below are written for 16 samples");
        double b = q * dl;
        b *= b; b *= b; b *= b; b *= b;
        double bound = m.svc_sabs * KN * b;
        if (!(bound <= RAY_BOUND)) {
            if (!(bound < 1.e300)) return;
            const double sh = sqrt(sqrt(sqrt(sqrt(RAY_BOUND / bound))));   // (RAY_BOUND / bound)^(1/16)
            lo = x0 - (x0 - lo) * sh;
            hi = x0 + (hi - x0) * sh;
            dl = (hi - lo) * (1. / (NS - 1));
            bound = RAY_BOUND;
        }
        // the recurrence multiplies by rho_k up to NS - 1 times: keep its exponent range harmless
        const double sD = sqrt(DD);
        if (!(2. * m.gamma * dl * (NS - 1) * sD * (sqrt(m.svc_vvmax) + sD * hi) < 60.) || !(hi > lo)) return;
        const double g = -m.gamma * LOG2E;
        const double A0 = g * DD * lo * lo, A1 = -2. * g * lo;   // log2(w_k / d_k) = A0 + A1 c_k + g |v_k|^2
        const double R1 = -2. * g * dl, R0 = -R1 * DD * lo;      // log2 rho_k = R1 c_k + R0
        double acc[NS];
#pragma unroll
        for (int j = 0; j < NS; j++) acc[j] = 0.;
        PROF_T0(a);
        for (int k = threadIdx.x & 63; k < npad; k += 64 * NC) {
            double v[NC][8];
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int c = 0; c < NC; c++) v[c][i] = dyn_lds[i * npad + k + 64 * c];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NC; c++)
                asm volatile("" : "+v"(v[c][0]), "+v"(v[c][1]), "+v"(v[c][2]), "+v"(v[c][3]), "+v"(v[c][4]), "+v"(v[c][5]),
                             "+v"(v[c][6]), "+v"(v[c][7]));
            double w[NC], rho[NC], ea[2 * NC], eo[2 * NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                double ck = 0.;
#pragma unroll
                for (int i = 0; i < 6; i++) ck = fma(D[i], v[c][i], ck);
                ea[c] = fma(g, v[c][7], fma(A1, ck, A0));
                ea[NC + c] = fma(R1, ck, R0);
            }
            exp2_neg_n<2 * NC>(ea, eo);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                w[c] = v[c][6] * eo[c];
                rho[c] = eo[NC + c];
            }
#pragma unroll
            for (int j = 0; j < NS; j++)
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    acc[j] += w[c];
                    if (j < NS - 1) w[c] *= rho[c];
                }
        }
        PROF_ADD(2, a);
        // f_j = b + G_j sum_k (...), G_j = 2^(g DD dl^2 j^2) by recurrence: G_{j+1} = G_j t_j, t_{j+1} = t_j G_1^2
        PROF_T0(b);
        const double G1 = exp2_neg(g * DD * dl * dl);
        double G = 1., t = G1;
        const double r = G1 * G1;
        const double *tab = poly_tab();
        const int li = threadIdx.x & (NS - 1);
        double ci = 0.;   // lane i (mod NS): Chebyshev coefficient i
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const double fj = fma(G, wave_allsum(acc[j]), m.intercept);
            ci = fma(tab[j * NS + li], fj, ci);
            G *= t;
            t *= r;
        }
#pragma unroll
        for (int i = 0; i < NS; i++) P.c[i] = readlane_f64(ci, i);
        P.lo = lo;
        P.hi = hi;
        P.ua = 2. / (hi - lo);
        P.ub = -(hi + lo) / (hi - lo);
        P.margin = bound + 4.e-12 * m.svc_sabs;
        P.ok = true;
        PROF_ADD(3, b);
    }
    __device__ __forceinline__ double decision_wave(const double *s) const
    {
        double x[6];
        svc_features(m, s, x);
        const double g = -m.gamma * LOG2E;
        double f[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) f[c] = 0.;
        for_trips([&](double (*v)[7]) {
            double h[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) h[c] = 0.;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const double d = x[i] - v[c][i];
                    h[c] = fma(d, d, h[c]);
                }
#pragma unroll
            for (int c = 0; c < NC; c++) f[c] = fma(v[c][6], exp2_neg(g * h[c]), f[c]);
        });
        double t = f[0];
#pragma unroll
        for (int c = 1; c < NC; c++) t += f[c];
        return wave_allsum(t) + m.intercept;
    }
    __device__ __forceinline__ void fgrad_wave(const double *s, double *a) const
    {
        double x[6], acc[6] = {0., 0., 0., 0., 0., 0.};
        svc_features(m, s, x);
        const double g = -m.gamma * LOG2E;
        for_trips([&](double (*v)[7]) {
            double h[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) h[c] = 0.;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    v[c][i] = x[i] - v[c][i];
                    h[c] = fma(v[c][i], v[c][i], h[c]);
                }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const double w = v[c][6] * exp2_neg(g * h[c]);
#pragma unroll
                for (int i = 0; i < 6; i++) acc[i] = fma(w, v[c][i], acc[i]);
            }
        });
        const double sc = -2. * m.gamma / m.scale_seq;
#pragma unroll
        for (int i = 0; i < 6; i++) a[i] = wave_allsum(acc[i]) * sc;
    }
    __device__ __forceinline__ double decision(const double *s) const
    {
        if (WAVE) return decision_wave(s);
        return NF == 6 ? svc_decision(m, sv, dual, s) : svc3_decision(m, sv, dual, s);
    }
    __device__ __forceinline__ double plain(const double *s, const double *epl) const
    {
        (void)epl;
        return decision(s);
    }
    // ML_full_yf (material.py:414-516): distance to the yield locus along the ray through s
    // (ld == nullptr) or along the loading direction ld.
    __device__ inline double full_ld(const double *s, const double *epl, const double *ld,
                                     int *status) const
    {
        double seqv = seq(s);
        double sflow = sflow_of(m, epl);
        if (status) *status = 0;
        if (seqv < 0.01 && ld == nullptr) return seqv - 0.85 * sflow;
        double su[6];
        if (ld == nullptr) {
#pragma unroll
            for (int i = 0; i < 6; i++) su[i] = s[i] / seqv;
        } else {
            // su = ld[0:sdim] * sqrt(1.5) / |ld[0:sdim]|  (material.py:455-462)
            double hh = 0.;
#pragma unroll
            for (int i = 0; i < 6; i++) hh += (i < (NF == 6 ? 6 : 3)) ? ld[i] * ld[i] : 0.;
            hh = sqrt(hh);
            if (hh < 1.e-3) {
                su[0] = sqrt(1.5);
#pragma unroll
                for (int i = 1; i < 6; i++) su[i] = 0.;
            } else {
#pragma unroll
                for (int i = 0; i < 6; i++) su[i] = (i < (NF == 6 ? 6 : 3)) ? ld[i] * sqrt(1.5) / hh : 0.;
            }
        }
        // find_yloc_scalar (material.py:547-574): f(x) = calc_yf(x*su).  The marching bracket (:468-486) and the
        // brentq search (:501-503) are driven as one state machine so that f has a single call site (the wave-mode
        // evaluation is a long unrolled loop).  Evaluation order and arithmetic are those of the straight-line form.
        RaySetup ray;
        if (!POLY) ray_setup(su, ray);
        double x0 = sflow;
        const bool halved = su[0] * su[1] < -1.e-5;
        if (halved) x0 *= 0.5;  // material.py:468-473
        RayPoly P;
        ray_sample(su, x0, halved, P);
#ifdef PLFX_PROF_REGIONS
        unsigned long long prof_m = __builtin_readcyclecounter();
#endif
        double x1 = x0, f0 = 0., f1 = 0., xs = 0.;
        bool conv = true;
        BrentState br;
        int phase = 0;  // 0 first point, 1 marching down (:475-480), 2 marching up (:481-486), 3 brentq
        double xq = x0;
        for (;;) {
            if (POLY && P.ok && (phase == 1 || phase == 2)) {
                // lane i looks at the i-th next point of the march (i = 0: xq itself): the leading run of points at which
                // p decides that the march goes on is skipped; the products are then repeated one by one so that the point
                // the march stops at (a bracket end) carries the reference's rounding
                const double fac = (phase == 1) ? 0.98 : 1.02;
                const double z = xq * poly_tab()[NS * NS + (phase == 1 ? 0 : 64) + (threadIdx.x & 63)];
                const double pv = P.eval(z);
                const bool on = z >= P.lo && z <= P.hi &&
                                ((phase == 1) ? (pv > P.margin && z > 0.010000001) : (pv < -P.margin && z < 4.9999999 * sflow));
                const unsigned long long stop = ~__ballot(on);
                const int k = stop ? __ffsll((long long)stop) - 1 : 64;
                double xx = xq;
                for (int i = 0; i < k; i++) xx *= fac;
                if (phase == 1) x0 = xx; else x1 = xx;
                xq = xx;
                if (k == 64) continue;
            }
            if (SCREEN && (phase == 1 || phase == 2)) {  // marching: the sign alone decides whether it goes on
                double mg;
                const double fe = ray_screen(ray, xq, mg);
                if (phase == 1 && fe - mg > 0. && x0 > 0.01) {  // certainly f >= 0: :475-480 marches on
                    x0 *= 0.98;
                    xq = x0;
                    continue;
                }
                if (phase == 2 && fe + mg < 0. && x1 < 5. * sflow) {  // certainly f < 0: :481-486 marches on
                    x1 *= 1.02;
                    xq = x1;
                    continue;
                }
            }
            double fq;
            if (POLY) {
                bool direct = true;
                if (P.covers(xq)) {
                    fq = P.eval(xq);
                    direct = phase < 3 && fabs(fq) <= P.margin;
                }
                if (direct) {
                    double xs6[6];
#pragma unroll
                    for (int i = 0; i < 6; i++) xs6[i] = xq * su[i];
                    fq = decision(xs6);
                }
            } else {
                fq = ray_eval(su, ray, xq);
            }
            if (phase == 0) {
                f0 = f1 = fq;
                phase = 1;
            } else if (phase == 1) {
                f0 = fq;
            } else if (phase == 2) {
                f1 = fq;
            } else {
                br.fcur = fq;
            }
            if (phase == 1) {
                if (f0 >= 0. && x0 > 0.01) {
                    x0 *= 0.98;
                    xq = x0;
                    continue;
                }
                phase = 2;
            }
            if (phase == 2) {
                if (f1 < 0. && x1 < 5. * sflow) {
                    x1 *= 1.02;
                    xq = x1;
                    continue;
                }
                if (f0 * f1 > 0.) {  // material.py:495-499
                    if (status) *status = 1;
                    return seqv - 0.85 * sflow;
                }
                phase = 3;
#ifdef PLFX_PROF_REGIONS
                if ((threadIdx.x & 63) == 0) atomicAdd(&g_prof[4], (unsigned long long)__builtin_readcyclecounter() - prof_m);
                prof_m = __builtin_readcyclecounter();
#endif
                if (!br.start(x0, x1, f0, f1)) break;
            }
            if (!br.next(1.e-5, 4. * 2.220446049250313e-16, 100)) break;
            xq = br.xcur;
        }
#ifdef PLFX_PROF_REGIONS
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_prof[5], (unsigned long long)__builtin_readcyclecounter() - prof_m);
#endif
        xs = br.root;
        conv = br.converged;
        if (conv && xs < 4. * sflow) return seqv - xs * seq(su);  // material.py:507
        if (status) *status = 2;
        return seqv - 0.85 * sflow;  // material.py:510
    }
    __device__ __forceinline__ double full(const double *s, const double *epl) const
    {
        return full_ld(s, epl, nullptr, nullptr);
    }
    // material.py:265: fy0 re-evaluated as full yield function with epl = 0
    __device__ __forceinline__ double full0(const double *s, double fy0) const
    {
        (void)fy0;
        const double z[6] = {0., 0., 0., 0., 0., 0.};
        return full_ld(s, z, nullptr, nullptr);
    }
    __device__ __forceinline__ void fgrad(const double *s, double *a) const
    {
        if (WAVE)
            fgrad_wave(s, a);
        else if (NF == 6)
            svc_fgrad(m, sv, dual, s, a);
        else
            svc3_fgrad(m, sv, dual, s, a);
    }
    __device__ __forceinline__ void fgrad(const double *s, const double *epl, double *a) const { (void)epl; fgrad(s, a); }
    __device__ __forceinline__ double kh() const { return m.khard; }
    __device__ __forceinline__ int touched() const { return 0; }
    __device__ __forceinline__ double sflow(const double *epl) const { return sflow_of(m, epl); }
    __device__ __forceinline__ double sflow_entry(const double *epl) const { return sflow_of(m, epl); }
};

// ---------------------------------------------------------------------------------------------
// RBF-SVC yield function with WORK-HARDENING features (material.py:2342-2346, 808-814): 15 features
//   x = [ (dev) sig / scale_seq (6) | epl / scale_wh (6) | accumulated strain | max. stress / scale_seq | flag ]
// of which the last three are zero on the path (Material.response / Model.solve never pass them, material.py:207-346).
// The hardening modulus is not a parameter but a by-product of the gradient:
//   khard = max(0, - sum_k dK/dx[6+k] * scale_seq / scale_wh)                                   (:808-814)
// and the reference keeps it in ONE mutable attribute of the Material object that every calc_fgrad call overwrites and
// every get_sflow / epl_dot / C_tan call reads -- carried from call to call, across elements, in index order.  Inside one
// response() call that is reproduced exactly (K below); across calls the data-parallel engine carries it per material
// point (k_sweep_*: kh_el[e]) and the single-call entry points take / return it explicitly.
// WAVE = 1 (round 4, k_sweep_wh_wave): one wave works on ONE material point -- every lane carries the same point, the
// support-vector sums are split over the lanes (vector k of lane L: L, L + 64, ...) and closed by a wave reduction; all other
// arithmetic runs redundantly in lock-step.  WAVE = 0: one thread per point (entry points, calc_scf, the thread sweeps).
template <int WAVE>
struct YfSvcWhT {
    const MatDev &m;
    const double *sv;
    const double *dual;
    const double K_in;   // hardening modulus at the entry of the call
    mutable double K;    // ... as of the last gradient evaluation
    mutable int touch = 0;   // a gradient evaluation has overwritten the modulus (else the call hands its entry value on)
    __device__ YfSvcWhT(const MatDev &mm, const double *s, const double *d, double k0) : m(mm), sv(s), dual(d), K_in(k0), K(k0) {}
    __device__ __forceinline__ int touched() const { return touch; }
    __device__ __forceinline__ double seq(const double *s) const { return hill_seq(m, s); }
    __device__ __forceinline__ void features(const double *s, const double *epl, double *x) const
    {
        svc_features(m, s, x);
#pragma unroll
        for (int i = 0; i < 6; i++) x[6 + i] = epl[i] / m.scale_wh;
    }
    // decision function; the three trailing features are zero: their squared distance is |sv[12..14]|^2
    __device__ inline double decision_x(const double *x) const
    {
        double f = 0.;
        const double g = -m.gamma * LOG2E;
        for (int k = WAVE ? (int)(threadIdx.x & 63) : 0; k < m.nsv; k += WAVE ? 64 : 1) {
            const double *v = sv + 15 * k;
            double hh = 0.;
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const double d = x[i] - v[i];
                hh = fma(d, d, hh);
            }
#pragma unroll
            for (int i = 12; i < 15; i++) hh = fma(v[i], v[i], hh);
            f = fma(dual[k], exp2_neg(g * hh), f);
        }
        if (WAVE) f = wave_allsum(f);
        return f + m.intercept;
    }
    __device__ __forceinline__ double plain(const double *s, const double *epl) const
    {
        double x[12];
        features(s, epl, x);
        return decision_x(x);
    }
    __device__ __forceinline__ double kh() const { return K; }
    __device__ __forceinline__ double sflow(const double *epl) const { return m.sy + eps_eq(epl) * K; }
    __device__ __forceinline__ double sflow_entry(const double *epl) const { return m.sy + eps_eq(epl) * K_in; }
    // gradient w.r.t. the stress (:797-807) and the hardening modulus it implies (:808-814, single point)
    __device__ inline double fgrad_raw(const double *s, const double *epl, double *a) const
    {
        double x[12], acc[12];
        features(s, epl, x);
#pragma unroll
        for (int i = 0; i < 12; i++) acc[i] = 0.;
        const double g = -m.gamma * LOG2E;
        for (int k = WAVE ? (int)(threadIdx.x & 63) : 0; k < m.nsv; k += WAVE ? 64 : 1) {
            const double *v = sv + 15 * k;
            double hv[12], hh = 0.;
#pragma unroll
            for (int i = 0; i < 12; i++) {
                hv[i] = x[i] - v[i];
                hh = fma(hv[i], hv[i], hh);
            }
#pragma unroll
            for (int i = 12; i < 15; i++) hh = fma(v[i], v[i], hh);
            const double w = dual[k] * exp2_neg(g * hh);
#pragma unroll
            for (int i = 0; i < 12; i++) acc[i] = fma(w, hv[i], acc[i]);
        }
        if (WAVE) {
#pragma unroll
            for (int i = 0; i < 12; i++) acc[i] = wave_allsum(acc[i]);
        }
        const double c2 = -2. * m.gamma;
#pragma unroll
        for (int i = 0; i < 6; i++) a[i] = acc[i] * c2 / m.scale_seq;
        double hk = 0.;
#pragma unroll
        for (int i = 6; i < 12; i++) hk -= acc[i] * c2 * m.scale_seq / m.scale_wh;
        return hk;
    }
    __device__ __forceinline__ void fgrad(const double *s, const double *epl, double *a) const
    {
        const double hk = fgrad_raw(s, epl, a);
        K = hk < 0. ? 0. : hk;  // strain softening not supported (:813-814)
        touch = 1;
    }
    // ML_full_yf (material.py:414-516) with the plastic strain in the features
    __device__ inline double full_ld(const double *s, const double *epl, const double *ld, int *status) const
    {
        const double seqv = seq(s);
        const double sfl = sflow(epl);
        if (status) *status = 0;
        if (seqv < 0.01 && ld == nullptr) return seqv - 0.85 * sfl;
        double su[6];
        if (ld == nullptr) {
#pragma unroll
            for (int i = 0; i < 6; i++) su[i] = s[i] / seqv;
        } else {
            double hh = 0.;
#pragma unroll
            for (int i = 0; i < 6; i++) hh += ld[i] * ld[i];
            hh = sqrt(hh);
            if (hh < 1.e-3) {
                su[0] = sqrt(1.5);
#pragma unroll
                for (int i = 1; i < 6; i++) su[i] = 0.;
            } else {
#pragma unroll
                for (int i = 0; i < 6; i++) su[i] = ld[i] * sqrt(1.5) / hh;
            }
        }
        auto f = [&](double x) {
            double xs[6];
#pragma unroll
            for (int i = 0; i < 6; i++) xs[i] = x * su[i];
            return plain(xs, epl);
        };
        double x0 = sfl;
        if (su[0] * su[1] < -1.e-5) x0 *= 0.5;
        double x1 = x0;
        double f0 = f(x0);
        double f1 = f0;
        while (f0 >= 0. && x0 > 0.01) {  // :475-480
            x0 *= 0.98;
            f0 = f(x0);
        }
        while (f1 < 0. && x1 < 5. * sfl) {  // :481-486
            x1 *= 1.02;
            f1 = f(x1);
        }
        if (f0 * f1 > 0.) {
            if (status) *status = 1;
            return seqv - 0.85 * sfl;
        }
        bool conv = true;
        const double xs = brentq_dev(f, x0, x1, f0, f1, 1.e-5, 4. * 2.220446049250313e-16, 100, conv);
        if (conv && xs < 4. * sfl) return seqv - xs * seq(su);
        if (status) *status = 2;
        return seqv - 0.85 * sfl;
    }
    __device__ __forceinline__ double full(const double *s, const double *epl) const { return full_ld(s, epl, nullptr, nullptr); }
    __device__ __forceinline__ double full0(const double *s, double fy0) const
    {
        (void)fy0;
        const double z[6] = {0., 0., 0., 0., 0., 0.};
        return full_ld(s, z, nullptr, nullptr);  // ML_full_yf(sig): epl = None -> zeros (:265, :437-438)
    }
};

// ---------------------------------------------------------------------------------------------
// 6-feature RBF-SVC, SIXTEEN LANES PER MATERIAL POINT (round 5, k_sweep_svc_row / k_full_yf_row).  The wave-per-point
// form above runs the scalar part of response() -- predictor / corrector algebra, the ray search's control flow, the
// polynomial evaluations of the sampled-ray form -- redundantly in all 64 lanes, and once the ray search needs one pass over
// the support vectors instead of thirty that part is half of the instructions of a sub-step.  Here a DPP row (16 lanes)
// carries one point and a wave four: the support-vector sums cost what they cost before (lane L of a row takes the vectors
// L, L + 16, ...; the four rows read the same LDS addresses), the scalar part a quarter.  A row is also exactly what the
// cross-lane hardware offers without LDS: sums by four DPP butterfly steps inside the row, broadcasts of one lane's value
// to its row by v_mov_b64_dpp row_newbcast.  Every value that steers control flow is bit-identical in the 16 lanes of a row
// (same inputs, same operations, commutative butterfly), so rows diverge from each other but never inside.
// The ray search is the sampled-ray form (see YfSvcT::ray_sample): NS = 16 samples, lane i of the row computes and keeps
// Chebyshev coefficient i; p(x) is Clenshaw's recurrence with the coefficients broadcast from their lanes.  With 16 lanes a
// row looks at 16 points per polynomial evaluation:
//  * the 2 % marching bracket (material.py:475-486): lane i tests the (i+1)-th next point of the march; the leading run of
//    points at which p decides the sign (|p| > margin, inside [lo, hi]) is skipped, the first point it does not decide is
//    evaluated as before (p, or the support-vector sum itself when p is closer to zero than its error bound or the point lies
//    outside the sampled interval) and the reference's loop condition is applied to that value;
//  * the root (:501-503): brentq(xtol = 1e-5) on [x0, x1] is replayed iterate for iterate (BrentState) with p as the function
//    (the support-vector sums where p does not cover an iterate), so the value response() branches on is the reference's
//    last iterate to ~1e-11, not a better root.
template <int NC, bool INLDS = true>
struct YfSvcRow {
    static constexpr int GS = 16, NS = RAYPOLY_N;
    static constexpr bool FUSED = true;   // fgrad_plain(): gradient at one point and decision function at another in one pass
    static_assert(NS == GS, "lane i of a row holds Chebyshev coefficient i");
    const MatDev &m;
    int npad;
    __device__ YfSvcRow(const MatDev &mm, int np) : m(mm), npad(np) {}
    // the tables: dynamic LDS (staged by stage_svc_wave), or the material's copy in device memory (INLDS = false: more support
    // vectors than the LDS holds; the 16 lanes of a row read 128 consecutive bytes, the four rows of a wave the same ones)
    __device__ __forceinline__ const double *tabs() const { return INLDS ? dyn_lds : m.rowtab; }
    __device__ __forceinline__ double seq(const double *s) const { return hill_seq(m, s); }
    // sum over the 16 lanes of a DPP row, result (bit-identical) in every lane of the row
    __device__ __forceinline__ static double row_allsum(double v)
    {
        v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
        v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
        v += dpp_f64<0x141>(v);  // row_half_mirror
        v += dpp_f64<0x140>(v);  // row_mirror
        return v;
    }
    template <int N>
    __device__ __forceinline__ static double row_bcast(double v)   // value of lane N of the row, in every lane of the row
    {
        return __builtin_amdgcn_update_dpp(0., v, 0x150 + N, 0xf, 0xf, false);   // row_newbcast:N
    }
    // features without the reference's six divisions (x = sig / scale_seq): one reciprocal, 1 ulp -- the row kernels are
    // held to 1e-6 sy like every SVC path (brentq's xtol), not to the bit
    __device__ __forceinline__ void features(const double *s, double *x) const
    {
        const double inv = 1. / m.scale_seq;
        const double p = m.dev_only ? (s[0] + s[1] + s[2]) / 3. : 0.;
        x[0] = (s[0] - p) * inv;
        x[1] = (s[1] - p) * inv;
        x[2] = (s[2] - p) * inv;
        x[3] = s[3] * inv;
        x[4] = s[4] * inv;
        x[5] = s[5] * inv;
    }
    __device__ __forceinline__ void load7(int k, double (*v)[7]) const
    {
        const double *T = tabs();
#pragma unroll
        for (int i = 0; i < 7; i++)
#pragma unroll
            for (int c = 0; c < NC; c++) v[c][i] = T[i * npad + k + GS * c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC; c++)
            asm volatile("" : "+v"(v[c][0]), "+v"(v[c][1]), "+v"(v[c][2]), "+v"(v[c][3]), "+v"(v[c][4]), "+v"(v[c][5]),
                         "+v"(v[c][6]));
    }
    __device__ __forceinline__ double decision_x(const double *x) const
    {
        const double g = -m.gamma * LOG2E;
        double f[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) f[c] = 0.;
        for (int k = threadIdx.x & (GS - 1); k < npad; k += GS * NC) {
            double v[NC][7], h[NC];
            load7(k, v);
#pragma unroll
            for (int c = 0; c < NC; c++) h[c] = 0.;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const double d = x[i] - v[c][i];
                    h[c] = fma(d, d, h[c]);
                }
            double ea[NC], eo[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) ea[c] = g * h[c];
            exp2_neg_n<NC>(ea, eo);
#pragma unroll
            for (int c = 0; c < NC; c++) f[c] = fma(v[c][6], eo[c], f[c]);
        }
        double t = f[0];
#pragma unroll
        for (int c = 1; c < NC; c++) t += f[c];
        return row_allsum(t) + m.intercept;
    }
    __device__ __forceinline__ double decision(const double *s) const
    {
        double x[6];
        features(s, x);
        return decision_x(x);
    }
    __device__ __forceinline__ double plain(const double *s, const double *epl) const
    {
        (void)epl;
        return decision(s);
    }
    // gradient at s (material.py:765-815) and, in the same pass over the support vectors, the decision function at s2
    // (epl_dot's calc_yf(sig + dsig), :1032): |x2 - v|^2 = |x - v|^2 + 2 (x2 - x).(x - v) + |x2 - x|^2
    template <bool WITH2>
    __device__ __forceinline__ double fgrad_impl(const double *s, const double *s2, double *a) const
    {
        double x[6], dx[6], acc[6] = {0., 0., 0., 0., 0., 0.}, f2[NC], dd = 0.;
        features(s, x);
        if (WITH2) {
            features(s2, dx);
#pragma unroll
            for (int i = 0; i < 6; i++) {
                dx[i] -= x[i];
                dd = fma(dx[i], dx[i], dd);
                dx[i] += dx[i];
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) f2[c] = 0.;
        const double g = -m.gamma * LOG2E;
        for (int k = threadIdx.x & (GS - 1); k < npad; k += GS * NC) {
            double v[NC][7], h[NC], h2[NC];
            load7(k, v);
#pragma unroll
            for (int c = 0; c < NC; c++) h[c] = 0., h2[c] = dd;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    v[c][i] = x[i] - v[c][i];
                    h[c] = fma(v[c][i], v[c][i], h[c]);
                    if (WITH2) h2[c] = fma(dx[i], v[c][i], h2[c]);
                }
            constexpr int NE = WITH2 ? 2 * NC : NC;
            double ea[NE], eo[NE];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                ea[c] = g * h[c];
                if (WITH2) ea[NE - NC + c] = g * (h[c] + h2[c]);
            }
            exp2_neg_n<NE>(ea, eo);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const double w = v[c][6] * eo[c];
                if (WITH2) f2[c] = fma(v[c][6], eo[NE - NC + c], f2[c]);
#pragma unroll
                for (int i = 0; i < 6; i++) acc[i] = fma(w, v[c][i], acc[i]);
            }
        }
        const double sc = -2. * m.gamma / m.scale_seq;
#pragma unroll
        for (int i = 0; i < 6; i++) a[i] = row_allsum(acc[i]) * sc;
        if (!WITH2) return 0.;
        double t = f2[0];
#pragma unroll
        for (int c = 1; c < NC; c++) t += f2[c];
        return row_allsum(t) + m.intercept;
    }
    __device__ __forceinline__ void fgrad(const double *s, double *a) const { fgrad_impl<false>(s, nullptr, a); }
    __device__ __forceinline__ void fgrad(const double *s, const double *epl, double *a) const { (void)epl; fgrad(s, a); }
    __device__ __forceinline__ double fgrad_plain(const double *s, const double *epl, const double *s2, double *a) const
    {
        (void)epl;
        return fgrad_impl<true>(s, s2, a);
    }
    // ---- sampled ray (see YfSvcT::ray_sample for the mathematics and the error bound)
    struct RowPoly {
        double ci;                 // Chebyshev coefficient (lane & 15) of the interpolant on [lo, hi]
        double lo, hi, ua, ub, margin;
        bool ok;
        __device__ __forceinline__ double eval(double x) const   // Clenshaw; x may differ from lane to lane
        {
            const double u = fma(ua, x, ub), u2 = u + u;
            double b1 = 0., b2 = 0., t;
#define PLFX_CLENSHAW(K) t = fma(u2, b1, row_bcast<K>(ci) - b2); b2 = b1; b1 = t;
            PLFX_CLENSHAW(15) PLFX_CLENSHAW(14) PLFX_CLENSHAW(13) PLFX_CLENSHAW(12) PLFX_CLENSHAW(11) PLFX_CLENSHAW(10)
            PLFX_CLENSHAW(9) PLFX_CLENSHAW(8) PLFX_CLENSHAW(7) PLFX_CLENSHAW(6) PLFX_CLENSHAW(5) PLFX_CLENSHAW(4)
            PLFX_CLENSHAW(3) PLFX_CLENSHAW(2) PLFX_CLENSHAW(1)
#undef PLFX_CLENSHAW
            return fma(u, b1, row_bcast<0>(ci) - b2);
        }
    };
    __device__ __forceinline__ const double *poly_tab() const { return tabs() + 9 * npad; }
    // xstate: ray parameter of the stress the search was called for (s = xstate su).  Inside the plastic corrector that stress lies
    // on the yield locus to a few per cent, i.e. xstate IS where the march ends and brentq works -- and for rays along which the
    // material is stronger than sflow by more than the default interval allows (config 4's loading direction: every search of the
    // corrector) the end of the march and brentq's iterates fell beyond hi and cost a pass over the support vectors each
    // (round 6, counted: 1.02 such passes per search, 13 % of the corrector).  The interval is stretched to 1.06 xstate when that is
    // a modest extension (<= 25 %; a trial stress far outside the locus says nothing about the root); the error bound below is
    // evaluated for the interval actually used and shrinks it as before if it has to.
    __device__ __forceinline__ void ray_sample(const double *su, double x0, bool halved, RowPoly &P, double xstate = 0.) const
    {
        P.ok = false;
        P.ci = 0.;
        P.lo = P.hi = P.ua = P.ub = P.margin = 0.;
        double D[6];
        features(su, D);
        double DD = 0.;
#pragma unroll
        for (int i = 0; i < 6; i++) DD = fma(D[i], D[i], DD);
        // the march starts at x0 = sflow and goes down or up, or at x0 = sflow / 2 (material.py:468-473) and goes up
        double lo = halved ? 0.94 * x0 : 0.72 * x0, hi = halved ? 2.7 * x0 : 1.30 * x0;
        if (1.06 * xstate > hi && 1.06 * xstate <= 1.25 * hi) hi = 1.06 * xstate;
        double dl = (hi - lo) * (1. / (NS - 1));
        const double q = sqrt(2. * m.gamma * DD);
        constexpr double KN = 1.0865 * 4574143.623 / (4. * NS);   // K sqrt(16!) / (4 N)
        static_assert(NS == 16, "KN and the 16th power below are written for 16 samples");
        double b = q * dl;
        b *= b; b *= b; b *= b; b *= b;
        double bound = m.svc_sabs * KN * b;
        if (!(bound <= RAY_BOUND)) {
            if (!(bound < 1.e300)) return;
            const double sh = sqrt(sqrt(sqrt(sqrt(RAY_BOUND / bound))));   // (RAY_BOUND / bound)^(1/16)
            lo = x0 - (x0 - lo) * sh;
            hi = x0 + (hi - x0) * sh;
            dl = (hi - lo) * (1. / (NS - 1));
            bound = RAY_BOUND;
        }
        const double sD = sqrt(DD);
        if (!(2. * m.gamma * dl * (NS - 1) * sD * (sqrt(m.svc_vvmax) + sD * hi) < 60.) || !(hi > lo)) return;
        const double g = -m.gamma * LOG2E;
        const double A0 = g * DD * lo * lo, A1 = -2. * g * lo;   // log2(w_k / d_k) = A0 + A1 c_k + g |v_k|^2
        const double R1 = -2. * g * dl, R0 = -R1 * DD * lo;      // log2 rho_k = R1 c_k + R0
        double acc[NS];
#pragma unroll
        for (int j = 0; j < NS; j++) acc[j] = 0.;
        const double *T = tabs();
        for (int k = threadIdx.x & (GS - 1); k < npad; k += GS * NC) {
            double v[NC][8];
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int c = 0; c < NC; c++) v[c][i] = T[i * npad + k + GS * c];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NC; c++)
                asm volatile("" : "+v"(v[c][0]), "+v"(v[c][1]), "+v"(v[c][2]), "+v"(v[c][3]), "+v"(v[c][4]), "+v"(v[c][5]),
                             "+v"(v[c][6]), "+v"(v[c][7]));
            double w[NC], rho[NC], ea[2 * NC], eo[2 * NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                double ck = 0.;
#pragma unroll
                for (int i = 0; i < 6; i++) ck = fma(D[i], v[c][i], ck);
                ea[c] = fma(g, v[c][7], fma(A1, ck, A0));
                ea[NC + c] = fma(R1, ck, R0);
            }
            exp2_neg_n<2 * NC>(ea, eo);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                w[c] = v[c][6] * eo[c];
                rho[c] = eo[NC + c];
            }
#pragma unroll
            for (int j = 0; j < NS; j++)
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    acc[j] += w[c];
                    if (j < NS - 1) w[c] *= rho[c];
                }
        }
        // f_j = b + G_j sum_k (...), G_j = 2^(g DD dl^2 j^2) by recurrence: G_{j+1} = G_j t_j, t_{j+1} = t_j G_1^2
        const double G1 = exp2_neg(g * DD * dl * dl);
        double G = 1., t = G1;
        const double r = G1 * G1;
        const double *tab = poly_tab();
        const int li = threadIdx.x & (NS - 1);
        double ci = 0.;
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const double fj = fma(G, row_allsum(acc[j]), m.intercept);
            ci = fma(tab[j * NS + li], fj, ci);
            G *= t;
            t *= r;
        }
        P.ci = ci;
        P.lo = lo;
        P.hi = hi;
        P.ua = 2. / (hi - lo);
        P.ub = -(hi + lo) / (hi - lo);
        P.margin = bound + 4.e-12 * m.svc_sabs;
        P.ok = true;
    }
    // f(x su): the polynomial where it decides (phase < 3: and is farther from zero than its error bound), the support-vector
    // sum otherwise
    __device__ __forceinline__ double evalx(const double *su, const RowPoly &P, double x, bool need_sign, int site = 0) const
    {
        double f = 0.;
        bool direct = true;
        if (P.ok && x >= P.lo && x <= P.hi) {
            f = P.eval(x);
            direct = need_sign && fabs(f) <= P.margin;
        }
        if (direct) {
            double xs[6];
#pragma unroll
            for (int i = 0; i < 6; i++) xs[i] = x * su[i];
            PROF_T0(dd);
            f = decision(xs);
            PROF_ADD(1, dd);
            PROF_CNT(9);
            if (!P.ok) PROF_CNT(10);
            else if (!(x >= P.lo && x <= P.hi)) { PROF_CNT(11); if (site == 1) PROF_CNT(15); if (site == 2) PROF_CNT(14); if (x < P.lo) PROF_CNT(13); }
            else PROF_CNT(12);
        }
        return f;
    }
    // the marching loops of material.py:475-480 (DOWN: while f0 >= 0 and x0 > 0.01: x0 *= 0.98) and :481-486 (up: while
    // f1 < 0 and x1 < 5 sflow: x1 *= 1.02), entered with the loop condition true at (x, fx); leaves the point the loop ends at
    // in (x, fx).  pw[i] = 0.98^i / 1.02^i as sequential products.
    template <bool DOWN>
    __device__ __forceinline__ void march(const double *su, const RowPoly &P, double sflow, double &x, double &fx) const
    {
        const double *pw = poly_tab() + NS * NS + (DOWN ? 0 : 64);
        const int l = threadIdx.x & (GS - 1);
        for (;;) {
            int k = 0;
            if (P.ok) {
                const double y = x * pw[l + 1];   // the (l + 1)-th next point of the march
                const double pv = P.eval(y);
                const bool on = y >= P.lo && y <= P.hi &&
                                (DOWN ? (pv > P.margin && y > 0.010000001) : (pv < -P.margin && y < 4.9999999 * sflow));
                const unsigned row = (unsigned)(__ballot(on) >> (threadIdx.x & 48)) & 0xffffu;
                k = (row == 0xffffu) ? GS : __ffs((int)~row) - 1;   // leading run of points at which the march goes on
            }
            // the products one by one: the point the march stops at is an end of brentq's bracket and carries the reference's rounding
            const int steps = (k == GS) ? GS : k + 1;
            for (int i = 0; i < steps; i++) x *= DOWN ? 0.98 : 1.02;
            if (k == GS) continue;
            fx = evalx(su, P, x, true);
            if (!(DOWN ? (fx >= 0. && x > 0.01) : (fx < 0. && x < 5. * sflow))) return;
        }
    }
    // ML_full_yf (material.py:414-516): distance to the yield locus along the ray through s (ld == nullptr) or along ld
    __device__ inline double full_ld(const double *s, const double *epl, const double *ld, int *status) const
    {
        const double seqv = seq(s);
        const double sflow = sflow_of(m, epl);
        if (status) *status = 0;
        if (seqv < 0.01 && ld == nullptr) return seqv - 0.85 * sflow;
        double su[6];
        if (ld == nullptr) {
            const double inv = 1. / seqv;
#pragma unroll
            for (int i = 0; i < 6; i++) su[i] = s[i] * inv;
        } else {
            double hh = 0.;   // su = ld * sqrt(1.5) / |ld|  (material.py:455-462)
#pragma unroll
            for (int i = 0; i < 6; i++) hh += ld[i] * ld[i];
            hh = sqrt(hh);
            if (hh < 1.e-3) {
                su[0] = sqrt(1.5);
#pragma unroll
                for (int i = 1; i < 6; i++) su[i] = 0.;
            } else {
#pragma unroll
                for (int i = 0; i < 6; i++) su[i] = ld[i] * sqrt(1.5) / hh;
            }
        }
        double x0 = sflow;
        const bool halved = su[0] * su[1] < -1.e-5;   // material.py:468-473
        if (halved) x0 *= 0.5;
        RowPoly P;
        PROF_T0(rs);
        ray_sample(su, x0, halved, P, ld == nullptr ? seqv : 0.);
        PROF_ADD(2, rs);
        PROF_CNT(8);

        PROF_T0(mm);
        double f0 = evalx(su, P, x0, true, 1), x1 = x0, f1 = f0;
        if (f0 >= 0. && x0 > 0.01) march<true>(su, P, sflow, x0, f0);
        if (f1 < 0. && x1 < 5. * sflow) march<false>(su, P, sflow, x1, f1);
        PROF_ADD(4, mm);
        if (f0 * f1 > 0.) {  // material.py:495-499
            if (status) *status = 1;
            return seqv - 0.85 * sflow;
        }
        // brentq(xtol = 1e-5) on the reference's bracket [x0, x1] (material.py:501-503), iterate for iterate (BrentState), with
        // the polynomial as the function wherever it covers the iterate: the reference's result is brentq's LAST ITERATE, up
        // to 1e-5 MPa away from the root, and the branch tests of response() (fy1 > toler, :310) read it -- a root finder
        // that converges further (measured: lane-parallel subdivision + secant, 1e-12) flips such a test in 1 of 60 000
        // seeded calls and moves that call's stress by 1.4e-3 sy; the replay on p differs from the reference by ~1e-11
        double xs;
        bool conv = true;
        {
            PROF_T0(bq);
            BrentState br;
            if (br.start(x0, x1, f0, f1)) {
                while (br.next(1.e-5, 4. * 2.220446049250313e-16, 100)) br.fcur = evalx(su, P, br.xcur, false, 2);
            }
            xs = br.root;
            conv = br.converged;
            PROF_ADD(5, bq);
        }
        if (conv && xs < 4. * sflow) return seqv - xs * seq(su);  // material.py:507
        if (status) *status = 2;
        return seqv - 0.85 * sflow;  // material.py:510
    }
    __device__ __forceinline__ double full(const double *s, const double *epl) const { return full_ld(s, epl, nullptr, nullptr); }
    __device__ __forceinline__ double full0(const double *s, double fy0) const   // material.py:265: epl = 0
    {
        (void)fy0;
        const double z[6] = {0., 0., 0., 0., 0., 0.};
        return full_ld(s, z, nullptr, nullptr);
    }
    __device__ __forceinline__ double kh() const { return m.khard; }
    __device__ __forceinline__ int touched() const { return 0; }
    __device__ __forceinline__ double sflow(const double *epl) const { return sflow_of(m, epl); }
    __device__ __forceinline__ double sflow_entry(const double *epl) const { return sflow_of(m, epl); }
};

typedef YfSvcWhT<0> YfSvcWh;
typedef YfSvcT<6> YfSvc;
typedef YfSvcT<2> YfSvc3;
template <int NC, bool POLY = false>
using YfSvcWave = YfSvcT<6, NC, POLY>;

// ---------------------------------------------------------------------------------------------
// Material.response (material.py:207-346) for one point, in two phases so that the sweep can run
// the cheap, memory-bound part at high occupancy and compact the expensive part (SURVEY "divergence"):
//
// response_light: elastic predictor (:243-256), elastic/plastic split (:259-274) and the trial step
//   with the full remaining increment (:277-293).  Returns 0 = purely elastic step, 1 = the trial
//   ended inside the tolerance (the reference then runs its loop exactly once on the same inputs and
//   reproduces the trial's numbers, so the trial IS the result), 2 = the increment has to be
//   sub-divided: sig then holds the stress at the start of the plastic part, deps_r the remaining
//   increment and st_scal the plastic share; fy/depl/Ct are not valid yet.
// response_heavy: the maxit = 50 sub-steps with radial scale-back (:295-344).
// In/out: sig (updated to the end of the step).  Out: fy, depl, Ct (21 symmetric entries).
// yf.fgrad(sig) followed by yf.plain(s2): one pass over the support vectors for policies that offer it (YfSvcRow)
template <class YF>
__device__ __forceinline__ auto grad_and_yf(const YF &yf, const double *sig, const double *epl, const double *s2, double *a)
    -> decltype(yf.fgrad_plain(sig, epl, s2, a))
{
    return yf.fgrad_plain(sig, epl, s2, a);
}
template <class YF, class... X>
__device__ __forceinline__ double grad_and_yf(const YF &yf, const double *sig, const double *epl, const double *s2, double *a, X...)
{
    yf.fgrad(sig, epl, a);
    return yf.plain(s2, epl);
}

template <class YF>
__device__ inline int response_light(const MatDev &m, const YF &yf, double *sig, const double *epl,
                                     const double *deps, double &fy, double *depl, double *Ct,
                                     double *deps_r, double &st_scal)
{
    const double *CV = m.CV;
    double dsig[6], tmp[6];
#pragma unroll
    for (int i = 0; i < 6; i++) depl[i] = 0.;
    const double sflow0 = yf.sflow_entry(epl);
    const double toler = YF_TOL * sflow0;  // :243
    symv(CV, deps, dsig);                  // :244
#pragma unroll
    for (int i = 0; i < 6; i++) tmp[i] = sig[i] + dsig[i];
    double fy1 = yf.full(tmp, epl);  // :249-252
    if (fy1 < toler) {               // purely elastic step :253-256
#pragma unroll
        for (int i = 0; i < 6; i++) sig[i] = tmp[i];
#pragma unroll
        for (int i = 0; i < 21; i++) Ct[i] = CV[i];
        fy = fy1;
        return 0;
    }
    st_scal = 1.;
    double fy0 = yf.plain(sig, epl);  // :259
    if (fy0 < SPLIT_THRESHOLD) {      // :260-270 split into elastic + plastic part
        fy0 = yf.full0(sig, fy0);
        st_scal += fy0 / yf.seq(dsig);
        const double wel = 1. - st_scal;
        double de[6], ds[6];
#pragma unroll
        for (int i = 0; i < 6; i++) de[i] = deps[i] * wel;
        symv(CV, de, ds);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            sig[i] += ds[i];
            deps_r[i] = deps[i] - de[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) deps_r[i] = deps[i];
    }
    double a[6], ca[6], dsr[6], ddepl[6], eplt[6];
    // trial step with the full remaining increment (:277-293)
    {
        symv(CV, deps_r, dsr);
#pragma unroll
        for (int i = 0; i < 6; i++) tmp[i] = sig[i] + dsr[i];
        const double yfun = grad_and_yf(yf, sig, epl, tmp, a);  // gradient at sig; epl_dot's calc_yf(sig + dsig) :1032, absolute tolerance :1041
        symv(CV, a, ca);
        const double hh = dot6(a, ca) + yf.kh();
        const double lam = (yfun <= YF_TOL) ? 0. : dot6(a, dsr) / hh;
        const double cd = dot6(ca, deps_r) / hh;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            ddepl[i] = lam * a[i];
            eplt[i] = epl[i] + ddepl[i];
            tmp[i] = sig[i] + (dsr[i] - ca[i] * cd);
        }
        fy1 = yf.full(tmp, eplt);
        if (!(fy1 > toler)) {  // nsteps = 1 (:292-293)
            const double w1 = st_scal / hh;
#pragma unroll
            for (int i = 0; i < 6; i++) {
                sig[i] = tmp[i];
                depl[i] = ddepl[i];
            }
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = i; j < 6; j++) Ct[sym_idx(i, j)] = fma(-w1 * ca[i], ca[j], CV[sym_idx(i, j)]);
            fy = fy1;
            return 1;
        }
    }
    return 2;
}

template <class YF>
__device__ inline void response_heavy(const MatDev &m, const YF &yf, double *sig, const double *epl,
                                      double *deps_r, double st_scal, double &fy, double *depl,
                                      double *Ct, const int maxit = MAXIT /* the sweeps of a model: the reference's default, a constant */)
{
    const double *CV = m.CV;
    const double toler = YF_TOL * yf.sflow_entry(epl);  // :243 (with the hardening modulus at the entry of the call)
    double a[6], ca[6], dsr[6], ddepl[6], eplt[6], tmp[6], fy1;
    // sub-divided step (:288-291): nsteps = maxit
    const int nsteps = maxit;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        deps_r[i] /= maxit;
        depl[i] = 0.;
    }
    symv(CV, deps_r, dsr);
    // R accumulates sum_it [ (Ca)(Ca)^T/h + corr3 ]; Ct = CV - (st_scal/nsteps) R at the end
    // (:343 with T = CV - (Ca)(Ca)^T/h - corr3 and the elastic share CV*(1-st_scal) of :269).
    double R[21];
#pragma unroll
    for (int i = 0; i < 21; i++) R[i] = 0.;
    PROF_T0(w);
    for (int it = 0; it < nsteps; it++) {  // :295
#pragma unroll
        for (int i = 0; i < 6; i++) tmp[i] = sig[i] + dsr[i];
        PROF_T0(g);
        const double yfun = grad_and_yf(yf, sig, epl, tmp, a);  // NB entry epl, absolute tolerance (:299, :1041)
        PROF_ADD(0, g);
        symv(CV, a, ca);
        const double hh = dot6(a, ca) + yf.kh();
        const double lam = (yfun <= YF_TOL) ? 0. : dot6(a, dsr) / hh;
        const double cd = dot6(ca, deps_r) / hh;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            ddepl[i] = lam * a[i];
            sig[i] += dsr[i] - ca[i] * cd;
            eplt[i] = epl[i] + depl[i] + ddepl[i];
        }
        {   // (before the ray search, which it does not depend on: a, ca, hh are dead across yf.full())
            const double ih = 1. / hh;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = i; j < 6; j++) R[sym_idx(i, j)] = fma(ca[i] * ih, ca[j], R[sym_idx(i, j)]);
        }
        PROF_T0(f);
        fy1 = yf.full(sig, eplt);
        PROF_ADD(6, f);
        if (fy1 > toler) {  // radial scale-back :310-342
            const double sq = yf.seq(sig);
            const double fr = fy1 / sq;
            double dsg[6], sd[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                dsg[i] = sig[i] * fr;
                sig[i] -= dsg[i];
            }
            symv(m.SV, dsg, sd);
#pragma unroll
            for (int i = 0; i < 6; i++) {
                ddepl[i] += sd[i];
                eplt[i] = epl[i] + depl[i] + ddepl[i];
            }
            // min-norm solution of the 3x6 system (:325-337): x = A^T G^{-1} b,
            // G = |d|^2 I + d d^T - diag(d^2), d = deps_r[0:3], b = dsg[0:3]
            const double d0 = deps_r[0], d1 = deps_r[1], d2 = deps_r[2];
            const double n2 = d0 * d0 + d1 * d1 + d2 * d2;
            if (n2 > 0.) {
                const double g01 = d0 * d1, g02 = d0 * d2, g12 = d1 * d2;
                const double c00 = n2 * n2 - g12 * g12, c01 = g02 * g12 - g01 * n2,
                             c02 = g01 * g12 - g02 * n2;
                const double c11 = n2 * n2 - g02 * g02, c12 = g01 * g02 - g12 * n2,
                             c22 = n2 * n2 - g01 * g01;
                const double det = n2 * c00 + g01 * c01 + g02 * c02;
                const double y0 = (c00 * dsg[0] + c01 * dsg[1] + c02 * dsg[2]) / det;
                const double y1 = (c01 * dsg[0] + c11 * dsg[1] + c12 * dsg[2]) / det;
                const double y2 = (c02 * dsg[0] + c12 * dsg[1] + c22 * dsg[2]) / det;
                R[sym_idx(0, 0)] += d0 * y0;            // x0
                R[sym_idx(1, 1)] += d1 * y1;            // x1
                R[sym_idx(2, 2)] += d2 * y2;            // x2
                R[sym_idx(1, 2)] += d2 * y1 + d1 * y2;  // x3
                R[sym_idx(0, 2)] += d2 * y0 + d0 * y2;  // x4
                R[sym_idx(0, 1)] += d1 * y0 + d0 * y1;  // x5
            }
            // "update yield function" (:339-342): inside the loop the value is overwritten by the next sub-step's :303-306
            // before anything reads it -- only the one of the last sub-step is returned, so only that one is evaluated
            // (for an SVC this is a whole ML_full_yf ray search per scaled-back sub-step)
            if (it == nsteps - 1) fy1 = yf.full(sig, eplt);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) depl[i] += ddepl[i];  // :344
    }
    PROF_ADD(7, w);
    const double w = st_scal / nsteps;
#pragma unroll
    for (int i = 0; i < 21; i++) Ct[i] = fma(-w, R[i], CV[i]);
    fy = fy1;
}

// both phases; returns msg['nsteps'] (last loop index, :345)
template <class YF>
__device__ inline int response_point(const MatDev &m, const YF &yf, double *sig, const double *epl,
                                     const double *deps, double &fy, double *depl, double *Ct, const int maxit = MAXIT)
{
    double deps_r[6], st_scal;
    const int st = response_light(m, yf, sig, epl, deps, fy, depl, Ct, deps_r, st_scal);
    if (st < 2) return 0;
    response_heavy(m, yf, sig, epl, deps_r, st_scal, fy, depl, Ct, maxit);
    return maxit - 1;
}

}  // namespace plfx
