// plfx_kernels.hpp — HIP kernels of libplfx (gfx950 / MI355X).
//
// Kernels (one wavefront = 64 lanes, blocks of 256 threads = 4 waves, one per SIMD):
//   k_response_batch  Material.response on N independent points (AoS in/out; façade + parity entry)
//   k_point_eval      calc_seq / calc_fgrad / calc_yf / ML_full_yf on N points
//   k_sweep           model.py:1340-1359: strain gather + response + tangent test/refresh, SoA state
//   k_assemble        model.py:954-977 as a deterministic gather into the block-ELL matrix
//   k_spmv            q = K p (block-ELL, one thread per node = two rows) [+ fused p update + p.q]
//   k_cg_update       x,r,z update + r.z, r.r partials
//   k_update_state    model.py:1383-1392
// All reductions are two-stage with a fixed grid and are summed in index order, so results are
// bitwise reproducible run to run.
#pragma once
#include "plfx_device.hpp"

namespace plfx {

constexpr int BLOCK = 256;
constexpr int MAXMAT = 16;
constexpr int MAXCLS = 16;
constexpr int MAXPART = 1024;  // max blocks of a reducing kernel (partials per scalar)
#ifndef PLFX_HEAVY_THREADS
#define PLFX_HEAVY_THREADS 512  // threads per workgroup of the wave-per-element SVC corrector: two waves per SIMD at 256 VGPRs + 592 B of scratch beat one wave per SIMD at 424 registers (766 -> 589 ms per 16 x 16384 element updates)
#endif
#ifndef PLFX_SWEEP_WAVES
#define PLFX_SWEEP_WAVES 1  // min waves per SIMD the sweep kernel is compiled for (register budget)
#endif

// Element class = (material, lx, ly): everything the element routines need besides the state.
// B-matrix structure (model.py:475-501): B[0][2a] = B[5][2a+1] = bx_a, B[1][2a+1] = B[5][2a] = by_a,
// plane stress adds B[2][j] = kappa (B[0][j] + B[1][j]) with kappa = -nu (C11+C12)/E.
struct ClassDev {
    double bxs[4], bys[4];             // sum over the 4 Gauss points (strain operator, model.py:387-411)
    double Sxx[16], Sxy[16], Syy[16];  // Jac * sum_gp b?_a b?_b  (stiffness integrals, model.py:365-370)
    double kappa, vel, lx, ly;
    int32_t mat, _pad;
};

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide sum in fixed order; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double *sh /* [BLOCK/64] */)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = 0.;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < BLOCK / 64; i++) t += sh[i];
    }
    return t;
}

// NA block-wide sums at once, written to part[k * gridDim.x + blockIdx.x]: the same operation order per scalar as block_sum
// (wave sums, then the BLOCK / 64 wave results added in wave order: bit-identical), but one barrier instead of 2 NA
template <int NA>
__device__ __forceinline__ void block_sums_to_partials(const double (&v)[NA], double *__restrict__ part)
{
    __shared__ double shn[NA][BLOCK / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NA; k++) {
        const double t = wave_sum(v[k]);
        if (lane == 0) shn[k][w] = t;
    }
    __syncthreads();
    if (threadIdx.x < NA) {
        double t = 0.;
#pragma unroll
        for (int i = 0; i < BLOCK / 64; i++) t += shn[threadIdx.x][i];
        part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = t;
    }
}

// every block sums the same partial array in the same order -> identical scalar on all blocks
__device__ __forceinline__ double sum_partials(const double *part, int n, double *sh)
{
    double v = 0.;
    for (int i = threadIdx.x; i < n; i += BLOCK) v += part[i];
    double t = block_sum(v, sh);
    __shared__ double bc;
    if (threadIdx.x == 0) bc = t;
    __syncthreads();
    t = bc;
    __syncthreads();
    return t;
}

// the same for up to three arrays at once: one round of loads and one set of barriers instead of one per scalar
// (identical operation order per array, hence bit-identical results)
template <int NA>
__device__ __forceinline__ void sum_partials_n(const double *const (&part)[NA], int n, double (&out)[NA])
{
    __shared__ double shn[NA][BLOCK / 64];
    __shared__ double bcn[NA];
    double v[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) v[a] = 0.;
    for (int i = threadIdx.x; i < n; i += BLOCK) {
#pragma unroll
        for (int a = 0; a < NA; a++) v[a] += part[a][i];
    }
#pragma unroll
    for (int a = 0; a < NA; a++) v[a] = wave_sum(v[a]);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < NA; a++) shn[a][w] = v[a];
    }
    __syncthreads();
    if (threadIdx.x < NA) {
        double t = 0.;
#pragma unroll
        for (int i = 0; i < BLOCK / 64; i++) t += shn[threadIdx.x][i];
        bcn[threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NA; a++) out[a] = bcn[a];
    __syncthreads();
}

__host__ __device__ constexpr bool kind_is_svc(int k) { return k == 3 || k == 6 || k == 7; }

// XCD-aware tile order: consecutive block ids land on different XCDs (b % 8); give each XCD a
// contiguous range of tiles so that the neighbour columns a tile gathers are in the same L2.
__device__ __forceinline__ int xcd_tile(int b, int nb)
{
    const int per = nb >> 3;
    if (per == 0 || (nb & 7)) return b;
    return (b & 7) * per + (b >> 3);
}

__device__ __forceinline__ void stage_materials(MatDev *smat, const MatDev *gmat, int nmat)
{
    const int words = nmat * (int)(sizeof(MatDev) / 8);
    const double *src = reinterpret_cast<const double *>(gmat);
    double *dst = reinterpret_cast<double *>(smat);
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
}

// stage the support vectors of the first SVC material into dynamic LDS (if they fit)
__device__ __forceinline__ void stage_svc(const MatDev *smat, int nmat, double *lds, int lds_doubles,
                                          int &svc_mat, const double *&sv, const double *&dual, int kind = 0)
{
    svc_mat = -1;
    sv = dual = nullptr;
    for (int k = 0; k < nmat; k++)
        if ((kind ? smat[k].kind == kind : kind_is_svc(smat[k].kind)) &&
            smat[k].nsv * (smat[k].nfeat + 1) <= lds_doubles) {
            svc_mat = k;
            break;
        }
    if (svc_mat < 0) return;
    const int n = smat[svc_mat].nsv, nf = smat[svc_mat].nfeat;
    for (int i = threadIdx.x; i < nf * n; i += blockDim.x) lds[i] = smat[svc_mat].sv[i];
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[nf * n + i] = smat[svc_mat].dual[i];
    sv = lds;
    dual = lds + nf * n;
}

// One material kind per kernel instantiation (KIND = 1 Hill-6p/J2 on Voigt, 2 Hill-3p/J2 on principal
// stresses, 3 RBF-SVC): the analytic kernels do not carry the SVC code (and its registers); every
// instantiation skips the elements of the other kinds.
template <int KIND>
struct YfOf;
template <>
struct YfOf<1> {
    typedef YfHill type;
    __device__ static YfHill make(const MatDev &m, const double *, const double *) { return YfHill(m); }
};
template <>
struct YfOf<2> {
    typedef YfPrinc3 type;
    __device__ static YfPrinc3 make(const MatDev &m, const double *, const double *) { return YfPrinc3(m); }
};
template <>
struct YfOf<3> {
    typedef YfSvc type;
    __device__ static YfSvc make(const MatDev &m, const double *sv, const double *dual)
    {
        return YfSvc(m, sv ? sv : m.sv, dual ? dual : m.dual);
    }
};
template <>
struct YfOf<5> {
    typedef YfBarlat type;
    __device__ static YfBarlat make(const MatDev &m, const double *, const double *) { return YfBarlat(m); }
};
template <>
struct YfOf<6> {
    typedef YfSvc3 type;
    __device__ static YfSvc3 make(const MatDev &m, const double *sv, const double *dual)
    {
        return YfSvc3(m, sv ? sv : m.sv, dual ? dual : m.dual);
    }
};
template <>
struct YfOf<7> {   // SVC with work-hardening features: the hardening modulus is state (YfSvcWh), set by make_wh
    typedef YfSvcWh type;
    __device__ static YfSvcWh make(const MatDev &m, const double *sv, const double *dual)
    {
        return YfSvcWh(m, sv ? sv : m.sv, dual ? dual : m.dual, m.khard);
    }
};
// policy object of one material point: KIND 7 takes the point's hardening modulus kh0, the others ignore it
template <int KIND>
__device__ __forceinline__ typename YfOf<KIND>::type make_policy(const MatDev &m, const double *sv, const double *dual, double kh0)
{
    if constexpr (KIND == 7)
        return YfSvcWh(m, sv ? sv : m.sv, dual ? dual : m.dual, kh0);
    else
        return YfOf<KIND>::make(m, sv, dual);
}



// ---------------------------------------------------------------------------------------------
// Material.response on n points, host-layout (AoS) arrays.
template <int KIND>
__global__ void __launch_bounds__(BLOCK)
k_response_batch(const MatDev *gmat, int nmat, int lds_doubles, int n, const int32_t *mat_id,
                 const double *sig_in, const double *epl_in, const double *deps_in, double *fy,
                 double *sig_out, double *depl_out, double *ct_out, int32_t *nsteps,
                 const double *kh_in = nullptr, double *kh_out = nullptr, unsigned skip_mask = 0u /* materials run by k_response_row */,
                 int maxit = MAXIT /* Material.response(..., maxit): sub-steps of a sub-divided increment (material.py:207, 288-291) */)
{
    __shared__ MatDev smat[MAXMAT];
    stage_materials(smat, gmat, nmat);
    __syncthreads();
    int svc_mat = -1;
    const double *sv = nullptr, *dual = nullptr;
    if (kind_is_svc(KIND)) {
        stage_svc(smat, nmat, dyn_lds, lds_doubles, svc_mat, sv, dual, KIND);
        __syncthreads();
    }
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const int mid = mat_id ? mat_id[i] : 0;
        const MatDev &m = smat[mid];
        if (m.kind != KIND && !(m.kind == 0 && KIND == 1)) continue;  // handled by another instantiation
        if ((skip_mask >> mid) & 1u) continue;
        double sig[6], epl[6], deps[6], depl[6], Ct[21], f = 0.;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            sig[c] = sig_in[6 * (size_t)i + c];
            epl[c] = epl_in[6 * (size_t)i + c];
            deps[c] = deps_in[6 * (size_t)i + c];
        }
        int ns = 0;
        if (m.kind == 0) {
#pragma unroll
            for (int c = 0; c < 6; c++) depl[c] = 0.;
#pragma unroll
            for (int c = 0; c < 21; c++) Ct[c] = m.CV[c];
        } else {
            const bool staged = (mid == svc_mat);
            const typename YfOf<KIND>::type yf =
                make_policy<KIND>(m, staged ? sv : nullptr, staged ? dual : nullptr, kh_in ? kh_in[i] : m.khard);
            ns = response_point(m, yf, sig, epl, deps, f, depl, Ct, maxit);
            if (KIND == 7 && kh_out) kh_out[i] = yf.kh();
        }
        fy[i] = f;
        nsteps[i] = ns;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            sig_out[6 * (size_t)i + c] = sig[c];
            depl_out[6 * (size_t)i + c] = depl[c];
        }
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) ct_out[36 * (size_t)i + r * 6 + c] = Ct[sym_idx(r, c)];
    }
}

// calc_fgrad(sig, seq=...) of the analytic Hill materials with the equivalent stress handed in (material.py:834-847): the
// deviator of the VOIGT components over 2 seq; sdim == 3 materials get no shear rows (:841 is skipped).  That is what the
// reference's point function returns for a (6,) stress of a principal-stress material -- seq from sig_princ's order, the
// deviator from the Voigt normals -- and differs from the principal-space normal that epl_dot / C_tan use (princ_fgrad)
// as soon as the state has shear.
__global__ void __launch_bounds__(BLOCK)
k_fgrad_seq(const MatDev *gmat, int mat, int n, const double *sig_in, const double *seq_in, double *out)
{
    const MatDev &m = gmat[mat];
    const bool six = (m.kind != 2);
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        double s[6];
#pragma unroll
        for (int c = 0; c < 6; c++) s[c] = sig_in[6 * (size_t)i + c];
        const double seq = seq_in[i];
        const double p = (s[0] + s[1] + s[2]) / 3.;
        const double s0 = s[0] - p, s1 = s[1] - p, s2 = s[2] - p;
        const double h0 = m.hill[0], h1 = m.hill[1], h2 = m.hill[2], d3 = m.d0 / 3.;
        double *a = out + 6 * (size_t)i;
        a[0] = ((h0 + h2) * s0 - h0 * s1 - h2 * s2) / (2. * seq) + d3;
        a[1] = ((h1 + h0) * s1 - h0 * s0 - h1 * s2) / (2. * seq) + d3;
        a[2] = ((h2 + h1) * s2 - h2 * s0 - h1 * s1) / (2. * seq) + d3;
        a[3] = six ? 3. * m.hill[3] * s[3] / seq : 0.;
        a[4] = six ? 3. * m.hill[4] * s[4] / seq : 0.;
        a[5] = six ? 3. * m.hill[5] * s[5] / seq : 0.;
    }
}

// what: 0 calc_seq, 1 calc_fgrad, 2 calc_yf, 3 ML_full_yf (SVC) / calc_yf (analytic)
__global__ void __launch_bounds__(BLOCK)
k_point_eval(const MatDev *gmat, int nmat, int lds_doubles, int what, int mat, int n,
             const double *sig_in, const double *epl_in, const double *ld, double *out,
             int32_t *status, double *kh_raw = nullptr /* kind 7, what 1: - sum dK/dx[wh] scale_seq / scale_wh per point */)
{
    __shared__ MatDev smat[MAXMAT];
    stage_materials(smat, gmat, nmat);
    __syncthreads();
    int svc_mat;
    const double *sv, *dual;
    stage_svc(smat, nmat, dyn_lds, lds_doubles, svc_mat, sv, dual);
    __syncthreads();
    const MatDev &m = smat[mat];
    const bool svc = kind_is_svc(m.kind);
    const bool svc3 = (m.kind == 6);
    const double *psv = (mat == svc_mat) ? sv : m.sv;
    const double *pdu = (mat == svc_mat) ? dual : m.dual;
    double ldv[6];
    if (ld) {
#pragma unroll
        for (int c = 0; c < 6; c++) ldv[c] = ld[c];
    }
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        double s[6], e[6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
            s[c] = sig_in[6 * (size_t)i + c];
            e[c] = epl_in ? epl_in[6 * (size_t)i + c] : 0.;
        }
        const int kd = m.kind;
        if (kd == 7 && what >= 1) {  // SVC with work-hardening features: the plastic strain is part of the feature vector
            const YfSvcWh yf(m, psv, pdu, m.khard);
            if (what == 1) {
                double a[6];
                const double hk = yf.fgrad_raw(s, e, a);
#pragma unroll
                for (int c = 0; c < 6; c++) out[6 * (size_t)i + c] = a[c];
                if (kh_raw) kh_raw[i] = hk;
            } else if (what == 2) {
                out[i] = yf.plain(s, e);
            } else {
                int st = 0;
                out[i] = yf.full_ld(s, e, ld ? ldv : nullptr, &st);
                if (status) status[i] = st;
            }
            continue;
        }
        if (what == 0) {
            out[i] = (kd == 2 || kd == 6) ? princ_seq(m, s) : kd == 4 ? tresca_seq(s) : kd == 5 ? barlat_seq(m, s) : hill_seq(m, s);
        } else if (what == 1) {
            double a[6];
            if (svc3)
                svc3_fgrad(m, psv, pdu, s, a);
            else if (svc)
                svc_fgrad(m, psv, pdu, s, a);
            else if (kd == 2)
                princ_fgrad(m, s, a);
            else if (kd == 5)
                barlat_seq_grad(m, s, a);
            else
                hill_fgrad(m, s, a);
#pragma unroll
            for (int c = 0; c < 6; c++) out[6 * (size_t)i + c] = a[c];
        } else if (what == 2) {
            out[i] = svc3 ? svc3_decision(m, psv, pdu, s)
                          : svc ? svc_decision(m, psv, pdu, s)
                                : (kd == 2 ? princ_seq(m, s) : kd == 5 ? barlat_seq(m, s) : hill_seq(m, s)) - sflow_of(m, e);
        } else {
            int st = 0;
            if (svc3) {
                YfSvc3 yf(m, psv, pdu);
                out[i] = yf.full_ld(s, e, ld ? ldv : nullptr, &st);
            } else if (svc) {
                YfSvc yf(m, psv, pdu);
                out[i] = yf.full_ld(s, e, ld ? ldv : nullptr, &st);
            } else {
                out[i] = (kd == 2 ? princ_seq(m, s) : kd == 5 ? barlat_seq(m, s) : hill_seq(m, s)) - sflow_of(m, e);
            }
            if (status) status[i] = st;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// element strain from nodal values: (sum_gp B) u_e   (model.py:387-411)
__device__ __forceinline__ void class_strain(const ClassDev &c, const double2 *u2, int n0, int n1,
                                             int n2, int n3, double *e)
{
    const double2 u0 = u2[n0], u1 = u2[n1], u2_ = u2[n2], u3 = u2[n3];
    const double ex = c.bxs[0] * u0.x + c.bxs[1] * u1.x + c.bxs[2] * u2_.x + c.bxs[3] * u3.x;
    const double ey = c.bys[0] * u0.y + c.bys[1] * u1.y + c.bys[2] * u2_.y + c.bys[3] * u3.y;
    const double gxy = c.bys[0] * u0.x + c.bxs[0] * u0.y + c.bys[1] * u1.x + c.bxs[1] * u1.y +
                       c.bys[2] * u2_.x + c.bxs[2] * u2_.y + c.bys[3] * u3.x + c.bxs[3] * u3.y;
    e[0] = ex;
    e[1] = ey;
    e[2] = c.kappa * (ex + ey);
    e[3] = 0.;
    e[4] = 0.;
    e[5] = gxy;
}

// compact element stiffness generator M = [X Y S]^T D [X Y S], X = e0 + kappa e2, Y = e1 + kappa e2,
// S = e5: the six numbers from which every 2x2 block of B^T D B follows (see k_assemble).
__device__ __forceinline__ void tangent_to_M(const double *D, double kappa, double *M)
{
    const double k = kappa;
    M[0] = D[sym_idx(0, 0)] + k * (2. * D[sym_idx(0, 2)] + k * D[sym_idx(2, 2)]);                  // XX
    M[1] = D[sym_idx(0, 1)] + k * (D[sym_idx(0, 2)] + D[sym_idx(1, 2)] + k * D[sym_idx(2, 2)]);    // XY
    M[2] = D[sym_idx(0, 5)] + k * D[sym_idx(2, 5)];                                                // XS
    M[3] = D[sym_idx(1, 1)] + k * (2. * D[sym_idx(1, 2)] + k * D[sym_idx(2, 2)]);                  // YY
    M[4] = D[sym_idx(1, 5)] + k * D[sym_idx(2, 5)];                                                // YS
    M[5] = D[sym_idx(5, 5)];                                                                       // SS
}

// Tail of the per-element sweep (model.py:1343-1357): store the response, yield-function ratio,
// tangent test ||elstiff - Ct||_F > 1e-3 and tangent / stiffness-generator refresh.
__device__ __forceinline__ void sweep_epilogue(const ClassDev &c, const MatDev &m, int e, int nel,
                                               const double *s, const double *ep, const double *depl,
                                               double *Ct, double fy, int ns, double *elstiff,
                                               double *Mel, int mel_stride, double *res_sig,
                                               double *res_depl, double *fyn, int32_t *max_steps, int nit,
                                               int &changed, int &nconv, double kh = -1.)
{
#pragma unroll
    for (int k = 0; k < 6; k++) {
        res_sig[(size_t)k * nel + e] = s[k];
        res_depl[(size_t)k * nel + e] = depl[k];
    }
    const double f = fy / (kh >= 0. ? m.sy + eps_eq(ep) * kh : sflow_of(m, ep));  // model.py:1345 (kh: the modulus a work-hardening SVC left behind)
    fyn[e] = f;
    if (!(f <= YF_TOL * 1.0001)) nconv = 1;  // model.py:1361
    // Frobenius norm of the tangent change over the full 6x6 (model.py:1346)
    double hh = 0.;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) {
            const double d = elstiff[(size_t)sym_idx(i, j) * nel + e] - Ct[sym_idx(i, j)];
            hh += (i == j ? 1. : 2.) * d * d;
        }
    hh = sqrt(hh);
    if (hh > 1.e-3) {  // model.py:1348-1355
        if (nit >= 15) {
#pragma unroll
            for (int k = 0; k < 21; k++) Ct[k] = 0.5 * (Ct[k] + elstiff[(size_t)k * nel + e]);
        }
#pragma unroll
        for (int k = 0; k < 21; k++) elstiff[(size_t)k * nel + e] = Ct[k];
        double M[6];
        tangent_to_M(Ct, c.kappa, M);
#pragma unroll
        for (int k = 0; k < 6; k++) Mel[(size_t)k * mel_stride + e] = M[k];
        changed += 1;  // counts the elements whose tangent (and generator) this thread rewrote
    }
    if (ns > max_steps[e]) max_steps[e] = ns;  // stat_nlin['max_steps'] (model.py:1356)
}

struct SweepTables {
    MatDev smat[MAXMAT];
    ClassDev scls[MAXCLS];
};

__device__ __forceinline__ void stage_tables(SweepTables &t, const MatDev *gmat, int nmat,
                                             const ClassDev *gcls, int ncls)
{
    stage_materials(t.smat, gmat, nmat);
    const int words = ncls * (int)(sizeof(ClassDev) / 8);
    const double *src = reinterpret_cast<const double *>(gcls);
    double *dst = reinterpret_cast<double *>(t.scls);
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
}

// Result flags of a sweep without atomics on one address (4096 per-wave atomics on one cache line cost ~40 us of the
// streaming sweep): every block adds its counts to its own slot of `bflags` (blocks of consecutive launches on the stream
// share slots, memset once per sweep), k_sweep_flags reduces the slots after the last launch of the sweep.
constexpr int SWEEP_SLOTS = 1024;  // >= every sweep grid (grid_xcd / grid_w cap their grids at MAXPART = 1024)
static_assert(SWEEP_SLOTS >= MAXPART, "one slot per block of a sweep grid");

__device__ __forceinline__ void post_block_flags(int changed, int nconv, int *__restrict__ bflags)
{
    __shared__ int sh_c[16], sh_n[16];
    int cw = changed, nw = nconv ? 1 : 0;
    for (int o = 32; o; o >>= 1) {
        cw += __shfl_xor(cw, o, 64);
        nw |= __shfl_xor(nw, o, 64);
    }
    const int wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh_c[wave] = cw;
        sh_n[wave] = nw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c2 = 0, n2 = 0;
        for (int w = 0; w < nwaves; w++) {
            c2 += sh_c[w];
            n2 |= sh_n[w];
        }
        if (c2) bflags[2 * blockIdx.x] += c2;
        if (n2) bflags[2 * blockIdx.x + 1] = 1;
    }
}

// flags[0] = any tangent changed, flags[1] = any element not converged, flags[3] = tangents rewritten (flags[2], the length
// of the compacted list, is maintained by the kernels themselves)
// The kernel leaves the slots and the list length zeroed for the next sweep (no memset launches between sweeps) and hands the
// four results over in out[0..3].
struct CgMbox;
__device__ void mbox_publish(CgMbox *mb, unsigned long long seq);   // (defined with CgMbox below)
// pin != nullptr (single GPU, mailbox): the four results also go to the pinned buffer and the sequence number is posted from
// here -- no separate k_mbox_post launch before the host may read them
// skip_setup / spec (plfx_load_step, single GPU): predicates of the two kernels enqueued BEHIND this one before the host has read
// the flags -- the set-up pass of the next stiffness iteration runs only if a tangent changed (*skip_setup = 0), the K du of the
// end of the load step only if the K-iteration loop ends here (spec->done = 0: nothing changed and every element converged)
struct CgScalars;
__device__ void spec_set(CgScalars *spec, int done);
__global__ void __launch_bounds__(BLOCK) k_sweep_flags(int *__restrict__ bflags, int *__restrict__ flags, int *__restrict__ out,
                                                       int *__restrict__ pin = nullptr, CgMbox *mb = nullptr, unsigned long long seq = 0ull,
                                                       int *__restrict__ skip_setup = nullptr, CgScalars *spec = nullptr)
{
    __shared__ int sc[BLOCK / 64], sn[BLOCK / 64];
    int cw = 0, nw = 0;
    for (int b = threadIdx.x; b < SWEEP_SLOTS; b += BLOCK) {
        cw += bflags[2 * b];
        nw |= bflags[2 * b + 1];
        bflags[2 * b] = 0;
        bflags[2 * b + 1] = 0;
    }
    for (int o = 32; o; o >>= 1) {
        cw += __shfl_xor(cw, o, 64);
        nw |= __shfl_xor(nw, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sc[threadIdx.x >> 6] = cw;
        sn[threadIdx.x >> 6] = nw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c2 = 0, n2 = 0;
        for (int w = 0; w < BLOCK / 64; w++) {
            c2 += sc[w];
            n2 |= sn[w];
        }
        out[0] = c2 ? 1 : 0;
        out[1] = n2;
        out[2] = flags[2];
        out[3] = c2;
        if (skip_setup) *skip_setup = c2 ? 0 : 1;
        if (spec) spec_set(spec, (c2 == 0 && n2 == 0) ? 0 : 1);
        if (pin) {
            pin[0] = c2 ? 1 : 0;
            pin[1] = n2;
            pin[2] = flags[2];
            pin[3] = c2;
            mbox_publish(mb, seq);
        }
        flags[2] = 0;
    }
}

// Material sweep over the owned elements (model.py:1340-1359), phase 1: elastic and one-step plastic
// elements are finished here (streaming, HBM-bound); elements whose increment must be sub-divided are
// appended to `list` (one atomic per wave) for k_sweep_heavy.  SoA state: component c of element e at
// [c*nel + e].  flags[0] |= changed, flags[1] |= not converged, flags[2] = length of `list`, flags[3] = tangents rewritten.
// Material / class tables are staged in LDS (wave-uniform addresses -> broadcast reads); holding them
// in SGPRs instead (scalar loads + waterfall over classes) was measured 35 % slower (SGPR spills).
// (KIND 2, principal-stress Hill: two waves per SIMD as before the out-of-line LAPACK replay of sig_princ_general was
// added -- the call must not cost the streaming path its occupancy; the spills it takes sit on the cold side of that branch)
template <int KIND>
__global__ void __launch_bounds__(BLOCK, (KIND == 2 && PLFX_SWEEP_WAVES < 2) ? 2 : PLFX_SWEEP_WAVES)
k_sweep_light(const MatDev *__restrict__ gmat, int nmat, const ClassDev *__restrict__ gcls, int ncls,
              int lds_doubles, int nel, int e_off, const int32_t *__restrict__ conn,
              const int32_t *__restrict__ cls, const double2 *__restrict__ du2,
              const double *__restrict__ sig, const double *__restrict__ epl, double *elstiff,
              double *Mel, int mel_stride, double *res_sig, double *res_depl, double *fyn,
              int32_t *max_steps, int nit, int *flags, int *bflags, int32_t *list, int first_kind, unsigned skip_mask,
              double *kh_el = nullptr /* KIND 7: hardening modulus of every material point, carried from sweep to sweep */,
              double *kh_out = nullptr, int32_t *kh_touch = nullptr /* sequential-carry mode: kh_el is read only; the exit
              modulus and "a gradient evaluation overwrote it" go here */)
{
    __shared__ SweepTables tb;
    stage_tables(tb, gmat, nmat, gcls, ncls);
    __syncthreads();
    int svc_mat = -1;
    const double *sv = nullptr, *dual = nullptr;
    if (kind_is_svc(KIND)) {
        stage_svc(tb.smat, nmat, dyn_lds, lds_doubles, svc_mat, sv, dual, KIND);
        __syncthreads();
    }
    int changed = 0, nconv = 0;
    const int nb = gridDim.x;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nel; t += nb) {
        const int e = t * BLOCK + threadIdx.x;
        bool heavy = false;
        if (e < nel) {
            const ClassDev &c = tb.scls[cls[e]];
            const MatDev &m = tb.smat[c.mat];
            if (m.kind == 0) {  // elastic material: skipped by the reference (model.py:1341, 1358)
                if (first_kind) fyn[e] = 0.;
            } else if (m.kind == KIND && !((skip_mask >> c.mat) & 1u)) {  // skip_mask: materials handled by the row / wave kernels
                const size_t ge = (size_t)e + e_off;
                double deps[6], s[6], ep[6], depl[6], Ct[21], dr[6], fy, st_scal;
                class_strain(c, du2, conn[ge * 4], conn[ge * 4 + 1], conn[ge * 4 + 2], conn[ge * 4 + 3], deps);
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    s[k] = sig[(size_t)k * nel + e];
                    ep[k] = epl[(size_t)k * nel + e];
                }
                const bool staged = (c.mat == svc_mat);
                const typename YfOf<KIND>::type yf =
                    make_policy<KIND>(m, staged ? sv : nullptr, staged ? dual : nullptr, (KIND == 7 && kh_el) ? kh_el[e] : m.khard);
                const int st = response_light(m, yf, s, ep, deps, fy, depl, Ct, dr, st_scal);
                if (st == 2)
                    heavy = true;  // (the corrector kernel repeats the prelude from the same entry modulus)
                else {
                    sweep_epilogue(c, m, e, nel, s, ep, depl, Ct, fy, 0, elstiff, Mel, mel_stride, res_sig,
                                   res_depl, fyn, max_steps, nit, changed, nconv, KIND == 7 ? yf.kh() : -1.);
                    if (KIND == 7 && kh_el) {
                        (kh_out ? kh_out : kh_el)[e] = yf.kh();
                        if (kh_touch) kh_touch[e] = yf.touched();
                    }
                }
            }
        }
        // compact the elements that need the 50-sub-step corrector: one atomic per wave
        const unsigned long long mask = __ballot(heavy);
        if (mask) {
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == 0) base = atomicAdd(&flags[2], __popcll(mask));
            base = __shfl(base, 0, 64);
            if (heavy) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = e;
        }
    }
    post_block_flags(changed, nconv, bflags);
}

// Phase 2: the sub-divided plastic corrector for the compacted element list.  Every lane runs the
// same 50 sub-steps (no divergence); FP64-VALU bound.
template <int KIND>
__global__ void __launch_bounds__(BLOCK)
k_sweep_heavy(const MatDev *__restrict__ gmat, int nmat, const ClassDev *__restrict__ gcls, int ncls,
              int lds_doubles, int nel, int e_off, const int32_t *__restrict__ conn,
              const int32_t *__restrict__ cls, const double2 *__restrict__ du2,
              const double *__restrict__ sig, const double *__restrict__ epl, double *elstiff,
              double *Mel, int mel_stride, double *res_sig, double *res_depl, double *fyn,
              int32_t *max_steps, int nit, int *flags, int *bflags, const int32_t *__restrict__ list, unsigned skip_mask,
              double *kh_el = nullptr, double *kh_out = nullptr, int32_t *kh_touch = nullptr)
{
    const int count = flags[2];
    if (count == 0) return;
    __shared__ SweepTables tb;
    stage_tables(tb, gmat, nmat, gcls, ncls);
    __syncthreads();
    int svc_mat = -1;
    const double *sv = nullptr, *dual = nullptr;
    if (kind_is_svc(KIND)) {
        stage_svc(tb.smat, nmat, dyn_lds, lds_doubles, svc_mat, sv, dual, KIND);
        __syncthreads();
    }
    int changed = 0, nconv = 0;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < count; i += gridDim.x * BLOCK) {
        const int e = list[i];
        const ClassDev &c = tb.scls[cls[e]];
        const MatDev &m = tb.smat[c.mat];
        if (m.kind != KIND || ((skip_mask >> c.mat) & 1u)) continue;
        const size_t ge = (size_t)e + e_off;
        double deps[6], s[6], ep[6], depl[6], Ct[21], dr[6], fy, st_scal;
        class_strain(c, du2, conn[ge * 4], conn[ge * 4 + 1], conn[ge * 4 + 2], conn[ge * 4 + 3], deps);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            s[k] = sig[(size_t)k * nel + e];
            ep[k] = epl[(size_t)k * nel + e];
        }
        const bool staged = (c.mat == svc_mat);
        const typename YfOf<KIND>::type yf =
            make_policy<KIND>(m, staged ? sv : nullptr, staged ? dual : nullptr, (KIND == 7 && kh_el) ? kh_el[e] : m.khard);
        response_light(m, yf, s, ep, deps, fy, depl, Ct, dr, st_scal);  // recompute the prelude
        response_heavy(m, yf, s, ep, dr, st_scal, fy, depl, Ct);
        sweep_epilogue(c, m, e, nel, s, ep, depl, Ct, fy, MAXIT - 1, elstiff, Mel, mel_stride, res_sig,
                       res_depl, fyn, max_steps, nit, changed, nconv, KIND == 7 ? yf.kh() : -1.);
        if (KIND == 7 && kh_el) {
            (kh_out ? kh_out : kh_el)[e] = yf.kh();
            if (kh_touch) kh_touch[e] = yf.touched();
        }
    }
    post_block_flags(changed, nconv, bflags);
}

// ---------------------------------------------------------------------------------------------
// Wave-per-element variants of the two sweep phases for the 6-feature SVC material `wave_mat` (YfSvcWave): the
// support-vector sums dominate an SVC update (1585 vectors x tens of yield-function evaluations), so one wave
// works on one element and splits every sum over its lanes.  Lane 0 stores.  The 50-sub-step list is shared with
// the thread-per-element kernels (flags[2]).
__device__ __forceinline__ int stage_svc_wave(const MatDev *smat, int wave_mat, int nc)
{
    const MatDev &m = smat[wave_mat];
    const int n = m.nsv, npad = (n + 64 * nc - 1) / (64 * nc) * (64 * nc);
    for (int i = threadIdx.x; i < npad; i += blockDim.x) {
#pragma unroll
        for (int c = 0; c < 6; c++) dyn_lds[c * npad + i] = (i < n) ? m.sv[6 * (size_t)i + c] : 0.;
        dyn_lds[6 * npad + i] = (i < n) ? m.dual[i] : 0.;
        double vv = 0.;
        if (i < n) {
#pragma unroll
            for (int c = 0; c < 6; c++) vv = fma(m.sv[6 * (size_t)i + c], m.sv[6 * (size_t)i + c], vv);
        }
        dyn_lds[7 * npad + i] = vv;  // |v_k|^2 for the evaluations along a ray (YfSvcT::ray_eval)
        // FP32 pair (dual, -gamma log2(e) |v_k|^2) for the sign screen of the marching bracket (YfSvcT::ray_screen)
        reinterpret_cast<float2 *>(dyn_lds + 8 * npad)[i] =
            make_float2((i < n) ? (float)m.dual[i] : 0.f, (float)(-m.gamma * LOG2E * vv));
    }
    // tables of the sampled-ray form (YfSvcT::ray_sample): samples -> Chebyshev coefficients, and the march factors
    // 0.98^i / 1.02^i as sequential products
    double *ext = dyn_lds + 9 * npad;
    for (int i = threadIdx.x; i < RAYPOLY_N * RAYPOLY_N; i += blockDim.x) ext[i] = RAYPOLY_MT[i];
    if (threadIdx.x < 64) {
        double a = 1., b = 1.;
        for (int i = 0; i < (int)threadIdx.x; i++) {
            a *= 0.98;
            b *= 1.02;
        }
        ext[RAYPOLY_N * RAYPOLY_N + threadIdx.x] = a;
        ext[RAYPOLY_N * RAYPOLY_N + 64 + threadIdx.x] = b;
    }
    return npad;
}
constexpr int SVC_WAVE_EXTRA = RAYPOLY_N * RAYPOLY_N + 128;   // doubles behind the 9 npad of the support-vector tables

// ML_full_yf on N points, one WAVE per point (round 4): the ray search of YfSvcWave<4> -- support-vector sums split over the
// lanes, FP32 sign screen of the marching bracket -- on its own needs 210 VGPRs and no scratch at two waves per SIMD (inside the
// sub-stepping corrector it shares 256 registers + 720 B of scratch with the loop state).  Entry point of plfx_full_yf_batch for the
// model's 6-feature SVC material (f3's callers: find_yloc / calc_properties / yield-locus grids call it on (N, 6) arrays).
template <bool POLY>
__global__ void __launch_bounds__(512)
k_full_yf_wave(const MatDev *__restrict__ gmat, int nmat, int mat, int n, const double *__restrict__ sig_in,
               const double *__restrict__ epl_in, const double *__restrict__ ld, double *__restrict__ out, int32_t *__restrict__ status)
{
    __shared__ MatDev smat[MAXMAT];
    stage_materials(smat, gmat, nmat);
    __syncthreads();
    const int npad = stage_svc_wave(smat, mat, 4);
    __syncthreads();
    const MatDev &m = smat[mat];
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    double ldv[6];
    if (ld) {
#pragma unroll
        for (int c = 0; c < 6; c++) ldv[c] = ld[c];
    }
    for (int i = blockIdx.x * wpb + (threadIdx.x >> 6); i < n; i += gridDim.x * wpb) {  // wave-uniform
        double s[6], e[6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
            s[c] = sig_in[6 * (size_t)i + c];
            e[c] = epl_in ? epl_in[6 * (size_t)i + c] : 0.;
        }
        const YfSvcWave<4, POLY> yf(m, nullptr, nullptr, npad);
        int st = 0;
        const double f = yf.full_ld(s, e, ld ? ldv : nullptr, &st);
        if (lane == 0) {
            out[i] = f;
            if (status) status[i] = st;
        }
    }
}

// Both phases run 512-thread workgroups (8 waves share the LDS tables: 2 waves per SIMD, 256 VGPRs).  HEAVY = 0: 2 vectors per
// lane and trip; HEAVY = 1: 4, with the FP32 sign screen of the marching bracket -- the sub-stepping loop wants 424 registers
// and ran one wave per SIMD in round 1; two waves per SIMD with 592 B of scratch are 1.3x faster (PLFX_HEAVY_THREADS).
template <int HEAVY, bool POLY>
__global__ void __launch_bounds__(HEAVY ? PLFX_HEAVY_THREADS : 512)
k_sweep_svc_wave(const MatDev *__restrict__ gmat, int nmat, const ClassDev *__restrict__ gcls, int ncls,
                 int nel, int e_off, const int32_t *__restrict__ conn, const int32_t *__restrict__ cls,
                 const double2 *__restrict__ du2, const double *__restrict__ sig, const double *__restrict__ epl,
                 double *elstiff, double *Mel, int mel_stride, double *res_sig, double *res_depl, double *fyn,
                 int32_t *max_steps, int nit, int *flags, int *bflags, int32_t *list, int first_kind, int wave_mat)
{
    const int count = HEAVY ? flags[2] : nel;
    if (count == 0) return;
    __shared__ SweepTables tb;
    stage_tables(tb, gmat, nmat, gcls, ncls);
    __syncthreads();
    constexpr int NC = HEAVY ? 4 : 2;
    const int npad = stage_svc_wave(tb.smat, wave_mat, NC);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int w = blockIdx.x * wpb + (threadIdx.x >> 6), nw = gridDim.x * wpb;
    int changed = 0, nconv = 0;
    for (int i = w; i < count; i += nw) {  // wave-uniform
        const int e = HEAVY ? list[i] : i;
        const ClassDev &c = tb.scls[cls[e]];
        const MatDev &m = tb.smat[c.mat];
        if (!HEAVY && m.kind == 0 && first_kind && lane == 0) fyn[e] = 0.;  // elastic: skipped by the reference
        if (c.mat != wave_mat) continue;
        const size_t ge = (size_t)e + e_off;
        double deps[6], s[6], ep[6], depl[6], Ct[21], dr[6], fy, st_scal;
        class_strain(c, du2, conn[ge * 4], conn[ge * 4 + 1], conn[ge * 4 + 2], conn[ge * 4 + 3], deps);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            s[k] = sig[(size_t)k * nel + e];
            ep[k] = epl[(size_t)k * nel + e];
        }
        const YfSvcWave<NC, POLY> yf(m, nullptr, nullptr, npad);
        const int st = response_light(m, yf, s, ep, deps, fy, depl, Ct, dr, st_scal);
        if (HEAVY) {
            response_heavy(m, yf, s, ep, dr, st_scal, fy, depl, Ct);
            if (lane == 0)
                sweep_epilogue(c, m, e, nel, s, ep, depl, Ct, fy, MAXIT - 1, elstiff, Mel, mel_stride, res_sig,
                               res_depl, fyn, max_steps, nit, changed, nconv);
        } else if (st == 2) {
            if (lane == 0) list[atomicAdd(&flags[2], 1)] = e;
        } else if (lane == 0) {
            sweep_epilogue(c, m, e, nel, s, ep, depl, Ct, fy, 0, elstiff, Mel, mel_stride, res_sig, res_depl, fyn,
                           max_steps, nit, changed, nconv);
        }
    }
    post_block_flags(lane == 0 ? changed : 0, lane == 0 ? nconv : 0, bflags);
}

// Row-per-element variants (round 5; YfSvcRow): 16 lanes = one DPP row per element, four elements per wave, the ray search in
// its sampled form.  Same tables in LDS, same two phases, same list and flags as k_sweep_svc_wave; lane 0 of a row stores.
template <bool INLDS>
__global__ void __launch_bounds__(512)
k_full_yf_row(const MatDev *__restrict__ gmat, int nmat, int mat, int n, const double *__restrict__ sig_in,
              const double *__restrict__ epl_in, const double *__restrict__ ld, double *__restrict__ out, int32_t *__restrict__ status)
{
    __shared__ MatDev smat[MAXMAT];
    stage_materials(smat, gmat, nmat);
    __syncthreads();
    const int npad = INLDS ? stage_svc_wave(smat, mat, 1) : smat[mat].rowpad;   // rows take 16 x 4 vectors per trip: padded to 64
    __syncthreads();
    const MatDev &m = smat[mat];
    const int l16 = threadIdx.x & 15, rpb = blockDim.x >> 4;
    double ldv[6];
    if (ld) {
#pragma unroll
        for (int c = 0; c < 6; c++) ldv[c] = ld[c];
    }
    for (int i = blockIdx.x * rpb + (threadIdx.x >> 4); i < n; i += gridDim.x * rpb) {  // row-uniform
        double s[6], e[6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
            s[c] = sig_in[6 * (size_t)i + c];
            e[c] = epl_in ? epl_in[6 * (size_t)i + c] : 0.;
        }
        const YfSvcRow<4, INLDS> yf(m, npad);
        int st = 0;
        const double f = yf.full_ld(s, e, ld ? ldv : nullptr, &st);
        if (l16 == 0) {
            out[i] = f;
            if (status) status[i] = st;
        }
    }
}

// Material.response on n points of the row-kernel SVC material `mat` (host-layout arrays as in k_response_batch; points of
// other materials are left to k_response_batch<3>): the same code path as the sweeps of a model, callable point by point
template <bool INLDS>
__global__ void __launch_bounds__(512)
k_response_row(const MatDev *__restrict__ gmat, int nmat, int mat, int n, const int32_t *__restrict__ mat_id,
               const double *__restrict__ sig_in, const double *__restrict__ epl_in, const double *__restrict__ deps_in,
               double *__restrict__ fy, double *__restrict__ sig_out, double *__restrict__ depl_out, double *__restrict__ ct_out,
               int32_t *__restrict__ nsteps, int maxit = MAXIT)
{
    __shared__ MatDev smat[MAXMAT];
    stage_materials(smat, gmat, nmat);
    __syncthreads();
    const int npad = INLDS ? stage_svc_wave(smat, mat, 1) : smat[mat].rowpad;
    __syncthreads();
    const MatDev &m = smat[mat];
    const int l16 = threadIdx.x & 15, rpb = blockDim.x >> 4;
    for (int i = blockIdx.x * rpb + (threadIdx.x >> 4); i < n; i += gridDim.x * rpb) {  // row-uniform
        if ((mat_id ? mat_id[i] : 0) != mat) continue;
        double sig[6], epl[6], deps[6], depl[6], Ct[21], f = 0.;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            sig[c] = sig_in[6 * (size_t)i + c];
            epl[c] = epl_in[6 * (size_t)i + c];
            deps[c] = deps_in[6 * (size_t)i + c];
        }
        const YfSvcRow<4, INLDS> yf(m, npad);
        const int ns = response_point(m, yf, sig, epl, deps, f, depl, Ct, maxit);
        if (l16 == 0) {
            fy[i] = f;
            nsteps[i] = ns;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                sig_out[6 * (size_t)i + c] = sig[c];
                depl_out[6 * (size_t)i + c] = depl[c];
            }
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c < 6; c++) ct_out[36 * (size_t)i + r * 6 + c] = Ct[sym_idx(r, c)];
        }
    }
}

template <int HEAVY, bool INLDS>
__global__ void __launch_bounds__(512)
k_sweep_svc_row(const MatDev *__restrict__ gmat, int nmat, const ClassDev *__restrict__ gcls, int ncls,
                int nel, int e_off, const int32_t *__restrict__ conn, const int32_t *__restrict__ cls,
                const double2 *__restrict__ du2, const double *__restrict__ sig, const double *__restrict__ epl,
                double *elstiff, double *Mel, int mel_stride, double *res_sig, double *res_depl, double *fyn,
                int32_t *max_steps, int nit, int *flags, int *bflags, int32_t *list, int first_kind, int wave_mat)
{
    const int count = HEAVY ? flags[2] : nel;
    if (count == 0) return;
    __shared__ SweepTables tb;
    stage_tables(tb, gmat, nmat, gcls, ncls);
    __syncthreads();
#ifndef PLFX_ROW_NC
#define PLFX_ROW_NC 4
#endif
    constexpr int NC = PLFX_ROW_NC;   // (2: measured 0 % -- see DESIGN 11.5)
    const int npad = INLDS ? stage_svc_wave(tb.smat, wave_mat, 1) : tb.smat[wave_mat].rowpad;   // rows take 16 x 4 vectors per trip: padded to 64
    __syncthreads();
    const int l16 = threadIdx.x & 15;
    const int rpb = blockDim.x >> 4;
    const int w = blockIdx.x * rpb + (threadIdx.x >> 4), nw = gridDim.x * rpb;
    int changed = 0, nconv = 0;
    for (int i = w; i < count; i += nw) {  // row-uniform
        const int e = HEAVY ? list[i] : i;
        const ClassDev &c = tb.scls[cls[e]];
        const MatDev &m = tb.smat[c.mat];
        if (!HEAVY && m.kind == 0 && first_kind && l16 == 0) fyn[e] = 0.;  // elastic: skipped by the reference
        if (c.mat != wave_mat) continue;
        const size_t ge = (size_t)e + e_off;
        double deps[6], s[6], ep[6], depl[6], Ct[21], dr[6], fy, st_scal;
        class_strain(c, du2, conn[ge * 4], conn[ge * 4 + 1], conn[ge * 4 + 2], conn[ge * 4 + 3], deps);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            s[k] = sig[(size_t)k * nel + e];
            ep[k] = epl[(size_t)k * nel + e];
        }
        const YfSvcRow<NC, INLDS> yf(m, npad);
        const int st = response_light(m, yf, s, ep, deps, fy, depl, Ct, dr, st_scal);
        if (HEAVY) {
            response_heavy(m, yf, s, ep, dr, st_scal, fy, depl, Ct);
            if (l16 == 0)
                sweep_epilogue(c, m, e, nel, s, ep, depl, Ct, fy, MAXIT - 1, elstiff, Mel, mel_stride, res_sig,
                               res_depl, fyn, max_steps, nit, changed, nconv);
        } else if (st == 2) {
            if (l16 == 0) list[atomicAdd(&flags[2], 1)] = e;
        } else if (l16 == 0) {
            sweep_epilogue(c, m, e, nel, s, ep, depl, Ct, fy, 0, elstiff, Mel, mel_stride, res_sig, res_depl, fyn,
                           max_steps, nit, changed, nconv);
        }
    }
    post_block_flags(l16 == 0 ? changed : 0, l16 == 0 ? nconv : 0, bflags);
}

// Wave-per-element sweep of the work-hardening SVC materials (kind 7; round 4): the same two phases with YfSvcWhT<1> -- a
// 15-feature SVC update costs ~1e8 flop in its support-vector sums, and one THREAD per element (k_sweep_light<7> / _heavy<7>)
// leaves a 16- or 144-element model on 16 or 144 lanes of the GPU (0.75 s per sweep on 4 x 4 elements).  Tables from LDS when
// they fit (stage_svc), else from the L2.  kh_el / kh_out / kh_touch as in the thread kernels (lane 0 stores).
template <int HEAVY>
__global__ void __launch_bounds__(BLOCK)
k_sweep_wh_wave(const MatDev *__restrict__ gmat, int nmat, const ClassDev *__restrict__ gcls, int ncls, int lds_doubles,
                int nel, int e_off, const int32_t *__restrict__ conn, const int32_t *__restrict__ cls,
                const double2 *__restrict__ du2, const double *__restrict__ sig, const double *__restrict__ epl,
                double *elstiff, double *Mel, int mel_stride, double *res_sig, double *res_depl, double *fyn,
                int32_t *max_steps, int nit, int *flags, int *bflags, int32_t *list, int first_kind,
                double *kh_el, double *kh_out, int32_t *kh_touch)
{
    const int count = HEAVY ? flags[2] : nel;
    if (count == 0) return;
    __shared__ SweepTables tb;
    stage_tables(tb, gmat, nmat, gcls, ncls);
    __syncthreads();
    int svc_mat = -1;
    const double *sv = nullptr, *dual = nullptr;
    stage_svc(tb.smat, nmat, dyn_lds, lds_doubles, svc_mat, sv, dual, 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int w = blockIdx.x * wpb + (threadIdx.x >> 6), nw = gridDim.x * wpb;
    int changed = 0, nconv = 0;
    for (int i = w; i < count; i += nw) {  // wave-uniform
        const int e = HEAVY ? list[i] : i;
        const ClassDev &c = tb.scls[cls[e]];
        const MatDev &m = tb.smat[c.mat];
        if (!HEAVY && m.kind == 0 && first_kind && lane == 0) fyn[e] = 0.;  // elastic: skipped by the reference
        if (m.kind != 7) continue;
        const size_t ge = (size_t)e + e_off;
        double deps[6], s[6], ep[6], depl[6], Ct[21], dr[6], fy, st_scal;
        class_strain(c, du2, conn[ge * 4], conn[ge * 4 + 1], conn[ge * 4 + 2], conn[ge * 4 + 3], deps);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            s[k] = sig[(size_t)k * nel + e];
            ep[k] = epl[(size_t)k * nel + e];
        }
        const bool staged = (c.mat == svc_mat);
        const YfSvcWhT<1> yf(m, staged ? sv : m.sv, staged ? dual : m.dual, kh_el ? kh_el[e] : m.khard);
        const int st = response_light(m, yf, s, ep, deps, fy, depl, Ct, dr, st_scal);
        if (HEAVY) response_heavy(m, yf, s, ep, dr, st_scal, fy, depl, Ct);
        if (!HEAVY && st == 2) {
            if (lane == 0) list[atomicAdd(&flags[2], 1)] = e;
            continue;
        }
        if (lane == 0) {
            sweep_epilogue(c, m, e, nel, s, ep, depl, Ct, fy, HEAVY ? MAXIT - 1 : 0, elstiff, Mel, mel_stride, res_sig, res_depl, fyn,
                           max_steps, nit, changed, nconv, yf.kh());
            if (kh_el) {
                (kh_out ? kh_out : kh_el)[e] = yf.kh();
                if (kh_touch) kh_touch[e] = yf.touched();
            }
        }
    }
    post_block_flags(lane == 0 ? changed : 0, lane == 0 ? nconv : 0, bflags);
}

// elstiff = CV, M from CV for all owned elements (model.py:1219-1221)
__global__ void __launch_bounds__(BLOCK)
k_init_tangent(const MatDev *gmat, const ClassDev *gcls, int nel, const int32_t *cls,
               double *elstiff, double *Mel, int mel_stride)
{
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= nel) return;
    const ClassDev &c = gcls[cls[e]];
    const MatDev &m = gmat[c.mat];
    double D[21];
#pragma unroll
    for (int k = 0; k < 21; k++) {
        D[k] = m.CV[k];
        elstiff[(size_t)k * nel + e] = D[k];
    }
    double M[6];
    tangent_to_M(D, c.kappa, M);
#pragma unroll
    for (int k = 0; k < 6; k++) Mel[(size_t)k * mel_stride + e] = M[k];
}

// SPD surrogate of the operator (DESIGN "Indefinite tangents"): the correction step of Material.response (material.py:317-338)
// can leave a tangent that is not positive semi-definite; the element stiffness matrix Kel = Jac sum_gp B^T D B is linear in
// the 3 x 3 generator matrix G = [XX XY XS; XY YY YS; XS YS SS] and PSD iff G is.  Msur = M (pair layout) with every
// indefinite G shifted by its most negative eigenvalue, G + |lambda_min| I (closed-form eigenvalues of the symmetric 3 x 3
// matrix): K_sur is positive semi-definite by construction, keeps the soft directions of the elastic-plastic tangents (the
// V-cycle built on it is as good a preconditioner as the one of K itself) and differs from K by sum_e |lambda_min,e| Kel(I) --
// a handful of elements early in config 5 (2 of 4.2 M), ~1e5 mildly indefinite ones late in its schedule (measured: replacing
// those by their ELASTIC matrices instead costs the preconditioner its quality: MINRES 600+, GMRES thousands of iterations).
// MINRES on the true, symmetric indefinite K needs exactly that: an SPD preconditioner.  nbad[block] = shifted elements.
__global__ void __launch_bounds__(BLOCK)
k_make_surrogate(const MatDev *__restrict__ gmat, const ClassDev *__restrict__ gcls, int nel, const int32_t *__restrict__ cls,
                 const double *__restrict__ Mop, double *__restrict__ Msur, int *__restrict__ nbad)
{
    (void)gmat; (void)gcls; (void)cls;
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const double2 *S = reinterpret_cast<const double2 *>(Mop);
    double2 *T = reinterpret_cast<double2 *>(Msur);
    int mine = 0;
    for (size_t e = blockIdx.x * (size_t)BLOCK + threadIdx.x; e < (size_t)nel; e += (size_t)gridDim.x * BLOCK) {
        double2 a01 = S[e], a23 = S[(size_t)nel + e], a45 = S[(size_t)2 * nel + e];
        const double a = a01.x, b = a01.y, cc = a23.x, d = a23.y, f = a45.x, g = a45.y;   // XX XY XS YY YS SS
        // smallest eigenvalue of [a b cc; b d f; cc f g] (trigonometric form)
        const double q = (a + d + g) / 3.;
        const double p1 = b * b + cc * cc + f * f;
        const double p2 = (a - q) * (a - q) + (d - q) * (d - q) + (g - q) * (g - q) + 2. * p1;
        const double sc1 = fabs(a) + fabs(d) + fabs(g);
        double lmin = fmin(a, fmin(d, g));
        if (p2 > 1e-30 * sc1 * sc1) {
            const double p = sqrt(p2 / 6.);
            const double ba = (a - q) / p, bd = (d - q) / p, bg = (g - q) / p, bb = b / p, bc = cc / p, bf = f / p;
            double r = 0.5 * (ba * (bd * bg - bf * bf) - bb * (bb * bg - bf * bc) + bc * (bb * bf - bd * bc));
            r = fmin(1., fmax(-1., r));
            const double phi = acos(r) / 3.;
            lmin = q + 2. * p * cos(phi + 2.0943951023931953);   // + 2 pi / 3: the smallest of the three
        }
        if (!(lmin >= -1e-10 * sc1)) {   // really indefinite (round-off of a singular PSD tangent stays far above this)
            const double sh = (lmin == lmin) ? -lmin * (1. + 1e-6) + 1e-14 * sc1 : sc1;   // NaN entries: a plain positive shift
            a01.x = a + sh;
            a23.y = d + sh;
            a45.y = g + sh;
            mine++;
        }
        T[e] = a01;
        T[(size_t)nel + e] = a23;
        T[(size_t)2 * nel + e] = a45;
    }
    if (mine) atomicAdd(&cnt, mine);  // integer count: order does not matter
    __syncthreads();
    if (threadIdx.x == 0) nbad[blockIdx.x] = cnt;
}

// Jacobi scaling of another diagonal with the Dirichlet mask of `mask_dinv` (zero = prescribed DOF)
__global__ void __launch_bounds__(BLOCK)
k_dinv_masked(size_t ndof, const double *__restrict__ diag, const double *__restrict__ mask_dinv, double *__restrict__ dinv)
{
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < ndof; i += (size_t)gridDim.x * BLOCK) {
        const double d = fabs(diag[i]);
        dinv[i] = (mask_dinv[i] != 0.) ? (d > 1e-300 ? 1. / d : 1.) : 0.;
    }
}

// M from CV for ALL elements of the mesh (sharded runs keep the stiffness generators of the whole
// mesh on every rank: the matrix hierarchy is replicated, only the sweep is sharded)
__global__ void __launch_bounds__(BLOCK)
k_init_M_all(const MatDev *gmat, const ClassDev *gcls, int nel, const int32_t *cls, double *Mel)
{
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= nel) return;
    const ClassDev &c = gcls[cls[e]];
    double M[6];
    tangent_to_M(gmat[c.mat].CV, c.kappa, M);
#pragma unroll
    for (int k = 0; k < 6; k++) Mel[(size_t)k * nel + e] = M[k];
}

// zero the stiffness generators of the elements this rank does not own (before the all-reduce that
// makes M consistent on every rank: x + 0 + ... + 0 = x exactly)
__global__ void __launch_bounds__(BLOCK)
k_zero_foreign_M(int nel_total, int e0, int e1, double *Mel)
{
    const size_t n = (size_t)6 * nel_total;
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const int e = (int)(i % nel_total);
        if (e < e0 || e >= e1) Mel[i] = 0.;
    }
}

// recompute M from elstiff (after plfx_state_set of the tangent)
__global__ void __launch_bounds__(BLOCK)
k_refresh_M(const ClassDev *gcls, int nel, const int32_t *cls, const double *elstiff, double *Mel,
            int mel_stride)
{
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    if (e >= nel) return;
    double D[21], M[6];
#pragma unroll
    for (int k = 0; k < 21; k++) D[k] = elstiff[(size_t)k * nel + e];
    tangent_to_M(D, gcls[cls[e]].kappa, M);
#pragma unroll
    for (int k = 0; k < 6; k++) Mel[(size_t)k * mel_stride + e] = M[k];
}

// ---------------------------------------------------------------------------------------------
// Layouts of the six stiffness generators (XX XY XS YY YS SS) of nel elements:
//   live array (written by the material sweep, exchanged between strips): SoA [6][stride]
//   operator side of the matrix-free path (the snapshot Mop, the generators of the multigrid levels >= 1, the replicated
//   coarse problem of a strip): PAIR layout [3][nel] double2 = (XX,XY) (XS,YY) (YS,SS) -- a node reads the generators of its
//   four elements with 12 sixteen-byte loads instead of 24 eight-byte ones; the operator kernels are bound by the loads a
//   wave keeps in flight, not by bytes (tools/probes/pair_probe.hip: fine-level smoother 27.6 -> 23.2 us).
__host__ __device__ __forceinline__ size_t gen_index(bool pair, int c, size_t nel, size_t e)
{
    return pair ? ((size_t)(c >> 1) * nel + e) * 2 + (c & 1) : (size_t)c * nel + e;
}

// ---------------------------------------------------------------------------------------------
// Block-ELL pattern of the reference's STRUCTURED grid in closed form (what build_pattern() derives from the connectivity
// by sorting: same slots, same gather codes, same order).  Node i = j * (ny + 1) + k, element e = ej * ny + ek with the
// local nodes 0:(ej,ek) 1:(ej,ek+1) 2:(ej+1,ek) 3:(ej+1,ek+1) (model.py:893, 935-948).  Slot s of node i = the s-th of its
// neighbour nodes (itself included) in ascending node order; its <= 4 gather codes e*16 + a*4 + b (a / b: local numbers of
// node i / the neighbour in element e) in ascending element order -- the reference's addition order.  Needs nx, ny >= 2
// (then nslot = 9, nq = 4).
__host__ __device__ inline void structured_slot(int nx, int ny, int i, int s, int32_t *col, int32_t *codes /* [4] */)
{
    const int nyn = ny + 1;
    const int j = i / nyn, k = i - j * nyn;
    const int jlo = j > 0 ? j - 1 : 0, jhi = j < nx ? j + 1 : nx;
    const int klo = k > 0 ? k - 1 : 0, khi = k < ny ? k + 1 : ny;
    const int nkk = khi - klo + 1, cnt = (jhi - jlo + 1) * nkk;
    codes[0] = codes[1] = codes[2] = codes[3] = -1;
    if (s >= cnt) {
        *col = -1;
        return;
    }
    const int jj = jlo + s / nkk, kk = klo + s % nkk;
    *col = jj * nyn + kk;
    int qn = 0;
    for (int ej = j - 1; ej <= j; ej++) {
        if (ej < 0 || ej >= nx || jj < ej || jj > ej + 1) continue;
        for (int ek = k - 1; ek <= k; ek++) {
            if (ek < 0 || ek >= ny || kk < ek || kk > ek + 1) continue;
            const int e = ej * ny + ek, a = (j - ej) * 2 + (k - ek), b = (jj - ej) * 2 + (kk - ek);
            codes[qn++] = e * 16 + a * 4 + b;
        }
    }
}

// col[s * nnode + i], contrib[(s * 4 + q) * nnode + i] of a structured nx x ny grid, filled on the device (no host pattern,
// no upload: 180 B per node)
__global__ void __launch_bounds__(256)
k_structured_pattern(int nx, int ny, int32_t *__restrict__ col, int32_t *__restrict__ contrib)
{
    const int nnode = (nx + 1) * (ny + 1);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nnode; i += gridDim.x * 256) {
#pragma unroll
        for (int s = 0; s < 9; s++) {
            int32_t cj, codes[4];
            structured_slot(nx, ny, i, s, &cj, codes);
            col[(size_t)s * nnode + i] = cj;
#pragma unroll
            for (int q = 0; q < 4; q++) contrib[((size_t)s * 4 + q) * nnode + i] = codes[q];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Assembly (model.py:954-977) as a gather: thread (node i, slot s) sums the <= nq element
// contributions of block K[i, col(s)] in ascending element order.  contrib code = e*16 + a*4 + b
// (e local owned-element index, a/b local node numbers), -1 = none.
// val layout: [(s*4 + r*2 + c) * nnode + i]
__global__ void __launch_bounds__(BLOCK)
k_assemble(const ClassDev *gcls, int ncls, int nnode, int nslot, int nq, int nel,
           const int32_t *contrib, const int32_t *cls, const double *Mel, const int32_t *col,
           double *val, double *diag, int pair /* layout of Mel: 0 SoA, 1 pairs */)
{
    __shared__ ClassDev scls[MAXCLS];
    {
        const int words = ncls * (int)(sizeof(ClassDev) / 8);
        const double *src = reinterpret_cast<const double *>(gcls);
        double *dst = reinterpret_cast<double *>(scls);
        for (int i = threadIdx.x; i < words; i += BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    const int s = blockIdx.y;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nnode; i += gridDim.x * BLOCK) {
        double k00 = 0., k01 = 0., k10 = 0., k11 = 0.;
        for (int q = 0; q < nq; q++) {
            const int code = contrib[((size_t)s * nq + q) * nnode + i];
            if (code < 0) continue;
            const int e = code >> 4, a = (code >> 2) & 3, b = code & 3;
            const ClassDev &c = scls[cls[e]];
            const double Mxx = Mel[gen_index(pair, 0, nel, e)], Mxy = Mel[gen_index(pair, 1, nel, e)],
                         Mxs = Mel[gen_index(pair, 2, nel, e)], Myy = Mel[gen_index(pair, 3, nel, e)],
                         Mys = Mel[gen_index(pair, 4, nel, e)], Mss = Mel[gen_index(pair, 5, nel, e)];
            const double sxx = c.Sxx[a * 4 + b], sxy = c.Sxy[a * 4 + b], syx = c.Sxy[b * 4 + a],
                         syy = c.Syy[a * 4 + b];
            k00 += Mxx * sxx + Mxs * (sxy + syx) + Mss * syy;
            k01 += Mxy * sxy + Mxs * sxx + Mys * syy + Mss * syx;
            k10 += Mxy * syx + Mys * syy + Mxs * sxx + Mss * sxy;
            k11 += Myy * syy + Mys * (syx + sxy) + Mss * sxx;
        }
        val[((size_t)s * 4 + 0) * nnode + i] = k00;
        val[((size_t)s * 4 + 1) * nnode + i] = k01;
        val[((size_t)s * 4 + 2) * nnode + i] = k10;
        val[((size_t)s * 4 + 3) * nnode + i] = k11;
        if (col[(size_t)s * nnode + i] == i) {
            diag[2 * (size_t)i] = k00;
            diag[2 * (size_t)i + 1] = k11;
        }
    }
}

// Row pair i of the block-ELL matrix times a vector given as a functor xf(j) -> double2.
// Nine slots (every structured Q4 grid): branch-free and fully unrolled, so that the 9 column ids, the 36
// matrix values and the 9 gathers are all in flight together (empty slots hold exact zeros, their gather
// is redirected to the node itself).  Other slot counts take the generic loop.
template <class XF>
__device__ __forceinline__ double2 bell_apply(int nnode, int nslot, const int32_t *__restrict__ col,
                                              const double *__restrict__ val, int i, XF xf)
{
    double qx = 0., qy = 0.;
    if (nslot == 9) {
        int j[9];
        double v[36];
#pragma unroll
        for (int s = 0; s < 9; s++) j[s] = col[(size_t)s * nnode + i];
#pragma unroll
        for (int k = 0; k < 36; k++) v[k] = val[(size_t)k * nnode + i];
#pragma unroll
        for (int s = 0; s < 9; s++) {
            const double2 pj = xf(j[s] < 0 ? i : j[s]);
            qx = fma(v[4 * s + 0], pj.x, fma(v[4 * s + 1], pj.y, qx));
            qy = fma(v[4 * s + 2], pj.x, fma(v[4 * s + 3], pj.y, qy));
        }
    } else {
        for (int s = 0; s < nslot; s++) {
            const int j = col[(size_t)s * nnode + i];
            if (j < 0) continue;
            const double2 pj = xf(j);
            qx = fma(val[((size_t)s * 4 + 0) * nnode + i], pj.x, fma(val[((size_t)s * 4 + 1) * nnode + i], pj.y, qx));
            qy = fma(val[((size_t)s * 4 + 2) * nnode + i], pj.x, fma(val[((size_t)s * 4 + 3) * nnode + i], pj.y, qy));
        }
    }
    return make_double2(qx, qy);
}

// ---------------------------------------------------------------------------------------------
// Operator descriptor.  GRID = 0: assembled block-ELL matrix (any mesh).  GRID = 1: matrix-free on the structured
// node grid (node id = j*nyn + k, element id = j*(nyn-1) + k, model.py:893/935): the row pair of node i is applied
// straight from the stiffness generators M of its <= 4 elements (48 B/element instead of 288 + 36 B/node of matrix)
// and one geometry table shared by all elements (plfx_set_grid requires a uniform element size).
struct KOp {
    int nnode, nslot;
    const int32_t *col;
    const double *val;
    int nxn, nyn, nel;   // nodes per row / column, elements
    const double *M;     // generators XX XY XS YY YS SS: pair layout [3][nel] double2 (gen_index), k_grid_setup<0> reads SoA [6][nel]
    const double *tab;   // [4 positions][4 b][sxx_ab, syy_ab, sxy_ab, sxy_ba]; position p = pj*2+pk <-> element (j-1+pj, k-1+pk)
    // size of the LAST element column / row relative to the others (1 on every grid whose size halves exactly; coarse levels
    // of a mesh with an odd number of elements end in a narrower cell so that the level covers exactly the fine grid, DESIGN
    // 10.7).  The stiffness integrals of a cell of size (rx lx, ry ly): Sxx ~ ly / lx scales with ry / rx, Syy with rx / ry,
    // Sxy does not depend on the size.
    double rx = 1., ry = 1.;
    // width of every element COLUMN relative to the column the tables were made for (non-proportional laminates: dx = LS[i] /
    // nes[i] per section, model.py:826-847), or null; only the Krylov operator of such a mesh carries it (instantiation 2 of the
    // kernels) -- its V-cycle runs on the operator of the uniform grid, DESIGN 10.8
    const double *colr = nullptr;
};

// Coarsening of one direction of a level with n cells, the last of relative size r (all others 1):
//   n even    pairs; the last coarse cell has children (1, r)                      -> n / 2 cells,       r' = (1 + r) / 2
//   n odd     pairs, the last coarse cell takes THREE children (1, 1, r)           -> (n - 1) / 2 cells, r' = (2 + r) / 2
// so that r stays in [1, 2) on every level: the odd cell is never NARROWER than the others.  (Until the end of round 5 an odd
// level with r >= 1 kept its last cell alone, r' = r / 2 in [1/2, 3/4): fewer iterations on random fields, but a cell narrower
// than its neighbours makes the area-scaled smoothing diagonal of k_grid_setup SMALLER than the true one -- not a stable
// Jacobi scaling -- and that diagonal is what keeps the homogeneous workload's error fields in their invariant subspace, see
// there.  PLFX_MG_ODD_RULE=0 at compile time restores the old rule.)  The last node line of every level is the edge of the grid.
#ifndef PLFX_MG_ODD_RULE
#define PLFX_MG_ODD_RULE 1
#endif
__host__ __device__ __forceinline__ bool mg_odd_triple(int n, double r) { return (n & 1) && (PLFX_MG_ODD_RULE || r < 1.); }
__host__ __device__ __forceinline__ int mg_coarse_cells(int n, double r) { return (n & 1) ? (mg_odd_triple(n, r) ? (n - 1) >> 1 : (n + 1) >> 1) : n >> 1; }
__host__ __device__ __forceinline__ double mg_coarse_ratio(int n, double r)
{
    return (n & 1) ? (mg_odd_triple(n, r) ? 0.5 * (2. + r) : 0.5 * r) : 0.5 * (1. + r);
}
// WHICH cell of a direction is the one of the other size: the LAST one (default), or -- PLFX_MG_RAGGED_FIRST=1 at compile
// time -- the first, by mirroring every index (node j <-> n - j, cell e <-> n - 1 - e).  Built to test whether the odd cell
// hurts because it lies next to the loaded edge (the reference's solve loop increments the right / top displacements only,
// model.py:1296-1312): it does not -- 128 x 127 under tension in y costs 227 PCG iterations with the odd row at the top and 226
// with it at the bottom (128 x 128: 28); what matters is the smoothing diagonal, see k_grid_setup.  Kept as a knob.
#ifndef PLFX_MG_RAGGED_FIRST
#define PLFX_MG_RAGGED_FIRST 0
#endif
__host__ __device__ __forceinline__ int mg_odd_cell(int n) { return PLFX_MG_RAGGED_FIRST ? 0 : n - 1; }
__host__ __device__ __forceinline__ int mg_fine_node_last(int J, int n, double r) { return J == mg_coarse_cells(n, r) ? n : 2 * J; }
// fine node of this level that coarse node J coincides with
__host__ __device__ __forceinline__ int mg_fine_node(int J, int n, double r)
{
    if (!PLFX_MG_RAGGED_FIRST) return mg_fine_node_last(J, n, r);
    return n - mg_fine_node_last(mg_coarse_cells(n, r) - J, n, r);
}
__host__ __device__ __forceinline__ void mg_tr1d_last(int j, int n, double r, int &J0, double &w0, double &w1);
// 1-d transfer stencil (bilinear interpolation): fine node j takes w0 of coarse node J0 and w1 of J0 + 1
__host__ __device__ __forceinline__ void mg_tr1d(int j, int n, double r, int &J0, double &w0, double &w1)
{
    if (!PLFX_MG_RAGGED_FIRST) {
        mg_tr1d_last(j, n, r, J0, w0, w1);
        return;
    }
    int Jm;
    double a0, a1;
    mg_tr1d_last(n - j, n, r, Jm, a0, a1);   // mirrored: coarse nodes nc - Jm (a0) and nc - Jm - 1 (a1)
    const int nc = mg_coarse_cells(n, r);
    if (a1 == 0.) {
        J0 = nc - Jm;
        w0 = 1.;
        w1 = 0.;
    } else {
        J0 = nc - Jm - 1;
        w0 = a1;
        w1 = a0;
    }
}
__host__ __device__ __forceinline__ void mg_tr1d_last(int j, int n, double r, int &J0, double &w0, double &w1)
{
    const bool odd = n & 1, triple = mg_odd_triple(n, r);
    if (j == n) {                       // the edge: last coarse node
        J0 = mg_coarse_cells(n, r);
        w0 = 1.;
        w1 = 0.;
    } else if (triple && j >= n - 2) {  // the two interior nodes of the last coarse cell (children 1, 1, r), left node (n - 3) / 2
        J0 = (n - 3) >> 1;
        w1 = (j == n - 2 ? 1. : 2.) / (2. + r);
        w0 = 1. - w1;
    } else if (!(j & 1)) {              // coincident
        J0 = j >> 1;
        w0 = 1.;
        w1 = 0.;
    } else {                            // between the two children of coarse cell (j - 1) / 2
        J0 = j >> 1;
        w1 = (!odd && j == n - 1) ? 1. / (1. + r) : 0.5;
        w0 = 1. - w1;
    }
}

// Row pair i of K times a vector given as a functor xf(node) -> double2, from the element generators.
// With a = local number of node i in the element and u_b the vector at the element's node b:
//   q_x += Mxx A1 + Mxs (A5+A7+A2) + Mss (A3+A8) + Mxy A6 + Mys A4
//   q_y += Mxy A7 + Mys (A3+A8+A6) + Mxs A1 + Mss (A5+A2) + Myy A4
// A1..A8 = sum_b {sxx,sxx,syy,syy,sxy,sxy,syx,syx}_ab * {ux,uy,...}_b   (the 2x2 blocks of k_assemble, regrouped)
// Elements outside the grid enter with M = 0 (indices clamped, so every load is in range).
// mf(q): generator number q = c * nel + e (global memory for the big levels, LDS in the single-workgroup tail)
template <class MF, class XF>
__device__ __forceinline__ double2 grid_apply_g(int nxn, int nyn, int nel, const double *tab, int i, MF mf, XF xf)
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
            for (int c = 0; c < 6; c++) {
                const double v = mf(c * nel + e);
                m[pj * 2 + pk][c] = ok ? v : 0.;
            }
        }
    double qx = 0., qy = 0.;
#pragma unroll
    for (int pj = 0; pj < 2; pj++)
#pragma unroll
        for (int pk = 0; pk < 2; pk++) {
            const int p = pj * 2 + pk;
            const double *T = tab + p * 16;  // wave-uniform -> scalar loads
            double A1 = 0., A2 = 0., A3 = 0., A4 = 0., A5 = 0., A6 = 0., A7 = 0., A8 = 0.;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const double2 ub = u[pj + (b >> 1)][pk + (b & 1)];
                const double sxx = T[b * 4 + 0], syy = T[b * 4 + 1], sxy = T[b * 4 + 2], syx = T[b * 4 + 3];
                A1 = fma(sxx, ub.x, A1);
                A2 = fma(sxx, ub.y, A2);
                A3 = fma(syy, ub.x, A3);
                A4 = fma(syy, ub.y, A4);
                A5 = fma(sxy, ub.x, A5);
                A6 = fma(sxy, ub.y, A6);
                A7 = fma(syx, ub.x, A7);
                A8 = fma(syx, ub.y, A8);
            }
            const double Mxx = m[p][0], Mxy = m[p][1], Mxs = m[p][2], Myy = m[p][3], Mys = m[p][4], Mss = m[p][5];
            qx = fma(Mxx, A1, fma(Mxs, A5 + A7 + A2, fma(Mss, A3 + A8, fma(Mxy, A6, fma(Mys, A4, qx)))));
            qy = fma(Mxy, A7, fma(Mys, A3 + A8 + A6, fma(Mxs, A1, fma(Mss, A5 + A2, fma(Myy, A4, qy)))));
        }
    return make_double2(qx, qy);
}

// the same with the generators in pair layout: mf2(q) = pair number q = c2 * nel + e; the vector comes as xjk(jj, kk) = entry
// of node (column jj, row kk) (callers that interpolate the entry on the fly need the grid position, not the node number)
// RAGGED: the level's last element column / row has another size than the rest (KOp::rx, ry) -- a compile-time variant, so
// that the code of levels whose cells are all alike is what it was (the extra registers of a run-time test cost the fine-level
// kernels 12-27 % and doubled the single-workgroup tail: measured, profiles/r05m)
template <bool RAGGED = false, class MF2, class XJK>
__device__ __forceinline__ double2 grid_apply_pairs_jk(int nxn, int nyn, int nel, const double *tab, int j, int k, MF2 mf2, XJK xjk,
                                                       double rx = 1., double ry = 1., const double *colr = nullptr)
{
    const int nye = nyn - 1, nxe = nxn - 1;
    double2 u[3][3];
#pragma unroll
    for (int dj = 0; dj < 3; dj++) {
        const int jj = min(max(j + dj - 1, 0), nxe);
#pragma unroll
        for (int dk = 0; dk < 3; dk++) {
            const int kk = min(max(k + dk - 1, 0), nye);
            u[dj][dk] = xjk(jj, kk);
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
            const double *T = tab + p * 16;  // wave-uniform -> scalar loads
            double A1 = 0., A2 = 0., A3 = 0., A4 = 0., A5 = 0., A6 = 0., A7 = 0., A8 = 0.;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const double2 ub = u[pj + (b >> 1)][pk + (b & 1)];
                const double sxx = T[b * 4 + 0], syy = T[b * 4 + 1], sxy = T[b * 4 + 2], syx = T[b * 4 + 3];
                A1 = fma(sxx, ub.x, A1);
                A2 = fma(sxx, ub.y, A2);
                A3 = fma(syy, ub.x, A3);
                A4 = fma(syy, ub.y, A4);
                A5 = fma(sxy, ub.x, A5);
                A6 = fma(sxy, ub.y, A6);
                A7 = fma(syx, ub.x, A7);
                A8 = fma(syx, ub.y, A8);
            }
            if (RAGGED) {   // a cell of the last column / row: Sxx ~ ly / lx, Syy ~ lx / ly
                const double sx = colr ? colr[min(max(j - 1 + pj, 0), nxe - 1)] : ((j - 1 + pj == mg_odd_cell(nxe)) ? rx : 1.);
                const double sy = (k - 1 + pk == mg_odd_cell(nye)) ? ry : 1.;
                const double f = sy / sx, fi = sx / sy;
                A1 *= f;
                A2 *= f;
                A3 *= fi;
                A4 *= fi;
            }
            const double Mxx = m[p][0], Mxy = m[p][1], Mxs = m[p][2], Myy = m[p][3], Mys = m[p][4], Mss = m[p][5];
            qx = fma(Mxx, A1, fma(Mxs, A5 + A7 + A2, fma(Mss, A3 + A8, fma(Mxy, A6, fma(Mys, A4, qx)))));
            qy = fma(Mxy, A7, fma(Mys, A3 + A8 + A6, fma(Mxs, A1, fma(Mss, A5 + A2, fma(Myy, A4, qy)))));
        }
    return make_double2(qx, qy);
}

template <bool RAGGED = false, class MF2, class XF>
__device__ __forceinline__ double2 grid_apply_pairs(int nxn, int nyn, int nel, const double *tab, int i, MF2 mf2, XF xf,
                                                    double rx = 1., double ry = 1., const double *colr = nullptr)
{
    const int j = i / nyn, k = i - j * nyn;
    return grid_apply_pairs_jk<RAGGED>(nxn, nyn, nel, tab, j, k, mf2, [&](int jj, int kk) { return xf(jj * nyn + kk); }, rx, ry, colr);
}

template <bool RAGGED = false, class XF>
__device__ __forceinline__ double2 grid_apply(const KOp &g, int i, XF xf)
{
    const double2 *M2 = reinterpret_cast<const double2 *>(g.M);
    return grid_apply_pairs<RAGGED>(g.nxn, g.nyn, g.nel, g.tab, i, [&](int q) { return M2[q]; }, xf, g.rx, g.ry, RAGGED ? g.colr : nullptr);
}

// GRID: 0 block-ELL matrix, 1 matrix-free (all cells alike), 2 matrix-free with cell shape factors: a level whose last column /
// row differs (KOp::rx, ry) or a fine grid with per-column widths (KOp::colr)
template <int GRID, class XF>
__device__ __forceinline__ double2 op_apply(const KOp &o, int i, XF xf)
{
    if (GRID == 2) return grid_apply<true>(o, i, xf);
    if (GRID) return grid_apply<false>(o, i, xf);
    return bell_apply(o.nnode, o.nslot, o.col, o.val, i, xf);
}

// ---------------------------------------------------------------------------------------------
// The same operator, MARCHING along x with the 3 x 3 stencil window in registers (round 3).  Measured with rocprofv3 at
// 2048^2 -- one pass = 470 MB, beyond the 256 MiB Infinity Cache -- the gather form above moves 1.15x (smoother) and 1.38x
// (PCG operator with its two gathered vectors) the algorithmic bytes: what the 9-node / 4-element gathers re-read has left
// the L2.  Here a WAVE owns 64 consecutive rows k and LC columns j; stepping j -> j + 1 it loads only the new vector column
// (3 entries per lane: rows k - 1, k, k + 1) and the new element column (2 elements x 3 generator pairs), the other six
// vector entries and two elements stay in registers: 3 (1 + 2/LC) + 6 (1 + 1/LC) loads per node instead of 9 + 12, all
// coalesced.  Same arithmetic in the same order as grid_apply_pairs: bit-identical results (tools/probes/march_probe.hip:
// 2048^2 smoother 122 -> 98 us, PCG operator 148 -> 103 us; 1024^2, Infinity-Cache resident: 23.1 -> 24.0 and 28.4 -> 26.7 us).
// Tasks (row chunk rk fastest, column range rj): XCD x = blockIdx % 8 takes the contiguous range [x ntask/8, (x+1) ntask/8).
//   xf(node) -> double2 vector entry;  emit(node, K-row result, centre entry) for every node of the task
struct Gen3 {
    double2 a, b, c;   // (XX,XY) (XS,YY) (YS,SS) of one element
};

__device__ __forceinline__ double2 stencil_window(const double2 (&u)[3][3], const Gen3 (&m)[2][2], const double *tab)
{
    double qx = 0., qy = 0.;
#pragma unroll
    for (int pj = 0; pj < 2; pj++)
#pragma unroll
        for (int pk = 0; pk < 2; pk++) {
            const int p = pj * 2 + pk;
            const double *T = tab + p * 16;  // wave-uniform -> scalar loads
            double A1 = 0., A2 = 0., A3 = 0., A4 = 0., A5 = 0., A6 = 0., A7 = 0., A8 = 0.;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const double2 ub = u[pj + (b >> 1)][pk + (b & 1)];
                const double sxx = T[b * 4 + 0], syy = T[b * 4 + 1], sxy = T[b * 4 + 2], syx = T[b * 4 + 3];
                A1 = fma(sxx, ub.x, A1);
                A2 = fma(sxx, ub.y, A2);
                A3 = fma(syy, ub.x, A3);
                A4 = fma(syy, ub.y, A4);
                A5 = fma(sxy, ub.x, A5);
                A6 = fma(sxy, ub.y, A6);
                A7 = fma(syx, ub.x, A7);
                A8 = fma(syx, ub.y, A8);
            }
            const double Mxx = m[pj][pk].a.x, Mxy = m[pj][pk].a.y, Mxs = m[pj][pk].b.x, Myy = m[pj][pk].b.y, Mys = m[pj][pk].c.x,
                         Mss = m[pj][pk].c.y;
            qx = fma(Mxx, A1, fma(Mxs, A5 + A7 + A2, fma(Mss, A3 + A8, fma(Mxy, A6, fma(Mys, A4, qx)))));
            qy = fma(Mxy, A7, fma(Mys, A3 + A8 + A6, fma(Mxs, A1, fma(Mss, A5 + A2, fma(Myy, A4, qy)))));
        }
    return make_double2(qx, qy);
}

constexpr int MARCH_LC = 8;   // columns per wave task

template <int LC, class XF, class EM>
__device__ __forceinline__ void grid_march(const KOp &g, XF xf, EM emit)
{
    const int nxn = g.nxn, nyn = g.nyn, nel = g.nel;
    const double2 *__restrict__ M2 = reinterpret_cast<const double2 *>(g.M);
    const double *tab = g.tab;
    const int nye = nyn - 1, nxe = nxn - 1;
    const int nrk = (nyn + 63) >> 6, nrj = (nxn + LC - 1) / LC, ntask = nrk * nrj;
    const int lane = threadIdx.x & 63;
    constexpr int wpb = BLOCK >> 6;
    int t0, t1, stride, first;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nbx = gridDim.x >> 3;
        t0 = (int)((long long)ntask * xcd / 8);
        t1 = (int)((long long)ntask * (xcd + 1) / 8);
        first = t0 + lb * wpb + (threadIdx.x >> 6);
        stride = nbx * wpb;
    } else {
        t0 = 0;
        t1 = ntask;
        first = blockIdx.x * wpb + (threadIdx.x >> 6);
        stride = gridDim.x * wpb;
    }
    for (int task = first; task < t1; task += stride) {
        const int rj = task / nrk, rk = task - rj * nrk;
        const int k = (rk << 6) + lane;
        const bool act = k < nyn;
        const int kc = min(k, nye), km = max(kc - 1, 0), kp = min(kc + 1, nye);
        const int j0 = rj * LC, j1 = min(j0 + LC, nxn);
        const int ek0 = min(max(kc - 1, 0), nye - 1), ek1 = min(kc, nye - 1);  // element rows k - 1, k (clamped: every load in range)
        const bool ok0 = kc - 1 >= 0, ok1 = kc < nye;
        double2 u[3][3];
        Gen3 m[2][2];
        auto load_col = [&](int jj, double2(&col)[3]) {
            const int jc = min(max(jj, 0), nxe);
            col[0] = xf(jc * nyn + km);
            col[1] = xf(jc * nyn + kc);
            col[2] = xf(jc * nyn + kp);
        };
        auto load_el = [&](int ej, Gen3(&gg)[2]) {
            const bool okj = ej >= 0 && ej < nxe;
            const int ec = min(max(ej, 0), nxe - 1);
            const size_t e0 = (size_t)ec * nye + ek0, e1 = (size_t)ec * nye + ek1;
            Gen3 g0 = {M2[e0], M2[(size_t)nel + e0], M2[(size_t)2 * nel + e0]};
            Gen3 g1 = {M2[e1], M2[(size_t)nel + e1], M2[(size_t)2 * nel + e1]};
            const double2 z = make_double2(0., 0.);
            if (!(okj && ok0)) g0 = {z, z, z};   // elements outside the grid enter with M = 0
            if (!(okj && ok1)) g1 = {z, z, z};
            gg[0] = g0;
            gg[1] = g1;
        };
        load_col(j0 - 1, u[0]);
        load_col(j0, u[1]);
        load_el(j0 - 1, m[0]);
        for (int j = j0; j < j1; j++) {
            load_col(j + 1, u[2]);
            load_el(j, m[1]);
            if (act) emit(j * nyn + k, stencil_window(u, m, tab), u[1][1]);
#pragma unroll
            for (int r = 0; r < 3; r++) {
                u[0][r] = u[1][r];
                u[1][r] = u[2][r];
            }
            m[0][0] = m[1][0];
            m[0][1] = m[1][1];
        }
    }
}

// Everything the matrix-free operator of one grid level needs after the generators changed, in ONE pass over them
// (node (j,k) reads its <= 4 elements):
//   diag   diagonal of K (what k_assemble writes for the block-ELL matrix) -> Jacobi smoother / preconditioner
//   Msnap  (finest level) copy of the generators: g.M is the live array that the material sweep keeps updating, but the
//          operator must stay the one of this moment (the reference's K = setupK() is a snapshot, model.py:1333, used
//          until the next setupK even when the last sweep of a load step changed tangents, :1384)
//   Mc     generators of the next coarser level = mean of the four children (node (2J,2K) owns coarse element (J,K))
//   dinv   (coarse levels, Dirichlet set known) free ? 1/|diag| : 0 with the mask of the coincident finest-grid node
//          (node (j << shift, k << shift) of the grid with mask_nyn nodes per column)
// SRC_PAIR: layout of g.M (0: the live SoA array of the finest level; 1: pair layout, every other level); Msnap and Mc are
// written in pair layout.
// Loads: a thread walks SETUP_COLS consecutive columns at one row k.  A node's four elements are the pair (j-1,k), (j,k) of
// its own row -- the second one loaded, the first one kept from the column before -- and the same pair of the node BELOW it,
// which is the neighbouring lane (lane 0 of a wave fetches it itself): 7.5 loads per node instead of 24, and the coarse
// generators are formed where all four children are already in registers -- at the odd/odd node (2J+1, 2K+1) -- instead of 18
// more loads at the even/even one.  Sums in the order of the node loop they replace: bit-identical (tools/probes/lib_ab.py).
// SETUP_COLS = 1 on the small levels (a walk of four columns is four dependent rounds of loads where the launch is all
// there is: 5.5 -> 8.5 us), 4 from 2^19 nodes (1024^2: 48.8 -> 35 us; 8 columns leave two waves per SIMD: not faster).
__host__ __device__ inline size_t grid_setup_tasks(int nxn, int nyn, int cols) { return (size_t)((nxn + cols - 1) / cols) * nyn; }
template <int SRC_PAIR, int SETUP_COLS = 1>
__global__ void __launch_bounds__(BLOCK)
k_grid_setup(KOp g, double2 *__restrict__ diag, double *__restrict__ Msnap, double *__restrict__ Mc,
             const double2 *__restrict__ mask_dinv, int mask_nyn, int shift, double2 *__restrict__ dinv, int mask_nxn = 0x7fffffff,
             const int *__restrict__ skip = nullptr /* speculative launch behind k_sweep_flags: nothing to do if no tangent changed */)
{
    if (skip && *skip) return;
    const int nyn = g.nyn, nye = nyn - 1, nxe = g.nxn - 1;
    const int nyc = nye >> 1;
    const size_t nel_c = (size_t)(nxe >> 1) * nyc;
    const int lane = threadIdx.x & 63;
    auto load6 = [&](size_t e, double(&o)[6]) {
        if (SRC_PAIR) {
            const double2 *M2 = reinterpret_cast<const double2 *>(g.M);
            const double2 a01 = M2[e], a23 = M2[(size_t)g.nel + e], a45 = M2[(size_t)2 * g.nel + e];
            o[0] = a01.x, o[1] = a01.y, o[2] = a23.x, o[3] = a23.y, o[4] = a45.x, o[5] = a45.y;
        } else {
#pragma unroll
            for (int c = 0; c < 6; c++) o[c] = g.M[(size_t)c * g.nel + e];
        }
    };
    const int ntask = (int)grid_setup_tasks(g.nxn, nyn, SETUP_COLS);
    for (int t = blockIdx.x * BLOCK + threadIdx.x; t < ntask; t += gridDim.x * BLOCK) {
      const int jg = t / nyn, k = t - jg * nyn;
      const bool okk[2] = {k >= 1, k < nye};
      double el[2][2][6];   // el[pj][pk] = generators of element (j-1+pj, k-1+pk), zero where it does not exist
      const int jlo = jg * SETUP_COLS, jhi = min(jlo + SETUP_COLS, g.nxn);
      for (int j = jlo; j < jhi; j++) {
        const int i = j * nyn + k;
        double dx = 0., dy = 0.;
        const bool okj[2] = {j >= 1, j < nxe};
#pragma unroll
        for (int pj = 0; pj < 2; pj++) {
            if (SETUP_COLS > 1 && pj == 0 && j > jlo) continue;   // the pair of column j-1 is the one kept from the step before
#pragma unroll
            for (int c = 0; c < 6; c++) el[pj][1][c] = 0.;
            if (okj[pj] && okk[1]) load6((size_t)(j - 1 + pj) * nye + k, el[pj][1]);
        }
#pragma unroll
        for (int pj = 0; pj < 2; pj++) {
            if (SETUP_COLS > 1 && pj == 0 && j > jlo) continue;
#pragma unroll
            for (int c = 0; c < 6; c++) el[pj][0][c] = __shfl_up(el[pj][1][c], 1);   // lane - 1 is node (j, k-1) when k >= 1
            if (lane == 0 && okk[0] && okj[pj]) load6((size_t)(j - 1 + pj) * nye + k - 1, el[pj][0]);
        }
#pragma unroll
        for (int pj = 0; pj < 2; pj++)
#pragma unroll
            for (int pk = 0; pk < 2; pk++) {
                if (!(okj[pj] && okk[pk])) continue;
                const int ej = j - 1 + pj, ek = k - 1 + pk;
                const int a = (1 - pj) * 2 + (1 - pk);
                const double *T = g.tab + (pj * 2 + pk) * 16 + a * 4;   // b = a
                double sxx = T[0], syy = T[1];
                const double sxy = T[2], syx = T[3];
                if (g.colr) {   // per-column widths of the finest grid: the true diagonal
                    const double csx = g.colr[ej];
                    sxx /= csx;
                    syy *= csx;
                } else if (g.rx != 1. || g.ry != 1.) {
                    // Coarse level whose last column / row has another size: NOT the true diagonal (Sxx ~ csy / csx, Syy ~ csx /
                    // csy) but every cell's contribution scaled with its AREA.  A field that is uniform along y has the residual
                    // h_k (A_x u)(x) at row k (h_k = tributary height), and only a scaling M_k ~ h_k keeps the damped-Jacobi
                    // sweep u + omega M^-1 r uniform along y -- the true diagonal a h_k + b (1 / h_below + 1 / h_above) / 2 is
                    // not.  The homogeneous workload lives on that: a tangent update changes its solution by the linear field
                    // (beta X, 0), the whole PCG iteration stays in the fields that are uniform along y and converges in 5-7
                    // steps; with the true diagonal on a level with an odd row the first V-cycle scatters the error over the
                    // soft modes of the plastic tangent and the same solve takes 26-60 (measured from the host with
                    // plfx_precond_apply, tools/probes/vcycle_pcg_host.py: 128 x 128 7, 128 x 127 60+, 127 x 128 7 iterations).
                    // Cells are never narrower than their neighbours (mg_coarse_ratio: r in [1, 2)), so this scaling is >= the
                    // true diagonal (stable, over-damped by <= r at the nodes of the odd cells only).
                    const double csx = (ej == mg_odd_cell(nxe)) ? g.rx : 1., csy = (ek == mg_odd_cell(nye)) ? g.ry : 1.;
                    sxx *= PLFX_MG_ODD_RULE ? csx * csy : csy / csx;
                    syy *= PLFX_MG_ODD_RULE ? csx * csy : csx / csy;
                }
                const double Mxx = el[pj][pk][0], Mxs = el[pj][pk][2], Myy = el[pj][pk][3], Mys = el[pj][pk][4], Mss = el[pj][pk][5];
                dx += Mxx * sxx + Mxs * (sxy + syx) + Mss * syy;
                dy += Myy * syy + Mys * (syx + sxy) + Mss * sxx;
            }
        if (Msnap && okj[1] && okk[1]) {   // node (j,k) "owns" element (j,k)
            const size_t e = (size_t)j * nye + k;
            double2 *S2 = reinterpret_cast<double2 *>(Msnap);
            S2[e] = make_double2(el[1][1][0], el[1][1][1]);
            S2[(size_t)g.nel + e] = make_double2(el[1][1][2], el[1][1][3]);
            S2[(size_t)2 * g.nel + e] = make_double2(el[1][1][4], el[1][1][5]);
        }
        diag[i] = make_double2(dx, dy);
        if (dinv) {
            // (interior node lines of a level are node lines j << level of the finest grid counted from the side of the regular
            // cells, the line beyond the odd cell is the edge)
            int mj, mk;
            if (PLFX_MG_RAGGED_FIRST) {
                mj = (j == 0) ? 0 : max(mask_nxn - 1 - ((nxe - j) << shift), 0);
                mk = (k == 0) ? 0 : max(mask_nyn - 1 - ((nye - k) << shift), 0);
            } else {
                mj = (j == nxe) ? mask_nxn - 1 : min(j << shift, mask_nxn - 1);
                mk = (k == nye) ? mask_nyn - 1 : min(k << shift, mask_nyn - 1);
            }
            const double2 df = mask_dinv[(size_t)mj * mask_nyn + mk];
            double2 o;
            o.x = (df.x != 0.) ? (fabs(dx) > 1e-300 ? 1. / fabs(dx) : 1.) : 0.;
            o.y = (df.y != 0.) ? (fabs(dy) > 1e-300 ? 1. / fabs(dy) : 1.) : 0.;
            dinv[i] = o;
        }
        // (levels that halve exactly; otherwise the host runs k_mg_coarsen_M): coarse element (J,K) has the children
        // (2J,2K) (2J,2K+1) (2J+1,2K) (2J+1,2K+1) = the four elements of node (2J+1, 2K+1), summed in that order
        if (Mc && (j & 1) && (k & 1) && okj[1] && okk[1]) {
            const size_t ec = (size_t)(j >> 1) * nyc + (k >> 1);
            double mc[6];
#pragma unroll
            for (int c = 0; c < 6; c++) mc[c] = 0.25 * (el[0][0][c] + el[0][1][c] + el[1][0][c] + el[1][1][c]);
            double2 *C2 = reinterpret_cast<double2 *>(Mc);
            C2[ec] = make_double2(mc[0], mc[1]);
            C2[nel_c + ec] = make_double2(mc[2], mc[3]);
            C2[2 * nel_c + ec] = make_double2(mc[4], mc[5]);
        }
        if (SETUP_COLS > 1) {
#pragma unroll
            for (int pk = 0; pk < 2; pk++)
#pragma unroll
                for (int c = 0; c < 6; c++) el[0][pk][c] = el[1][pk][c];
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// SpMV, one thread per node (two rows); GRID selects the operator form (KOp).

// MODE 0: q = K p                                   (plain; used for K w, K du, residual)
// MODE 1: PCG step: beta = rz_new/rz_old from the partial sums of the previous update kernel,
//         p_new = z + beta p_old (written to pnew), q = K p_new, partial sums of p_new . q.
// MODE 2: first PCG step: p_new = z, q = K z, partial sums of z . q (p_old is neither initialised nor read)
// MODE 3: end of a load step (model.py:1383-1384): u += du, f += K du in one pass -- p = du, pnew = u, q = f; K du itself is not stored
struct CgScalars {
    double thresh2;  // (rtol * |b|)^2
    int32_t done;    // sticky convergence flag
    int32_t iters;   // iteration count at convergence
    double rr_final;
    double aux;      // step of the interpolated start (k_pred_try; travels to the host with the flags)
};

template <int MODE, int GRID>
__global__ void __launch_bounds__(BLOCK)
k_spmv(KOp op, int n_begin, int n_end,
       const double2 *__restrict__ p, const double2 *__restrict__ z, double2 *__restrict__ pnew,
       double2 *__restrict__ q, const double *__restrict__ part_rz_new, const double *__restrict__ part_rz_old,
       const double *__restrict__ part_rr, int npart_prev, double *__restrict__ part_pq, CgScalars *__restrict__ sc, int it,
       int own_lo, int own_hi /* nodes whose p.q this rank sums (strip: owned columns; else all) */)
{
    __shared__ double sh[BLOCK / 64];
    double beta = 0.;
    if ((MODE == 0 || MODE == 3) && sc != nullptr && sc->done) return;   // (speculatively enqueued K d of the interpolated start / K du of the end of a load step)
    if (MODE == 1 || MODE == 2) {
        if (sc->done) return;
        double rr, rzn = 0., rzo = 1.;
        if (MODE == 1) {  // MODE 2 = first iteration: p = z (beta = 0, p_old is not initialised and never read)
            const double *const arr[3] = {part_rr, part_rz_new, part_rz_old};
            double o[3];
            sum_partials_n<3>(arr, npart_prev, o);
            rr = o[0];
            rzn = o[1];
            rzo = o[2];
        } else {
            rr = sum_partials(part_rr, npart_prev, sh);
        }
        if (rr <= sc->thresh2 || !(rr == rr)) {  // all blocks take the same decision from the same partials
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                sc->done = (rr == rr) ? 1 : 2;   // 2 = breakdown (NaN residual)
                sc->iters = it;
                sc->rr_final = rr;
            }
            return;
        }
        if (MODE == 1) beta = rzn / rzo;
    }
    double acc_pq = 0.;
    const int nb = gridDim.x;
    const int span = n_end - n_begin;
    (void)op.nnode;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < span; t += nb) {
        const int i = n_begin + t * BLOCK + threadIdx.x;
        if (i >= n_end) continue;
        double2 qv;
        if (MODE == 1)
            qv = op_apply<GRID>(op, i, [&](int j) {
                const double2 zj = z[j], po = p[j];
                return make_double2(fma(beta, po.x, zj.x), fma(beta, po.y, zj.y));
            });
        else if (MODE == 2)
            qv = op_apply<GRID>(op, i, [&](int j) { return z[j]; });
        else
            qv = op_apply<GRID>(op, i, [&](int j) { return p[j]; });
        const double qx = qv.x, qy = qv.y;
        if (MODE == 3) {   // pnew = u, q = f: u += du, f += K du in the pass that forms K du (k_axpy_uf's two additions)
            const double2 d = p[i];
            double2 uu = pnew[i], ff = q[i];
            uu.x += d.x;
            uu.y += d.y;
            ff.x += qx;
            ff.y += qy;
            pnew[i] = uu;
            q[i] = ff;
        } else
            q[i] = make_double2(qx, qy);
        if (MODE == 1) {
            const double2 zi = z[i], po = p[i];
            double2 pn;
            pn.x = fma(beta, po.x, zi.x);
            pn.y = fma(beta, po.y, zi.y);
            pnew[i] = pn;
            if (i >= own_lo && i < own_hi) acc_pq = fma(pn.x, qx, fma(pn.y, qy, acc_pq));
        } else if (MODE == 2) {
            const double2 zi = z[i];
            pnew[i] = zi;
            if (i >= own_lo && i < own_hi) acc_pq = fma(zi.x, qx, fma(zi.y, qy, acc_pq));
        }
    }
    if (MODE == 1 || MODE == 2) {
        const double t = block_sum(acc_pq, sh);
        if (threadIdx.x == 0) part_pq[blockIdx.x] = t;
    }
}

// k_spmv<MODE, 1> (MODE 1 / 2) of the finest grid in marching form (grid_march): same scalars, same partial-sum slots --
// one partial per block, summed over the nodes its waves emit
template <int MODE>
__global__ void __launch_bounds__(BLOCK)
k_spmv_march(KOp op, const double2 *__restrict__ p, const double2 *__restrict__ z, double2 *__restrict__ pnew,
             double2 *__restrict__ q, const double *__restrict__ part_rz_new, const double *__restrict__ part_rz_old,
             const double *__restrict__ part_rr, int npart_prev, double *__restrict__ part_pq, CgScalars *__restrict__ sc, int it,
             int own_lo, int own_hi)
{
    __shared__ double sh[BLOCK / 64];
    double beta = 0.;
    if (sc->done) return;
    double rr, rzn = 0., rzo = 1.;
    if (MODE == 1) {
        const double *const arr[3] = {part_rr, part_rz_new, part_rz_old};
        double o[3];
        sum_partials_n<3>(arr, npart_prev, o);
        rr = o[0];
        rzn = o[1];
        rzo = o[2];
    } else {
        rr = sum_partials(part_rr, npart_prev, sh);
    }
    if (rr <= sc->thresh2 || !(rr == rr)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            sc->done = (rr == rr) ? 1 : 2;
            sc->iters = it;
            sc->rr_final = rr;
        }
        return;
    }
    if (MODE == 1) beta = rzn / rzo;
    double acc_pq = 0.;
    grid_march<MARCH_LC>(
        op,
        [&](int n) {
            const double2 zj = z[n];
            if (MODE == 2) return zj;
            const double2 po = p[n];
            return make_double2(fma(beta, po.x, zj.x), fma(beta, po.y, zj.y));
        },
        [&](int i, double2 qv, double2 pn) {
            q[i] = qv;
            pnew[i] = pn;
            if (i >= own_lo && i < own_hi) acc_pq = fma(pn.x, qv.x, fma(pn.y, qv.y, acc_pq));
        });
    const double t = block_sum(acc_pq, sh);
    if (threadIdx.x == 0) part_pq[blockIdx.x] = t;
}

// partial sums of p.q over all nodes (multi-GPU: after the all-reduce of q), and the p update for
// nodes outside the rank's own SpMV range
__global__ void __launch_bounds__(BLOCK)
k_dot_pq(int nnode, const double2 *__restrict__ p, const double2 *__restrict__ q, double *__restrict__ part_pq, const CgScalars *__restrict__ sc)
{
    __shared__ double sh[BLOCK / 64];
    if (sc->done) return;
    double acc = 0.;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nnode; i += gridDim.x * BLOCK) {
        const double2 a = p[i], b = q[i];
        acc = fma(a.x, b.x, fma(a.y, b.y, acc));
    }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) part_pq[blockIdx.x] = t;
}

// p_new = z + beta p_old for nodes outside [n_begin, n_end) (multi-GPU only)
__global__ void __launch_bounds__(BLOCK)
k_p_update_outside(int nnode, int n_begin, int n_end, const double2 *__restrict__ p, const double2 *__restrict__ z,
                   double2 *__restrict__ pnew, double2 *__restrict__ q, const double *__restrict__ part_rz_new, const double *__restrict__ part_rz_old,
                   const double *__restrict__ part_rr, int npart_prev, const CgScalars *__restrict__ sc)
{
    __shared__ double sh[BLOCK / 64];
    if (sc->done) return;
    const double rr = sum_partials(part_rr, npart_prev, sh);
    if (rr <= sc->thresh2) return;
    const double rzn = sum_partials(part_rz_new, npart_prev, sh);
    const double rzo = sum_partials(part_rz_old, npart_prev, sh);
    const double beta = rzn / rzo;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nnode; i += gridDim.x * BLOCK) {
        if (i >= n_begin && i < n_end) continue;
        const double2 zi = z[i], po = p[i];
        pnew[i] = make_double2(fma(beta, po.x, zi.x), fma(beta, po.y, zi.y));
        q[i] = make_double2(0., 0.);
    }
}

// alpha = rz/pq;  x += alpha p;  r -= alpha q (free DOFs only);  z = dinv r;  partials r.z, r.r
__global__ void __launch_bounds__(BLOCK)
k_cg_update(int nnode, const double2 *__restrict__ p, const double2 *__restrict__ q, const double2 *__restrict__ dinv, double2 *__restrict__ x,
            double2 *__restrict__ r, double2 *__restrict__ z, const double *__restrict__ part_pq, int npart_pq, const double *__restrict__ part_rz,
            const double *__restrict__ part_rr_prev, int npart_prev, double *__restrict__ part_rz_out, double *__restrict__ part_rr_out,
            CgScalars *__restrict__ sc)
{
    __shared__ double sh[BLOCK / 64];
    if (sc->done) return;
    // same decision as the k_spmv<1> of this iteration (it set sc->done only if rr <= thresh2,
    // and that write is visible here because it happened in an earlier kernel)
    const double pq = sum_partials(part_pq, npart_pq, sh);
    const double rz = sum_partials(part_rz, npart_prev, sh);
    (void)part_rr_prev;
    if (!(pq > 0.)) {  // p^T K p <= 0: operator or preconditioner not positive definite -> stop, keep x
        if (blockIdx.x == 0 && threadIdx.x == 0) sc->done = 2;
        return;
    }
    const double alpha = rz / pq;
    double a_rz = 0., a_rr = 0.;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nnode; i += gridDim.x * BLOCK) {
        const double2 pi = p[i], qi = q[i], di = dinv[i];
        double2 xi = x[i], ri = r[i], zi;
        xi.x = fma(alpha, pi.x, xi.x);
        xi.y = fma(alpha, pi.y, xi.y);
        ri.x = (di.x != 0.) ? fma(-alpha, qi.x, ri.x) : 0.;
        ri.y = (di.y != 0.) ? fma(-alpha, qi.y, ri.y) : 0.;
        zi.x = di.x * ri.x;
        zi.y = di.y * ri.y;
        x[i] = xi;
        r[i] = ri;
        z[i] = zi;
        a_rz = fma(ri.x, zi.x, fma(ri.y, zi.y, a_rz));
        a_rr = fma(ri.x, ri.x, fma(ri.y, ri.y, a_rr));
    }
    const double t1 = block_sum(a_rz, sh);
    const double t2 = block_sum(a_rr, sh);
    if (threadIdx.x == 0) {
        part_rz_out[blockIdx.x] = t1;
        part_rr_out[blockIdx.x] = t2;
    }
}

// Start of a PCG solve in one pass: q = K x0 (warm start), r = mask (b - q), z = dinv r and the partial sums of r.z,
// r.r, b.b -- k_spmv<0> + k_cg_init without the round trip of q through memory.
template <int GRID>
__global__ void __launch_bounds__(BLOCK)
k_cg_start(KOp op, int nnode, int warm, const double2 *__restrict__ x, const double2 *__restrict__ b,
           const double2 *__restrict__ dinv, double2 *__restrict__ r, double2 *__restrict__ z,
           double *__restrict__ part_rz_out, double *__restrict__ part_rr_out, double *__restrict__ part_bb_out,
           int own_lo, int own_hi /* nodes whose sums this rank takes (strip: owned columns; else all) */)
{
    __shared__ double sh[BLOCK / 64];
    double a_rz = 0., a_rr = 0., a_bb = 0.;
    const int nb = gridDim.x;
    for (int t = xcd_tile(blockIdx.x, nb); t * BLOCK < nnode; t += nb) {
        const int i = t * BLOCK + threadIdx.x;
        if (i >= nnode) continue;
        double2 qi = make_double2(0., 0.);
        if (warm) qi = op_apply<GRID>(op, i, [&](int j) { return x[j]; });
        const double2 bi = b[i], di = dinv[i];
        double2 ri, zi;
        ri.x = (di.x != 0.) ? bi.x - qi.x : 0.;
        ri.y = (di.y != 0.) ? bi.y - qi.y : 0.;
        zi.x = di.x * ri.x;
        zi.y = di.y * ri.y;
        r[i] = ri;
        z[i] = zi;
        if (i >= own_lo && i < own_hi) {
            a_rz = fma(ri.x, zi.x, fma(ri.y, zi.y, a_rz));
            a_rr = fma(ri.x, ri.x, fma(ri.y, ri.y, a_rr));
            const double bx = (di.x != 0.) ? bi.x : 0., by = (di.y != 0.) ? bi.y : 0.;
            a_bb = fma(bx, bx, fma(by, by, a_bb));
        }
    }
    const double t1 = block_sum(a_rz, sh);
    const double t2 = block_sum(a_rr, sh);
    const double t3 = block_sum(a_bb, sh);
    if (threadIdx.x == 0) {
        part_rz_out[blockIdx.x] = t1;
        part_rr_out[blockIdx.x] = t2;
        part_bb_out[blockIdx.x] = t3;
    }
}

// r = mask (b - q), z = dinv r, partial r.z, r.r   (initial residual; q = K x0)
__global__ void __launch_bounds__(BLOCK)
k_cg_init(int nnode, const double2 *__restrict__ b, const double2 *__restrict__ q, const double2 *__restrict__ dinv, double2 *__restrict__ r, double2 *__restrict__ z,
          double *__restrict__ part_rz_out, double *__restrict__ part_rr_out, double *__restrict__ part_bb_out)
{
    __shared__ double sh[BLOCK / 64];
    double a_rz = 0., a_rr = 0., a_bb = 0.;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nnode; i += gridDim.x * BLOCK) {
        const double2 bi = b[i], di = dinv[i];
        double2 ri, zi;
        const double2 qi = q ? q[i] : make_double2(0., 0.);
        ri.x = (di.x != 0.) ? bi.x - qi.x : 0.;
        ri.y = (di.y != 0.) ? bi.y - qi.y : 0.;
        zi.x = di.x * ri.x;
        zi.y = di.y * ri.y;
        r[i] = ri;
        z[i] = zi;
        a_rz = fma(ri.x, zi.x, fma(ri.y, zi.y, a_rz));
        a_rr = fma(ri.x, ri.x, fma(ri.y, ri.y, a_rr));
        const double bx = (di.x != 0.) ? bi.x : 0., by = (di.y != 0.) ? bi.y : 0.;
        a_bb = fma(bx, bx, fma(by, by, a_bb));
    }
    const double t1 = block_sum(a_rz, sh);
    const double t2 = block_sum(a_rr, sh);
    const double t3 = block_sum(a_bb, sh);
    if (threadIdx.x == 0) {
        part_rz_out[blockIdx.x] = t1;
        part_rr_out[blockIdx.x] = t2;
        part_bb_out[blockIdx.x] = t3;
    }
}

// finalise set-up scalars on one block: thresh2 = rtol^2 * max(bb, tiny), reset flags
__global__ void k_cg_setup(const double *part_bb, int npart, double rtol, CgScalars *sc)
{
    __shared__ double sh[BLOCK / 64];
    const double bb = sum_partials(part_bb, npart, sh);
    if (threadIdx.x == 0) {
        sc->thresh2 = rtol * rtol * bb;
        sc->done = 0;
        sc->iters = -1;
        sc->rr_final = -1.;
        sc->aux = 0.;
    }
}

// k_cg_setup + k_cg_check in one launch (start of a solve)
struct CgMbox;
__global__ void k_cg_setup_check(const double *part_bb, const double *part_rr, int npart, double rtol, CgScalars *sc, int it_done,
                                 CgMbox *mb, unsigned long long seq);

// Pinned, host-visible copy of the PCG scalars: a kernel posts them with a sequence number and the host spins on the
// number instead of enqueueing a device->host copy and synchronising the stream (12 instead of 19 us per round trip on
// this system, tools/probes/sync_probe.hip).
struct CgMbox {
    unsigned long long seq;
    CgScalars sc;
};

__device__ void spec_set(CgScalars *spec, int done) { spec->done = done; }

__device__ void mbox_publish(CgMbox *mb, unsigned long long seq)
{
    __threadfence_system();
    __atomic_store_n(&mb->seq, seq, __ATOMIC_RELEASE);
}

// convergence test on the residual partials an update kernel just wrote (one block): lets the host stop
// BEFORE it launches the next preconditioner application (a V-cycle is the most expensive part of an iteration)
__global__ void k_cg_check(const double *part_rr, int npart, CgScalars *sc, int it_done, CgMbox *mb,
                           unsigned long long seq)
{
    __shared__ double sh[BLOCK / 64];
    if (!sc->done) {  // uniform: every thread reads the same flag
        const double rr = sum_partials(part_rr, npart, sh);
        if (threadIdx.x == 0 && (rr <= sc->thresh2 || !(rr == rr))) {
            sc->done = (rr == rr) ? 1 : 2;
            sc->iters = it_done;
            sc->rr_final = rr;
        }
    }
    if (mb && threadIdx.x == 0) {
        mb->sc = *sc;
        __threadfence_system();
        __atomic_store_n(&mb->seq, seq, __ATOMIC_RELEASE);
    }
}

__global__ void k_cg_setup_check(const double *part_bb, const double *part_rr, int npart, double rtol, CgScalars *sc, int it_done,
                                 CgMbox *mb, unsigned long long seq)
{
    __shared__ double sh[BLOCK / 64];
    const double bb = sum_partials(part_bb, npart, sh);
    const double rr = sum_partials(part_rr, npart, sh);
    if (threadIdx.x == 0) {
        const double th = rtol * rtol * bb;
        sc->thresh2 = th;
        sc->aux = 0.;
        if (rr <= th || !(rr == rr)) {
            sc->done = (rr == rr) ? 1 : 2;
            sc->iters = it_done;
            sc->rr_final = rr;
        } else {
            sc->done = 0;
            sc->iters = -1;
            sc->rr_final = -1.;
        }
        if (mb) {
            mb->sc = *sc;
            __threadfence_system();
            __atomic_store_n(&mb->seq, seq, __ATOMIC_RELEASE);
        }
    }
}

// copy n doubles of results to pinned host memory and post the sequence number (one block)
__global__ void __launch_bounds__(BLOCK)
k_mbox_post(const double *__restrict__ src, int n, double *__restrict__ dst, CgMbox *mb, unsigned long long seq)
{
    for (int i = threadIdx.x; i < n; i += BLOCK) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __atomic_store_n(&mb->seq, seq, __ATOMIC_RELEASE);
}

// final rr for reporting when the iteration limit was hit
__global__ void k_cg_final(const double *part_rr, int npart, CgScalars *sc)
{
    __shared__ double sh[BLOCK / 64];
    const double rr = sum_partials(part_rr, npart, sh);
    if (threadIdx.x == 0 && !sc->done) sc->rr_final = rr;
}

__global__ void __launch_bounds__(BLOCK) k_fill(double *a, size_t n, double v)
{
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) a[i] = v;
}

// y[idx[k]] = v[k]
__global__ void __launch_bounds__(BLOCK) k_scatter(int n, const int32_t *idx, const double *v, double *y)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < n) y[idx[k]] = v[k];
}

// y[idx[k]] = a[k], z[idx[k]] = b[k], (flag ? m[idx[k]] = 1)  : prescribed-DOF data of apply_bc in one launch
__global__ void __launch_bounds__(BLOCK)
k_scatter_bc(int n, const int32_t *idx, const double *a, const double *b, double *y, double *z, double *m, int flag)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    const int i = idx[k];
    y[i] = a[k];
    z[i] = b[k];
    if (flag) m[i] = 1.;
}

// the same for a registered BC plan: every prescribed DOF takes the value of its first entry's segment (y) and the sum
// over the segments of all its entries in entry order (z; a DOF on two edges counts twice, model.py:1115-1122); the
// per-segment values travel as kernel arguments
struct BcSegVals {
    static constexpr int N = 16;
    double v[N];
};

__global__ void __launch_bounds__(BLOCK)
k_scatter_bc_plan(int n, const int32_t *__restrict__ idx, const int32_t *__restrict__ seg4, BcSegVals sv,
                  double *__restrict__ y, double *__restrict__ z, double *__restrict__ m, int flag)
{
    __shared__ double val[BcSegVals::N];
    if (threadIdx.x < BcSegVals::N) val[threadIdx.x] = sv.v[threadIdx.x];
    __syncthreads();
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    const int i = idx[k];
    const double first = val[seg4[4 * k]];
    double w = 0.;
    w += first;
#pragma unroll
    for (int q = 1; q < 4; q++) {
        const int sg = seg4[4 * k + q];
        if (sg >= 0) w += val[sg];
    }
    y[i] = first;
    z[i] = w;
    if (flag) m[i] = 1.;
}

// va[k] = a[idx[k]], vb[k] = b[idx[k]]   (boundary values of u and f at the end of a load step: one launch)
__global__ void __launch_bounds__(BLOCK)
k_gather2(int n, const int32_t *__restrict__ idx, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ va,
          double *__restrict__ vb)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < n) {
        const int i = idx[k];
        va[k] = a[i];
        vb[k] = b[i];
    }
}

__global__ void __launch_bounds__(BLOCK) k_gather(int n, const int32_t *idx, const double *y, double *v)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < n) v[k] = y[idx[k]];
}

// q[i] = (K w)[i] for the listed nodes only: w is non-zero on prescribed DOFs, so K w vanishes except on the
// rows of nodes that touch a prescribed node (O(boundary) work instead of a full matrix pass)
template <int GRID>
__global__ void __launch_bounds__(BLOCK)
k_spmv_rows(int nlist, const int32_t *__restrict__ list, KOp op, const double2 *__restrict__ w,
            double2 *__restrict__ q)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= nlist) return;
    const int i = list[k];
    q[i] = op_apply<GRID>(op, i, [&](int j) { return w[j]; });
}

// rhs = fext - K w (q holds K w);  dinv = free ? 1/|diag| : 0
__global__ void __launch_bounds__(BLOCK)
k_bc_finish(size_t ndof, const double *__restrict__ fext, const double *__restrict__ kw, const double *__restrict__ diag,
            const double *__restrict__ is_presc, double *__restrict__ rhs, double *__restrict__ dinv)
{
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < ndof; i += (size_t)gridDim.x * BLOCK) {
        const bool free_dof = (is_presc[i] == 0.);
        rhs[i] = free_dof ? (fext ? fext[i] : 0.) - kw[i] : 0.;
        const double d = fabs(diag[i]);
        dinv[i] = free_dof ? (d > 1e-300 ? 1. / d : 1.) : 0.;
    }
}

// du = x (free) + du_presc (prescribed)
__global__ void __launch_bounds__(BLOCK)
k_compose_du(size_t ndof, const double *__restrict__ x, const double *__restrict__ dup, const double *__restrict__ is_presc, double *__restrict__ du)
{
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < ndof; i += (size_t)gridDim.x * BLOCK)
        du[i] = (is_presc[i] != 0.) ? dup[i] : x[i];
}

// x0 = warm ? du on free DOFs : 0
__global__ void __launch_bounds__(BLOCK)
k_x0(size_t ndof, const double *__restrict__ du, const double *__restrict__ is_presc, int warm, double scale, double *__restrict__ x)
{
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < ndof; i += (size_t)gridDim.x * BLOCK)
        x[i] = (warm && is_presc[i] == 0.) ? scale * du[i] : 0.;
}

// ---- initial guess of a warm-started solve from the last two solutions (plfx_solve)
// d = x - xprev   (x = the solution of the previous solve, xprev = the last solution that differed from it)
__global__ void __launch_bounds__(BLOCK)
k_pred_diff(size_t ndof, const double *__restrict__ x, const double *__restrict__ xprev, const double *__restrict__ dinv,
            double *__restrict__ d, const CgScalars *__restrict__ sc)
{
    if (sc->done) return;   // the plain start satisfies the tolerance: nothing to interpolate (enqueued before the host knows)
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < ndof; i += (size_t)gridDim.x * BLOCK)
        d[i] = (dinv[i] != 0.) ? x[i] - xprev[i] : 0.;   // (the Dirichlet set may have changed since xprev was a solution)
}

// xprev += d: the solution this solve started from becomes "the one before" (called when the solve moved x)
__global__ void __launch_bounds__(BLOCK)
k_pred_advance(size_t ndof, double *__restrict__ xprev, const double *__restrict__ d)
{
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < ndof; i += (size_t)gridDim.x * BLOCK) xprev[i] += d[i];
}

// partials of (K d) . r and (K d) . (K d) over the free DOFs (r = P (b - K x)): the step alpha that minimises | r - alpha P K d |
__global__ void __launch_bounds__(BLOCK)
k_pred_dots(size_t dof_lo, size_t dof_hi /* owned DOFs (a strip: its owned node columns) */, const double *__restrict__ dinv,
            const double *__restrict__ r, const double *__restrict__ kd, double *__restrict__ part, const CgScalars *__restrict__ sc)
{
    __shared__ double sh[BLOCK / 64];
    if (sc->done) return;
    double a0 = 0., a1 = 0.;
    for (size_t i = dof_lo + blockIdx.x * (size_t)BLOCK + threadIdx.x; i < dof_hi; i += (size_t)gridDim.x * BLOCK) {
        if (dinv[i] == 0.) continue;
        const double v = kd[i];
        a0 = fma(v, r[i], a0);
        a1 = fma(v, v, a1);
    }
    const double t0 = block_sum(a0, sh);
    const double t1 = block_sum(a1, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = t0;
        part[(size_t)MAXPART + blockIdx.x] = t1;
    }
}

// | P (r - alpha K d) |^2 over the owned free DOFs, nothing written: would x + alpha d satisfy the tolerance as it is?  (Round 6: the
// interpolated start is taken only when it FINISHES the solve; a solve that iterates starts from x, bit for bit the plain warm start.)
__global__ void __launch_bounds__(BLOCK)
k_pred_try(int nnode, CgScalars *__restrict__ sc, const double *__restrict__ part_dots, int gp, const double2 *__restrict__ r,
           const double2 *__restrict__ kd, const double2 *__restrict__ dinv, double *__restrict__ part_rr_out, int own_lo, int own_hi)
{
    __shared__ double sh[BLOCK / 64];
    if (sc->done) return;
    // alpha = (K d . r) / (K d . K d) clamped to [0, 1], steps below 0.01 dropped: every block sums the partials of k_pred_dots
    // itself (the same loads in the same order: the same alpha everywhere); block 0 leaves it in sc->aux for the commit and the host
    const double o0 = sum_partials(part_dots, gp, sh);
    const double o1 = sum_partials(part_dots + MAXPART, gp, sh);
    double alpha = (o1 > 0.) ? o0 / o1 : 0.;
    if (!(alpha == alpha) || alpha > 1.e300 || alpha < -1.e300) alpha = 0.;
    alpha = fmin(1., fmax(0., alpha));
    if (alpha < 0.01) alpha = 0.;
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->aux = alpha;
    if (alpha == 0.) {   // no step: the test that follows must fail
        if (threadIdx.x == 0) part_rr_out[blockIdx.x] = 1.e300;
        return;
    }
    double a_rr = 0.;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < nnode; i += gridDim.x * BLOCK) {
        if (i < own_lo || i >= own_hi) continue;
        const double2 ki = kd[i], dv = dinv[i], ri = r[i];
        const double rx = (dv.x != 0.) ? fma(-alpha, ki.x, ri.x) : 0.;
        const double ry = (dv.y != 0.) ? fma(-alpha, ki.y, ri.y) : 0.;
        a_rr = fma(rx, rx, fma(ry, ry, a_rr));
    }
    const double t = block_sum(a_rr, sh);
    if (threadIdx.x == 0) part_rr_out[blockIdx.x] = t;
}

// x += alpha d: the accepted start IS the solution (r and z of the finished solve are not read again)
__global__ void __launch_bounds__(BLOCK)
k_pred_commit(size_t ndof, const CgScalars *__restrict__ sc, double *__restrict__ x, const double *__restrict__ d)
{
    const double alpha = sc->aux;
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < ndof; i += (size_t)gridDim.x * BLOCK) x[i] = fma(alpha, d[i], x[i]);
}

// the accepted start in one pass (single GPU): x += alpha d, du = x on the free DOFs / the prescribed increment elsewhere
// (k_compose_du), and the history of the initial guess advances (xprev += d: the solution this solve started from)
__global__ void __launch_bounds__(BLOCK)
k_pred_finish(size_t ndof, const CgScalars *__restrict__ sc, double *__restrict__ x, const double *__restrict__ d,
              double *__restrict__ xprev, const double *__restrict__ dup, const double *__restrict__ is_presc, double *__restrict__ du)
{
    // enqueued BEFORE the host has seen the test (its round trip overlaps this pass): does nothing unless the interpolated
    // start was accepted (k_cg_check marks that with iters = -2)
    if (!(sc->done == 1 && sc->iters == -2)) return;
    const double alpha = sc->aux;
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < ndof; i += (size_t)gridDim.x * BLOCK) {
        const double di = d[i];
        const double xi = fma(alpha, di, x[i]);
        x[i] = xi;
        du[i] = (is_presc[i] != 0.) ? dup[i] : xi;
        xprev[i] += di;
    }
}

// u += du ; f += q  (q = K du)     (model.py:1383-1384)
__global__ void __launch_bounds__(BLOCK)
k_axpy_uf(size_t ndof, const double *__restrict__ du, const double *__restrict__ q, double *__restrict__ u, double *__restrict__ f)
{
    for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < ndof; i += (size_t)gridDim.x * BLOCK) {
        u[i] += du[i];
        f[i] += q[i];
    }
}

// element state update at the end of a load step (model.py:1385-1392); u already updated
template <int SUMS>
__global__ void __launch_bounds__(BLOCK)
k_update_state(const MatDev *__restrict__ gmat, const ClassDev *__restrict__ gcls, int nel, int e_off, const int32_t *__restrict__ conn,
               const int32_t *__restrict__ cls, const double2 *__restrict__ du2, const double2 *__restrict__ u2, double *__restrict__ sig, double *__restrict__ epl,
               double *__restrict__ eps, const double *__restrict__ elstiff, const double *__restrict__ res_sig, const double *__restrict__ res_depl,
               int nonlin, double *__restrict__ part /* SUMS: [18][gridDim.x] volume-weighted sums of the new sig, eps, epl */,
               int sum_lo = 0, int sum_hi = 0x7fffffff /* elements that enter the sums (strip: owned columns) */)
{
    // SUMS = 1 fuses calc_global's element sums (k_global_partials: same grid, same order -> identical numbers)
    double acc[18];
#pragma unroll
    for (int k = 0; k < 18; k++) acc[k] = 0.;
    for (int e = blockIdx.x * BLOCK + threadIdx.x; e < nel; e += gridDim.x * BLOCK) {
        const ClassDev &c = gcls[cls[e]];
        const MatDev &m = gmat[c.mat];
        const size_t ge = (size_t)e + e_off;
        const int n0 = conn[ge * 4], n1 = conn[ge * 4 + 1], n2 = conn[ge * 4 + 2], n3 = conn[ge * 4 + 3];
        double sn[6], pn[6];
        if (m.kind != 0 && nonlin) {  // el.res_sig is set (model.py:1390-1391)
#pragma unroll
            for (int k = 0; k < 6; k++) {
                pn[k] = epl[(size_t)k * nel + e] + res_depl[(size_t)k * nel + e];
                sn[k] = res_sig[(size_t)k * nel + e];
                epl[(size_t)k * nel + e] = pn[k];
                sig[(size_t)k * nel + e] = sn[k];
            }
        } else {  // el.sig += elstiff @ deps ; depl = 0 for elastic materials (model.py:1387-1388)
            double de[6], D[21], ds[6];
            class_strain(c, du2, n0, n1, n2, n3, de);
#pragma unroll
            for (int k = 0; k < 21; k++) D[k] = elstiff[(size_t)k * nel + e];
            symv(D, de, ds);
#pragma unroll
            for (int k = 0; k < 6; k++) {
                sn[k] = sig[(size_t)k * nel + e] + ds[k];
                sig[(size_t)k * nel + e] = sn[k];
                if (SUMS) pn[k] = epl[(size_t)k * nel + e];
            }
        }
        double et[6];
        class_strain(c, u2, n0, n1, n2, n3, et);  // el.eps = el.eps_t() (model.py:1392)
#pragma unroll
        for (int k = 0; k < 6; k++) eps[(size_t)k * nel + e] = et[k];
        if (SUMS && e >= sum_lo && e < sum_hi) {
            const double v = c.vel;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                acc[k] = fma(sn[k], v, acc[k]);
                acc[6 + k] = fma(et[k], v, acc[6 + k]);
                acc[12 + k] = fma(pn[k], v, acc[12 + k]);
            }
        }
    }
    if (SUMS) block_sums_to_partials(acc, part);
}

// calc_global sums (model.py:1500-1507): partials of sum(x*Vel) for the 18 components
__global__ void __launch_bounds__(BLOCK)
k_global_partials(const ClassDev *__restrict__ gcls, int nel, const int32_t *__restrict__ cls, const double *__restrict__ sig,
                  const double *__restrict__ eps, const double *__restrict__ epl, double *__restrict__ part /* [18][gridDim.x] */)
{
    double acc[18];
#pragma unroll
    for (int k = 0; k < 18; k++) acc[k] = 0.;
    for (int e = blockIdx.x * BLOCK + threadIdx.x; e < nel; e += gridDim.x * BLOCK) {
        const double v = gcls[cls[e]].vel;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            acc[k] = fma(sig[(size_t)k * nel + e], v, acc[k]);
            acc[6 + k] = fma(eps[(size_t)k * nel + e], v, acc[6 + k]);
            acc[12 + k] = fma(epl[(size_t)k * nel + e], v, acc[12 + k]);
        }
    }
    block_sums_to_partials(acc, part);
}

// out[k] = sum of row k of part[nrows][npart]; one block per row (launch with nrows blocks), fixed order
__global__ void __launch_bounds__(BLOCK) k_reduce_rows(const double *part, int nrows, int npart, double *out)
{
    __shared__ double sh[BLOCK / 64];
    const int k = blockIdx.x;
    if (k >= nrows) return;
    double v = 0.;
    for (int i = threadIdx.x; i < npart; i += BLOCK) v += part[(size_t)k * npart + i];
    const double t = block_sum(v, sh);
    if (threadIdx.x == 0) out[k] = t;
}

// calc_scf per element (model.py:1036-1054): hh value and multiplicity (0, 1 or 2 appends)
__global__ void __launch_bounds__(BLOCK)
k_scf_elements(const MatDev *__restrict__ gmat, int nmat, const ClassDev *__restrict__ gcls, int ncls, int lds_doubles, int nel,
               int e_off, const int32_t *__restrict__ conn, const int32_t *__restrict__ cls, const double2 *__restrict__ du2,
               const double *__restrict__ sig, const double *__restrict__ epl, const double *__restrict__ elstiff, const double *__restrict__ sld,
               double *__restrict__ hh_out, int32_t *__restrict__ mult_out, const double *__restrict__ kh_el = nullptr,
               unsigned skip_mask = 0u /* materials done by k_scf_row */)
{
    __shared__ MatDev smat[MAXMAT];
    stage_materials(smat, gmat, nmat);
    __syncthreads();
    int svc_mat;
    const double *sv, *dual;
    stage_svc(smat, nmat, dyn_lds, lds_doubles, svc_mat, sv, dual);
    __syncthreads();
    (void)ncls;
    double ld[6];
#pragma unroll
    for (int k = 0; k < 6; k++) ld[k] = sld[k];
    for (int e = blockIdx.x * BLOCK + threadIdx.x; e < nel; e += gridDim.x * BLOCK) {
        const ClassDev &c = gcls[cls[e]];
        if ((skip_mask >> c.mat) & 1u) continue;
        const MatDev &m = smat[c.mat];
        const size_t ge = (size_t)e + e_off;
        double de[6], D[21], ds[6];
        class_strain(c, du2, conn[ge * 4], conn[ge * 4 + 1], conn[ge * 4 + 2], conn[ge * 4 + 3], de);
#pragma unroll
        for (int k = 0; k < 21; k++) D[k] = elstiff[(size_t)k * nel + e];
        symv(D, de, ds);
        int mult = 0;
        double hh = 0.;
        if (m.kind != 0) {
            const double sref = (m.kind == 2 || m.kind == 6) ? princ_seq(m, ds) : m.kind == 5 ? barlat_seq(m, ds)
                                                                                             : hill_seq(m, ds);  // Stress(el.dsig()).seq(el.Mat) (model.py:1040)
            if (sref > 0.1) {
                double s[6], ep[6];
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    s[k] = sig[(size_t)k * nel + e];
                    ep[k] = epl[(size_t)k * nel + e];
                }
                double yf0;
                if (m.kind == 7) {  // work-hardening SVC: features with the plastic strain, flow stress with the point's modulus
                    const bool st = (c.mat == svc_mat);
                    const YfSvcWh yw(m, st ? sv : m.sv, st ? dual : m.dual, kh_el ? kh_el[e] : m.khard);
                    yf0 = yw.plain(s, ep);
                    if (yf0 < SPLIT_THRESHOLD) {
                        yf0 = yw.full_ld(s, ep, ld, nullptr);
                        hh = fmin(1., -yf0 / sref);
                        mult = 2;
                    } else {
                        hh = fmin(1., sqrt(1.5) * yw.sflow(ep) / sref);
                        mult = 1;
                    }
                } else if (kind_is_svc(m.kind)) {
                    const bool st = (c.mat == svc_mat);
                    const double *psv = st ? sv : m.sv, *pdu = st ? dual : m.dual;
                    yf0 = (m.kind == 3) ? YfSvc(m, psv, pdu).plain(s, ep) : YfSvc3(m, psv, pdu).plain(s, ep);
                    if (yf0 < SPLIT_THRESHOLD) {  // branch test on the decision function (model.py:1046-1048)
                        yf0 = (m.kind == 3) ? YfSvc(m, psv, pdu).full_ld(s, ep, ld, nullptr)
                                            : YfSvc3(m, psv, pdu).full_ld(s, ep, ld, nullptr);  // model.py:1049-1052
                        hh = fmin(1., -yf0 / sref);
                        mult = 2;  // appended twice (model.py:1054 and :1058)
                    } else {
                        hh = fmin(1., sqrt(1.5) * sflow_of(m, ep) / sref);
                        mult = 1;
                    }
                } else {
                    yf0 = (m.kind == 2 ? princ_seq(m, s) : m.kind == 5 ? barlat_seq(m, s) : hill_seq(m, s)) - sflow_of(m, ep);
                    if (yf0 < SPLIT_THRESHOLD) {
                        hh = fmin(1., -yf0 / sref);
                        mult = 2;
                    } else {
                        hh = fmin(1., sqrt(1.5) * sflow_of(m, ep) / sref);
                        mult = 1;
                    }
                }
            }
        }
        hh_out[e] = hh;
        mult_out[e] = mult;
    }
}

// calc_scf for the elements of the 6-feature SVC material `mat`, 16 lanes per element (YfSvcRow; ML_full_yf along the
// loading direction is a ray search, model.py:1049-1052) -- the thread-per-element form above took 4.5 ms per call on
// 16 384 elements, a quarter of a 50-sub-step corrector launch
template <bool INLDS>
__global__ void __launch_bounds__(512)
k_scf_row(const MatDev *__restrict__ gmat, int nmat, const ClassDev *__restrict__ gcls, int mat, int nel,
          int e_off, const int32_t *__restrict__ conn, const int32_t *__restrict__ cls, const double2 *__restrict__ du2,
          const double *__restrict__ sig, const double *__restrict__ epl, const double *__restrict__ elstiff, const double *__restrict__ sld,
          double *__restrict__ hh_out, int32_t *__restrict__ mult_out)
{
    __shared__ MatDev smat[MAXMAT];
    stage_materials(smat, gmat, nmat);
    __syncthreads();
    const int npad = INLDS ? stage_svc_wave(smat, mat, 1) : smat[mat].rowpad;
    __syncthreads();
    const MatDev &m = smat[mat];
    const int l16 = threadIdx.x & 15, rpb = blockDim.x >> 4;
    double ld[6];
#pragma unroll
    for (int k = 0; k < 6; k++) ld[k] = sld[k];
    for (int e = blockIdx.x * rpb + (threadIdx.x >> 4); e < nel; e += gridDim.x * rpb) {  // row-uniform
        const ClassDev &c = gcls[cls[e]];
        if (c.mat != mat) continue;
        const size_t ge = (size_t)e + e_off;
        double de[6], D[21], ds[6];
        class_strain(c, du2, conn[ge * 4], conn[ge * 4 + 1], conn[ge * 4 + 2], conn[ge * 4 + 3], de);
#pragma unroll
        for (int k = 0; k < 21; k++) D[k] = elstiff[(size_t)k * nel + e];
        symv(D, de, ds);
        int mult = 0;
        double hh = 0.;
        const double sref = hill_seq(m, ds);   // Stress(el.dsig()).seq(el.Mat) (model.py:1040)
        if (sref > 0.1) {
            double s[6], ep[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                s[k] = sig[(size_t)k * nel + e];
                ep[k] = epl[(size_t)k * nel + e];
            }
            const YfSvcRow<4, INLDS> yf(m, npad);
            double yf0 = yf.plain(s, ep);
            if (yf0 < SPLIT_THRESHOLD) {  // branch test on the decision function (model.py:1046-1048)
                yf0 = yf.full_ld(s, ep, ld, nullptr);  // model.py:1049-1052
                hh = fmin(1., -yf0 / sref);
                mult = 2;  // appended twice (model.py:1054 and :1058)
            } else {
                hh = fmin(1., sqrt(1.5) * sflow_of(m, ep) / sref);
                mult = 1;
            }
        }
        if (l16 == 0) {
            hh_out[e] = hh;
            mult_out[e] = mult;
        }
    }
}

// pass 0: sum(mult*hh), min(hh | mult>0), count ; pass 1: sum(mult*(hh-mean)^2)
__global__ void __launch_bounds__(BLOCK)
k_scf_reduce(int nel, const double *hh, const int32_t *mult, double mean, int pass,
             double *part /* [3][gridDim.x] */, const double *mean_dev = nullptr, int e_lo = 0, int e_hi = 0x7fffffff)
{
    __shared__ double sh[BLOCK / 64];
    __shared__ double shmin[BLOCK / 64];
    if (mean_dev) mean = *mean_dev;
    double s = 0., cnt = 0., mn = 1.e300;
    for (int e = blockIdx.x * BLOCK + threadIdx.x; e < nel; e += gridDim.x * BLOCK) {
        const int m = (e >= e_lo && e < e_hi) ? mult[e] : 0;  // strip: halo elements belong to the neighbour's statistics
        if (m == 0) continue;
        const double h = hh[e];
        if (pass == 0) {
            s += m * h;
            cnt += m;
            mn = fmin(mn, h);
        } else {
            s += m * (h - mean) * (h - mean);
        }
    }
    const double ts = block_sum(s, sh);
    const double tc = block_sum(cnt, sh);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_down(mn, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) shmin[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = shmin[0];
        for (int i = 1; i < BLOCK / 64; i++) t = fmin(t, shmin[i]);
        part[blockIdx.x] = ts;
        part[gridDim.x + blockIdx.x] = tc;
        part[2 * gridDim.x + blockIdx.x] = t;
    }
}

// final sums of the k_scf_reduce partials (one block, fixed order):
// pass 0 -> out[0..3] = sum, count, min, mean;  pass 1 -> out[4] = centred sum of squares
__global__ void __launch_bounds__(BLOCK) k_scf_finish(const double *part, int g, int pass, double *out)
{
    __shared__ double sh[BLOCK / 64];
    __shared__ double shmin[BLOCK / 64];
    double s = 0., cnt = 0., mn = 1.e300;
    for (int b = threadIdx.x; b < g; b += BLOCK) {
        s += part[b];
        cnt += part[g + b];
        mn = fmin(mn, part[2 * g + b]);
    }
    const double ts = block_sum(s, sh);
    const double tc = block_sum(cnt, sh);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mn = fmin(mn, __shfl_down(mn, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) shmin[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = shmin[0];
        for (int i = 1; i < BLOCK / 64; i++) t = fmin(t, shmin[i]);
        if (pass == 0) {
            out[0] = ts;
            out[1] = tc;
            out[2] = t;
            out[3] = tc > 0. ? ts / (double)(long long)(tc + 0.5) : 0.;
        } else {
            out[4] = ts;
        }
    }
}

// mean of the (all-reduced) calc_scf statistics: out[3] = out[0] / out[1]
__global__ void k_scf_mean(double *out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out[3] = out[1] > 0. ? out[0] / (double)(long long)(out[1] + 0.5) : 0.;
}


// ---------------------------------------------------------------------------------------------
// Work-hardening SVC, SEQUENTIAL CARRY (the reference's semantics): Material.khard is ONE mutable attribute that every
// calc_fgrad overwrites (material.py:808-814) and that the element loop of Model.solve hands from element to element in index
// order (model.py:1340-1359).  Element e therefore ENTERS its response() with the exit modulus of the last element before it
// (same material) whose call evaluated a gradient -- calls that stay elastic pass their entry value on -- or, if there is
// none, with the value the material object held when the sweep began.  Given the exit moduli and "touched" flags of a
// data-parallel sweep, these three kernels compute that entry value for every element (an exclusive prefix maximum of the
// touched element indices) and count the elements whose entry value differs from the one the sweep was run with; the host
// repeats the sweep until the count is zero (fixed point = the sequential loop's result, plfx.hip: sweep_wh_sequential).
struct WhCarry {
    double v[16];
};
__global__ void __launch_bounds__(BLOCK)
k_wh_fill(int nel, const ClassDev *__restrict__ gcls, const int32_t *__restrict__ cls, WhCarry w, double *__restrict__ kh)
{
    for (int e = blockIdx.x * BLOCK + threadIdx.x; e < nel; e += gridDim.x * BLOCK) kh[e] = w.v[gcls[cls[e]].mat & 15];
}

__global__ void __launch_bounds__(BLOCK)
k_wh_blockmax(int nel, int mat, const ClassDev *__restrict__ gcls, const int32_t *__restrict__ cls, const int32_t *__restrict__ touch,
              int32_t *__restrict__ bmax)
{
    __shared__ int sh[BLOCK / 64];
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    int key = -1;
    if (e < nel && gcls[cls[e]].mat == mat && touch[e]) key = e;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) key = max(key, __shfl_xor(key, off, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        int m = sh[0];
        for (int w = 1; w < BLOCK / 64; w++) m = max(m, sh[w]);
        bmax[blockIdx.x] = m;
    }
}

// exclusive prefix maximum over the block maxima (in place), total maximum -> last[0]; one workgroup
__global__ void __launch_bounds__(BLOCK)
k_wh_scan(int nblk, int32_t *__restrict__ bmax, int32_t *__restrict__ last)
{
    if (threadIdx.x != 0) return;
    int run = -1;
    for (int b = 0; b < nblk; b++) {
        const int v = bmax[b];
        bmax[b] = run;
        run = max(run, v);
    }
    last[0] = run;
}

__global__ void __launch_bounds__(BLOCK)
k_wh_entry(int nel, int mat, const ClassDev *__restrict__ gcls, const int32_t *__restrict__ cls, const int32_t *__restrict__ touch,
           const int32_t *__restrict__ bpre, const double *__restrict__ kh_out, double carry, const double *__restrict__ kh_in,
           double *__restrict__ kh_new, int32_t *__restrict__ nchanged)
{
    __shared__ int key[BLOCK];
    const int e = blockIdx.x * BLOCK + threadIdx.x;
    const bool mine = e < nel && gcls[cls[e]].mat == mat;
    key[threadIdx.x] = (mine && touch[e]) ? e : -1;
    __syncthreads();
    int lm = bpre[blockIdx.x];
    for (int t = 0; t < (int)threadIdx.x; t++) lm = max(lm, key[t]);   // (256 LDS reads per thread: the scan is not the cost of a sweep)
    int diff = 0;
    if (mine) {
        const double v = lm >= 0 ? kh_out[lm] : carry;
        kh_new[e] = v;
        diff = __double_as_longlong(v) != __double_as_longlong(kh_in[e]);
    }
    const unsigned long long m = __ballot(diff);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(nchanged, __popcll(m));
}

}  // namespace plfx

