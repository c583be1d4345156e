/*
 * plfx_oracle.h — CPU oracle for the pyLabFEA hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, line-faithful restatement of the reference algorithm
 * (pyLabFEA v4.4.2, /root/reference/src/pylabfea) used as the checker in tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * (pylabfea_amd/ + libplfx.so) never includes, links or calls anything in here.
 *
 * Parity status: PINNED — every function is checked against golden vectors dumped
 * from the imported reference by oracle/gen_golden.py (tests/test_oracle_golden.py).
 *
 * Conventions: Voigt order (11,22,33,23,13,12), engineering shear strains, IEEE FP64.
 * 6x6 matrices are row-major double[36].
 */
#ifndef PLFX_ORACLE_H
#define PLFX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { PLFO_ELASTIC = 0, PLFO_HILL6 = 1, PLFO_PRINC3 = 2, PLFO_SVC6 = 3, PLFO_TRESCA = 4, PLFO_BARLAT = 5,
       PLFO_SVC3 = 6, /* sdim=3 ML material: 2 features (seq_J2/scale - 1, polar angle/pi), material.py:2330-2333 */
       PLFO_SVC_WH = 7 /* 6-feature SVC + work-hardening features (material.py:2342-2346): ndof = 15, khard is mutable state */ };

typedef struct plfo_material {
    int kind;            /* PLFO_* */
    int sdim;            /* 3 or 6 (material.py:2520) */
    double E, nu;        /* isotropic constants (material.py:2440-2441) */
    double sy, khard;    /* material.py:2512-2513 */
    double hill[6];      /* material.py:2573 */
    double dp[3];        /* d0 of calc_seq: lhs or ones*drucker (material.py:640-645) */
    /* SVC (material.py:398-405, 765-807) */
    int nsv, ndof, dev_only;
    double gamma, intercept, scale_seq;
    const double *sv;    /* [nsv*ndof] row-major support vectors */
    const double *dual;  /* [nsv] dual coefficients */
    /* Barlat Yld2004-18p (material.py:2575-2591): 18 coefficients and exponent; calc_seq only */
    double barlat[18];
    double barlat_exp;
    double scale_wh;     /* PLFO_SVC_WH: scaling of the plastic-strain features (material.py:1165-1172, 2343) */
} plfo_material;

/* basic.py:304 sig_dev, :328 eps_eq */
void plfo_sig_dev(const double sig[6], double out[6]);
double plfo_eps_eq(const double eps[6]);

/* basic.py:107-179 sig_princ: principal stresses in the reference's axis-tracking order.  Exact for
 * plane states (s23 = s13 = 0); for general 3-d states the reference's order depends on LAPACK's
 * dgeev output order and only the natural rule (axis i -> eigenvector with the largest |component i|)
 * is restated. */
void plfo_sig_princ(const double sig[6], double sp[3]);
/* material.py:576 calc_seq (Hill-6p/J2 on Voigt, Hill-3p/J2 on principal stresses, Tresca, Barlat) */
double plfo_calc_seq(const plfo_material *m, const double sig[6]);
/* material.py:974 get_sflow */
double plfo_get_sflow(const plfo_material *m, const double epl[6]);
/* material.py:348 calc_yf : analytic or SVC decision function */
double plfo_calc_yf(const plfo_material *m, const double sig[6], const double epl[6]);
/* material.py:704 calc_fgrad */
void plfo_calc_fgrad(const plfo_material *m, const double sig[6], double a[6]);
/* material.py:414 ML_full_yf (ld=None path); *status: 0 ok, 1 bracket failure, 2 no convergence */
double plfo_ML_full_yf(const plfo_material *m, const double sig[6], const double epl[6], int *status);
/* material.py:414 ML_full_yf with a loading direction ld (:454-462), as calc_scf calls it (model.py:1049-1053) */
double plfo_ML_full_yf_ld(const plfo_material *m, const double sig[6], const double epl[6], const double *ld, int *status);
void plfo_full_yf_ld_batch(const plfo_material *m, int n, const double *sig, const double *epl, const double *ld,
                           double *out);
/* material.py:1009 epl_dot, :1057 C_tan */
void plfo_epl_dot(const plfo_material *m, const double sig[6], const double epl[6],
                  const double Cel[36], const double deps[6], double pdot[6]);
void plfo_C_tan(const plfo_material *m, const double sig[6], const double Cel[36], double Ct[36]);
/* material.py:207 response.  Returns msg['nsteps'] (= last loop index). */
int plfo_response(const plfo_material *m, const double sig[6], const double epl[6],
                  const double deps[6], const double CV[36],
                  double *fy, double sig_out[6], double depl[6], double Ct[36]);

/* batched drivers (OpenMP over points); arrays are AoS [n*6] / [n*36] */
void plfo_seq_batch(const plfo_material *m, int n, const double *sig, double *seq);
void plfo_fgrad_batch(const plfo_material *m, int n, const double *sig, double *a);
void plfo_yf_batch(const plfo_material *m, int n, const double *sig, const double *epl, double *yf);
void plfo_full_yf_batch(const plfo_material *m, int n, const double *sig, const double *epl,
                        double *yf, int *status);
void plfo_response_batch(const plfo_material *mats, int n, const int *mat_id,
                         const double *sig, const double *epl, const double *deps,
                         const double *CV /* [nmat*36] element CV per material */,
                         double *fy, double *sig_out, double *depl, double *ct, int *nsteps,
                         int nthreads);

/* model.py:439 calc_Bmat (2-d, linear shape functions) ; B is 6x8 row-major */
void plfo_calc_Bmat(double lx, double ly, double x, double y, int planestress,
                    const double CV[36], double E, double nu, double B[48]);
/* model.py:262-348 Gauss points + model.py:365 calc_Kel: Kel = Jac*wght*sum_gp B^T D B */
void plfo_calc_Kel(double lx, double ly, double thick, int planestress, const double CV[36],
                   double E, double nu, const double D[36], double Kel[64]);
/* model.py:387 deps / :400 eps_t : (sum_gp B) u_e */
void plfo_strain(double lx, double ly, int planestress, const double CV[36], double E, double nu,
                 const double ue[8], double eps[6]);

/* batched element routines over a mesh (OpenMP): per element class arrays lxy[nel*2], mat_id[nel];
 * CV/E/nu per material [nmat*36], [nmat], [nmat]. */
void plfo_kel_batch(int nel, const double *lxy, const int *mat_id, double thick, int planestress,
                    const double *CV, const double *E, const double *nu, const double *D /* [nel*36] */,
                    double *Kel /* [nel*64] */);
void plfo_strain_batch(int nel, const int *conn /* [nel*4] */, const double *lxy, const int *mat_id,
                       int planestress, const double *CV, const double *E, const double *nu,
                       const double *u /* [ndof] */, double *eps /* [nel*6] */);

/* Kred + np.linalg.solve (model.py:1028-1033, 1291) as Jacobi-PCG on the CSR matrix, free DOFs only (OpenMP rows) */
int plfo_pcg_csr(int n, const int *indptr, const int *indices, const double *data, const double *b,
                 const unsigned char *free_mask, double *x, double rtol, int maxit, int nthreads, double *relres);

/* PLFO_SVC_WH: the reference's Material.khard is ONE mutable attribute that every calc_fgrad call overwrites
 * (material.py:808-814) and get_sflow / epl_dot / C_tan read.  plfo_calc_fgrad_wh is calc_fgrad(sig, epl) for a single point: it
 * writes the new value into m->khard THROUGH the const pointer (callers pass a private copy of the material, exactly like the
 * Python object is mutated); *kh_raw receives the unclipped value.  plfo_response_wh_batch: response() per point on a private
 * copy that starts with khard_in[i] (NULL: m->khard) and leaves khard_out[i]; sequential != 0 instead carries ONE copy through
 * the points in index order -- what a loop of response() calls over the elements does in Model.solve. */
void plfo_calc_fgrad_wh(const plfo_material *m, const double sig[6], const double epl[6], double a[6], double *kh_raw);
void plfo_fgrad_wh_batch(const plfo_material *m, int n, const double *sig, const double *epl, double *a, double *kh_raw);
void plfo_yf_wh_batch(const plfo_material *m, int n, const double *sig, const double *epl, double *yf);
void plfo_full_yf_wh_batch(const plfo_material *m, int n, const double *sig, const double *epl, const double *khard, double *yf);
void plfo_response_wh_batch(const plfo_material *mats, int n, const int *mat_id, const double *sig, const double *epl,
                            const double *deps, const double *CV, const double *khard_in, double *fy, double *sig_out,
                            double *depl, double *ct, int *nsteps, double *khard_out, int sequential, int nthreads,
                            double *kh_point /* [n] or NULL: Material.khard right after each point's call */);

/* scipy 1.15.3 optimize.brentq (Brent 1973) on a scalar callback */
typedef double (*plfo_fn)(double x, void *ctx);
double plfo_brentq(plfo_fn f, void *ctx, double xa, double xb, double xtol, double rtol,
                   int maxiter, int *converged);

#ifdef __cplusplus
}
#endif
#endif
