/*
 * plfx.h — C-ABI of libplfx.so, the MI355X (gfx950) engine for pyLabFEA's hot path.
 *
 * The reference (pyLabFEA v4.4.2, pure Python) has no FFI: its boundary is the Python object
 * API Model.solve() -> Element.* -> Material.response().  Every entry point below replaces the
 * reference function cited next to it (paths relative to /root/reference/src/pylabfea) and is
 * what a ctypes binding inside the reference would call (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; all host arrays are caller-owned, C-contiguous,
 *     `double` / `int32_t`, and are not retained after the call returns (copied to HBM).
 *   - Voigt order (11,22,33,23,13,12), engineering shear strains; 6x6 matrices row-major [36].
 *   - host-side per-point arrays are AoS ([n*6], [n*36]) exactly like the NumPy arrays of the
 *     reference; the SoA layout in HBM is private to the library (DESIGN.md "Data layout").
 *   - every function returns 0 on success, <0 on error (plfx_last_error gives the text);
 *     soft failures of the reference (warnings.warn) are returned as counters/flags.
 *   - one context per host thread; a context is bound to one GPU and one HIP stream.
 */
#ifndef PLFX_H
#define PLFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plfx_ctx plfx_ctx;

/* yield-function kinds */
enum {
    PLFX_ELASTIC = 0, /* Material.sy is None: no response() call (model.py:1341) */
    PLFX_HILL6 = 1,   /* analytic Hill-6p on the full Voigt stress, J2 = all ones (material.py:650-661) */
    PLFX_PRINC3 = 2,  /* sdim=3: Hill-3p/J2 on principal stresses in sig_princ's axis-tracking order
                         (material.py:662-670, basic.py:107-179); exact for plane stress states */
    PLFX_SVC6 = 3,    /* RBF-SVC yield function on 6 stress features (material.py:398-405, 765-807) */
    PLFX_TRESCA = 4,  /* Tresca equivalent stress (material.py:630-632): calc_seq only, the reference has
                         no flow rule for it (calc_fgrad raises, material.py:824) */
    PLFX_BARLAT = 5,  /* Barlat Yld2004-18p equivalent stress (material.py:678-702): calc_seq only (:822) unless
                         plfx_material.barlat_normal is set (native normal, extension) */
    PLFX_SVC3 = 6,    /* sdim=3 ML material: RBF-SVC on 2 features (seq_J2/scale - 1, polar angle/pi) of the
                         principal stresses, gradient through the Jacobian (material.py:779-807, 2331-2333) */
    PLFX_SVC_WH = 7   /* RBF-SVC with work-hardening features (material.py:2342-2346): 15 features = 6 stress features, the
                         plastic strain / scale_wh (6), accumulated strain, max. stress / scale_seq, flag (the last three are
                         zero on the path); the hardening modulus is read off the gradient (material.py:808-814) */
};

/* error codes */
enum {
    PLFX_OK = 0,
    PLFX_ERR_HIP = -1,      /* HIP runtime failure (no device, OOM, launch error) */
    PLFX_ERR_ARG = -2,      /* invalid argument */
    PLFX_ERR_STATE = -3,    /* call order violated (e.g. solve before set_mesh) */
    PLFX_ERR_UNSUPPORTED = -4
};

/* One material as seen by an element: mirrors Material.elasticity/plasticity (material.py:2401-2594)
 * plus the trained SVC parameters (svm_yf.support_vectors_, dual_coef_, intercept_, gam_yf, scale_seq). */
typedef struct plfx_material {
    int32_t kind;      /* PLFX_* */
    int32_t sdim;      /* 6, or 3 for PLFX_PRINC3 */
    double CV[36];     /* ELEMENT elastic matrix, i.e. after the plane-stress/strain choice of
                          Element.__init__ (model.py:272-303); must be symmetric */
    double E, nu;      /* used by the plane-stress B-matrix row (model.py:498-501) */
    double sy, khard;  /* material.py:2512-2513 */
    double hill[6];    /* material.py:2573 */
    double drucker;    /* material.py:2514 (d0 = ones*drucker) */
    int32_t nsv;       /* SVC: number of support vectors */
    int32_t nfeat;     /* SVC: features per support vector (6, or 2 for PLFX_SVC3) */
    int32_t dev_only;  /* SVC: deviatoric features (material.py:2336) */
    int32_t _pad;
    double gamma, intercept, scale_seq;
    const double *sv;   /* [nsv*nfeat] row-major */
    const double *dual; /* [nsv] */
    double barlat[18];  /* Yld2004-18p coefficients c'_12.. (material.py:2578-2591) */
    double barlat_exp;  /* exponent a */
    int32_t barlat_normal; /* PLFX_BARLAT only.  0: like the reference, the material has an equivalent stress but no flow rule
                            * (calc_fgrad raises, material.py:822-825) -- plfx_fgrad_batch / plfx_response_batch / plfx_sweep
                            * refuse it.  1 (EXTENSION, north star "Barlat ... with their normals"): the analytic normal
                            * d seq / d sigma of Yld2004-18p through the eigen-decompositions of the two transformed deviators,
                            * associated flow rule like the Hill materials.  NO REFERENCE VALUE EXISTS for this branch: the
                            * reference raises for Barlat normals (material.py:822-825), so its parity is "unpinned" in the sense
                            * of DESIGN.md 4 -- it is held by central finite differences of the pinned calc_seqB, Euler's theorem
                            * (a . sigma = seq) and the identity "unit coefficients, a = 2" == J2 (tests/test_barlat_normal.py). */
    int32_t _pad2;
    double scale_wh;    /* PLFX_SVC_WH: scaling of the plastic-strain features (Material.scale_wh, material.py:1165-1172) */
} plfx_material;

/* ---------------------------------------------------------------- context */
int plfx_create(int device, plfx_ctx **out);
void plfx_destroy(plfx_ctx *ctx);
const char *plfx_last_error(plfx_ctx *ctx);
const char *plfx_version(void);
/* number of GPUs visible to the process (0 when there is none or the HIP runtime cannot start); needs no context */
int plfx_device_count(void);
/* name[<=len], number of CUs, bytes of HBM */
int plfx_device_info(plfx_ctx *ctx, char *name, int len, int *cus, int64_t *hbm_bytes);
/* HIP stream the library launches on (for hipEvent timing by the caller) */
void *plfx_stream(plfx_ctx *ctx);
int plfx_sync(plfx_ctx *ctx);

/* ---------------------------------------------------------------- materials */
/* replaces the parameter set-up consumed by Material.response (material.py:2401-2594) */
int plfx_set_materials(plfx_ctx *ctx, int nmat, const plfx_material *mats);

/* ---------------------------------------------------------------- batched point evaluations
 * Material.calc_seq (material.py:576), calc_fgrad (:704), calc_yf (:348), ML_full_yf (:414),
 * response (:207) on N points; also the kernel-level parity entry points. */
int plfx_seq_batch(plfx_ctx *ctx, int mat, int n, const double *sig, double *seq);
int plfx_fgrad_batch(plfx_ctx *ctx, int mat, int n, const double *sig, double *fgrad);
/* calc_fgrad(sig, seq=seq) of an analytic Hill material (PLFX_HILL6 / PLFX_PRINC3), material.py:834-847: the deviator of the
 * Voigt components over 2 seq[i] with the equivalent stress handed in -- the reference's `seq` argument, and the form its
 * point function takes for a (6,) stress of a principal-stress (sdim = 3) material: seq in sig_princ's order, deviator of
 * the Voigt normals, no shear rows (plfx_fgrad_batch returns the principal-space normal epl_dot / C_tan use, :1044-1047). */
int plfx_fgrad_seq_batch(plfx_ctx *ctx, int mat, int n, const double *sig, const double *seq, double *fgrad);
int plfx_yf_batch(plfx_ctx *ctx, int mat, int n, const double *sig, const double *epl, double *yf);
/* ld: NULL or one loading direction [6] shared by all points (model.py:1052); status[n]: 0 ok,
 * 1 bracket failure, 2 root not accepted (both fall back to seq-0.85*sflow as the reference) */
int plfx_full_yf_batch(plfx_ctx *ctx, int mat, int n, const double *sig, const double *epl,
                       const double *ld, double *yf, int32_t *status);
int plfx_response_batch(plfx_ctx *ctx, int n, const int32_t *mat_id, const double *sig,
                        const double *epl, const double *deps, double *fy, double *sig_out,
                        double *depl, double *ct /* [n*36] */, int32_t *nsteps);
/* The `maxit` argument of Material.response (material.py:207, 288-291: an increment whose trial step ends outside the yield
 * locus is sub-divided into maxit sub-steps; nsteps = maxit - 1 then) for the point entries plfx_response_batch(_kh) of this
 * context; default 50, the reference's default and what the load-step loop always uses (model.py:1346 passes none). */
int plfx_set_response_maxit(plfx_ctx *ctx, int maxit);
/* basic.sig_princ (basic.py:107-179) on n Voigt stresses, computed on the HOST by the routine the PRINC3 / SVC3 kernels run on
 * states with out-of-plane shear: np.linalg.eig's eigenpair ORDER (LAPACK dgeev replayed for symmetric 3 x 3 matrices,
 * csrc/plfx_lapack3.hpp) and the reference's row-argmax re-ordering.  No context, no GPU.  sp[n*3].  plfx_eig3_host returns the
 * eigenvalues w[n*3] in dgeev's order and (V != NULL) the unit eigenvectors V[n*9], V[i*3+k] = component i of eigenvector k.
 * Return 1 if a matrix did not converge / gave a complex pair (numpy would, too). */
int plfx_sig_princ_host(int n, const double *sig, double *sp);
int plfx_eig3_host(int n, const double *sig, double *w, double *V);
/* Work-hardening-aware SVC materials (PLFX_SVC_WH).  The reference keeps the hardening modulus in ONE mutable attribute of
 * the Material object: every calc_fgrad call overwrites it (khard = max(0, -sum dK/dx[wh] scale_seq/scale_wh),
 * material.py:808-814), every get_sflow / epl_dot / C_tan reads it, and it is carried from call to call.  Here it is an
 * explicit input and output per point: khard_in[n] (NULL: the material's khard) is what Material.khard holds when
 * response() is entered, khard_out[n] what it holds on return.
 * Inside the load-step loop (plfx_sweep / plfx_load_step) the reference hands ONE value from element to element in index order
 * (model.py:1340-1359): reproduced exactly by default, see plfx_set_wh_mode below (round 4; rounds 2-3 carried one modulus
 * per material point, which remains available and is the only form on several GPUs). */
int plfx_response_batch_kh(plfx_ctx *ctx, int n, const int32_t *mat_id, const double *sig, const double *epl,
                           const double *deps, const double *khard_in, double *fy, double *sig_out, double *depl,
                           double *ct, int32_t *nsteps, double *khard_out);
/* calc_fgrad(sig, epl) of a PLFX_SVC_WH material on n points: gradient w.r.t. the stress and, per point, the raw value
 * -sum_k dK/dx[6+k] * scale_seq / scale_wh whose mean over the points, clipped at 0, the reference stores in khard */
int plfx_fgrad_batch_wh(plfx_ctx *ctx, int mat, int n, const double *sig, const double *epl, double *fgrad, double *khard_raw);

/* Index products of Model.mesh for the reference's structured NX x NY grid, computed on the host (no context, no GPU):
 * conn[NX*NY*4] = [n1, n1+1, n1+NnodeY, n1+NnodeY+1] with n1 = (ih / NY) * NnodeY + ih % NY for element ih = j*NY + k
 * (model.py:935-948); node sets noleft (j = 0), noright (j = NX), nobot (k = 0), notop (k = NY) in ascending node order
 * (model.py:897-911), each [NY+1] or [NX+1] long.  Any output pointer may be NULL.  Integer work: bit-exact. */
int plfx_gen_structured(int NX, int NY, int32_t *conn, int32_t *noleft, int32_t *noright, int32_t *nobot, int32_t *notop);

/* ---------------------------------------------------------------- mesh (Model.mesh products, model.py:758-952)
 * conn[nel*4]: Q4 connectivity in the reference's node order; mat_id[nel]; lxy[nel*2] element
 * sizes (Lelx, Lely).  Elements are rectangles aligned with the axes (model.py:262).
 * The element range [el_begin, el_end) is the part owned by this rank (x-strip shard);
 * pass 0, nel for a single GPU. */
int plfx_set_mesh(plfx_ctx *ctx, int nel, int nnode, const int32_t *conn, const int32_t *mat_id,
                  const double *lxy, double thick, int planestress, int el_begin, int el_end);
/* The same for the structured grids Model.mesh produces (model.py:758-952): node j * (NY + 1) + k, element j * NY + k,
 * connectivity [n1, n1 + 1, n1 + NY + 1, n1 + NY + 2] (:893, :935-948) written by the library; dx_col[NX] = element width of every
 * column (laminate sections, :847), dy the element height; material numbers per column (mat_col[NX]) or per element
 * (mat_el[NX * NY], the `elmts` form of Model.mesh); exactly one of the two non-NULL. */
int plfx_set_mesh_structured(plfx_ctx *ctx, int NX, int NY, const int32_t *mat_col, const int32_t *mat_el, const double *dx_col,
                             double dy, double thick, int planestress, int el_begin, int el_end);
/* Host-only self-test (no GPU needed): the closed-form block-ELL pattern that plfx_set_mesh / plfx_set_grid write for the
 * reference's structured numbering equals the one derived generically from the connectivity (slots, gather codes, order).
 * 0 = identical. */
int plfx_pattern_selftest(int nx, int ny);
/* Declare that the mesh is the reference's structured NX x NY grid (node j*(NY+1)+k, element j*NY+k,
 * model.py:893, 935); verified against conn.  Enables the geometric-multigrid preconditioner of
 * plfx_solve when all elements have one shape and NX, NY halve down to a small grid. */
int plfx_set_grid(plfx_ctx *ctx, int nx, int ny);
/* preconditioner of plfx_solve: kind 0 = Jacobi, 1 = multigrid V(nu,nu) with damped-Jacobi smoothing
 * (falls back to Jacobi when no hierarchy exists); omega <= 0 / nu <= 0 keep the defaults (0.65, 2) */
int plfx_set_precond(plfx_ctx *ctx, int kind, double omega, int nu);
int plfx_precond_info(plfx_ctx *ctx, int *kind_in_use, int *levels);
/* z = B r: one application of the preconditioner of plfx_solve (the multigrid V-cycle) to a host vector [ndof], for tests of
 * its symmetry / of what it does to a given field (single GPU, after plfx_assemble + plfx_apply_bc). */
int plfx_precond_apply(plfx_ctx *ctx, const double *r, double *z);
/* measurement hook (bench.py: `vcycle`): `reps` applications of the multigrid preconditioner back to back on the current
 * residual vector, one pair of HIP events around them -> microseconds per V-cycle; us_coarse (may be NULL): the part of a cycle
 * below the fine level (transfers to / from level 1, the launch-latency-bound levels, the single-workgroup tail).  Only scratch
 * vectors are written.  Single-GPU hierarchies (no strip). */
int plfx_precond_bench(plfx_ctx *ctx, int reps, double *us_per_cycle, double *us_coarse);
/* number of plfx_solve calls so far that PCG could not finish: a direction of negative curvature was met (the tangents of
 * Material.response are not always positive semi-definite, material.py:324-338) and the indefinite-system solver completed
 * the solve from the last iterate (right-preconditioned GMRES by default; PLFX_INDEFINITE_SOLVER=surrogate / minres:
 * preconditioned MINRES, see plfx_indefinite_info) -- the reference's LU does not need definiteness either -- or multigrid-PCG did not
 * converge within 300 iterations and Jacobi-PCG took over */
int plfx_solve_fallbacks(plfx_ctx *ctx, int64_t *count);
/* The solves with an indefinite tangent stiffness among them (p.Kp <= 0 met by PCG): how many there were, how many were
 * completed by MINRES with the V-cycle of the SPD surrogate operator (PLFX_INDEFINITE_SOLVER=surrogate: every indefinite
 * element matrix -- Kel is PSD iff the 3 x 3 matrix of its stiffness generators is -- shifted by its most negative
 * eigenvalue; uniform structured grids, one GPU or a replicated solve), how many by GMRES (the default; strips; MINRES not
 * converged), the number of surrogate hierarchies built (one per operator that needed it) and the elements shifted in the
 * last one.  Any pointer may be NULL. */
int plfx_indefinite_info(plfx_ctx *ctx, int64_t *solves, int64_t *by_minres_surrogate, int64_t *by_gmres,
                         int64_t *surrogates_built, int64_t *elements_shifted);
/* PLFX_INDEFINITE_SOLVER=sqmr (round 5, not the default): SQMR (simplified QMR for symmetric indefinite systems, Freund &
 * Nachtigal 1994) completes such a solve from PCG's last iterate -- the recurrences of preconditioned CG without its positivity
 * requirements plus a quasi-minimal-residual smoothing of the iterates, preconditioned by the V-cycle of the operator as it is
 * (symmetric, not necessarily definite); no Krylov basis.  GMRES continues from SQMR's iterate on a breakdown or when the true
 * residual stalls.  *by_sqmr: solves SQMR completed on its own. */
int plfx_sqmr_info(plfx_ctx *ctx, int64_t *by_sqmr);
/* Form of the stiffness operator in plfx_solve / plfx_update_state / plfx_apply_bc: kind 1 (default) applies
 * K matrix-free from the element stiffness generators (Element.calc_Kel never materialised, Model.setupK reduced to
 * the diagonal) wherever plfx_set_grid found a structured grid with one element shape; kind 0 always assembles the
 * block-ELL matrix (Model.setupK, model.py:954-977).  plfx_get_csr assembles on demand in both cases.
 * Environment override of the default: PLFX_MATFREE=0|1. */
int plfx_set_operator(plfx_ctx *ctx, int kind);
/* y = K x over all DOFs with the current operator (the product of model.py:1384, `K @ du`); host arrays [ndof] */
int plfx_matvec(plfx_ctx *ctx, const double *x, double *y);
int plfx_operator_info(plfx_ctx *ctx, int *matrix_free, int *levels_matrix_free);
/* Unchanged inputs are not recomputed (environment PLFX_REUSE=0 switches it off): plfx_assemble returns at once when no
 * sweep reported a changed tangent and nothing was written into the tangents since the last assembly (Model.setupK of an
 * unchanged model, model.py:1333); plfx_apply_bc_plan with the values of the previous call on the same operator keeps the
 * right-hand side; plfx_solve(warm=1) of the system the previous converged solve solved returns that solution with 0
 * iterations (the reference repeats the last solve of a load step as predictor and first stiffness iteration of the next,
 * model.py:1291/1335).  Counters of the three cases since plfx_create. */
int plfx_reuse_info(plfx_ctx *ctx, int *assemblies, int *bc_applications, int *solves);
/* Initial guess of a warm-started solve from the last two solutions (environment PLFX_PREDICT=0 at plfx_create switches it off):
 * with x the previous solution, d its difference to the one before and alpha in [0, 1] of smallest residual
 * | P (b - K (x + alpha d)) |, plfx_solve(warm=1) returns x + alpha d when -- and only when -- it satisfies the tolerance as it is
 * (0 iterations); otherwise the solve iterates from x exactly as with PLFX_PREDICT=0 (bit-identical).  Tried on multigrid-PCG
 * solves of meshes of >= 16384 nodes (strips: of the whole grid; the sums are all-reduced).  d is taken to the last solution that
 * differed from x (repeated solves of one system keep the history).  x itself is never rescaled.
 * applied (accepted as the solution) / skipped (alpha < 0.01) / rejected (failed the tolerance test) since plfx_create. */
int plfx_predict_info(plfx_ctx *ctx, int64_t *applied, int64_t *skipped, int64_t *rejected);
/* Sweeps since plfx_create and the number of element tangents they rewrote (model.py:1346-1355: a tangent is stored, and
 * Kel refreshed, only where it changed by more than 1e-3) -- whole mesh in sharded runs.  A sweep moves 412 B per element
 * plus 216 B per rewritten tangent. */
int plfx_sweep_info(plfx_ctx *ctx, int64_t *sweeps, int64_t *tangents_rewritten);
/* Which kernels run the 6-feature SVC materials (Material.response with an ML yield function, material.py:398-405 evaluates
 * any trained svm_yf): bit k of *row_materials = material k runs with 16 lanes per element / point (one launch per material
 * and sweep phase), its support-vector tables staged in LDS when they fit the 160 KB of a CU (up to ~2200 vectors) and read
 * from device memory otherwise -- no limit on the number of 6-feature SVC materials or of their support vectors;
 * bit k of *thread_materials = material k runs one thread per element / point (only when the row kernels are switched off
 * by PLFX_SVC_POLY / PLFX_SVC_WAVE).  Launch counters of the sweep kernels of either form since plfx_create. */
int plfx_svc_info(plfx_ctx *ctx, int *row_materials, int *thread_materials, int64_t *row_launches, int64_t *thread_launches);
/* B matrices of element e at its 4 Gauss points, [4*6*8] (Element.calc_Bmat, model.py:439) */
int plfx_get_bmat(plfx_ctx *ctx, int e, double *B);
/* element stiffness of element e from its current tangent (Element.calc_Kel, model.py:365), [64] */
int plfx_get_kel(plfx_ctx *ctx, int e, double *Kel);

/* ---------------------------------------------------------------- state (el.sig/eps/epl/elstiff, Model.u/f/du)
 * which: 0 sig, 1 eps, 2 epl, 3 res_sig, 4 res_depl  -> [nel_owned*6];  5 elstiff -> [nel_owned*36];
 *        6 u, 7 f, 8 du -> [ndof];  9 fyn (fy/sflow of the last sweep) -> [nel_owned];
 *        10 max_steps (stat_nlin) -> [nel_owned] as double;  11 hardening modulus of every material point (PLFX_SVC_WH
 *        materials: carried from sweep to sweep; others: the material's khard) -> [nel_owned] */
int plfx_state_get(plfx_ctx *ctx, int which, double *out);
int plfx_state_set(plfx_ctx *ctx, int which, const double *in);
/* solve() first-call initialisation (model.py:1212-1234): zero u,f,sig,eps,epl; elstiff = CV */
int plfx_state_reset(plfx_ctx *ctx);
/* gather entries of u (which=6), f (7) or du (8) at idx[n] */
int plfx_gather(plfx_ctx *ctx, int which, int n, const int32_t *idx, double *out);

/* ---------------------------------------------------------------- assembly (Model.setupK, model.py:954-977) */
int plfx_assemble(plfx_ctx *ctx);
/* export the assembled matrix as CSR (scalar rows, sorted columns).  Call with NULL arrays to get nnz. */
int plfx_get_csr(plfx_ctx *ctx, int64_t *nnz, int32_t *rowptr, int32_t *colidx, double *val);

/* ---------------------------------------------------------------- boundary conditions + solve
 * calc_BC (model.py:1070-1206) reduced to data: presc_idx[n] DOFs with displacement BC (ascending or
 * not), du_presc[n] the value written to du (first application), w[n] the multiplicity-weighted
 * value that enters the right-hand side (the reference re-applies a DOF shared by two edges),
 * fext[ndof] consistent nodal forces (or NULL).  Builds rhs = P (fext - K w) and the free mask. */
int plfx_apply_bc(plfx_ctx *ctx, int n, const int32_t *presc_idx, const double *du_presc,
                  const double *w, const double *fext);
/* calc_BC's index structure registered once (it only depends on the BC flags, model.py:1070-1206): nseg segments of
 * prescribed DOFs in the reference's order (left, bottom, right, top, node set; x before y), seg_len[nseg] entries each,
 * idx = their DOF numbers concatenated.  plfx_apply_bc_plan then takes ONE value per segment and forms, exactly like
 * plfx_apply_bc fed by the host: the ascending unique DOF set, the value written to du (first occurrence) and the
 * multiplicity-weighted right-hand-side value (a DOF shared by two edges is applied twice, :1115-1122).
 * inconsistent_entry: first entry whose value differs from the first one on its DOF (the reference's warning), -1 if none. */
int plfx_set_bc_plan(plfx_ctx *ctx, int nseg, const int32_t *seg_len, const int32_t *idx);
int plfx_apply_bc_plan(plfx_ctx *ctx, const double *seg_val, const double *fext, int *inconsistent_entry);
/* Kred + np.linalg.solve (model.py:1028-1033, 1291, 1335) as an iterative solve on the free DOFs: PCG with the multigrid
 * V-cycle (uniform structured grids) or the Jacobi scaling as preconditioner, stopped at |P(b - K du)| <= rtol |P b|.
 * Systems PCG cannot finish are completed from its last iterate: tangent stiffness that is not positive definite (a
 * direction with p.Kp <= 0 was met) by right-preconditioned GMRES(400), stalled multigrid (300 iterations) by Jacobi-PCG;
 * plfx_solve_fallbacks counts them.  warm != 0 starts from the previous du on the free DOFs.  Result in du (state 8);
 * returns 1 when maxit iterations did not reach rtol (iters / relres report what was reached). */
int plfx_solve(plfx_ctx *ctx, double rtol, int maxit, int warm, int *iters, double *relres);

/* ---------------------------------------------------------------- non-linear driver pieces */
/* material sweep (model.py:1340-1359) over owned elements with the current du:
 * response + |elstiff - Ct|_F > 1e-3 test + tangent refresh (averaging when nit >= 15).
 * changed: any tangent updated; conv: all fy/sflow <= yf_tolerance*1.0001 */
int plfx_sweep(plfx_ctx *ctx, int nit, int *changed, int *conv);
/* Work-hardening SVC materials (PLFX_SVC_WH) inside the load-step loop.  The reference keeps the hardening modulus in ONE
 * mutable attribute of the Material object: every calc_fgrad overwrites it (material.py:808-814), get_sflow / epl_dot / C_tan
 * read it, and the element loop of Model.solve (model.py:1340-1359) hands it from element to element in index order.
 *   sequential = 1 (default; one GPU, no shard): exactly that.  plfx_sweep resolves the chain as a fixed point -- sweep with
 *       guessed entry moduli, derive the entry moduli that sweep implies (exit modulus of the last element before e, same
 *       material, whose call evaluated a gradient; else the value the material held when the sweep began), repeat while any
 *       entry changed; tangents / generators are restored before every repetition.  plfx_wh_info counts sweeps and passes.
 *       State 11 then returns the EXIT modulus of every element's last call; plfx_wh_carry reads / sets the value a material
 *       object holds now (what Material.khard is after / before Model.solve in the reference).
 *   sequential = 0: one modulus per material point, carried from sweep to sweep (state 11) -- the data-parallel contract of
 *       rounds 2-3, the only form on several GPUs (the chain would cross every shard); deviates from the reference by ~1e-4
 *       on its 4 x 4 trace (tests/test_workhard_svc.py). */
int plfx_set_wh_mode(plfx_ctx *ctx, int sequential);
int plfx_wh_info(plfx_ctx *ctx, int *sequential_in_use, int64_t *sweeps, int64_t *passes,
                 int64_t *unresolved /* sweeps whose chain had not settled after 64 passes (accepted as they were; never observed) */);
int plfx_wh_carry(plfx_ctx *ctx, int mat, const double *set, double *get);
/* calc_scf statistics (model.py:1036-1067): sum of entries, sum of squares about the mean, min, count.
 * sld[6] loading direction for SVC materials. */
int plfx_scf_stats(plfx_ctx *ctx, const double *sld, double *sum, double *sumsq_c, double *minv,
                   int64_t *count, double mean_in, int pass);
/* both passes of calc_scf's statistics in one call (single GPU: the mean of pass 0 stays on the device) */
int plfx_scf_all(plfx_ctx *ctx, const double *sld, int64_t *count, double *minv, double *sum, double *sumsq_c);
/* end-of-load-step update (model.py:1383-1392): u += du, f += K du, sig/epl/eps update */
int plfx_update_state(plfx_ctx *ctx);
/* End of a load step in one call and one host synchronisation: plfx_update_state, then u and f at the registered DOFs
 * (the boundary nodes calc_global averages, model.py:1452-1471) and the 18 element sums of plfx_global_sums. */
int plfx_set_finish_set(plfx_ctx *ctx, int n, const int32_t *idx);
int plfx_finish_step(plfx_ctx *ctx, double *u_at /* [n] */, double *f_at /* [n] */, double *sums18);
/* calc_global element sums (model.py:1500-1511): out[18] = sum(sig*Vel), sum(eps*Vel), sum(epl*Vel) */
int plfx_global_sums(plfx_ctx *ctx, double *out18);

/* ---------------------------------------------------------------- one load step of Model.solve (model.py:1262-1392)
 * The whole body of the reference's load-step loop as ONE call: elastic predictor with the stiffness of the previous
 * step (:1290-1291), load-step scaling calc_scf while il < 10 (:1296), the stiffness iterations
 * [halving of the increment while il < 6 (:1308-1330) -> setupK -> calc_BC -> solve -> material sweep] until no tangent
 * changed and the yield function converged or 16 iterations (:1306-1381), state update (:1383-1392) and the data of
 * calc_global (= plfx_finish_step).  The host keeps the loop over load steps and its bookkeeping.
 * Needs plfx_set_bc_plan (with the segment sources below) and plfx_set_finish_set.
 * Segment source codes: 0 left (value bcl0[k]), 1 bottom (bcb0[k]), 2 right (dbcr[k]), 3 top (dbct[k]), 4 node set (dbcn[k]). */
typedef struct plfx_step {
    /* in */
    int32_t il, nonlin, has_nodeset, warm, maxit;
    int32_t defer_slot; /* 0: the call returns the end-of-step data; 1 or 2: it returns as soon as the end-of-step kernels are
                         * enqueued, the data are collected later with plfx_finish_fetch(slot - 1) -- the caller's bookkeeping
                         * and the predictor of the next step overlap the state update of this one */
    double rtol;
    double bcl0[2], bcb0[2];
    double max_dbcr[2], max_dbct[2], max_dbcn[2]; /* increments planned for this step (max_dbcn is updated like the reference's alias, :1285) */
    double bcr[2], bct[2], bcn[2];                /* target totals */
    double bcr0[2], bct0[2], bcn0[2];             /* reached before this step */
    double sld[6];                                /* loading direction for calc_scf (:1245-1258) */
    /* out */
    double dbcr[2], dbct[2], dbcn[2];             /* increments applied in the end */
    double scale_bc;
    int32_t nit, nconv, nsweeps, nsolves, soft_fail, inconsistent_entry;
    int32_t its[40];
    double relres[40];
} plfx_step;
/* which segments of the registered BC plan take which value, and the force-controlled segments (edge force split
 * over the nodes, model.py:1145-1151): src/k per segment; nf force segments with flen[nf] DOFs each, fidx / fshare
 * concatenated */
int plfx_set_bc_sources(plfx_ctx *ctx, int nseg, const int32_t *src, const int32_t *k, int nf, const int32_t *fsrc,
                        const int32_t *fk, const int32_t *flen, const int32_t *fidx, const double *fshare);
int plfx_load_step(plfx_ctx *ctx, plfx_step *step, double *u_at, double *f_at, double *sums18);
int plfx_finish_fetch(plfx_ctx *ctx, int slot, double *u_at, double *f_at, double *sums18);

/* ---------------------------------------------------------------- strip-local engine
 * Multi-GPU form of the whole path for the reference's structured grids: the global NX x NY grid (node j*(NY+1)+k, element
 * j*NY+k, model.py:893, 935) is cut into x-strips of whole element columns; every rank holds ONE strip as a standalone local
 * mesh (plfx_set_mesh with el_begin = 0, el_end = nel; plfx_set_grid with its local column count) made of its owned columns
 * [own_col0, own_col1) of the local grid plus `halo` columns on each interior side.  Nothing is replicated except the
 * coarse problem; per PCG iteration the ranks exchange one halo slab of the residual (contiguous node columns, ncclSend /
 * ncclRecv), one all-reduce of the owned level-`coarse_level` residual into the replicated global coarse grid, and three
 * all-reduces of <= 8 KB of partial sums.  The V-cycle is arithmetically the one a single GPU runs on the global grid
 * (validity widths in DESIGN.md section 6), so iteration counts do not depend on the number of strips.  Material state and
 * sweep: when plfx_set_mesh was given exactly the owned columns as its owned element range [el_begin, el_end), the strip
 * holds and sweeps only those, and the six stiffness generators of the halo columns arrive from the neighbour that owns
 * them after every sweep that rewrote a tangent anywhere (6 * halo * NY doubles per side); with el_begin = 0, el_end = nel
 * the sweep runs on the halo elements too (state recomputed from the exchanged displacement increment, nothing else sent).
 * global_col0 = global element column of local column 0; halo (derived) must be at least 4 * 2^coarse_level (32 for level 3, 64 for
 * level 4); all column numbers and NY must be multiples of 2^coarse_level.  Needs the matrix-free operator and, for more
 * than one strip, a communicator (plfx_comm_init / plfx_comm_init_callback) created BEFORE this call.
 * plfx_sweep flags, plfx_scf_all statistics and plfx_finish_step element sums then refer to the whole grid; u_at / f_at of
 * plfx_finish_step and the nodal arrays are local (owned + halo columns); element state arrays hold the owned element range. */
int plfx_set_strip(plfx_ctx *ctx, int own_col0, int own_col1, int global_col0, int global_nx, int coarse_level);
/* diagnostic: one ncclSend / ncclRecv pair of this rank with itself inside a group plus one all-reduce on scratch buffers,
 * results checked on the host (the entry points the library binds by dlsym, on whatever hardware is at hand) */
int plfx_comm_selftest(plfx_ctx *ctx);
int plfx_strip_info(plfx_ctx *ctx, int *active, int *halo, int *coarse_level, int *coarse_levels, int64_t *halo_refreshes,
                    int64_t *coarse_gathers, int64_t *partial_allreduces, int64_t *generator_exchanges);
/* in-place all-reduce of n <= 32 host doubles over the context's communicator (op 0 = sum, 3 = min): the boundary sums of
 * calc_global (model.py:1452-1471) over the strips.  No-op without a communicator. */
int plfx_allreduce_host(plfx_ctx *ctx, double *buf, int n, int op);

/* Host-staged transport for the same collectives (tests on a single GPU, hosts without RCCL): every in-place all-reduce
 * the library needs is staged through host memory and handed to `fn` (dtype 0 = double, 1 = int32; op 0 = sum, 3 = min;
 * return 0 on success).  op 100 = halo exchange of a strip: buf holds [slab for the left neighbour | slab for the right
 * neighbour] (count / 2 doubles each) and must come back as [slab from the left | slab from the right]; every rank calls.  Slow by construction -- the product transport is plfx_comm_init (RCCL). */
typedef int (*plfx_allreduce_fn)(void *user, void *buf, size_t count, int dtype, int op);
int plfx_comm_init_callback(plfx_ctx *ctx, int rank, int nranks, plfx_allreduce_fn fn, void *user);
/* device_collectives = 1: this context shards the elements over an RCCL communicator; plfx_sweep (flags),
 * plfx_scf_all (statistics) and plfx_finish_step (element sums) then return values of the WHOLE mesh (all-reduced on
 * the device, on the library's stream) and the caller needs no collective of its own. */
int plfx_comm_info(plfx_ctx *ctx, int *rank, int *nranks, int *device_collectives);
/* ---------------------------------------------------------------- multi-GPU
 * RCCL communicator for the per-CG-step all-reduce of the global vector.  id is the 128-byte
 * ncclUniqueId created on rank 0 by plfx_comm_unique_id and broadcast by the caller. */
int plfx_comm_unique_id(char id[128]);
int plfx_comm_init(plfx_ctx *ctx, const char id[128], int rank, int nranks);

/* ---------------------------------------------------------------- instrumentation */
/* accumulated HIP-event time (ms) and launch count of a named kernel family since the last reset:
 * which: 0 streaming phase of the material sweep (k_sweep_light / k_sweep_svc_wave<0>), 1 spmv(+dot), 2 cg vector
 *        update, 3 assemble, 4 multigrid V-cycle (whole cycle), 5 fine-level multigrid smoother launches,
 *        6 sub-stepping phase of the material sweep (k_sweep_heavy / k_sweep_svc_wave<1>),
 *        7 collectives on the library's stream (RCCL all-reduces, halo and generator exchanges; every call is timed, the time
 *          includes the wait for the slowest peer) */
int plfx_timing_get(plfx_ctx *ctx, int which, double *ms, int64_t *launches);
int plfx_timing_reset(plfx_ctx *ctx);
int plfx_timing_enable(plfx_ctx *ctx, int on);
/* Bit f of mask = time family f (default: all).  Every timed launch costs two hipEventRecord calls (about 4 us of host time
 * each and a bubble on the stream); bench.py times only the two kernels its roofline objects need. */
int plfx_timing_select(plfx_ctx *ctx, unsigned mask);
/* time only every n-th launch of a selected family (n >= 1; the averages stay representative, the host cost drops n-fold) */
int plfx_timing_sample(plfx_ctx *ctx, int every);

#ifdef __cplusplus
}
#endif
#endif /* PLFX_H */
