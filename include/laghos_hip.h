/* laghos_hip.h — C ABI of the MI355X-native Laghos partial-assembly hot path.
 *
 * This is the drop-in boundary: the reference's C++ operator
 * classes keep their shape (laghos_amd/host/) and their Mult / MultTranspose /
 * UpdateQuadratureData bodies become calls into this library of hand-written
 * HIP kernels for gfx950.  Plain C types only; no torch / MFEM types.
 *
 * Conventions
 *  - Every `const double*` / `double*` vector argument is a DEVICE pointer owned
 *    by the caller (the reference's mfem::Vector device memory).  Pointers inside
 *    lgh_config are HOST pointers, copied at creation.
 *  - L-vectors use the reference layouts: H1 vectors are byNODES (component c at
 *    [c*N, (c+1)*N)), element-local dofs lexicographic; L2 dofs are e*L1D^dim + l
 *    (SURVEY A2-A4).  S = [x | v | e] with offsets {0, dim*N, 2*dim*N}
 *    (/root/reference/laghos_solver.cpp:166-169).
 *  - All work is enqueued on the context's HIP stream; calls are asynchronous
 *    unless they return a scalar to the host (documented per function).
 *  - Return value: 0 = LGH_OK, non-zero = error; lgh_last_error() gives the text.
 *    The reference aborts on the same conditions (MFEM_ABORT "Unknown kernel",
 *    laghos_assembly.cpp:549-553); the C++ shells turn non-zero into abort().
 *  - Not re-entrant per context; one host thread / one process per GPU.
 */
#ifndef LAGHOS_HIP_H
#define LAGHOS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define LGH_OK 0
#define LGH_ERR_ARG 1         /* bad argument */
#define LGH_ERR_UNSUPPORTED 2 /* (dim,D1D,Q1D) has no kernel: "Unknown kernel 0x..." */
#define LGH_ERR_HIP 3         /* HIP runtime error */
#define LGH_ERR_COMM 4        /* RCCL error */

#define LGH_SPACE_H1 0 /* scalar H1 space "H1c" (laghos_solver.cpp:121) */
#define LGH_SPACE_L2 1

typedef struct lgh_ctx lgh_ctx;

/* Everything ForcePAOperator / MassPAOperator / QUpdate constructors pull out of
 * the MFEM spaces (laghos_assembly.cpp:123-143, :80-96; laghos_solver.hpp:72-89). */
typedef struct lgh_config
{
   int dim, NE;            /* mesh dimension (2|3), local elements */
   int D1D, Q1D, L1D;      /* H1 dofs, quadrature points, L2 dofs per direction */
   int N;                  /* local scalar H1 nodes (H1c.GetVSize()) */
   const int *h1_map;      /* NE*D1D^dim: node of element-local lexicographic dof
                              (ElementRestriction, LEXICOGRAPHIC; assembly.cpp:133) */
   const double *B_h1;     /* Q1D*D1D, B[q + Q1D*d]  (DofToQuad::TENSOR, assembly.cpp:141-142) */
   const double *G_h1;     /* Q1D*D1D, derivative table */
   const double *B_l2;     /* Q1D*L1D */
   const double *weights;  /* NQ = Q1D^dim integration weights (ir.GetWeights()) */
   const double *gamma;    /* NE: order-0 L2 gamma grid function (laghos.cpp:628-632) */
   int ess_count[3];       /* c_tdofs[c].Size()  (laghos_solver.cpp:187-195) */
   const int *ess[3];      /* c_tdofs[c]: scalar nodes with v_c = 0 */
   const double *owner;    /* N, 1.0 where this rank owns the node, 0.0 where a lower
                              rank does (T-vector dot products); NULL = all owned */
   int use_viscosity, use_vorticity;
   double cfl;
   int order_v;            /* H1.GetOrder(0): h1order in QUpdateBody */
   int device;             /* HIP device ordinal */
   void *stream;           /* hipStream_t to use, or NULL to create one */
} lgh_config;

const char *lgh_last_error(void);
const char *lgh_version(void);

int lgh_create(const lgh_config *cfg, lgh_ctx **out);
int lgh_destroy(lgh_ctx *ctx);
int lgh_sync(lgh_ctx *ctx); /* hipStreamSynchronize on the context stream */
void *lgh_stream(lgh_ctx *ctx);

/* ---- QuadratureData (laghos_assembly.hpp:31-62); device arrays owned by ctx.
 *   stressJinvT[(e*NQ+q) + NE*NQ*(gd + dim*vd)]     (laghos_solver.cpp:1160-1167)
 *   Jac0inv[i + dim*(j + dim*(e*NQ+q))]              (laghos_solver.cpp:1195)
 *   rho0DetJ0w[e*NQ+q];  mass_D = w*detJ0*rho0(x_q)  (laghos_assembly.cpp:92-95)   */
double *lgh_qdata_stressJinvT(lgh_ctx *ctx);
double *lgh_qdata_Jac0inv(lgh_ctx *ctx);
double *lgh_qdata_rho0DetJ0w(lgh_ctx *ctx);
double *lgh_mass_D(lgh_ctx *ctx);
/* Form of the mass quadrature data the plane / slab mass kernels read: 1 = compact, D[q, e] = W[q] s_e (the data of
 * laghos_assembly.cpp:92-95 whenever rho0 detJ0 is constant within each element - every benchmark mesh of
 * BASELINE.json; the kernels then read one double per element instead of NQ), 0 = the stored table.  The test is made
 * on the device (every entry to 1e-12 relative, the size of the rounding of the stored entries; LGH_MASS_RANK1_TOL) at the first mass apply after lgh_setup_rho0detj0() or after lgh_mass_D() was
 * called: a caller that writes through the lgh_mass_D() pointer calls lgh_mass_D() again after its last write. */
int lgh_mass_data_form(lgh_ctx *ctx, int *form);
/* Form of qdata.Jac0inv (laghos_solver.cpp:1195) the row-form quadrature update reads: *compact = 1: one inverse Jacobian per
 * zone - lgh_setup_rho0detj0 found Jac0inv the same at every point of every zone (each entry against the zone's first point,
 * to 1e-12 of the zone's largest entry; LGH_JAC0_TOL), which is what an affine initial zone has - nine doubles per zone
 * instead of nine per quadrature point (15.5 of the 20.5 KB the update streams per zone at Q3Q2); 0: the point values
 * (a curved initial mesh; LGH_JAC0_COMPACT=0).  The arrays of lgh_qdata_Jac0inv() are written in full either way. */
int lgh_jac0inv_form(lgh_ctx *ctx, int *compact);
/* A caller that keeps the lgh_mass_D() pointer and writes the table after a mass apply has run says so with this call
 * (calling lgh_mass_D() again does the first half too): the compact form is tested for again at the next apply,
 * and the Jacobi diagonal (lgh_mass_diag, OperatorJacobiSmoother of laghos_solver.cpp:266-270) is reassembled from the
 * new table, so that the operator every kernel form applies and its preconditioner stay the same matrix. */
int lgh_mass_data_changed(lgh_ctx *ctx);
double *lgh_mass_diag(lgh_ctx *ctx); /* Jacobi diagonal of the scalar H1 mass (N) */
int lgh_set_h0(lgh_ctx *ctx, double h0);
int lgh_get_h0(lgh_ctx *ctx, double *h0);
/* qdata.dt_est: host-visible scalar; the get synchronises the stream. */
int lgh_set_dt_est(lgh_ctx *ctx, double dt_est);
int lgh_get_dt_est(lgh_ctx *ctx, double *dt_est);

/* Rho0DetJ0Vol (laghos_solver.cpp:1170-1261) + mass PA data + Jacobi diagonal
 * (OperatorJacobiSmoother, laghos_solver.cpp:266-270).  x0: H1 L-vector of the
 * initial nodes; rho0_l2: rho0 grid function (L2 dofs); rho0_q: the function rho0
 * at the NE*NQ physical quadrature points.  Synchronous; returns the local volume. */
int lgh_setup_rho0detj0(lgh_ctx *ctx, const double *x0, const double *rho0_l2,
                        const double *rho0_q, double *volume);

/* ---- ForcePAOperator (laghos_assembly.hpp:94-112) */
/* Mult (laghos_assembly.cpp:557-565): x L2 L-vector -> y H1 L-vector (dim*N). */
int lgh_force_mult(lgh_ctx *ctx, const double *x_l2, double *y_h1);
/* MultTranspose (laghos_assembly.cpp:965-973): v H1 L-vector -> y L2 L-vector. */
int lgh_force_mult_transpose(lgh_ctx *ctx, const double *v_h1, double *y_l2);

/* ---- MassPAOperator (laghos_assembly.hpp:115-131) */
/* SetEssentialTrueDofs(c_tdofs[comp]) (assembly.cpp:98-110); comp = -1 clears. */
int lgh_mass_set_essential_tdofs(lgh_ctx *ctx, int comp);
/* EliminateRHS (assembly.cpp:112-115): b[ess] = 0. */
int lgh_mass_eliminate_rhs(lgh_ctx *ctx, double *b);
/* Mult (assembly.cpp:117-121): y = M x, then y[ess] = 0.  space = LGH_SPACE_*. */
int lgh_mass_mult(lgh_ctx *ctx, int space, const double *x, double *y);
/* MultFull (assembly.hpp:127): no essential-row elimination. */
int lgh_mass_mult_full(lgh_ctx *ctx, int space, const double *x, double *y);

/* ---- CG solves configured at laghos_solver.cpp:264-284 (upstream CGSolver,
 * SURVEY §3.2).  space H1: Jacobi-preconditioned, iterative_mode (x = initial
 * guess), active essential list; space L2: plain CG, x overwritten.  Dot products
 * are wave-shuffle reductions, owner-masked and all-reduced over RCCL when a
 * communicator is attached.  Synchronous; *iters = GetNumIterations(). */
int lgh_cg_solve(lgh_ctx *ctx, int space, const double *b, double *x, double rel_tol,
                 int max_iter, int *iters);

/* ---- QUpdate::UpdateQuadratureData (laghos_solver.cpp:1354-1411): fused
 * E-restriction + reference-gradient + QKernel; updates stressJinvT and folds the
 * point-wise dt estimate into qdata.dt_est (device side, no host sync). */
int lgh_qupdate(lgh_ctx *ctx, const double *S);
/* The viscosity branch of QUpdateBody (laghos_solver.cpp:1086-1134) eigen-decomposes sym(grad v) at every
 * point.  When all 64 points of a wavefront have |sym grad v|_max <= tiny_grad (units 1/time) the kernel takes
 * the decomposition's own result for a tensor without deviatoric part (mu = tr/3, direction e_x) instead:
 * identical for exact zeros; for values below the threshold the stress differs by < visc_coeff * tiny_grad.
 * Default 1e-30 (LGH_Q_TINY_GRAD overrides); 0 = exact zeros only; negative = always decompose. */
int lgh_qupdate_set_tiny_grad(lgh_ctx *ctx, double tiny_grad);
/* lgh_qupdate also forms the two force products of the state it is called for - F.1 as E-vector (3D) and
 * F^T v for the state's own velocity - from the stress values it has in registers; lgh_solve_velocity and
 * lgh_solve_energy use them instead of a pass over stressJinvT each.  When they are used (no address is ever
 * compared):
 *   - both products belong to the quadrature data as lgh_qupdate left it: lgh_reset_quadrature_data, a set-up
 *     call, lgh_set_fused_forces and handing out the mutable lgh_qdata_stressJinvT pointer all discard them;
 *   - F.1 only for the all-ones L2 function (one_l2 == NULL, or a vector that is checked on that call);
 *   - F^T v only for a v whose every element equals the velocity block lgh_qupdate saw (a copy is kept and
 *     compared on the device inside lgh_solve_energy; any other v runs ForceMultTranspose, as
 *     laghos_solver.cpp:473 does for whatever v it is given).
 * on = 0 switches the fusion off: the products then always come from the ForcePAOperator kernels (per-kernel
 * timing, A/B).  Default on. */
int lgh_set_fused_forces(lgh_ctx *ctx, int on);
/* QuadratureData.stressJinvT is the hand-over between UpdateQuadratureData and the two ForcePA products in the
 * reference (laghos_solver.cpp:1158-1167 -> laghos_assembly.cpp:307-308, :859-872).  With both products formed inside
 * lgh_qupdate the nine planes are written and never read in a PA run whose SolveEnergy is always given the state's own
 * velocity (RK1-4, RK6): 43 % of the update's memory traffic.  on = 0: lgh_qupdate keeps the stress in registers and
 * does not write stressJinvT (3D, force fusion on); every reader then refuses with LGH_ERR_ARG instead of using stale
 * data - lgh_force_mult, lgh_force_mult_transpose, lgh_solve_velocity for a vector other than one; a SolveEnergy for a
 * velocity other than the state's (RK2Avg) turns its right-hand side into NaN on the device and the next
 * lgh_get_dt_est returns LGH_ERR_ARG.  lgh_qdata_stressJinvT() still returns the array (for inspection), but planes
 * nobody wrote do not become current by asking for them.
 * Default on = 1 (the reference's behaviour); a change takes effect with the next lgh_qupdate. */
int lgh_qupdate_store_stress(lgh_ctx *ctx, int on);
int lgh_qupdate_stores_stress(lgh_ctx *ctx, int *on);
/* Kernel lgh_qupdate launches for this context: 1 = qrows_kernel (3D up to Q4Q3: contraction stages with row-owning
 * threads, lgh_qrows.hpp), 0 = qpoint_kernel (every thread owns one output of every stage; 2D, Q5Q4, LGH_Q_FORM=0). */
int lgh_qupdate_form(lgh_ctx *ctx, int *form); /* whether the next lgh_qupdate writes stressJinvT (bench.py's byte accounting) */
/* ResetQuadratureData (laghos_solver.hpp: qdata_is_current = false): the state has changed, the quadrature data -
 * and the force products formed with it - are stale.  The shells call it wherever the reference does. */
int lgh_reset_quadrature_data(lgh_ctx *ctx);
/* The two fused products as vectors, for callers (and tests) that want ForcePA->Mult(one) / MultTranspose(v_state)
 * of the current quadrature data without another pass over it: F.1 summed to the H1 L-vector (dim*N), F^T v as
 * L2 vector.  LGH_ERR_ARG when the product is not on hand. */
int lgh_fused_force_mult(lgh_ctx *ctx, double *y_h1);
int lgh_fused_force_mult_transpose(lgh_ctx *ctx, double *y_l2);
/* *gen counts lgh_qupdate calls and invalidations; *f1_valid / *ftv_valid: a fused F.1 / F^T v is on hand (tests). */
int lgh_quadrature_generation(lgh_ctx *ctx, unsigned long *gen, int *f1_valid, int *ftv_valid);
/* *f1 / *ftv = 1 when lgh_qupdate forms F.1 / F^T v (what the region timers then see: the "Forces" region of
 * lgh_get_timers only holds the E->L sum and right-hand-side set-up, the products are inside "UpdateQuadData") */
int lgh_get_fused_forces(lgh_ctx *ctx, int *f1, int *ftv);

/* ---- LagrangianHydroOperator pieces kept together for launch efficiency
 * (laghos_solver.cpp:329-399, :442-490).  dS_dt = [dx|dv|de]; one_l2 is the
 * constant-one L2 vector (laghos_solver.cpp:170-171) or NULL for the operator's own (the reference's
 * SolveVelocity takes none: it is a member); rhs_h1 / e_rhs / work are
 * caller scratch (dim*N, L2 size, N).  e_source may be NULL.
 * *h1_iters / *l2_iters accumulate CG iteration counts (timer.H1iter, L2iter). */
int lgh_solve_velocity(lgh_ctx *ctx, const double *S, double *dS_dt, const double *one_l2,
                       double *rhs_h1, double *work_B, double rel_tol, int max_iter,
                       int *h1_iters);
int lgh_solve_energy(lgh_ctx *ctx, const double *S, const double *v_h1, double *dS_dt,
                     double *e_rhs, const double *e_source, double rel_tol, int max_iter,
                     int *l2_iters);
/* The same SolveEnergy split in two so that LagrangianHydroOperator::Mult
 * (laghos_solver.cpp:308-326) can overlap it with SolveVelocity, which it does not
 * depend on: _begin enqueues F^T v and the L2 CG on a second stream, _end completes
 * the solve and joins the context stream.  Call order: _begin, lgh_solve_velocity,
 * _end; all pointers must stay valid until _end returns.  The pair is equivalent to
 * one lgh_solve_energy call (identical iterates) and degrades to it when overlap is
 * not possible (timers on, several ranks, 2D). */
int lgh_solve_energy_begin(lgh_ctx *ctx, const double *S, const double *v_h1, double *dS_dt,
                           double *e_rhs, const double *e_source, double rel_tol, int max_iter);
int lgh_solve_energy_end(lgh_ctx *ctx, int *l2_iters);
/* Several ranks on ONE communicator (no second channel: the default over RCCL): _begin sets the energy CG up on the main stream
 * and lgh_solve_velocity enqueues one of its iterations behind each of its own - the mass apply behind K1, the update behind K2 -
 * with (d, M d) as a fourth scalar on the halo messages and (r, r) in a spare word of the accumulator-word exchange: the energy
 * solve costs no exchange and no host look of its own, as SolveEnergy costs none beside SolveVelocity on one rank.  _end runs
 * what is left (an energy solve that needs more iterations than the velocity solve took).  The iterates are those of
 * lgh_solve_energy up to the order in which the ranks' (r, r) and (d, M d) are added (rank order instead of RCCL's).
 * LGH_ENERGY_LOCKSTEP=0: the energy solve after the velocity solve.  Needs what the velocity solve's word exchange needs (every
 * rank a neighbour of every other, the slab form of K1) and region timers off.
 *   out[0] = energy solves run this way, out[1] = their iterations enqueued inside velocity solves, out[2] = ... after them,
 *   out[3] = 1: the next lgh_solve_energy_begin would take this path.
 * Replaces nothing in the reference (laghos_solver.cpp:400-493 runs the two solves one after the other). */
int lgh_energy_lockstep_stats(lgh_ctx *ctx, long out[4]);

/* ---- vector helpers on the context stream (device pointers) */
int lgh_vec_set(lgh_ctx *ctx, double *y, double a, long n);              /* y = a */
int lgh_vec_copy(lgh_ctx *ctx, double *y, const double *x, long n);      /* y = x */
/* z1 = a1 x1 + b1 y and z2 = a2 x2 + b2 y in one pass over y: the two combinations an explicit RK stage forms from the same
 * increment k (upstream RK4Solver::Step: the next stage state and the running sum of the solution).  The same expressions
 * as lgh_vec_axpby - the same bits as two calls.  z1 may be x1 and z2 may be x2 (the very same vector); any other overlap
 * between a result and an operand of the OTHER combination (z1 with z2, x2 or y; z2 with x1 or y) is refused with
 * LGH_ERR_ARG: two sequential calls would see the first result, the fused pass would not. */
int lgh_vec_axpby_pair(lgh_ctx *ctx, double *z1, double a1, const double *x1, double b1, double *z2, double a2, const double *x2, double b2,
                       const double *y, long n);
int lgh_vec_axpby(lgh_ctx *ctx, double *z, double a, const double *x, double b,
                  const double *y, long n);                                /* z = a x + b y */
int lgh_vec_dot(lgh_ctx *ctx, const double *x, const double *y, long n, double *result); /* sync */

/* ---- energies (laghos_solver.cpp:640-697); synchronous, all-reduced */
/* Acceleration source of SolveVelocity (source_type == 2, problem 7; laghos_solver.cpp:340-347,
 * :371-380): accel_h1 = nodal projection of RTCoefficient on the H1 space (dim*N, byNODES, device;
 * must stay valid), NULL switches it off.  lgh_solve_velocity then adds VMassPA->MultFull(accel_c)
 * to the right-hand side of every component before EliminateRHS. */
int lgh_set_velocity_source(lgh_ctx *ctx, const double *accel_h1);

/* 2D Taylor-Green energy source (SolveEnergy's source_type == 1 branch,
 * laghos_solver.cpp:448-467 with TaylorCoefficient laghos_solver.hpp:208-218):
 * e_source (L2 size, device) = DomainLFIntegrator(TaylorCoefficient) assembled on the
 * mesh positions in S; pass it to lgh_solve_energy(_begin). */
int lgh_tg_source_2d(lgh_ctx *ctx, const double *S, double *e_source);

int lgh_internal_energy(lgh_ctx *ctx, const double *e_l2, double *result);
int lgh_kinetic_energy(lgh_ctx *ctx, const double *v_h1, double *result);

/* ---- `-err`: density against the exact Sedov blast wave (laghos.cpp:1007-1086) ---------
 * Replaces the host-side SedovSol class (sedov/sedov_sol.hpp:21-76, sedov_sol.cpp),
 * LagrangianHydroOperator::ComputeDensity (laghos_solver.cpp:542-563) and the error loop
 * of the driver.  The solution's parameter block `par` is 21 host doubles:
 *   dim gamma rho0 E omega | a b c d e | alpha0..alpha5 | V0 Vv V2 Vs | alpha
 * (the members of SedovSol, sedov_sol.hpp:24-55). */
/* SedovSol::SedovSol (sedov_sol.cpp:27-117): constants and the energy integral alpha
 * (adaptive 21-point Gauss-Kronrod, host).  omega must be 0 (uniform initial density). */
int lgh_sedov_setup(int dim, double gamma, double rho0, double blast_energy, double omega, double par[21]);
/* SedovSol::SetTime (sedov_sol.cpp:119-130): shock[6] = r2 U rho1 rho2 v2 p2 at time t. */
int lgh_sedov_shock(const double par[21], double t, double shock[6]);
/* SedovSol::EvalSol (sedov_sol.cpp:132-198) at one radius, on the host (scalar API of the class). */
int lgh_sedov_eval_point(const double par[21], double t, double r, double *rho, double *v, double *P);
/* The same for n radii on the GPU: r, rho, v, P are device arrays; asynchronous on the context's stream. */
int lgh_sedov_eval(lgh_ctx *ctx, const double par[21], double t, long n, const double *r, double *rho, double *v,
                   double *P);
/* ComputeDensity (laghos_solver.cpp:542-563): zone-local L2 projection of the density on the
 * mesh positions x_h1 (= S, device) into rho_l2 (L2 size, device).  Needs lgh_setup_rho0detj0. */
int lgh_compute_density(lgh_ctx *ctx, const double *x_h1, double *rho_l2);
/* laghos.cpp:1027-1080: err2 = integral over the current mesh of (rho_exact(|x - origin|, t) - rho_h)^2
 * with the n1d^dim tensor Gauss-Legendre rule given by HOST tables on [0,1]: weights[n1d],
 * B_h1/G_h1 [p + n1d*d] (H1 basis values / derivatives), B_l2 [p + n1d*l] (L2 basis).
 * Summed over the ranks; synchronous. */
int lgh_sedov_density_error(lgh_ctx *ctx, const double *x_h1, const double *rho_l2, const double par[21], double t,
                            const double origin[3], int n1d, const double *weights, const double *B_h1,
                            const double *G_h1, const double *B_l2, double *err2);

/* ---- timing data (TimingData, laghos_solver.hpp:39-56): seconds measured with
 * HIP events around the same regions as the reference stopwatches.
 * t[0..3] = cgH1, cgL2, force, qdata; c[0..2] = H1iter, L2iter, quad_tstep */
int lgh_get_timers(lgh_ctx *ctx, double t[4], long c[3]);
int lgh_reset_timers(lgh_ctx *ctx);
int lgh_enable_timers(lgh_ctx *ctx, int on);

/* ---- per-kernel timing with HIP events on the context stream (bench.py roofline):
 * between begin and end every launch of kernel `which` is bracketed by an event
 * pair (up to max_samples launches); end synchronises and returns the number of
 * sampled launches and their mean duration in seconds. */
#define LGH_KERNEL_MASS_CG_H1 0   /* K1: gather + B^T D B + (d,Ad), H1 CG */
#define LGH_KERNEL_CG_UPDATE_H1 1 /* K2: transpose gather + vector updates + (r,z) */
#define LGH_KERNEL_QUPDATE 2
#define LGH_KERNEL_FORCE_MULT 3
#define LGH_KERNEL_FORCE_MULT_T 4
#define LGH_KERNEL_MASS_CG_L2 5
#define LGH_KERNEL_HALO 6         /* one shared-node / scalar exchange over RCCL: pack, grouped send/recv, combine */
#define LGH_KERNEL_ALLREDUCE 7    /* one ncclAllReduce of device scalars */
int lgh_ktime_begin(lgh_ctx *ctx, int which, int max_samples);
int lgh_ktime_end(lgh_ctx *ctx, int *launches, double *mean_seconds);
/* Whether lgh_create found the 1-D H1 / L2 tables mirror symmetric, B[q,d] = B[Q-1-q, D-1-d] (any nodal or
 * Bernstein basis on symmetric points is): the plane-form mass kernels keep half a table in scalar registers
 * and are only dispatched then (the column forms run otherwise). */
int lgh_table_symmetry(lgh_ctx *ctx, int *h1, int *l2);
/* Which form of the mass-apply kernel K1 the lockstep velocity solve (lgh_solve_velocity) launches for this
 * context: 0 = column form, 2 = plane form, 4 = slab form (sum factorisation in registers, lane-group transposes by v_permlane swaps, Q3Q2 only),
 * 5 = Kronecker form (compact mass data on a tensor-product rule: the element matrix as s_e M1 (x) M1 (x) M1 with the
 * 1-D mass tile M1 = B^T diag(w) B; the slab form applies the same where it is dispatched),
 * -1 = no lockstep solve for this kernel id (the scalar CG runs).  Tests use it to make sure a requested
 * form (LGH_VCG_VARIANT) is the one that ran. */
int lgh_k1_form(lgh_ctx *ctx, int *form);
/* Which kernel MassPAOperator::Mult on the L2 space (the energy CG, laghos_solver.cpp:480-488) launches for this context:
 * 2 = Kronecker form s_e M1l (x) M1l (x) M1l (compact mass data on a tensor-product rule, L1D <= 5), 1 = plane form,
 * 0 = column form; *compact = 1 when that kernel reads one factor per element instead of the NQ-entry table
 * (bench.py's byte accounting follows the kernel that runs, not the kernel id). */
int lgh_l2_mass_form(lgh_ctx *ctx, int *form, int *compact);
/* What the mass-apply kernel K1 of the lockstep velocity solve hands to its node kernel K2 in this context, for byte
 * accounting (bench.py): out[0] = doubles per velocity component of the E-vector between them (NE * D1D^3 in the
 * element-local layout; about a fifth less where the slab form of K1 sums the shared x-faces of a set of five zones
 * itself), out[1] = bytes of the transposed-restriction table K2 always reads (16 per node: contributions 1..4),
 * out[2] = bytes of its second half (contributions 5..8) in the 64-node blocks that need it, out[3] = E-vector entries
 * K1 has summed into a neighbour's.  Replaces nothing in the reference: there the E -> L sum is H1R^T
 * (laghos_assembly.cpp:121) inside every operator apply. */
int lgh_vcg_layout_stats(lgh_ctx *ctx, long out[4]);

/* The order in which the library walks the zones and numbers the nodes of its OWN vectors (the velocity solve's r, d, x,
 * 1/diag, E-vector and tables; the zone order of lgh_qupdate).  The reference hands its operators whatever numbering the mesh
 * library has - H1.GetElementRestriction(LEXICOGRAPHIC) of an MFEM space: vertex, edge, face, interior dofs; zones in the
 * order UniformRefinement leaves them (laghos_assembly.cpp:133-134, laghos.cpp:391) - and only the element-local dof order is
 * part of the interface.  lgh_create therefore finds the structure itself, from h1_map alone: zones are neighbours along a
 * local axis when the D1D x D1D nodes of their facing faces coincide position by position; a flood fill gives the zones of a
 * structured block integer coordinates; zones are then taken row by row and the nodes numbered along those rows.  The
 * caller's vectors keep the caller's numbering: they are read / written through the permutation once per solve.
 *   out[0] = 1: the mesh is a set of structured blocks (else the caller's order is kept: general path);
 *   out[1] = 1: the internal order IS the caller's (nothing is permuted; e.g. a lexicographic Cartesian generator);
 *   out[2] = connected components; out[3..5] = zones of the first component along its local x, y, z.
 * LGH_ORDER=0 keeps the caller's order everywhere (A/B).  Replaces nothing in the reference: MFEM's numbering is what it is. */
int lgh_mesh_order(lgh_ctx *ctx, long out[8]);
/* The same analysis as a host helper (no GPU, no context; tests, and callers that want to look at the order before they
 * create a context): zorder[i] = the caller's zone that comes i-th, nnum[n] = internal number of the caller's node n (both
 * the identity when out[0] == 0 or dim != 3); out as above. */
int lgh_mesh_order_host(int dim, int NE, int N, int D1D, const int *h1_map, int *zorder, int *nnum, long out[8]);

/* ---- multi-GPU: element blocks per rank, shared H1 nodes summed
 * over RCCL, dot products / dt all-reduced.  unique_id is the 128-byte
 * ncclUniqueId produced by lgh_comm_unique_id on rank 0 and broadcast by the
 * caller (e.g. torch.distributed).  Neighbour lists: for each of n_nbr peers,
 * the local node indices shared with that peer, sorted identically on both
 * sides (global lexicographic order). */
int lgh_comm_unique_id(char id_out[128]);
int lgh_comm_init(lgh_ctx *ctx, int nranks, int rank, const char unique_id[128]);
/* An id for the cross-process loopback transport instead of RCCL (tests, `bench.py --transport shm`): the ranks are
 * processes as on a node, but may share one GPU (RCCL refuses that); exchanges are staged through a POSIX
 * shared-memory segment named by the id.  Everything above the transport is the code a multi-GPU node runs. */
int lgh_comm_unique_id_shm(char id_out[128]);
int lgh_comm_set_neighbors(lgh_ctx *ctx, int n_nbr, const int *nbr_rank, const int *nbr_count,
                           const int *const *nbr_nodes);
/* What the exchanges of this rank look like (bench.py's `comm` block): neighbours, nodes of the largest message,
 * distinct shared nodes, whether every rank neighbours every other (sums then ride on the halo messages) and
 * whether the second channel (energy solve beside the velocity solve) is up. */
int lgh_comm_stats(lgh_ctx *ctx, int *n_neighbours, long *max_nodes_per_neighbour, long *shared_nodes, int *all_pairs,
                   int *second_channel);
/* Host helper (no GPU, no context): the `owner` mask of lgh_config and the neighbour lists of
 * lgh_comm_set_neighbors from a description of the shared dofs by GROUPS - the form in which MFEM holds them
 * (ParFiniteElementSpace::GroupComm(): GroupTopology = the rank set and master of every group,
 * GroupLDofTable = its L-dofs in an order common to all members; this is what P / P^T of
 * laghos_solver.cpp:368, :393 are built from).  INTEGRATION.md §4 shows the MFEM side.
 *   group g: ranks  group_ranks[group_off[g] .. group_off[g+1])  (my_rank among them, any order),
 *            master group_master[g] (NULL: the lowest rank of the group),
 *            L-dofs ldofs[ldof_off[g] .. ldof_off[g+1]) in the group's common order.
 * Outputs: owner[N] = 0 for dofs of groups mastered elsewhere, 1 otherwise;  *n_nbr peers in ascending rank
 * order, nbr_rank[k], nbr_count[k], and their node lists concatenated in nbr_nodes (neighbour k starts at the sum
 * of the counts before it).  For a peer the groups are taken in the order of their sorted rank sets (a key that
 * is the same on both sides) and the dofs inside a group in the group's order, so the two ranks of a pair
 * enumerate their shared dofs identically.  Capacities: nbr_rank / nbr_count >= number of distinct peers,
 * nbr_nodes >= sum over the groups of (size - 1) * number of dofs; cap_nbr / cap_nodes are checked. */
int lgh_groups_to_neighbors(int my_rank, int N, int n_groups, const int *group_off, const int *group_ranks,
                            const int *group_master, const int *ldof_off, const int *ldofs, double *owner,
                            int *n_nbr, int *nbr_rank, int *nbr_count, int cap_nbr, int *nbr_nodes, long cap_nodes);
/* in-place sum of shared nodes of an H1 L-vector with ncomp components */
int lgh_halo_sum(lgh_ctx *ctx, double *v_h1, int ncomp);
/* op 0 = sum, 1 = min; synchronous */
int lgh_allreduce(lgh_ctx *ctx, double *value, int op);

/* ---- E-vector level entry points (kernel-granularity parity tests only) */
int lgh_force_mult_E(lgh_ctx *ctx, const double *sJit, const double *x_E, double *y_E);
int lgh_force_mult_transpose_E(lgh_ctx *ctx, const double *sJit, const double *v_E, double *y_E);
int lgh_mass_apply_E(lgh_ctx *ctx, int space, const double *x_E, double *y_E);
/* ONE launch of the mass-apply kernel K1 of the lockstep velocity solve - in whichever form lgh_k1_form() reports
 * (column / plane / slab / Kronecker), i.e. the kernel lgh_solve_velocity spends most of its time in - exactly as
 * the solve launches it: first != 0: the first iteration (d = r/diag), else a later one (d = r/diag + beta d_old,
 * beta = rz[c] / rz_prev[c]).  r, d_old: H1 vectors of 3 components (byNODES, device); rz, rz_prev, den: 3 host
 * doubles; y_E: 3 planes of NE*D1D^3 (device) = the element contributions A_e d_e (MassPAOperator::Mult before the
 * E->L sum, laghos_assembly.cpp:117-121); den[c] = (d_c, A d_c).  rz[c] must be (r_c, r_c/diag) of the vectors
 * given (the slab form scales its exact accumulators by it).  Single rank; synchronous; overwrites the solve's own
 * work vectors only. */
int lgh_test_vcg_k1(lgh_ctx *ctx, const double *r, const double *d_old, const double rz[3], const double rz_prev[3],
                    int first, double *y_E, double den[3]);
/* The slab form of K1 stores a set of five x-neighbouring zones with the shared x-faces already summed (merged E-vector
 * layout, round 5: what K2 reads shrinks by a fifth and comes in whole cache lines).  lgh_test_vcg_k1 then reports such a
 * sum in the LEFT zone's entry (dx = 3) and 0.0 in the right zone's (dx = 0); mask[e * D1D^3 + d] (host, NE * D1D^3
 * bytes) = 1 marks those right-hand entries, *n_merged counts them (0 with every other form of K1 or LGH_SLAB_MERGE=0).
 * lgh_test_vcg_k2 takes the element-local E-vector either way (it forms the sums K1 would have stored). */
int lgh_test_vcg_merged_faces(lgh_ctx *ctx, unsigned char *mask, long *n_merged);
/* ONE launch of the node kernel K2 of the lockstep velocity solve (the kernel with the largest share of the step: the
 * E -> L sum of K1's element contributions, essential rows, r -= alpha A d, d = r_old/diag + beta d, the deferred update
 * of x, (r, r/diag)) exactly as the one-rank solve launches it in iteration it >= 1: y_E = 3 planes of NE*D1D^3
 * (device) as lgh_test_vcg_k1 returns them; r, d, x: H1 vectors of 3 components (device), updated in place - d in: the
 * direction of iteration it-1 (ignored for it = 1), d out: that of iteration it; den[c] = (d_c, A d_c) of this
 * iteration, rz[c] / rz_prev[c] = (r, z) after iterations it-1 / it-2, alpha_prev[c] = alpha of iteration it-1 (host
 * doubles).  rz_out[c] = (r, z) of the new residual.  *deferred_x = 1: the bounded-grid kernel ran - x is updated
 * only in even iterations, with the terms of iterations it and it-1 (it = 2: x := both terms, the old content is not
 * read); 0: the round-1 kernel (LGH_K2P=0), x += alpha d every iteration.  Single rank; synchronous. */
int lgh_test_vcg_k2(lgh_ctx *ctx, int it, const double *y_E, double *r, double *d, double *x, const double den[3],
                    const double rz[3], const double rz_prev[3], const double alpha_prev[3], double rz_out[3],
                    int *deferred_x);
/* halo pieces without the RCCL transport (tests emulate the exchange between
 * several contexts on one GPU): rank bookkeeping, pack into / combine from caller
 * buffers of 3*total doubles laid out as the send / receive buffers are. */
int lgh_test_set_rank(lgh_ctx *ctx, int nranks, int rank);
int lgh_test_halo_pack(lgh_ctx *ctx, const double *v_h1, int ncomp, double *sendbuf_out);
int lgh_test_halo_combine(lgh_ctx *ctx, const double *recvbuf_in, double *v_h1, int ncomp);
/* the peer buffer of the exact word exchange (the (r, z) accumulator words of the velocity CG on all-pairs partitions) as
 * the solve would size it for `nwords` words per peer: *capacity_words = peers x nwords.  A second
 * lgh_comm_set_neighbors with more neighbours must grow it (round-5 advisor). */
int lgh_test_word_peers(lgh_ctx *ctx, int nwords, long *capacity_words);
/* grouped ncclSend / ncclRecv of n doubles from this rank to itself on the RCCL communicator (what
 * halo_sum does with its neighbours); *max_abs_diff = max |received - sent| */
int lgh_test_rccl_self_sendrecv(lgh_ctx *ctx, int n, double *max_abs_diff);
/* device small-matrix probes: n matrices (column-major, 9 or 4 doubles each) */
int lgh_test_eig(lgh_ctx *ctx, int dim, int n, const double *A, double *lambda, double *vec);
/* y[i] = the device square root the small-matrix routines use (rsq + Goldschmidt step + correction), n device doubles */
int lgh_test_sqrt(lgh_ctx *ctx, int n, const double *x, double *y);
int lgh_test_singular(lgh_ctx *ctx, int dim, int n, const double *A, double *sv_min);

#ifdef __cplusplus
}
#endif
#endif /* LAGHOS_HIP_H */
