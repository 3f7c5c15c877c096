// laghos_solver.cpp — host orchestration of the hot path over the C ABI.
// Reference: /root/reference/laghos_solver.cpp:104-540 (operator), :699-797
// (timing report), :1436-1487 (RK2Avg); upstream RK4Solver.
#include "laghos_solver.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <limits>

namespace laghos
{
namespace hydrodynamics
{

void AbortWithLghError(const char *where)
{
   // the reference aborts on the same conditions (MFEM_ABORT, laghos_assembly.cpp:549-553)
   std::fprintf(stderr, "%s failed: %s\n", where, lgh_last_error());
   std::abort();
}

LagrangianHydroOperator::LagrangianHydroOperator(const Discretization &d, const std::vector<double> &S0,
                                                 const std::vector<double> &rho0_l2,
                                                 const std::vector<double> &gamma,
                                                 const std::vector<double> &rho0_q, double cfl,
                                                 double cgt, int cgiter, int device, const char *nccl_id)
   : disc(d), ctx(nullptr), dim(d.dim), NE(d.NE), H1Vsize(d.H1V), L2Vsize(d.L2V),
     H1GTVSize(d.dim * d.global_N), L2GTVSize(d.global_NE * d.NL), cg_rel_tol(cgt),
     cg_max_iter(cgiter), qdata_is_current(false), volume(0.0)
{
   // everything the reference's PA constructors pull from the spaces (laghos_solver.cpp:173-200)
   lgh_config cfg;
   std::memset(&cfg, 0, sizeof(cfg));
   cfg.dim = dim;
   cfg.NE = NE;
   cfg.D1D = d.tab.D1D;
   cfg.Q1D = d.tab.Q1D;
   cfg.L1D = d.tab.L1D;
   cfg.N = d.N;
   cfg.h1_map = d.h1map.data();
   cfg.B_h1 = d.tab.B.data();
   cfg.G_h1 = d.tab.G.data();
   cfg.B_l2 = d.tab.Bl.data();
   cfg.weights = d.W.data();
   cfg.gamma = gamma.data();
   for (int c = 0; c < 3; c++)
   {
      cfg.ess_count[c] = (c < dim) ? (int)d.ess[c].size() : 0;
      cfg.ess[c] = (c < dim) ? d.ess[c].data() : nullptr;
   }
   // LGH_FORCE_MULTI=1 drives the multi-rank code path (unfused E->L sums, RCCL
   // all-reduces of the device scalars) with a communicator of size 1: this is how
   // that path is exercised on a single-GPU box.
   const char *force_env = std::getenv("LGH_FORCE_MULTI");
   const bool force_multi = (force_env && force_env[0] == '1' && d.part.nranks == 1);
   const bool multi = d.part.nranks > 1 || force_multi;
   cfg.owner = multi ? d.owner.data() : nullptr;
   cfg.use_viscosity = d.UseViscosity();
   cfg.use_vorticity = d.UseVorticity() ? 1 : 0;
   cfg.cfl = cfl;
   cfg.order_v = d.tab.order_v;
   cfg.device = device;
   cfg.stream = nullptr;
   LGH_VERIFY(lgh_create(&cfg, &ctx));
   if (multi)
   {
      char self_id[128];
      if (!nccl_id && force_multi)
      {
         LGH_VERIFY(lgh_comm_unique_id(self_id));
         nccl_id = self_id;
      }
      if (!nccl_id) { std::fprintf(stderr, "multi-rank run needs an RCCL unique id\n"); std::abort(); }
      LGH_VERIFY(lgh_comm_init(ctx, d.part.nranks, d.part.rank, nccl_id));
      std::vector<int> cnt;
      std::vector<const int *> ptrs;
      for (auto &l : d.nbr_nodes) { cnt.push_back((int)l.size()); ptrs.push_back(l.data()); }
      LGH_VERIFY(lgh_comm_set_neighbors(ctx, (int)d.nbr_rank.size(), d.nbr_rank.data(), cnt.data(), ptrs.data()));
   }
   qdata.reset(new QuadratureData(ctx));
   qupdate.reset(new QUpdate(ctx));
   ForcePA.reset(new ForcePAOperator(*qdata, ctx));
   VMassPA.reset(new MassPAOperator(ctx, LGH_SPACE_H1));
   EMassPA.reset(new MassPAOperator(ctx, LGH_SPACE_L2));
   timer.L2dof = L2Vsize;

   // Rho0DetJ0Vol, h0 from the global volume (laghos_solver.cpp:223-262), Jacobi diagonal (:266-270)
   Vector x0((long)H1Vsize), r_l2, r_q;
   {
      std::vector<double> xh(S0.begin(), S0.begin() + H1Vsize);
      x0.FromHost(xh);
   }
   r_l2.FromHost(rho0_l2);
   r_q.FromHost(rho0_q);
   double vol = 0.0;
   LGH_VERIFY(lgh_setup_rho0detj0(ctx, x0.Read(), r_l2.Read(), r_q.Read(), &vol));
   double ne = (double)NE;
   if (multi)
   {
      LGH_VERIFY(lgh_allreduce(ctx, &vol, 0));
      LGH_VERIFY(lgh_allreduce(ctx, &ne, 0));
   }
   volume = vol;
   qdata->h0 = std::pow(vol / ne, 1.0 / dim) / (double)d.tab.order_v;
   LGH_VERIFY(lgh_set_h0(ctx, qdata->h0));

   one.SetSize(L2Vsize);
   LGH_VERIFY(lgh_vec_set(ctx, one.Write(), 1.0, L2Vsize)); // :170-171
   rhs.SetSize(H1Vsize);
   e_rhs.SetSize(L2Vsize);
   source_type = d.SourceType(); // laghos.cpp:636-647
   if (source_type == 1) { e_source.SetSize(L2Vsize); }
   if (source_type == 2)
   {
      // accel_src_gf.ProjectCoefficient(RTCoefficient) (:340-347): (0, -1) at every node
      std::vector<double> acc((size_t)H1Vsize, 0.0);
      for (long i = H1Vsize / dim; i < 2 * (H1Vsize / dim); i++) { acc[i] = -1.0; }
      accel_src.FromHost(acc);
      LGH_VERIFY(lgh_set_velocity_source(ctx, accel_src.Read()));
   }
   B.SetSize(d.N);
   LGH_VERIFY(lgh_sync(ctx));
}

LagrangianHydroOperator::~LagrangianHydroOperator() { lgh_destroy(ctx); }

void LagrangianHydroOperator::Mult(const Vector &S, Vector &dS_dt) const
{
   UpdateMesh(S);
   // dx_dt = v (laghos_solver.cpp:323)
   LGH_VERIFY(lgh_vec_copy(ctx, dS_dt.Write(), S.Read() + H1Vsize, H1Vsize));
   // SolveVelocity(S, dS_dt); SolveEnergy(S, v, dS_dt) (:324-325).  SolveEnergy takes v
   // from S, not from SolveVelocity's result, so the library may overlap the two: the
   // energy solve is enqueued first (second stream) and completed after the velocity solve.
   UpdateQuadratureData(S); // :332 / :445
   if (source_type == 1) { LGH_VERIFY(lgh_tg_source_2d(ctx, S.Read(), e_source.Write())); } // :448-467
   LGH_VERIFY(lgh_solve_energy_begin(ctx, S.Read(), S.Read() + H1Vsize, dS_dt.Write(), e_rhs.Write(),
                                     source_type == 1 ? e_source.Read() : nullptr, cg_rel_tol, cg_max_iter));
   SolveVelocity(S, dS_dt);
   int it = 0;
   LGH_VERIFY(lgh_solve_energy_end(ctx, &it));
   qdata_is_current = false; // :326
}

void LagrangianHydroOperator::SolveVelocity(const Vector &S, Vector &dS_dt) const
{
   UpdateQuadratureData(S); // :332
   int it = 0;
   // ForcePA->Mult(one, rhs); rhs.Neg(); per-component EliminateRHS + CG_VMass (:354-398)
   // (one: NULL = the operator's own constant-one function, :170-171; the member `one` above stays for ForcePAOperator::Mult callers)
   LGH_VERIFY(lgh_solve_velocity(ctx, S.Read(), dS_dt.Write(), nullptr, rhs.Write(), B.Write(),
                                 cg_rel_tol, cg_max_iter, &it));
}

void LagrangianHydroOperator::SolveEnergy(const Vector &S, const Vector &v, Vector &dS_dt) const
{
   UpdateQuadratureData(S); // :445
   int it = 0;
   // ForcePA->MultTranspose(v, e_rhs); CG_EMass.Mult(e_rhs, de) (:473-486)
   if (source_type == 1) { LGH_VERIFY(lgh_tg_source_2d(ctx, S.Read(), e_source.Write())); } // :448-467
   LGH_VERIFY(lgh_solve_energy(ctx, S.Read(), v.Read(), dS_dt.Write(), e_rhs.Write(),
                               source_type == 1 ? e_source.Read() : nullptr,
                               cg_rel_tol, cg_max_iter, &it));
}

void LagrangianHydroOperator::UpdateQuadratureData(const Vector &S) const
{
   // (:809) current for the host AND for the library: anything that discards the force products behind the operator's
   // back - lgh_set_fused_forces, lgh_qupdate_store_stress, a request for the mutable stressJinvT - bumps the
   // library's generation counter, and the data is then updated again instead of being read stale
   unsigned long gen = 0;
   int f1 = 0, ftv = 0;
   LGH_VERIFY(lgh_quadrature_generation(ctx, &gen, &f1, &ftv));
   if (qdata_is_current && gen == qdata_gen) { return; }
   qdata_is_current = true;
   qupdate->UpdateQuadratureData(S, *qdata); // :814
   LGH_VERIFY(lgh_quadrature_generation(ctx, &gen, &f1, &ftv));
   qdata_gen = gen;
}

double LagrangianHydroOperator::GetTimeStepEstimate(const Vector &S) const
{
   UpdateMesh(S);
   UpdateQuadratureData(S);
   double dt = qdata->GetDtEst();
   if (disc.part.nranks > 1) { LGH_VERIFY(lgh_allreduce(ctx, &dt, 1)); } // MPI_MIN (:533)
   return dt;
}

void LagrangianHydroOperator::ResetTimeStepEstimate() const
{
   qdata->SetDtEst(std::numeric_limits<double>::infinity()); // :539
}

double LagrangianHydroOperator::InternalEnergy(const Vector &S) const
{
   double r;
   LGH_VERIFY(lgh_internal_energy(ctx, S.Read() + 2 * (long)H1Vsize, &r));
   return r;
}
double LagrangianHydroOperator::KineticEnergy(const Vector &S) const
{
   double r;
   LGH_VERIFY(lgh_kinetic_energy(ctx, S.Read() + H1Vsize, &r));
   return r;
}
double LagrangianHydroOperator::ENorm(const Vector &S) const
{
   double n2;
   const double *e = S.Read() + 2 * (long)H1Vsize;
   LGH_VERIFY(lgh_vec_dot(ctx, e, e, L2Vsize, &n2));
   if (disc.part.nranks > 1) { LGH_VERIFY(lgh_allreduce(ctx, &n2, 0)); }
   return std::sqrt(n2);
}

void LagrangianHydroOperator::ComputeDensity(const Vector &S, Vector &rho) const
{
   if (rho.Size() != L2Vsize) { rho.SetSize(L2Vsize); }
   LGH_VERIFY(lgh_compute_density(ctx, S.Read(), rho.Write()));
}

double LagrangianHydroOperator::SedovDensityError(const Vector &S, const Vector &rho, const double par[21], double t,
                                                  const double origin[3], int err_order) const
{
   // IntRules.Get(cube, err_order): tensor Gauss-Legendre with err_order/2+1 points per direction
   const int n1 = err_order / 2 + 1;
   std::vector<double> pts, wts, Bh, Gh, Bl;
   GaussLegendre(n1, pts, wts);
   LagrangeTables(disc.tab.gll, pts, Bh, Gh);
   BernsteinTable(disc.tab.order_e, pts, Bl);
   double err2 = 0;
   LGH_VERIFY(lgh_sedov_density_error(ctx, S.Read(), rho.Read(), par, t, origin, n1, wts.data(), Bh.data(), Gh.data(),
                                      Bl.data(), &err2));
   return std::sqrt(err2);
}

double LagrangianHydroOperator::AllReduce(double v, int op) const
{
   if (disc.part.nranks > 1) { LGH_VERIFY(lgh_allreduce(ctx, &v, op)); }
   return v;
}
void LagrangianHydroOperator::Add(Vector &z, double a, const Vector &x, double b, const Vector &y) const
{
   LGH_VERIFY(lgh_vec_axpby(ctx, z.Write(), a, x.Read(), b, y.Read(), z.Size()));
}
void LagrangianHydroOperator::Add2(Vector &z1, double a1, const Vector &x1, double b1, Vector &z2, double a2, const Vector &x2, double b2,
                                   const Vector &y) const
{
   LGH_VERIFY(lgh_vec_axpby_pair(ctx, z1.Write(), a1, x1.Read(), b1, z2.Write(), a2, x2.Read(), b2, y.Read(), z1.Size()));
}
void LagrangianHydroOperator::Copy(Vector &y, const Vector &x) const
{
   LGH_VERIFY(lgh_vec_copy(ctx, y.Write(), x.Read(), y.Size()));
}
void LagrangianHydroOperator::Sync() const { LGH_VERIFY(lgh_sync(ctx)); }

const TimingData &LagrangianHydroOperator::Timing() const
{
   double t[4];
   long c[3];
   LGH_VERIFY(lgh_get_timers(ctx, t, c));
   timer.sw_cgH1 = t[0];
   timer.sw_cgL2 = t[1];
   timer.sw_force = t[2];
   timer.sw_qdata = t[3];
   timer.H1iter = c[0];
   timer.L2iter = c[1];
   timer.quad_tstep = c[2];
   return timer;
}
void LagrangianHydroOperator::ResetTiming() { LGH_VERIFY(lgh_reset_timers(ctx)); }
void LagrangianHydroOperator::EnableTimers(bool on) { LGH_VERIFY(lgh_enable_timers(ctx, on ? 1 : 0)); }

// laghos_solver.cpp:699-797 (same formulas and wording)
void LagrangianHydroOperator::PrintTimingData(bool IamRoot, int steps, bool fom) const
{
   const TimingData &tm = Timing();
   double T[5] = {tm.sw_cgH1, tm.sw_cgL2, tm.sw_force, tm.sw_qdata, 0.0};
   T[4] = T[0] + T[2] + T[3];
   for (int i = 0; i < 5; i++) { T[i] = -AllReduce(-T[i], 1); } // MPI_MAX as -min(-x)
   double l2work = (double)tm.L2dof * (double)tm.L2iter, quads = (double)tm.quad_tstep, zones = (double)NE;
   l2work = AllReduce(l2work, 0);
   quads = AllReduce(quads, 0);
   zones = AllReduce(zones, 0);
   if (!IamRoot) { return; }
   using namespace std;
   const long H1iter = tm.H1iter / dim;
   const int NQ = disc.NQ;
   const double FOM1 = 1e-6 * H1GTVSize * H1iter / T[0];
   const double FOM2 = 1e-6 * steps * (H1GTVSize + L2GTVSize) / T[2];
   const double FOM3 = 1e-6 * quads * NQ / T[3];
   const double FOM = (FOM1 * T[0] + FOM2 * T[2] + FOM3 * T[3]) / T[4];
   const double FOM0 = 1e-6 * steps * (H1GTVSize + L2GTVSize) / T[4];
   cout << endl;
   cout << "CG (H1) total time: " << T[0] << endl;
   cout << "CG (H1) rate (megadofs x cg_iterations / second): " << FOM1 << endl;
   cout << endl;
   cout << "CG (L2) total time: " << T[1] << endl;
   cout << "CG (L2) rate (megadofs x cg_iterations / second): " << 1e-6 * l2work / T[1] << endl;
   cout << endl;
   cout << "Forces total time: " << T[2] << endl;
   cout << "Forces rate (megadofs x timesteps / second): " << FOM2 << endl;
   {
      int f1 = 0, ftv = 0;
      if (lgh_get_fused_forces(ctx, &f1, &ftv) == LGH_OK && (f1 || ftv))
      {
         cout << "(the force products" << (f1 && ftv ? "" : (f1 ? " F.1" : " F^T v"))
              << " are formed inside UpdateQuadData: this region holds their E->L sum / right-hand-side set-up only)" << endl;
      }
   }
   cout << endl;
   cout << "UpdateQuadData total time: " << T[3] << endl;
   cout << "UpdateQuadData rate (megaquads x timesteps / second): " << FOM3 << endl;
   cout << endl;
   cout << "Major kernels total time (seconds): " << T[4] << endl;
   cout << "Major kernels total rate (megadofs x time steps / second): " << FOM << endl;
   if (!fom) { return; }
   const long ndofs = 2 * H1GTVSize + L2GTVSize + (long)NQ * (long)zones;
   cout << endl;
   cout << "| Ranks | Zones   | H1 dofs | L2 dofs | QP | N dofs   | FOM0   | FOM1   | T1   | FOM2   | T2   "
           "| FOM3   | T3   | FOM    | TT   |" << endl;
   cout << setprecision(3);
   cout << "| " << setw(6) << disc.part.nranks << "| " << setw(8) << (long)zones << "| " << setw(8)
        << H1GTVSize << "| " << setw(8) << L2GTVSize << "| " << setw(3) << NQ << "| " << setw(9) << ndofs
        << "| " << setw(7) << FOM0 << "| " << setw(7) << FOM1 << "| " << setw(5) << T[0] << "| " << setw(7)
        << FOM2 << "| " << setw(5) << T[2] << "| " << setw(7) << FOM3 << "| " << setw(5) << T[3] << "| "
        << setw(7) << FOM << "| " << setw(5) << T[4] << "| " << endl;
}

} // namespace hydrodynamics

// ---- ODE solvers ------------------------------------------------------------------------------
void ForwardEulerSolver::Init(hydrodynamics::LagrangianHydroOperator &op)
{
   ODESolver::Init(op);
   k.SetSize(op.Size());
}
void ForwardEulerSolver::Step(Vector &S, double &t, double &dt)
{
   f->Mult(S, k);
   f->Add(S, 1.0, S, dt, k); // x += dt f(x)
   t += dt;
}

void RK2Solver::Init(hydrodynamics::LagrangianHydroOperator &op)
{
   ODESolver::Init(op);
   k.SetSize(op.Size());
   x1.SetSize(op.Size());
}
void RK2Solver::Step(Vector &S, double &t, double &dt)
{
   //  0 |
   //  a |  a
   // ---+--------
   //    | 1-b  b      b = 1/(2a)
   const double b = 0.5 / a;
   f->Mult(S, k);
   f->Add(x1, 1.0, S, (1. - b) * dt, k);
   f->Add(S, 1.0, S, a * dt, k);
   f->Mult(S, k);
   f->Add(S, 1.0, x1, b * dt, k);
   t += dt;
}

void RK3SSPSolver::Init(hydrodynamics::LagrangianHydroOperator &op)
{
   ODESolver::Init(op);
   k.SetSize(op.Size());
   y.SetSize(op.Size());
}
void RK3SSPSolver::Step(Vector &S, double &t, double &dt)
{
   // x1 = x + dt f(x); x2 = 3/4 x + 1/4 (x1 + dt f(x1)); x3 = 1/3 x + 2/3 (x2 + dt f(x2))
   f->Mult(S, k);
   f->Add(y, 1.0, S, dt, k);
   f->Mult(y, k);
   f->Add(y, 1.0, y, dt, k);
   f->Add(y, 3. / 4, S, 1. / 4, y);
   f->Mult(y, k);
   f->Add(y, 1.0, y, dt, k);
   f->Add(S, 1. / 3, S, 2. / 3, y);
   t += dt;
}

void RK4Solver::Init(hydrodynamics::LagrangianHydroOperator &op)
{
   ODESolver::Init(op);
   k.SetSize(op.Size());
   y.SetSize(op.Size());
   z.SetSize(op.Size());
}

void RK4Solver::Step(Vector &S, double &t, double &dt)
{
   //   0  |
   //  1/2 | 1/2
   //  1/2 |  0   1/2
   //   1  |  0    0    1
   // -----+-------------------
   //      | 1/6  1/3  1/3  1/6
   // (the two combinations of a stage share their increment k: one pass over it - Add2 - with the bits of the two Adds)
   f->Mult(S, k);
   f->Add2(y, 1.0, S, dt / 2, z, 1.0, S, dt / 6, k);
   f->Mult(y, k);
   f->Add2(y, 1.0, S, dt / 2, z, 1.0, z, dt / 3, k);
   f->Mult(y, k);
   f->Add2(y, 1.0, S, dt, z, 1.0, z, dt / 3, k);
   f->Mult(y, k);
   f->Add(S, 1.0, z, dt / 6, k);
   t += dt;
}

void ExplicitRKSolver::Init(hydrodynamics::LagrangianHydroOperator &op)
{
   ODESolver::Init(op);
   y.SetSize(op.Size());
   k.resize(s);
   for (int i = 0; i < s; i++) { k[i].SetSize(op.Size()); }
}
void ExplicitRKSolver::Step(Vector &S, double &t, double &dt)
{
   //   0     |
   //  c[0]   | a[0]
   //  c[1]   | a[1] a[2]
   //  ...    |    ...
   //  c[s-2] | ...   a[s(s-1)/2-1]
   // --------+---------------------
   //         | b[0] b[1] ... b[s-1]          (same order of operations as upstream's Step)
   f->Mult(S, k[0]);
   for (int l = 0, i = 1; i < s; i++)
   {
      f->Add(y, 1.0, S, a[l++] * dt, k[0]);
      for (int j = 1; j < i; j++) { f->Add(y, 1.0, y, a[l++] * dt, k[j]); }
      f->Mult(y, k[i]); // the operator has no explicit time dependence (c unused, as f->SetTime upstream)
   }
   for (int i = 0; i < s; i++) { f->Add(S, 1.0, S, b[i] * dt, k[i]); }
   t += dt;
}
// upstream RK6Solver (Verner); the coefficients satisfy the order conditions through order 6 to 1e-30
// (checked in tests/test_host_setup.py)
const double RK6Solver::a[] = {
   .6e-1,
   .1923996296296296296296296296296296296296e-1,
   .7669337037037037037037037037037037037037e-1,
   .35975e-1,
   0.,
   .107925,
   1.318683415233148260919747276431735612861,
   0.,
   -5.042058063628562225427761634715637693344,
   4.220674648395413964508014358284402080483,
   -41.87259166432751461803757780644346812905,
   0.,
   159.4325621631374917700365669070346830453,
   -122.1192135650100309202516203389242140663,
   5.531743066200053768252631238332999150076,
   -54.43015693531650433250642051294142461271,
   0.,
   207.0672513650184644273657173866509835987,
   -158.6108137845899991828742424365058599469,
   6.991816585950242321992597280791793907096,
   -.1859723106220323397765171799549294623692e-1,
   -54.66374178728197680241215648050386959351,
   0.,
   207.9528062553893734515824816699834244238,
   -159.2889574744995071508959805871426654216,
   7.018743740796944434698170760964252490817,
   -.1833878590504572306472782005141738268361e-1,
   -.5119484997882099077875432497245168395840e-3};
const double RK6Solver::b[] = {
   .3438957868357036009278820124728322386520e-1,
   0.,
   0.,
   .2582624555633503404659558098586120858767,
   .4209371189673537150642551514069801967032,
   4.405396469669310170148836816197095664891,
   -176.4831190242986576151740942499002125029,
   172.3641334014150730294022582711902413315};
const double RK6Solver::c[] = {
   .6e-1,
   .9593333333333333333333333333333333333333e-1,
   .1439,
   .4973,
   .9725,
   .9995,
   1.};

void RK2AvgSolver::Init(hydrodynamics::LagrangianHydroOperator &op)
{
   ODESolver::Init(op);
   V.SetSize(op.H1VSize());
   dS_dt.SetSize(op.Size());
   S0.SetSize(op.Size());
}

void RK2AvgSolver::Step(Vector &S, double &t, double &dt)
{
   // laghos_solver.cpp:1447-1487.  Blocks of the monolithic vectors:
   // (position, velocity, specific internal energy).
   const long h1v = f->H1VSize();
   Vector v0, dx_dt, dv_dt;
   f->Copy(S0, S);
   v0.MakeRef(S0, h1v, h1v);
   dx_dt.MakeRef(dS_dt, 0, h1v);
   dv_dt.MakeRef(dS_dt, h1v, h1v);
   // -- 1. S is S0
   f->UpdateMesh(S);
   f->SolveVelocity(S, dS_dt);
   f->Add(V, 1.0, v0, 0.5 * dt, dv_dt); // V = v0 + 0.5 dt dv_dt
   f->SolveEnergy(S, V, dS_dt);
   f->Copy(dx_dt, V);
   // -- 2. S = S0 + 0.5 dt dS_dt
   f->Add(S, 1.0, S0, 0.5 * dt, dS_dt);
   f->ResetQuadratureData();
   f->UpdateMesh(S);
   f->SolveVelocity(S, dS_dt);
   f->Add(V, 1.0, v0, 0.5 * dt, dv_dt);
   f->SolveEnergy(S, V, dS_dt);
   f->Copy(dx_dt, V);
   // -- 3. S = S0 + dt dS_dt
   f->Add(S, 1.0, S0, dt, dS_dt);
   f->ResetQuadratureData();
   t += dt;
}

} // namespace laghos
