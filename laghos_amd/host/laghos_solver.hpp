// laghos_solver.hpp — LagrangianHydroOperator, QUpdate, TimingData and the ODE
// solvers with the reference's API surface (/root/reference/laghos_solver.hpp:36-255),
// re-expressed over the C ABI of the HIP library.
#pragma once
#include <memory>

#include "fem.hpp"
#include "laghos_assembly.hpp"

namespace laghos
{
namespace hydrodynamics
{

// laghos_solver.hpp:39-56; times come from HIP events inside the library.
struct TimingData
{
   double sw_cgH1 = 0, sw_cgL2 = 0, sw_force = 0, sw_qdata = 0;
   long L2dof = 0, H1iter = 0, L2iter = 0, quad_tstep = 0;
};

// laghos_solver.hpp:58-93
class QUpdate
{
   lgh_ctx *ctx;

public:
   explicit QUpdate(lgh_ctx *c) : ctx(c) {}
   void UpdateQuadratureData(const Vector &S, QuadratureData &) { LGH_VERIFY(lgh_qupdate(ctx, S.Read())); }
};

// Given a state (x, v, e) evaluates the slopes (dx_dt, dv_dt, de_dt)
// (laghos_solver.hpp:97-205, laghos_solver.cpp:104-540; PA branch, dim >= 2).
class LagrangianHydroOperator
{
protected:
   const Discretization &disc;
   lgh_ctx *ctx;
   const int dim, NE;
   const int H1Vsize, L2Vsize;
   const long H1GTVSize, L2GTVSize;
   const double cg_rel_tol;
   const int cg_max_iter;
   std::unique_ptr<QuadratureData> qdata;
   mutable bool qdata_is_current;
   mutable unsigned long qdata_gen = 0; // lgh_quadrature_generation() after the operator's last update
   std::unique_ptr<ForcePAOperator> ForcePA;
   std::unique_ptr<MassPAOperator> VMassPA, EMassPA;
   std::unique_ptr<QUpdate> qupdate;
   mutable Vector one, rhs, e_rhs, B;
   Vector accel_src;        // source_type == 2 (problem 7 gravity, laghos_solver.cpp:340-347)
   mutable Vector e_source; // source_type == 1 (2D Taylor-Green, laghos_solver.cpp:448-467), else empty
   int source_type = 0;
   mutable TimingData timer;
   double volume;

public:
   // nccl_id: 128-byte unique id for multi-rank runs (nullptr when nranks == 1)
   LagrangianHydroOperator(const Discretization &disc, const std::vector<double> &S0,
                           const std::vector<double> &rho0_l2, const std::vector<double> &gamma,
                           const std::vector<double> &rho0_q, double cfl, double cgt, int cgiter,
                           int device, const char *nccl_id);
   ~LagrangianHydroOperator();

   int Size() const { return 2 * H1Vsize + L2Vsize; }
   int H1VSize() const { return H1Vsize; }
   int L2VSize() const { return L2Vsize; }
   // Solve for dx_dt, dv_dt and de_dt (laghos_solver.cpp:308-327)
   void Mult(const Vector &S, Vector &dS_dt) const;
   void SolveVelocity(const Vector &S, Vector &dS_dt) const;
   void SolveEnergy(const Vector &S, const Vector &v, Vector &dS_dt) const;
   void UpdateMesh(const Vector &) const {} // nodes alias S.x: nothing to move
   void UpdateQuadratureData(const Vector &S) const;
   // Calls UpdateQuadratureData (laghos_solver.cpp:527-535); all-reduce MIN
   double GetTimeStepEstimate(const Vector &S) const;
   void ResetTimeStepEstimate() const;
   // (laghos_solver.hpp:193) - the force products lgh_qupdate formed with the stale data are discarded with it
   void ResetQuadratureData() const { qdata_is_current = false; LGH_VERIFY(lgh_reset_quadrature_data(ctx)); }
   double InternalEnergy(const Vector &S) const;
   double KineticEnergy(const Vector &S) const;
   double ENorm(const Vector &S) const; // ||e||_2, all-reduced (laghos.cpp:794-795)
   // zone-local L2 projection of the density on the current mesh (laghos_solver.cpp:542-563)
   void ComputeDensity(const Vector &S, Vector &rho) const;
   // sqrt of the integral of (rho_exact - rho)^2 against the exact Sedov solution `par` at time t,
   // error rule of order err_order (laghos.cpp:1007-1086); all-reduced
   double SedovDensityError(const Vector &S, const Vector &rho, const double par[21], double t,
                            const double origin[3], int err_order) const;
   void PrintTimingData(bool IamRoot, int steps, bool fom) const;
   const TimingData &Timing() const;
   void ResetTiming();
   void EnableTimers(bool on);
   lgh_ctx *Context() const { return ctx; }
   long GlobalH1Size() const { return H1GTVSize; }
   long GlobalL2Size() const { return L2GTVSize; }
   int Rank() const { return disc.part.rank; }
   int NRanks() const { return disc.part.nranks; }
   double AllReduce(double v, int op) const;
   // z = a x + b y on the context stream
   void Add(Vector &z, double a, const Vector &x, double b, const Vector &y) const;
   // z1 = a1 x1 + b1 y, z2 = a2 x2 + b2 y: two stage combinations of one increment in one pass (the bits of two Adds)
   void Add2(Vector &z1, double a1, const Vector &x1, double b1, Vector &z2, double a2, const Vector &x2, double b2, const Vector &y) const;
   void Copy(Vector &y, const Vector &x) const;
   void Sync() const;
};

} // namespace hydrodynamics

// ODE solvers (upstream MFEM ODESolver API: Init / Step)
class ODESolver
{
protected:
   hydrodynamics::LagrangianHydroOperator *f = nullptr;

public:
   virtual ~ODESolver() {}
   virtual void Init(hydrodynamics::LagrangianHydroOperator &op) { f = &op; }
   virtual void Step(Vector &S, double &t, double &dt) = 0;
   virtual int Stages() const = 0;
};

// upstream ForwardEulerSolver, RK2Solver(0.5) (midpoint) and RK3SSPSolver (laghos.cpp:521-523)
class ForwardEulerSolver : public ODESolver
{
   Vector k;

public:
   void Init(hydrodynamics::LagrangianHydroOperator &op) override;
   void Step(Vector &S, double &t, double &dt) override;
   int Stages() const override { return 1; }
};
class RK2Solver : public ODESolver
{
   Vector k, x1;
   double a;

public:
   explicit RK2Solver(double a_ = 0.5) : a(a_) {}
   void Init(hydrodynamics::LagrangianHydroOperator &op) override;
   void Step(Vector &S, double &t, double &dt) override;
   int Stages() const override { return 2; }
};
class RK3SSPSolver : public ODESolver
{
   Vector k, y;

public:
   void Init(hydrodynamics::LagrangianHydroOperator &op) override;
   void Step(Vector &S, double &t, double &dt) override;
   int Stages() const override { return 3; }
};

// classical RK4 (upstream RK4Solver; laghos.cpp:524)
class RK4Solver : public ODESolver
{
   Vector k, y, z;

public:
   void Init(hydrodynamics::LagrangianHydroOperator &op) override;
   void Step(Vector &S, double &t, double &dt) override;
   int Stages() const override { return 4; }
};

// upstream ExplicitRKSolver: s stages, Butcher tableau (a strictly lower triangular by rows, b, c),
// and RK6Solver (laghos.cpp:525): Verner's 8-stage 6th-order method with upstream's coefficients
class ExplicitRKSolver : public ODESolver
{
   int s;
   const double *a, *b, *c;
   Vector y;
   std::vector<Vector> k;

public:
   ExplicitRKSolver(int s_, const double *a_, const double *b_, const double *c_) : s(s_), a(a_), b(b_), c(c_) {}
   void Init(hydrodynamics::LagrangianHydroOperator &op) override;
   void Step(Vector &S, double &t, double &dt) override;
   int Stages() const override { return s; }
};
class RK6Solver : public ExplicitRKSolver
{
   static const double a[28], b[8], c[7];

public:
   RK6Solver() : ExplicitRKSolver(8, a, b, c) {}
};

// energy-conserving midpoint scheme (laghos_solver.cpp:1436-1487)
class RK2AvgSolver : public ODESolver
{
   Vector V, dS_dt, S0;

public:
   void Init(hydrodynamics::LagrangianHydroOperator &op) override;
   void Step(Vector &S, double &t, double &dt) override;
   int Stages() const override { return 2; }
};

} // namespace laghos
