// laghos_assembly.hpp — C++ shells with the reference's partial-assembly operator
// API (/root/reference/laghos_assembly.hpp:31-131): QuadratureData,
// ForcePAOperator, MassPAOperator.  Their Mult / MultTranspose bodies are calls
// through the C ABI (include/laghos_hip.h) into the HIP kernels; the MFEM space
// and integration-rule arguments of the reference constructors are replaced by
// the shared lgh_ctx, which was created from exactly that data
// (laghos_assembly.cpp:123-143, :80-96).
#pragma once
#include "../../include/laghos_hip.h"
#include "vector.hpp"

namespace laghos
{
namespace hydrodynamics
{

[[noreturn]] void AbortWithLghError(const char *where);
#define LGH_VERIFY(call)                                                          \
   do                                                                             \
   {                                                                              \
      if ((call) != LGH_OK) { ::laghos::hydrodynamics::AbortWithLghError(#call); } \
   } while (0)

// Container for all data needed at quadrature points (assembly.hpp:31-62).
// The arrays live in the context (device memory); this struct gives the
// reference's field names as views plus the host scalar h0.
struct QuadratureData
{
   lgh_ctx *ctx;
   double *Jac0inv;     // [i + dim*(j + dim*(e*NQ+q))]
   double *stressJinvT; // [(e*NQ+q) + NE*NQ*(gd + dim*vd)]
   double *rho0DetJ0w;  // [e*NQ+q]
   double h0;
   // dt_est lives on the device (folded by the QUpdate kernel); accessors below
   explicit QuadratureData(lgh_ctx *c)
      : ctx(c), Jac0inv(lgh_qdata_Jac0inv(c)), stressJinvT(lgh_qdata_stressJinvT(c)),
        rho0DetJ0w(lgh_qdata_rho0DetJ0w(c)), h0(0.0) {}
   void SetDtEst(double v) { LGH_VERIFY(lgh_set_dt_est(ctx, v)); }
   double GetDtEst() const
   {
      double v;
      LGH_VERIFY(lgh_get_dt_est(ctx, &v));
      return v;
   }
};

// Performs partial assembly for the force operator (assembly.hpp:94-112).
class ForcePAOperator
{
   lgh_ctx *ctx;
   const QuadratureData &qdata;

public:
   ForcePAOperator(const QuadratureData &qd, lgh_ctx *c) : ctx(c), qdata(qd) {}
   // x: L2 L-vector, y: H1 L-vector (laghos_assembly.cpp:557-565)
   void Mult(const Vector &x, Vector &y) const { LGH_VERIFY(lgh_force_mult(ctx, x.Read(), y.Write())); }
   // x: H1 L-vector, y: L2 L-vector (laghos_assembly.cpp:965-973)
   void MultTranspose(const Vector &x, Vector &y) const
   {
      LGH_VERIFY(lgh_force_mult_transpose(ctx, x.Read(), y.Write()));
   }
};

// Performs partial assembly for the velocity / energy mass matrix (assembly.hpp:115-131).
class MassPAOperator
{
   lgh_ctx *ctx;
   const int space; // LGH_SPACE_H1 (the scalar space H1c) or LGH_SPACE_L2

public:
   MassPAOperator(lgh_ctx *c, int space_) : ctx(c), space(space_) {}
   void Mult(const Vector &x, Vector &y) const { LGH_VERIFY(lgh_mass_mult(ctx, space, x.Read(), y.Write())); }
   void MultFull(const Vector &x, Vector &y) const { LGH_VERIFY(lgh_mass_mult_full(ctx, space, x.Read(), y.Write())); }
   // SetEssentialTrueDofs(c_tdofs[c]) (assembly.cpp:98-110): the per-component
   // lists were handed to the context at creation (laghos_solver.cpp:187-195)
   void SetEssentialTrueDofs(int comp) { LGH_VERIFY(lgh_mass_set_essential_tdofs(ctx, comp)); }
   void EliminateRHS(Vector &b) const { LGH_VERIFY(lgh_mass_eliminate_rhs(ctx, b.ReadWrite())); }
   int Space() const { return space; }
};

} // namespace hydrodynamics
} // namespace laghos
