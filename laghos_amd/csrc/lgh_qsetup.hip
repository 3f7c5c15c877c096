// lgh_qsetup.hip — Rho0DetJ0Vol (laghos_solver.cpp:1170-1261): the set-up mode of the quadrature-point kernel, compiled
// with the default flags (IEEE division and library roots): Jac0inv, rho0DetJ0w and the mass data are computed once and
// read by every kernel afterwards.
#include "lgh_qpoint.hpp"

namespace lgh
{

int setup_rho0detj0(lgh_ctx *c, const double *x0, const double *rho0_l2, const double *rho0_q,
                    double *volume)
{
   QArgs a = q_base(c);
   a.x = x0;
   a.e = rho0_l2;
   a.rho0_q = rho0_q;
   a.Jac0inv_out = c->Jac0inv;
   a.Jac0inv_soa_out = c->Jac0inv_soa;
   a.rho0DetJ0w_out = c->rho0DetJ0w;
   a.massD_out = c->massD;
   a.result = c->scal;
   int rc = launch_q<QMODE_SETUP>(c, a);
   if (rc) { return rc; }
   LGH_HIP_CHECK(hipMemcpyAsync(c->host_pinned, c->scal, sizeof(double), hipMemcpyDeviceToHost, c->stream));
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   *volume = c->host_pinned[0];
   return LGH_OK;
}

} // namespace lgh
