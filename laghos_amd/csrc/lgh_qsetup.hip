// lgh_qsetup.hip — Rho0DetJ0Vol (laghos_solver.cpp:1170-1261): the set-up mode of the quadrature-point kernel, compiled
// with the default flags (IEEE division and library roots): Jac0inv, rho0DetJ0w and the mass data are computed once and
// read by every kernel afterwards.
#include "lgh_qpoint.hpp"

namespace lgh
{

// one thread per zone: Je[k + n2 e] = Jac0inv(point 0 of zone e)[k]; *flag = 1 when a point of some zone differs
constexpr double kJac0Tol = 1e-12;
__global__ void __launch_bounds__(64)
jac0_compact_k(const int NE, const int NQ, const int n2, const double *__restrict__ soa, const double tol, double *__restrict__ Je, int *flag)
{
   const int e = blockIdx.x * blockDim.x + threadIdx.x;
   if (e >= NE) { return; }
   const size_t plane = (size_t)NE * NQ, base = (size_t)e * NQ;
   double ref[9], big = 0.0;
   for (int k = 0; k < n2; k++)
   {
      ref[k] = soa[base + plane * k];
      big = fmax(big, fabs(ref[k]));
      Je[(size_t)n2 * e + k] = ref[k];
   }
   bool ok = isfinite(big);
   for (int q = 1; q < NQ && ok; q++)
   {
      for (int k = 0; k < n2; k++) { ok = ok && fabs(soa[base + q + plane * k] - ref[k]) <= tol * big; }
   }
   if (!ok) { *flag = 1; }
}

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
   // Jac0inv the same at every point of a zone?  (An affine initial zone - every mesh of BASELINE.json - has ONE inverse
   // Jacobian; the 216 stored copies of it are 15.5 of the 20.5 KB the row-form update has to stream per zone at Q3Q2.)
   // Every entry of every point against the zone's first point, to kJac0Tol of the zone's largest entry; the update then
   // reads nine doubles per zone (lgh_jac0inv_form; LGH_JAC0_COMPACT=0: never).
   {
      const char *env = getenv("LGH_JAC0_COMPACT"), *tenv = getenv("LGH_JAC0_TOL");
      int *flag = c->dev_flags + 5, h = 1;
      LGH_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
      hipLaunchKernelGGL(jac0_compact_k, dim3(ceil_div(c->NE, 64)), dim3(64), 0, c->stream, c->NE, c->NQ, c->dim * c->dim, c->Jac0inv_soa,
                         // (a tolerance above 1e-10 would turn curved zones into affine ones: clamped - round-5 advisor)
                         tenv ? std::min(std::max(atof(tenv), 0.0), 1e-10) : kJac0Tol, c->Jac0inv_e, flag);
      LGH_HIP_CHECK(hipGetLastError());
      LGH_HIP_CHECK(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      c->jac0_compact = (h == 0 && !(env && env[0] == '0')) ? 1 : 0;
   }
   *volume = c->host_pinned[0];
   return LGH_OK;
}

} // namespace lgh
