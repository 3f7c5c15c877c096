// sedov_exact.hpp — the reference's SedovSol interface (Taylor–von Neumann–Sedov blast wave,
// /root/reference/sedov/sedov_sol.hpp:21-76) over the C ABI of liblaghos_hip.so: the
// constants and the energy integral are computed by lgh_sedov_setup, point values by
// lgh_sedov_eval_point (host, scalar) or lgh_sedov_eval (GPU, arrays).
#pragma once
#include "../../include/laghos_hip.h"

#include <cstdio>
#include <cstdlib>

struct SedovSol
{
   /// 1 for plane wave, 2 for cylinder, 3 for sphere
   int dim;
   double t = 0;
   double gamma, rho_0, omega, blast_energy;
   /// the flat parameter block handed to the library (include/laghos_hip.h)
   double par[21];
   /// the energy integral
   double alpha;
   /// time dependent: shock position and speed, pre- and post-shock state
   double r2 = 0, U = 0, rho1 = 0, rho2 = 0, v2 = 0, p2 = 0;

   SedovSol(int dim_, double gamma_, double rho_0_, double blast_energy_, double omega_ = 0)
      : dim(dim_), gamma(gamma_), rho_0(rho_0_), omega(omega_), blast_energy(blast_energy_)
   {
      Verify(lgh_sedov_setup(dim, gamma, rho_0, blast_energy, omega, par));
      alpha = par[20];
   }
   void SetTime(double t_)
   {
      t = t_;
      double s[6];
      Verify(lgh_sedov_shock(par, t, s));
      r2 = s[0]; U = s[1]; rho1 = s[2]; rho2 = s[3]; v2 = s[4]; p2 = s[5];
   }
   void EvalSol(double r, double &rho, double &v, double &P) const { Verify(lgh_sedov_eval_point(par, t, r, &rho, &v, &P)); }

private:
   static void Verify(int rc)
   {
      if (rc != LGH_OK)
      {
         std::fprintf(stderr, "SedovSol: %s\n", lgh_last_error());
         std::abort();
      }
   }
};
