// ORACLE — test infrastructure only.  C entry points around the REFERENCE's own SedovSol
// class (/root/reference/sedov/sedov_sol.hpp:21-76), which compiles from its own sources
// (sedov_sol.cpp + two headers, std-only includes).  Built by `make ref` into
// oracle/_ref/libsedov_ref.so from the sources where they lie; nothing of the reference is
// copied into this repository.  Used to pin oracle/sedov_exact.cpp and to emit
// tests/golden/sedov_exact.json (tests/golden/make_sedov_exact.py).
#include "sedov_sol.hpp"

extern "C"
{
// out[21] in the order of oracle/sedov_exact.cpp's parameter block
void ref_sedov_setup(int dim, double gamma, double rho0, double E, double omega, double *out)
{
   SedovSol s(dim, gamma, rho0, E, omega);
   const double v[21] = {(double)s.dim, s.gamma, s.rho_0, s.blast_energy, s.omega, s.a, s.b, s.c, s.d, s.e,
                         s.alpha0, s.alpha1, s.alpha2, s.alpha3, s.alpha4, s.alpha5, s.V0, s.Vv, s.V2, s.Vs, s.alpha};
   for (int i = 0; i < 21; i++) { out[i] = v[i]; }
}

// shock[6] = r2, U, rho1, rho2, v2, p2;  then (rho, v, P)(r_i)
void ref_sedov_eval(int dim, double gamma, double rho0, double E, double omega, double t, long n, const double *r,
                    double *shock, double *rho, double *v, double *P)
{
   SedovSol s(dim, gamma, rho0, E, omega);
   s.SetTime(t);
   shock[0] = s.r2; shock[1] = s.U; shock[2] = s.rho1; shock[3] = s.rho2; shock[4] = s.v2; shock[5] = s.p2;
   for (long i = 0; i < n; i++) { s.EvalSol(r[i], rho[i], v[i], P[i]); }
}
}
