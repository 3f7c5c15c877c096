// ORACLE — test infrastructure only (never linked or called by the product path).
//
// CPU restatement of the reference's Taylor–von Neumann–Sedov exact solution
// (Kamm, LA-UR-00-6055) used by `laghos -err`:
//   constants and the energy integral alpha ... /root/reference/sedov/sedov_sol.cpp:27-117
//   shock state at time t ..................... /root/reference/sedov/sedov_sol.cpp:119-130
//   point evaluation (rho, v, P)(r) ........... /root/reference/sedov/sedov_sol.cpp:132-198
//   adaptive 21-point Gauss–Kronrod rule ...... /root/reference/sedov/adaptive_quad.hpp:26-146
//   acceptance test of a panel ................ /root/reference/sedov/adaptive_quad.hpp:148-172
//   bisection ................................. /root/reference/sedov/bisect.hpp:27-96
// Pinned: against oracle/_ref/libsedov_ref.so (the reference's own sedov_sol.cpp compiled
// where it lies, oracle/Makefile target `ref`) and against tests/golden/sedov_exact.json
// (values emitted by that library, tests/golden/make_sedov_exact.py).
#include <cmath>
#include <algorithm>
#include <functional>

namespace
{

struct Sedov
{
   int dim;
   double gamma, rho0, E, omega;
   double a, b, c, d, e;
   double al0, al1, al2, al3, al4, al5;
   double V0, Vv, V2, Vs;
   double alpha;
};
constexpr int kNumPar = 21; // flattened Sedov

// Kronrod extension of the 10-point Gauss rule; panel = [lo, hi].  Nodes are visited Gauss
// points first (negative, then positive abscissae), then the Kronrod-only points, as the
// reference does, so that the sums round the same way.
static const double kGaussX[5] = {1.488743389816312108848260011297200e-01, 4.333953941292471907992659431657842e-01,
                                  6.794095682990244062343273651148736e-01, 8.650633666889845107320966884234930e-01,
                                  9.739065285171717200779640120844521e-01};
static const double kGaussW[5] = {2.955242247147528701738929946513383e-01, 2.692667193099963550912269215694694e-01,
                                  2.190863625159820439955349342281632e-01, 1.494513491505805931457763396576973e-01,
                                  6.667134430868813759356880989333179e-02};
static const double kGaussWK[5] = {1.477391049013384913748415159720680e-01, 1.347092173114733259280540017717068e-01,
                                   1.093871588022976418992105903258050e-01, 7.503967481091995276704314091619001e-02,
                                   3.255816230796472747881897245938976e-02};
static const double kKronX[6] = {0.0,
                                 2.943928627014601981311266031038656e-01,
                                 5.627571346686046833390000992726941e-01,
                                 7.808177265864168970637175783450424e-01,
                                 9.301574913557082260012071800595083e-01,
                                 9.956571630258080807355272806890028e-01};
static const double kKronW[6] = {1.494455540029169056649364683898212e-01, 1.427759385770600807970942731387171e-01,
                                 1.234919762620658510779581098310742e-01, 9.312545458369760553506546508336634e-02,
                                 5.475589657435199603138130024458018e-02, 1.169463886737187427806439606219205e-02};

static bool panel_accepted(double hi_order, double lo_order, double eps_abs, double eps_rel)
{
   if (!std::isfinite(hi_order)) { return true; }
   const double delta = std::fabs(hi_order - lo_order);
   if (delta < eps_abs) { return true; }
   return delta < eps_rel * std::max(std::fabs(hi_order), std::fabs(lo_order));
}

static double gk21_panel(const std::function<double(double)> &f, double lo, double hi, int depth, int max_depth,
                         double eps_abs, double eps_rel)
{
   const double half = (hi - lo) * 0.5;
   double g = 0.0, k = 0.0;
   for (int s = -1; s <= 1; s += 2)
   {
      for (int i = 0; i < 5; i++)
      {
         const double fx = f((s * kGaussX[i] + 1) * half + lo);
         g += fx * kGaussW[i];
         k += fx * kGaussWK[i];
      }
   }
   k += f((kKronX[0] + 1) * half + lo) * kKronW[0];
   for (int s = -1; s <= 1; s += 2)
   {
      for (int i = 1; i < 6; i++) { k += f((s * kKronX[i] + 1) * half + lo) * kKronW[i]; }
   }
   k *= half;
   g *= half;
   if (depth < max_depth && !panel_accepted(k, g, eps_abs, eps_rel))
   {
      k = gk21_panel(f, lo, lo + half, depth + 1, max_depth, eps_abs, eps_rel);
      k += gk21_panel(f, lo + half, hi, depth + 1, max_depth, eps_abs, eps_rel);
   }
   return k;
}

static double gk21(const std::function<double(double)> &f, double lo, double hi, int segments, int max_depth,
                   double eps_abs, double eps_rel)
{
   const double dx = (hi - lo) / segments;
   double sum = 0.0, left = lo;
   for (int i = 0; i < segments; i++)
   {
      const double right = lo + (i + 1) * dx;
      sum += gk21_panel(f, left, right, 1, max_depth, eps_abs, eps_rel);
      left = right;
   }
   return sum;
}

// bisect.hpp:27-96.  Returns NaN where the reference throws.
static double bisect(const std::function<double(double)> &f, double lo, double hi)
{
   const double tol = 1e-20;
   double flo = f(lo);
   if (std::fabs(flo) < tol) { return lo; }
   double fhi = f(hi);
   if (std::fabs(fhi) < tol) { return hi; }
   if (std::signbit(flo) == std::signbit(fhi)) { return std::nan(""); }
   const double width0 = hi - lo;
   double prev = width0;
   for (;;)
   {
      const double mid = 0.5 * (lo + hi);
      const double dx = mid - lo;
      const double fmid = f(mid);
      if (dx < width0 * 1e-16 || dx >= prev)
      {
         // no further progress in double precision: the end point with the smallest residual
         const double am = std::fabs(fmid), al = std::fabs(flo), ah = std::fabs(fhi);
         if (am < al) { return (am < ah) ? mid : ((ah < al) ? hi : lo); }
         return (ah < al) ? hi : lo;
      }
      if (std::fabs(fmid) < tol) { return mid; }
      if (std::signbit(flo) != std::signbit(fmid)) { hi = mid; fhi = fmid; }
      else if (std::signbit(fhi) != std::signbit(fmid)) { lo = mid; flo = fmid; }
      else { return std::nan(""); }
      prev = dx;
   }
}

static void setup(Sedov &s)
{
   const double n = s.dim, g = s.gamma, w = s.omega;
   const double j2w = n + 2 - w; // sedov_sol.cpp:32-49
   s.a = j2w * (g + 1) * 0.25;
   s.b = (g + 1) / (g - 1);
   s.c = j2w * g * 0.5;
   s.d = (j2w * (g + 1) / (j2w * (g + 1) - 2 * (2 + n * (g - 1))));
   s.e = (2 + n * (g - 1)) * 0.5;
   s.al0 = 2. / j2w;
   s.al2 = -(g - 1) / (2 * (g - 1) + n - g * w);
   s.al1 = (j2w * g / (2 + n * (g - 1)) * (2 * (n * (2 - g) - w) / (g * std::pow(j2w, 2)) - s.al2));
   s.al3 = (n - w) / (2 * (g - 1) + n - n * w);
   s.al4 = j2w * (n - w) * s.al1 / (n * (2 - g) - w);
   s.al5 = (w * (1 + g) - 2 * n) / (n * (2 - g) - w);
   s.V0 = 2. / (j2w * g);
   s.Vv = 2. / j2w;
   s.V2 = 4. / (j2w * (g + 1));
   s.Vs = 2. / ((g - 1) * n + 2);
   if (s.V2 == s.Vs)
   {
      s.alpha = (g + 1) / (g - 1) * std::pow(2, n) / std::pow(n * ((g - 1) * n + 2), 2); // :58-65
      if (s.dim > 1) { s.alpha *= M_PI; }
      return;
   }
   const Sedov p = s;
   // the part common to both energy integrands (:76-82, :98-103)
   auto common = [p, j2w, g](double V)
   {
      return std::pow((std::pow((p.a * V), p.al0) * std::pow((p.b * (p.c * V - 1)), p.al2) *
                       std::pow((p.d * (1 - p.e * V)), p.al1)),
                      (-j2w)) *
             std::pow((p.b * (p.c * V - 1)), p.al3) * std::pow((p.d * (1 - p.e * V)), p.al4) *
             std::pow((p.b * (1 - p.c * V / g)), p.al5);
   };
   auto J1f = [p, g, common](double V)
   {
      return -(g + 1) / (g - 1) * std::pow(V, 2) *
             (p.al0 / V + p.al2 * p.c / (p.c * V - 1) - p.al1 * p.e / (1 - p.e * V)) * common(V);
   };
   auto J2f = [p, g, common](double V)
   {
      double den = 1 - p.c * V;
      if (std::fabs(den) <= 1e-15) { den = std::copysign(1e-15, den); }
      return -(g + 1) / (2 * g) * std::pow(V, 2) * (p.c * V - g) / den *
             (p.al0 / V + p.al2 * p.c / -den - p.al1 * p.e / (1 - p.e * V)) * common(V);
   };
   const double Vmin = std::min(s.V0, s.Vv);
   const double J1 = gk21(J1f, Vmin, s.V2, 20, 64, 1.49e-15, 1.49e-15); // :87
   const double J2 = gk21(J2f, Vmin, s.V2, 20, 64, 1.49e-15, 1.49e-15); // :106
   double I1 = std::pow(2, n - 2) * J1;
   double I2 = std::pow(2, (n - 1)) / (g - 1) * J2;
   if (s.dim > 1) { I1 *= M_PI; I2 *= M_PI; }
   s.alpha = I1 + I2;
}

struct Shock { double r2, U, rho1, rho2, v2, p2; };

static Shock shock_at(const Sedov &s, double t) // :119-130
{
   Shock k;
   const double j2w = s.dim + 2 - s.omega;
   k.r2 = std::pow((s.E / (s.alpha * s.rho0)), (1. / j2w)) * std::pow(t, (2. / j2w));
   k.U = (2 / j2w) * (k.r2 / t);
   k.rho1 = s.rho0 * std::pow(k.r2, -s.omega);
   k.rho2 = ((s.gamma + 1) / (s.gamma - 1)) * k.rho1;
   k.v2 = (2 / (s.gamma + 1)) * k.U;
   k.p2 = (2 / (s.gamma + 1)) * k.rho1 * k.U * k.U;
   return k;
}

static void eval(const Sedov &s, const Shock &k, double r, double &rho, double &v, double &P) // :132-198
{
   if (r >= k.r2) { rho = s.rho0 * std::pow(r, -s.omega); v = 0; P = 0; return; }
   if (s.V2 == s.Vs)
   {
      rho = k.rho2 * std::pow((r / k.r2), (s.dim - 2));
      v = k.v2 * r / k.r2;
      P = k.p2 * std::pow((r / k.r2), s.dim);
      return;
   }
   auto x1 = [&](double V) { return s.a * V; };
   auto x2 = [&](double V) { return s.b * (s.c * V - 1); };
   auto x3 = [&](double V) { return s.d * (1 - s.e * V); };
   auto x4 = [&](double V) { return s.b * (1 - s.c * V / s.gamma); };
   auto lambda = [&](double V) { return std::pow(x1(V), -s.al0) * std::pow(x2(V), -s.al2) * std::pow(x3(V), -s.al1); };
   auto resid = [&](double V) { return k.r2 * lambda(V) - r; };
   double V;
   if (s.V2 < s.Vs) { V = bisect(resid, s.V0, s.V2); }
   else
   {
      V = bisect(resid, s.Vv, s.V2);
      if (r <= k.r2 * lambda(s.Vv)) { rho = 0; v = 0; P = 0; return; }
   }
   rho = k.rho2 * (std::pow(x1(V), s.al0 * s.omega) * std::pow(x2(V), (s.al3 + s.al2 * s.omega)) *
                   std::pow(x3(V), (s.al4 + s.al1 * s.omega)) * std::pow(x4(V), s.al5));
   v = k.v2 * (x1(V) * lambda(V));
   P = k.p2 * (std::pow(x1(V), (s.al0 * s.dim)) * std::pow(x3(V), (s.al4 + s.al1 * (s.omega - 2))) *
               std::pow(x4(V), (1 + s.al5)));
}

static void pack(const Sedov &s, double *o)
{
   const double v[kNumPar] = {(double)s.dim, s.gamma, s.rho0, s.E, s.omega, s.a, s.b, s.c, s.d, s.e, s.al0,
                              s.al1, s.al2, s.al3, s.al4, s.al5, s.V0, s.Vv, s.V2, s.Vs, s.alpha};
   std::copy(v, v + kNumPar, o);
}
static Sedov unpack(const double *o)
{
   Sedov s;
   s.dim = (int)o[0]; s.gamma = o[1]; s.rho0 = o[2]; s.E = o[3]; s.omega = o[4];
   s.a = o[5]; s.b = o[6]; s.c = o[7]; s.d = o[8]; s.e = o[9];
   s.al0 = o[10]; s.al1 = o[11]; s.al2 = o[12]; s.al3 = o[13]; s.al4 = o[14]; s.al5 = o[15];
   s.V0 = o[16]; s.Vv = o[17]; s.V2 = o[18]; s.Vs = o[19]; s.alpha = o[20];
   return s;
}

} // namespace

extern "C"
{
int lgo_sedov_num_par() { return kNumPar; }

// par[21]: dim, gamma, rho0, E, omega, a..e, alpha0..5, V0, Vv, V2, Vs, alpha
void lgo_sedov_setup(int dim, double gamma, double rho0, double E, double omega, double *par)
{
   Sedov s;
   s.dim = dim; s.gamma = gamma; s.rho0 = rho0; s.E = E; s.omega = omega;
   setup(s);
   pack(s, par);
}

// shock[6]: r2, U, rho1, rho2, v2, p2
void lgo_sedov_shock(const double *par, double t, double *shock)
{
   const Shock k = shock_at(unpack(par), t);
   shock[0] = k.r2; shock[1] = k.U; shock[2] = k.rho1; shock[3] = k.rho2; shock[4] = k.v2; shock[5] = k.p2;
}

void lgo_sedov_eval(const double *par, double t, long n, const double *r, double *rho, double *v, double *P)
{
   const Sedov s = unpack(par);
   const Shock k = shock_at(s, t);
#pragma omp parallel for schedule(static)
   for (long i = 0; i < n; i++) { eval(s, k, r[i], rho[i], v[i], P[i]); }
}
}
