// lgh_sedov.hip — the `-err` row of SURVEY §8(f): density of the final state compared with
// the exact Taylor–von Neumann–Sedov blast wave (Kamm, LA-UR-00-6055).
//
//   exact solution ........ SedovSol, /root/reference/sedov/sedov_sol.{hpp,cpp}
//                           (constants :27-49, energy integral alpha :51-117 with the
//                           adaptive Gauss–Kronrod rule of sedov/adaptive_quad.hpp, shock
//                           state :119-130, point evaluation :132-198 with sedov/bisect.hpp)
//   density grid function .. LagrangianHydroOperator::ComputeDensity, laghos_solver.cpp:542-563
//                           (+ DensityIntegrator, laghos_assembly.cpp:26-41)
//   error integral ......... laghos.cpp:1007-1086
//
// The reference evaluates all of this on the host, point by point.  Here the set-up
// (a handful of scalars, two 1-D integrals) stays on the host and everything that scales
// with the mesh runs on the GPU: one workgroup per zone builds and solves the local L2
// projection, and the error integral evaluates the exact solution (a bisection per
// point, ~55 iterations of three pow calls) at the (n1d)^dim points of the error rule of
// every zone.  Set-up-grade kernels: called once per run, written for clarity.
#include "lgh_common.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace lgh
{

// Parameter block, 21 doubles in the order documented in include/laghos_hip.h.
struct SedovPar
{
   double dim, gamma, rho0, E, omega;
   double a, b, c, d, e;
   double al0, al1, al2, al3, al4, al5;
   double V0, Vv, V2, Vs;
   double alpha;
};
static_assert(sizeof(SedovPar) == 21 * sizeof(double), "SedovPar is the flat parameter block");
struct SedovShock { double r2, U, rho1, rho2, v2, p2; };

// ---- host set-up ---------------------------------------------------------------------

// 21-point Kronrod extension of the 10-point Gauss rule on [-1,1] (positive half; the
// rule is symmetric).  Gauss points carry both weights.
static const double kGx[5] = {1.488743389816312108848260011297200e-01, 4.333953941292471907992659431657842e-01,
                              6.794095682990244062343273651148736e-01, 8.650633666889845107320966884234930e-01,
                              9.739065285171717200779640120844521e-01};
static const double kGw[5] = {2.955242247147528701738929946513383e-01, 2.692667193099963550912269215694694e-01,
                              2.190863625159820439955349342281632e-01, 1.494513491505805931457763396576973e-01,
                              6.667134430868813759356880989333179e-02};
static const double kGwk[5] = {1.477391049013384913748415159720680e-01, 1.347092173114733259280540017717068e-01,
                               1.093871588022976418992105903258050e-01, 7.503967481091995276704314091619001e-02,
                               3.255816230796472747881897245938976e-02};
static const double kKx[6] = {0.0,
                              2.943928627014601981311266031038656e-01,
                              5.627571346686046833390000992726941e-01,
                              7.808177265864168970637175783450424e-01,
                              9.301574913557082260012071800595083e-01,
                              9.956571630258080807355272806890028e-01};
static const double kKw[6] = {1.494455540029169056649364683898212e-01, 1.427759385770600807970942731387171e-01,
                              1.234919762620658510779581098310742e-01, 9.312545458369760553506546508336634e-02,
                              5.475589657435199603138130024458018e-02, 1.169463886737187427806439606219205e-02};

// One panel of the adaptive rule; splits in two while the Gauss and Kronrod sums disagree
// (adaptive_quad.hpp:31-117; acceptance :148-172).  Evaluation order as in the reference.
template <class F> static double kronrod_panel(const F &f, double lo, double hi, int depth, int max_depth, double eps)
{
   const double half = 0.5 * (hi - lo);
   double gauss = 0.0, kron = 0.0;
   for (int sign = -1; sign <= 1; sign += 2)
   {
      for (int i = 0; i < 5; i++)
      {
         const double fx = f((sign * kGx[i] + 1) * half + lo);
         gauss += fx * kGw[i];
         kron += fx * kGwk[i];
      }
   }
   kron += f((kKx[0] + 1) * half + lo) * kKw[0];
   for (int sign = -1; sign <= 1; sign += 2)
   {
      for (int i = 1; i < 6; i++) { kron += f((sign * kKx[i] + 1) * half + lo) * kKw[i]; }
   }
   kron *= half;
   gauss *= half;
   bool ok = !std::isfinite(kron);
   if (!ok)
   {
      const double delta = std::fabs(kron - gauss);
      ok = delta < eps || delta < eps * std::max(std::fabs(kron), std::fabs(gauss));
   }
   if (ok || depth >= max_depth) { return kron; }
   const double left = kronrod_panel(f, lo, lo + half, depth + 1, max_depth, eps);
   return left + kronrod_panel(f, lo + half, hi, depth + 1, max_depth, eps);
}

template <class F> static double kronrod(const F &f, double lo, double hi, int panels, int max_depth, double eps)
{
   const double width = (hi - lo) / panels;
   double total = 0.0, left = lo;
   for (int i = 0; i < panels; i++)
   {
      const double right = lo + (i + 1) * width;
      total += kronrod_panel(f, left, right, 1, max_depth, eps);
      left = right;
   }
   return total;
}

static void sedov_setup(SedovPar &s)
{
   const double n = s.dim, g = s.gamma, w = s.omega, m = n + 2 - w;
   s.a = m * (g + 1) * 0.25;
   s.b = (g + 1) / (g - 1);
   s.c = m * g * 0.5;
   s.d = (m * (g + 1) / (m * (g + 1) - 2 * (2 + n * (g - 1))));
   s.e = (2 + n * (g - 1)) * 0.5;
   s.al0 = 2. / m;
   s.al2 = -(g - 1) / (2 * (g - 1) + n - g * w);
   s.al1 = (m * g / (2 + n * (g - 1)) * (2 * (n * (2 - g) - w) / (g * std::pow(m, 2)) - s.al2));
   s.al3 = (n - w) / (2 * (g - 1) + n - n * w);
   s.al4 = m * (n - w) * s.al1 / (n * (2 - g) - w);
   s.al5 = (w * (1 + g) - 2 * n) / (n * (2 - g) - w);
   s.V0 = 2. / (m * g);
   s.Vv = 2. / m;
   s.V2 = 4. / (m * (g + 1));
   s.Vs = 2. / ((g - 1) * n + 2);
   if (s.V2 == s.Vs) // singular case: closed form
   {
      s.alpha = (g + 1) / (g - 1) * std::pow(2, n) / std::pow(n * ((g - 1) * n + 2), 2);
      if (n > 1) { s.alpha *= M_PI; }
      return;
   }
   // standard / vacuum case: alpha = I1 + I2, two integrals over the similarity variable V
   const SedovPar p = s;
   auto tail = [&p, m, g](double V)
   {
      return std::pow((std::pow((p.a * V), p.al0) * std::pow((p.b * (p.c * V - 1)), p.al2) *
                       std::pow((p.d * (1 - p.e * V)), p.al1)),
                      (-m)) *
             std::pow((p.b * (p.c * V - 1)), p.al3) * std::pow((p.d * (1 - p.e * V)), p.al4) *
             std::pow((p.b * (1 - p.c * V / g)), p.al5);
   };
   auto kinetic = [&p, g, &tail](double V)
   {
      return -(g + 1) / (g - 1) * std::pow(V, 2) *
             (p.al0 / V + p.al2 * p.c / (p.c * V - 1) - p.al1 * p.e / (1 - p.e * V)) * tail(V);
   };
   auto internal = [&p, g, &tail](double V)
   {
      double den = 1 - p.c * V;
      if (std::fabs(den) <= 1e-15) { den = std::copysign(1e-15, den); }
      return -(g + 1) / (2 * g) * std::pow(V, 2) * (p.c * V - g) / den *
             (p.al0 / V + p.al2 * p.c / -den - p.al1 * p.e / (1 - p.e * V)) * tail(V);
   };
   const double Vmin = std::min(s.V0, s.Vv), eps = 1.49e-15;
   const double J1 = kronrod(kinetic, Vmin, s.V2, 20, 64, eps);
   const double J2 = kronrod(internal, Vmin, s.V2, 20, 64, eps);
   double I1 = std::pow(2, n - 2) * J1, I2 = std::pow(2, (n - 1)) / (g - 1) * J2;
   if (n > 1) { I1 *= M_PI; I2 *= M_PI; }
   s.alpha = I1 + I2;
}

// ---- point evaluation (host and device) ----------------------------------------------------------

__host__ __device__ static inline SedovShock sedov_shock(const SedovPar &s, const double t)
{
   SedovShock k;
   const double m = s.dim + 2 - s.omega;
   k.r2 = pow((s.E / (s.alpha * s.rho0)), (1. / m)) * pow(t, (2. / m));
   k.U = (2 / m) * (k.r2 / t);
   k.rho1 = s.rho0 * pow(k.r2, -s.omega);
   k.rho2 = ((s.gamma + 1) / (s.gamma - 1)) * k.rho1;
   k.v2 = (2 / (s.gamma + 1)) * k.U;
   k.p2 = (2 / (s.gamma + 1)) * k.rho1 * k.U * k.U;
   return k;
}

__host__ __device__ static inline double sedov_lambda(const SedovPar &s, const double V)
{
   return pow(s.a * V, -s.al0) * pow(s.b * (s.c * V - 1), -s.al2) * pow(s.d * (1 - s.e * V), -s.al1);
}

__host__ __device__ static inline bool neg_bit(const double x) { return __builtin_signbit(x) != 0; }

// Root of r2*lambda(V) - r on [lo, hi] by bisection down to the last bit (bisect.hpp:27-96).
// NaN where the reference throws (no sign change).
__host__ __device__ static inline double sedov_similarity_V(const SedovPar &s, const double r2, const double r,
                                                            double lo, double hi)
{
   const double tiny = 1e-20, nan = __builtin_nan("");
   double flo = r2 * sedov_lambda(s, lo) - r;
   if (fabs(flo) < tiny) { return lo; }
   double fhi = r2 * sedov_lambda(s, hi) - r;
   if (fabs(fhi) < tiny) { return hi; }
   if (neg_bit(flo) == neg_bit(fhi)) { return nan; }
   const double width0 = hi - lo;
   double last = width0;
   for (int it = 0; it < 4096; it++)
   {
      const double mid = 0.5 * (lo + hi), step = mid - lo;
      const double fmid = r2 * sedov_lambda(s, mid) - r;
      if (step < width0 * 1e-16 || step >= last)
      {
         const double am = fabs(fmid), al = fabs(flo), ah = fabs(fhi);
         if (am < al) { return (am < ah) ? mid : ((ah < al) ? hi : lo); }
         return (ah < al) ? hi : lo;
      }
      if (fabs(fmid) < tiny) { return mid; }
      if (neg_bit(flo) != neg_bit(fmid)) { hi = mid; fhi = fmid; }
      else if (neg_bit(fhi) != neg_bit(fmid)) { lo = mid; flo = fmid; }
      else { return nan; }
      last = step;
   }
   return nan;
}

__host__ __device__ static inline void sedov_point(const SedovPar &s, const SedovShock &k, const double r,
                                                   double &rho, double &v, double &P)
{
   if (r >= k.r2) // undisturbed gas ahead of the shock
   {
      rho = s.rho0 * pow(r, -s.omega);
      v = 0;
      P = 0;
      return;
   }
   if (s.V2 == s.Vs)
   {
      rho = k.rho2 * pow((r / k.r2), (s.dim - 2));
      v = k.v2 * r / k.r2;
      P = k.p2 * pow((r / k.r2), s.dim);
      return;
   }
   double V;
   if (s.V2 < s.Vs) { V = sedov_similarity_V(s, k.r2, r, s.V0, s.V2); }
   else
   {
      V = sedov_similarity_V(s, k.r2, r, s.Vv, s.V2);
      if (r <= k.r2 * sedov_lambda(s, s.Vv)) { rho = 0; v = 0; P = 0; return; }
   }
   const double x1 = s.a * V, x2 = s.b * (s.c * V - 1), x3 = s.d * (1 - s.e * V), x4 = s.b * (1 - s.c * V / s.gamma);
   rho = k.rho2 * (pow(x1, s.al0 * s.omega) * pow(x2, (s.al3 + s.al2 * s.omega)) * pow(x3, (s.al4 + s.al1 * s.omega)) *
                   pow(x4, s.al5));
   v = k.v2 * (x1 * sedov_lambda(s, V));
   P = k.p2 * (pow(x1, (s.al0 * s.dim)) * pow(x3, (s.al4 + s.al1 * (s.omega - 2))) * pow(x4, (1 + s.al5)));
}

__global__ void __launch_bounds__(256)
sedov_eval_k(const SedovPar s, const SedovShock k, const long n, const double *__restrict__ r,
             double *__restrict__ rho, double *__restrict__ v, double *__restrict__ P)
{
   const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) { return; }
   double a, b, c;
   sedov_point(s, k, r[i], a, b, c);
   rho[i] = a;
   v[i] = b;
   P[i] = c;
}

// ---- reference-space geometry at one point of a tensor rule ---------------------------------------
// Position and Jacobian of the zone map at the point (p0,p1,p2) of an n1^dim tensor rule,
// from the zone's nodes xe[c*ND + d] and 1-D tables T[p + n1*d] (values Bt, derivatives Gt).
template <int DIM>
__device__ static inline void zone_map_at(const int D, const int n1, const int *p, const double *__restrict__ Bt,
                                          const double *__restrict__ Gt, const double *xe, double *X, double *J)
{
   const int ND = (DIM == 3) ? D * D * D : D * D;
   for (int i = 0; i < DIM; i++) { X[i] = 0; }
   for (int i = 0; i < DIM * DIM; i++) { J[i] = 0; }
   for (int dz = 0; dz < (DIM == 3 ? D : 1); dz++)
   {
      const double bz = (DIM == 3) ? Bt[p[2] + n1 * dz] : 1.0, gz = (DIM == 3) ? Gt[p[2] + n1 * dz] : 0.0;
      for (int dy = 0; dy < D; dy++)
      {
         const double by = Bt[p[1] + n1 * dy], gy = Gt[p[1] + n1 * dy];
         for (int dx = 0; dx < D; dx++)
         {
            const double bx = Bt[p[0] + n1 * dx], gx = Gt[p[0] + n1 * dx];
            const int d = dx + D * (dy + D * dz);
            const double phi = bx * by * bz;
            double g[3] = {gx * by * bz, bx * gy * bz, bx * by * gz};
            for (int c = 0; c < DIM; c++)
            {
               const double xc = xe[c * ND + d];
               X[c] += xc * phi;
               for (int k = 0; k < DIM; k++) { J[c + DIM * k] += xc * g[k]; } // J[c,k] = d x_c / d xi_k
            }
         }
      }
   }
}
template <int DIM> __device__ static inline double det_small(const double *J)
{
   if (DIM == 2) { return J[0] * J[3] - J[1] * J[2]; }
   return J[0] * (J[4] * J[8] - J[5] * J[7]) - J[3] * (J[1] * J[8] - J[2] * J[7]) + J[6] * (J[1] * J[5] - J[2] * J[4]);
}
template <int DIM>
__device__ static inline double l2_shape(const int L, const int n1, const int *p, const int l, const double *__restrict__ Bl)
{
   const int lx = l % L, ly = (l / L) % L, lz = l / (L * L);
   double s = Bl[p[0] + n1 * lx] * Bl[p[1] + n1 * ly];
   if (DIM == 3) { s *= Bl[p[2] + n1 * lz]; }
   return s;
}

// ---- ComputeDensity: zone-local L2 projection -------------------------------------------------------
// rho_z = M_z^{-1} b_z,  M_ij = sum_q w_q detJ(q) psi_i psi_j on the current mesh,
// b_i = sum_q rho0DetJ0w(q) psi_i(q)  (rho detJ is conserved point-wise).  One workgroup per
// zone; the augmented matrix [M | b] lives in a per-workgroup global scratch (NL up to 125).
template <int DIM>
__global__ void __launch_bounds__(256)
density_project_k(const int NE, const int N, const int D, const int Q, const int L, const int *__restrict__ map,
                  const double *__restrict__ B, const double *__restrict__ G, const double *__restrict__ Bl,
                  const double *__restrict__ W, const double *__restrict__ x, const double *__restrict__ rho0DetJ0w,
                  double *__restrict__ scratch, double *__restrict__ rho)
{
   extern __shared__ double sm[];
   const int ND = (DIM == 3) ? D * D * D : D * D, NQ = (DIM == 3) ? Q * Q * Q : Q * Q, NL = (DIM == 3) ? L * L * L : L * L;
   double *xe = sm, *wdet = xe + DIM * ND, *sol = wdet + NQ;
   const int t = threadIdx.x, nt = blockDim.x, ld = NL + 1;
   double *M = scratch + (size_t)blockIdx.x * NL * ld;
   for (int e = blockIdx.x; e < NE; e += gridDim.x)
   {
      for (int i = t; i < DIM * ND; i += nt)
      {
         const int c = i / ND, d = i - c * ND;
         xe[i] = x[(size_t)c * N + map[(size_t)e * ND + d]];
      }
      __syncthreads();
      for (int q = t; q < NQ; q += nt)
      {
         const int p[3] = {q % Q, (q / Q) % Q, q / (Q * Q)};
         double X[3], J[9];
         zone_map_at<DIM>(D, Q, p, B, G, xe, X, J);
         wdet[q] = W[q] * det_small<DIM>(J);
      }
      __syncthreads();
      for (int idx = t; idx < NL * ld; idx += nt)
      {
         const int i = idx / ld, j = idx - i * ld;
         double s = 0;
         for (int q = 0; q < NQ; q++)
         {
            const int p[3] = {q % Q, (q / Q) % Q, q / (Q * Q)};
            const double pi = l2_shape<DIM>(L, Q, p, i, Bl);
            s += (j < NL) ? pi * l2_shape<DIM>(L, Q, p, j, Bl) * wdet[q] : pi * rho0DetJ0w[(size_t)e * NQ + q];
         }
         M[idx] = s;
      }
      __syncthreads();
      // elimination without pivoting (M is symmetric positive definite)
      for (int k = 0; k < NL - 1; k++)
      {
         const int rows = NL - 1 - k, cols = ld - 1 - k;
         const double piv = M[k * ld + k];
         for (int idx = t; idx < rows * cols; idx += nt)
         {
            const int i = k + 1 + idx / cols, j = k + 1 + idx % cols;
            M[i * ld + j] -= (M[i * ld + k] / piv) * M[k * ld + j];
         }
         __syncthreads();
      }
      if (t == 0)
      {
         for (int i = NL - 1; i >= 0; i--)
         {
            double s = M[i * ld + NL];
            for (int j = i + 1; j < NL; j++) { s -= M[i * ld + j] * sol[j]; }
            sol[i] = s / M[i * ld + i];
         }
      }
      __syncthreads();
      for (int i = t; i < NL; i += nt) { rho[(size_t)e * NL + i] = sol[i]; }
      __syncthreads();
   }
}

// ---- error integral ---------------------------------------------------------------------------
// part[e] = sum_p w_p detJ(p) (rho_exact(|x(p) - x0|) - rho_h(p))^2 over the n1^dim points of
// the error rule (laghos.cpp:1027-1080).
template <int DIM>
__global__ void __launch_bounds__(256)
sedov_density_error_k(const int NE, const int N, const int D, const int L, const int n1, const int *__restrict__ map,
                      const double *__restrict__ Bt, const double *__restrict__ Gt, const double *__restrict__ Blt,
                      const double *__restrict__ w1, const double *__restrict__ x, const double *__restrict__ rho_l2,
                      const SedovPar s, const SedovShock k, const double ox, const double oy, const double oz,
                      double *__restrict__ part)
{
   extern __shared__ double sm[];
   const int ND = (DIM == 3) ? D * D * D : D * D, NL = (DIM == 3) ? L * L * L : L * L;
   const int NP = (DIM == 3) ? n1 * n1 * n1 : n1 * n1;
   double *xe = sm, *re = xe + DIM * ND, *red = re + NL;
   const int t = threadIdx.x, nt = blockDim.x;
   const double org[3] = {ox, oy, oz};
   for (int e = blockIdx.x; e < NE; e += gridDim.x)
   {
      for (int i = t; i < DIM * ND; i += nt)
      {
         const int c = i / ND, d = i - c * ND;
         xe[i] = x[(size_t)c * N + map[(size_t)e * ND + d]];
      }
      for (int i = t; i < NL; i += nt) { re[i] = rho_l2[(size_t)e * NL + i]; }
      __syncthreads();
      double acc = 0;
      for (int q = t; q < NP; q += nt)
      {
         const int p[3] = {q % n1, (q / n1) % n1, q / (n1 * n1)};
         double X[3], J[9];
         zone_map_at<DIM>(D, n1, p, Bt, Gt, xe, X, J);
         double w = w1[p[0]] * w1[p[1]];
         if (DIM == 3) { w *= w1[p[2]]; }
         double rho_h = 0;
         for (int l = 0; l < NL; l++) { rho_h += re[l] * l2_shape<DIM>(L, n1, p, l, Blt); }
         double r = 0;
         for (int c = 0; c < DIM; c++) { r += (X[c] - org[c]) * (X[c] - org[c]); }
         r = sqrt(r);
         double rho_x, vx, px;
         sedov_point(s, k, r, rho_x, vx, px);
         const double diff = rho_x - rho_h;
         acc += w * det_small<DIM>(J) * (diff * diff);
      }
      red[t] = acc;
      __syncthreads();
      for (int off = nt / 2; off > 0; off >>= 1) // nt is a power of two
      {
         if (t < off) { red[t] += red[t + off]; }
         __syncthreads();
      }
      if (t == 0) { part[e] = red[0]; }
      __syncthreads();
   }
}

// fixed-order sum of n partials by one workgroup
__global__ void __launch_bounds__(1024) sum_ordered_k(const int n, const double *__restrict__ v, double *__restrict__ out)
{
   __shared__ double red[1024];
   const int t = threadIdx.x;
   double s = 0;
   for (int i = t; i < n; i += 1024) { s += v[i]; }
   red[t] = s;
   __syncthreads();
   for (int off = 512; off > 0; off >>= 1)
   {
      if (t < off) { red[t] += red[t + off]; }
      __syncthreads();
   }
   if (t == 0) { out[0] = red[0]; }
}

struct DevBuf
{
   void *p = nullptr;
   ~DevBuf() { if (p) { (void)hipFree(p); } }
   int alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 8)) == hipSuccess ? LGH_OK : LGH_ERR_HIP; }
};

} // namespace lgh

using namespace lgh;

static SedovPar par_from(const double *par)
{
   SedovPar s;
   std::copy(par, par + 21, reinterpret_cast<double *>(&s));
   return s;
}

extern "C"
{

int lgh_sedov_setup(int dim, double gamma, double rho0, double blast_energy, double omega, double par[21])
{
   LGH_CHECK_ARG(par && dim >= 1 && dim <= 3 && gamma > 1.0 && rho0 > 0.0 && blast_energy > 0.0);
   if (omega != 0.0)
   {
      // "currently only supports uniform initial density" (sedov_sol.hpp:36): with omega != 0 the
      // reference's own set-up returns NaN or does not terminate (adaptive rule at depth 64)
      set_error("lgh_sedov_setup: only omega = 0 (uniform initial density) is supported");
      return LGH_ERR_UNSUPPORTED;
   }
   SedovPar s;
   s.dim = dim; s.gamma = gamma; s.rho0 = rho0; s.E = blast_energy; s.omega = omega;
   sedov_setup(s);
   std::copy(reinterpret_cast<double *>(&s), reinterpret_cast<double *>(&s) + 21, par);
   return LGH_OK;
}

int lgh_sedov_shock(const double par[21], double t, double shock[6])
{
   LGH_CHECK_ARG(par && shock && t > 0.0);
   const SedovShock k = sedov_shock(par_from(par), t);
   shock[0] = k.r2; shock[1] = k.U; shock[2] = k.rho1; shock[3] = k.rho2; shock[4] = k.v2; shock[5] = k.p2;
   return LGH_OK;
}

int lgh_sedov_eval_point(const double par[21], double t, double r, double *rho, double *v, double *P)
{
   LGH_CHECK_ARG(par && rho && v && P && t > 0.0);
   const SedovPar s = par_from(par);
   sedov_point(s, sedov_shock(s, t), r, *rho, *v, *P);
   return LGH_OK;
}

int lgh_sedov_eval(lgh_ctx *c, const double par[21], double t, long n, const double *r, double *rho, double *v, double *P)
{
   LGH_CHECK_ARG(c && par && t > 0.0 && n >= 0 && (n == 0 || (r && rho && v && P)));
   if (n == 0) { return LGH_OK; }
   const SedovPar s = par_from(par);
   hipLaunchKernelGGL(sedov_eval_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, s, sedov_shock(s, t), n, r,
                      rho, v, P);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

int lgh_compute_density(lgh_ctx *c, const double *x_h1, double *rho_l2)
{
   LGH_CHECK_ARG(c && x_h1 && rho_l2);
   const int grid = std::min(c->NE, 2048);
   DevBuf scratch;
   if (scratch.alloc((size_t)grid * c->NL * (c->NL + 1) * sizeof(double))) { set_error("lgh_compute_density: hipMalloc failed"); return LGH_ERR_HIP; }
   const size_t lds = ((size_t)c->dim * c->ND + c->NQ + c->NL) * sizeof(double);
   if (c->dim == 3)
   {
      hipLaunchKernelGGL(density_project_k<3>, dim3(grid), dim3(256), lds, c->stream, c->NE, c->N, c->D1D, c->Q1D, c->L1D,
                         c->h1map, c->B, c->G, c->Bl, c->W, x_h1, c->rho0DetJ0w, (double *)scratch.p, rho_l2);
   }
   else
   {
      hipLaunchKernelGGL(density_project_k<2>, dim3(grid), dim3(256), lds, c->stream, c->NE, c->N, c->D1D, c->Q1D, c->L1D,
                         c->h1map, c->B, c->G, c->Bl, c->W, x_h1, c->rho0DetJ0w, (double *)scratch.p, rho_l2);
   }
   LGH_HIP_CHECK(hipGetLastError());
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream)); // the scratch is released on return
   return LGH_OK;
}

int lgh_sedov_density_error(lgh_ctx *c, const double *x_h1, const double *rho_l2, const double par[21], double t,
                            const double origin[3], int n1d, const double *weights, const double *B_h1,
                            const double *G_h1, const double *B_l2, double *err2)
{
   LGH_CHECK_ARG(c && x_h1 && rho_l2 && par && origin && weights && B_h1 && G_h1 && B_l2 && err2 && t > 0.0 &&
                 n1d >= 1 && n1d <= 64);
   const int D = c->D1D, L = c->L1D;
   // host tables of the error rule -> device: [w | B_h1 | G_h1 | B_l2]
   std::vector<double> tab;
   tab.insert(tab.end(), weights, weights + n1d);
   tab.insert(tab.end(), B_h1, B_h1 + (size_t)n1d * D);
   tab.insert(tab.end(), G_h1, G_h1 + (size_t)n1d * D);
   tab.insert(tab.end(), B_l2, B_l2 + (size_t)n1d * L);
   DevBuf dtab, part;
   if (dtab.alloc(tab.size() * sizeof(double)) || part.alloc((size_t)c->NE * sizeof(double)))
   {
      set_error("lgh_sedov_density_error: hipMalloc failed");
      return LGH_ERR_HIP;
   }
   LGH_HIP_CHECK(hipMemcpy(dtab.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
   const double *dw = (const double *)dtab.p, *dB = dw + n1d, *dG = dB + (size_t)n1d * D, *dBl = dG + (size_t)n1d * D;
   const SedovPar s = par_from(par);
   const SedovShock k = sedov_shock(s, t);
   const size_t lds = ((size_t)c->dim * c->ND + c->NL + 256) * sizeof(double);
   const int grid = std::min(c->NE, 1 << 16);
   if (c->dim == 3)
   {
      hipLaunchKernelGGL(sedov_density_error_k<3>, dim3(grid), dim3(256), lds, c->stream, c->NE, c->N, D, L, n1d, c->h1map,
                         dB, dG, dBl, dw, x_h1, rho_l2, s, k, origin[0], origin[1], origin[2], (double *)part.p);
   }
   else
   {
      hipLaunchKernelGGL(sedov_density_error_k<2>, dim3(grid), dim3(256), lds, c->stream, c->NE, c->N, D, L, n1d, c->h1map,
                         dB, dG, dBl, dw, x_h1, rho_l2, s, k, origin[0], origin[1], 0.0, (double *)part.p);
   }
   LGH_HIP_CHECK(hipGetLastError());
   hipLaunchKernelGGL(sum_ordered_k, dim3(1), dim3(1024), 0, c->stream, c->NE, (const double *)part.p, c->scal);
   LGH_HIP_CHECK(hipGetLastError());
   if (c->multi != 0)
   {
      const int rc = allreduce_dev(c, c->scal, 1, 0);
      if (rc) { return rc; }
   }
   LGH_HIP_CHECK(hipMemcpyAsync(c->host_pinned, c->scal, sizeof(double), hipMemcpyDeviceToHost, c->stream));
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   *err2 = c->host_pinned[0];
   return LGH_OK;
}
}
