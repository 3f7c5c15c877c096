// lgh_smallmat.hpp — per-quadrature-point dense 2x2 / 3x3 kernels for gfx950.
//
// Device-side counterparts of the mfem::kernels:: calls made by QUpdateBody
// (/root/reference/laghos_solver.cpp:1078-1080, :1095, :1105, :1113, :1117-1121,
// :1133, :1139, :1158).  Upstream MFEM is not part of the reference tree; the
// algorithms are the published ones (scaled trigonometric root of the deviator's
// characteristic cubic, Householder deflation onto a 2x2 block, Parlett's
// rotation) so that results agree with the reference CPU path to round-off.
// Column-major storage: A(i,j) = a[i + n*j].  Everything is fp64.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>

#define LGH_HD __host__ __device__ __forceinline__

namespace lgh
{
namespace sm
{

LGH_HD void swap2(double &a, double &b) { const double t = a; a = b; b = t; }

// sqrt on the device: v_rsq_f64, one Goldschmidt step and one residual correction - 8 instructions and <= 1 ulp on
// normal operands, against 22 for the compiler's correctly rounded expansion (which rescales tiny and huge
// operands and handles the special values; the update kernel is bound by vector issue and takes ~10 roots per
// point).  +-0 and +inf are returned as they are (one class test); no denormal rescue.
// tests/test_gpu_kernels.py::test_device_sqrt.  Building with -DLGH_IEEE_SQRT (make IEEE_SQRT=1) takes the correctly
// rounded library root instead - the reference's arithmetic, for parity runs that want it.
LGH_HD double fsqrt(const double x)
{
#if defined(__HIP_DEVICE_COMPILE__) && defined(LGH_IEEE_SQRT)
   return __builtin_sqrt(x);
#elif defined(__HIP_DEVICE_COMPILE__)
   const double y = __builtin_amdgcn_rsq(x);
   double g = x * y, h = 0.5 * y;
   const double r = fma(-h, g, 0.5);
   g = fma(g, r, g);
   h = fma(h, r, h);
   const double d = fma(-g, g, x);
   g = fma(d, h, g);
   return __builtin_amdgcn_class(x, 0x260) ? x : g; // -0, +0, +inf
#else
   return std::sqrt(x);
#endif
}

LGH_HD double det2(const double *J) { return J[0] * J[3] - J[1] * J[2]; }
LGH_HD double det3(const double *J)
{
   return J[0] * (J[4] * J[8] - J[5] * J[7]) + J[3] * (J[2] * J[7] - J[1] * J[8]) +
          J[6] * (J[1] * J[5] - J[2] * J[4]);
}
template <int DIM> LGH_HD double det(const double *J) { return DIM == 2 ? det2(J) : det3(J); }

// adjugate / det
template <int DIM> LGH_HD void inverse(const double *J, const double detJ, double *Ji)
{
   const double d = 1.0 / detJ;
   if (DIM == 2)
   {
      Ji[0] = J[3] * d;
      Ji[1] = -J[1] * d;
      Ji[2] = -J[2] * d;
      Ji[3] = J[0] * d;
   }
   else
   {
      Ji[0] = (J[4] * J[8] - J[5] * J[7]) * d;
      Ji[3] = (J[5] * J[6] - J[3] * J[8]) * d;
      Ji[6] = (J[3] * J[7] - J[4] * J[6]) * d;
      Ji[1] = (J[2] * J[7] - J[1] * J[8]) * d;
      Ji[4] = (J[0] * J[8] - J[2] * J[6]) * d;
      Ji[7] = (J[1] * J[6] - J[0] * J[7]) * d;
      Ji[2] = (J[1] * J[5] - J[2] * J[4]) * d;
      Ji[5] = (J[2] * J[3] - J[0] * J[5]) * d;
      Ji[8] = (J[0] * J[4] - J[1] * J[3]) * d;
   }
}

// C = A B (all n x n)
template <int N> LGH_HD void matmul(const double *A, const double *B, double *C)
{
#pragma unroll
   for (int j = 0; j < N; j++)
#pragma unroll
      for (int i = 0; i < N; i++)
      {
         double s = 0.0;
#pragma unroll
         for (int l = 0; l < N; l++) { s += A[i + N * l] * B[l + N * j]; }
         C[i + N * j] = s;
      }
}
// C = A B^T
template <int N> LGH_HD void matmul_abt(const double *A, const double *B, double *C)
{
#pragma unroll
   for (int j = 0; j < N; j++)
#pragma unroll
      for (int i = 0; i < N; i++)
      {
         double s = 0.0;
#pragma unroll
         for (int l = 0; l < N; l++) { s += A[i + N * l] * B[j + N * l]; }
         C[i + N * j] = s;
      }
}
template <int N> LGH_HD void matvec(const double *A, const double *x, double *y)
{
#pragma unroll
   for (int i = 0; i < N; i++)
   {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < N; j++) { s += A[i + N * j] * x[j]; }
      y[i] = s;
   }
}
template <int N> LGH_HD void symmetrize(double *A)
{
#pragma unroll
   for (int i = 0; i < N; i++)
#pragma unroll
      for (int j = 0; j < i; j++)
      {
         const double a = 0.5 * (A[i + N * j] + A[j + N * i]);
         A[i + N * j] = A[j + N * i] = a;
      }
}
// power-of-two scale with d_max / mult in [0.5, 1): mult = 2^ex (the value the
// reference obtains as d_max / frexp-mantissa, an exact quotient), and its exact
// reciprocal, so the 6-9 scalings cost a multiply each instead of an fp64 divide
// (bit-identical results: scaling by a power of two is exact either way).
LGH_HD double scaling_factor(const double d_max, double &inv_mult)
{
   if (!(d_max > 0.))
   {
      inv_mult = 1.;
      return 1.;
   }
   int ex;
   (void)frexp(d_max, &ex);
   if (ex == DBL_MAX_EXP) { ex -= 1; }
   inv_mult = ldexp(1.0, -ex);
   return ldexp(1.0, ex);
}

// overflow-safe Euclidean norm.  The reference's kernels::Norml2 rescales the running sum by the largest entry so
// far, with a division per entry; scaling all entries by one exact power of two near the largest gives the same
// protection and the same value to round-off (<= 2 ulp) for a tenth of the instructions.
template <int N> LGH_HD double norml2(const double *v)
{
   double mx = 0.0;
#pragma unroll
   for (int i = 0; i < N; i++) { mx = fmax(mx, fabs(v[i])); }
   if (!(mx > 0.0)) { return 0.0; }
   double inv;
   const double mult = scaling_factor(mx, inv);
   double sum = 0.0;
#pragma unroll
   for (int i = 0; i < N; i++) { const double e = v[i] * inv; sum = fma(e, e, sum); }
   return mult * fsqrt(sum);
}
template <int N> LGH_HD double trace(const double *A)
{
   double t = 0.0;
#pragma unroll
   for (int i = 0; i < N; i++) { t += A[i + i * N]; }
   return t;
}
// Frobenius norm scaled by (a power of two near) the max entry (laghos_solver.cpp:997-1040)
template <int N> LGH_HD double fnorm(const double *A)
{
   double mx = 0.0;
#pragma unroll
   for (int i = 0; i < N * N; i++) { mx = fmax(mx, fabs(A[i])); }
   if (!(mx > 0.0)) { return 0.0; }
   double inv;
   const double mult = scaling_factor(mx, inv);
   double f2 = 0.0;
#pragma unroll
   for (int i = 0; i < N * N; i++) { const double e = A[i] * inv; f2 = fma(e, e, f2); }
   return mult * fsqrt(f2);
}

// sqrt(a^2 + b^2) for operands already scaled to O(1) (no overflow guard needed)
LGH_HD double hypot_scaled(const double a, const double b) { return fsqrt(a * a + b * b); }

// Relative size below which the deviatoric part of a symmetric 3x3 matrix is
// round-off of its isotropic part: Q <= (8 eps)^2 (tr/3)^2 perturbs eigenvalues
// by <= 2e-15 relative.  In the singular-VALUE routine, treating it as Q = 0 keeps
// whole wavefronts on the cheap path over undisturbed mesh (J = h I + round-off),
// where otherwise random signs of R/Q^1.5 send some lane of every wave through
// every branch (measured: 4300 VALU instructions per wave).
#define LGH_SM_QTINY 3.2e-30

// cos(acos(t)/3) for t in [-0.9, 1], i.e. the largest root c of 4c^3 - 3c = t (c in [0.62, 1]):
// Newton's method from a quadratic fit (error <= 0.023; the cubic is convex for c > 0 and its
// slope is >= 1.6 on the range, so the fourth step is at round-off: <= 2 ulp against the
// library pair, checked over the range on the host).  The characteristic-cubic roots of the
// eigenvalue / singular-value routines need nothing else, and an acos plus a cos call are
// 93 + 148 VALU instructions against ~25 here.
LGH_HD double cos_third_acos(const double t)
{
   double c = fma(t, fma(t, -0.0709, 0.2049), 0.8660254037844386);
#pragma unroll
   for (int i = 0; i < 4; i++)
   {
      const double c2 = c * c;
      const double p = fma(c, fma(4.0, c2, -3.0), -t);
      const double dp = fma(12.0, c2, -3.0);
#if defined(__HIP_DEVICE_COMPILE__)
      c = fma(-p, __builtin_amdgcn_rcp(dp), c); // Newton corrects the reciprocal's last bits
#else
      c -= p / dp;
#endif
   }
   return c;
}

// Jacobi rotation (c,s) for [d1 d12; d12 d2]; d1,d2 become the eigenvalues.
LGH_HD void eigensystem2s(const double d12, double &d1, double &d2, double &c, double &s)
{
   if (d12 != 0.)
   {
      const double sqrt_1_eps = 67108864.0; // sqrt(1/DBL_EPSILON) = 2^26
      double t;
      const double zeta = (d2 - d1) / (2 * d12);
      const double az = fabs(zeta);
      if (az < sqrt_1_eps) { t = copysign(1. / (az + fsqrt(1. + zeta * zeta)), zeta); }
      else { t = copysign(0.5 / az, zeta); }
      c = fsqrt(1. / (1. + t * t));
      s = c * t;
      t *= d12;
      d1 -= t;
      d2 += t;
   }
   else
   {
      c = 1.;
      s = 0.;
   }
}

LGH_HD void normalize3_lead(const double x1, const double x2, const double x3, double &n1, double &n2,
                            double &n3)
{
   const double m = fabs(x1);
   double r = x2 / m;
   double t = 1. + r * r;
   r = x3 / m;
   t = fsqrt(1. / (t + r * r));
   n1 = copysign(t, x1);
   t /= m;
   n2 = x2 * t;
   n3 = x3 * t;
}
LGH_HD void normalize3(const double x1, const double x2, const double x3, double &n1, double &n2,
                       double &n3)
{
   // The leading (largest) entry is picked with value selects and normalize3_lead is
   // instantiated once: three inlined copies with permuted reference arguments are
   // merged by the compiler into one body working on SELECTED POINTERS, which keeps
   // the operands in scratch memory (91 scratch instructions in the fused QUpdate).
   const double a1 = fabs(x1), a2 = fabs(x2), a3 = fabs(x3);
   const int lead = (a1 >= a2 && a1 >= a3) ? 0 : ((a1 < a2 && a2 >= a3) ? 1 : 2);
   const double xl = (lead == 0) ? x1 : ((lead == 1) ? x2 : x3);
   const double xa = (lead == 0) ? x2 : x1;
   const double xb = (lead == 2) ? x2 : x3;
   double ml = 0., ma = 0., mb = 0.;
   if (xl != 0.) { normalize3_lead(xl, xa, xb, ml, ma, mb); } // xl == 0 only if all are
   n1 = (lead == 0) ? ml : ma;
   n2 = (lead == 0) ? ma : ((lead == 1) ? ml : mb);
   n3 = (lead == 2) ? ml : mb;
}

// near-kernel vector of the general 2x2 [d1 d12; d21 d2] (pivoted Householder
// QR), returned in (d1,d2) with |z1|+|z2| = 1; 0 for the zero matrix.
LGH_HD int kernel_vector_2g(const int mode, double &d1, double &d12, double &d21, double &d2)
{
   double n1 = fabs(d1) + fabs(d21);
   double n2 = fabs(d2) + fabs(d12);
   const bool swap_columns = (n2 > n1);
   double mu;
   if (!swap_columns)
   {
      if (n1 == 0.) { return 0; }
      const bool sw = (mode == 0) ? (fabs(d1) > fabs(d21)) : (fabs(d1) < fabs(d21));
      if (sw) { swap2(d1, d21); swap2(d12, d2); }
   }
   else
   {
      const bool sw = (mode == 0) ? (fabs(d12) > fabs(d2)) : (fabs(d12) < fabs(d2));
      if (sw) { swap2(d1, d2); swap2(d12, d21); }
      else { swap2(d1, d12); swap2(d21, d2); }
   }
   n1 = hypot_scaled(d1, d21);
   if (d21 != 0.)
   {
      mu = copysign(n1, d1);
      n1 = -d21 * (d21 / (d1 + mu));
      d1 = mu;
      if (fabs(n1) <= fabs(d21))
      {
         n1 = n1 / d21;
         mu = (2. / (1. + n1 * n1)) * (n1 * d12 + d2);
         d2 = d2 - mu;
         d12 = d12 - mu * n1;
      }
      else
      {
         n2 = d21 / n1;
         mu = (2. / (1. + n2 * n2)) * (d12 + n2 * d2);
         d2 = d2 - mu * n2;
         d12 = d12 - mu;
      }
   }
   mu = -d12 / d1;
   n2 = 1. / (1. + fabs(mu));
   if (fabs(d1) <= n2 * fabs(d2))
   {
      d2 = 0.;
      d1 = 1.;
   }
   else
   {
      d2 = n2;
      d1 = mu * n2;
   }
   if (swap_columns) { swap2(d1, d2); }
   return 1;
}

// general 3x3 with dominant first column: Householder on column 1 + 2x2 problem
LGH_HD int kernel_vector_3g(const int mode, double &d1, double &d2, double &d3, double &c12,
                            double &c13, double &c23, double &c21, double &c31, double &c32)
{
   int kdim;
   double mu, n1, n2, n3, s1, s2, s3;
   s1 = hypot_scaled(c21, c31);
   n1 = hypot_scaled(d1, s1);
   if (s1 != 0.)
   {
      mu = copysign(n1, d1);
      n1 = -s1 * (s1 / (d1 + mu));
      d1 = mu;
      const double a1 = fabs(n1), a2 = fabs(c21), a3 = fabs(c31);
      if (a1 >= a2 && a1 >= a3)
      {
         s2 = c21 / n1;
         s3 = c31 / n1;
         mu = 2. / (1. + s2 * s2 + s3 * s3);
         n2 = mu * (c12 + s2 * d2 + s3 * c32);
         n3 = mu * (c13 + s2 * c23 + s3 * d3);
         c12 = c12 - n2;
         d2 = d2 - s2 * n2;
         c32 = c32 - s3 * n2;
         c13 = c13 - n3;
         c23 = c23 - s2 * n3;
         d3 = d3 - s3 * n3;
      }
      else if (a1 < a2 && a2 >= a3)
      {
         s1 = n1 / c21;
         s3 = c31 / c21;
         mu = 2. / (1. + s1 * s1 + s3 * s3);
         n2 = mu * (s1 * c12 + d2 + s3 * c32);
         n3 = mu * (s1 * c13 + c23 + s3 * d3);
         c12 = c12 - s1 * n2;
         d2 = d2 - n2;
         c32 = c32 - s3 * n2;
         c13 = c13 - s1 * n3;
         c23 = c23 - n3;
         d3 = d3 - s3 * n3;
      }
      else
      {
         s1 = n1 / c31;
         s2 = c21 / c31;
         mu = 2. / (1. + s1 * s1 + s2 * s2);
         n2 = mu * (s1 * c12 + s2 * d2 + c32);
         n3 = mu * (s1 * c13 + s2 * c23 + d3);
         c12 = c12 - s1 * n2;
         d2 = d2 - s2 * n2;
         c32 = c32 - n2;
         c13 = c13 - s1 * n3;
         c23 = c23 - s2 * n3;
         d3 = d3 - n3;
      }
   }
   if (kernel_vector_2g(mode, d2, c23, c32, d3))
   {
      d1 = -(c12 * d2 + c13 * d3) / d1;
      kdim = 1;
   }
   else
   {
      d2 = c12 / d1;
      d3 = c13 / d1;
      d1 = 1.;
      kdim = 2;
   }
   normalize3(d1, d2, d3, d1, d2, d3);
   return kdim;
}

// unit near-kernel vector of symmetric [d1 d12 d13; . d2 d23; . . d3] in (d1,d2,d3);
// returns kernel dimension (3 = zero matrix, vector undefined)
LGH_HD int kernel_vector_3s(const int mode, const double d12, const double d13, const double d23,
                            double &d1, double &d2, double &d3)
{
   double c12 = d12, c13 = d13, c23 = d23;
   const double l1 = fabs(d1) + fabs(c12) + fabs(c13);
   const double l2 = fabs(d2) + fabs(c12) + fabs(c23);
   const double l3 = fabs(d3) + fabs(c13) + fabs(c23);
   int col;
   if (l1 >= l3) { col = (l1 >= l2) ? 1 : 2; }
   else { col = (l2 >= l3) ? 2 : 3; }
   if (col == 1) { if (l1 == 0.) { return 3; } }
   else if (col == 2)
   {
      if (l2 == 0.) { return 3; }
      swap2(c13, c23);
      swap2(d1, d2);
   }
   else
   {
      if (l3 == 0.) { return 3; }
      swap2(c12, c23);
      swap2(d1, d3);
   }
   int row;
   if (mode == 0)
   {
      if (fabs(d1) <= fabs(c13)) { row = (fabs(d1) <= fabs(c12)) ? 1 : 2; }
      else { row = (fabs(c12) <= fabs(c13)) ? 2 : 3; }
   }
   else
   {
      if (fabs(d1) >= fabs(c13)) { row = (fabs(d1) >= fabs(c12)) ? 1 : 2; }
      else { row = (fabs(c12) >= fabs(c13)) ? 2 : 3; }
   }
   const double s11 = d1, s12 = c12, s13 = c13, s22 = d2, s23 = c23, s33 = d3;
   double c21, c31, c32;
   if (row == 1)
   {
      d1 = s11; c12 = s12; c13 = s13;
      c21 = s12; d2 = s22; c23 = s23;
      c31 = s13; c32 = s23; d3 = s33;
   }
   else if (row == 2)
   {
      d1 = s12; c12 = s22; c13 = s23;
      c21 = s11; d2 = s12; c23 = s13;
      c31 = s13; c32 = s23; d3 = s33;
   }
   else
   {
      d1 = s13; c12 = s23; c13 = s33;
      c21 = s12; d2 = s22; c23 = s23;
      c31 = s11; c32 = s12; d3 = s13;
   }
   const int kdim = kernel_vector_3g(mode, d1, d2, d3, c12, c13, c23, c21, c31, c32);
   if (col == 2) { swap2(d1, d2); }
   else if (col == 3) { swap2(d1, d3); }
   return kdim;
}

// deflate symmetric A with unit eigenvector z: B = Q P A P Q = diag(b1,[b2 b23; b23 b3])
LGH_HD int reduce_3s(const int mode, double &d1, double &d2, double &d3, double &d12, double &d13,
                     double &d23, double &z1, double &z2, double &z3, double &v1, double &v2,
                     double &v3, double &g)
{
   int k;
   if (mode == 0)
   {
      if (fabs(z1) <= fabs(z3)) { k = (fabs(z1) <= fabs(z2)) ? 1 : 2; }
      else { k = (fabs(z2) <= fabs(z3)) ? 2 : 3; }
   }
   else
   {
      if (fabs(z1) >= fabs(z3)) { k = (fabs(z1) >= fabs(z2)) ? 1 : 2; }
      else { k = (fabs(z2) >= fabs(z3)) ? 2 : 3; }
   }
   if (k == 2)
   {
      swap2(d13, d23);
      swap2(d1, d2);
      swap2(z1, z2);
   }
   else if (k == 3)
   {
      swap2(d12, d23);
      swap2(d1, d3);
      swap2(z1, z3);
   }
   double s = hypot_scaled(z2, z3);
   if (s == 0.)
   {
      v1 = v2 = v3 = 0.;
      g = 1.;
   }
   else
   {
      g = copysign(1., z1);
      v1 = -s * (s / (z1 + g));
      g = fabs(v1);
      if (fabs(z2) > g) { g = fabs(z2); }
      if (fabs(z3) > g) { g = fabs(z3); }
      v1 = v1 / g;
      v2 = z2 / g;
      v3 = z3 / g;
      g = 2. / (v1 * v1 + v2 * v2 + v3 * v3);
      double w1 = g * (d1 * v1 + d12 * v2 + d13 * v3);
      double w2 = g * (d12 * v1 + d2 * v2 + d23 * v3);
      double w3 = g * (d13 * v1 + d23 * v2 + d3 * v3);
      s = (g / 2) * (v1 * w1 + v2 * w2 + v3 * w3);
      w1 -= s * v1;
      w2 -= s * v2;
      w3 -= s * v3;
      d1 -= 2 * v1 * w1;
      d2 -= 2 * v2 * w2;
      d23 -= v2 * w3 + v3 * w2;
      d3 -= 2 * v3 * w3;
   }
   if (k == 2) { swap2(z1, z2); }
   else if (k == 3) { swap2(z1, z3); }
   return k;
}

// Smallest eigenvalue of the symmetric matrix (upper triangle of the column-major
// data) and a corresponding eigenvector.  QUpdateBody only consumes lambda[0] and
// vec[0..DIM-1] (laghos_solver.cpp:1115, :1124), so only those are formed.
LGH_HD void min_eigenpair2(const double *data, double &lambda, double *vec)
{
   double d0 = data[0], d2 = data[2], d3 = data[3], c, s;
   eigensystem2s(d2, d0, d3, c, s);
   if (d0 <= d3)
   {
      lambda = d0;
      vec[0] = c;
      vec[1] = -s;
   }
   else
   {
      lambda = d3;
      vec[0] = s;
      vec[1] = c;
   }
}

LGH_HD void min_eigenpair3(const double *data, double &lambda, double *vec)
{
   double d11 = data[0], d12 = data[3], d22 = data[4];
   double d13 = data[6], d23 = data[7], d33 = data[8];
   double d_max = fabs(d11);
   if (d_max < fabs(d22)) { d_max = fabs(d22); }
   if (d_max < fabs(d33)) { d_max = fabs(d33); }
   if (d_max < fabs(d12)) { d_max = fabs(d12); }
   if (d_max < fabs(d13)) { d_max = fabs(d13); }
   if (d_max < fabs(d23)) { d_max = fabs(d23); }
   double imult;
   const double mult = scaling_factor(d_max, imult);
   d11 *= imult; d22 *= imult; d33 *= imult;
   d12 *= imult; d13 *= imult; d23 *= imult;
   double aa = (d11 + d22 + d33) / 3;
   double c1 = d11 - aa, c2 = d22 - aa, c3 = d33 - aa;
   const double Q = (2 * (d12 * d12 + d13 * d13 + d23 * d23) + c1 * c1 + c2 * c2 + c3 * c3) / 6;
   double R = (c1 * (d23 * d23 - c2 * c3) + d12 * (d12 * c3 - 2 * d13 * d23) + d13 * d13 * c2) / 2;
   bool triple = (Q <= 0.); // no tiny-Q shortcut here: the eigenVECTOR is not continuous in Q
   if (!triple)
   {
      const double sqrtQ = fsqrt(Q);
      const double sqrtQ3 = Q * sqrtQ;
      double r;
      if (fabs(R) >= sqrtQ3) { r = (R < 0.) ? 2 * sqrtQ : -2 * sqrtQ; }
      else
      {
         R = R / sqrtQ3;
         // reference: cos(acos(R)/3) for R >= 0, cos((acos(R) + 2 pi)/3) for R < 0; the second
         // is -cos(acos(-R)/3), so both are the isolated root +-cos_third_acos(|R|)
         r = -2 * sqrtQ * copysign(cos_third_acos(fabs(R)), R);
      }
      aa += r;
      c1 = d11 - aa;
      c2 = d22 - aa;
      c3 = d33 - aa;
      if (kernel_vector_3s(0, d12, d13, d23, c1, c2, c3) == 3) { triple = true; }
      else
      {
         double v1, v2, v3, g;
         const int k = reduce_3s(0, d11, d22, d33, d12, d13, d23, c1, c2, c3, v1, v2, v3, g);
         double c, s;
         eigensystem2s(d23, d22, d33, c, s);
         // candidates: d11 <-> (c1,c2,c3); d22 <-> P Q (0,c,-s); d33 <-> P Q (0,s,c)
         int which; // 1, 2 or 3: the smallest with the reference's tie-breaking
         if (d11 <= d22) { which = (d22 <= d33) ? 1 : ((d11 <= d33) ? 1 : 3); }
         else { which = (d11 <= d33) ? 2 : ((d22 <= d33) ? 2 : 3); }
         if (which == 1)
         {
            lambda = d11;
            vec[0] = c1;
            vec[1] = c2;
            vec[2] = c3;
         }
         else
         {
            double w0, w1, w2;
            if (which == 2)
            {
               const double t = g * (v2 * c - v3 * s);
               lambda = d22;
               w0 = -v1 * t;
               w1 = c - v2 * t;
               w2 = -s - v3 * t;
            }
            else
            {
               const double t = g * (v2 * s + v3 * c);
               lambda = d33;
               w0 = -v1 * t;
               w1 = s - v2 * t;
               w2 = c - v3 * t;
            }
            if (k == 2) { swap2(w0, w1); }
            else if (k == 3) { swap2(w0, w2); }
            vec[0] = w0;
            vec[1] = w1;
            vec[2] = w2;
         }
      }
   }
   if (triple)
   {
      lambda = aa;
      vec[0] = 1.;
      vec[1] = 0.;
      vec[2] = 0.;
   }
   lambda *= mult;
}
template <int DIM> LGH_HD void min_eigenpair(const double *A, double &lambda, double *vec)
{
   if (DIM == 2) { min_eigenpair2(A, lambda, vec); }
   else { min_eigenpair3(A, lambda, vec); }
}

// smallest singular value
LGH_HD double min_singular2(const double *data)
{
   double d0 = data[0], d1 = data[1], d2 = data[2], d3 = data[3];
   double d_max = fabs(d0);
   if (d_max < fabs(d1)) { d_max = fabs(d1); }
   if (d_max < fabs(d2)) { d_max = fabs(d2); }
   if (d_max < fabs(d3)) { d_max = fabs(d3); }
   double imult;
   const double mult = scaling_factor(d_max, imult);
   d0 *= imult; d1 *= imult; d2 *= imult; d3 *= imult;
   double t = 0.5 * ((d0 + d2) * (d0 - d2) + (d1 - d3) * (d1 + d3));
   double s = d0 * d2 + d1 * d3;
   s = fsqrt(0.5 * (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) + fsqrt(t * t + s * s));
   if (s == 0.0) { return 0.0; }
   t = fabs(d0 * d3 - d1 * d2) / s;
   return (t > s) ? s * mult : t * mult;
}

LGH_HD double min_singular3(const double *data)
{
   double d0 = data[0], d1 = data[1], d2 = data[2];
   double d3 = data[3], d4 = data[4], d5 = data[5];
   double d6 = data[6], d7 = data[7], d8 = data[8];
   double d_max = fabs(d0);
   if (d_max < fabs(d1)) { d_max = fabs(d1); }
   if (d_max < fabs(d2)) { d_max = fabs(d2); }
   if (d_max < fabs(d3)) { d_max = fabs(d3); }
   if (d_max < fabs(d4)) { d_max = fabs(d4); }
   if (d_max < fabs(d5)) { d_max = fabs(d5); }
   if (d_max < fabs(d6)) { d_max = fabs(d6); }
   if (d_max < fabs(d7)) { d_max = fabs(d7); }
   if (d_max < fabs(d8)) { d_max = fabs(d8); }
   double imult;
   const double mult = scaling_factor(d_max, imult);
   d0 *= imult; d1 *= imult; d2 *= imult;
   d3 *= imult; d4 *= imult; d5 *= imult;
   d6 *= imult; d7 *= imult; d8 *= imult;
   double b11 = d0 * d0 + d1 * d1 + d2 * d2;
   double b12 = d0 * d3 + d1 * d4 + d2 * d5;
   double b13 = d0 * d6 + d1 * d7 + d2 * d8;
   double b22 = d3 * d3 + d4 * d4 + d5 * d5;
   double b23 = d3 * d6 + d4 * d7 + d5 * d8;
   double b33 = d6 * d6 + d7 * d7 + d8 * d8;
   double aa = (b11 + b22 + b33) / 3;
   double c1, c2, c3;
   {
      const double b11_b22 = ((d0 - d3) * (d0 + d3) + (d1 - d4) * (d1 + d4) + (d2 - d5) * (d2 + d5));
      const double b22_b33 = ((d3 - d6) * (d3 + d6) + (d4 - d7) * (d4 + d7) + (d5 - d8) * (d5 + d8));
      const double b33_b11 = ((d6 - d0) * (d6 + d0) + (d7 - d1) * (d7 + d1) + (d8 - d2) * (d8 + d2));
      c1 = (b11_b22 - b33_b11) / 3;
      c2 = (b22_b33 - b11_b22) / 3;
      c3 = (b33_b11 - b22_b33) / 3;
   }
   const double Q = (2 * (b12 * b12 + b13 * b13 + b23 * b23) + c1 * c1 + c2 * c2 + c3 * c3) / 6;
   double R = (c1 * (b23 * b23 - c2 * c3) + b12 * (b12 * c3 - 2 * b13 * b23) + b13 * b13 * c2) / 2;
   if (Q > LGH_SM_QTINY * (aa * aa))
   {
      const double sqrtQ = fsqrt(Q);
      const double sqrtQ3 = Q * sqrtQ;
      double r = 0.;
      bool have = false;
      if (fabs(R) >= sqrtQ3) { r = (R < 0.) ? 2 * sqrtQ : -2 * sqrtQ; }
      else
      {
         R = R / sqrtQ3;
         const bool mid = (fabs(R) <= 0.9), neg = (R < 0.);
         // mid: cos(acos(R)/3); else the isolated root, cos((acos(R) + 2 pi)/3) = -cos(acos(-R)/3)
         // for R < 0 and cos(acos(R)/3) for R > 0
         const double cta = cos_third_acos(mid ? R : fabs(R));
         const double cs = (!mid && neg) ? -cta : cta;
         if (mid)
         {
            aa -= 2 * sqrtQ * cs; // min root directly
            have = true;
         }
         else if (neg) { r = -2 * sqrtQ * cs; } // max is isolated
         else
         {
            r = -2 * sqrtQ * cs; // min is isolated
            aa += r;
            have = true;
         }
      }
      if (!have)
      {
         c1 -= r;
         c2 -= r;
         c3 -= r;
         if (kernel_vector_3s(1, b12, b13, b23, c1, c2, c3) == 3) { aa += r; }
         else
         {
            double v1, v2, v3, g;
            reduce_3s(1, b11, b22, b33, b12, b13, b23, c1, c2, c3, v1, v2, v3, g);
            double c, s;
            eigensystem2s(b23, b22, b33, c, s);
            aa = fmin(fmin(b11, b22), b33);
         }
      }
   }
   return fsqrt(fabs(aa)) * mult;
}
template <int DIM> LGH_HD double min_singular(const double *J)
{
   return DIM == 2 ? min_singular2(J) : min_singular3(J);
}

} // namespace sm
} // namespace lgh
