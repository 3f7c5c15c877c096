// ORACLE (test infrastructure only — never linked into the product path).
//
// Small dense-matrix kernels used by the quadrature-point update, restated on
// the CPU.  The reference calls these as mfem::kernels::{Det, CalcInverse, Mult,
// MultABt, Symmetrize, CalcEigenvalues, CalcSingularvalue, Norml2, Add}
// (call sites: /root/reference/laghos_solver.cpp:1078-1080, :1095, :1105,
// :1113, :1117-1121, :1133, :1139, :1158).  Their bodies live in upstream MFEM
// (linalg/kernels.hpp, MFEM `master`, unpinned: makefile:307-314), which is NOT
// part of /root/reference and not installed here, so the algorithms below are
// restated from the published MFEM algorithm (scaled characteristic-polynomial
// root + Householder deflation for the 3x3 symmetric eigenproblem, Parlett's
// 2x2 rotation, singular values as sqrt of the eigenvalues of J^T J).  Parity of
// this file is pinned only end-to-end through the reference's `--checks` |e|
// table (laghos.cpp:1441-1463) and README runs; tests/ additionally compare
// against numpy.linalg.{eigh,svd}.
//
// All matrices are column-major: A(i,j) = a[i + n*j].
#pragma once
#include <cmath>
#include <limits>

namespace sm
{

template <typename T> static inline void Swap(T &a, T &b) { T t = a; a = b; b = t; }

template <int DIM> static inline double Det(const double *J);
template <> inline double Det<2>(const double *J) { return J[0] * J[3] - J[1] * J[2]; }
template <> inline double Det<3>(const double *J)
{
   return J[0] * (J[4] * J[8] - J[5] * J[7]) + J[3] * (J[2] * J[7] - J[1] * J[8]) +
          J[6] * (J[1] * J[5] - J[2] * J[4]);
}

// inverse = adjugate / det
template <int DIM> static inline void CalcInverse(const double *J, double *Ji);
template <> inline void CalcInverse<2>(const double *J, double *Ji)
{
   const double d = 1.0 / Det<2>(J);
   Ji[0] = J[3] * d;
   Ji[1] = -J[1] * d;
   Ji[2] = -J[2] * d;
   Ji[3] = J[0] * d;
}
template <> inline void CalcInverse<3>(const double *J, double *Ji)
{
   const double d = 1.0 / Det<3>(J);
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

// C(h x w) = A(h x k) * B(k x w)
static inline void Mult(int h, int w, int k, const double *A, const double *B, double *C)
{
   for (int j = 0; j < w; j++)
      for (int i = 0; i < h; i++)
      {
         double s = 0.0;
         for (int l = 0; l < k; l++) { s += A[i + h * l] * B[l + k * j]; }
         C[i + h * j] = s;
      }
}
// y(h) = A(h x w) x(w)
static inline void MultV(int h, int w, const double *A, const double *x, double *y)
{
   for (int i = 0; i < h; i++)
   {
      double s = 0.0;
      for (int j = 0; j < w; j++) { s += A[i + h * j] * x[j]; }
      y[i] = s;
   }
}
// C(ah x bh) = A(ah x aw) * B(bh x aw)^T
static inline void MultABt(int ah, int aw, int bh, const double *A, const double *B, double *C)
{
   for (int j = 0; j < bh; j++)
      for (int i = 0; i < ah; i++)
      {
         double s = 0.0;
         for (int l = 0; l < aw; l++) { s += A[i + ah * l] * B[j + bh * l]; }
         C[i + ah * j] = s;
      }
}
static inline void Symmetrize(int n, double *A)
{
   for (int i = 0; i < n; i++)
      for (int j = 0; j < i; j++)
      {
         const double a = 0.5 * (A[i + n * j] + A[j + n * i]);
         A[i + n * j] = A[j + n * i] = a;
      }
}
static inline double Norml2(int n, const double *v)
{
   // scaled 2-norm (overflow safe), as upstream Vector::Norml2
   if (n == 0) { return 0.0; }
   if (n == 1) { return std::fabs(v[0]); }
   double scale = 0.0, sum = 0.0;
   for (int i = 0; i < n; i++)
   {
      if (v[i] != 0.0)
      {
         const double a = std::fabs(v[i]);
         if (scale <= a)
         {
            const double r = scale / a;
            sum = 1.0 + sum * (r * r);
            scale = a;
         }
         else
         {
            const double r = a / scale;
            sum += r * r;
         }
      }
   }
   return scale * std::sqrt(sum);
}
// C = A + alpha*B
static inline void Add(int h, int w, double alpha, const double *A, const double *B, double *C)
{
   for (int i = 0; i < h * w; i++) { C[i] = A[i] + alpha * B[i]; }
}

// d_max in [0.5,1)*mult with mult a power of two
static inline void GetScalingFactor(const double d_max, double &mult)
{
   int d_exp;
   if (d_max > 0.)
   {
      mult = std::frexp(d_max, &d_exp);
      if (d_exp == std::numeric_limits<double>::max_exponent)
      {
         mult *= std::numeric_limits<double>::radix;
      }
      mult = d_max / mult;
   }
   else { mult = 1.; }
}

// Parlett, "The Symmetric Eigenvalue Problem", pp.189-190: rotation (c,s)
// diagonalising [d1 d12; d12 d2]; on return d1,d2 hold the eigenvalues.
static inline void Eigensystem2S(const double d12, double &d1, double &d2, double &c, double &s)
{
   const double sqrt_1_eps = std::sqrt(1. / std::numeric_limits<double>::epsilon());
   if (d12 != 0.)
   {
      double t;
      const double zeta = (d2 - d1) / (2 * d12);
      const double azeta = std::fabs(zeta);
      if (azeta < sqrt_1_eps) { t = std::copysign(1. / (azeta + std::sqrt(1. + zeta * zeta)), zeta); }
      else { t = std::copysign(0.5 / azeta, zeta); }
      c = std::sqrt(1. / (1. + t * t));
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
static inline void Eigenvalues2S(const double d12, double &d1, double &d2)
{
   double c, s;
   Eigensystem2S(d12, d1, d2, c, s);
}

static inline void Vec_normalize3_aux(const double x1, const double x2, const double x3, double &n1,
                                      double &n2, double &n3)
{
   // |x1| is the largest entry
   const double m = std::fabs(x1);
   double r = x2 / m;
   double t = 1. + r * r;
   r = x3 / m;
   t = std::sqrt(1. / (t + r * r));
   n1 = std::copysign(t, x1);
   t /= m;
   n2 = x2 * t;
   n3 = x3 * t;
}
static inline void Vec_normalize3(const double x1, const double x2, const double x3, double &n1,
                                  double &n2, double &n3)
{
   if (std::fabs(x1) >= std::fabs(x2))
   {
      if (std::fabs(x1) >= std::fabs(x3))
      {
         if (x1 != 0.) { Vec_normalize3_aux(x1, x2, x3, n1, n2, n3); }
         else { n1 = n2 = n3 = 0.; }
         return;
      }
   }
   else if (std::fabs(x2) >= std::fabs(x3))
   {
      Vec_normalize3_aux(x2, x1, x3, n2, n1, n3);
      return;
   }
   Vec_normalize3_aux(x3, x1, x2, n3, n1, n2);
}

// Vector (z1,z2) in the near-kernel of [d1 d12; d21 d2] by pivoted Householder
// QR; returned in (d1,d2) with |z1|+|z2| = 1.  Returns 0 for the zero matrix.
static inline int KernelVector2G(const int mode, double &d1, double &d12, double &d21, double &d2)
{
   double n1 = std::fabs(d1) + std::fabs(d21);
   double n2 = std::fabs(d2) + std::fabs(d12);
   const bool swap_columns = (n2 > n1);
   double mu;
   if (!swap_columns)
   {
      if (n1 == 0.) { return 0; }
      if (mode == 0)
      {
         if (std::fabs(d1) > std::fabs(d21)) { Swap(d1, d21); Swap(d12, d2); }
      }
      else
      {
         if (std::fabs(d1) < std::fabs(d21)) { Swap(d1, d21); Swap(d12, d2); }
      }
   }
   else
   {
      if (mode == 0)
      {
         if (std::fabs(d12) > std::fabs(d2)) { Swap(d1, d2); Swap(d12, d21); }
         else { Swap(d1, d12); Swap(d21, d2); }
      }
      else
      {
         if (std::fabs(d12) < std::fabs(d2)) { Swap(d1, d2); Swap(d12, d21); }
         else { Swap(d1, d12); Swap(d21, d2); }
      }
   }
   n1 = std::hypot(d1, d21);
   if (d21 != 0.)
   {
      // Householder: Q (d1,d21)^t = (mu,0)^t
      mu = std::copysign(n1, d1);
      n1 = -d21 * (d21 / (d1 + mu)); // = d1 - mu
      d1 = mu;
      if (std::fabs(n1) <= std::fabs(d21))
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
   // solve [d1 d12; 0 d2] z = 0 approximately, |z1|+|z2| = 1
   mu = -d12 / d1;
   n2 = 1. / (1. + std::fabs(mu));
   if (std::fabs(d1) <= n2 * std::fabs(d2))
   {
      d2 = 0.;
      d1 = 1.;
   }
   else
   {
      d2 = n2;
      d1 = mu * n2;
   }
   if (swap_columns) { Swap(d1, d2); }
   return 1;
}

// General 3x3 [d1 c12 c13; c21 d2 c23; c31 c32 d3] whose first column has the
// largest norm: Householder on column 1, then the 2x2 problem.
static inline int KernelVector3G_aux(const int mode, double &d1, double &d2, double &d3, double &c12,
                                     double &c13, double &c23, double &c21, double &c31,
                                     double &c32)
{
   int kdim;
   double mu, n1, n2, n3, s1, s2, s3;
   s1 = std::hypot(c21, c31);
   n1 = std::hypot(d1, s1);
   if (s1 != 0.)
   {
      mu = std::copysign(n1, d1);
      n1 = -s1 * (s1 / (d1 + mu)); // = d1 - mu
      d1 = mu;
      if (std::fabs(n1) >= std::fabs(c21) && std::fabs(n1) >= std::fabs(c31))
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
      else if (std::fabs(c21) >= std::fabs(c31))
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
   if (KernelVector2G(mode, d2, c23, c32, d3))
   {
      // back-substitute for z1
      d1 = -(c12 * d2 + c13 * d3) / d1;
      kdim = 1;
   }
   else
   {
      // 2-dimensional kernel: return the vector orthogonal to it
      d2 = c12 / d1;
      d3 = c13 / d1;
      d1 = 1.;
      kdim = 2;
   }
   Vec_normalize3(d1, d2, d3, d1, d2, d3);
   return kdim;
}

// Unit vector in the near-kernel of the symmetric [d1 d12 d13; d12 d2 d23;
// d13 d23 d3], returned in (d1,d2,d3).  Returns the kernel dimension (never 0):
// 3 -> zero matrix (vector undefined), 2 -> vector orthogonal to the kernel.
static inline int KernelVector3S(const int mode, const double d12, const double d13, const double d23,
                                 double &d1, double &d2, double &d3)
{
   double c12 = d12, c13 = d13, c23 = d23;
   double c21, c31, c32;
   int col, row;
   // l1 norms of columns 1,2,3
   c32 = std::fabs(d1) + std::fabs(c12) + std::fabs(c13);
   c31 = std::fabs(d2) + std::fabs(c12) + std::fabs(c23);
   c21 = std::fabs(d3) + std::fabs(c13) + std::fabs(c23);
   if (c32 >= c21) { col = (c32 >= c31) ? 1 : 2; }
   else { col = (c31 >= c21) ? 2 : 3; }
   // symmetric permutation 1 <-> col
   switch (col)
   {
      case 1:
         if (c32 == 0.) { return 3; }
         break;
      case 2:
         if (c31 == 0.) { return 3; }
         Swap(c13, c23);
         Swap(d1, d2);
         break;
      case 3:
         if (c21 == 0.) { return 3; }
         Swap(c12, c23);
         Swap(d1, d3);
   }
   // row pivoting within column 1 = (d1, c12, c13)
   if (mode == 0)
   {
      if (std::fabs(d1) <= std::fabs(c13)) { row = (std::fabs(d1) <= std::fabs(c12)) ? 1 : 2; }
      else { row = (std::fabs(c12) <= std::fabs(c13)) ? 2 : 3; }
   }
   else
   {
      if (std::fabs(d1) >= std::fabs(c13)) { row = (std::fabs(d1) >= std::fabs(c12)) ? 1 : 2; }
      else { row = (std::fabs(c12) >= std::fabs(c13)) ? 2 : 3; }
   }
   // general matrix G = rows of the symmetric matrix, with rows 1 <-> row
   // swapped.  Symmetric S = [d1 c12 c13; c12 d2 c23; c13 c23 d3].
   double g11, g12, g13, g21, g22, g23, g31, g32, g33;
   const double s11 = d1, s12 = c12, s13 = c13, s22 = d2, s23 = c23, s33 = d3;
   switch (row)
   {
      case 1:
         g11 = s11; g12 = s12; g13 = s13;
         g21 = s12; g22 = s22; g23 = s23;
         g31 = s13; g32 = s23; g33 = s33;
         break;
      case 2:
         g11 = s12; g12 = s22; g13 = s23;
         g21 = s11; g22 = s12; g23 = s13;
         g31 = s13; g32 = s23; g33 = s33;
         break;
      default:
         g11 = s13; g12 = s23; g13 = s33;
         g21 = s12; g22 = s22; g23 = s23;
         g31 = s11; g32 = s12; g33 = s13;
   }
   d1 = g11; d2 = g22; d3 = g33;
   c12 = g12; c13 = g13; c23 = g23;
   c21 = g21; c31 = g31; c32 = g32;
   row = KernelVector3G_aux(mode, d1, d2, d3, c12, c13, c23, c21, c31, c32);
   // undo the column permutation on the kernel vector
   switch (col)
   {
      case 2: Swap(d1, d2); break;
      case 3: Swap(d1, d3);
   }
   return row;
}

// With unit eigenvector z of the symmetric A, B = Q P A P Q = diag(b1, [b2 b23;
// b23 b3]); P swaps entries 1<->k, Q = I - g v v^t.  Returns k.
static inline int Reduce3S(const int mode, double &d1, double &d2, double &d3, double &d12,
                           double &d13, double &d23, double &z1, double &z2, double &z3, double &v1,
                           double &v2, double &v3, double &g)
{
   int k;
   double s, w1, w2, w3;
   if (mode == 0)
   {
      if (std::fabs(z1) <= std::fabs(z3)) { k = (std::fabs(z1) <= std::fabs(z2)) ? 1 : 2; }
      else { k = (std::fabs(z2) <= std::fabs(z3)) ? 2 : 3; }
   }
   else
   {
      if (std::fabs(z1) >= std::fabs(z3)) { k = (std::fabs(z1) >= std::fabs(z2)) ? 1 : 2; }
      else { k = (std::fabs(z2) >= std::fabs(z3)) ? 2 : 3; }
   }
   switch (k)
   {
      case 2:
         Swap(d13, d23);
         Swap(d1, d2);
         Swap(z1, z2);
         break;
      case 3:
         Swap(d12, d23);
         Swap(d1, d3);
         Swap(z1, z3);
   }
   s = std::hypot(z2, z3);
   if (s == 0.)
   {
      v1 = v2 = v3 = 0.;
      g = 1.;
   }
   else
   {
      g = std::copysign(1., z1);
      v1 = -s * (s / (z1 + g)); // = z1 - g
      g = std::fabs(v1);
      if (std::fabs(z2) > g) { g = std::fabs(z2); }
      if (std::fabs(z3) > g) { g = std::fabs(z3); }
      v1 = v1 / g;
      v2 = z2 / g;
      v3 = z3 / g;
      g = 2. / (v1 * v1 + v2 * v2 + v3 * v3);
      // Q A Q = A - v w^t - w v^t,  w = u - (g/2)(v^t u) v,  u = g A v
      w1 = g * (d1 * v1 + d12 * v2 + d13 * v3);
      w2 = g * (d12 * v1 + d2 * v2 + d23 * v3);
      w3 = g * (d13 * v1 + d23 * v2 + d3 * v3);
      s = (g / 2) * (v1 * w1 + v2 * w2 + v3 * w3);
      w1 -= s * v1;
      w2 -= s * v2;
      w3 -= s * v3;
      d1 -= 2 * v1 * w1;
      d2 -= 2 * v2 * w2;
      d23 -= v2 * w3 + v3 * w2;
      d3 -= 2 * v3 * w3;
   }
   switch (k)
   {
      case 2: Swap(z1, z2); break;
      case 3: Swap(z1, z3);
   }
   return k;
}

// Eigenvalues ascending in lambda[], eigenvector k in vec[k*DIM .. k*DIM+DIM-1].
// Uses the upper-triangular entries of the (symmetric, column-major) data.
template <int DIM> static inline void CalcEigenvalues(const double *data, double *lambda, double *vec);

template <> inline void CalcEigenvalues<2>(const double *data, double *lambda, double *vec)
{
   double d0 = data[0];
   double d2 = data[2];
   double d3 = data[3];
   double c, s;
   Eigensystem2S(d2, d0, d3, c, s);
   if (d0 <= d3)
   {
      lambda[0] = d0;
      lambda[1] = d3;
      vec[0] = c;
      vec[1] = -s;
      vec[2] = s;
      vec[3] = c;
   }
   else
   {
      lambda[0] = d3;
      lambda[1] = d0;
      vec[0] = s;
      vec[1] = c;
      vec[2] = c;
      vec[3] = -s;
   }
}

template <> inline void CalcEigenvalues<3>(const double *data, double *lambda, double *vec)
{
   double d11 = data[0];
   double d12 = data[3];
   double d22 = data[4];
   double d13 = data[6];
   double d23 = data[7];
   double d33 = data[8];
   double mult;
   {
      double d_max = std::fabs(d11);
      if (d_max < std::fabs(d22)) { d_max = std::fabs(d22); }
      if (d_max < std::fabs(d33)) { d_max = std::fabs(d33); }
      if (d_max < std::fabs(d12)) { d_max = std::fabs(d12); }
      if (d_max < std::fabs(d13)) { d_max = std::fabs(d13); }
      if (d_max < std::fabs(d23)) { d_max = std::fabs(d23); }
      GetScalingFactor(d_max, mult);
   }
   d11 /= mult; d22 /= mult; d33 /= mult;
   d12 /= mult; d13 /= mult; d23 /= mult;

   double aa = (d11 + d22 + d33) / 3; // tr(A)/3
   double c1 = d11 - aa;
   double c2 = d22 - aa;
   double c3 = d33 - aa;
   double Q, R;
   Q = (2 * (d12 * d12 + d13 * d13 + d23 * d23) + c1 * c1 + c2 * c2 + c3 * c3) / 6;
   R = (c1 * (d23 * d23 - c2 * c3) + d12 * (d12 * c3 - 2 * d13 * d23) + d13 * d13 * c2) / 2;

   bool identity = false;
   if (Q <= 0.) { identity = true; }
   else
   {
      const double sqrtQ = std::sqrt(Q);
      const double sqrtQ3 = Q * sqrtQ;
      double r;
      if (std::fabs(R) >= sqrtQ3) { r = (R < 0.) ? 2 * sqrtQ : -2 * sqrtQ; }
      else
      {
         R = R / sqrtQ3;
         if (R < 0.) { r = -2 * sqrtQ * std::cos((std::acos(R) + 2.0 * M_PI) / 3); } // max
         else { r = -2 * sqrtQ * std::cos(std::acos(R) / 3); }                      // min
      }
      aa += r; // the root best separated from the other two
      c1 = d11 - aa;
      c2 = d22 - aa;
      c3 = d33 - aa;
      const int mode = 0;
      if (KernelVector3S(mode, d12, d13, d23, c1, c2, c3) == 3) { identity = true; }
      else
      {
         double v1, v2, v3, g;
         const int k = Reduce3S(mode, d11, d22, d33, d12, d13, d23, c1, c2, c3, v1, v2, v3, g);
         double c, s;
         Eigensystem2S(d23, d22, d33, c, s);
         double *vec_1, *vec_2, *vec_3;
         if (d11 <= d22)
         {
            if (d22 <= d33)
            {
               lambda[0] = d11; vec_1 = vec;
               lambda[1] = d22; vec_2 = vec + 3;
               lambda[2] = d33; vec_3 = vec + 6;
            }
            else if (d11 <= d33)
            {
               lambda[0] = d11; vec_1 = vec;
               lambda[1] = d33; vec_3 = vec + 3;
               lambda[2] = d22; vec_2 = vec + 6;
            }
            else
            {
               lambda[0] = d33; vec_3 = vec;
               lambda[1] = d11; vec_1 = vec + 3;
               lambda[2] = d22; vec_2 = vec + 6;
            }
         }
         else
         {
            if (d11 <= d33)
            {
               lambda[0] = d22; vec_2 = vec;
               lambda[1] = d11; vec_1 = vec + 3;
               lambda[2] = d33; vec_3 = vec + 6;
            }
            else if (d22 <= d33)
            {
               lambda[0] = d22; vec_2 = vec;
               lambda[1] = d33; vec_3 = vec + 3;
               lambda[2] = d11; vec_1 = vec + 6;
            }
            else
            {
               lambda[0] = d33; vec_3 = vec;
               lambda[1] = d22; vec_2 = vec + 3;
               lambda[2] = d11; vec_1 = vec + 6;
            }
         }
         vec_1[0] = c1;
         vec_1[1] = c2;
         vec_1[2] = c3;
         d22 = g * (v2 * c - v3 * s);
         d33 = g * (v2 * s + v3 * c);
         vec_2[0] = -v1 * d22;     vec_3[0] = -v1 * d33;
         vec_2[1] = c - v2 * d22;  vec_3[1] = s - v2 * d33;
         vec_2[2] = -s - v3 * d22; vec_3[2] = c - v3 * d33;
         switch (k)
         {
            case 2:
               Swap(vec_2[0], vec_2[1]);
               Swap(vec_3[0], vec_3[1]);
               break;
            case 3:
               Swap(vec_2[0], vec_2[2]);
               Swap(vec_3[0], vec_3[2]);
         }
      }
   }
   if (identity)
   {
      lambda[0] = lambda[1] = lambda[2] = aa;
      vec[0] = 1.; vec[3] = 0.; vec[6] = 0.;
      vec[1] = 0.; vec[4] = 1.; vec[7] = 0.;
      vec[2] = 0.; vec[5] = 0.; vec[8] = 1.;
   }
   lambda[0] *= mult;
   lambda[1] *= mult;
   lambda[2] *= mult;
}

// i-th singular value, descending (i = DIM-1 is the minimum).
template <int DIM> static inline double CalcSingularvalue(const double *data, const int i);

template <> inline double CalcSingularvalue<2>(const double *data, const int i)
{
   double d0 = data[0], d1 = data[1], d2 = data[2], d3 = data[3];
   double mult;
   {
      double d_max = std::fabs(d0);
      if (d_max < std::fabs(d1)) { d_max = std::fabs(d1); }
      if (d_max < std::fabs(d2)) { d_max = std::fabs(d2); }
      if (d_max < std::fabs(d3)) { d_max = std::fabs(d3); }
      GetScalingFactor(d_max, mult);
   }
   d0 /= mult; d1 /= mult; d2 /= mult; d3 /= mult;
   double t = 0.5 * ((d0 + d2) * (d0 - d2) + (d1 - d3) * (d1 + d3));
   double s = d0 * d2 + d1 * d3;
   s = std::sqrt(0.5 * (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) + std::sqrt(t * t + s * s));
   if (s == 0.0) { return 0.0; }
   t = std::fabs(d0 * d3 - d1 * d2) / s;
   if (t > s) { return (i == 0) ? t * mult : s * mult; }
   return (i == 0) ? s * mult : t * mult;
}

template <> inline double CalcSingularvalue<3>(const double *data, const int i)
{
   double d0 = data[0], d1 = data[1], d2 = data[2];
   double d3 = data[3], d4 = data[4], d5 = data[5];
   double d6 = data[6], d7 = data[7], d8 = data[8];
   double mult;
   {
      double d_max = std::fabs(d0);
      if (d_max < std::fabs(d1)) { d_max = std::fabs(d1); }
      if (d_max < std::fabs(d2)) { d_max = std::fabs(d2); }
      if (d_max < std::fabs(d3)) { d_max = std::fabs(d3); }
      if (d_max < std::fabs(d4)) { d_max = std::fabs(d4); }
      if (d_max < std::fabs(d5)) { d_max = std::fabs(d5); }
      if (d_max < std::fabs(d6)) { d_max = std::fabs(d6); }
      if (d_max < std::fabs(d7)) { d_max = std::fabs(d7); }
      if (d_max < std::fabs(d8)) { d_max = std::fabs(d8); }
      GetScalingFactor(d_max, mult);
   }
   d0 /= mult; d1 /= mult; d2 /= mult;
   d3 /= mult; d4 /= mult; d5 /= mult;
   d6 /= mult; d7 /= mult; d8 /= mult;

   // B = J^t J
   double b11 = d0 * d0 + d1 * d1 + d2 * d2;
   double b12 = d0 * d3 + d1 * d4 + d2 * d5;
   double b13 = d0 * d6 + d1 * d7 + d2 * d8;
   double b22 = d3 * d3 + d4 * d4 + d5 * d5;
   double b23 = d3 * d6 + d4 * d7 + d5 * d8;
   double b33 = d6 * d6 + d7 * d7 + d8 * d8;

   double aa = (b11 + b22 + b33) / 3; // tr(B)/3
   double c1, c2, c3;
   {
      const double b11_b22 = ((d0 - d3) * (d0 + d3) + (d1 - d4) * (d1 + d4) + (d2 - d5) * (d2 + d5));
      const double b22_b33 = ((d3 - d6) * (d3 + d6) + (d4 - d7) * (d4 + d7) + (d5 - d8) * (d5 + d8));
      const double b33_b11 = ((d6 - d0) * (d6 + d0) + (d7 - d1) * (d7 + d1) + (d8 - d2) * (d8 + d2));
      c1 = (b11_b22 - b33_b11) / 3;
      c2 = (b22_b33 - b11_b22) / 3;
      c3 = (b33_b11 - b22_b33) / 3;
   }
   double Q, R;
   Q = (2 * (b12 * b12 + b13 * b13 + b23 * b23) + c1 * c1 + c2 * c2 + c3 * c3) / 6;
   R = (c1 * (b23 * b23 - c2 * c3) + b12 * (b12 * c3 - 2 * b13 * b23) + b13 * b13 * c2) / 2;

   if (Q <= 0.) { /* B = aa*I */ }
   else
   {
      const double sqrtQ = std::sqrt(Q);
      const double sqrtQ3 = Q * sqrtQ;
      double r;
      bool have_aa = false;
      if (std::fabs(R) >= sqrtQ3) { r = (R < 0.) ? 2 * sqrtQ : -2 * sqrtQ; }
      else
      {
         R = R / sqrtQ3;
         if (std::fabs(R) <= 0.9)
         {
            if (i == 2) { aa -= 2 * sqrtQ * std::cos(std::acos(R) / 3); }                       // min
            else if (i == 0) { aa -= 2 * sqrtQ * std::cos((std::acos(R) + 2.0 * M_PI) / 3); }  // max
            else { aa -= 2 * sqrtQ * std::cos((std::acos(R) - 2.0 * M_PI) / 3); }              // mid
            have_aa = true;
         }
         else if (R < 0.)
         {
            r = -2 * sqrtQ * std::cos((std::acos(R) + 2.0 * M_PI) / 3); // max
            if (i == 0) { aa += r; have_aa = true; }
         }
         else
         {
            r = -2 * sqrtQ * std::cos(std::acos(R) / 3); // min
            if (i == 2) { aa += r; have_aa = true; }
         }
      }
      if (!have_aa)
      {
         // (tr(B)/3 + r) is the isolated root; the wanted root is one of the two
         // close ones: deflate with its eigenvector and solve the 2x2 problem.
         c1 -= r;
         c2 -= r;
         c3 -= r;
         const int mode = 1;
         if (KernelVector3S(mode, b12, b13, b23, c1, c2, c3) == 3) { aa += r; }
         else
         {
            double v1, v2, v3, g;
            Reduce3S(mode, b11, b22, b33, b12, b13, b23, c1, c2, c3, v1, v2, v3, g);
            Eigenvalues2S(b23, b22, b33);
            if (i == 2) { aa = std::fmin(std::fmin(b11, b22), b33); }
            else if (i == 1)
            {
               if (b11 <= b22) { aa = (b22 <= b33) ? b22 : std::fmax(b11, b33); }
               else { aa = (b11 <= b33) ? b11 : std::fmax(b33, b22); }
            }
            else { aa = std::fmax(std::fmax(b11, b22), b33); }
         }
      }
   }
   return std::sqrt(std::fabs(aa)) * mult;
}

} // namespace sm
