// fem.cpp — see fem.hpp.  Host-side setup for the MI355X Laghos hot path.
#include "fem.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <stdexcept>

namespace laghos
{

// Legendre P_n(x) and derivative on [-1,1] by the three-term recurrence
static void Legendre(int n, double x, double &p, double &dp)
{
   if (n == 0) { p = 1.0; dp = 0.0; return; }
   double p0 = 1.0, p1 = x;
   for (int k = 2; k <= n; k++)
   {
      const double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
      p0 = p1;
      p1 = pk;
   }
   p = p1;
   dp = n * (x * p1 - p0) / (x * x - 1.0);
}

void GaussLegendre(int n, std::vector<double> &x, std::vector<double> &w)
{
   x.resize(n);
   w.resize(n);
   const int m = (n + 1) / 2;
   for (int i = 0; i < m; i++)
   {
      // Chebyshev-like initial guess, Newton on P_n
      double z = std::cos(M_PI * (i + 0.75) / (n + 0.5));
      double p, dp;
      for (int it = 0; it < 100; it++)
      {
         Legendre(n, z, p, dp);
         const double dz = p / dp;
         z -= dz;
         if (std::fabs(dz) < 1e-16) { break; }
      }
      Legendre(n, z, p, dp);
      const double wt = 2.0 / ((1.0 - z * z) * dp * dp);
      // z is the i-th largest root: map to [0,1], ascending order
      x[n - 1 - i] = 0.5 * (1.0 + z);
      x[i] = 0.5 * (1.0 - z);
      w[n - 1 - i] = w[i] = 0.5 * wt;
   }
   if (n % 2 == 1) { x[n / 2] = 0.5; }
}

void GaussLobatto(int n, std::vector<double> &x)
{
   x.resize(n);
   x[0] = 0.0;
   x[n - 1] = 1.0;
   const int N = n - 1; // interior nodes are the roots of P'_N
   for (int i = 1; i <= (n - 2 + 1) / 2; i++)
   {
      // initial guess: Chebyshev-Gauss-Lobatto point, Newton on q(z) = P'_N(z)
      double z = std::cos(M_PI * i / N);
      for (int it = 0; it < 100; it++)
      {
         double p, dp;
         Legendre(N, z, p, dp);
         // P''_N from the Legendre ODE: (1-z^2) P'' = 2 z P' - N(N+1) P
         const double d2p = (2.0 * z * dp - N * (N + 1) * p) / (1.0 - z * z);
         const double dz = dp / d2p;
         z -= dz;
         if (std::fabs(dz) < 1e-16) { break; }
      }
      x[n - 1 - i] = 0.5 * (1.0 + z);
      x[i] = 0.5 * (1.0 - z);
   }
   if (n % 2 == 1) { x[n / 2] = 0.5; }
}

void LagrangeTables(const std::vector<double> &nodes, const std::vector<double> &pts,
                    std::vector<double> &B, std::vector<double> &G)
{
   const int nd = (int)nodes.size(), nq = (int)pts.size();
   B.assign((size_t)nq * nd, 0.0);
   G.assign((size_t)nq * nd, 0.0);
   for (int d = 0; d < nd; d++)
   {
      double denom = 1.0;
      for (int m = 0; m < nd; m++) { if (m != d) { denom *= nodes[d] - nodes[m]; } }
      for (int q = 0; q < nq; q++)
      {
         const double x = pts[q];
         double val = 1.0;
         for (int m = 0; m < nd; m++) { if (m != d) { val *= x - nodes[m]; } }
         double der = 0.0;
         for (int k = 0; k < nd; k++)
         {
            if (k == d) { continue; }
            double t = 1.0;
            for (int m = 0; m < nd; m++) { if (m != d && m != k) { t *= x - nodes[m]; } }
            der += t;
         }
         B[q + (size_t)nq * d] = val / denom;
         G[q + (size_t)nq * d] = der / denom;
      }
   }
}

static double Binom(int n, int k)
{
   double r = 1.0;
   for (int i = 1; i <= k; i++) { r = r * (n - k + i) / i; }
   return r;
}

void BernsteinTable(int p, const std::vector<double> &pts, std::vector<double> &B)
{
   const int nq = (int)pts.size();
   B.assign((size_t)nq * (p + 1), 0.0);
   for (int l = 0; l <= p; l++)
      for (int q = 0; q < nq; q++)
      {
         B[q + (size_t)nq * l] = Binom(p, l) * std::pow(pts[q], l) * std::pow(1.0 - pts[q], p - l);
      }
}

Tables::Tables(int ov, int oe, int oq) : order_v(ov), order_e(oe), D1D(ov + 1), L1D(oe + 1)
{
   const int order = (oq > 0) ? oq : 3 * ov + oe - 1;
   Q1D = order / 2 + 1;
   GaussLegendre(Q1D, qpts, qwts);
   GaussLobatto(D1D, gll);
   LagrangeTables(gll, qpts, B, G);
   BernsteinTable(oe, qpts, Bl);
}

// ---- mesh ---------------------------------------------------------------------------
CartMesh CartMesh::Named(const std::string &name_in)
{
   // accept "data/cube01_hex.mesh", "cube01_hex.mesh" or "cube01_hex"
   std::string name = name_in;
   const size_t slash = name.find_last_of('/');
   if (slash != std::string::npos) { name = name.substr(slash + 1); }
   const size_t dot = name.rfind(".mesh");
   if (dot != std::string::npos) { name = name.substr(0, dot); }
   CartMesh m;
   if (name == "square01_quad")
   {
      m.dim = 2;
      m.brk[0] = {0.0, 0.5, 1.0};
      m.brk[1] = {0.0, 0.5, 1.0};
   }
   else if (name == "cube01_hex")
   {
      m.dim = 3;
      for (int a = 0; a < 3; a++) { m.brk[a] = {0.0, 0.5, 1.0}; }
   }
   else if (name == "box01_hex")
   {
      m.dim = 3;
      m.brk[0] = {0.0, 1.0, 3.0, 5.0, 7.0};
      m.brk[1] = {0.0, 1.5, 3.0};
      m.brk[2] = {0.0, 1.5, 3.0};
   }
   else if (name == "rectangle01_quad")
   {
      m.dim = 2;
      m.brk[0] = {0., 1., 2., 3., 4., 5., 6., 7.};
      m.brk[1] = {0., 1., 2., 3.};
   }
   else if (name == "rt2D")
   {
      m.dim = 2;
      m.brk[0] = {0.0, 0.5};
      m.brk[1] = {-1.0, -0.5, 0.0, 0.5, 1.0};
   }
   else if (name == "square_gresho")
   {
      m.dim = 2;
      m.brk[0] = {-0.5, 0.0, 0.5};
      m.brk[1] = {-0.5, 0.0, 0.5};
   }
   else
   {
      throw std::runtime_error("mesh '" + name_in + "' is not one of the structured meshes this "
                               "harness supports (square01_quad, cube01_hex, box01_hex, rectangle01_quad, "
                               "square_gresho, rt2D)");
   }
   return m;
}

CartMesh CartMesh::Cartesian(int dim, int nx, int ny, int nz, double sx, double sy, double sz)
{
   CartMesh m;
   m.dim = dim;
   const int n[3] = {nx, ny, nz};
   const double s[3] = {sx, sy, sz};
   for (int a = 0; a < dim; a++)
   {
      m.brk[a].resize(n[a] + 1);
      for (int i = 0; i <= n[a]; i++) { m.brk[a][i] = s[a] * i / n[a]; }
   }
   return m;
}

void CartMesh::UniformRefinement()
{
   for (int a = 0; a < dim; a++)
   {
      const std::vector<double> &b = brk[a];
      std::vector<double> r(2 * b.size() - 1);
      for (size_t i = 0; i < b.size(); i++) { r[2 * i] = b[i]; }
      for (size_t i = 0; i + 1 < b.size(); i++) { r[2 * i + 1] = 0.5 * (b[i] + b[i + 1]); }
      brk[a] = r;
   }
}

long CartMesh::NE() const
{
   long n = 1;
   for (int a = 0; a < dim; a++) { n *= ne(a); }
   return n;
}

Partition::Partition(const CartMesh &mesh, int nranks_, int rank_)
   : dim(mesh.dim), nranks(nranks_), rank(rank_)
{
   // split the axis with the most local elements by the smallest prime factor
   // of what is left, so blocks stay as cubic as possible (8 ranks -> 2x2x2)
   std::array<int, 3> loc{1, 1, 1};
   for (int a = 0; a < dim; a++) { loc[a] = mesh.ne(a); }
   int left = nranks;
   while (left > 1)
   {
      int f = 2;
      while (left % f) { f++; }
      int best = -1;
      for (int a = 0; a < dim; a++)
      {
         if (loc[a] % f == 0 && (best < 0 || loc[a] > loc[best])) { best = a; }
      }
      if (best < 0) { throw std::runtime_error("element grid cannot be split evenly over the ranks"); }
      loc[best] /= f;
      pgrid[best] *= f;
      left /= f;
   }
   int r = rank;
   for (int a = 0; a < dim; a++)
   {
      rcoord[a] = r % pgrid[a];
      r /= pgrid[a];
      ne[a] = mesh.ne(a) / pgrid[a];
      eoff[a] = rcoord[a] * ne[a];
   }
}

// ---- discretisation ----------------------------------------------------------------------
Discretization::Discretization(const CartMesh &mesh_, int order_v, int order_e, int problem_,
                               int nranks, int rank, int order_q, double blast)
   : dim(mesh_.dim), problem(problem_), tab(order_v, order_e, order_q), mesh(mesh_),
     part(mesh_, nranks, rank), blast_energy(blast)
{
   const int p = order_v, D = tab.D1D, Q = tab.Q1D, L = tab.L1D;
   NE = 1;
   N = 1;
   global_N = 1;
   global_NE = mesh.NE();
   for (int a = 0; a < dim; a++)
   {
      brk[a].assign(mesh.brk[a].begin() + part.eoff[a], mesh.brk[a].begin() + part.eoff[a] + part.ne[a] + 1);
      nn[a] = part.ne[a] * p + 1;
      NE *= part.ne[a];
      N *= nn[a];
      global_N *= (long)mesh.ne(a) * p + 1;
   }
   ND = NQ = NL = 1;
   for (int a = 0; a < dim; a++) { ND *= D; NQ *= Q; NL *= L; }
   H1V = dim * N;
   L2V = NE * NL;
   // tensor weights, q = qx + Q*(qy + Q*qz)
   W.resize(NQ);
   for (int q = 0; q < NQ; q++)
   {
      double w = 1.0;
      int r = q;
      for (int a = 0; a < dim; a++) { w *= tab.qwts[r % Q]; r /= Q; }
      W[q] = w;
   }
   // lexicographic element restriction
   h1map.resize((size_t)NE * ND);
   for (int e = 0; e < NE; e++)
   {
      int ec[3] = {0, 0, 0}, r = e;
      for (int a = 0; a < dim; a++) { ec[a] = r % part.ne[a]; r /= part.ne[a]; }
      for (int d = 0; d < ND; d++)
      {
         int dc[3] = {0, 0, 0}, rr = d;
         for (int a = 0; a < dim; a++) { dc[a] = rr % D; rr /= D; }
         long node = 0, stride = 1;
         for (int a = 0; a < dim; a++)
         {
            node += stride * (ec[a] * p + dc[a]);
            stride *= nn[a];
         }
         h1map[(size_t)e * ND + d] = (int)node;
      }
   }
   // essential dofs: attribute a+1 = faces normal to axis a of the GLOBAL boundary
   // (data/cube01_hex.mesh:28-53, laghos.cpp:499-515); ownership: low faces shared
   // with a lower rank are not owned.
   owner.assign(N, 1.0);
   for (int n = 0; n < N; n++)
   {
      int ic[3] = {0, 0, 0}, r = n;
      for (int a = 0; a < dim; a++) { ic[a] = r % nn[a]; r /= nn[a]; }
      for (int a = 0; a < dim; a++)
      {
         const bool lo = (ic[a] == 0), hi = (ic[a] == nn[a] - 1);
         if ((lo && part.rcoord[a] == 0) || (hi && part.rcoord[a] == part.pgrid[a] - 1)) { ess[a].push_back(n); }
         if (lo && part.rcoord[a] > 0) { owner[n] = 0.0; }
      }
   }
   // neighbours: ranks whose block touches this one (faces, edges, corners)
   int off[3] = {0, 0, 0};
   const int lo3[3] = {-1, -1, dim == 3 ? -1 : 0}, hi3[3] = {1, 1, dim == 3 ? 1 : 0};
   for (off[2] = lo3[2]; off[2] <= hi3[2]; off[2]++)
      for (off[1] = lo3[1]; off[1] <= hi3[1]; off[1]++)
         for (off[0] = lo3[0]; off[0] <= hi3[0]; off[0]++)
         {
            if (off[0] == 0 && off[1] == 0 && off[2] == 0) { continue; }
            int nc[3], nr = 0, stride = 1;
            bool ok = true;
            for (int a = 0; a < dim; a++)
            {
               nc[a] = part.rcoord[a] + off[a];
               if (nc[a] < 0 || nc[a] >= part.pgrid[a]) { ok = false; }
            }
            if (!ok) { continue; }
            for (int a = 0; a < dim; a++) { nr += stride * nc[a]; stride *= part.pgrid[a]; }
            // shared nodes: plane index fixed where off != 0, full range elsewhere,
            // enumerated in local lexicographic order (identical on both sides)
            int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
            for (int a = 0; a < dim; a++)
            {
               if (off[a] < 0) { lo[a] = hi[a] = 0; }
               else if (off[a] > 0) { lo[a] = hi[a] = nn[a] - 1; }
               else { lo[a] = 0; hi[a] = nn[a] - 1; }
            }
            std::vector<int> nodes;
            for (int k = lo[2]; k <= hi[2]; k++)
               for (int j = lo[1]; j <= hi[1]; j++)
                  for (int i = lo[0]; i <= hi[0]; i++)
                  {
                     nodes.push_back(i + nn[0] * (j + nn[1] * k));
                  }
            nbr_rank.push_back(nr);
            nbr_nodes.push_back(nodes);
         }
}

void Discretization::ElemPoint(int e, const double *ref, double *x) const
{
   int r = e;
   for (int a = 0; a < dim; a++)
   {
      const int ec = r % part.ne[a];
      r /= part.ne[a];
      x[a] = brk[a][ec] + (brk[a][ec + 1] - brk[a][ec]) * ref[a];
   }
}

double Discretization::rho0(const double *x) const
{
   switch (problem)
   {
      case 0: return 1.0;
      case 1: return 1.0;
      case 2: return (x[0] < 0.5) ? 1.0 : 0.1;
      case 3:
         return (dim == 2) ? ((x[0] > 1.0 && x[1] > 1.5) ? 0.125 : 1.0)
                           : ((x[0] > 1.0 && ((x[1] < 1.5 && x[2] < 1.5) || (x[1] > 1.5 && x[2] > 1.5))) ? 0.125 : 1.0);
      case 4: return 1.0;
      case 5: // laghos.cpp:1105-1110
         if (x[0] >= 0.5 && x[1] >= 0.5) { return 0.5313; }
         if (x[0] < 0.5 && x[1] < 0.5) { return 0.8; }
         return 1.0;
      case 6: // laghos.cpp:1111-1116
         if (x[0] < 0.5 && x[1] >= 0.5) { return 2.0; }
         if (x[0] >= 0.5 && x[1] < 0.5) { return 3.0; }
         return 1.0;
      case 7: return x[1] >= 0.0 ? 2.0 : 1.0; // laghos.cpp:1117
      default: throw std::runtime_error("Bad number given for problem id!");
   }
}
double Discretization::gamma_func(const double *x) const
{
   switch (problem)
   {
      case 0: return 5.0 / 3.0;
      case 1: return 1.4;
      case 2: return 1.4;
      case 3: return (x[0] > 1.0 && x[1] <= 1.5) ? 1.4 : 1.5;
      case 4: return 5.0 / 3.0;
      case 5: return 1.4;
      case 6: return 1.4;
      case 7: return 5.0 / 3.0;
      default: throw std::runtime_error("Bad number given for problem id!");
   }
}
void Discretization::v0(const double *x, double *v) const
{
   for (int a = 0; a < dim; a++) { v[a] = 0.0; }
   if (problem == 0)
   {
      v[0] = std::sin(M_PI * x[0]) * std::cos(M_PI * x[1]);
      v[1] = -std::cos(M_PI * x[0]) * std::sin(M_PI * x[1]);
      if (dim == 3)
      {
         v[0] *= std::cos(M_PI * x[2]);
         v[1] *= std::cos(M_PI * x[2]);
         v[2] = 0.0;
      }
   }
   else if (problem == 5 || problem == 6) // laghos.cpp:1144-1145, :1178-1197
   {
      const double atn = std::pow(x[0] * (1.0 - x[0]) * 4 * x[1] * (1.0 - x[1]) * 4.0, 0.4);
      const bool hx = x[0] >= 0.5, hy = x[1] >= 0.5;
      if (problem == 5)
      {
         v[0] = (!hx && hy) ? 0.7276 * atn : 0.0 * atn;
         v[1] = (hx && !hy) ? 0.7276 * atn : 0.0 * atn;
      }
      else
      {
         v[0] = hy ? 0.75 * atn : -0.75 * atn;
         v[1] = hx ? -0.5 * atn : 0.5 * atn;
      }
   }
   else if (problem == 7) // laghos.cpp:1198-1203
   {
      v[1] = 0.02 * std::exp(-2 * M_PI * x[1] * x[1]) * std::cos(2 * M_PI * x[0]);
   }
   else if (problem == 4) // Gresho vortex, laghos.cpp:1161-1177
   {
      const double r = std::sqrt(x[0] * x[0] + x[1] * x[1]);
      if (r < 0.2)
      {
         v[0] = 5.0 * x[1];
         v[1] = -5.0 * x[0];
      }
      else if (r < 0.4)
      {
         v[0] = 2.0 * x[1] / r - 5.0 * x[1];
         v[1] = -2.0 * x[0] / r + 5.0 * x[0];
      }
   }
}
double Discretization::e0(const double *x) const
{
   switch (problem)
   {
      case 0:
      {
         const double denom = 2.0 / 3.0;
         double val;
         if (dim == 2) { val = 1.0 + (std::cos(2 * M_PI * x[0]) + std::cos(2 * M_PI * x[1])) / 4.0; }
         else
         {
            val = 100.0 + ((std::cos(2 * M_PI * x[2]) + 2) * (std::cos(2 * M_PI * x[0]) + std::cos(2 * M_PI * x[1])) - 2) / 16.0;
         }
         return val / denom;
      }
      case 1: return 0.0;
      case 3: return ((x[0] > 1.0) ? 0.1 : 1.0) / rho0(x) / (gamma_func(x) - 1.0);
      case 2: return ((x[0] < 0.5) ? 1.0 : 0.1) / rho0(x) / (gamma_func(x) - 1.0); // :1228-1229
      case 5: // :1248-1257
      {
         const double irg = 1.0 / rho0(x) / (gamma_func(x) - 1.0);
         return ((x[0] >= 0.5 && x[1] >= 0.5) ? 0.4 : 1.0) * irg;
      }
      case 6: return 1.0 / rho0(x) / (gamma_func(x) - 1.0); // :1258-1267
      case 7: // laghos.cpp:1268-1272
      {
         const double rho = rho0(x), gamma = gamma_func(x);
         return (6.0 - rho * x[1]) / (gamma - 1.0) / rho;
      }
      case 4: // laghos.cpp:1232-1247
      {
         const double rsq = x[0] * x[0] + x[1] * x[1], r = std::sqrt(rsq);
         const double gamma = 5.0 / 3.0;
         if (r < 0.2) { return (5.0 + 25.0 / 2.0 * rsq) / (gamma - 1.0); }
         else if (r < 0.4)
         {
            const double t1 = 9.0 - 4.0 * std::log(0.2) + 25.0 / 2.0 * rsq;
            const double t2 = 20.0 * r - 4.0 * std::log(r);
            return (t1 - t2) / (gamma - 1.0);
         }
         return (3.0 + 4.0 * std::log(2.0)) / (gamma - 1.0);
      }
      default: throw std::runtime_error("problem not supported by this harness");
   }
}

// solve the small dense system A X = rhs in place (Gaussian elimination, partial pivoting)
static void SolveDense(int n, std::vector<double> A, std::vector<double> &X, int nrhs)
{
   for (int k = 0; k < n; k++)
   {
      int piv = k;
      for (int i = k + 1; i < n; i++) { if (std::fabs(A[i * n + k]) > std::fabs(A[piv * n + k])) { piv = i; } }
      if (piv != k)
      {
         for (int j = 0; j < n; j++) { std::swap(A[k * n + j], A[piv * n + j]); }
         for (int j = 0; j < nrhs; j++) { std::swap(X[k * nrhs + j], X[piv * nrhs + j]); }
      }
      for (int i = k + 1; i < n; i++)
      {
         const double f = A[i * n + k] / A[k * n + k];
         for (int j = k; j < n; j++) { A[i * n + j] -= f * A[k * n + j]; }
         for (int j = 0; j < nrhs; j++) { X[i * nrhs + j] -= f * X[k * nrhs + j]; }
      }
   }
   for (int k = n - 1; k >= 0; k--)
   {
      for (int j = 0; j < nrhs; j++)
      {
         double s = X[k * nrhs + j];
         for (int i = k + 1; i < n; i++) { s -= A[k * n + i] * X[i * nrhs + j]; }
         X[k * nrhs + j] = s / A[k * n + k];
      }
   }
}

// Nodal (Gauss-Legendre) L2 values -> Bernstein coefficients, element by element.
// GridFunction::ProjectGridFunction onto the positive basis is a local L2
// projection (laghos.cpp:583-595, :622); both bases span Q_p on an affine element,
// so it equals this exact change of basis, applied one tensor direction at a time.
void Discretization::NodalToBernstein(std::vector<double> &vals) const
{
   const int L = tab.L1D, p = tab.order_e;
   std::vector<double> nodes, wts, V;
   GaussLegendre(L, nodes, wts);
   BernsteinTable(p, nodes, V); // V[i + L*l] = B_l(node_i)
   // Vinv via solving V * Vinv = I (row-major copies)
   std::vector<double> A((size_t)L * L), Vinv((size_t)L * L, 0.0);
   for (int i = 0; i < L; i++)
      for (int l = 0; l < L; l++) { A[i * L + l] = V[i + L * l]; }
   for (int i = 0; i < L; i++) { Vinv[i * L + i] = 1.0; }
   SolveDense(L, A, Vinv, L); // Vinv[l*L + i]
   std::vector<double> tmp(NL);
   for (int e = 0; e < NE; e++)
   {
      double *u = vals.data() + (size_t)e * NL;
      int stride = 1;
      for (int a = 0; a < dim; a++)
      {
         for (int idx = 0; idx < NL; idx++)
         {
            const int ia = (idx / stride) % L;
            const int base = idx - ia * stride;
            double s = 0.0;
            for (int i = 0; i < L; i++) { s += Vinv[ia * L + i] * u[base + i * stride]; }
            tmp[idx] = s;
         }
         for (int idx = 0; idx < NL; idx++) { u[idx] = tmp[idx]; }
         stride *= L;
      }
   }
}

void Discretization::InitialState(std::vector<double> &S, std::vector<double> &rho0_l2,
                                  std::vector<double> &gamma, std::vector<double> &rho0_q) const
{
   const int p = tab.order_v, L = tab.L1D, Q = tab.Q1D;
   S.assign((size_t)2 * H1V + L2V, 0.0);
   // node positions: affine image of the Gauss-Lobatto points (pmesh.SetNodalGridFunction)
   std::array<std::vector<double>, 3> c1;
   for (int a = 0; a < dim; a++)
   {
      c1[a].resize(nn[a]);
      for (int e = 0; e < part.ne[a]; e++)
      {
         const double h = brk[a][e + 1] - brk[a][e];
         for (int d = 0; d < p; d++) { c1[a][e * p + d] = brk[a][e] + h * tab.gll[d]; }
      }
      c1[a][nn[a] - 1] = brk[a].back();
   }
   for (int n = 0; n < N; n++)
   {
      int r = n;
      double x[3] = {0, 0, 0}, v[3] = {0, 0, 0};
      for (int a = 0; a < dim; a++) { x[a] = c1[a][r % nn[a]]; r /= nn[a]; }
      v0(x, v); // ProjectCoefficient: point-wise at the nodes (laghos.cpp:574-575)
      for (int a = 0; a < dim; a++)
      {
         S[(size_t)a * N + n] = x[a];
         S[(size_t)H1V + (size_t)a * N + n] = v[a];
      }
   }
   // (the essential velocity dofs are zeroed at the end: `ess` is in the numbering the operators get, laghos.cpp:576-579)
   // rho0 and e0: nodal L2 interpolation, then projection to Bernstein (laghos.cpp:589-622)
   std::vector<double> glx, glw;
   GaussLegendre(L, glx, glw);
   rho0_l2.assign(L2V, 0.0);
   std::vector<double> e_l2(L2V, 0.0);
   for (int e = 0; e < NE; e++)
      for (int l = 0; l < NL; l++)
      {
         double ref[3] = {0, 0, 0}, x[3];
         int r = l;
         for (int a = 0; a < dim; a++) { ref[a] = glx[r % L]; r /= L; }
         ElemPoint(e, ref, x);
         rho0_l2[(size_t)e * NL + l] = rho0(x);
         if (problem != 1) { e_l2[(size_t)e * NL + l] = e0(x); }
      }
   if (problem == 1)
   {
      // Sedov: DeltaCoefficient at the origin, scale E0/2^dim (laghos.cpp:597-616).
      // Upstream ProjectDeltaCoefficient: in the element having the origin as a
      // vertex the nodal values are prod_a (1 - x_a)^p, scaled so the integral of
      // the interpolant equals the scale.
      bool origin_rank = true;
      for (int a = 0; a < dim; a++) { origin_rank = origin_rank && part.rcoord[a] == 0; }
      double vol0 = 1.0; // volume of the global origin element
      for (int a = 0; a < dim; a++) { vol0 *= mesh.brk[a][1] - mesh.brk[a][0]; }
      const int pe = tab.order_e;
      double integral1d = 0.0;
      for (int i = 0; i < L; i++) { integral1d += glw[i] * std::pow(1.0 - glx[i], pe); }
      const double integral = std::pow(integral1d, dim) * vol0;
      const double scale = blast_energy / std::pow(2.0, dim);
      if (origin_rank)
      {
         for (int l = 0; l < NL; l++)
         {
            double s = 1.0;
            int r = l;
            for (int a = 0; a < dim; a++) { s *= std::pow(1.0 - glx[r % L], pe); r /= L; }
            e_l2[l] = s * scale / integral; // element 0 is the origin element
         }
      }
   }
   NodalToBernstein(rho0_l2);
   NodalToBernstein(e_l2);
   for (int i = 0; i < L2V; i++) { S[(size_t)2 * H1V + i] = e_l2[i]; }
   // gamma: order-0 L2 projection = value at the element centre (laghos.cpp:628-632)
   gamma.resize(NE);
   rho0_q.resize((size_t)NE * NQ);
   for (int e = 0; e < NE; e++)
   {
      const double half[3] = {0.5, 0.5, 0.5};
      double x[3];
      ElemPoint(e, half, x);
      gamma[e] = gamma_func(x);
      for (int q = 0; q < NQ; q++)
      {
         double ref[3] = {0, 0, 0};
         int r = q;
         for (int a = 0; a < dim; a++) { ref[a] = tab.qpts[r % Q]; r /= Q; }
         ElemPoint(e, ref, x);
         rho0_q[(size_t)e * NQ + q] = rho0(x); // mass coefficient at the qpts
      }
   }
   if (!node_perm.empty()) // -renumber: the same fields in the numbering Renumber() gave the nodes and zones
   {
      std::vector<double> T(S.size());
      for (int b = 0; b < 2 * dim; b++)
         for (int n = 0; n < N; n++) { T[(size_t)b * N + node_perm[n]] = S[(size_t)b * N + n]; }
      auto zones = [&](const double *in, double *out, int per) {
         for (int j = 0; j < NE; j++) { std::copy(in + (size_t)elem_perm[j] * per, in + (size_t)(elem_perm[j] + 1) * per, out + (size_t)j * per); }
      };
      zones(S.data() + 2 * (size_t)H1V, T.data() + 2 * (size_t)H1V, NL);
      S.swap(T);
      std::vector<double> r(rho0_l2.size()), g(gamma.size()), rq(rho0_q.size());
      zones(rho0_l2.data(), r.data(), NL);
      zones(gamma.data(), g.data(), 1);
      zones(rho0_q.data(), rq.data(), NQ);
      rho0_l2.swap(r);
      gamma.swap(g);
      rho0_q.swap(rq);
   }
   for (int a = 0; a < dim; a++)
      for (int n : ess[a]) { S[(size_t)H1V + (size_t)a * N + n] = 0.0; } // laghos.cpp:576-579
}

// ---- renumbering (`-renumber`) ---------------------------------------------------------------------
namespace
{
// MFEM's local numbering of the vertices, edges and faces of a quadrilateral / hexahedron (upstream mesh/geom; the vertex
// order is the one of the reference's data/*.mesh files, e.g. /root/reference/data/cube01_hex.mesh:17-24)
const int kVert[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
const int kHexEdge[12][2] = {{0, 1}, {1, 2}, {3, 2}, {0, 3}, {4, 5}, {5, 6}, {7, 6}, {4, 7}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
const int kHexFace[6][4] = {{3, 2, 1, 0}, {0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {3, 0, 4, 7}, {4, 5, 6, 7}};
const int kQuadEdge[4][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}};

struct Lattice // one int per point of the integer lattice [0, n0] x [0, n1] x [0, n2]
{
   int n[3];
   std::vector<int> v;
   explicit Lattice(const int m[3])
   {
      for (int a = 0; a < 3; a++) { n[a] = m[a]; }
      v.assign((size_t)(n[0] + 1) * (n[1] + 1) * (n[2] + 1), -1);
   }
   int &at(const int c[3]) { return v[c[0] + (size_t)(n[0] + 1) * (c[1] + (size_t)(n[1] + 1) * c[2])]; }
};

// The numbering MFEM would give `levels` uniform refinements of the lexicographic (nloc >> levels) base mesh with an H1
// space of order p on it.  Restated from upstream MFEM (none of it is under /root/reference; SURVEY Appendix A style):
//  * Mesh::UniformRefinement (quad / hex): the 2^dim children of element i become elements 2^dim i + k, child k at the
//    parent's vertex k, same orientation; old vertices keep their numbers, then one new vertex per edge, per face (3D), per
//    element, each class in the coarse mesh's numbering of that entity;
//  * edges and faces are numbered in the order the elements first meet them (element by element, local edge / face order);
//  * FiniteElementSpace: vertex dofs, then (p-1) per edge running from its lower to its higher vertex, then (p-1)^2 per
//    face in the frame of the face's vertices as its first element lists them, then (p-1)^dim interior dofs per element.
// What matters for the kernels is the CHARACTER of that numbering - nodes of a zone scattered over four ranges, zones along
// a space-filling tree order - not the last detail of the order inside an entity.
void MfemLikeNumbering(int dim, const int nloc[3], int levels, int p, std::vector<int> &node_perm, std::vector<int> &elem_perm)
{
   int g[3] = {0, 0, 0};
   for (int a = 0; a < dim; a++)
   {
      while (levels > 0 && nloc[a] % (1 << levels)) { levels--; } // (a block that is not 2^levels zones wide: fewer levels)
   }
   for (int a = 0; a < dim; a++) { g[a] = nloc[a] >> levels; }
   const int nchild = 1 << dim, nedge = (dim == 3) ? 12 : 4;
   const int(*edges)[2] = (dim == 3) ? kHexEdge : kQuadEdge;
   std::vector<std::array<int, 3>> elems;
   for (int k = 0; k < std::max(g[2], 1); k++)
      for (int j = 0; j < g[1]; j++)
         for (int i = 0; i < g[0]; i++) { elems.push_back({i, j, k}); }
   Lattice vid(g);
   int nv = 0;
   for (size_t i = 0; i < vid.v.size(); i++) { vid.v[i] = nv++; }
   for (int lev = 0; lev < levels; lev++)
   {
      int f[3] = {2 * g[0], 2 * g[1], 2 * g[2]};
      Lattice fv(f);
      {
         int c[3];
         for (c[2] = 0; c[2] <= g[2]; c[2]++)
            for (c[1] = 0; c[1] <= g[1]; c[1]++)
               for (c[0] = 0; c[0] <= g[0]; c[0]++)
               {
                  int c2[3] = {2 * c[0], 2 * c[1], 2 * c[2]};
                  fv.at(c2) = vid.at(c);
               }
      }
      for (const auto &e : elems)
         for (int le = 0; le < nedge; le++)
         {
            int m[3];
            for (int c = 0; c < 3; c++) { m[c] = 2 * e[c] + kVert[edges[le][0]][c] + kVert[edges[le][1]][c]; }
            if (fv.at(m) < 0) { fv.at(m) = nv++; }
         }
      if (dim == 3)
      {
         for (const auto &e : elems)
            for (int lf = 0; lf < 6; lf++)
            {
               int m[3];
               for (int c = 0; c < 3; c++)
               {
                  int sum = 0;
                  for (int k = 0; k < 4; k++) { sum += kVert[kHexFace[lf][k]][c]; }
                  m[c] = 2 * e[c] + sum / 2;
               }
               if (fv.at(m) < 0) { fv.at(m) = nv++; }
            }
      }
      for (const auto &e : elems)
      {
         int m[3] = {2 * e[0] + 1, 2 * e[1] + 1, dim == 3 ? 2 * e[2] + 1 : 0};
         fv.at(m) = nv++;
      }
      std::vector<std::array<int, 3>> fine;
      fine.reserve(elems.size() * nchild);
      for (const auto &e : elems)
         for (int k = 0; k < nchild; k++) { fine.push_back({2 * e[0] + kVert[k][0], 2 * e[1] + kVert[k][1], 2 * e[2] + kVert[k][2]}); }
      elems.swap(fine);
      vid = fv;
      for (int a = 0; a < 3; a++) { g[a] = f[a]; }
   }
   const int NE = (int)elems.size();
   elem_perm.resize(NE);
   for (int j = 0; j < NE; j++) { elem_perm[j] = elems[j][0] + nloc[0] * (elems[j][1] + nloc[1] * elems[j][2]); }
   // H1 dofs on the lattice of the nodes
   int P[3] = {g[0] * p, g[1] * p, g[2] * p};
   Lattice nid(P);
   {
      int c[3];
      for (c[2] = 0; c[2] <= g[2]; c[2]++)
         for (c[1] = 0; c[1] <= g[1]; c[1]++)
            for (c[0] = 0; c[0] <= g[0]; c[0]++)
            {
               int cp[3] = {p * c[0], p * c[1], p * c[2]};
               nid.at(cp) = vid.at(c);
            }
   }
   int next = nv;
   auto vertex = [&](const std::array<int, 3> &e, int k, int out[3]) {
      for (int c = 0; c < 3; c++) { out[c] = e[c] + kVert[k][c]; }
   };
   if (p > 1)
   {
      for (const auto &e : elems)
         for (int le = 0; le < nedge; le++)
         {
            int A[3], B[3];
            vertex(e, edges[le][0], A);
            vertex(e, edges[le][1], B);
            if (vid.at(A) > vid.at(B)) { std::swap_ranges(A, A + 3, B); }
            int c[3];
            for (int t = 1; t < p; t++)
            {
               for (int a = 0; a < 3; a++) { c[a] = p * A[a] + t * (B[a] - A[a]); }
               if (t == 1 && nid.at(c) >= 0) { break; }
               nid.at(c) = next++;
            }
         }
      if (dim == 3)
      {
         for (const auto &e : elems)
            for (int lf = 0; lf < 6; lf++)
            {
               int A[3], B[3], Dv[3];
               vertex(e, kHexFace[lf][0], A);
               vertex(e, kHexFace[lf][1], B);
               vertex(e, kHexFace[lf][3], Dv);
               int c[3];
               bool seen = false;
               for (int j = 1; j < p && !seen; j++)
                  for (int i = 1; i < p; i++)
                  {
                     for (int a = 0; a < 3; a++) { c[a] = p * A[a] + i * (B[a] - A[a]) + j * (Dv[a] - A[a]); }
                     if (i == 1 && j == 1 && nid.at(c) >= 0) { seen = true; break; }
                     nid.at(c) = next++;
                  }
            }
      }
      for (const auto &e : elems)
      {
         int c[3] = {0, 0, 0};
         for (int k = 1; k < (dim == 3 ? p : 2); k++)
            for (int j = 1; j < p; j++)
               for (int i = 1; i < p; i++)
               {
                  c[0] = p * e[0] + i;
                  c[1] = p * e[1] + j;
                  c[2] = (dim == 3) ? p * e[2] + k : 0;
                  nid.at(c) = next++;
               }
      }
   }
   if ((size_t)next != nid.v.size()) { throw std::runtime_error("MfemLikeNumbering: the dof count does not add up"); }
   node_perm.assign(nid.v.begin(), nid.v.end()); // the lattice is stored x fastest: the structured node order
   for (int id : node_perm) { if (id < 0) { throw std::runtime_error("MfemLikeNumbering: a node was left without a number"); } }
}
} // namespace

void Discretization::Renumber(const std::string &mode, int levels, unsigned seed)
{
   if (mode.empty() || mode == "none" || mode == "lexicographic") { return; }
   if (!node_perm.empty()) { throw std::runtime_error("Discretization::Renumber: already renumbered"); }
   std::vector<int> np, ep;
   if (mode == "mfem")
   {
      const int nloc[3] = {part.ne[0], part.ne[1], dim == 3 ? part.ne[2] : 0};
      MfemLikeNumbering(dim, nloc, levels, tab.order_v, np, ep);
   }
   else if (mode == "random")
   {
      np.resize(N);
      ep.resize(NE);
      std::iota(np.begin(), np.end(), 0);
      std::iota(ep.begin(), ep.end(), 0);
      std::mt19937_64 rng(seed);
      std::shuffle(np.begin(), np.end(), rng);
      std::shuffle(ep.begin(), ep.end(), rng);
   }
   else { throw std::runtime_error("-renumber " + mode + ": expected mfem, random or none"); }
   if ((int)np.size() != N || (int)ep.size() != NE) { throw std::runtime_error("Discretization::Renumber: sizes do not match the block"); }
   std::vector<int> hm(h1map.size());
   for (int j = 0; j < NE; j++)
      for (int k = 0; k < ND; k++) { hm[(size_t)j * ND + k] = np[h1map[(size_t)ep[j] * ND + k]]; }
   h1map.swap(hm);
   for (auto &list : ess)
   {
      for (int &n : list) { n = np[n]; }
      std::sort(list.begin(), list.end());
   }
   std::vector<double> own(N);
   for (int n = 0; n < N; n++) { own[np[n]] = owner[n]; }
   owner.swap(own);
   for (auto &list : nbr_nodes)
      for (int &n : list) { n = np[n]; } // (the order of a list is what the two ranks of a pair share: kept)
   node_perm.swap(np);
   elem_perm.swap(ep);
   numbering = mode;
}

} // namespace laghos
