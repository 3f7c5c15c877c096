// lgh_qupdate.hip — the update and energy-integral modes of the quadrature-point kernel (lgh_qpoint.hpp), the 2D
// Taylor-Green source and the small-matrix probes.  This file is compiled with the relaxed fp64 division (Makefile).
#include "lgh_qpoint.hpp"
#include "lgh_qrows.hpp"

namespace lgh
{

int qupdate_form(lgh_ctx *c);
int qupdate(lgh_ctx *c, const double *S)
{
   QArgs a = q_base(c);
   a.x = S;
   a.v = S + c->H1V;
   a.e = S + 2 * (size_t)c->H1V;
   a.result = c->dt_est_dev;
   // lgh_qupdate_store_stress(ctx, 0): with both force products formed from registers nobody reads the nine stressJinvT
   // planes (43 % of this kernel's bytes) - they are not written; every reader checks stress_current and refuses
   a.Jac0inv_e = (c->jac0_compact == 1) ? c->Jac0inv_e : nullptr;
   const bool keep = !c->stress_store && a.erhs_q && a.force_e && c->v_snap;
   if (keep) { a.stressJinvT = nullptr; }
   // 3D up to Q4Q3: the form with row-owned contraction stages (lgh_qrows.hpp); LGH_Q_FORM=0: the point form (A/B, tests)
   const int form = qupdate_form(c);
   const int rc = (form == 1) ? launch_qrows(c, a) : launch_q<QMODE_UPDATE>(c, a);
   // F^T v of this state's velocity block is now in c->erhs_q, F.1 in c->force_e_q; lgh_solve_energy compares the
   // velocity it is given with the one the product was formed from
   c->qgen++;
   c->stress_current = (rc == LGH_OK && !keep) ? 1 : 0;
   c->fused_ftv_valid = (rc == LGH_OK && a.erhs_q && c->v_snap) ? 1 : 0;
   c->fused_f1_valid = (rc == LGH_OK && a.force_e) ? 1 : 0;
   if (c->fused_ftv_valid)
   {
      LGH_HIP_CHECK(hipMemcpyAsync(c->v_snap, S + c->H1V, sizeof(double) * (size_t)c->H1V, hipMemcpyDeviceToDevice, c->stream));
   }
   return rc;
}

// which form lgh_qupdate launches for this context: 1 = row form (lgh_qrows.hpp), 0 = point form (qpoint_kernel)
int qupdate_form(lgh_ctx *c)
{
   const char *fenv = getenv("LGH_Q_FORM");
   return (qrows_available(c) && !(fenv && fenv[0] == '0')) ? 1 : 0;
}

int interp_energy(lgh_ctx *c, int which, const double *vec, double *result)
{
   QArgs a = q_base(c);
   a.result = c->scal;
   int rc;
   if (which == 0)
   {
      a.e = vec;
      rc = launch_q<QMODE_IE>(c, a);
   }
   else
   {
      a.v = vec;
      rc = launch_q<QMODE_KE>(c, a);
   }
   if (rc) { return rc; }
   if (c->multi != 0)
   {
      rc = allreduce_dev(c, c->scal, 1, 0);
      if (rc) { return rc; }
   }
   LGH_HIP_CHECK(hipMemcpyAsync(c->host_pinned, c->scal, sizeof(double), hipMemcpyDeviceToHost, c->stream));
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   *result = (which == 0) ? c->host_pinned[0] : 0.5 * c->host_pinned[0];
   return LGH_OK;
}

// ---- 2D Taylor-Green energy source (laghos_solver.cpp:448-467, TaylorCoefficient
// laghos_solver.hpp:208-218): e_src_l = sum_q w_q detJ(x_q) f(x_q) phi_l(q) on the CURRENT
// mesh, f = 3/8 pi (cos 3 pi x cos pi y - cos pi x cos 3 pi y).  One workgroup per element,
// one thread per quadrature point; a set-up-grade kernel (2D problem 0 only).
__global__ void __launch_bounds__(128)
tg_source_2d_k(const int NE, const int N, const int D, const int Q, const int L, const int *__restrict__ map,
               const double *__restrict__ B, const double *__restrict__ G, const double *__restrict__ Bl,
               const double *__restrict__ W, const double *__restrict__ x, double *__restrict__ out)
{
   __shared__ double sx[2 * 36], ss[100];
   const int e = blockIdx.x, t = threadIdx.x;
   const int ND = D * D, NQ = Q * Q, NL = L * L;
   for (int i = t; i < 2 * ND; i += blockDim.x)
   {
      const int c = i / ND, d = i - c * ND;
      sx[i] = x[(size_t)c * N + map[(size_t)e * ND + d]];
   }
   __syncthreads();
   if (t < NQ)
   {
      const int qx = t % Q, qy = t / Q;
      double v[2] = {0.0, 0.0}, gx[2] = {0.0, 0.0}, gy[2] = {0.0, 0.0};
      for (int dy = 0; dy < D; dy++)
      {
         for (int dx = 0; dx < D; dx++)
         {
            const double bb = B[qx + Q * dx] * B[qy + Q * dy];
            const double gb = G[qx + Q * dx] * B[qy + Q * dy];
            const double bg = B[qx + Q * dx] * G[qy + Q * dy];
            for (int c = 0; c < 2; c++)
            {
               const double u = sx[c * ND + dx + D * dy];
               v[c] += bb * u;
               gx[c] += gb * u;
               gy[c] += bg * u;
            }
         }
      }
      const double det = gx[0] * gy[1] - gx[1] * gy[0];
      const double f = 3.0 / 8.0 * M_PI * (cos(3.0 * M_PI * v[0]) * cos(M_PI * v[1]) - cos(M_PI * v[0]) * cos(3.0 * M_PI * v[1]));
      ss[t] = W[t] * det * f;
   }
   __syncthreads();
   if (t < NL)
   {
      const int lx = t % L, ly = t / L;
      double s = 0.0;
      for (int qy = 0; qy < Q; qy++)
      {
         for (int qx = 0; qx < Q; qx++) { s += ss[qx + Q * qy] * Bl[qx + Q * lx] * Bl[qy + Q * ly]; }
      }
      out[t + (size_t)NL * e] = s;
   }
}
int tg_source_2d(lgh_ctx *c, const double *S, double *out)
{
   if (c->dim != 2) { set_error("the Taylor-Green energy source is the 2D one (laghos_solver.cpp:448)"); return LGH_ERR_ARG; }
   if (c->D1D > 6 || c->Q1D > 10 || c->NQ > 128) { return LGH_ERR_UNSUPPORTED; }
   hipLaunchKernelGGL(tg_source_2d_k, dim3(c->NE), dim3(128), 0, c->stream, c->NE, c->N, c->D1D, c->Q1D, c->L1D,
                      c->h1map, c->B, c->G, c->Bl, c->W, S, out);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

// ---- device probes for the small-matrix kernels (tests) ----------------------
template <int DIM>
__global__ void test_eig_k(int n, const double *A, double *lambda, double *vec)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) { return; }
   double a[DIM * DIM], v[DIM], l;
   for (int k = 0; k < DIM * DIM; k++) { a[k] = A[(size_t)i * DIM * DIM + k]; }
   sm::min_eigenpair<DIM>(a, l, v);
   lambda[i] = l;
   for (int k = 0; k < DIM; k++) { vec[(size_t)i * DIM + k] = v[k]; }
}
template <int DIM>
__global__ void test_sv_k(int n, const double *A, double *sv)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) { return; }
   double a[DIM * DIM];
   for (int k = 0; k < DIM * DIM; k++) { a[k] = A[(size_t)i * DIM * DIM + k]; }
   sv[i] = sm::min_singular<DIM>(a);
}
__global__ void test_sqrt_k(int n, const double *x, double *y)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { y[i] = sm::fsqrt(x[i]); }
}
int test_sqrt(lgh_ctx *c, int n, const double *x, double *y)
{
   hipLaunchKernelGGL(test_sqrt_k, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, n, x, y);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int test_eig(lgh_ctx *c, int dim, int n, const double *A, double *lambda, double *vec)
{
   if (dim == 2) { hipLaunchKernelGGL(test_eig_k<2>, dim3(ceil_div(n, 128)), dim3(128), 0, c->stream, n, A, lambda, vec); }
   else { hipLaunchKernelGGL(test_eig_k<3>, dim3(ceil_div(n, 128)), dim3(128), 0, c->stream, n, A, lambda, vec); }
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int test_singular(lgh_ctx *c, int dim, int n, const double *A, double *sv)
{
   if (dim == 2) { hipLaunchKernelGGL(test_sv_k<2>, dim3(ceil_div(n, 128)), dim3(128), 0, c->stream, n, A, sv); }
   else { hipLaunchKernelGGL(test_sv_k<3>, dim3(ceil_div(n, 128)), dim3(128), 0, c->stream, n, A, sv); }
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

} // namespace lgh
