#pragma once
// lgh_qbatch.hpp — the 3D quadrature-data update at Q3Q2 as a persistent kernel over BATCHES of seven elements
// (round 4, experimental form 2: LGH_Q_FORM=2).
//
// Same reference code as qrows_kernel / qpoint_kernel (QUpdate::UpdateQuadratureData + QKernel / QUpdateBody,
// /root/reference/laghos_solver.cpp:1354-1411, :1263-1352, :1042-1168, with ForcePA->Mult(one) and
// ForcePA->MultTranspose(v) of /root/reference/laghos_assembly.cpp:296-514, :715-924 fused); same point body.
//
// Why.  One element is 216 points = 3 wavefronts + 24 lanes: every form with one element per workgroup runs its point
// body four times per element for 3.375 wavefronts of work, and shares each SIMD between three workgroups that wait on
// each other at seven barriers per element.  Seven elements are 1 512 points = 5.9 x 256: ONE workgroup of four full
// wavefronts per CU (one per SIMD, up to 512 registers), six body rounds per batch at 98 % lane use, barriers only
// around the contraction stages of a whole batch (19 per seven elements instead of 49).
// What has to fit: the y-contracted arrays of a batch are 7 x 2 592 doubles for six fields - too much LDS - so the
// fields go in two passes: positions first (their z stage leaves the Jacobian of the thread's six points in
// registers), velocities second (their z stage runs inside the body round).  The stress of the six points waits in
// registers for the transposed contractions, which run one velocity component at a time through the same LDS.
#include "lgh_qrows.hpp"

namespace lgh
{

template <int NEB>
__global__ void __launch_bounds__(256, 1)
qbatch346_kernel(const QArgs a, const int nbatch)
{
   constexpr int D = 4, Q = 6, L = 3;
   constexpr int ND = 64, NQ = 216, NL = 27, DD = 16, QQ = 36, NT = 256;
   constexpr int NR = (NEB * NQ + NT - 1) / NT; // body rounds per batch
   constexpr int SU = NEB * 6 * ND;             // gathered fields [el][f][64]
   constexpr int SX = 2 * NEB * 3 * Q * DD;     // x-contracted, three fields [which][el][f][qx][dz][dy]
   constexpr int SY = NEB * 9 * QQ * D;         // y-contracted, three fields [el][part][f][qy][qx][dz]
   constexpr int SEa = NEB * NL, SE1 = NEB * L * L * Q, SE2 = NEB * L * QQ;
   constexpr int ST = 2 * Q * D + Q * L;
   __shared__ __attribute__((aligned(16))) double smem[SU + SX + SY + ST + SEa + SE1 + SE2];
   __shared__ double red[16];
   double *const sU = smem;
   double *const sX = sU + SU;
   double *const sY = sX + SX;
   double *const sTB = sY + SY;
   double *const sTG = sTB + Q * D;
   double *const sTL = sTG + Q * D;
   double *const sE = sTL + Q * L;   // [el][lz][ly][lx]
   double *const sE1 = sE + SEa;     // [el][lz][ly][qx]
   double *const sE2 = sE1 + SE1;    // [el][lz][qy][qx]
   // transposed contractions, one velocity component at a time
   double *const sF = sY;                       // [el][gd][qy][qx][qz]
   double *const sA = sY + NEB * 3 * NQ;        // [el][gd][dz][qx][qy]
   double *const sW = sX;                       // [el][gd][dz][dy][qx]
   double *const sS = sU;                       // [el][qy][qx][qz]
   double *const sT1 = sU + NEB * NQ;           // [el][lz][qy][qx]
   double *const sT2 = sT1 + NEB * L * QQ;      // [el][lz][ly][qx]
   static_assert(NEB * 3 * NQ + NEB * 3 * D * QQ <= SY && NEB * 3 * DD * Q <= SX, "LDS aliases of the transposed stages");
   static_assert(NEB * NQ + NEB * L * QQ + NEB * L * L * Q <= SU, "LDS aliases of the F^T v stages");

   const int tid = threadIdx.x;
   const size_t plane = (size_t)a.NE * NQ;
   for (int i = tid; i < Q * D; i += NT)
   {
      const int q = i / D, d = i - q * D;
      sTB[i] = a.B[q + Q * d];
      sTG[i] = a.G[q + Q * d];
   }
   for (int i = tid; i < Q * L; i += NT)
   {
      const int q = i / L, l = i - q * L;
      sTL[i] = a.Bl[q + Q * l];
   }
   const bool do_f = (a.force_e != nullptr), do_t = (a.erhs_q != nullptr);
   double cand = INFINITY;

   for (int vb = blockIdx.x; vb < nbatch; vb += gridDim.x)
   {
      const int b = xcd_swizzle(vb, nbatch);
      const int e0 = b * NEB;
      const int nel = min(NEB, a.NE - e0);
      const int npt = nel * NQ;
      // ---- gather the six fields and the energies of the batch
      for (int i = tid; i < nel * 3 * ND; i += NT)
      {
         const int el = i / (3 * ND), r = i - el * (3 * ND), c = r / ND, d = r - c * ND;
         const size_t n = (size_t)c * a.N + a.map[(size_t)(e0 + el) * ND + d];
         sU[(el * 6 + c) * ND + d] = a.x[n];
         sU[(el * 6 + 3 + c) * ND + d] = a.v[n];
      }
      for (int i = tid; i < nel * NL; i += NT) { sE[i] = a.e[(size_t)e0 * NL + i]; }
      __syncthreads();

      double Jr[NR][9], ev[NR];
      // ---- two passes over the fields: f0 = 0 positions, f0 = 3 velocities
#pragma unroll
      for (int pass = 0; pass < 2; pass++)
      {
         const int f0 = 3 * pass;
         // X stage: rows (el, which, f, dz, dy)
         for (int i = tid; i < nel * 2 * 3 * DD; i += NT)
         {
            const int el = i / (2 * 3 * DD), r0 = i - el * (2 * 3 * DD), which = r0 / (3 * DD), r = r0 - which * (3 * DD); // r = dy + D*(dz + D*f)
            const int f = r / DD, zy = r - f * DD;
            const double *T = sTB + which * (Q * D);
            double u[D], o[Q];
            row_load<D>(sU + (el * 6 + f0 + f) * ND + D * zy, u);
            row_fwd<D, Q>(T, u, o);
            double *dst = sX + ((which * NEB + el) * 3 + f) * (Q * DD) + zy;
#pragma unroll
            for (int q = 0; q < Q; q++) { dst[q * DD] = o[q]; }
         }
         if (pass == 0)
         {
            for (int i = tid; i < nel * L * L; i += NT) // energy: rows (el, lz, ly)
            {
               double u[L], o[Q];
               row_load<L>(sE + L * i, u);
               row_fwd<L, Q>(sTL, u, o);
#pragma unroll
               for (int q = 0; q < Q; q++) { sE1[i * Q + q] = o[q]; }
            }
         }
         __syncthreads();
         // Y stage: rows (el, part, f, qx, dz): part 0 = B on the G array (d/dx), 1 = G on the B array (d/dy), 2 = B on B (for d/dz)
         for (int i = tid; i < nel * 3 * 3 * Q * D; i += NT)
         {
            const int el = i / (9 * Q * D), r0 = i - el * (9 * Q * D), part = r0 / (3 * Q * D), r = r0 - part * (3 * Q * D); // r = dz + D*(qx + Q*f)
            const int f = r / (Q * D), xz = r - f * (Q * D);
            const double *T = sTB + ((part == 1) ? Q * D : 0);
            double u[D], o[Q];
            row_load<D>(sX + ((((part == 0) ? 1 : 0) * NEB + el) * 3 + f) * (Q * DD) + D * xz, u);
            row_fwd<D, Q>(T, u, o);
            double *dst = sY + ((el * 3 + part) * 3 + f) * (QQ * D) + xz;
#pragma unroll
            for (int q = 0; q < Q; q++) { dst[q * (Q * D)] = o[q]; }
         }
         if (pass == 0)
         {
            for (int i = tid; i < nel * L * Q; i += NT) // energy: rows (el, lz, qx)
            {
               const int qx = i % Q, ez = i / Q; // ez = lz + L*el
               double u[L], o[Q];
#pragma unroll
               for (int ly = 0; ly < L; ly++) { u[ly] = sE1[(ez * L + ly) * Q + qx]; }
               row_fwd<L, Q>(sTL, u, o);
#pragma unroll
               for (int q = 0; q < Q; q++) { sE2[(ez * Q + q) * Q + qx] = o[q]; }
            }
         }
         __syncthreads();
         if (pass == 0)
         {
            // Z stage of the positions: the Jacobian (and the energy) of this thread's points, kept in registers
#pragma unroll
            for (int r = 0; r < NR; r++)
            {
               const int p = min(tid + NT * r, npt - 1);
               const int el = p / NQ, q = p - el * NQ;
               const int tx = q % Q, ty = (q / Q) % Q, tz = q / QQ;
               double tb[D], tg[D];
               row_load<D>(sTB + D * tz, tb);
               row_load<D>(sTG + D * tz, tg);
               const double *col = sY + (el * 9) * (QQ * D) + (ty * Q + tx) * D;
#pragma unroll
               for (int f = 0; f < 3; f++)
               {
                  double gb[D], bg[D], bb[D];
                  row_load<D>(col + (0 * 3 + f) * (QQ * D), gb);
                  row_load<D>(col + (1 * 3 + f) * (QQ * D), bg);
                  row_load<D>(col + (2 * 3 + f) * (QQ * D), bb);
                  double d0 = tb[0] * gb[0], d1 = tb[0] * bg[0], d2 = tg[0] * bb[0];
#pragma unroll
                  for (int dz = 1; dz < D; dz++)
                  {
                     d0 = fma(tb[dz], gb[dz], d0);
                     d1 = fma(tb[dz], bg[dz], d1);
                     d2 = fma(tg[dz], bb[dz], d2);
                  }
                  Jr[r][f] = d0;
                  Jr[r][f + 3] = d1;
                  Jr[r][f + 6] = d2;
               }
               double e_val = 0.0;
#pragma unroll
               for (int lz = 0; lz < L; lz++) { e_val = fma(sTL[tz * L + lz], sE2[((el * L + lz) * Q + ty) * Q + tx], e_val); }
               ev[r] = e_val;
            }
            __syncthreads(); // the velocity pass overwrites sX / sY
         }
      }
      // ---- body rounds: z stage of the velocities, the point body; the stress of the six points stays in registers
      double sj[NR][9], ft[NR];
#pragma unroll
      for (int r = 0; r < NR; r++)
      {
         const int pr = tid + NT * r;
         const bool live = pr < npt;
         const int p = min(pr, npt - 1);
         const int el = p / NQ, q = p - el * NQ;
         const int tx = q % Q, ty = (q / Q) % Q, tz = q / QQ;
         const size_t eq = (size_t)(e0 + el) * NQ + q;
         double J0i[9];
#pragma unroll
         for (int k = 0; k < 9; k++) { J0i[k] = a.Jac0inv_soa[eq + plane * k]; }
         const double rdw = a.rho0DetJ0w_in[eq];
         const double weight = a.W[q];
         double tb[D], tg[D], dV[9];
         row_load<D>(sTB + D * tz, tb);
         row_load<D>(sTG + D * tz, tg);
         const double *col = sY + (el * 9) * (QQ * D) + (ty * Q + tx) * D;
#pragma unroll
         for (int f = 0; f < 3; f++)
         {
            double gb[D], bg[D], bb[D];
            row_load<D>(col + (0 * 3 + f) * (QQ * D), gb);
            row_load<D>(col + (1 * 3 + f) * (QQ * D), bg);
            row_load<D>(col + (2 * 3 + f) * (QQ * D), bb);
            double d0 = tb[0] * gb[0], d1 = tb[0] * bg[0], d2 = tg[0] * bb[0];
#pragma unroll
            for (int dz = 1; dz < D; dz++)
            {
               d0 = fma(tb[dz], gb[dz], d0);
               d1 = fma(tb[dz], bg[dz], d1);
               d2 = fma(tg[dz], bb[dz], d2);
            }
            dV[f] = d0;
            dV[f + 3] = d1;
            dV[f + 6] = d2;
         }
         double ftv = 0.0, sjw[9];
         // (lanes past the end of a ragged batch repeat its last point: the same values to the same addresses)
         const double c_p = qpoint_body<3>(a, e0 + el, eq, weight, Jr[r], dV, ev[r], plane, J0i, rdw, ftv, sjw);
         cand = fmin(cand, c_p);
         (void)live;
#pragma unroll
         for (int k = 0; k < 9; k++) { sj[r][k] = sjw[k]; }
         ft[r] = ftv;
         __builtin_amdgcn_sched_barrier(0); // (one point after the other)
      }
      __syncthreads(); // everybody is done with sY (and sU has been dead since the X stage of the velocities)
      if (do_f || do_t)
      {
#pragma unroll 1
         for (int c = 0; c < (do_f ? 3 : 1); c++)
         {
            // the stress of component c (and, with c = 0, the integrand of F^T v) to LDS
#pragma unroll
            for (int r = 0; r < NR; r++)
            {
               const int pr = tid + NT * r;
               if (pr < npt)
               {
                  const int el = pr / NQ, q = pr - el * NQ;
                  const int tx = q % Q, ty = (q / Q) % Q, tz = q / QQ;
                  const int pq = (ty * Q + tx) * Q + tz;
                  if (do_f)
                  {
#pragma unroll
                     for (int gd = 0; gd < 3; gd++)
                     {
                        // (component chosen with selects: sj is indexed by compile-time constants only)
                        const double v = (c == 0) ? sj[r][gd] : ((c == 1) ? sj[r][gd + 3] : sj[r][gd + 6]);
                        sF[(el * 3 + gd) * NQ + pq] = v;
                     }
                  }
                  if (do_t && c == 0) { sS[el * NQ + pq] = ft[r]; }
               }
            }
            __syncthreads();
            // contraction over qz: rows (el, gd, qy, qx); F^T v rows (el, qy, qx)
            if (do_f)
            {
               for (int i = tid; i < nel * 3 * QQ; i += NT)
               {
                  const int eg = i / QQ, r = i - eg * QQ, gd = eg % 3; // eg = gd + 3*el, r = qx + Q*qy
                  const double *T = sTB + ((gd == 2) ? Q * D : 0);
                  double u[Q], o[D];
                  row_load<Q>(sF + (size_t)i * Q, u);
                  row_bwd<Q, D>(T, u, o);
                  const int qx = r % Q, qy = r / Q;
                  double *dst = sA + eg * (D * QQ) + qx * Q + qy;
#pragma unroll
                  for (int d = 0; d < D; d++) { dst[d * QQ] = o[d]; }
               }
            }
            if (do_t && c == 0)
            {
               for (int i = tid; i < nel * QQ; i += NT)
               {
                  const int el = i / QQ, j = i - el * QQ;
                  double u[Q], o[L];
                  row_load<Q>(sS + i * Q, u);
                  row_bwd<Q, L>(sTL, u, o);
#pragma unroll
                  for (int l = 0; l < L; l++) { sT1[(el * L + l) * QQ + j] = o[l]; }
               }
            }
            __syncthreads();
            // contraction over qy: rows (el, gd, dz, qx); F^T v rows (el, lz, qx)
            if (do_f)
            {
               for (int i = tid; i < nel * 3 * D * Q; i += NT)
               {
                  const int eg = i / (D * Q), gd = eg % 3;
                  const double *T = sTB + ((gd == 1) ? Q * D : 0);
                  double u[Q], o[D];
                  row_load<Q>(sA + (size_t)i * Q, u);
                  row_bwd<Q, D>(T, u, o);
                  const int r = i - eg * (D * Q), qx = r % Q, dz = r / Q;
                  double *dst = sW + ((eg * D + dz) * D) * Q + qx;
#pragma unroll
                  for (int d = 0; d < D; d++) { dst[d * Q] = o[d]; }
               }
            }
            if (do_t && c == 0)
            {
               for (int i = tid; i < nel * L * Q; i += NT)
               {
                  const int qx = i % Q, ez = i / Q; // ez = lz + L*el
                  double u[Q], o[L];
#pragma unroll
                  for (int qy = 0; qy < Q; qy++) { u[qy] = sT1[(ez * Q + qy) * Q + qx]; }
                  row_bwd<Q, L>(sTL, u, o);
#pragma unroll
                  for (int l = 0; l < L; l++) { sT2[(ez * L + l) * Q + qx] = o[l]; }
               }
            }
            __syncthreads();
            // contraction over qx and the sum over the three reference directions: rows (el, dz, dy) -> E-vector
            if (do_f)
            {
               const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;
               for (int i = tid; i < nel * DD; i += NT)
               {
                  const int el = i / DD, r = i - el * DD; // r = dy + D*dz
                  double wg[Q], w1[Q], w2[Q], og[D], ob[D];
                  row_load<Q>(sW + (((el * 3 + 0) * DD) + r) * Q, wg);
                  row_load<Q>(sW + (((el * 3 + 1) * DD) + r) * Q, w1);
                  row_load<Q>(sW + (((el * 3 + 2) * DD) + r) * Q, w2);
#pragma unroll
                  for (int q = 0; q < Q; q++) { w1[q] += w2[q]; }
                  row_bwd<Q, D>(sTG, wg, og);
                  row_bwd<Q, D>(sTB, w1, ob);
                  double *dst = a.force_e + (size_t)ND * (c + 3 * (size_t)(e0 + el)) + D * r;
#pragma unroll
                  for (int d = 0; d < D; d++)
                  {
                     double v = og[d] + ob[d];
                     if (fabs(v) < eps2) { v = 0.0; } // laghos_assembly.cpp:495-512
                     dst[d] = v;
                  }
               }
            }
            if (do_t && c == 0)
            {
               for (int i = tid; i < nel * L * L; i += NT) // rows (el, lz, ly)
               {
                  double u[Q], o[L];
                  row_load<Q>(sT2 + i * Q, u);
                  row_bwd<Q, L>(sTL, u, o);
#pragma unroll
                  for (int l = 0; l < L; l++) { a.erhs_q[(size_t)e0 * NL + i * L + l] = o[l]; }
               }
            }
            __syncthreads(); // the next component (or the next batch) writes the same LDS
         }
      }
   }
   const double bmin = block_min(cand, red);
   double total;
   if (grid_min_last_block(bmin, a.partials, a.ticket, red, total))
   {
      if (tid == 0) { *a.result = fmin(*a.result, total); }
   }
}

static int launch_qbatch(lgh_ctx *c, const QArgs &a)
{
   constexpr int NEB = 7;
   if (c->ncu <= 0)
   {
      hipDeviceProp_t prop;
      c->ncu = (hipGetDeviceProperties(&prop, c->device) == hipSuccess) ? prop.multiProcessorCount : 256;
   }
   const int nbatch = ceil_div(c->NE, NEB);
   const int grid = std::min(nbatch, c->ncu);
   hipLaunchKernelGGL((qbatch346_kernel<NEB>), dim3(grid), dim3(256), 0, c->stream, a, nbatch);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

} // namespace lgh
