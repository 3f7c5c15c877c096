// lgh_force.hip — ForcePAOperator actions for gfx950.
//
// Replaces ForceMult{2D,3D} / ForceMultTranspose{2D,3D}
// (/root/reference/laghos_assembly.cpp:145-514, :567-924) and the restriction
// calls around them (:557-565, :965-973).
//
// MI355X design (3D): the kernel is an HBM stream of the nine stressJinvT planes
// (15.5 of 17.3 KB per Q3Q2 element), so the layout of work follows the loads:
//   * a workgroup owns NEB consecutive elements = NEB*NQ contiguous doubles of
//     every plane; thread (qx,qy,eb) owns the z-column {q = qx + Q*qy + Q*Q*qz},
//     so each load instruction of a wave covers contiguous Q*Q-double runs;
//   * all 3*Q loads of one velocity component are issued before they are used
//     (>= 18 outstanding 8-byte loads per lane at Q=6);
//   * the z contraction runs in registers with the 1-D tables held in SGPRs
//     (uniform scalar loads), only the x and y contractions go through LDS,
//     two barriers per component;
//   * results are written as an E-vector and summed into the L-vector by a
//     gather-based (deterministic, atomic-free) transpose, as MFEM's
//     ElementRestriction::MultTranspose does.
// Summation order inside an element is z,x,y instead of the reference's x,y,z:
// results agree to round-off (tests/ pin <= 1e-13 relative to the oracle).
#include "lgh_common.hpp"

namespace lgh
{

// ---------------------------------------------------------------------------
// 3D  y_E = F x_E
// ---------------------------------------------------------------------------
template <int D, int Q, int L, int NEB>
__global__ void __launch_bounds__(Q *Q *NEB, (Q <= 6) ? 3 : 2)
force_mult_3d(const int NE, const double *__restrict__ Bl, // [q + Q*l]
              const double *__restrict__ B,                 // [q + Q*d]
              const double *__restrict__ G, const double *__restrict__ sJit,
              const double *__restrict__ xE, double *__restrict__ yE)
{
   constexpr int NQ = Q * Q * Q, ND = D * D * D, NL = L * L * L;
   constexpr int S1 = 3 * D * Q * Q; // [gd][dz][qy][qx]
   constexpr int S2 = 3 * D * Q * D; // [gd][dz][qy][dx]
   constexpr int SE = NL + L * L * Q; // E and LQ[lz][ly][qx]
   constexpr int PER = S1 + S2 + SE + 1; // +1: odd stride spreads banks
   __shared__ double smem[NEB * PER];

   const int tid = threadIdx.x;
   const int tx = tid % Q, ty = (tid / Q) % Q, eb = tid / (Q * Q);
   const int e = blockIdx.x * NEB + eb;
   const bool active = (e < NE);
   double *s1 = smem + eb * PER;
   double *s2 = s1 + S1;
   double *sE = s2 + S2;
   double *sLQ = sE + NL;
   const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;

   // per-thread rows of the H1 tables: row tx / ty of B^T, G^T when tx,ty < D
   double btx[Q], gtx[Q], bty[Q], gty[Q];
#pragma unroll
   for (int q = 0; q < Q; q++)
   {
      btx[q] = (tx < D) ? B[q + Q * tx] : 0.0;
      gtx[q] = (tx < D) ? G[q + Q * tx] : 0.0;
      bty[q] = (ty < D) ? B[q + Q * ty] : 0.0;
      gty[q] = (ty < D) ? G[q + Q * ty] : 0.0;
   }

   // ---- energy at the quadrature points: QQQ[qz] for this (qx,qy) column
   if (active)
   {
      for (int i = tx + Q * ty; i < NL; i += Q * Q) { sE[i] = xE[i + (size_t)NL * e]; }
   }
   __syncthreads();
   if (active && ty < L)
   {
#pragma unroll
      for (int lz = 0; lz < L; lz++)
      {
         double u = 0.0;
#pragma unroll
         for (int lx = 0; lx < L; lx++) { u += Bl[tx + Q * lx] * sE[lx + L * (ty + L * lz)]; }
         sLQ[tx + Q * (ty + L * lz)] = u;
      }
   }
   __syncthreads();
   double qqq[Q];
   {
      double qq[L];
#pragma unroll
      for (int lz = 0; lz < L; lz++)
      {
         double u = 0.0;
#pragma unroll
         for (int ly = 0; ly < L; ly++) { u += Bl[ty + Q * ly] * sLQ[tx + Q * (ly + L * lz)]; }
         qq[lz] = u;
      }
#pragma unroll
      for (int qz = 0; qz < Q; qz++)
      {
         double u = 0.0;
#pragma unroll
         for (int lz = 0; lz < L; lz++) { u += Bl[qz + Q * lz] * qq[lz]; }
         qqq[qz] = u;
      }
   }

   const size_t plane = (size_t)NE * NQ;
   const size_t col = (size_t)e * NQ + tx + Q * ty;
   // one component at a time: unrolling lets the scheduler hoist all 54 plane loads
   // and blows the register budget (256 VGPRs, 1 wave/SIMD)
#pragma unroll 1
   for (int c = 0; c < 3; c++)
   {
      // stream the three planes (gd = 0,1,2) of component c for this column
      double t0[Q], t1[Q], t2[Q];
      if (active)
      {
         const double *p0 = sJit + plane * (0 + 3 * c) + col;
         const double *p1 = sJit + plane * (1 + 3 * c) + col;
         const double *p2 = sJit + plane * (2 + 3 * c) + col;
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            t0[qz] = p0[Q * Q * qz];
            t1[qz] = p1[Q * Q * qz];
            t2[qz] = p2[Q * Q * qz];
         }
      }
      else
      {
#pragma unroll
         for (int qz = 0; qz < Q; qz++) { t0[qz] = t1[qz] = t2[qz] = 0.0; }
      }
      // z contraction in registers: gd 0,1 with B^T, gd 2 with G^T
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
         double u = 0.0, v = 0.0, w = 0.0;
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            const double b = B[qz + Q * dz], g = G[qz + Q * dz];
            u += b * (qqq[qz] * t0[qz]);
            v += b * (qqq[qz] * t1[qz]);
            w += g * (qqq[qz] * t2[qz]);
         }
         s1[tx + Q * (ty + Q * (dz + D * 0))] = u;
         s1[tx + Q * (ty + Q * (dz + D * 1))] = v;
         s1[tx + Q * (ty + Q * (dz + D * 2))] = w;
      }
      __syncthreads();
      // x contraction: thread (dx = tx < D, qy = ty): gd 0 with G^T, gd 1,2 with B^T
      if (tx < D)
      {
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            double u = 0.0, v = 0.0, w = 0.0;
#pragma unroll
            for (int qx = 0; qx < Q; qx++)
            {
               u += gtx[qx] * s1[qx + Q * (ty + Q * (dz + D * 0))];
               v += btx[qx] * s1[qx + Q * (ty + Q * (dz + D * 1))];
               w += btx[qx] * s1[qx + Q * (ty + Q * (dz + D * 2))];
            }
            s2[tx + D * (ty + Q * (dz + D * 0))] = u;
            s2[tx + D * (ty + Q * (dz + D * 1))] = v;
            s2[tx + D * (ty + Q * (dz + D * 2))] = w;
         }
      }
      __syncthreads();
      // y contraction: thread (dx = tx < D, dy = ty < D): gd 1 with G^T, gd 0,2 with B^T
      if (tx < D && ty < D && active)
      {
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            double u = 0.0, v = 0.0, w = 0.0;
#pragma unroll
            for (int qy = 0; qy < Q; qy++)
            {
               u += bty[qy] * s2[tx + D * (qy + Q * (dz + D * 0))];
               v += gty[qy] * s2[tx + D * (qy + Q * (dz + D * 1))];
               w += bty[qy] * s2[tx + D * (qy + Q * (dz + D * 2))];
            }
            double r = u + v + w;
            if (fabs(r) < eps2) { r = 0.0; } // laghos_assembly.cpp:495-512
            yE[tx + D * (ty + D * dz) + (size_t)ND * (c + 3 * (size_t)e)] = r;
         }
      }
      // s1 of the next component is written only after the barrier that followed
      // its last reads; s2 is re-written after the next barrier: no hazard.
   }
}

// ---------------------------------------------------------------------------
// 3D  y_l2 = F^T v   (v given as L-vector through the gather map, or as E-vector)
// ---------------------------------------------------------------------------
template <int D, int Q, int L, int NEB>
__global__ void __launch_bounds__(Q *Q *NEB, (Q <= 6) ? 3 : 2)
force_mult_t_3d(const int NE, const int N, const double *__restrict__ Bl,
                const double *__restrict__ B, const double *__restrict__ G,
                const double *__restrict__ sJit, const double *__restrict__ v,
                const int *__restrict__ map /* null: v is an E-vector */,
                double *__restrict__ y, const int *__restrict__ guard /* non-null: run only when *guard != 0 */)
{
   if (guard && *guard == 0) { return; }
   constexpr int NQ = Q * Q * Q, ND = D * D * D, NL = L * L * L;
   constexpr int SV = 3 * ND;        // V[c][dz][dy][dx]
   constexpr int SA = 2 * D * D * Q; // Bx/Gx [k][dz][dy][qx]
   constexpr int ST1 = L * Q * Q;    // [lz][qy][qx]
   constexpr int ST2 = L * Q * L;    // [lz][qy][lx]
   constexpr int SW = (SA > ST1 + ST2) ? SA : (ST1 + ST2);
   constexpr int PER = SV + SW + 1;
   __shared__ double smem[NEB * PER];

   const int tid = threadIdx.x;
   const int tx = tid % Q, ty = (tid / Q) % Q, eb = tid / (Q * Q);
   const int e0 = blockIdx.x * NEB;
   const int e = e0 + eb;
   const bool active = (e < NE);
   double *sV = smem + eb * PER;
   double *sA = sV + SV;

   // cooperative gather of the block's velocity dofs (fused H1R->Mult)
   {
      const int nthr = Q * Q * NEB;
      const int nel = min(NEB, NE - e0);
      for (int i = tid; i < nel * SV; i += nthr)
      {
         const int el = i / SV, r = i % SV, c = r / ND, d = r % ND;
         const size_t eg = (size_t)(e0 + el);
         double val;
         if (map) { val = v[(size_t)c * N + map[eg * ND + d]]; }
         else { val = v[d + (size_t)ND * (c + 3 * eg)]; }
         smem[el * PER + r] = val;
      }
   }
   double bx[D], gx[D], by[D], gy[D];
#pragma unroll
   for (int d = 0; d < D; d++)
   {
      bx[d] = B[tx + Q * d];
      gx[d] = G[tx + Q * d];
      by[d] = B[ty + Q * d];
      gy[d] = G[ty + Q * d];
   }
   __syncthreads();

   double acc[Q];
#pragma unroll
   for (int qz = 0; qz < Q; qz++) { acc[qz] = 0.0; }
   const size_t plane = (size_t)NE * NQ;
   const size_t col = (size_t)e * NQ + tx + Q * ty;

   // one component at a time: unrolling lets the scheduler hoist all 54 plane loads
   // and blows the register budget (256 VGPRs, 1 wave/SIMD)
#pragma unroll 1
   for (int c = 0; c < 3; c++)
   {
      // issue this component's plane loads early
      double t0[Q], t1[Q], t2[Q];
      if (active)
      {
         const double *p0 = sJit + plane * (0 + 3 * c) + col;
         const double *p1 = sJit + plane * (1 + 3 * c) + col;
         const double *p2 = sJit + plane * (2 + 3 * c) + col;
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            t0[qz] = p0[Q * Q * qz];
            t1[qz] = p1[Q * Q * qz];
            t2[qz] = p2[Q * Q * qz];
         }
      }
      else
      {
#pragma unroll
         for (int qz = 0; qz < Q; qz++) { t0[qz] = t1[qz] = t2[qz] = 0.0; }
      }
      // x stage: thread (qx = tx, dy = ty < D), loop dz
      if (ty < D)
      {
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            double u = 0.0, w = 0.0;
#pragma unroll
            for (int dx = 0; dx < D; dx++)
            {
               const double in = sV[dx + D * (ty + D * dz) + ND * c];
               u += bx[dx] * in;
               w += gx[dx] * in;
            }
            sA[tx + Q * (ty + D * (dz + D * 0))] = u; // B in x
            sA[tx + Q * (ty + D * (dz + D * 1))] = w; // G in x
         }
      }
      __syncthreads();
      // y stage in registers per dz, then z stage: reference gradient at (qx,qy,qz)
      double gb[D], bg[D], bb[D]; // d/dx, d/dy, value-in-xy
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
         double u = 0.0, w = 0.0, z = 0.0;
#pragma unroll
         for (int dy = 0; dy < D; dy++)
         {
            const double vb = sA[tx + Q * (dy + D * (dz + D * 0))];
            const double vg = sA[tx + Q * (dy + D * (dz + D * 1))];
            u += by[dy] * vg; // d/dxi_0
            w += gy[dy] * vb; // d/dxi_1
            z += by[dy] * vb; // for d/dxi_2
         }
         gb[dz] = u;
         bg[dz] = w;
         bb[dz] = z;
      }
#pragma unroll
      for (int qz = 0; qz < Q; qz++)
      {
         double g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            const double b = B[qz + Q * dz], g = G[qz + Q * dz];
            g0 += b * gb[dz];
            g1 += b * bg[dz];
            g2 += g * bb[dz];
         }
         acc[qz] += g0 * t0[qz] + g1 * t1[qz] + g2 * t2[qz];
      }
      __syncthreads(); // sA is rewritten by the next component
   }

   // ---- test with the L2 basis: z in registers, x and y through LDS
   double *sT1 = sA;
   double *sT2 = sA + ST1;
#pragma unroll
   for (int lz = 0; lz < L; lz++)
   {
      double u = 0.0;
#pragma unroll
      for (int qz = 0; qz < Q; qz++) { u += Bl[qz + Q * lz] * acc[qz]; }
      sT1[tx + Q * (ty + Q * lz)] = u;
   }
   __syncthreads();
   if (tx < L)
   {
#pragma unroll
      for (int lz = 0; lz < L; lz++)
      {
         double u = 0.0;
#pragma unroll
         for (int qx = 0; qx < Q; qx++) { u += Bl[qx + Q * tx] * sT1[qx + Q * (ty + Q * lz)]; }
         sT2[tx + L * (ty + Q * lz)] = u;
      }
   }
   __syncthreads();
   if (tx < L && ty < L && active)
   {
#pragma unroll
      for (int lz = 0; lz < L; lz++)
      {
         double u = 0.0;
#pragma unroll
         for (int qy = 0; qy < Q; qy++) { u += Bl[qy + Q * ty] * sT2[tx + L * (qy + Q * lz)]; }
         y[tx + L * (ty + L * lz) + (size_t)NL * e] = u;
      }
   }
}

// ---------------------------------------------------------------------------
// 2D kernels: one thread per quadrature point, NEB elements per workgroup.
// (Configuration C1 is a plumbing / |e| check; these are not tuned.)
// ---------------------------------------------------------------------------
template <int D, int Q, int L, int NEB>
__global__ void __launch_bounds__(Q *Q *NEB)
force_mult_2d(const int NE, const double *__restrict__ Bl, const double *__restrict__ B,
              const double *__restrict__ G, const double *__restrict__ sJit,
              const double *__restrict__ xE, double *__restrict__ yE)
{
   constexpr int NQ = Q * Q, ND = D * D, NL = L * L;
   constexpr int PER = NL + L * Q + 2 * NQ + 2 * D * Q + 1;
   __shared__ double smem[NEB * PER];
   const int tid = threadIdx.x;
   const int tx = tid % Q, ty = (tid / Q) % Q, eb = tid / (Q * Q);
   const int e = blockIdx.x * NEB + eb;
   const bool active = (e < NE);
   double *sE = smem + eb * PER;
   double *sLQ = sE + NL;      // [ly][qx]
   double *sQ0 = sLQ + L * Q;  // [qy][qx]
   double *sQ1 = sQ0 + NQ;
   double *sD0 = sQ1 + NQ;     // [qy][dx]
   double *sD1 = sD0 + D * Q;
   const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;

   if (active)
   {
      for (int i = tx + Q * ty; i < NL; i += NQ) { sE[i] = xE[i + (size_t)NL * e]; }
   }
   __syncthreads();
   if (ty < L)
   {
      double u = 0.0;
      for (int lx = 0; lx < L; lx++) { u += Bl[tx + Q * lx] * sE[lx + L * ty]; }
      sLQ[tx + Q * ty] = u;
   }
   __syncthreads();
   double qq = 0.0;
   for (int ly = 0; ly < L; ly++) { qq += Bl[ty + Q * ly] * sLQ[tx + Q * ly]; }
   const size_t plane = (size_t)NE * NQ;
   const size_t qi = (size_t)e * NQ + tx + Q * ty;
   for (int c = 0; c < 2; c++)
   {
      const double s0 = active ? sJit[plane * (0 + 2 * c) + qi] : 0.0;
      const double s1 = active ? sJit[plane * (1 + 2 * c) + qi] : 0.0;
      __syncthreads();
      sQ0[tx + Q * ty] = qq * s0;
      sQ1[tx + Q * ty] = qq * s1;
      __syncthreads();
      if (tx < D)
      {
         double u = 0.0, v = 0.0;
         for (int qx = 0; qx < Q; qx++)
         {
            u += G[qx + Q * tx] * sQ0[qx + Q * ty];
            v += B[qx + Q * tx] * sQ1[qx + Q * ty];
         }
         sD0[tx + D * ty] = u;
         sD1[tx + D * ty] = v;
      }
      __syncthreads();
      if (tx < D && ty < D && active)
      {
         double u = 0.0, v = 0.0;
         for (int qy = 0; qy < Q; qy++)
         {
            u += sD0[tx + D * qy] * B[qy + Q * ty];
            v += sD1[tx + D * qy] * G[qy + Q * ty];
         }
         double r = u + v;
         if (fabs(r) < eps2) { r = 0.0; }
         yE[tx + D * ty + (size_t)ND * (c + 2 * (size_t)e)] = r;
      }
   }
}

template <int D, int Q, int L, int NEB>
__global__ void __launch_bounds__(Q *Q *NEB)
force_mult_t_2d(const int NE, const int N, const double *__restrict__ Bl,
                const double *__restrict__ B, const double *__restrict__ G,
                const double *__restrict__ sJit, const double *__restrict__ v,
                const int *__restrict__ map, double *__restrict__ y, const int *__restrict__ guard)
{
   if (guard && *guard == 0) { return; }
   constexpr int NQ = Q * Q, ND = D * D, NL = L * L;
   constexpr int PER = 2 * ND + 2 * D * Q + Q * L + 1;
   __shared__ double smem[NEB * PER];
   const int tid = threadIdx.x;
   const int tx = tid % Q, ty = (tid / Q) % Q, eb = tid / (Q * Q);
   const int e0 = blockIdx.x * NEB;
   const int e = e0 + eb;
   const bool active = (e < NE);
   double *sV = smem + eb * PER; // [c][dy][dx]
   double *sB = sV + 2 * ND;     // [dy][qx] B in x
   double *sG = sB + D * Q;      // [dy][qx] G in x
   double *sQL = sG + D * Q;     // [qy][lx]
   {
      const int nthr = Q * Q * NEB;
      const int nel = min(NEB, NE - e0);
      for (int i = tid; i < nel * 2 * ND; i += nthr)
      {
         const int el = i / (2 * ND), r = i % (2 * ND), c = r / ND, d = r % ND;
         const size_t eg = (size_t)(e0 + el);
         smem[el * PER + r] = map ? v[(size_t)c * N + map[eg * ND + d]] : v[d + (size_t)ND * (c + 2 * eg)];
      }
   }
   __syncthreads();
   const size_t plane = (size_t)NE * NQ;
   const size_t qi = (size_t)e * NQ + tx + Q * ty;
   double acc = 0.0;
   for (int c = 0; c < 2; c++)
   {
      if (ty < D)
      {
         double u = 0.0, w = 0.0;
         for (int dx = 0; dx < D; dx++)
         {
            const double in = sV[dx + D * ty + ND * c];
            u += B[tx + Q * dx] * in;
            w += G[tx + Q * dx] * in;
         }
         sB[tx + Q * ty] = u;
         sG[tx + Q * ty] = w;
      }
      __syncthreads();
      double g0 = 0.0, g1 = 0.0;
      for (int dy = 0; dy < D; dy++)
      {
         g0 += sG[tx + Q * dy] * B[ty + Q * dy];
         g1 += sB[tx + Q * dy] * G[ty + Q * dy];
      }
      if (active) { acc += g0 * sJit[plane * (0 + 2 * c) + qi] + g1 * sJit[plane * (1 + 2 * c) + qi]; }
      __syncthreads();
   }
   // L2 test: reuse sB as QQ[qy][qx]
   double *sQQ = sB;
   sQQ[tx + Q * ty] = acc;
   __syncthreads();
   if (tx < L)
   {
      double u = 0.0;
      for (int qx = 0; qx < Q; qx++) { u += sQQ[qx + Q * ty] * Bl[qx + Q * tx]; }
      sQL[tx + L * ty] = u;
   }
   __syncthreads();
   if (tx < L && ty < L && active)
   {
      double u = 0.0;
      for (int qy = 0; qy < Q; qy++) { u += sQL[tx + L * qy] * Bl[qy + Q * ty]; }
      y[tx + L * ty + (size_t)NL * e] = u;
   }
}

// ---------------------------------------------------------------------------
// E -> L transpose of the lexicographic restriction: deterministic gather over
// the CSR transpose built at context creation (ascending element order).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
h1_transpose_gather_k(const int N, const int ncomp, const int ND,
                      const int *__restrict__ off, const int *__restrict__ idx,
                      const double *__restrict__ YE, double *__restrict__ yL)
{
   const int n = blockIdx.x * blockDim.x + threadIdx.x;
   if (n >= N) { return; }
   const int k0 = off[n], k1 = off[n + 1];
   for (int c = 0; c < ncomp; c++)
   {
      double s = 0.0;
      for (int k = k0; k < k1; k++)
      {
         const int p = idx[k]; // e*ND + d
         const int e = p / ND, d = p - e * ND;
         s += YE[d + (size_t)ND * (c + ncomp * (size_t)e)];
      }
      yL[(size_t)c * N + n] = s;
   }
}

int h1_transpose_gather(lgh_ctx *c, int ncomp, const double *YE, double *yL)
{
   hipLaunchKernelGGL(h1_transpose_gather_k, dim3(ceil_div(c->N, 256)), dim3(256), 0, c->stream,
                      c->N, ncomp, c->ND, c->t_off, c->t_idx, YE, yL);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

// ---- dispatch: same ids as the reference tables (assembly.cpp:534-548, :944-956)
static int unknown_kernel(int id)
{
   set_error("Unknown kernel 0x%x", id);
   return LGH_ERR_UNSUPPORTED;
}

template <int D, int Q, int L> constexpr int neb3() { return (256 / (Q * Q)) > 0 ? (256 / (Q * Q)) : 1; }

#define LGH_LAUNCH_3D(K, D_, Q_, L_, ...)                                                     \
   {                                                                                          \
      constexpr int NEB_ = neb3<D_, Q_, L_>();                                                \
      hipLaunchKernelGGL((K<D_, Q_, L_, NEB_>), dim3(ceil_div(c->NE, NEB_)),                  \
                         dim3(Q_ * Q_ * NEB_), 0, c->stream, __VA_ARGS__);                    \
   }

int force_mult_E(lgh_ctx *c, const double *sJit, const double *xE, double *yE)
{
   switch (c->kid)
   {
      case 0x222: LGH_LAUNCH_3D(force_mult_2d, 2, 2, 1, c->NE, c->Bl, c->B, c->G, sJit, xE, yE); break;
      case 0x234: LGH_LAUNCH_3D(force_mult_2d, 3, 4, 2, c->NE, c->Bl, c->B, c->G, sJit, xE, yE); break;
      case 0x246: LGH_LAUNCH_3D(force_mult_2d, 4, 6, 3, c->NE, c->Bl, c->B, c->G, sJit, xE, yE); break;
      case 0x258: LGH_LAUNCH_3D(force_mult_2d, 5, 8, 4, c->NE, c->Bl, c->B, c->G, sJit, xE, yE); break;
      case 0x26A: LGH_LAUNCH_3D(force_mult_2d, 6, 10, 5, c->NE, c->Bl, c->B, c->G, sJit, xE, yE); break;
      case 0x322: LGH_LAUNCH_3D(force_mult_3d, 2, 2, 1, c->NE, c->Bl, c->B, c->G, sJit, xE, yE); break;
      case 0x334: LGH_LAUNCH_3D(force_mult_3d, 3, 4, 2, c->NE, c->Bl, c->B, c->G, sJit, xE, yE); break;
      case 0x346: LGH_LAUNCH_3D(force_mult_3d, 4, 6, 3, c->NE, c->Bl, c->B, c->G, sJit, xE, yE); break;
      case 0x358: LGH_LAUNCH_3D(force_mult_3d, 5, 8, 4, c->NE, c->Bl, c->B, c->G, sJit, xE, yE); break;
      // 0x36A is not instantiated by the reference (assembly.cpp:544-547); extension.
      case 0x36A: LGH_LAUNCH_3D(force_mult_3d, 6, 10, 5, c->NE, c->Bl, c->B, c->G, sJit, xE, yE); break;
      default: return unknown_kernel(c->kid);
   }
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

static int force_mult_t_any(lgh_ctx *c, const double *sJit, const double *v, const int *map, double *y, const int *guard)
{
   switch (c->kid)
   {
      case 0x222: LGH_LAUNCH_3D(force_mult_t_2d, 2, 2, 1, c->NE, c->N, c->Bl, c->B, c->G, sJit, v, map, y, guard); break;
      case 0x234: LGH_LAUNCH_3D(force_mult_t_2d, 3, 4, 2, c->NE, c->N, c->Bl, c->B, c->G, sJit, v, map, y, guard); break;
      case 0x246: LGH_LAUNCH_3D(force_mult_t_2d, 4, 6, 3, c->NE, c->N, c->Bl, c->B, c->G, sJit, v, map, y, guard); break;
      case 0x258: LGH_LAUNCH_3D(force_mult_t_2d, 5, 8, 4, c->NE, c->N, c->Bl, c->B, c->G, sJit, v, map, y, guard); break;
      case 0x26A: LGH_LAUNCH_3D(force_mult_t_2d, 6, 10, 5, c->NE, c->N, c->Bl, c->B, c->G, sJit, v, map, y, guard); break;
      case 0x322: LGH_LAUNCH_3D(force_mult_t_3d, 2, 2, 1, c->NE, c->N, c->Bl, c->B, c->G, sJit, v, map, y, guard); break;
      case 0x334: LGH_LAUNCH_3D(force_mult_t_3d, 3, 4, 2, c->NE, c->N, c->Bl, c->B, c->G, sJit, v, map, y, guard); break;
      case 0x346: LGH_LAUNCH_3D(force_mult_t_3d, 4, 6, 3, c->NE, c->N, c->Bl, c->B, c->G, sJit, v, map, y, guard); break;
      case 0x358: LGH_LAUNCH_3D(force_mult_t_3d, 5, 8, 4, c->NE, c->N, c->Bl, c->B, c->G, sJit, v, map, y, guard); break;
      case 0x36A: LGH_LAUNCH_3D(force_mult_t_3d, 6, 10, 5, c->NE, c->N, c->Bl, c->B, c->G, sJit, v, map, y, guard); break;
      default: return unknown_kernel(c->kid);
   }
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
// guard: device flag; the kernel does nothing when it is 0 (lgh_solve_energy: the fused F^T v is current)
int force_mult_t_L(lgh_ctx *c, const double *sJit, const double *v_h1, double *y_l2, const int *guard)
{
   return force_mult_t_any(c, sJit, v_h1, c->h1map, y_l2, guard);
}
int force_mult_t_E(lgh_ctx *c, const double *sJit, const double *vE, double *y_l2)
{
   return force_mult_t_any(c, sJit, vE, nullptr, y_l2, nullptr);
}

} // namespace lgh
