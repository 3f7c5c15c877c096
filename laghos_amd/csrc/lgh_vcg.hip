// lgh_vcg.hip — the three velocity-component mass solves of SolveVelocity run in
// lockstep on the device.
//
// Reference: /root/reference/laghos_solver.cpp:363-398 calls CG_VMass.Mult once
// per velocity component, each a Jacobi-PCG on the same scalar H1 mass operator
// with its own right-hand side and essential-dof list.  Here the dim independent
// CGSolver::Mult recurrences advance together: every component keeps
// its own alpha, beta, residual norms, convergence flag and iteration count, and
// performs exactly the operations of its stand-alone solve in the same order —
// only the kernels are shared.
//
// Why (MI355X): the quadrature data D = w detJ0 rho0 is 63 % of the bytes of a mass
// apply and the element->node maps are another 10 %; applying the operator to the
// three directions in one pass reads them once instead of three times, halves the
// launches, and amortises the grid reduction (one ticket, three sums).
#include "lgh_vcg.hpp"

namespace lgh
{
typedef unsigned v4u_ __attribute__((ext_vector_type(4)));


// ---- K1: y_e^c = B^T D_e B d_e^c for the unconverged components, d^c = z^c + beta_c d^c.
// Persistent workgroups: the grid is one resident wave of workgroups; each walks
// batches b, b+G, ... of NEB elements.  Profiling the one-batch-per-workgroup form
// showed it bound by the per-workgroup latency chain (map -> gather -> 4 barriers
// -> store -> partial/ticket round trip) times 2.3 waves of workgroups, not by
// bytes or FMAs.  Here the map entries, the gathers of all components and the
// quadrature data of batch i+1 are in flight while batch i is contracted, and
// the dot-product partials are published once per workgroup.
template <int D, int Q, int NEB>
__global__ void __launch_bounds__(Q *Q *NEB, (Q >= 8) ? 1 : 2)
vcg_apply_3d(const VcgArgs a, const int nbatch)
{
   constexpr int NQ = Q * Q * Q, ND = D * D * D;
   constexpr int SX = (ND > D * Q * Q) ? ND : D * Q * Q;
   constexpr int SA = D * D * Q;
   constexpr int SAE = (SA > D * Q * D) ? SA : D * Q * D;
   constexpr int OFF_A = kVC * SX, OFF_IN = OFF_A + kVC * SAE;
   constexpr int PER = OFF_IN + kVC * ND + 1; // per component: C, A/E work buffers, gathered direction
   constexpr int NT = Q * Q * NEB;
   constexpr int GPT = (NEB * ND + NT - 1) / NT; // gather items per thread
   __shared__ double smem[NEB * PER];
   __shared__ double red[48];

   if (a.s->all_done) { return; }
   const int tid = threadIdx.x;
   const int tx = tid % Q, ty = (tid / Q) % Q, eb = tid / (Q * Q);
   const int G = gridDim.x;
   double *sX = smem + eb * PER;  // [c][SX]
   double *sA = sX + OFF_A;       // [c][SAE]
   double *sIn = sX + OFF_IN;     // [c][ND]
   const bool first = a.s->first != 0;
   bool todo[kVC];
   double beta[kVC];
#pragma unroll
   for (int c = 0; c < kVC; c++) { todo[c] = a.s->done[c] == 0; }
   if (a.multi && !first && !vcg_pending_update(a.s, a.iter, blockIdx.x == 0 && tid == 0, todo)) { return; }
#pragma unroll
   for (int c = 0; c < kVC; c++) { beta[c] = (first || !todo[c]) ? 0.0 : a.s->rz[c] / a.s->rz_prev[c]; }
   // the 1-D table lives in LDS (thread-dependent rows are read at use: the
   // persistent kernel is register-, not LDS-limited)
   __shared__ double sB[Q * D];
   for (int i = tid; i < Q * D; i += NT) { sB[i] = a.B[i]; }
   const double *bxp = sB + tx;         // B[tx + Q*d]
   const double *byp = sB + ty;         // B[ty + Q*d]
   const double *brxp = sB + Q * (tx < D ? tx : 0); // B[q + Q*tx]
   const double *bryp = sB + Q * (ty < D ? ty : 0); // B[q + Q*ty]

   // in-flight state of the NEXT batch
   int mi[GPT];
   double gz[kVC][GPT], gd[kVC][GPT], gi[GPT], dqn[Q];
   auto load_map = [&](const int b) {
      const int e0 = b * NEB, nel = min(NEB, a.NE - e0);
#pragma unroll
      for (int k = 0; k < GPT; k++)
      {
         const int i = tid + k * NT;
         mi[k] = (i < nel * ND) ? a.map[(size_t)e0 * ND + i] : -1;
      }
   };
   auto load_data = [&](const int b) {
      const int e = b * NEB + eb;
#pragma unroll
      for (int k = 0; k < GPT; k++) { gi[k] = (mi[k] >= 0) ? a.dinv[mi[k]] : 0.0; }
#pragma unroll
      for (int c = 0; c < kVC; c++)
      {
         if (!todo[c]) { continue; }
#pragma unroll
         for (int k = 0; k < GPT; k++)
         {
            gz[c][k] = (mi[k] >= 0) ? a.r[(size_t)c * a.N + mi[k]] : 0.0;
            gd[c][k] = (mi[k] >= 0 && !first) ? a.d[(size_t)c * a.N + mi[k]] : 0.0;
         }
      }
      if (e < a.NE)
      {
         const double *p = a.DqFull + (size_t)e * NQ + tx + Q * ty;
#pragma unroll
         for (int qz = 0; qz < Q; qz++) { dqn[qz] = p[Q * Q * qz]; }
      }
      else
      {
#pragma unroll
         for (int qz = 0; qz < Q; qz++) { dqn[qz] = 0.0; }
      }
   };

   double dots[kVC] = {0.0, 0.0, 0.0};
   int b = xcd_swizzle(blockIdx.x, G);
   if (b < nbatch)
   {
      load_map(b);
      load_data(b);
   }
   for (; b < nbatch; b += G)
   {
      const int e0 = b * NEB;
      const int e = e0 + eb;
      const bool active = (e < a.NE);
      // take ownership of the prefetched batch: quadrature data to registers, the
      // directions d = z + beta d of every active component to LDS (K2 stores the
      // same values later); after that the prefetch registers are free again
      double dq[Q];
#pragma unroll
      for (int qz = 0; qz < Q; qz++) { dq[qz] = dqn[qz]; }
      __syncthreads(); // previous batch finished with the LDS buffers
#pragma unroll
      for (int c = 0; c < kVC; c++)
      {
         if (!todo[c]) { continue; }
#pragma unroll
         for (int k = 0; k < GPT; k++)
         {
            const int i = tid + k * NT;
            if (mi[k] >= 0)
            {
               const int el = i / ND, dd = i - el * ND;
               smem[el * PER + OFF_IN + c * ND + dd] = fma(beta[c], gd[c][k], __dmul_rn(gz[c][k], gi[k]));
            }
         }
      }
      const int bn = b + G;
      const bool have_next = bn < nbatch;
      if (have_next) { load_map(bn); } // its gathers are issued one stage later
      __syncthreads();
      // All active components advance through each stage together: 5 barriers per
      // batch instead of 5 per component.
      // forward x: thread (qx = tx, dy = ty < D)
      double xcol[kVC][D];
#pragma unroll
      for (int c = 0; c < kVC; c++)
      {
         if (!todo[c]) { continue; }
         const double *sXc = sIn + c * ND;
         double *sAc = sA + c * SAE;
#pragma unroll
         for (int dz = 0; dz < D; dz++) { xcol[c][dz] = (tx < D && ty < D) ? sXc[tx + D * (ty + D * dz)] : 0.0; }
         if (ty < D)
         {
#pragma unroll
            for (int dz = 0; dz < D; dz++)
            {
               double u = 0.0;
#pragma unroll
               for (int dx = 0; dx < D; dx++) { u += bxp[Q * dx] * sXc[dx + D * (ty + D * dz)]; }
               sAc[tx + Q * (ty + D * dz)] = u;
            }
         }
      }
      __syncthreads();
      if (have_next) { load_data(bn); } // map entries have had a stage to arrive
      // forward y, z; scale by the quadrature data; backward z (registers)
#pragma unroll
      for (int c = 0; c < kVC; c++)
      {
         if (!todo[c]) { continue; }
         const double *sAc = sA + c * SAE;
         double *sCc = sX + c * SX;
         double bb[D];
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            double u = 0.0;
#pragma unroll
            for (int dy = 0; dy < D; dy++) { u += byp[Q * dy] * sAc[tx + Q * (dy + D * dz)]; }
            bb[dz] = u;
         }
         double qv[Q];
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            double u = 0.0;
#pragma unroll
            for (int dz = 0; dz < D; dz++) { u += sB[qz + Q * dz] * bb[dz]; }
            qv[qz] = u * dq[qz];
         }
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            double u = 0.0;
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { u += sB[qz + Q * dz] * qv[qz]; }
            sCc[tx + Q * (ty + Q * dz)] = u;
         }
      }
      __syncthreads();
      // backward x: thread (dx = tx < D, qy = ty)
      if (tx < D)
      {
#pragma unroll
         for (int c = 0; c < kVC; c++)
         {
            if (!todo[c]) { continue; }
            const double *sCc = sX + c * SX;
            double *sEc = sA + c * SAE;
#pragma unroll
            for (int dz = 0; dz < D; dz++)
            {
               double u = 0.0;
#pragma unroll
               for (int qx = 0; qx < Q; qx++) { u += brxp[qx] * sCc[qx + Q * (ty + Q * dz)]; }
               sEc[tx + D * (ty + Q * dz)] = u;
            }
         }
      }
      __syncthreads();
      // backward y: thread (dx = tx < D, dy = ty < D)
      if (tx < D && ty < D && active)
      {
#pragma unroll
         for (int c = 0; c < kVC; c++)
         {
            if (!todo[c]) { continue; }
            const double *sEc = sA + c * SAE;
            double *yc = a.YE + (size_t)c * a.ye_stride;
            double dot = 0.0;
#pragma unroll
            for (int dz = 0; dz < D; dz++)
            {
               double u = 0.0;
#pragma unroll
               for (int qy = 0; qy < Q; qy++) { u += bryp[qy] * sEc[tx + D * (qy + Q * dz)]; }
               yc[tx + D * (ty + D * dz) + (size_t)ND * e] = u;
               dot += xcol[c][dz] * u;
            }
            dots[c] += dot;
         }
      }
   }
   double bp[kVC];
   block_sum3(dots[0], dots[1], dots[2], red, bp);
   double total[kVC];
   if (grid_sum3_last_block(bp, a.partials, a.stride, a.ticket, red, total))
   {
      if (tid == 0)
      {
         VcgScalars *s = a.s;
         for (int c = 0; c < kVC; c++)
         {
            if (!todo[c]) { continue; }
            s->den[c] = total[c];
            if (total[c] == 0.0 && !a.multi) { s->done[c] = 1; } // breakdown, as upstream
         }
         s->first = 0;
      }
   }
}

// ---- K1, plane-per-thread form --------------------------------------------------
// The (qx, qy)-column form above is bound by LDS bandwidth: every contraction
// stage re-reads its operands (and the 1-D table) from LDS, 474 doubles per thread
// and batch, 80 % of its run time.  Here a thread owns one x-index of one
// component of one element and keeps the whole (y, z) plane of that index in
// registers: only the two x contractions exchange data through LDS (64 + 96
// doubles read per thread), the y and z contractions and the scaling by the
// quadrature data run on registers with the 1-D table in scalar registers (its
// indices are loop constants there), i.e. as FMAs with an SGPR operand.
// The quadrature data of a batch is staged through LDS (NQ/TE values in flight per
// thread, shared by the three components), which keeps the register count low
// enough for two workgroups per CU.
// Software pipeline: the element->node map runs two batches ahead, the gathers and
// the quadrature data one batch ahead.

template <int D, int Q, int NEB, bool SYM>
__global__ void __launch_bounds__(kVC *Q *NEB, 2)
vcg_apply_plane(const VcgArgs a, const int nbatch)
{
   constexpr int NQ = Q * Q * Q, ND = D * D * D, DD = D * D;
   constexpr int TE = kVC * Q; // threads per element
   constexpr int NT = TE * NEB;
   // LDS strides: consecutive (element, component) groups of a wave are skewed by
   // 16 B modulo 128 B, so the broadcast reads of different groups hit different banks
   constexpr int CS = (ND + 3) & ~1;     // direction d of one component
   constexpr int CE = (DD * Q + 3) & ~1; // x-contracted result of one component, [dy,dz][qx]
   constexpr int PER0 = kVC * (CS + CE);
   constexpr int PER = PER0 + ((6 - PER0 % 16) + 16) % 16;
   constexpr int GPT = (NEB * ND + NT - 1) / NT; // gather items per thread
   constexpr int DPT = (NQ + TE - 1) / TE;       // quadrature values staged per thread
   constexpr int DSTR = (NQ + 7) & ~1;
   __shared__ double smem[NEB * (PER + DSTR)];
   __shared__ double red[48];

   const int tid = threadIdx.x;
   unsigned long long t_start = 0, t_loop = 0;
   unsigned long long c_start = 0;
   if (a.trace) { t_start = wall_clock64(); c_start = clock64(); }
   const int eb = tid / TE, lt = tid - eb * TE;
   const int c = lt / Q, qx = lt - c * Q;
   const int G = gridDim.x;
   double *sIn = smem + eb * PER + c * CS;           // [dx + D*(dy + D*dz)]
   double *sE = smem + eb * PER + kVC * CS + c * CE; // [qx + Q*(dy + D*dz)]
   double *sD = smem + NEB * PER + eb * DSTR;        // [qx + Q*(qy + Q*qz)]

   int mi[GPT];
   auto load_map = [&](const int b) {
      // (no predicates on the loads of the pipeline: items beyond a ragged last batch read a valid entry, their values
      // are never written to LDS - predicated loads cost a branch each and cut the loop into 40 basic blocks that the
      // scheduler cannot interleave with the contractions)
      const int e0 = b * NEB, last = min(NEB, a.NE - e0) * ND - 1;
#pragma unroll
      for (int k = 0; k < GPT; k++)
      {
         const int i = tid + k * NT;
         mi[k] = a.map[(size_t)e0 * ND + min(i, last)];
      }
   };
   int b = xcd_swizzle(blockIdx.x, G);
   if (b < nbatch) { load_map(b); } // in flight while the scalars are read

   if (a.s->all_done) { return; }
   const bool first = a.s->first != 0;
   bool todo[kVC];
   double beta[kVC];
#pragma unroll
   for (int k = 0; k < kVC; k++) { todo[k] = a.s->done[k] == 0; }
   if (a.multi && !first && !vcg_pending_update(a.s, a.iter, blockIdx.x == 0 && tid == 0, todo)) { return; }
#pragma unroll
   for (int k = 0; k < kVC; k++) { beta[k] = (first || !todo[k]) ? 0.0 : a.s->rz[k] / a.s->rz_prev[k]; }
   const bool mine = (c == 0) ? todo[0] : (c == 1) ? todo[1] : todo[2];
   // 1-D table: scalar registers for the register-resident contractions, this
   // thread's row / column for the two x contractions
   // SYM: the table is mirror symmetric, B[q,d] = B[Q-1-q, D-1-d] (Gauss-Lobatto or Bernstein basis at Gauss-Legendre
   // points; checked by lgh_create), and half of it serves all indices - 24 scalar registers less, without which
   // the kernel spills 40 of them to vector lanes and reads them back ~50 times per batch
   constexpr int QD = Q * D, HB = SYM ? (QD + 1) / 2 : QD;
   double Bsr[HB];
#pragma unroll
   for (int i = 0; i < HB; i++) { Bsr[i] = uniform_f64(a.B[i]); }
   auto Bs = [&](const int idx) -> double { return (SYM && idx >= HB) ? Bsr[QD - 1 - idx] : Bsr[idx]; };
   double bx[D], bt[Q];
#pragma unroll
   for (int dx = 0; dx < D; dx++) { bx[dx] = a.B[qx + Q * dx]; }
#pragma unroll
   for (int q = 0; q < Q; q++) { bt[q] = a.B[q + Q * (qx < D ? qx : 0)]; }
   // Everything loaded so far (map, scalars, tables) is complete before the pipelined
   // loop starts: the compiler's wait-count analysis is loop-conservative and otherwise
   // treats these one-time loads as possibly outstanding in EVERY iteration, which with
   // in-order vmcnt put an s_waitcnt vmcnt(6) - i.e. a wait for the next batch's
   // gathers - ahead of the first FMA of each batch.
   __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)

   double gz[kVC][GPT], gd[kVC][GPT], gi[GPT], dq[DPT];
   auto load_gather = [&]() { // nodes of mi[]: residual r, direction d, 1/diag
#pragma unroll
      for (int k = 0; k < GPT; k++) { gi[k] = a.dinv[mi[k]]; }
#pragma unroll
      for (int k2 = 0; k2 < kVC; k2++)
      {
#pragma unroll
         for (int k = 0; k < GPT; k++)
         {
            gz[k2][k] = a.r[(size_t)k2 * a.N + mi[k]];
            gd[k2][k] = first ? 0.0 : a.d[(size_t)k2 * a.N + mi[k]];
         }
      }
   };
   auto load_dq = [&](const int bb) {
      const int e = min(bb * NEB + eb, a.NE - 1);
      const double se = a.Se[e];
#pragma unroll
      for (int k = 0; k < DPT; k++)
      {
         const int j = lt + k * TE;
         dq[k] = a.Dq[(size_t)e * a.dqs + ((DPT * TE == NQ) ? j : min(j, NQ - 1))] * se;
      }
   };

   double dot = 0.0;
   if (b < nbatch)
   {
      load_gather();
      load_dq(b);
      if (b + G < nbatch) { load_map(b + G); }
   }
   for (; b < nbatch; b += G)
   {
      const int e = b * NEB + eb;
      const bool active = (e < a.NE) && mine;
      const int nit = min(NEB, a.NE - b * NEB) * ND;
      __syncthreads(); // previous batch finished with the LDS buffers
      // directions d = z + beta d of every active component (K2 stores the same
      // values), quadrature data
#pragma unroll
      for (int k = 0; k < GPT; k++)
      {
         const int i = tid + k * NT;
         if (i < nit)
         {
            const int el = i / ND, dd = i - el * ND;
#pragma unroll
            for (int k2 = 0; k2 < kVC; k2++)
            {
               smem[el * PER + k2 * CS + dd] = fma(beta[k2], gd[k2][k], __dmul_rn(gz[k2][k], gi[k]));
            }
         }
      }
#pragma unroll
      for (int k = 0; k < DPT; k++)
      {
         const int j = lt + k * TE;
         if ((DPT * TE == NQ) || j < NQ) { sD[j] = dq[k]; }
      }
      // next batch: gathers and quadrature data now, the map of the one after
      const int bn = b + G;
      if (bn < nbatch)
      {
         load_gather();
         load_dq(bn);
         if (bn + G < nbatch) { load_map(bn + G); }
      }
      __syncthreads();
      // forward x: t[dy,dz] = sum_dx B[qx,dx] d[dx,dy,dz]
      double t[DD];
#pragma unroll
      for (int k = 0; k < DD; k++)
      {
         double u = 0.0;
#pragma unroll
         for (int dx = 0; dx < D; dx++) { u = fma(bx[dx], sIn[dx + D * k], u); }
         t[k] = u;
      }
      // forward y: w[qy][dz] = sum_dy B[qy,dy] t[dy,dz]
      double w[Q][D];
#pragma unroll
      for (int qy = 0; qy < Q; qy++)
      {
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            double u = 0.0;
#pragma unroll
            for (int dy = 0; dy < D; dy++) { u = fma(Bs(qy + Q * dy), t[dy + D * dz], u); }
            w[qy][dz] = u;
         }
      }
      // per qy row: forward z, scale by the quadrature data, backward z (in place)
#pragma unroll
      for (int qy = 0; qy < Q; qy++)
      {
         double cz[Q];
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            double u = 0.0;
#pragma unroll
            for (int dz = 0; dz < D; dz++) { u = fma(Bs(qz + Q * dz), w[qy][dz], u); }
            cz[qz] = u * sD[qx + Q * (qy + Q * qz)];
         }
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            double u = 0.0;
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { u = fma(Bs(qz + Q * dz), cz[qz], u); }
            w[qy][dz] = u;
         }
      }
      // backward y: sum_qy B[qy,dy] w[qy][dz]; hand the plane over
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
#pragma unroll
         for (int dy = 0; dy < D; dy++)
         {
            double u = 0.0;
#pragma unroll
            for (int qy = 0; qy < Q; qy++) { u = fma(Bs(qy + Q * dy), w[qy][dz], u); }
            sE[qx + Q * (dy + D * dz)] = u;
         }
      }
      __syncthreads();
      // backward x: thread dx = qx < D sums over the Q planes
      if (qx < D && active)
      {
         double *yc = a.YE + (size_t)c * a.ye_stride + (size_t)ND * e;
#pragma unroll
         for (int k = 0; k < DD; k++)
         {
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < Q; q++) { u = fma(bt[q], sE[q + Q * k], u); }
            yc[qx + D * k] = u;
            dot = fma(sIn[qx + D * k], u, dot);
         }
      }
   }
   if (a.trace) { t_loop = wall_clock64(); }
   double bp[kVC];
   block_sum3(c == 0 ? dot : 0.0, c == 1 ? dot : 0.0, c == 2 ? dot : 0.0, red, bp);
   double total[kVC];
   if (grid_sum3_last_block_flat(bp, a.partials, a.stride, a.ticket, red, total))
   {
      if (tid == 0)
      {
         VcgScalars *s = a.s;
         for (int k = 0; k < kVC; k++)
         {
            if (!todo[k]) { continue; }
            s->den[k] = total[k];
            if (total[k] == 0.0 && !a.multi) { s->done[k] = 1; } // breakdown, as upstream
         }
         s->first = 0;
      }
   }
   if (a.trace && tid == 0)
   {
      unsigned xcc = 0, hwid = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      a.trace[kTraceRec * blockIdx.x + 0] = t_start;
      a.trace[kTraceRec * blockIdx.x + 1] = t_loop;
      a.trace[kTraceRec * blockIdx.x + 2] = wall_clock64();
      a.trace[kTraceRec * blockIdx.x + 3] = clock64() - c_start; // shader cycles between the two wall-clock stamps 0 and 2
   }
}

// ---- K1, plane form for D1D >= 5 (Q4Q3 0x358, Q5Q4 0x36A) ------------------------
// Same thread layout as vcg_apply_plane, sized for the larger planes: at Q1D = 10 the
// (y, z) plane of one x-index is 100 doubles, so HY = 2 threads (adjacent lanes) share
// it, each taking Q/HY of its qy rows; their partial backward-y sums are combined with
// one DPP lane swap per pair of values.  The 1-D table sits in scalar registers in
// half: the Gauss-Lobatto basis at Gauss-Legendre points is mirror symmetric,
// B[q,d] = B[Q-1-q, D-1-d], so Q*D/2 values serve all Q*D indices (60 doubles would
// not fit the scalar file, 30 do).  One batch of NEB elements per workgroup; its
// quadrature data is fetched in one coalesced sweep at the start (when the registers
// are still free) into the LDS region that later takes the x-contracted planes.
__device__ __forceinline__ double lane_pair_swap(const double v)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
   hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
   return __hiloint2double(hi, lo);
}

template <int D, int Q, int HY, int NEB, bool SEP>
__global__ void __launch_bounds__(kVC *Q *HY *NEB, 2)
vcg_apply_plane_ho(const VcgArgs a)
{
   constexpr int NQ = Q * Q * Q, ND = D * D * D, DD = D * D, QD = Q * D, HB = QD / 2;
   constexpr int TE = kVC * Q * HY, NT = TE * NEB, QH = Q / HY;
   static_assert(Q % HY == 0 && (HY == 1 || HY == 2), "qy rows split evenly over one or two lanes");
   static_assert(QD % 2 == 0 && TE % 2 == 0, "mirror-symmetric table, lane pairs inside a wave");
   constexpr int CS = (ND + 3) & ~1;
   constexpr int CE = (DD * Q + 3) & ~1;
   constexpr int PER0 = kVC * (CS + CE);
   constexpr int PER = PER0 + ((6 - PER0 % 16) + 16) % 16;
   constexpr int GPT = (NEB * ND + NT - 1) / NT;
   constexpr int DPT = (NEB * NQ + NT - 1) / NT;
   static_assert(kVC * CE >= NQ, "the quadrature data of an element is staged where its x-contracted planes go later");
   // SEP (compact mass data of a tensor-product rule): value(q, e) = Se[e] w[qx] w[qy] w[qz] with the weights in scalar
   // registers - no sweep over the quadrature data, nothing staged in LDS, one barrier less
   __shared__ double smem[NEB * PER];
   __shared__ double sB[QD];
   __shared__ double red[48];

   const int tid = threadIdx.x;
   const int eb = tid / TE, lt = tid - eb * TE;
   const int c = lt / (Q * HY), r2 = lt - c * (Q * HY);
   const int qx = r2 / HY, h = r2 - qx * HY;
   const int e0 = xcd_swizzle(blockIdx.x, gridDim.x) * NEB, e = e0 + eb;
   const int nel = min(NEB, a.NE - e0);
   double *sIn = smem + eb * PER + c * CS;           // [dx + D*(dy + D*dz)]
   double *sE = smem + eb * PER + kVC * CS + c * CE; // [qx + Q*(dy + D*dz)]
   const double *sD = smem + eb * PER + kVC * CS;    // [qx + Q*(qy + Q*qz)] until the planes are handed over

   int mi[GPT];
#pragma unroll
   for (int k = 0; k < GPT; k++)
   {
      const int i = tid + k * NT;
      mi[k] = (i < nel * ND) ? a.map[(size_t)e0 * ND + i] : -1;
   }
   if (a.s->all_done) { return; }
   // quadrature data of the batch: one coalesced sweep while the registers are still free
   double dst[SEP ? 1 : DPT];
   if (!SEP)
   {
#pragma unroll
      for (int k = 0; k < DPT; k++)
      {
         const int i = tid + k * NT, ei = i / NQ;
         dst[k] = (i < nel * NQ) ? a.Dq[(size_t)(e0 + ei) * a.dqs + (i - ei * NQ)] * a.Se[e0 + ei] : 0.0;
      }
   }
   double ws[SEP ? (Q + 1) / 2 : 1]; // (mirror symmetric, checked by lgh_create: w[q] = w[Q - 1 - q])
   auto wq = [&](const int q) -> double { return ws[SEP ? (q < (Q + 1) / 2 ? q : Q - 1 - q) : 0]; };
   double wxe = 0.0;
   if (SEP)
   {
#pragma unroll
      for (int q = 0; q < (Q + 1) / 2; q++) { ws[q] = uniform_f64(a.w1[q]); }
      wxe = a.w1[qx] * a.Se[min(e, a.NE - 1)];
   }
   const bool first = a.s->first != 0;
   bool todo[kVC];
   double beta[kVC];
#pragma unroll
   for (int k = 0; k < kVC; k++) { todo[k] = a.s->done[k] == 0; }
   if (a.multi && !first && !vcg_pending_update(a.s, a.iter, blockIdx.x == 0 && tid == 0, todo)) { return; }
#pragma unroll
   for (int k = 0; k < kVC; k++) { beta[k] = (first || !todo[k]) ? 0.0 : a.s->rz[k] / a.s->rz_prev[k]; }
   const bool active = (e < a.NE) && ((c == 0) ? todo[0] : (c == 1) ? todo[1] : todo[2]);
   double Bs[HB];
#pragma unroll
   for (int i = 0; i < HB; i++) { Bs[i] = uniform_f64(a.B[i]); }
   auto Bc = [&](const int q, const int d) -> double {
      const int idx = q + Q * d;
      return idx < HB ? Bs[idx] : Bs[QD - 1 - idx];
   };
   for (int i = tid; i < QD; i += NT) { sB[i] = a.B[i]; }
   if (!SEP)
   {
#pragma unroll
      for (int k = 0; k < DPT; k++)
      {
         const int i = tid + k * NT;
         if (i < NEB * NQ)
         {
            const int el = i / NQ, j = i - el * NQ;
            smem[el * PER + kVC * CS + j] = dst[k];
         }
      }
   }
   // directions d = z + beta d of the three components (K2 stores the same values)
#pragma unroll
   for (int k = 0; k < GPT; k++)
   {
      const int i = tid + k * NT;
      if (i < nel * ND)
      {
         const int el = i / ND, dd = i - el * ND;
         const int n = mi[k];
         const double gi = a.dinv[n];
#pragma unroll
         for (int k2 = 0; k2 < kVC; k2++)
         {
            const double gz = a.r[(size_t)k2 * a.N + n];
            const double gd = first ? 0.0 : a.d[(size_t)k2 * a.N + n];
            smem[el * PER + k2 * CS + dd] = fma(beta[k2], gd, __dmul_rn(gz, gi));
         }
      }
   }
   __syncthreads();
   // forward x: t[dy,dz] = sum_dx B[qx,dx] d[dx,dy,dz] (each lane of a pair forms it)
   double t[DD];
   {
      double bx[D];
#pragma unroll
      for (int dx = 0; dx < D; dx++) { bx[dx] = sB[qx + Q * dx]; }
#pragma unroll
      for (int k = 0; k < DD; k++)
      {
         double u = 0.0;
#pragma unroll
         for (int dx = 0; dx < D; dx++) { u = fma(bx[dx], sIn[dx + D * k], u); }
         t[k] = u;
      }
   }
   // this lane's qy rows: forward y, then per row forward z, scaling, backward z; backward y accumulated
   double acc[DD];
#pragma unroll
   for (int k = 0; k < DD; k++) { acc[k] = 0.0; }
#pragma unroll
   for (int r = 0; r < QH; r++)
   {
      double by[D]; // table row of qy = h*QH + r
#pragma unroll
      for (int dy = 0; dy < D; dy++)
      {
         double v = Bc(r, dy);
         if (HY == 2) { v = h ? Bc(QH + r, dy) : v; }
         by[dy] = v;
      }
      double wr[D];
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
         double u = 0.0;
#pragma unroll
         for (int dy = 0; dy < D; dy++) { u = fma(by[dy], t[dy + D * dz], u); }
         wr[dz] = u;
      }
      double wrow = 0.0; // SEP: s_e w[qx] w[qy] of this row
      if (SEP)
      {
         double wy = wq(r);
         if (HY == 2) { wy = h ? wq(QH + r) : wy; }
         wrow = wxe * wy;
      }
      double cz[Q];
#pragma unroll
      for (int qz = 0; qz < Q; qz++)
      {
         double u = 0.0;
#pragma unroll
         for (int dz = 0; dz < D; dz++) { u = fma(Bc(qz, dz), wr[dz], u); }
         cz[qz] = SEP ? (u * wrow) * wq(qz) : u * sD[qx + Q * (h * QH + r) + Q * Q * qz];
      }
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
         double u = 0.0;
#pragma unroll
         for (int qz = 0; qz < Q; qz++) { u = fma(Bc(qz, dz), cz[qz], u); }
#pragma unroll
         for (int dy = 0; dy < D; dy++) { acc[dy + D * dz] = fma(by[dy], u, acc[dy + D * dz]); }
      }
   }
   if (!SEP) { __syncthreads(); } // every lane of the element has read its quadrature data
   // hand the plane over: lane h of a pair delivers the (dy,dz) entries k = h (mod 2)
   if (HY == 1)
   {
#pragma unroll
      for (int k = 0; k < DD; k++) { sE[qx + Q * k] = acc[k]; }
   }
   else
   {
#pragma unroll
      for (int k = 0; k + 1 < DD; k += 2)
      {
         const double give = h ? acc[k] : acc[k + 1];
         const double keep = h ? acc[k + 1] : acc[k];
         sE[qx + Q * (k + h)] = keep + lane_pair_swap(give);
      }
      if (DD & 1)
      {
         const double tot = acc[DD - 1] + lane_pair_swap(acc[DD - 1]);
         if (h == 0) { sE[qx + Q * (DD - 1)] = tot; }
      }
   }
   __syncthreads();
   // backward x: lane (dx = qx < D, h) sums the Q planes for its share of the (dy,dz) pairs
   double dot = 0.0;
   if (qx < D && active)
   {
      double bt[Q];
#pragma unroll
      for (int q = 0; q < Q; q++) { bt[q] = sB[q + Q * qx]; }
      double *yc = a.YE + (size_t)c * a.ye_stride + (size_t)ND * e;
      constexpr int KH = (DD + HY - 1) / HY;
#pragma unroll
      for (int kk = 0; kk < KH; kk++)
      {
         const int k = h * KH + kk;
         if (k < DD)
         {
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < Q; q++) { u = fma(bt[q], sE[q + Q * k], u); }
            yc[qx + D * k] = u;
            dot = fma(sIn[qx + D * k], u, dot);
         }
      }
   }
   double bp[kVC];
   block_sum3(c == 0 ? dot : 0.0, c == 1 ? dot : 0.0, c == 2 ? dot : 0.0, red, bp);
   double total[kVC];
   if (grid_sum3_last_block(bp, a.partials, a.stride, a.ticket, red, total))
   {
      if (tid == 0)
      {
         VcgScalars *s = a.s;
         for (int k = 0; k < kVC; k++)
         {
            if (!todo[k]) { continue; }
            s->den[k] = total[k];
            if (total[k] == 0.0 && !a.multi) { s->done[k] = 1; } // breakdown, as upstream
         }
         s->first = 0;
      }
   }
}

// ---- K1, Kronecker form, any order (compact mass data on a tensor-product rule; a.M1 = 1-D mass tile B^T diag(w) B):
// A_e = s_e M1 (x) M1 (x) M1 - three contractions of D on the D^3 dofs of an (element, component) item instead of the
// six of D x Q and the pass over the quadrature points: 3 D^4 FMAs per item (D = 6: 3 888 against ~21 000), no
// quadrature-point values; what is left is the traffic: the gathers of r, d_old, 1/diag and the E-vector out.  The
// Q3Q2 meshes of 20 000 zones and more take the slab form of the same operator (lgh_vcg_slab.hip), everything else
// this kernel.  One workgroup of 256 threads = NEB consecutive elements x 3 components; a thread owns one row of D
// values per stage (x rows (item, dz, dy), y rows (item, dz, dx), z rows (item, dy, dx)); (d, A d) = sum over the item's
// dofs of d y.  Same operator in exact arithmetic as vcg_apply_plane(_ho); rounding differs (tests/test_gpu_k1.py).
template <int D, int NEB>
__global__ void __launch_bounds__(256)
vcg_apply_kron(const VcgArgs a)
{
   constexpr int ND = D * D * D, DD = D * D, NT = 256, NI = NEB * kVC;
   __shared__ double sIn[NI * ND], sT1[NI * ND], sT2[NI * ND];
   __shared__ double red[48];
   const int tid = threadIdx.x;
   const int e0 = xcd_swizzle(blockIdx.x, gridDim.x) * NEB;
   const int nel = min(NEB, a.NE - e0);
   constexpr int GPT = (NEB * ND + NT - 1) / NT;
   int mi[GPT];
#pragma unroll
   for (int k = 0; k < GPT; k++)
   {
      const int i = tid + k * NT;
      mi[k] = (i < nel * ND) ? a.map[(size_t)e0 * ND + i] : -1;
   }
   if (a.s->all_done) { return; }
   const bool first = a.s->first != 0;
   bool todo[kVC];
   double beta[kVC];
#pragma unroll
   for (int k = 0; k < kVC; k++) { todo[k] = a.s->done[k] == 0; }
   if (a.multi && !first && !vcg_pending_update(a.s, a.iter, blockIdx.x == 0 && tid == 0, todo)) { return; }
#pragma unroll
   for (int k = 0; k < kVC; k++) { beta[k] = (first || !todo[k]) ? 0.0 : a.s->rz[k] / a.s->rz_prev[k]; }
   double M[DD]; // M[i + D j], symmetric, in scalar registers
#pragma unroll
   for (int i = 0; i < DD; i++) { M[i] = uniform_f64(a.M1[i]); }
   // directions d = z + beta d of the three components (K2 stores the same values); item = k2 * NEB + el
#pragma unroll
   for (int k = 0; k < GPT; k++)
   {
      const int i = tid + k * NT;
      if (i < nel * ND)
      {
         const int el = i / ND, dd = i - el * ND;
         const int n = mi[k];
         const double gi = a.dinv[n];
#pragma unroll
         for (int k2 = 0; k2 < kVC; k2++)
         {
            const double gz = a.r[(size_t)k2 * a.N + n];
            const double gd = first ? 0.0 : a.d[(size_t)k2 * a.N + n];
            sIn[(k2 * NEB + el) * ND + dd] = fma(beta[k2], gd, __dmul_rn(gz, gi));
         }
      }
   }
   __syncthreads();
   auto live = [&](const int item) -> bool {
      const int k2 = item / NEB, el = item - k2 * NEB;
      return el < nel && ((k2 == 0) ? todo[0] : (k2 == 1) ? todo[1] : todo[2]);
   };
   auto row = [&](const double (&u)[D], double (&o)[D]) {
#pragma unroll
      for (int i = 0; i < D; i++)
      {
         double t = M[i] * u[0];
#pragma unroll
         for (int j = 1; j < D; j++) { t = fma(M[i + D * j], u[j], t); }
         o[i] = t;
      }
   };
   // x: rows of D consecutive values
   for (int r = tid; r < NI * DD; r += NT)
   {
      if (!live(r / DD)) { continue; }
      double u[D], o[D];
#pragma unroll
      for (int j = 0; j < D; j++) { u[j] = sIn[r * D + j]; }
      row(u, o);
#pragma unroll
      for (int i = 0; i < D; i++) { sT1[r * D + i] = o[i]; }
   }
   __syncthreads();
   // y: rows (item, dz, dx), stride D
   for (int r = tid; r < NI * DD; r += NT)
   {
      if (!live(r / DD)) { continue; }
      const int dx = r % D, iz = r / D; // iz = dz + D item
      const int base = iz * DD + dx;
      double u[D], o[D];
#pragma unroll
      for (int j = 0; j < D; j++) { u[j] = sT1[base + D * j]; }
      row(u, o);
#pragma unroll
      for (int i = 0; i < D; i++) { sT2[base + D * i] = o[i]; }
   }
   __syncthreads();
   // z: rows (item, dy, dx), stride D^2; the element factor; E-vector out and the partials of (d, A d)
   double dot[kVC] = {0.0, 0.0, 0.0};
   for (int r = tid; r < NI * DD; r += NT)
   {
      const int item = r / DD;
      if (!live(item)) { continue; }
      const int yx = r - item * DD;
      const int k2 = item / NEB, el = item - k2 * NEB;
      const int base = item * ND + yx;
      const double se = a.Se[e0 + el];
      double u[D], o[D];
#pragma unroll
      for (int j = 0; j < D; j++) { u[j] = sT2[base + DD * j]; }
      row(u, o);
      double *yc = a.YE + (size_t)k2 * a.ye_stride + (size_t)(e0 + el) * ND + yx;
      double part = 0.0;
#pragma unroll
      for (int i = 0; i < D; i++)
      {
         const double v = o[i] * se;
         yc[DD * i] = v;
         part = fma(sIn[base + DD * i], v, part);
      }
      dot[0] += (k2 == 0) ? part : 0.0;
      dot[1] += (k2 == 1) ? part : 0.0;
      dot[2] += (k2 == 2) ? part : 0.0;
   }
   double bp[kVC];
   block_sum3(dot[0], dot[1], dot[2], red, bp);
   double total[kVC];
   if (grid_sum3_last_block(bp, a.partials, a.stride, a.ticket, red, total))
   {
      if (tid == 0)
      {
         VcgScalars *s = a.s;
         for (int k = 0; k < kVC; k++)
         {
            if (!todo[k]) { continue; }
            s->den[k] = total[k];
            if (total[k] == 0.0 && !a.multi) { s->done[k] = 1; } // breakdown, as upstream
         }
         s->first = 0;
      }
   }
}

// base pointer + 32-bit byte offset: one SGPR pair and one VGPR per address
__device__ __forceinline__ double vcg_ld(const double *base, const unsigned off) { return *(const double *)((const char *)base + off); }
__device__ __forceinline__ double *vcg_ptr(double *base, const unsigned off) { return (double *)((char *)base + off); }

// a.reset: the first kernel of a solve does what vcg_set_tol_k did in a launch of its own in front of it (round 6: one
// kernel boundary and a serial one-thread loop over ~400 words less per solve).  Workgroup 0 clears the exact accumulators
// and the set counters (nobody touches them before K1 of the first iteration); the thread that writes the scalars of the
// solve resets them first.
__device__ __forceinline__ void vcg_reset_accumulators(const VcgArgs &a)
{
   if (!a.reset || blockIdx.x != 0) { return; }
   if (a.limbs) { for (int i = threadIdx.x; i < 2 * kLimbWords + 8 * 16; i += blockDim.x) { a.limbs[i] = 0; } }
   if (a.rzl) { for (int i = threadIdx.x; i < 3 * kLimbWords; i += blockDim.x) { a.rzl[i] = 0; } }
}
__device__ __forceinline__ void vcg_reset_scalars(const VcgArgs &a, VcgScalars *s)
{
   if (!a.reset) { return; }
   s->rel_tol2 = a.reset_tol2;
   s->all_done = 0;
   s->first = 1;
   for (int c = 0; c < kVC; c++) { s->done[c] = 0; s->iters[c] = 0; s->nupd[c] = 0; s->alpha_last[c] = 0.0; s->alpha_hist[0][c] = s->alpha_hist[1][c] = 0.0; s->rzh[0][c] = s->rzh[1][c] = 0.0; }
}

// ---- init: r = b (x = 0), z = r/diag, nom_c = (z_c, r_c)
__global__ void __launch_bounds__(256)
vcg_init_k(const VcgArgs a)
{
   vcg_reset_accumulators(a);
   __shared__ double red[48];
   const int n = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; // as K2 and vcg_init_force_k
   double part[kVC] = {0.0, 0.0, 0.0};
   if (n < a.N)
   {
      const double di = a.dinv[n];
      const double ow = a.owner ? a.owner[n] : 1.0;
#pragma unroll
      for (int c = 0; c < kVC; c++)
      {
         const size_t i = (size_t)c * a.N + n;
         const double rv = a.b[(size_t)c * a.N + (a.ncaller ? a.ncaller[n] : n)]; // (b is the caller's vector)
         a.r[i] = rv;
         part[c] = ow * __dmul_rn(rv, di) * rv; // z = r/diag is recomputed where it is used
      }
   }
   double bp[kVC], total[kVC];
   block_sum3(part[0], part[1], part[2], red, bp);
   if (grid_sum3_last_block(bp, a.partials, a.stride, a.ticket, red, total))
   {
      if (threadIdx.x == 0)
      {
         VcgScalars *s = a.s;
         int all = 1;
         vcg_reset_scalars(a, s);
         for (int c = 0; c < kVC; c++)
         {
            s->rz[c] = s->rz_prev[c] = total[c];
            s->iters[c] = 0;
            if (!a.multi)
            {
               s->r0[c] = fmax(total[c] * s->rel_tol2, 0.0);
               s->done[c] = (total[c] < 0.0 || total[c] <= s->r0[c]) ? 1 : 0;
               all = all && s->done[c];
            }
         }
         s->first = 1;
         if (!a.multi) { s->all_done = all; }
      }
   }
}
// ---- init fused with the tail of ForcePA->Mult(one) (SolveVelocity, laghos_solver.cpp:354-384):
// b = -(H1R^T Y_E) with the essential rows of every component zeroed, x = 0, r = b, nom.  FE is
// the force E-vector in the reference layout (D1D^3, dim, NE) (laghos_assembly.cpp:312); the sum
// runs in the order of h1_transpose_gather_k, so b has the bits of the separate
// gather / negate / EliminateRHS kernels it replaces (four passes over the vectors less).
// ellc: the transpose of the restriction for the FORCE E-vector (slot j of node n at j * N + n), which is in the caller's zone
// order whatever order the solve's own E-vector has
template <int DEG>
__global__ void __launch_bounds__(256)
vcg_init_force_k(const VcgArgs a, const double *__restrict__ FE, const int ND, double *__restrict__ bout, const int *__restrict__ ellc, const int degc)
{
   vcg_reset_accumulators(a);
   __shared__ double red[48];
   const int n = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; // node ranges of vcg_init_k: same partial sums
   double part[kVC] = {0.0, 0.0, 0.0};
   if (n < a.N)
   {
      long pos[DEG];
#pragma unroll
      for (int j = 0; j < DEG; j++)
      {
         const int p = (j < degc) ? ellc[(size_t)j * a.N + n] : -1; // e*ND + d
         const int e = p / ND;
         pos[j] = (p >= 0) ? (long)kVC * ND * e + (p - e * ND) : -1;
      }
      const double di = a.dinv[n];
#pragma unroll
      for (int c = 0; c < kVC; c++)
      {
         double s = 0.0;
#pragma unroll
         for (int j = 0; j < DEG; j++) { if (pos[j] >= 0) { s += FE[pos[j] + (long)ND * c]; } }
         double bv = -s;
         if (a.ess[c] && a.ess[c][n]) { bv = 0.0; }
         const size_t i = (size_t)c * a.N + n;
         bout[(size_t)c * a.N + (a.ncaller ? a.ncaller[n] : n)] = bv; // (the caller's vector)
         a.r[i] = bv;
         a.x[i] = 0.0;
         part[c] = __dmul_rn(bv, di) * bv;
      }
   }
   double bp[kVC], total[kVC];
   block_sum3(part[0], part[1], part[2], red, bp);
   if (grid_sum3_last_block(bp, a.partials, a.stride, a.ticket, red, total))
   {
      if (threadIdx.x == 0)
      {
         VcgScalars *s = a.s;
         int all = 1;
         vcg_reset_scalars(a, s);
         for (int c = 0; c < kVC; c++)
         {
            s->rz[c] = s->rz_prev[c] = total[c];
            s->iters[c] = 0;
            s->r0[c] = fmax(total[c] * s->rel_tol2, 0.0);
            s->done[c] = (total[c] < 0.0 || total[c] <= s->r0[c]) ? 1 : 0;
            all = all && s->done[c];
         }
         s->first = 1;
         s->all_done = all;
      }
   }
}
// The same kernel over a table of byte offsets into the force E-vector ([e][c][d]; absent contributions point at
// a zero element behind it - element NE, all components): no index arithmetic, no predicates, and slots no node of the wavefront uses are not
// fetched (as in vcg_update_p_k).  Same node -> workgroup map and summation orders as above: the same bits.
__global__ void __launch_bounds__(256)
vcg_init_force_z_k(const VcgArgs a, const double *__restrict__ FE, const unsigned compb_fe, const unsigned *__restrict__ ellf,
                   double *__restrict__ bout)
{
   vcg_reset_accumulators(a);
   __shared__ double red[48];
   const int n = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
   const bool ok = n < a.N;
   const unsigned nn = (unsigned)(ok ? n : a.N - 1);
   const unsigned rowb = 4u * (unsigned)a.N;
   unsigned ix[8];
#pragma unroll
   for (int j = 0; j < 8; j++) { ix[j] = *(const unsigned *)((const char *)ellf + (4u * nn + (unsigned)j * rowb)); }
   const double di = vcg_ld(a.dinv, 8u * nn);
   const unsigned es = a.essbits[nn];
   const unsigned zoff = 8u * (unsigned)kVC * (unsigned)(a.ye_stride - kYePad); // = 8 * kVC * ND * NE
   double fe[kVC][8];
#pragma unroll
   for (int j = 0; j < 8; j++)
   {
      if (j == 0 || __any(ix[j] != zoff))
      {
#pragma unroll
         for (int c = 0; c < kVC; c++) { fe[c][j] = vcg_ld(FE, ix[j] + (unsigned)c * compb_fe); }
      }
      else
      {
#pragma unroll
         for (int c = 0; c < kVC; c++) { fe[c][j] = 0.0; }
      }
   }
   double part[kVC];
#pragma unroll
   for (int c = 0; c < kVC; c++)
   {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 8; j++) { s += fe[c][j]; }
      double bv = -s;
      if ((es >> c) & 1u) { bv = 0.0; }
      part[c] = ok ? __dmul_rn(bv, di) * bv : 0.0;
      if (ok)
      {
         const size_t i = (size_t)c * a.N + n;
         bout[(size_t)c * a.N + (a.ncaller ? a.ncaller[n] : n)] = bv; // (the caller's vector)
         a.r[i] = bv;
         a.x[i] = 0.0;
      }
   }
   double bp[kVC], total[kVC];
   block_sum3(part[0], part[1], part[2], red, bp);
   if (grid_sum3_last_block(bp, a.partials, a.stride, a.ticket, red, total))
   {
      if (threadIdx.x == 0)
      {
         VcgScalars *s = a.s;
         int all = 1;
         vcg_reset_scalars(a, s);
         for (int c = 0; c < kVC; c++)
         {
            s->rz[c] = s->rz_prev[c] = total[c];
            s->iters[c] = 0;
            s->r0[c] = fmax(total[c] * s->rel_tol2, 0.0);
            s->done[c] = (total[c] < 0.0 || total[c] <= s->r0[c]) ? 1 : 0;
            all = all && s->done[c];
         }
         s->first = 1;
         s->all_done = all;
      }
   }
}
__global__ void __launch_bounds__(256)
vcg_ellf_k(const int *__restrict__ ell, unsigned *__restrict__ ellf, const size_t n_have, const size_t n_all, const int ND, const int NE)
{
   const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_all) { return; }
   const int p = (i < n_have) ? ell[i] : -1; // e*ND + d
   const int e = p / ND;
   ellf[i] = 8u * (unsigned)(p < 0 ? kVC * ND * NE : kVC * ND * e + (p - e * ND));
}
__global__ void __launch_bounds__(256)
vcg_mapb_k(const int *__restrict__ map, unsigned *__restrict__ mapb, const size_t n)
{
   const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { mapb[i] = 8u * (unsigned)map[i]; }
}
// flag = 1 unless the D nodes of every x-row of every element are consecutive node numbers
__global__ void __launch_bounds__(256)
vcg_map_xrows_k(const int *__restrict__ map, const size_t nrows, const int D, int *flag)
{
   const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= nrows) { return; }
   const int *p = map + i * D;
   bool ok = true;
   for (int k = 1; k < D; k++) { ok = ok && p[k] == p[0] + k; }
   if (!ok) { *flag = 1; }
}
__global__ void vcg_init_finish_k(VcgScalars *s)
{
   int all = 1;
   for (int c = 0; c < kVC; c++)
   {
      s->rz_prev[c] = s->rz[c];
      s->r0[c] = fmax(s->rz[c] * s->rel_tol2, 0.0);
      s->done[c] = (s->rz[c] < 0.0 || s->rz[c] <= s->r0[c]) ? 1 : 0;
      all = all && s->done[c];
   }
   s->all_done = all;
}
__global__ void vcg_set_tol_k(VcgScalars *s, double rel_tol2, long long *limbs, long long *rzl)
{
   if (limbs) { for (int i = 0; i < 2 * kLimbWords + 8 * 16; i++) { limbs[i] = 0; } } // accumulators and the set counters behind them
   if (rzl) { for (int i = 0; i < 3 * kLimbWords; i++) { rzl[i] = 0; } }
   s->rel_tol2 = rel_tol2;
   s->all_done = 0;
   s->first = 1;
   for (int c = 0; c < kVC; c++) { s->done[c] = 0; s->iters[c] = 0; s->nupd[c] = 0; s->alpha_last[c] = 0.0; s->alpha_hist[0][c] = s->alpha_hist[1][c] = 0.0; s->rzh[0][c] = s->rzh[1][c] = 0.0; }
}
// Several ranks: the host enqueues the iteration count of the previous solve without looking at the flags,
// so a few launches may follow convergence.  Their kernels return at once, but the exchanges between them
// still run and would re-sum the (already global) den / rz of a finished component once per surplus
// iteration - growing by the number of ranks each time, to inf after a long solve.  Whoever marks a
// component done (vcg_pending_update / vcg_pending_den inside the kernels, this kernel before the host
// looks) therefore zeroes its scalars: sums of zeros stay zero.  den[c] and rz[c] are undefined (0) once
// done[c] is set; nothing reads them after that.
__global__ void vcg_update_finish_k(VcgScalars *s, int iter, VcgScalars *host_out, unsigned long long *host_token, const unsigned long long token)
{
   int all = 1;
   for (int c = 0; c < kVC; c++)
   {
      if (!s->done[c])
      {
         s->iters[c] = iter;
         if (s->rz[c] < 0.0 || s->rz[c] <= s->r0[c]) { s->done[c] = 1; }
      }
      if (s->done[c]) { s->rz[c] = 0.0; s->den[c] = 0.0; }
      all = all && s->done[c];
   }
   s->all_done = all;
   if (host_out) // (the host looks next: straight into its pinned memory, the token last)
   {
      *host_out = *s;
      __threadfence_system();
      __hip_atomic_store(host_token, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   }
}

// rz_limbs mode: the outcome of the last enqueued iteration `last`, committed for the host (what workgroup 0 of
// K1(last + 1) would write); host_out: pinned host memory the host reads after its stream synchronisation (nullptr: test hook)
__global__ void vcg_rz_finish_k(VcgScalars *s, const long long *rzl, const int last, const long long *peers, const int n_peers, VcgScalars *host_out,
                                unsigned long long *host_token, const unsigned long long token)
{
   // several ranks: own words + the peers' (exchange_words) = the sum over the ranks; folded into a scratch set - the
   // flag word rides along (any rank's overflow turns the sum into NaN on every rank)
   long long set[kLimbWords]; // (a copy: K1(last + 1), if the host enqueues it after its look, folds own + peers itself)
   for (int i = 0; i <= kLimbShards * kVC * kLimbs; i++)
   {
      long long w = rzl[(last % 3) * kLimbWords + i];
      for (int p = 0; p < n_peers; p++) { w += peers[p * kLimbWords + i]; }
      set[i] = w;
   }
   double cur[kVC];
   bool dn[kVC];
   for (int k = 0; k < kVC; k++)
   {
      const double before = (last > 1) ? s->rzh[(last - 1) & 1][k] : s->rz[k];
      cur[k] = exact_fold(set, k, exact_scale(before));
      dn[k] = s->done[k] != 0 || vcg_rz_converged(last + 1, cur[k], s->r0[k]);
   }
   vcg_rz_commit(s, last + 1, cur, dn);
   if (host_out)
   {
      *host_out = *s;
      __threadfence_system();
      __hip_atomic_store(host_token, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
   }
}

// ---- K2: per node and component: A d = sum of element contributions, ess rows,
// d = r/diag + beta d, x += alpha d, r -= alpha A d, (r, r/diag).  The preconditioned
// residual z = r/diag is never stored: K1 and K2 recompute it from r (one rounded
// multiply, so both see the same value), which saves a vector read and a write.
template <bool FUSED_GATHER, int DEG>
__global__ void __launch_bounds__(256)
vcg_update_k(const VcgArgs a)
{
   __shared__ double red[48];
   if (a.s->all_done) { return; }
   const bool it1 = (a.iter == 1);
   const int n = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
   const bool ok = n < a.N;
   const int nn = ok ? n : a.N - 1;
   double part[kVC] = {0.0, 0.0, 0.0};
   int pidx[DEG > 0 ? DEG : 1];
   if (FUSED_GATHER)
   {
#pragma unroll
      for (int j = 0; j < DEG; j++) { pidx[j] = (j < a.deg) ? a.ell[(size_t)j * a.N + nn] : -1; }
   }
   const double di = a.dinv[nn];
   const double ow = a.owner ? a.owner[nn] : 1.0;
   const bool shared = FUSED_GATHER && a.hmask && a.hmask[nn];
#pragma unroll 1
   for (int c = 0; c < kVC; c++)
   {
      if (a.s->done[c]) { continue; } // uniform
      if (a.multi && vcg_pending_den(a.s, c, blockIdx.x == 0 && threadIdx.x == 0)) { continue; }
      const double alpha = a.s->rz[c] / a.s->den[c];
      const double beta = it1 ? 0.0 : a.s->rz[c] / a.s->rz_prev[c];
      const size_t i = (size_t)c * a.N + nn;
      double zv;
      if (FUSED_GATHER)
      {
         if (shared) { zv = a.yL[i]; } // summed over the ranks by halo_sum
         else
         {
            const double *yc = a.YE + (size_t)c * a.ye_stride;
            zv = 0.0;
#pragma unroll
            for (int j = 0; j < DEG; j++) { if (pidx[j] >= 0) { zv += yc[pidx[j]]; } }
         }
      }
      else { zv = a.yL[i]; }
      if (a.ess[c] && a.ess[c][nn]) { zv = 0.0; }
      const double ro = a.r[i];
      double dv = __dmul_rn(ro, di); // z of the previous iterate, not stored
      if (!it1) { dv = fma(beta, a.d[i], dv); }
      const double xv = a.x[i] + alpha * dv;
      const double rv = ro - alpha * zv;
      const double pz = __dmul_rn(rv, di);
      if (ok)
      {
         a.d[i] = dv;
         a.x[i] = xv;
         a.r[i] = rv;
         part[c] = ow * rv * pz;
      }
   }
   double bp[kVC], total[kVC];
   block_sum3(part[0], part[1], part[2], red, bp);
   if (grid_sum3_last_block(bp, a.partials, a.stride, a.ticket, red, total))
   {
      if (threadIdx.x == 0)
      {
         VcgScalars *s = a.s;
         int all = 1;
         for (int c = 0; c < kVC; c++)
         {
            if (!s->done[c] && !(a.multi && s->den[c] == 0.0)) // (several ranks: breakdown found by this launch)
            {
               s->rz_prev[c] = s->rz[c];
               s->rz[c] = total[c]; // betanom
               if (!a.multi)
               {
                  s->iters[c] = a.iter;
                  if (total[c] < 0.0 || total[c] <= s->r0[c]) { s->done[c] = 1; }
               }
            }
            all = all && s->done[c];
         }
         if (!a.multi) { s->all_done = all; }
      }
   }
}

// ---- K2, bounded-grid form (one rank) ---------------------------------------------------------------
// The same node update as vcg_update_k, organised the way the node phase of the persistent kernel
// (round 2's persistent-solve experiment, profiles/r2_pcg_trace_*.txt) turned out to run fastest - its measurements carry over to a kernel of its own:
//  * four workgroups of 512 threads per CU (more on large meshes: ranges of at most ~1000 nodes), each with a contiguous node range of equal COST (a node costs a
//    fixed part plus a part per element contribution; with equal counts the ranges that cover
//    element-boundary planes take 40 % longer), all ~35 loads of a node issued before the first use,
//    straight-line code.  One node per thread and pass (U = 1: 106 VGPRs, two workgroups resident per CU)
//    beats two (U = 2: 182 VGPRs, one resident; what the persistent kernel's node phase used): 55.4 -> 49.7 us
//    at C2 (LGH_K2_U=2, LGH_K2_GRID=<workgroups per CU> for A/B);
//  * the ELL row holds byte offsets and absent contributions point at a zero slot behind the Y_E plane: no
//    predicated loads, addresses are one SGPR base + a 32-bit VGPR offset;
//  * x is only needed at the end of the solve: it is updated every second iteration with both terms,
//      x = (x + alpha_{it-1} d_{it-1}) + alpha_it d_it     (the roundings of two single updates),
//    and vcg_xfix_k adds the pending term of a component that stopped after an odd number of updates -
//    22 MB less traffic per iteration on average at C2;
//  * 1024 workgroup partials instead of 3566 for the ticketed reduction;
//  * write-through and non-temporal stores of d, r, x were tried (no gain) and are gone.

// MINW (round 5): wavefronts per SIMD the register allocation admits (the second argument of HIP's __launch_bounds__):
// 4 = the budget of rounds 2-4 (up to 128 VGPRs: two workgroups of eight wavefronts per CU), 6 = at most 80 (three).  The kernel is a chain of dependent memory round trips per pass
// (table -> element contributions -> stores), and what hides them is the number of wavefronts a CU holds; the second
// table (slots 4..7) is summed in a branch of its own so that its 24 registers exist only there.
template <bool XU, int U, int MINW = 4>
__global__ void __launch_bounds__(512, MINW)
vcg_update_p_k(const VcgArgs a)
{
   constexpr int NT = 512;
   __shared__ double red[48];
   if (a.s->all_done)
   {
      // (several ranks, pack_rz: the exchange that follows must not carry the sums of an earlier iteration)
      if (a.pack_rz && blockIdx.x == 0 && threadIdx.x < a.hp.n_nbr * kVC) { a.hp.sbuf[(size_t)a.hp.base[threadIdx.x / kVC] + threadIdx.x % kVC] = 0.0; }
      return;
   }
   const int it = a.iter;
   const bool first = (it == 1);
   const int tid = threadIdx.x;
   const int w = xcd_swizzle(blockIdx.x, gridDim.x);
   const int n0 = a.nstart[w], n1 = a.nstart[w + 1];
   bool todo[kVC];
   double alpha[kVC], alpha_prev[kVC], beta[kVC], den[kVC];
   // rz_limbs mode (lgh_vcg.hpp): (r, z) of the last two iterations as workgroup 0 of K1 left them
   double rzc[kVC], rzp[kVC];
   int rzE[kVC] = {0, 0, 0};
#pragma unroll
   for (int k = 0; k < kVC; k++)
   {
      rzc[k] = (a.rzl && it > 1) ? a.s->rzh[(it - 1) & 1][k] : a.s->rz[k];
      rzp[k] = a.rzl ? a.s->rzh[it & 1][k] : a.s->rz_prev[k]; // (only looked at from the second iteration on)
      rzE[k] = exact_scale(rzc[k]);
   }
#pragma unroll
   for (int k = 0; k < kVC; k++) { den[k] = (a.den_limbs == 1) ? exact_den(a.limbs + (it & 1) * kLimbWords, k, rzc[k]) : a.s->den[k]; }
   if (a.den_limbs == 1 && blockIdx.x == 0 && tid < kLimbWords) { a.limbs[((it + 1) & 1) * kLimbWords + tid] = 0; } // the set of the next K1
#pragma unroll
   for (int k = 0; k < kVC; k++)
   {
      todo[k] = a.s->done[k] == 0;
      if (a.den_limbs == 1 && den[k] == 0.0) { todo[k] = false; } // breakdown, as upstream (marked below)
      // (several ranks: breakdown is looked at here, after the sum of (d, A d) over the ranks - vcg_pending_den)
      if (a.multi && todo[k] && vcg_pending_den(a.s, k, blockIdx.x == 0 && tid == 0)) { todo[k] = false; }
      alpha[k] = todo[k] ? rzc[k] / den[k] : 0.0;
      alpha_prev[k] = todo[k] ? (a.rzl ? a.s->alpha_hist[(it - 1) & 1][k] : a.s->alpha_last[k]) : 0.0;
      beta[k] = (first || !todo[k]) ? 0.0 : rzc[k] / rzp[k];
   }
   if (a.rzl && !(todo[0] || todo[1] || todo[2]))
   {
      // nothing iterates any more (launches enqueued past convergence): only the bookkeeping of a breakdown is left
      if (blockIdx.x == 0 && tid == 0)
      {
         for (int k = 0; k < kVC; k++) { if (!a.s->done[k] && den[k] == 0.0) { a.s->done[k] = 1; } }
      }
      return;
   }
   const bool xload = XU && it > 2;
   const unsigned compb = 8u * (unsigned)a.N;
   const unsigned zoff = 8u * (unsigned)(a.ye_stride - kYePad); // byte offset of the zero slot (make_ellz)
   const char *const ellhi = (const char *)a.ellz + (size_t)16 * (size_t)a.N; // slots 4..7
   double part[kVC] = {0.0, 0.0, 0.0};
   for (int base = n0; base < n1; base += NT * U)
   {
      unsigned nn[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; u++)
      {
         const int n = base + tid + u * NT;
         ok[u] = n < n1;
         nn[u] = (unsigned)(ok[u] ? n : n0);
      }
      // ELL rows as two tables of four offsets per node (16 bytes each): slots 0..3 with one load per node; slots 4..7 -
      // only a vertex node of the mesh has more than four contributions, with the merged E-vector of the slab K1 only
      // where a set boundary meets an element edge in y and z - are fetched when some node of the wavefront uses them
      // (rows hold their contributions first, so slot 3 tells): a ninth of the wavefronts, 16 of the table's 32 bytes per
      // node for the others
      unsigned ix[U][4];
      bool hi_any = false;
#pragma unroll
      for (int u = 0; u < U; u++)
      {
         const v4u_ q0 = *(const v4u_ *)((const char *)a.ellz + 16u * nn[u]);
         ix[u][0] = q0[0]; ix[u][1] = q0[1]; ix[u][2] = q0[2]; ix[u][3] = q0[3];
         hi_any = hi_any || (q0[3] != zoff);
      }
      const bool hi = !a.k2_skip || __any(hi_any);
      double di[U], ro[U][kVC], dol[U][kVC], xo[U][kVC];
      unsigned es[U];
#pragma unroll
      for (int u = 0; u < U; u++)
      {
         di[u] = vcg_ld(a.dinv, 8u * nn[u]);
         es[u] = a.essbits[nn[u]];
#pragma unroll
         for (int k = 0; k < kVC; k++)
         {
            const unsigned vb = 8u * nn[u] + (unsigned)k * compb;
            ro[u][k] = vcg_ld(a.r, vb);
            dol[u][k] = vcg_ld(a.d, vb);
            xo[u][k] = 0.0;
            if (XU && MINW < 6) { xo[u][k] = vcg_ld(a.x, vb); }
         }
      }
      // Slot j of the ELL rows is only fetched when some node of the wavefront has a j-th contribution (rows hold
      // their contributions first): a wave of consecutive nodes of a tensor-product mesh rarely needs all 8 (a
      // node needs 1, 2, 4 or 8), and the loads of absent slots - all lanes at the zero slot - still occupy the
      // address path.
      bool need[4];
      need[0] = true;
#pragma unroll
      for (int j = 1; j < 4; j++)
      {
         bool any = false;
#pragma unroll
         for (int u = 0; u < U; u++) { any = any || (ix[u][j] != zoff); }
         need[j] = !a.k2_skip || __any(any);
      }
      double zsum[U][kVC]; // ascending contribution order (absent slots add 0.0): the sum of vcg_update_k
      {
         double ye[U][kVC][4];
#pragma unroll
         for (int j = 0; j < 4; j++)
         {
            if (need[j])
            {
#pragma unroll
               for (int k = 0; k < kVC; k++)
               {
                  const double *yc = a.YE + (size_t)k * a.ye_stride;
#pragma unroll
                  for (int u = 0; u < U; u++) { ye[u][k][j] = vcg_ld(yc, ix[u][j]); }
               }
            }
            else
            {
#pragma unroll
               for (int k = 0; k < kVC; k++)
               {
#pragma unroll
                  for (int u = 0; u < U; u++) { ye[u][k][j] = 0.0; }
               }
            }
         }
#pragma unroll
         for (int u = 0; u < U; u++)
         {
#pragma unroll
            for (int k = 0; k < kVC; k++) { zsum[u][k] = ((0.0 + ye[u][k][0]) + ye[u][k][1]) + ye[u][k][2]; zsum[u][k] += ye[u][k][3]; }
         }
      }
      if (XU && MINW >= 6 && xload)
      {
         // (six wavefronts per SIMD: x is asked for once the element contributions have been summed - its three registers
         //  per node would otherwise be live beside their 24, and the budget of 80 does not hold both; the other
         //  wavefronts of the SIMD cover the round trip)
#pragma unroll
         for (int u = 0; u < U; u++)
         {
#pragma unroll
            for (int k = 0; k < kVC; k++) { asm volatile("" : "+v"(zsum[u][k])); xo[u][k] = vcg_ld(a.x, 8u * nn[u] + (unsigned)k * compb); }
         }
      }
      if (hi)
      {
         // contributions 5..8 (a ninth of the wavefronts): second table, then the values, added behind the first four
#pragma unroll
         for (int u = 0; u < U; u++)
         {
            const v4u_ q1 = *(const v4u_ *)(ellhi + 16u * nn[u]);
#pragma unroll
            for (int k = 0; k < kVC; k++)
            {
               const double *yc = a.YE + (size_t)k * a.ye_stride;
               const double y4 = vcg_ld(yc, q1[0]), y5 = vcg_ld(yc, q1[1]), y6 = vcg_ld(yc, q1[2]), y7 = vcg_ld(yc, q1[3]);
               zsum[u][k] = (((zsum[u][k] + y4) + y5) + y6) + y7;
            }
         }
      }
      // several ranks: a node shared with another rank takes its A d from the L-vector the halo exchange has summed
      double ysh[U][kVC];
      bool anysh = false;
#pragma unroll
      for (int u = 0; u < U; u++) { anysh = anysh || ((es[u] >> 3) & 1u); }
      anysh = a.multi && __any(anysh);
#pragma unroll
      for (int u = 0; u < U; u++)
      {
#pragma unroll
         for (int k = 0; k < kVC; k++) { ysh[u][k] = anysh ? vcg_ld(a.yL, 8u * nn[u] + (unsigned)k * compb) : 0.0; }
      }
#pragma unroll
      for (int u = 0; u < U; u++)
      {
         const double ow = ((es[u] >> 4) & 1u) ? 0.0 : 1.0; // (1 on one rank)
#pragma unroll
         for (int k = 0; k < kVC; k++)
         {
            const unsigned vb = 8u * nn[u] + (unsigned)k * compb;
            double zs = zsum[u][k];
            if (anysh && ((es[u] >> 3) & 1u)) { zs = ysh[u][k]; }
            const double z_ = ((es[u] >> k) & 1u) ? 0.0 : zs;
            const double zold = __dmul_rn(ro[u][k], di[u]); // z of the previous iterate, not stored
            const double dnew = first ? zold : fma(beta[k], dol[u][k], zold);
            const double rnew = ro[u][k] - alpha[k] * z_;
            if (ok[u] && todo[k])
            {
               *vcg_ptr(a.d, vb) = dnew;
               *vcg_ptr(a.r, vb) = rnew;
               if (XU)
               {
                  const double x0 = xload ? xo[u][k] : 0.0;
                  *vcg_ptr(a.x, vb) = fma(alpha[k], dnew, fma(alpha_prev[k], first ? 0.0 : dol[u][k], x0));
               }
               part[k] += (a.multi ? ow : 1.0) * rnew * __dmul_rn(rnew, di[u]);
            }
         }
      }
   }
   if (a.rzl)
   {
      // no last workgroup: the share of (r, z) of this workgroup into set it % 3 (exact limbs, fire-and-forget atomics);
      // the next kernels fold it.  Workgroup 0 leaves what K2 of the next iteration and the host read.
      // (the workgroup's share as ONE double per component - a fixed tree: wave sums, then the eight wavefronts in
      //  order - split into limbs by one thread per component: exactness is only needed ACROSS workgroups, whose
      //  atomics arrive in any order.  Round 4, first version: every thread split its own partial and the limbs went
      //  through twelve 64-bit wave sums - 0.5 us more per launch, profiles/r4_k2_tail.txt.)
      const int lane = tid & 63, wid = tid >> 6;
      const double w0 = wave_sum(part[0], lane, 64), w1 = wave_sum(part[1], lane, 64), w2 = wave_sum(part[2], lane, 64);
      if (lane == 0) { red[3 * wid + 0] = w0; red[3 * wid + 1] = w1; red[3 * wid + 2] = w2; }
      __syncthreads();
      long long *L = a.rzl + (it % 3) * kLimbWords;
      if (tid < kVC)
      {
         double t = 0.0;
#pragma unroll
         for (int w = 0; w < NT / 64; w++) { t += red[3 * w + tid]; }
         const bool live = (tid == 0) ? todo[0] : (tid == 1) ? todo[1] : todo[2];
         const int E = (tid == 0) ? rzE[0] : (tid == 1) ? rzE[1] : rzE[2];
         long long acc[kLimbs] = {0, 0, 0, 0};
         const bool bad = live && !exact_add(acc, t, E);
         if (live && !bad)
         {
#pragma unroll
            for (int j = 0; j < kLimbs; j++)
            {
               if (acc[j] != 0) { (void)__hip_atomic_fetch_add(&L[(blockIdx.x % kLimbShards) * (kVC * kLimbs) + kLimbs * tid + j], acc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            }
         }
         if (bad) { (void)__hip_atomic_fetch_or(&L[kLimbShards * kVC * kLimbs], 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
      if (blockIdx.x == 0 && tid == 0)
      {
         VcgScalars *s = a.s;
         s->first = 0;
         for (int k = 0; k < kVC; k++)
         {
            if (s->done[k]) { continue; }
            s->den[k] = den[k];
            if (den[k] == 0.0) { s->done[k] = 1; continue; } // breakdown, as upstream
            s->alpha_hist[it & 1][k] = alpha[k]; // (K2 of the next iteration: the deferred update of x)
            s->alpha_last[k] = alpha[k];         // (vcg_xfix_k after the solve)
            s->nupd[k] = it;
         }
      }
      return;
   }
   double bp[kVC], total[kVC];
   block_sum3(part[0], part[1], part[2], red, bp);
   if (grid_sum3_last_block(bp, a.partials, a.stride, a.ticket, red, total))
   {
      if (tid == 0)
      {
         VcgScalars *s = a.s;
         int all = 1;
         if (a.den_limbs == 1) { s->first = 0; }
         for (int k = 0; k < kVC; k++)
         {
            if (a.den_limbs == 1 && !s->done[k])
            {
               s->den[k] = den[k];
               if (den[k] == 0.0) { s->done[k] = 1; }
            }
            if (!s->done[k] && !(a.multi && s->den[k] == 0.0)) // (several ranks: breakdown found by this launch)
            {
               s->alpha_last[k] = s->rz[k] / den[k]; // the alpha this launch used
               s->nupd[k] = it;
               s->rz_prev[k] = s->rz[k];
               s->rz[k] = total[k]; // betanom: on several ranks the local part, summed and looked at by the next K1
               if (!a.multi)
               {
                  s->iters[k] = it;
                  if (total[k] < 0.0 || total[k] <= s->r0[k]) { s->done[k] = 1; }
               }
            }
            all = all && s->done[k];
         }
         if (!a.multi) { s->all_done = all; }
         if (a.pack_rz)
         {
            // the local (r, z) straight into the send buffer of the exchange that sums it over the ranks
            for (int k = 0; k < a.hp.n_nbr; k++)
            {
               for (int e = 0; e < kVC; e++) { a.hp.sbuf[(size_t)a.hp.base[k] + e] = s->rz[e]; }
            }
         }
      }
   }
}
// components whose last update of x is still pending (an odd number of updates)
__global__ void __launch_bounds__(256)
vcg_xfix_k(const VcgArgs a)
{
   const int n = blockIdx.x * blockDim.x + threadIdx.x;
   if (n >= a.N) { return; }
   for (int k = 0; k < kVC; k++)
   {
      if ((a.s->nupd[k] & 1) == 0) { continue; }
      const size_t i = (size_t)k * a.N + n;
      a.x[i] = fma(a.s->alpha_last[k], a.d[i], a.x[i]);
   }
}

// The solve ran in the library's own node numbering (a.ncaller): its solution goes to the caller's vector, with the pending
// update of vcg_xfix_k applied on the way (the same fma: the same bits as fix, then copy)
__global__ void __launch_bounds__(256)
vcg_xout_k(const VcgArgs a, double *__restrict__ X)
{
   const int n = blockIdx.x * blockDim.x + threadIdx.x;
   if (n >= a.N) { return; }
   const int nc = a.ncaller[n];
   for (int k = 0; k < kVC; k++)
   {
      const size_t i = (size_t)k * a.N + n;
      double xv = a.x[i];
      if (a.s->nupd[k] & 1) { xv = fma(a.s->alpha_last[k], a.d[i], xv); }
      X[(size_t)k * a.N + nc] = xv;
   }
}

// unfused E -> L sum of the kVC planes (multi-GPU path, unusual valence)
__global__ void __launch_bounds__(256)
vcg_gather_k(const VcgArgs a)
{
   const int n = blockIdx.x * blockDim.x + threadIdx.x;
   if (n >= a.N) { return; }
   for (int c = 0; c < kVC; c++)
   {
      if (a.s->done[c]) { continue; }
      const double *yc = a.YE + (size_t)c * a.ye_stride;
      double s = 0.0;
      for (int j = 0; j < a.deg; j++)
      {
         const int p = a.ell[(size_t)j * a.N + n];
         if (p >= 0) { s += yc[p]; }
      }
      a.yL[(size_t)c * a.N + n] = s;
   }
}

// Several ranks, slab-form K1 with exact accumulators (den_limbs == 2): the local part of (d, A d) has to exist as a
// double before the exchange that sums it over the ranks - one workgroup (of the kernel that runs between K1 and the
// exchange anyway) folds the set K1 added into and clears the other one for the next K1.  256 threads.
__device__ __forceinline__ void vcg_fold_den(const VcgArgs &a)
{
   const int tid = threadIdx.x, it = a.iter;
   // (the scale of the accumulators is that of (r, z) before this iteration: in rz_limbs mode K1 of this iteration left it in rzh)
   if (tid < kVC && !a.s->done[tid])
   {
      const double ref = (a.rzl && it > 1) ? a.s->rzh[(it - 1) & 1][tid] : a.s->rz[tid];
      __hip_atomic_store(&a.s->den[tid], exact_den(a.limbs + (it & 1) * kLimbWords, tid, ref), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   if (tid < kLimbWords) { a.limbs[((it + 1) & 1) * kLimbWords + tid] = 0; }
   if (tid == 0) { a.s->first = 0; }
}
__global__ void __launch_bounds__(256)
vcg_fold_den_k(const VcgArgs a) { vcg_fold_den(a); }

// E -> L sum at the listed (rank-shared) nodes only: what the halo exchange needs
// pack_halo: every sum also goes to its places in the send buffer (one per neighbour that shares the node: the CSR the
// combine kernel reads), and workgroup 0 puts the local (d, A d) behind every neighbour's block (nx_den) - the exchange
// then starts without the pack kernel (halo_sum(..., packed)).  Components that are done send zeros.
__global__ void __launch_bounds__(256)
vcg_gather_list_k(const VcgArgs a)
{
   if (blockIdx.x == 0 && (a.den_limbs == 2 || a.nx_den > 0))
   {
      if (a.den_limbs == 2) { vcg_fold_den(a); }
      __syncthreads(); // (den of this workgroup's own fold: visible to its other threads)
      const int t = threadIdx.x;
      if (t < a.hp.n_nbr * a.nx_den)
      {
         const int k = t / a.nx_den, e = t - k * a.nx_den;
         a.hp.sbuf[(size_t)a.hp.base[k] + (size_t)kVC * a.hp.ncnt[k] + e] = __hip_atomic_load(e < kVC ? &a.s->den[e] : &a.s->den_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
   }
   const int u = blockIdx.x * blockDim.x + threadIdx.x;
   if (u >= a.n_shared) { return; }
   const int n = a.sh_node[u];
   for (int c = 0; c < kVC; c++)
   {
      const bool live = a.s->done[c] == 0;
      double s = 0.0;
      if (live)
      {
         const double *yc = a.YE + (size_t)c * a.ye_stride;
         for (int j = 0; j < a.deg; j++)
         {
            const int p = a.ell[(size_t)j * a.N + n];
            if (p >= 0) { s += yc[p]; }
         }
         a.yL[(size_t)c * a.N + n] = s;
      }
      if (a.pack_halo)
      {
         for (int k = a.hp.sh_off[u]; k < a.hp.sh_off[u + 1]; k++)
         {
            const int j = a.hp.sh_src[k];
            if (j >= 0) { a.hp.sbuf[(size_t)a.hp.pos[j] + (size_t)c * a.hp.cnt[j]] = s; }
         }
      }
   }
}

// ---- tables of the node kernel K2 ---------------------------------------------------------
__global__ void __launch_bounds__(256)
vcg_essbits_k(const uint8_t *e0, const uint8_t *e1, const uint8_t *e2, const uint8_t *hmask, const double *owner, uint8_t *bits,
              const int N)
{
   const int n = blockIdx.x * blockDim.x + threadIdx.x;
   if (n >= N) { return; }
   // bits 0-2: essential for component k; several ranks: bit 3 = shared with another rank (its A d comes halo-summed
   // from the L-vector), bit 4 = owned by another rank (weight 0 in the dot products)
   bits[n] = (uint8_t)(((e0 && e0[n]) ? 1 : 0) | ((e1 && e1[n]) ? 2 : 0) | ((e2 && e2[n]) ? 4 : 0) | ((hmask && hmask[n]) ? 8 : 0) |
                       ((owner && owner[n] == 0.0) ? 16 : 0));
}
__global__ void __launch_bounds__(256)
vcg_ellz_k(const int *__restrict__ ell, unsigned *__restrict__ ellz, const size_t n_have, const size_t n_all, const int zslot)
{
   const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_all) { return; }
   const int p = (i < n_have) ? ell[i] : -1; // rows beyond the mesh's valence: absent
   // in: slot-major (slot j of node n at j * N + n); out: two tables of four slots per node, [n][j] for j < 4 and, behind
   // it, [n][j - 4] for the rest (vcg_update_p_k reads a table entry as one 16-byte load)
   const size_t N = n_all / 8, j = i / N, n = i - j * N;
   ellz[(j < 4 ? 0 : 4 * N) + 4 * n + (j & 3)] = 8u * (unsigned)(p < 0 ? zslot : p);
}

// Node phase: a node costs a fixed part (its vectors, the ELL row) plus a part per element contribution
// (measured: t = 4.2..5.3 ns + 1.7..1.8 ns * valence per node and workgroup, the traces of the persistent-solve experiment, profiles/r2_pcg_trace_*.txt); with equal
// node counts the workgroups whose range covers element-boundary planes take 40 % longer and every barrier
// waits for them.  Ranges of equal cost instead, boundaries rounded to 16 nodes (128 B).
int partition_nodes_by_cost(lgh_ctx *c, const int W, int **out, const std::vector<int> *valence)
{
   const int N = c->N;
   std::vector<int> off((size_t)N + 1);
   if (valence) // (the merged layout of the slab K1: contributions per node as THAT layout has them)
   {
      off[0] = 0;
      for (int n = 0; n < N; n++) { off[(size_t)n + 1] = off[n] + (*valence)[n]; }
   }
   else { LGH_HIP_CHECK(hipMemcpy(off.data(), c->t_off, off.size() * sizeof(int), hipMemcpyDeviceToHost)); }
   const char *wenv = getenv("LGH_K2_NODE_WEIGHT"); // fixed part in units of half a contribution; <0: equal counts
   const long fixed = wenv ? atol(wenv) : 5;
   std::vector<long> cum((size_t)N + 1, 0);
   for (int n = 0; n < N; n++) { cum[(size_t)n + 1] = cum[n] + (fixed < 0 ? 1 : fixed + 2 * (long)(off[(size_t)n + 1] - off[n])); }
   std::vector<int> ns((size_t)std::max(W, 1) + 1, N);
   ns[0] = 0;
   int n = 0;
   for (int w = 1; w < W; w++)
   {
      const long target = cum[N] * w / W;
      while (n < N && cum[n] < target) { n++; }
      int nb = (n + 8) & ~15; // nearest multiple of 16
      nb = std::max(nb, ns[(size_t)w - 1]);
      ns[w] = std::min(nb, N);
   }
   ns[std::max(W, 1)] = N;
   LGH_HIP_CHECK(hipMalloc((void **)out, ns.size() * sizeof(int)));
   LGH_HIP_CHECK(hipMemcpy(*out, ns.data(), ns.size() * sizeof(int), hipMemcpyHostToDevice));
   return LGH_OK;
}
// ELL transpose of the restriction as byte offsets into a Y_E plane of NE*ND + pad doubles whose slot NE*ND is
// zero: absent contributions point there, so the gather needs no predicate
int make_ellz(lgh_ctx *c, unsigned **out, const int *ell, const int deg)
{
   const size_t ell_n = (size_t)8 * c->N;
   LGH_HIP_CHECK(hipMalloc((void **)out, ell_n * sizeof(unsigned)));
   hipLaunchKernelGGL(vcg_ellz_k, dim3((unsigned)((ell_n + 255) / 256)), dim3(256), 0, nullptr, ell ? ell : c->t_ell, *out,
                      (size_t)(ell ? deg : c->t_deg) * c->N, ell_n, c->NE * c->ND);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

// ---- merged E-vector layout of the slab-form K1 (round 5) ---------------------------------------------------------
// K1 hands K2 the element contributions A_e d_e; in the element-local layout [e][dz][dy][dx] every node of an x-face
// between two zones travels twice (64 values per zone for 27 nodes) and a wavefront of K2 - consecutive nodes of an
// x-line - uses one y-row (32 bytes) of every 128-byte line it touches.  The slab K1 works on sets of five consecutive
// zones; where those are x-neighbours ("x-chain": zone i + 1's dx = 0 nodes ARE zone i's dx = 3 nodes, read off the
// element -> node map, nothing about the mesh is assumed) the set is stored as 16 rows (dz, dy) of the set's 16 x-nodes
// with the four shared faces summed by K1: 256 instead of 320 doubles per component, every row one cache line.  Sets
// that are not chains (a row of zones ends inside them; the short last set; any unstructured numbering) keep the
// element-local layout.  The plane of a component is the concatenation of the sets' slices; everything K2 and the
// several-rank gathers know about the layout is the ELL table built here.
//   pos[e * ND + d]: where the entry lives in a plane;  sec[...] = 1: the right-hand member of a merged pair
//   settab[s]: (offset of the set's slice / 64) | chain << 31
struct SlabLayout
{
   std::vector<unsigned> settab;
   std::vector<int> pos;
   std::vector<uint8_t> sec;
   size_t n_merged = 0;
};
// (the map of the order the solve runs in: the caller's, or the library's own - vcg_view_map)
static int vcg_view_map(lgh_ctx *c, std::vector<int> &map);
static int slab_merge_layout(lgh_ctx *c, SlabLayout &L)
{
   constexpr int ES = 5, D = 4, ND = 64;
   if (c->D1D != D || c->dim != 3) { set_error("slab_merge_layout: Q3 only"); return LGH_ERR_UNSUPPORTED; }
   const int NE = c->NE, nset = ceil_div(NE, ES);
   std::vector<int> map;
   {
      const int rc = vcg_view_map(c, map);
      if (rc) { return rc; }
   }
   L.settab.assign((size_t)nset, 0u);
   L.pos.assign((size_t)NE * ND, 0);
   L.sec.assign((size_t)NE * ND, 0);
   L.n_merged = 0;
   size_t off = 0; // doubles
   for (int s = 0; s < nset; s++)
   {
      const int e0 = ES * s, nel = std::min(ES, NE - e0);
      bool chain = (nel == ES);
      for (int i = 0; chain && i + 1 < ES; i++)
      {
         const int *ma = &map[(size_t)(e0 + i) * ND], *mb = &map[(size_t)(e0 + i + 1) * ND];
         for (int k = 0; chain && k < D * D; k++) { chain = (ma[D * k + (D - 1)] == mb[D * k]); }
      }
      if (off % 64 != 0 || off / 64 > 0x7fffffffull) { set_error("slab_merge_layout: plane offset out of range"); return LGH_ERR_UNSUPPORTED; }
      L.settab[s] = (unsigned)(off / 64) | (chain ? 0x80000000u : 0u);
      for (int i = 0; i < nel; i++)
      {
         for (int d = 0; d < ND; d++)
         {
            const size_t idx = (size_t)(e0 + i) * ND + d;
            if (!chain) { L.pos[idx] = (int)(off + (size_t)i * ND + d); continue; }
            const int dx = d % D, row = d / D; // row = dy + 4 dz
            L.pos[idx] = (int)(off + (size_t)row * 16 + 3 * i + dx);
            if (i > 0 && dx == 0) { L.sec[idx] = 1; L.n_merged++; }
         }
      }
      off += chain ? 256 : (size_t)nel * ND;
   }
   return LGH_OK;
}
// bit k of entry n: node n is essential for component k
int make_essbits(lgh_ctx *c, uint8_t **out)
{
   LGH_HIP_CHECK(hipMalloc((void **)out, (size_t)c->N));
   const uint8_t *hmask = nullptr;
   const int *sh_node = nullptr;
   int n_shared = 0;
   if (c->multi) { comm_shared_nodes(c, &hmask, &sh_node, &n_shared); }
   hipLaunchKernelGGL(vcg_essbits_k, dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, nullptr, c->essmask[0], c->essmask[1],
                      c->essmask[2], hmask, c->multi ? c->owner : nullptr, *out, c->N);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}


// ---- host side -----------------------------------------------------------------------
struct VcgAux
{
   double *ye = nullptr;     // kVC planes of NE*ND + kYePad doubles (slot NE*ND of each plane stays 0.0)
   unsigned *ellz = nullptr;
   uint8_t *essbits = nullptr;
   int *nstart = nullptr;    // cost-balanced node ranges of the grid2 workgroups of vcg_update_p_k
   unsigned *ellf = nullptr; // ELL transpose as byte offsets into a force E-vector ([e][c][d] + zero slot): vcg_init_force_z_k
   unsigned *mapb = nullptr; // element -> node map as byte offsets into a node vector: vcg_apply_slab346
   int map_xrows = 0;        // the nodes of every x-row of every element are consecutive (vcg_apply_slab346 then loads rows, not nodes)
   long long *limbs = nullptr; // exact accumulators of (d, A d), 2 parities (lgh_vcg.hpp)
   long long *rzl = nullptr;   // exact accumulators of (r, z), 3 sets (rz_limbs mode)
   int grid2 = 0;
   // merged E-vector layout of the slab K1 (slab_merge_layout): its set table and the tables of K2 for that layout
   int rz_words_all = -1;      // several ranks: every rank can exchange (r, z) as accumulator words (-1: not asked yet)
   int rz_words_key = -1;      // ... and the local inputs of that question at the time it was asked
   int ls_all = 0;             // ... every rank could also run the energy CG in lockstep with this solve (asked in the same exchange)
   int ls_ready = 0;           // the last solve exchanged (r, z) as accumulator words and packed its halo itself (or had no neighbour), here
                               // and on every other rank: the energy CG may run in lockstep with the next one (vcg_lockstep_ready)
   unsigned *settab = nullptr;
   int *ellm = nullptr;
   int degm = 0;
   unsigned *ellzm = nullptr;
   int *nstartm = nullptr;
   // The library's own order (lgh_order.hip), when it differs from the caller's: the tables and vectors of the solve in the
   // internal zone order / node numbering - everything the kernels see comes from here then, the context's own tables (the
   // caller's numbering) serve every other entry point.  ord == nullptr: the caller's order is the solve's.
   const MeshOrder *ord = nullptr;
   int *o_map = nullptr;       // NE*ND: internal node of (internal zone, local dof)
   int *o_ell = nullptr;       // t_deg*N: transpose of o_map, E-positions of the solve's own E-vector (internal zone order)
   int *o_ellc = nullptr;      // t_deg*N: per internal node, the positions in an E-vector of the CALLER's zone order (the force E-vector)
   std::vector<int> o_valence; // contributions per internal node
   uint8_t *o_ess[kVC] = {nullptr, nullptr, nullptr};
   double *o_dinv = nullptr, *o_Se = nullptr, *o_massD = nullptr;
   double *o_x = nullptr;      // kVC*N: the solution in internal numbering (vcg_xout_k hands it to the caller)
   // several ranks: owner weights, shared-node flags and the node lists of the exchanges (lgh_comm.hip) in internal numbering
   double *o_owner = nullptr;
   uint8_t *o_hmask = nullptr;
   int *o_nodes = nullptr, *o_shnode = nullptr;
   HaloNodeAlias o_alias = {nullptr, nullptr};
   unsigned long o_mass_gen = ~0ul; // c->mass_gen the copies above were taken at
};
static int vcg_view_map(lgh_ctx *c, std::vector<int> &map)
{
   map.resize((size_t)c->NE * c->ND);
   const VcgAux *x = (const VcgAux *)c->vcg_aux;
   LGH_HIP_CHECK(hipMemcpy(map.data(), (x && x->o_map) ? x->o_map : c->h1map, map.size() * sizeof(int), hipMemcpyDeviceToHost));
   return LGH_OK;
}
void vcg_free(lgh_ctx *c)
{
   VcgAux *x = (VcgAux *)c->vcg_aux;
   if (!x) { return; }
   (void)hipFree(x->settab);
   (void)hipFree(x->ellm);
   (void)hipFree(x->ellzm);
   (void)hipFree(x->nstartm);
   (void)hipFree(x->ye);
   (void)hipFree(x->ellz);
   (void)hipFree(x->essbits);
   (void)hipFree(x->nstart);
   (void)hipFree(x->ellf);
   (void)hipFree(x->mapb);
   (void)hipFree(x->limbs);
   (void)hipFree(x->rzl);
   (void)hipFree(x->o_map);
   (void)hipFree(x->o_ell);
   (void)hipFree(x->o_ellc);
   for (int k = 0; k < kVC; k++) { (void)hipFree(x->o_ess[k]); }
   (void)hipFree(x->o_dinv);
   (void)hipFree(x->o_Se);
   (void)hipFree(x->o_massD);
   (void)hipFree(x->o_x);
   (void)hipFree(x->o_owner);
   (void)hipFree(x->o_hmask);
   (void)hipFree(x->o_nodes);
   (void)hipFree(x->o_shnode);
   delete x;
   c->vcg_aux = nullptr;
}

static bool vcg_supported(const lgh_ctx *c)
{
   if (c->dim != 3) { return false; }
   switch (c->kid)
   {
      case 0x322: case 0x334: case 0x346: case 0x358: case 0x36A: return true;
   }
   return false;
}

template <int D, int Q> static void launch_vcg_plane(lgh_ctx *c, const VcgArgs &a)
{
   // 160 KB of LDS per CU: with the staged quadrature data one element less per
   // workgroup keeps two workgroups resident for Q = 6
   constexpr int NEB0 = (256 / (kVC * Q)) > 0 ? (256 / (kVC * Q)) : 1;
   constexpr int NEB = (Q == 6) ? NEB0 - 1 : NEB0;
   const int nbatch = ceil_div(c->NE, NEB);
   if (c->vcg_grid <= 0)
   {
      int per_cu = 0, ncu = 256;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) { ncu = prop.multiProcessorCount; }
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vcg_apply_plane<D, Q, NEB, true>, kVC * Q * NEB, 0) != hipSuccess || per_cu <= 0)
      {
         per_cu = 2;
      }
      c->vcg_grid = per_cu * ncu;
   }
   const int grid = std::min(nbatch, c->vcg_grid);
   if (c->b_h1_sym) { hipLaunchKernelGGL((vcg_apply_plane<D, Q, NEB, true>), dim3(grid), dim3(kVC * Q * NEB), 0, c->stream, a, nbatch); }
   else { hipLaunchKernelGGL((vcg_apply_plane<D, Q, NEB, false>), dim3(grid), dim3(kVC * Q * NEB), 0, c->stream, a, nbatch); }
}

template <int D, int Q, int HY, int NEB> static void launch_vcg_plane_ho(lgh_ctx *c, const VcgArgs &a)
{
   if (a.w1) { hipLaunchKernelGGL((vcg_apply_plane_ho<D, Q, HY, NEB, true>), dim3(ceil_div(c->NE, NEB)), dim3(kVC * Q * HY * NEB), 0, c->stream, a); }
   else { hipLaunchKernelGGL((vcg_apply_plane_ho<D, Q, HY, NEB, false>), dim3(ceil_div(c->NE, NEB)), dim3(kVC * Q * HY * NEB), 0, c->stream, a); }
}

template <int D, int Q> static void launch_vcg_apply(lgh_ctx *c, const VcgArgs &a)
{
   constexpr int NEB = (256 / (Q * Q)) > 0 ? (256 / (Q * Q)) : 1;
   const int nbatch = ceil_div(c->NE, NEB);
   // one resident wave of workgroups (occupancy query is cached per context)
   if (c->vcg_grid <= 0)
   {
      int per_cu = 0, ncu = 256;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) { ncu = prop.multiProcessorCount; }
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vcg_apply_3d<D, Q, NEB>, Q * Q * NEB, 0) != hipSuccess || per_cu <= 0)
      {
         per_cu = 2;
      }
      c->vcg_grid = per_cu * ncu;
   }
   const int grid = std::min(nbatch, c->vcg_grid);
   hipLaunchKernelGGL((vcg_apply_3d<D, Q, NEB>), dim3(grid), dim3(Q * Q * NEB), 0, c->stream, a, nbatch);
}

template <int D, int NEB> static void launch_vcg_kron(lgh_ctx *c, const VcgArgs &a)
{
   hipLaunchKernelGGL((vcg_apply_kron<D, NEB>), dim3(ceil_div(c->NE, NEB)), dim3(256), 0, c->stream, a);
}

// LGH_VCG_VARIANT: 0 = (qx,qy)-column K1, otherwise the plane-per-thread K1
#define VCG_DISPATCH(D_, Q_)                                                       \
   do {                                                                            \
      if (c->vcg_variant == 0) { launch_vcg_apply<D_, Q_>(c, a); }                 \
      else { launch_vcg_plane<D_, Q_>(c, a); }                                     \
   } while (0)

bool vcg_available(const lgh_ctx *c) { return vcg_supported(c); }
// the form of K1 that vcg_solve launches (kept in one place: the dispatch below asks the same questions)
int vcg_k1_form(lgh_ctx *c)
{
   if (!vcg_supported(c)) { return -1; }
   if (c->kid == 0x346 && c->vcg_variant == 4 && vcg_slab_available(c)) { return 4; }
   // Default at Q3Q2 from kSlabMinElements zones per rank: 40.6 vs 48.5 us per launch at 32^3 zones, 316 vs 383 at 64^3
   // (profiles/README.md); smaller meshes do not fill its one workgroup per CU.
   if (c->kid == 0x346 && c->vcg_variant < 0 && c->NE >= kSlabMinElements && vcg_slab_available(c)) { return 4; }
   // default at D1D >= 5 when the mass data is compact on a tensor-product rule: the Kronecker form (vcg_apply_kron;
   // config 5: 438 against 585 us).  At Q3Q2 the plane form stays the default below the slab threshold (32^3 zones:
   // plane 49.9, generic Kronecker 55.3 us - its LDS stages and one-node gathers cost more than the FMAs it saves,
   // profiles/r4_k1_forms.txt); LGH_VCG_VARIANT=5 selects it at any order (tests)
   if (((c->vcg_variant < 0 && c->D1D >= 5) || c->vcg_variant == 5) && c->M1h && c->w1d)
   {
      const double *Dq, *Se;
      int dqs = 1;
      if (mass_data(c, &Dq, &dqs, &Se) == LGH_OK && dqs == 0) { return 5; }
   }
   if (c->vcg_variant == 0) { return 0; }
   if ((c->kid == 0x358 || c->kid == 0x36A) && !c->b_h1_sym) { return 0; }
   return 2;
}
bool vcg_fused_init_ok(const lgh_ctx *c)
{
   const char *env = getenv("LGH_FUSED_INIT"); // A/B switch
   if (env && env[0] == '0') { return false; }
   return vcg_supported(c) && c->multi == 0 && c->t_deg <= 8;
}

// B, X: dim*N (byNODES).  X must be zero on entry (dv = 0, laghos_solver.cpp:338, :382).
// force_E != nullptr: B and X are outputs - the init kernel forms B = -(H1R^T force_E) with the
// essential rows zeroed and X = 0 itself (single rank only; see vcg_init_force_k).
// iters[c] = GetNumIterations() of component c.
// Everything a launch of the lockstep kernels needs: work vectors, tables of K2, the argument block (vcg_solve, and the
// one-launch test hook vcg_test_k1 below).  Launches vcg_set_tol_k on the context stream.
static const char *trace_path = nullptr; // debug: LGH_VCG_TRACE=<file>
static bool trace_asked = false;
static unsigned long long *trace_dev = nullptr;
struct VcgPlan
{
   VcgArgs a;
   VcgAux *aux;
   int k1form;
   bool k2p;
};
// the tables of K2 (and of the several-rank gathers) for the merged layout: the ELL transpose with every merged pair as
// ONE entry, its byte-offset form, node ranges balanced by the contributions that are left
static int vcg_build_merged_tables(lgh_ctx *c, VcgAux *x)
{
   SlabLayout L;
   int rc = slab_merge_layout(c, L);
   if (rc) { return rc; }
   const size_t N = (size_t)c->N;
   const int deg = c->t_deg;
   std::vector<int> ell((size_t)deg * N), ellm((size_t)8 * N, -1), val(N, 0);
   LGH_HIP_CHECK(hipMemcpy(ell.data(), x->o_ell ? x->o_ell : c->t_ell, ell.size() * sizeof(int), hipMemcpyDeviceToHost));
   int degm = 0;
   for (size_t n = 0; n < N; n++)
   {
      int cnt = 0;
      for (int j = 0; j < deg; j++)
      {
         const int p = ell[(size_t)j * N + n];
         if (p < 0 || L.sec[p]) { continue; } // (the right-hand member of a merged pair: summed into its partner's entry by K1)
         ellm[(size_t)cnt * N + n] = L.pos[p];
         cnt++;
      }
      val[n] = cnt;
      degm = std::max(degm, cnt);
   }
   x->degm = degm;
   LGH_HIP_CHECK(hipMalloc((void **)&x->settab, L.settab.size() * sizeof(unsigned)));
   LGH_HIP_CHECK(hipMemcpy(x->settab, L.settab.data(), L.settab.size() * sizeof(unsigned), hipMemcpyHostToDevice));
   LGH_HIP_CHECK(hipMalloc((void **)&x->ellm, ellm.size() * sizeof(int)));
   LGH_HIP_CHECK(hipMemcpy(x->ellm, ellm.data(), ellm.size() * sizeof(int), hipMemcpyHostToDevice));
   rc = make_ellz(c, &x->ellzm, x->ellm, 8);
   if (rc) { return rc; }
   return partition_nodes_by_cost(c, x->grid2, &x->nstartm, &val);
}

// The order the lockstep solve runs in: the library's own (lgh_order.hip) unless it is the caller's anyway.  Several ranks:
// every rank orders its own block; the node lists of the exchanges (lgh_comm.hip: the caller's numbering) are translated
// once (HaloNodeAlias) - which nodes travel, in which order and how they are summed does not depend on their numbers.
static const MeshOrder *vcg_order_for(const lgh_ctx *c)
{
   const char *env = getenv("LGH_ORDER_MULTI"); // A/B: 0 = several ranks keep the caller's numbering in the solve
   if (c->multi != 0 && env && env[0] == '0') { return nullptr; }
   return mesh_order(c);
}
// Tables and vectors of the solve in the internal order (VcgAux::o_*): the element -> node map (internal zone i, internal
// node numbers), its transpose for the solve's own E-vector and for an E-vector in the caller's zone order, essential masks,
// room for 1/diag, the mass data and the solution.
static int vcg_build_internal_tables(lgh_ctx *c, VcgAux *x)
{
   const MeshOrder *o = x->ord;
   const size_t N = (size_t)c->N, NE = (size_t)c->NE, ND = (size_t)c->ND, nmap = NE * ND;
   const int deg = c->t_deg;
   std::vector<int> hm(nmap), om(nmap);
   LGH_HIP_CHECK(hipMemcpy(hm.data(), c->h1map, nmap * sizeof(int), hipMemcpyDeviceToHost));
   for (size_t i = 0; i < NE; i++)
   {
      const size_t e = (size_t)o->zorder[i];
      for (size_t d = 0; d < ND; d++) { om[i * ND + d] = o->nnum[hm[e * ND + d]]; }
   }
   LGH_HIP_CHECK(hipMalloc((void **)&x->o_map, nmap * sizeof(int)));
   LGH_HIP_CHECK(hipMemcpy(x->o_map, om.data(), nmap * sizeof(int), hipMemcpyHostToDevice));
   // transpose, ascending E-position per node (as lgh_create builds the caller's)
   std::vector<int> off(N + 1, 0);
   for (size_t i = 0; i < nmap; i++) { off[(size_t)om[i] + 1]++; }
   x->o_valence.assign(N, 0);
   for (size_t n = 0; n < N; n++) { x->o_valence[n] = off[n + 1]; off[n + 1] += off[n]; }
   std::vector<int> pos(off.begin(), off.end() - 1), ell((size_t)deg * N, -1);
   for (size_t i = 0; i < nmap; i++)
   {
      const size_t n = (size_t)om[i];
      const int k = pos[n]++ - off[n];
      if (k >= deg) { set_error("vcg_build_internal_tables: valence above the mesh's"); return LGH_ERR_ARG; }
      ell[(size_t)k * N + n] = (int)i;
   }
   LGH_HIP_CHECK(hipMalloc((void **)&x->o_ell, ell.size() * sizeof(int)));
   LGH_HIP_CHECK(hipMemcpy(x->o_ell, ell.data(), ell.size() * sizeof(int), hipMemcpyHostToDevice));
   // the caller's transpose, rows taken in internal node order: positions in an E-vector of the CALLER's zone order, in the
   // caller's (ascending) order of contributions - the right-hand side has the bits of the unfused E -> L sum
   std::vector<int> tell((size_t)deg * N);
   LGH_HIP_CHECK(hipMemcpy(tell.data(), c->t_ell, tell.size() * sizeof(int), hipMemcpyDeviceToHost));
   for (int k = 0; k < deg; k++)
      for (size_t m = 0; m < N; m++) { ell[(size_t)k * N + m] = tell[(size_t)k * N + (size_t)o->ncaller[m]]; }
   LGH_HIP_CHECK(hipMalloc((void **)&x->o_ellc, ell.size() * sizeof(int)));
   LGH_HIP_CHECK(hipMemcpy(x->o_ellc, ell.data(), ell.size() * sizeof(int), hipMemcpyHostToDevice));
   for (int k = 0; k < kVC; k++)
   {
      if (!c->essmask[k]) { continue; }
      LGH_HIP_CHECK(hipMalloc((void **)&x->o_ess[k], N));
      const int rc = order_gather_bytes(c, c->essmask[k], x->o_ess[k]);
      if (rc) { return rc; }
   }
   if (c->multi != 0)
   {
      if (c->owner)
      {
         LGH_HIP_CHECK(hipMalloc((void **)&x->o_owner, N * sizeof(double)));
         const int rc = order_gather_nodes(c, c->owner, x->o_owner, 1);
         if (rc) { return rc; }
      }
      const uint8_t *hmask = nullptr;
      const int *sh_node = nullptr, *nodes = nullptr;
      int n_shared = 0, total = 0;
      comm_shared_nodes(c, &hmask, &sh_node, &n_shared);
      if (hmask)
      {
         LGH_HIP_CHECK(hipMalloc((void **)&x->o_hmask, N));
         const int rc = order_gather_bytes(c, hmask, x->o_hmask);
         if (rc) { return rc; }
      }
      comm_node_lists(c, &nodes, &total, &sh_node, &n_shared);
      auto translate = [&](const int *src, const int n, int **dst) -> int {
         std::vector<int> h((size_t)std::max(n, 1), 0);
         if (n > 0) { LGH_HIP_CHECK(hipMemcpy(h.data(), src, (size_t)n * sizeof(int), hipMemcpyDeviceToHost)); }
         for (int i = 0; i < n; i++) { h[i] = o->nnum[h[i]]; }
         LGH_HIP_CHECK(hipMalloc((void **)dst, h.size() * sizeof(int)));
         LGH_HIP_CHECK(hipMemcpy(*dst, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
         return LGH_OK;
      };
      if (nodes && sh_node)
      {
         int rc = translate(nodes, total, &x->o_nodes);
         if (rc == LGH_OK) { rc = translate(sh_node, n_shared, &x->o_shnode); }
         if (rc) { return rc; }
         x->o_alias = {x->o_nodes, x->o_shnode};
      }
   }
   LGH_HIP_CHECK(hipMalloc((void **)&x->o_dinv, N * sizeof(double)));
   LGH_HIP_CHECK(hipMalloc((void **)&x->o_Se, NE * sizeof(double)));
   LGH_HIP_CHECK(hipMalloc((void **)&x->o_x, kVC * N * sizeof(double)));
   LGH_HIP_CHECK(hipMemset(x->o_x, 0, kVC * N * sizeof(double)));
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   x->o_mass_gen = ~0ul;
   return LGH_OK;
}

// in_solve: the call comes from vcg_solve, which every rank enters together - only there may a collective run (the decision
// below); the statistics and test entry points take the conservative answer until a solve has decided (round-5 advisor)
static int vcg_prepare(lgh_ctx *c, double *B, double *X, double rel_tol, VcgPlan &plan, const bool in_solve = false)
{
   const bool multi = c->multi != 0;
   const size_t N = (size_t)c->N;
   int rc;
   if (!c->vcg_s)
   {
      LGH_HIP_CHECK(hipMalloc((void **)&c->vcg_s, sizeof(VcgScalars)));
      LGH_HIP_CHECK(hipMemset(c->vcg_s, 0, sizeof(VcgScalars)));
      if (!c->vcg_vec)
      {
         LGH_HIP_CHECK(hipMalloc((void **)&c->vcg_vec, 3 * kVC * N * sizeof(double))); // r, d, yL (z = r/diag is never stored)
         LGH_HIP_CHECK(hipMemset(c->vcg_vec, 0, 3 * kVC * N * sizeof(double)));
      }
      // block partials of the largest reducing launch: K1 (<= NE batches), vcg_update_k (N/256 blocks), vcg_update_p_k (2 per CU)
      c->vcg_stride = (unsigned)(std::max<size_t>(std::max<size_t>((size_t)c->NE, (N + 255) / 256), 4096) + kShards);
      LGH_HIP_CHECK(hipMalloc((void **)&c->vcg_partials, 2 * kVC * (size_t)c->vcg_stride * sizeof(double)));
      LGH_HIP_CHECK(hipMemset(c->vcg_partials, 0, 2 * kVC * (size_t)c->vcg_stride * sizeof(double)));
      LGH_HIP_CHECK(hipMalloc((void **)&c->vcg_tickets, 2 * kTicketSlot * sizeof(unsigned int)));
      LGH_HIP_CHECK(hipMemset(c->vcg_tickets, 0, 2 * kTicketSlot * sizeof(unsigned int)));
   }
   if (!c->vcg_aux) // (first solve, or the communicator has changed since: lgh_comm_init / lgh_comm_set_neighbors drop the tables)
   {
      // the CG's own E-vector (the force E-vector of the fused init stays in c->YE) and the tables of K2
      VcgAux *x = new VcgAux();
      c->vcg_aux = x;
      const size_t ye_n = (size_t)kVC * ((size_t)c->NE * c->ND + kYePad);
      LGH_HIP_CHECK(hipMalloc((void **)&x->ye, ye_n * sizeof(double)));
      LGH_HIP_CHECK(hipMemset(x->ye, 0, ye_n * sizeof(double)));
      // The library's own order of zones and nodes (lgh_order.hip), when the caller's is another one: the solve's tables are
      // built for THAT order.  (Several ranks: the exchange tables of lgh_comm.hip are in the caller's numbering - see
      // vcg_order_for.)
      x->ord = vcg_order_for(c);
      if (x->ord)
      {
         rc = vcg_build_internal_tables(c, x);
         if (rc) { return rc; }
      }
      const int *v_map = x->o_map ? x->o_map : c->h1map;
      if (c->t_deg <= 8 && (size_t)c->N * 8 * kVC < 0xffffffffull && ((size_t)c->NE * c->ND + kYePad) * 8 < 0xffffffffull)
      {
         int ncu = 256;
         hipDeviceProp_t prop;
         if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) { ncu = prop.multiProcessorCount; }
         // node ranges (= workgroups) of vcg_update_p_k: four per CU, and not more than ~1000 nodes (two passes) each -
         // on large meshes long static ranges lose to the hardware's dynamic distribution of many short ones
         // (64^3 zones: 367 vs 424 us; 128^3: H1 CG 2.72 vs 3.04 s).  LGH_K2_GRID=<ranges per CU> for A/B.
         const char *genv = getenv("LGH_K2_GRID");
         if (genv && atoi(genv) > 0) { x->grid2 = atoi(genv) * ncu; }
         else { x->grid2 = (int)std::max<long>(4L * ncu, (((long)c->N + 1023) / 1024 + 7) & ~7L); }
         x->grid2 = std::min<long>(x->grid2, (long)c->vcg_stride - (long)kShards); // (one partial per workgroup in a reduction slot)
         rc = make_ellz(c, &x->ellz, x->o_ell, x->o_ell ? c->t_deg : 0);
         if (rc) { return rc; }
         rc = make_essbits(c, &x->essbits);
         if (rc) { return rc; }
         if (x->ord)
         {
            // (the flag bytes were built from the caller's masks: into the internal numbering)
            uint8_t *tmp = nullptr;
            LGH_HIP_CHECK(hipMalloc((void **)&tmp, (size_t)c->N));
            LGH_HIP_CHECK(hipStreamSynchronize(nullptr));
            rc = order_gather_bytes(c, x->essbits, tmp);
            if (rc) { (void)hipFree(tmp); return rc; }
            LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
            (void)hipFree(x->essbits);
            x->essbits = tmp;
         }
         rc = partition_nodes_by_cost(c, x->grid2, &x->nstart, x->ord ? &x->o_valence : nullptr);
         if (rc) { return rc; }
         if ((size_t)kVC * (c->NE + 1) * c->ND * 8 < 0xffffffffull)
         {
            // (the FORCE E-vector is in the caller's zone order: o_ellc)
            const size_t ell_n = (size_t)8 * c->N;
            LGH_HIP_CHECK(hipMalloc((void **)&x->ellf, ell_n * sizeof(unsigned)));
            hipLaunchKernelGGL(vcg_ellf_k, dim3((unsigned)((ell_n + 255) / 256)), dim3(256), 0, nullptr, x->o_ellc ? x->o_ellc : c->t_ell, x->ellf,
                               (size_t)c->t_deg * c->N, ell_n, c->ND, c->NE);
            LGH_HIP_CHECK(hipGetLastError());
         }
      }
      if (vcg_k1_form(c) == 4)
      {
         const size_t nm = (size_t)c->NE * c->ND;
         LGH_HIP_CHECK(hipMalloc((void **)&x->mapb, nm * sizeof(unsigned)));
         hipLaunchKernelGGL(vcg_mapb_k, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, nullptr, v_map, x->mapb, nm);
         LGH_HIP_CHECK(hipGetLastError());
         int *flag = (int *)c->scal, h = 1;
         LGH_HIP_CHECK(hipMemset(flag, 0, sizeof(int)));
         hipLaunchKernelGGL(vcg_map_xrows_k, dim3((unsigned)((nm / c->D1D + 255) / 256)), dim3(256), 0, nullptr, v_map, nm / c->D1D, c->D1D, flag);
         LGH_HIP_CHECK(hipMemcpy(&h, flag, sizeof(int), hipMemcpyDeviceToHost));
         x->map_xrows = (h == 0) ? 1 : 0;
         LGH_HIP_CHECK(hipMalloc((void **)&x->limbs, (2 * kLimbWords + 8 * 16 + 32 * 4096) * sizeof(long long)));
         LGH_HIP_CHECK(hipMemset(x->limbs, 0, (2 * kLimbWords + 8 * 16 + 32 * 4096) * sizeof(long long)));
         LGH_HIP_CHECK(hipMalloc((void **)&x->rzl, 3 * kLimbWords * sizeof(long long)));
         LGH_HIP_CHECK(hipMemset(x->rzl, 0, 3 * kLimbWords * sizeof(long long)));
      }
      {
         const char *menv = getenv("LGH_SLAB_MERGE"); // A/B: 0 = element-local E-vector for every set (the layout of rounds 3 and 4)
         if (vcg_k1_form(c) == 4 && x->ellz && !(menv && menv[0] == '0'))
         {
            rc = vcg_build_merged_tables(c, x);
            if (rc) { return rc; }
         }
      }
      LGH_HIP_CHECK(hipStreamSynchronize(nullptr)); // the fills run asynchronously on the null stream
   }
   VcgAux *aux = (VcgAux *)c->vcg_aux;
   const int k1form = vcg_k1_form(c);
   const char *k2env = getenv("LGH_K2P"); // A/B: 0 = vcg_update_k (one node per thread, x every iteration)
   const bool k2p = aux->ellz != nullptr && !(k2env && k2env[0] == '0');
   VcgScalars *ds = (VcgScalars *)c->vcg_s;
   // exact accumulators of (d, A d): slab-form K1 with the bounded-grid K2 (LGH_SLAB_EXACT=0: ticketed fold)
   long long *limbs = (c->slab_exact && k2p && k1form == 4) ? aux->limbs : nullptr;
   // rz_limbs mode: one rank, exact accumulators with the deferred fold (LGH_RZ_LIMBS=0: the ticketed fold of (r, z) in K2)
   const char *rzenv = getenv("LGH_RZ_LIMBS");
   // (several ranks, round 5: all-pairs partitions, where the words of every rank reach every other in one exchange)
   // (a communicator of size 1 - LGH_FORCE_MULTI - has nobody to exchange with: the words are complete as they are)
   bool rz_words = limbs && (!multi || ((halo_can_piggyback(c) || c->nranks == 1) && c->t_deg <= 8)) && !(rzenv && rzenv[0] == '0');
   // ... and whether the energy CG can ride on this solve's exchanges (lockstep): a fourth scalar on the halo
   // messages - their LENGTH, which both ends of a message must agree on - so this too is every rank's answer or nobody's
   bool ls_mine = false;
   if (multi && rz_words)
   {
      const char *e0 = getenv("LGH_HALO_FUSED_PACK");
      HaloPackTables hp0;
      const uint8_t *hm0 = nullptr; const int *sn0 = nullptr; int nsh0 = 0;
      comm_shared_nodes(c, &hm0, &sn0, &nsh0);
      ls_mine = k2p && c->t_deg <= 8 && l2_lockstep_possible(c) && comm_ranks_before(c) >= 0 &&
                (nsh0 == 0 || (!(e0 && e0[0] == '0') && halo_can_piggyback(c) && comm_pack_tables(c, &hp0)));
   }
   if (multi && c->nranks == 1) { aux->ls_all = ls_mine ? 1 : 0; }
   if (multi && c->nranks > 1)
   {
      // Which exchange follows K2 - accumulator words or three doubles - must be the same on every rank, and whether a rank
      // CAN use the words depends on its own kernels (the slab K1 is dispatched by the rank's zone count): decided
      // collectively, once per set of tables (a MIN over the ranks; every rank builds its tables in its first solve)
      // (what this rank can do depends on switches that may be flipped between solves: the answer is cached with them)
      const int key = (rz_words ? 1 : 0) | (c->slab_exact ? 2 : 0) | ((c->vcg_variant & 0xff) << 2) | (k1form << 10);
      if (aux->rz_words_key != key) { aux->rz_words_all = -1; }
      if (aux->rz_words_all < 0 && !in_solve) { rz_words = false; }
      else if (aux->rz_words_all < 0)
      {
         aux->rz_words_key = key;
         const double mine[2] = {rz_words ? 1.0 : 0.0, ls_mine ? 1.0 : 0.0};
         LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
         LGH_HIP_CHECK(hipMemcpy(c->scal + 12, mine, sizeof(mine), hipMemcpyHostToDevice)); // (pageable host memory: synchronous copies)
         rc = allreduce_dev(c, c->scal + 12, 2, 1);
         if (rc) { return rc; }
         double all[2] = {0.0, 0.0};
         LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
         LGH_HIP_CHECK(hipMemcpy(all, c->scal + 12, sizeof(all), hipMemcpyDeviceToHost));
         aux->rz_words_all = (all[0] > 0.5) ? 1 : 0;
         aux->ls_all = (all[0] > 0.5 && all[1] > 0.5) ? 1 : 0;
      }
      rz_words = rz_words && aux->rz_words_all == 1;
   }
   long long *rzl = rz_words ? aux->rzl : nullptr;
   {
      const char *e0 = getenv("LGH_SLAB_DEFER");
      if (e0 && e0[0] == '0') { rzl = nullptr; } // (needs the deferred fold of (d, A d) as well)
   }
   // (a solve resets its scalars and accumulators in its first kernel: a.reset below; the hooks and statistics entry points,
   //  which launch single kernels on state of their own, keep the separate launch)
   if (!in_solve) { hipLaunchKernelGGL(vcg_set_tol_k, dim3(1), dim3(1), 0, c->stream, ds, rel_tol * rel_tol, limbs, rzl); }

   VcgArgs a;
   memset(&a, 0, sizeof(a));
   a.NE = c->NE;
   a.N = c->N;
   a.B = c->B;
   a.DqFull = c->massD;
   rc = mass_data(c, &a.Dq, &a.dqs, &a.Se);
   a.w1 = (a.dqs == 0) ? c->w1d : nullptr;
   a.M1 = (a.dqs == 0 && c->w1d) ? c->M1h : nullptr; // (nullptr with LGH_MASS_KRON=0)
   if (rc) { return rc; }
   a.map = c->h1map;
   a.ell = c->t_ell;
   a.deg = c->t_deg;
   for (int k = 0; k < kVC; k++) { a.ess[k] = c->essmask[k]; }
   a.dinv = c->dinvV;
   a.owner = multi ? c->owner : nullptr;
   a.b = B;
   a.x = X;
   if (aux->ord)
   {
      // the library's own order: per-zone and per-node data of the context through the permutation (copies, refreshed when
      // the mass data or the Jacobi diagonal have changed since they were taken)
      const bool need_table = a.dqs != 0 || k1form == 0; // (the column form of K1 reads the stored table whatever the compact form says)
      if (aux->o_mass_gen != c->mass_gen)
      {
         rc = order_gather_nodes(c, c->dinvV, aux->o_dinv, 1);
         if (rc) { return rc; }
         if (a.dqs == 0) { rc = order_zone_blocks(c, a.Se, aux->o_Se, 1, false); }
         if (rc) { return rc; }
         if (need_table)
         {
            if (!aux->o_massD) { LGH_HIP_CHECK(hipMalloc((void **)&aux->o_massD, ((size_t)c->NE * c->NQ + 2048) * sizeof(double))); LGH_HIP_CHECK(hipMemset(aux->o_massD, 0, ((size_t)c->NE * c->NQ + 2048) * sizeof(double))); }
            rc = order_zone_blocks(c, c->massD, aux->o_massD, c->NQ, false);
            if (rc) { return rc; }
         }
         aux->o_mass_gen = c->mass_gen;
      }
      if (a.dqs == 0) { a.Se = aux->o_Se; }
      else { a.Dq = aux->o_massD; } // (a.Se: ones)
      a.DqFull = need_table ? aux->o_massD : nullptr;
      a.map = aux->o_map;
      a.ell = aux->o_ell;
      for (int k = 0; k < kVC; k++) { a.ess[k] = aux->o_ess[k]; }
      a.dinv = aux->o_dinv;
      a.owner = multi ? aux->o_owner : nullptr;
      a.ncaller = aux->ord->ncaller_d;
      a.x = aux->o_x;
   }
   a.r = c->vcg_vec;
   a.d = c->vcg_vec + kVC * N;
   a.yL = c->vcg_vec + 2 * kVC * N;
   a.YE = aux->ye;
   a.ye_stride = (size_t)c->NE * c->ND + kYePad;
   {
      const char *w0 = getenv("LGH_SLAB_YE_WIDE"); // test hook: 1 = the per-set 64-bit store base of the slab K1 on a mesh that does not need it
      a.ye_wide = (a.ye_stride * 8 * kVC >= 0xffffffffull || (w0 && w0[0] == '1')) ? 1 : 0;
   }
   a.ellz = aux->ellz;
   a.essbits = aux->essbits;
   a.nstart = aux->nstart;
   a.mapb = aux->mapb;
   a.map_xrows = aux->map_xrows;
   if (k1form == 4 && aux->settab)
   {
      // merged E-vector layout: K1 stores by the set table, everybody who reads Y_E goes through the tables of that layout
      a.settab = aux->settab;
      a.ell = aux->ellm;
      a.deg = aux->degm;
      a.ellz = aux->ellzm;
      a.nstart = aux->nstartm;
   }
   a.limbs = limbs;
   a.rzl = rzl;
   a.reset = in_solve ? 1 : 0;
   a.reset_tol2 = rel_tol * rel_tol;
   if (rzl && multi)
   {
      rc = comm_word_peers(c, kLimbWords, &a.rzl_peers, &a.n_rz_peers);
      if (rc) { return rc; }
   }
   a.queue = limbs ? (unsigned *)(limbs + 2 * kLimbWords) : nullptr;
   {
      const char *e0 = getenv("LGH_SLAB_DEFER"); // A/B: 0 = the last workgroup of K1 folds the accumulators (ticket), K2 reads the result
      a.den_limbs = (limbs && !(e0 && e0[0] == '0')) ? (multi ? 2 : 1) : 0; // (several ranks: folded before the exchange, vcg_fold_den)
      e0 = getenv("LGH_SLAB_STORE_WAIT"); // A/B: 1 = a wavefront waits for the stores of a pass before the next one (it paid at 64^3 while the stores were partial lines: 306 vs 370 us; with whole lines 275 vs 248)
      a.store_wait = (e0 && e0[0] == '1') ? 1 : 0;
   }
   {
      const char *e0 = getenv("LGH_K2_SKIP");
      a.k2_skip = (e0 && e0[0] == '0') ? 0 : 1;
   }
   a.s = ds;
   a.stride = c->vcg_stride;
   a.multi = multi ? 1 : 0;
   // debug: LGH_VCG_TRACE=<file> dumps "block start loop_end end (xcc<<32|hw_id)" of the
   // last K1 launch of every solve, in 10 ns ticks
   if (!trace_asked) { trace_path = getenv("LGH_VCG_TRACE"); trace_asked = true; }
   if (trace_path && !trace_dev) { (void)hipMalloc((void **)&trace_dev, kTraceRec * 4096 * sizeof(unsigned long long)); (void)hipMemset(trace_dev, 0, kTraceRec * 4096 * sizeof(unsigned long long)); }
   a.trace = trace_dev;
   if (trace_dev) { (void)hipMemsetAsync(trace_dev, 0, kTraceRec * 4096 * sizeof(unsigned long long), c->stream); }
   if (multi)
   {
      comm_shared_nodes(c, &a.hmask, &a.sh_node, &a.n_shared);
      if (aux->ord && a.n_shared > 0) { a.hmask = aux->o_hmask; a.sh_node = aux->o_shnode; } // (the same nodes by their internal numbers)
   }
   {
      // several ranks, bounded-grid K2: the shared-node gather fills the send buffer of the halo exchange itself (and
      // the local (d, A d) with it where the sums ride on the messages), the last workgroup of K2 the one of the
      // (r, z) exchange - two launches less per iteration (A/B: LGH_HALO_FUSED_PACK=0)
      const char *e0 = getenv("LGH_HALO_FUSED_PACK");
      const bool want = !(e0 && e0[0] == '0');
      const bool mixed0 = multi && c->t_deg <= 8;
      if (want && mixed0 && k2p && a.n_shared > 0 && comm_pack_tables(c, &a.hp))
      {
         a.pack_halo = 1;
         a.nx_den = halo_can_piggyback(c) ? (c->e_lockstep ? kVC + 1 : kVC) : 0; // (lockstep energy CG: its (d, M d) rides along, VcgScalars::den_e)
         a.pack_rz = (halo_can_piggyback(c) && !rzl) ? 1 : 0; // (rz_limbs mode: (r, z) travels as accumulator words, exchange_words)
      }
   }
   if (in_solve)
   {
      // (the collective answer, and nothing of this rank's own has changed since it was given)
      aux->ls_ready = (multi && aux->ls_all == 1 && ls_mine && a.rzl && (a.n_shared == 0 || (a.pack_halo && halo_can_piggyback(c)))) ? 1 : 0;
   }
   plan.a = a;
   plan.aux = aux;
   plan.k1form = k1form;
   plan.k2p = k2p;
   return LGH_OK;
}
bool vcg_lockstep_ready(const lgh_ctx *c)
{
   const VcgAux *x = (const VcgAux *)c->vcg_aux;
   return x && x->ls_ready != 0;
}

// one launch of K1 in the form vcg_k1_form() names (a.iter, a.partials, a.ticket set by the caller)
static void vcg_launch_k1(lgh_ctx *c, const VcgPlan &plan, const VcgArgs &a)
{
   VcgAux *aux = plan.aux;
   const int k1form = plan.k1form;
   const char *knenv = getenv("LGH_KRON_NEB");
   const bool kron_small = !(knenv && knenv[0] == '0');
   if (k1form == 5)
   {
      switch (c->D1D)
      {
         case 2: launch_vcg_kron<2, 64>(c, a); break;
         case 3: launch_vcg_kron<3, 32>(c, a); break;
         case 4: launch_vcg_kron<4, 16>(c, a); break;
         // (zones per workgroup at D = 5, 6, round 6: three / two - every thread at most one row of a stage and five workgroups
         //  in a CU's LDS - instead of eight / four (two workgroups per CU): 433 -> 345 us per launch at config 5, Q4Q3 steps
         //  - 2 %, profiles/r6_kron_neb.txt; LGH_KRON_NEB=0: eight / four)
         case 5: if (kron_small) { launch_vcg_kron<5, 3>(c, a); } else { launch_vcg_kron<5, 8>(c, a); } break;
         default: if (kron_small) { launch_vcg_kron<6, 2>(c, a); } else { launch_vcg_kron<6, 4>(c, a); } break;
      }
      return;
   }
   switch (c->kid)
   {
      case 0x322: VCG_DISPATCH(2, 2); break;
      case 0x334: VCG_DISPATCH(3, 4); break;
      case 0x346:
         if (aux->mapb && k1form == 4) { launch_vcg_slab(c, a); } // slab form: all in registers (lgh_vcg_slab.hip)
         else { VCG_DISPATCH(4, 6); }
         break;
      case 0x358: // LGH_VCG_VARIANT=1: the two-lanes-per-plane split at Q1D = 8 as well (A/B, tests)
         if (c->vcg_variant == 0 || !c->b_h1_sym) { launch_vcg_apply<5, 8>(c, a); } // (the plane form uses half a table)
         else if (c->vcg_variant == 1) { launch_vcg_plane_ho<5, 8, 2, 5>(c, a); }
         else { launch_vcg_plane_ho<5, 8, 1, 5>(c, a); }
         break;
      case 0x36A:
         if (c->vcg_variant == 0 || !c->b_h1_sym) { launch_vcg_apply<6, 10>(c, a); }
         else { launch_vcg_plane_ho<6, 10, 2, 4>(c, a); }
         break;
   }
}

// one launch of K2 as the one-rank solve issues it in iteration `it` (a.iter, a.partials, a.ticket set by the caller):
// the bounded-grid kernel (x updated every second iteration) or the round-1 kernel (LGH_K2P=0, unusual valence)
static void vcg_launch_k2p(lgh_ctx *c, const VcgPlan &plan, const VcgArgs &a, const int it)
{
   kt_begin(c, LGH_KERNEL_CG_UPDATE_H1);
   const char *uenv = getenv("LGH_K2_U"); // A/B: nodes per thread and pass (2: 182 VGPRs, one workgroup per CU resident)
   const int u2 = (uenv && uenv[0] == '2') ? 1 : 0;
   // LGH_K2_OCC (A/B): 4 = the register budget of rounds 2-4 for both launches, 6 = at most 80 registers for both (the launch
   // that updates x then spills 7), default: 80 for the launch without x, the old budget for the one with it
   const char *oenv = getenv("LGH_K2_OCC"); // (per launch, like LGH_K2_U: the switch tests flip it inside one process)
   const int occ = (oenv && oenv[0] == '4') ? 0 : (oenv && oenv[0] == '6') ? 2 : 1;
   const int occ3 = (it & 1) ? (occ >= 1) : (occ == 2);
#define LGH_K2P_LAUNCH(XU_, U_, MINW_) hipLaunchKernelGGL((vcg_update_p_k<XU_, U_, MINW_>), dim3(plan.aux->grid2), dim3(512), 0, c->stream, a)
   if (it & 1) { if (u2) { LGH_K2P_LAUNCH(false, 2, 2); } else if (occ3) { LGH_K2P_LAUNCH(false, 1, 6); } else { LGH_K2P_LAUNCH(false, 1, 4); } }
   else { if (u2) { LGH_K2P_LAUNCH(true, 2, 2); } else if (occ3) { LGH_K2P_LAUNCH(true, 1, 6); } else { LGH_K2P_LAUNCH(true, 1, 4); } }
#undef LGH_K2P_LAUNCH
   kt_end(c, LGH_KERNEL_CG_UPDATE_H1);
}

int vcg_solve(lgh_ctx *c, double *B, double *X, double rel_tol, int max_iter, int iters[3], const double *force_E)
{
   if (!vcg_supported(c)) { return LGH_ERR_UNSUPPORTED; }
   VcgPlan plan;
   int rc = vcg_prepare(c, B, X, rel_tol, plan, true);
   if (rc) { return rc; }
   VcgArgs &a = plan.a;
   VcgAux *aux = plan.aux;
   const bool k2p = plan.k2p;
   const bool multi = c->multi != 0;
   const size_t N = (size_t)c->N;
   VcgScalars *ds = (VcgScalars *)c->vcg_s;
   const int nb = ceil_div((long)N, 256);
   const HaloNodeAlias *alias = (aux->ord && aux->o_alias.nodes) ? &aux->o_alias : nullptr; // (a.yL is in the solve's own numbering)

   // init (vector kernels use reduction slot 0, the element kernel slot 1)
   a.partials = c->vcg_partials;
   a.ticket = c->vcg_tickets;
   if (force_E)
   {
      if (multi || c->t_deg > 8) { set_error("vcg_solve: fused init is single-rank, degree <= 8"); return LGH_ERR_ARG; }
      if (aux->ellf && aux->essbits)
      {
         hipLaunchKernelGGL(vcg_init_force_z_k, dim3(nb), dim3(256), 0, c->stream, a, force_E, 8u * (unsigned)c->ND, aux->ellf, B);
      }
      else { hipLaunchKernelGGL(vcg_init_force_k<8>, dim3(nb), dim3(256), 0, c->stream, a, force_E, c->ND, B, aux->o_ellc ? aux->o_ellc : c->t_ell, c->t_deg); }
   }
   else
   {
      if (aux->ord) { LGH_HIP_CHECK(hipMemsetAsync(a.x, 0, kVC * N * sizeof(double), c->stream)); } // (X = 0 on entry: the solve's own copy as well)
      hipLaunchKernelGGL(vcg_init_k, dim3(nb), dim3(256), 0, c->stream, a);
   }
   LGH_HIP_CHECK(hipGetLastError());
   if (multi)
   {
      rc = allreduce_dev(c, ds->rz, kVC, 0);
      if (rc) { return rc; }
      hipLaunchKernelGGL(vcg_init_finish_k, dim3(1), dim3(1), 0, c->stream, ds);
   }

   VcgScalars *hs = (VcgScalars *)(c->host_pinned + 32);
   static_assert(sizeof(VcgScalars) <= 64 * sizeof(double), "pinned staging too small");
   int it = 0, looks = 0;
   bool energy_polled = false;
   // The energy CG in lockstep (lgh_solve_energy_begin set it up on this stream; lgh_mass.hip): one of its iterations per
   // velocity iteration - apply behind K1, update behind K2 - while it has some left of the count it needed last time; its
   // (d, M d) rides on the halo messages (den_e, the fourth scalar), its (r, r) in a word of the accumulator-word exchange.
   const bool ls = c->e_lockstep == 1 && multi && a.rzl && aux->ls_ready && (a.n_shared == 0 || a.nx_den == kVC + 1);
   const int ls_limit = ls ? std::min(l2_lockstep_limit(c), max_iter) : 0;
   const int ls_before = ls ? comm_ranks_before(c) : 0;
   auto ls_words = [&](const int set_it) { return LockstepWords{a.rzl + (set_it % 3) * kLimbWords, a.rzl_peers, a.n_rz_peers, ls_before}; };
   // ... with its kernels on the second stream (c->ls_side): `side(f)` enqueues f there; the velocity iteration and the energy
   // iteration meet at four events - the apply after the word exchange of the iteration before (whose words it reads) and before
   // the halo messages are packed (its (d, M d) rides on them), the update after the halo sums and before the word exchange
   // (which carries its (r, r)).  In between the apply runs beside K1 and the update beside K2.
   const bool ls_side = ls && c->ls_side != 0 && c->stream2 != nullptr;
   auto side = [&](auto &&f) -> int {
      if (!ls_side) { return f(); }
      std::swap(c->stream, c->stream2);
      const int r = f();
      std::swap(c->stream, c->stream2);
      return r;
   };
   // (record on `from`, wait on the other stream)
   auto hand_over = [&](hipEvent_t ev, const bool from_side) -> int {
      if (!ls_side) { return LGH_OK; }
      LGH_HIP_CHECK(hipEventRecord(ev, from_side ? c->stream2 : c->stream));
      LGH_HIP_CHECK(hipStreamWaitEvent(from_side ? c->stream : c->stream2, ev, 0));
      return LGH_OK;
   };
   if (ls_side) { rc = hand_over(c->ev_fork, false); if (rc) { return rc; } } // (the energy solve's set-up and this solve's are on the main stream)
   // first chunk = iteration count of the previous velocity solve (see cg_solve)
   int chunk = c->vcg_last > 0 ? c->vcg_last : 8;
   bool first_look = true;
   while (true)
   {
      // The first chunk is enqueued without looking at the flag first (an all-zero
      // right-hand side just makes its launches return at once): one host round trip,
      // with the GPU idle meanwhile, less per solve.
      if (!first_look || max_iter <= 0)
      {
         // several ranks: the outcome of the last enqueued update is still pending (vcg_pending_update) - commit it
         // (the one-thread kernel that commits the outcome also puts the scalars into the host's pinned memory: no copy
         //  kernel between it and the synchronisation)
         VcgScalars *hs_dev = (VcgScalars *)(c->host_pinned_dev + 32);
         unsigned long long *tok_dev = (unsigned long long *)(c->host_pinned_dev + 80);
         volatile unsigned long long *tok = (volatile unsigned long long *)(c->host_pinned + 80);
         const unsigned long long token = ++c->look_token;
         bool copied = false;
         // the first look of a solve comes after a whole chunk of iterations: the energy solve on the second stream is
         // brought to its end meanwhile (its looks used to wait behind the velocity solve, with the GPU idle)
         if (!energy_polled) { rc = energy_overlap_poll(c); energy_polled = true; if (rc) { return rc; } }
         if (multi && !a.rzl && it > 0) { hipLaunchKernelGGL(vcg_update_finish_k, dim3(1), dim3(1), 0, c->stream, ds, it, hs_dev, tok_dev, token); copied = true; }
         if (a.rzl && it > 0) { hipLaunchKernelGGL(vcg_rz_finish_k, dim3(1), dim3(1), 0, c->stream, ds, a.rzl, it, a.rzl_peers, a.n_rz_peers, hs_dev, tok_dev, token); copied = true; }
         if (copied)
         {
            // (on several ranks the exchanges that follow are host-synchronous on the loopback transports and enqueue-only over
            //  RCCL: either way nothing of this rank is in flight behind the finishing kernel)
            rc = host_wait_token(c, tok, token);
            if (rc) { return rc; }
         }
         else
         {
            LGH_HIP_CHECK(hipMemcpyAsync(hs, ds, sizeof(VcgScalars), hipMemcpyDeviceToHost, c->stream));
            LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
         }
         if (hs->all_done || it >= max_iter) { break; }
         chunk = (++looks <= 2) ? 2 : std::min(64, 2 * chunk); // (as cg_solve, lgh_mass.hip)
      }
      first_look = false;
      const int upto = std::min(max_iter, it + chunk);
      while (it < upto)
      {
         ++it;
         a.iter = it;
         a.partials = c->vcg_partials + (size_t)kVC * c->vcg_stride;
         a.ticket = c->vcg_tickets + kTicketSlot;
         kt_begin(c, LGH_KERNEL_MASS_CG_H1);
         vcg_launch_k1(c, plan, a);
         kt_end(c, LGH_KERNEL_MASS_CG_H1);
         LGH_HIP_CHECK(hipGetLastError());
         const bool ls_it = ls && it <= ls_limit;
         if (ls_it)
         {
            rc = side([&] { return l2_lockstep_apply(c, it, ls_words(it - 1), &ds->den_e); });
            if (rc) { return rc; }
            rc = hand_over(c->ls_ev[1][it & 3], true); // (the messages are packed after this)
            if (rc) { return rc; }
         }
         a.partials = c->vcg_partials;
         a.ticket = c->vcg_tickets;
         auto launch_k2p = [&]() { vcg_launch_k2p(c, plan, a, it); };
         if (k2p && !multi) { launch_k2p(); }
         else if (!multi && c->t_deg <= 8)
         {
            kt_begin(c, LGH_KERNEL_CG_UPDATE_H1);
            hipLaunchKernelGGL((vcg_update_k<true, 8>), dim3(nb), dim3(256), 0, c->stream, a);
            kt_end(c, LGH_KERNEL_CG_UPDATE_H1);
         }
         else
         {
            // A d as L-vectors: only at the nodes shared with other ranks when K2 can
            // gather the rest itself; sum shared nodes across ranks, all-reduce the scalars
            const bool mixed = multi && c->t_deg <= 8;
            if (mixed)
            {
               if (a.n_shared > 0)
               {
                  hipLaunchKernelGGL(vcg_gather_list_k, dim3(ceil_div(a.n_shared, 256)), dim3(256), 0, c->stream, a);
               }
               else if (a.den_limbs == 2) { hipLaunchKernelGGL(vcg_fold_den_k, dim3(1), dim3(256), 0, c->stream, a); }
            }
            else
            {
               if (a.den_limbs == 2) { hipLaunchKernelGGL(vcg_fold_den_k, dim3(1), dim3(256), 0, c->stream, a); }
               hipLaunchKernelGGL(vcg_gather_k, dim3(nb), dim3(256), 0, c->stream, a);
            }
            LGH_HIP_CHECK(hipGetLastError());
            if (multi)
            {
               // (d, A d): summed over the ranks by the halo messages themselves when every
               // rank is a neighbour of every other, by an all-reduce otherwise
               if (halo_can_piggyback(c))
               {
                  rc = halo_sum(c, a.yL, kVC, ds->den, a.nx_den > kVC ? a.nx_den : kVC, a.pack_halo != 0, alias);
                  if (rc) { return rc; }
               }
               else
               {
                  rc = halo_sum(c, a.yL, kVC, nullptr, 0, a.pack_halo != 0, alias);
                  if (rc) { return rc; }
                  rc = allreduce_dev(c, ds->den, kVC, 0);
                  if (rc) { return rc; }
               }
            }
            if (ls_it) { rc = hand_over(c->ls_ev[2][it & 3], false); if (rc) { return rc; } } // (the energy update reads the summed (d, M d))
            // (the bounded-grid K2 knows the shared nodes and the owner weights from its flag bytes)
            if (mixed && k2p) { launch_k2p(); }
            else if (mixed) { hipLaunchKernelGGL((vcg_update_k<true, 8>), dim3(nb), dim3(256), 0, c->stream, a); }
            else { hipLaunchKernelGGL((vcg_update_k<false, 0>), dim3(nb), dim3(256), 0, c->stream, a); }
            if (ls_it)
            {
               rc = side([&] { return l2_lockstep_update(c, it, &ds->den_e, a.rzl + (it % 3) * kLimbWords); });
               if (rc) { return rc; }
               rc = hand_over(c->ls_ev[3][it & 3], true); // (the word exchange carries its (r, r))
               if (rc) { return rc; }
            }
            if (multi && a.rzl)
            {
               // exact all-reduce of (r, z): the accumulator words K2 just added into go to every peer as they are - no fold,
               // no pack, no combine kernel; the next K1 (or vcg_rz_finish_k) adds own and peers' words before it folds
               rc = exchange_words(c, a.rzl + (it % 3) * kLimbWords, kLimbWords);
               if (rc) { return rc; }
               if (ls_it) { rc = hand_over(c->ls_ev[0][it & 3], false); if (rc) { return rc; } } // (the next apply, or the fold, reads these words)
               if (ls_it && (it == ls_limit || it == upto))
               {
                  // the energy CG's last interleaved iteration of this chunk: its outcome is committed now, while the peers'
                  // words of THIS exchange are still in the buffer (the next apply, if there is one, finds the same sum)
                  rc = side([&] { return l2_lockstep_fold(c, it, ls_words(it)); });
                  if (rc) { return rc; }
               }
            }
            else if (multi)
            {
               rc = allreduce_dev(c, ds->rz, kVC, 0, a.pack_rz != 0 && mixed && k2p); // convergence is looked at by the next K1 (vcg_pending_update)
               if (rc) { return rc; }
            }
         }
         LGH_HIP_CHECK(hipGetLastError());
      }
   }
   if (ls_side) { rc = hand_over(c->ev_join, true); if (rc) { return rc; } } // (lgh_solve_energy_end continues on the main stream)
   if (aux->ord)
   {
      // the solution, in the caller's numbering, into the caller's vector (with the pending update of vcg_xfix_k)
      hipLaunchKernelGGL(vcg_xout_k, dim3(nb), dim3(256), 0, c->stream, a, X);
      LGH_HIP_CHECK(hipGetLastError());
   }
   else if (k2p && ((hs->nupd[0] | hs->nupd[1] | hs->nupd[2]) & 1))
   {
      // x lags one update behind for the components that stopped after an odd number of updates
      hipLaunchKernelGGL(vcg_xfix_k, dim3(nb), dim3(256), 0, c->stream, a);
      LGH_HIP_CHECK(hipGetLastError());
   }
   if (trace_dev)
   {
      std::vector<unsigned long long> h(kTraceRec * 4096);
      (void)hipMemcpy(h.data(), trace_dev, h.size() * 8, hipMemcpyDeviceToHost);
      FILE *f = fopen(trace_path, "w");
      if (f)
      {
         for (int i = 0; i < 4096; i++)
         {
            if (h[(size_t)kTraceRec * i] == 0) { continue; } // (no such workgroup in the last launch)
            fprintf(f, "%d", i);
            for (int k = 0; k < kTraceRec; k++) { fprintf(f, " %llu", h[(size_t)kTraceRec * i + k]); }
            fprintf(f, "\n");
         }
         fclose(f);
      }
   }
   int mx = 0;
   for (int k = 0; k < kVC; k++)
   {
      // upstream: final_iter = max_iter when the loop runs out without converging
      int fin = hs->iters[k];
      if (!hs->done[k] && it >= max_iter) { fin = max_iter; }
      iters[k] = fin;
      mx = std::max(mx, fin);
   }
   c->vcg_last = mx;
   return LGH_OK;
}

__global__ void vcg_test_put_rz_k(long long *set, double v0, double v1, double v2, double s0, double s1, double s2)
{
   const double v[kVC] = {v0, v1, v2}, sc[kVC] = {s0, s1, s2};
   for (int k = 0; k < kVC; k++)
   {
      long long acc[kLimbs] = {0, 0, 0, 0};
      const bool ok = exact_add(acc, v[k], exact_scale(sc[k]));
      for (int j = 0; j < kLimbs; j++) { set[kLimbs * k + j] = acc[j]; }
      if (!ok) { set[kLimbShards * kVC * kLimbs] = 1; }
   }
}
// Test hooks and the merged layout: the hooks speak the element-local E-vector ([e][d] per component) on both sides.
// Out of K1: every entry from its place in the plane; the right-hand member of a merged pair reads 0.0 and the sum K1
// formed is reported in the left-hand zone's entry (lgh_test_vcg_merged_faces tells the caller which entries those are).
// Into K2: a merged place receives the sum of its two entries (what K1 would have stored there).
__global__ void __launch_bounds__(256)
vcg_unpack_merged_k(const double *__restrict__ plane, const int *__restrict__ pos, const uint8_t *__restrict__ sec, double *__restrict__ out, const size_t n)
{
   const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { out[i] = sec[i] ? 0.0 : plane[pos[i]]; }
}
__global__ void __launch_bounds__(256)
vcg_pack_merged_k(const double *__restrict__ in, const int *__restrict__ pos, const uint8_t *__restrict__ sec, const int ND, double *__restrict__ plane, const size_t n)
{
   const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n || sec[i]) { return; }
   double v = in[i];
   const size_t j = i + (size_t)ND - 3; // same (dy, dz), dx = 0 of the next zone - the partner of a dx = 3 entry of a chain
   if ((i % 4) == 3 && j < n && sec[j] && pos[j] == pos[i]) { v += in[j]; }
   plane[pos[i]] = v;
}
struct MergedDev
{
   int *pos = nullptr;
   uint8_t *sec = nullptr;
   ~MergedDev() { (void)hipFree(pos); (void)hipFree(sec); }
};
static int merged_to_device(lgh_ctx *c, MergedDev &m)
{
   SlabLayout L;
   const int rc = slab_merge_layout(c, L);
   if (rc) { return rc; }
   LGH_HIP_CHECK(hipMalloc((void **)&m.pos, L.pos.size() * sizeof(int)));
   LGH_HIP_CHECK(hipMalloc((void **)&m.sec, L.sec.size()));
   LGH_HIP_CHECK(hipMemcpy(m.pos, L.pos.data(), L.pos.size() * sizeof(int), hipMemcpyHostToDevice));
   LGH_HIP_CHECK(hipMemcpy(m.sec, L.sec.data(), L.sec.size(), hipMemcpyHostToDevice));
   return LGH_OK;
}
// lgh_test_vcg_merged_faces: mask[e * ND + d] = 1 where K1 of this context has already summed the entry into its left
// neighbour's (all zero when the layout is element-local: every other form of K1, LGH_SLAB_MERGE=0)
int vcg_test_merged_faces(lgh_ctx *c, unsigned char *mask, long *n_merged)
{
   const size_t nE = (size_t)c->NE * c->ND;
   memset(mask, 0, nE);
   *n_merged = 0;
   if (!vcg_supported(c)) { return LGH_OK; }
   VcgPlan plan;
   const int rc = vcg_prepare(c, nullptr, nullptr, 0.0, plan);
   if (rc) { return rc; }
   if (!plan.a.settab) { return LGH_OK; }
   SlabLayout L;
   const int rc2 = slab_merge_layout(c, L);
   if (rc2) { return rc2; }
   // (the layout is that of the order the solve runs in; the caller's E-vector is in the caller's zone order)
   const MeshOrder *ord = plan.aux->ord;
   if (ord)
   {
      const size_t ND = (size_t)c->ND;
      for (size_t i = 0; i < (size_t)c->NE; i++) { memcpy(mask + (size_t)ord->zorder[i] * ND, L.sec.data() + i * ND, ND); }
   }
   else { memcpy(mask, L.sec.data(), nE); }
   *n_merged = (long)L.n_merged;
   return LGH_OK;
}

// lgh_vcg_layout_stats: what K1 hands to K2 in this context, for byte accounting (bench.py's moved_bytes): out[0] = doubles
// per component plane K1 writes and K2 reads (NE * ND element-local; less with the merged layout), out[1] = bytes of the
// ELL table K2 always reads (slots 0..3: 16 per node), out[2] = bytes of its second table (slots 4..7) in the 64-node
// blocks that hold a node with more than four contributions (what the wavefronts of K2 fetch of it, to the granularity
// of a wavefront), out[3] = E-vector entries K1 has summed into a neighbour's (0: element-local layout)
int vcg_layout_stats(lgh_ctx *c, long out[4])
{
   out[0] = (long)c->NE * c->ND; out[1] = 16L * c->N; out[2] = 16L * c->N; out[3] = 0;
   if (!vcg_supported(c)) { return LGH_OK; }
   VcgPlan plan;
   int rc = vcg_prepare(c, nullptr, nullptr, 0.0, plan);
   if (rc) { return rc; }
   const VcgArgs &a = plan.a;
   if (a.settab)
   {
      SlabLayout L;
      rc = slab_merge_layout(c, L);
      if (rc) { return rc; }
      long used = 0;
      for (size_t i = 0; i < L.pos.size(); i++) { used = std::max(used, (long)L.pos[i] + 1); }
      out[0] = used;
      out[3] = (long)L.n_merged;
   }
   const size_t N = (size_t)c->N;
   if (a.deg > 4)
   {
      std::vector<int> row((size_t)(a.deg - 4) * N);
      LGH_HIP_CHECK(hipMemcpy(row.data(), a.ell + 4 * N, row.size() * sizeof(int), hipMemcpyDeviceToHost));
      long blocks = 0;
      for (size_t b = 0; b < N; b += 64)
      {
         bool any = false;
         for (size_t n = b; n < std::min(N, b + 64) && !any; n++) { any = row[n] >= 0; } // (rows hold their contributions first: slot 4 tells)
         blocks += any ? 1 : 0;
      }
      out[2] = 16L * 64 * blocks;
   }
   else { out[2] = 0; }
   return LGH_OK;
}

// Test hook (lgh_test_vcg_k1): ONE launch of K1, in whichever form vcg_solve dispatches for this context, exactly as
// the solve would launch it in its first iteration (first != 0: d = r/diag) or in a later one (d = r/diag + beta d_old
// with beta = rz / rz_prev), on the caller's vectors.  Returns what K1 hands to K2: the element contributions
// A_e d_e of the three components as E-vectors (kVC planes of NE*ND, element-local lexicographic) and (d, A d).
// Single rank.  The vectors of the lockstep solve (r, d) are overwritten; nothing else of the context changes.
int vcg_test_k1(lgh_ctx *c, const double *r, const double *d_old, const double rz[3], const double rz_prev[3],
                int first, double *YE_out, double den_out[3])
{
   if (!vcg_supported(c)) { set_error("lgh_test_vcg_k1: no lockstep solve for kernel 0x%x", c->kid); return LGH_ERR_UNSUPPORTED; }
   if (c->multi != 0) { set_error("lgh_test_vcg_k1: single rank only"); return LGH_ERR_ARG; }
   VcgPlan plan;
   int rc = vcg_prepare(c, nullptr, nullptr, 0.0, plan);
   if (rc) { return rc; }
   VcgArgs &a = plan.a;
   const size_t N = (size_t)c->N, nE = (size_t)c->NE * c->ND;
   VcgScalars *ds = (VcgScalars *)c->vcg_s;
   const MeshOrder *ord = plan.aux->ord; // (the solve's own order: the caller's vectors and E-vector go through the permutation)
   if (ord) { rc = order_gather_nodes(c, r, a.r, kVC); if (rc) { return rc; } }
   else { LGH_HIP_CHECK(hipMemcpyAsync(a.r, r, kVC * N * sizeof(double), hipMemcpyDeviceToDevice, c->stream)); }
   if (first) { LGH_HIP_CHECK(hipMemsetAsync(a.d, 0, kVC * N * sizeof(double), c->stream)); }
   else if (ord) { rc = order_gather_nodes(c, d_old, a.d, kVC); if (rc) { return rc; } }
   else { LGH_HIP_CHECK(hipMemcpyAsync(a.d, d_old, kVC * N * sizeof(double), hipMemcpyDeviceToDevice, c->stream)); }
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream)); // (vcg_set_tol_k has cleared the accumulators and the set counters)
   VcgScalars h;
   memset(&h, 0, sizeof(h));
   for (int k = 0; k < kVC; k++) { h.rz[k] = rz[k]; h.rz_prev[k] = rz_prev[k]; }
   h.first = first ? 1 : 0;
   if (a.rzl && !first)
   {
      // rz_limbs mode: the second iteration takes (r, z) of the first out of set 1 of the exact accumulators and the
      // one before (which also fixes the scale of that set) out of the scalars
      for (int k = 0; k < kVC; k++) { h.rzh[0][k] = rz_prev[k]; }
   }
   LGH_HIP_CHECK(hipMemcpy(ds, &h, sizeof(h), hipMemcpyHostToDevice));
   if (a.rzl && !first)
   {
      hipLaunchKernelGGL(vcg_test_put_rz_k, dim3(1), dim3(1), 0, c->stream, a.rzl + 1 * kLimbWords, rz[0], rz[1], rz[2], rz_prev[0], rz_prev[1], rz_prev[2]);
      LGH_HIP_CHECK(hipGetLastError());
   }
   a.iter = first ? 1 : 2;
   a.partials = c->vcg_partials + (size_t)kVC * c->vcg_stride;
   a.ticket = c->vcg_tickets + kTicketSlot;
   vcg_launch_k1(c, plan, a);
   LGH_HIP_CHECK(hipGetLastError());
   // (slab form with deferred fold: K2 would form den from the accumulators - the one-workgroup fold of the several-rank path does the same)
   if (a.den_limbs) { hipLaunchKernelGGL(vcg_fold_den_k, dim3(1), dim3(256), 0, c->stream, a); }
   LGH_HIP_CHECK(hipGetLastError());
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   LGH_HIP_CHECK(hipMemcpy(&h, ds, sizeof(h), hipMemcpyDeviceToHost));
   MergedDev md;
   if (a.settab) { rc = merged_to_device(c, md); if (rc) { return rc; } }
   struct Tmp { double *p = nullptr; ~Tmp() { (void)hipFree(p); } } tmp; // (internal order: a plane in the solve's zone order on its way to the caller's)
   if (ord) { LGH_HIP_CHECK(hipMalloc((void **)&tmp.p, nE * sizeof(double))); }
   for (int k = 0; k < kVC; k++)
   {
      den_out[k] = h.den[k];
      double *dst = ord ? tmp.p : YE_out + (size_t)k * nE;
      if (a.settab)
      {
         hipLaunchKernelGGL(vcg_unpack_merged_k, dim3((unsigned)((nE + 255) / 256)), dim3(256), 0, c->stream, a.YE + (size_t)k * a.ye_stride, md.pos, md.sec, dst, nE);
         LGH_HIP_CHECK(hipGetLastError());
      }
      else { LGH_HIP_CHECK(hipMemcpyAsync(dst, a.YE + (size_t)k * a.ye_stride, nE * sizeof(double), hipMemcpyDeviceToDevice, c->stream)); }
      if (ord)
      {
         rc = order_zone_blocks(c, tmp.p, YE_out + (size_t)k * nE, c->ND, true);
         if (rc) { return rc; }
      }
   }
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   return LGH_OK;
}

// Test hook (lgh_test_vcg_k2): ONE launch of K2 - the node kernel of the lockstep solve, the largest kernel of the
// step - as the one-rank solve issues it in iteration `it` (1: first, beta = 0; even: the launch that also updates x
// with the terms of two iterations; odd > 1: x untouched), on the caller's vectors: the E-vector K1 would have
// written (kVC planes of NE * ND), r, the old direction d and x (kVC * N each, updated in place), the scalars K1 and
// the iteration before leave: (d, A d), (r, z) after iterations it-1 and it-2, alpha of iteration it-1.  Returns
// (r, z) of the new residual as the solve would see it (ticketed fold, or the exact accumulators folded as the next
// K1 does).  One rank only.
int vcg_test_k2(lgh_ctx *c, int it, const double *YE_in, double *r, double *d, double *x, const double den[3],
                const double rz[3], const double rz_prev[3], const double alpha_prev[3], double rz_out[3], int *deferred_x)
{
   if (!vcg_supported(c)) { set_error("lgh_test_vcg_k2: no lockstep solve for kernel 0x%x", c->kid); return LGH_ERR_UNSUPPORTED; }
   if (c->multi != 0 || it < 1) { set_error("lgh_test_vcg_k2: single rank, it >= 1"); return LGH_ERR_ARG; }
   VcgPlan plan;
   int rc = vcg_prepare(c, nullptr, x, 0.0, plan);
   if (rc) { return rc; }
   VcgArgs &a = plan.a;
   const size_t N = (size_t)c->N, nE = (size_t)c->NE * c->ND;
   VcgScalars *ds = (VcgScalars *)c->vcg_s;
   const MeshOrder *ord = plan.aux->ord; // (the solve's own order: the caller's vectors and E-vector go through the permutation)
   if (ord)
   {
      rc = order_gather_nodes(c, r, a.r, kVC);
      if (rc == LGH_OK) { rc = order_gather_nodes(c, d, a.d, kVC); }
      if (rc == LGH_OK) { rc = order_gather_nodes(c, x, a.x, kVC); }
      if (rc) { return rc; }
   }
   else
   {
      LGH_HIP_CHECK(hipMemcpyAsync(a.r, r, kVC * N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      LGH_HIP_CHECK(hipMemcpyAsync(a.d, d, kVC * N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
   }
   MergedDev md;
   if (a.settab) { rc = merged_to_device(c, md); if (rc) { return rc; } }
   struct Tmp { double *p = nullptr; ~Tmp() { (void)hipFree(p); } } tmp;
   if (ord) { LGH_HIP_CHECK(hipMalloc((void **)&tmp.p, nE * sizeof(double))); }
   for (int k = 0; k < kVC; k++)
   {
      const double *src = YE_in + (size_t)k * nE;
      if (ord)
      {
         rc = order_zone_blocks(c, src, tmp.p, c->ND, false);
         if (rc) { return rc; }
         src = tmp.p;
      }
      if (a.settab)
      {
         hipLaunchKernelGGL(vcg_pack_merged_k, dim3((unsigned)((nE + 255) / 256)), dim3(256), 0, c->stream, src, md.pos, md.sec, c->ND, a.YE + (size_t)k * a.ye_stride, nE);
         LGH_HIP_CHECK(hipGetLastError());
      }
      else { LGH_HIP_CHECK(hipMemcpyAsync(a.YE + (size_t)k * a.ye_stride, src, nE * sizeof(double), hipMemcpyDeviceToDevice, c->stream)); }
   }
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream)); // (vcg_set_tol_k has cleared the accumulators)
   VcgScalars h;
   memset(&h, 0, sizeof(h));
   for (int k = 0; k < kVC; k++)
   {
      h.rz[k] = rz[k];
      h.rz_prev[k] = rz_prev[k];
      h.den[k] = den[k];
      h.alpha_last[k] = alpha_prev[k];
      h.rzh[(it - 1) & 1][k] = rz[k];
      h.rzh[it & 1][k] = rz_prev[k];
      h.alpha_hist[(it - 1) & 1][k] = alpha_prev[k];
      h.nupd[k] = it - 1;
   }
   h.first = (it == 1) ? 1 : 0;
   LGH_HIP_CHECK(hipMemcpy(ds, &h, sizeof(h), hipMemcpyHostToDevice));
   a.den_limbs = 0; // (d, A d) comes from the scalars here - the accumulators K1 fills are its own test's subject
   *deferred_x = plan.k2p ? 1 : 0;
   a.iter = it;
   a.partials = c->vcg_partials;
   a.ticket = c->vcg_tickets;
   if (plan.k2p) { vcg_launch_k2p(c, plan, a, it); }
   else if (c->t_deg <= 8) { hipLaunchKernelGGL((vcg_update_k<true, 8>), dim3(ceil_div((long)N, 256)), dim3(256), 0, c->stream, a); }
   else { set_error("lgh_test_vcg_k2: unusual valence (the unfused gather runs in the solve)"); return LGH_ERR_UNSUPPORTED; }
   LGH_HIP_CHECK(hipGetLastError());
   if (a.rzl && plan.k2p) { hipLaunchKernelGGL(vcg_rz_finish_k, dim3(1), dim3(1), 0, c->stream, ds, a.rzl, it, (const long long *)nullptr, 0, (VcgScalars *)nullptr, (unsigned long long *)nullptr, 0ull); }
   LGH_HIP_CHECK(hipGetLastError());
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   LGH_HIP_CHECK(hipMemcpy(&h, ds, sizeof(h), hipMemcpyDeviceToHost));
   for (int k = 0; k < kVC; k++) { rz_out[k] = (a.rzl && plan.k2p) ? h.rzh[it & 1][k] : h.rz[k]; }
   if (ord)
   {
      rc = order_scatter_nodes(c, a.r, r, kVC);
      if (rc == LGH_OK) { rc = order_scatter_nodes(c, a.d, d, kVC); }
      if (rc == LGH_OK) { rc = order_scatter_nodes(c, a.x, x, kVC); }
      if (rc) { return rc; }
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      return LGH_OK;
   }
   LGH_HIP_CHECK(hipMemcpy(r, a.r, kVC * N * sizeof(double), hipMemcpyDeviceToDevice));
   LGH_HIP_CHECK(hipMemcpy(d, a.d, kVC * N * sizeof(double), hipMemcpyDeviceToDevice));
   return LGH_OK;
}

} // namespace lgh
