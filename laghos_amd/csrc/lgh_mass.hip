// lgh_mass.hip — MassPAOperator action, Jacobi diagonal and the device-resident
// CG solves for gfx950.
//
// Replaces MassPAOperator::{Mult,MultFull,EliminateRHS}
// (/root/reference/laghos_assembly.cpp:80-121; the contraction itself is upstream
// MFEM MassIntegrator::AddMultPA, same math as amr/laghos_assembly.cpp:878-963),
// OperatorJacobiSmoother (laghos_solver.cpp:266-270) and the two CGSolver
// instances configured at laghos_solver.cpp:264-284 (algorithm: SURVEY §3.2).
//
// MI355X design: one CG iteration is two kernels and no host round trip.
//   K1 (element batches): beta = rz/rz_prev; the new direction d = z + beta*d is
//       formed on the fly inside the gather (not stored: elements sharing a node
//       all read the old z, d); y_e = B^T D B d_e with the z contraction in
//       registers and tables in SGPRs; den = sum_e d_e.y_e (the E-vector form of
//       (d, A d), so no extra pass over the L-vector is needed).
//   K2 (nodes): forms the same d = z + beta*d in place (one thread per node, no
//       race), z = sum of element contributions (deterministic gather over an
//       ELL-format transpose, coalesced index loads, no atomics), essential rows
//       zeroed, alpha = rz/den, x += alpha d, r -= alpha z, z = r/diag,
//       betanom = (r,z) with a wave64 shuffle reduction; the last block to
//       finish folds the block partials in fixed order, updates the device
//       scalars and the convergence flag.
// The host only enqueues iterations in chunks and looks at the flag between
// chunks; once `done` is set the remaining kernels of a chunk exit immediately.
#include "lgh_common.hpp"

namespace lgh
{

// ---------------------------------------------------------------------------
// element kernel: y_e = B^T diag(D_e) B x_e
//   MODE 0: x is an E-vector (or L2 vector), plain apply
//   MODE 1: x gathered from an L-vector through `map`
//   MODE 2: CG K1 (H1): d = z[map] + beta * d_old[map] (not stored); den
//   MODE 3: CG K1 (L2, no map): d = r + beta*d_old, stored in place (element-local)
// ---------------------------------------------------------------------------
struct MassArgs
{
   int NE;
   const double *B;   // [q + Q*d]
   const double *Dq;  // [q + NQ*e] - mass_apply_l2_plane: value(q, e) = Dq[q + dqs e] * Se[e] (mass_data)
   const double *Se;
   int dqs;
   const double *M1;  // mass_apply_l2_kron: the 1-D mass tile of the L2 basis (L x L)
   const double *w1;  // mass_apply_l2_plane<.., SEP = true>: one-dimensional weights, value(q, e) = Se[e] w1[qx] w1[qy] w1[qz] (compact data of a tensor-product rule)
   const double *x;   // MODE 0/1 input; MODE 2/3: z (H1) or r (L2)
   const int *map;    // NE*ND or null
   double *y;         // E-vector (H1) or L2 vector output
   // CG
   double *d;         // direction vector (read; MODE 3 also writes it in place)
   CgScalars *cgs;
   double *partials;
   unsigned int *ticket;
   int multi; // multi-GPU: den is all-reduced before any decision is taken on it
   int iter;  // CG iteration this launch belongs to (several ranks: see cg_pending_update)
   // mass_apply_l2_kron, fused update of the energy CG (round 6): MODE 3 with no_y does not store M d - MODE 4, the update of the
   // same iteration, forms it again from d (the L2 mass matrix is block diagonal: M d of a zone needs that zone's d only) and
   // updates ux (x) and ur (r) with it: ten vector passes per iteration become eight
   int no_y;
   double *ux, *ur;
   // lockstep with the velocity CG on several ranks (lgh_common.hpp): the rank sums of this solve's two dot products travel with
   // the velocity iteration's exchanges instead of in exchanges of their own
   LockstepWords ls;        // MODE 3: the accumulator-word sets of the iteration before, own and peers': word kLsWord = that rank's (r, r)
   int ls_on;               // 1: MODE 3 takes (r, r) from ls instead of cgs->rz; MODE 4 takes (d, M d) from ls_den_src
   double *ls_den_mirror;   // MODE 3: the local (d, M d) also goes here (VcgScalars::den_e: behind the halo messages)
   const double *ls_den_src;// MODE 4: the rank-summed (d, M d)
   long long *ls_word_out;  // MODE 4: the local (r, r), as bits, into word kLsWord of the set the exchange after K2 sends
};

// lockstep: (r, r) of the iteration before = the ranks' values in rank order (own value between the lower and the higher peers)
__device__ __forceinline__ double lockstep_rz(const LockstepWords &w)
{
   double rz = 0.0;
   for (int p = 0; p < w.before; p++) { const double v = __longlong_as_double(w.peers[(size_t)p * kLsWordsPerSet + kLsWord]); rz = (p == 0) ? v : rz + v; }
   {
      const double v = __longlong_as_double(w.own[kLsWord]);
      rz = (w.before == 0) ? v : rz + v;
   }
   for (int p = w.before; p < w.n_peers; p++) { rz += __longlong_as_double(w.peers[(size_t)p * kLsWordsPerSet + kLsWord]); }
   return rz;
}
// ... and what the next kernel of the sequence does with it (cg_pending_update with the sum formed here): convergence looked at,
// the outcome committed by one thread; returns true when the solve has converged
__device__ __forceinline__ bool lockstep_pending_update(CgScalars *s, const int iter, const double rz, const bool commit)
{
   const bool conv = rz < 0.0 || rz <= s->r0;
   if (commit)
   {
      __hip_atomic_store(&s->iters, iter - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (conv)
      {
         __hip_atomic_store(&s->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         __hip_atomic_store(&s->rz, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         __hip_atomic_store(&s->den, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      else { __hip_atomic_store(&s->rz, rz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } // (the update of this iteration divides it by (d, M d))
   }
   return conv;
}

// Several ranks: the sums of den and (r, z) over the ranks complete outside the kernels that produce the
// local parts, so the decisions they feed (breakdown, convergence, iteration count) are taken by the NEXT
// kernel of the sequence instead of a single-thread kernel after every exchange: each workgroup evaluates the
// same predicate on the same reduced values, and thread 0 of workgroup 0 also commits the outcome to the
// state.  A commit only writes values under which the predicate stays true (done = 1, rz = den = 0: sums of
// zeros stay zero in the exchanges of launches enqueued past convergence), so a thread that reads the
// state after the commit decides like one that read it before.
// K1 of iteration iter >= 2: outcome of the update of iteration iter - 1.
__device__ __forceinline__ bool cg_pending_update(CgScalars *s, const int iter, const bool commit)
{
   const double rz = s->rz;
   const bool conv = rz < 0.0 || rz <= s->r0;
   if (commit)
   {
      // write-through stores: another XCD's L2 may hold the same line dirty (K1's last workgroup writes den / first)
      __hip_atomic_store(&s->iters, iter - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (conv)
      {
         __hip_atomic_store(&s->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         __hip_atomic_store(&s->rz, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         __hip_atomic_store(&s->den, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
   }
   return conv;
}
// K2: breakdown (den == 0 after the sum over the ranks), as upstream
__device__ __forceinline__ bool cg_pending_den(CgScalars *s, const bool commit)
{
   const bool brk = s->den == 0.0;
   if (brk && commit)
   {
      __hip_atomic_store(&s->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&s->rz, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   return brk;
}

template <int D, int Q, int NEB, int MODE>
__global__ void __launch_bounds__(Q *Q *NEB)
mass_apply_3d(const MassArgs a)
{
   constexpr int NQ = Q * Q * Q, ND = D * D * D;
   constexpr int SX = (ND > D * Q * Q) ? ND : D * Q * Q; // X then C[dz][qy][qx]
   constexpr int SA = D * D * Q;                         // A[dz][dy][qx] then E[dz][qy][dx]
   constexpr int SAE = (SA > D * Q * D) ? SA : D * Q * D;
   constexpr int PER = SX + SAE + 1;
   __shared__ double smem[NEB * PER];
   __shared__ double red[16];

   const int tid = threadIdx.x;
   const int tx = tid % Q, ty = (tid / Q) % Q, eb = tid / (Q * Q);
   const int nblocks = gridDim.x;
   const int blk = (MODE == 2 || MODE == 1) ? xcd_swizzle(blockIdx.x, nblocks) : blockIdx.x;
   const int e0 = blk * NEB;
   const int e = e0 + eb;
   const bool active = (e < a.NE);
   double *sX = smem + eb * PER;
   double *sA = sX + SX;

   double beta = 0.0;
   bool first = false;
   if (MODE >= 2)
   {
      if (a.cgs->done) { return; }
      first = a.cgs->first != 0;
      if (a.multi && !first && cg_pending_update(a.cgs, a.iter, blockIdx.x == 0 && threadIdx.x == 0)) { return; }
      beta = first ? 0.0 : a.cgs->rz / a.cgs->rz_prev;
   }

   // ---- cooperative load of the block's element dofs
   {
      const int nthr = Q * Q * NEB;
      const int nel = min(NEB, a.NE - e0);
      for (int i = tid; i < nel * ND; i += nthr)
      {
         const int el = i / ND, d = i - el * ND;
         const size_t p = (size_t)(e0 + el) * ND + d;
         double val;
         if (MODE == 0) { val = a.x[p]; }
         else if (MODE == 1) { val = a.x[a.map[p]]; }
         else if (MODE == 2)
         {
            const int n = a.map[p];
            val = a.x[n]; // K2 stores the same value later
            if (!first) { val += beta * a.d[n]; }
         }
         else
         {
            val = a.x[p];
            if (!first) { val += beta * a.d[p]; }
            a.d[p] = val;
         }
         smem[el * PER + d] = val;
      }
   }
   double bx[D], by[D];
#pragma unroll
   for (int d = 0; d < D; d++)
   {
      bx[d] = a.B[tx + Q * d];
      by[d] = a.B[ty + Q * d];
   }
   // quadrature data of this column, issued early
   double dq[Q];
   if (active)
   {
      const double *p = a.Dq + (size_t)e * NQ + tx + Q * ty;
#pragma unroll
      for (int qz = 0; qz < Q; qz++) { dq[qz] = p[Q * Q * qz]; }
   }
   else
   {
#pragma unroll
      for (int qz = 0; qz < Q; qz++) { dq[qz] = 0.0; }
   }
   __syncthreads();

   // E-level dot needs x_e for the (dx,dy) threads: keep the column in registers
   double xcol[D];
   if (MODE >= 2)
   {
#pragma unroll
      for (int dz = 0; dz < D; dz++) { xcol[dz] = (tx < D && ty < D) ? sX[tx + D * (ty + D * dz)] : 0.0; }
   }

   // forward x: thread (qx = tx, dy = ty < D)
   if (ty < D)
   {
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
         double u = 0.0;
#pragma unroll
         for (int dx = 0; dx < D; dx++) { u += bx[dx] * sX[dx + D * (ty + D * dz)]; }
         sA[tx + Q * (ty + D * dz)] = u;
      }
   }
   __syncthreads();
   // forward y (registers per dz), forward z, scale, backward z
   double r[D];
   {
      double bb[D];
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
         double u = 0.0;
#pragma unroll
         for (int dy = 0; dy < D; dy++) { u += by[dy] * sA[tx + Q * (dy + D * dz)]; }
         bb[dz] = u;
      }
      double qv[Q];
#pragma unroll
      for (int qz = 0; qz < Q; qz++)
      {
         double u = 0.0;
#pragma unroll
         for (int dz = 0; dz < D; dz++) { u += a.B[qz + Q * dz] * bb[dz]; }
         qv[qz] = u * dq[qz];
      }
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
         double u = 0.0;
#pragma unroll
         for (int qz = 0; qz < Q; qz++) { u += a.B[qz + Q * dz] * qv[qz]; }
         r[dz] = u;
      }
   }
   // sX is dead (read before the previous barrier): reuse as C[dz][qy][qx]
#pragma unroll
   for (int dz = 0; dz < D; dz++) { sX[tx + Q * (ty + Q * dz)] = r[dz]; }
   __syncthreads();
   // backward x: thread (dx = tx < D, qy = ty)
   if (tx < D)
   {
      double brow[Q];
#pragma unroll
      for (int q = 0; q < Q; q++) { brow[q] = a.B[q + Q * tx]; }
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
         double u = 0.0;
#pragma unroll
         for (int qx = 0; qx < Q; qx++) { u += brow[qx] * sX[qx + Q * (ty + Q * dz)]; }
         sA[tx + D * (ty + Q * dz)] = u;
      }
   }
   __syncthreads();
   // backward y: thread (dx = tx < D, dy = ty < D)
   double dot = 0.0;
   if (tx < D && ty < D && active)
   {
      double brow[Q];
#pragma unroll
      for (int q = 0; q < Q; q++) { brow[q] = a.B[q + Q * ty]; }
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
         double u = 0.0;
#pragma unroll
         for (int qy = 0; qy < Q; qy++) { u += brow[qy] * sA[tx + D * (qy + Q * dz)]; }
         const size_t po = tx + D * (ty + D * dz) + (size_t)ND * e;
         a.y[po] = u;
         if (MODE >= 2) { dot += xcol[dz] * u; }
      }
   }
   if (MODE >= 2)
   {
      const double bsum = block_sum(dot, red);
      double total;
      if (grid_sum_last_block(bsum, a.partials, a.ticket, red, total))
      {
         if (tid == 0)
         {
            a.cgs->den = total;
            if (a.cgs->first) { a.cgs->first = 0; }
            if (total == 0.0 && !a.multi) { a.cgs->done = 1; } // breakdown (den == 0): stop, as upstream
         }
      }
   }
}

// ---------------------------------------------------------------------------
// L2 mass apply for the high orders (Q1D >= 8: Q4Q3, Q5Q4), plane-per-thread form.
// The column form above keeps Q*Q threads per element and the quadrature data of one z-column per thread:
// at Q1D = 10 that is one wave per SIMD (NEB = 2 elements per workgroup) and the kernel reaches 1.7 TB/s,
// and the unpreconditioned energy CG (237 iterations per solve at Q5Q4) makes it the dominant kernel of
// BASELINE config 5.  Here a thread owns one x-index qx of an element and keeps (y, z) data of that index in
// registers (as vcg_apply_plane does for the H1 solve): only the two x contractions go through LDS, and
// 256 / (Q HY) elements share a workgroup, two workgroups per CU.
// Register budget: the plane w[qy][dz] (Q*L doubles) AND the 1-D table (Q*L doubles - too large for the
// scalar registers at 50 entries, and re-reading it from LDS per use would make the kernel LDS-issue bound)
// both want registers.  At Q = 10 that is 2 x 100 VGPRs and does not fit, so the qy rows of a plane are split
// over HY = 2 threads: thread (qx, h) runs the rows qy in [h Q/HY, (h+1) Q/HY) through the z stage (the rows are
// independent there) and contributes a partial sum to the backward y contraction; the HY partial planes are
// summed by the backward x contraction, which reads them from LDS anyway.
// The quadrature data is read straight from memory, one qy row of Q values ahead of its use (each value is
// needed by exactly one thread: LDS staging would only add traffic); rows of Q consecutive doubles.
// MODE 0: y = M x.  MODE 3: CG K1 of the L2 solve (d = r + beta d stored in place, den = (d, M d)).
// value of the other lane of an adjacent pair (quad_perm [1,0,3,2], as lane_pair_swap of lgh_vcg.hip)
__device__ __forceinline__ double pair_swap(const double v)
{
   int lo = __double2loint(v), hi = __double2hiint(v);
   lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
   hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
   return __hiloint2double(hi, lo);
}

template <int L, int Q, int HY, int NEB, int MODE, bool SEP>
__global__ void __launch_bounds__(Q *HY *NEB, 2)
mass_apply_l2_plane(const MassArgs a)
{
   constexpr int NQ = Q * Q * Q, NL = L * L * L, LL = L * L;
   constexpr int TE = Q * HY, NT = TE * NEB, QH = Q / HY;
   static_assert(Q % HY == 0, "qy rows split evenly");
   constexpr int CS = NL | 1;            // odd strides: the broadcast reads of different elements hit different banks
   static_assert(HY == 1 || HY == 2, "one lane or a pair of adjacent lanes per plane");
   constexpr int CE = (LL * Q) | 1; // [dy,dz][qx]: the two lanes of a pair (HY = 2) hand over one summed plane (lane_pair_swap)
   __shared__ double sIn[NEB * CS], sE[NEB * CE], sB[Q * L];
   __shared__ double red[16];
   const int tid = threadIdx.x;
   const int eb = tid / TE, lt = tid - eb * TE;
   const int qx = lt / HY, h = lt - qx * HY; // (the lanes of a pair are adjacent: TE is even, so h is the parity of the lane)
   const int e0 = blockIdx.x * NEB, e = e0 + eb;
   const bool active = (e < a.NE);
   double beta = 0.0;
   bool first = false;
   if (MODE == 3)
   {
      if (a.cgs->done) { return; }
      first = a.cgs->first != 0;
      if (a.multi && !first && cg_pending_update(a.cgs, a.iter, blockIdx.x == 0 && threadIdx.x == 0)) { return; }
      beta = first ? 0.0 : a.cgs->rz / a.cgs->rz_prev;
   }
   for (int i = tid; i < Q * L; i += NT) { sB[i] = a.B[i]; }
   {
      const int nel = min(NEB, a.NE - e0);
      for (int i = tid; i < nel * NL; i += NT)
      {
         const int el = i / NL, d = i - el * NL;
         const size_t p = (size_t)e0 * NL + i;
         double val = a.x[p];
         if (MODE == 3)
         {
            if (!first) { val += beta * a.d[p]; }
            a.d[p] = val;
         }
         sIn[el * CS + d] = val;
      }
   }
   // first row of this thread's quadrature data, in flight across the barrier
   const double *Dp = a.Dq + (size_t)(active ? e : 0) * a.dqs + qx + Q * (h * QH);
   const double se = a.Se[active ? e : 0];
   double dq[SEP ? 1 : Q];
   if (!SEP)
   {
#pragma unroll
      for (int qz = 0; qz < Q; qz++) { dq[qz] = Dp[Q * Q * qz] * se; }
   }
   // SEP: value(q, e) = se w[qx] w[qy] w[qz] - the weights in scalar registers, no loads and no row held ahead
   double ws[SEP ? (Q + 1) / 2 : 1]; // (mirror symmetric, checked by lgh_create: w[q] = w[Q - 1 - q])
   auto wq = [&](const int q) -> double { return ws[SEP ? (q < (Q + 1) / 2 ? q : Q - 1 - q) : 0]; };
   double wxe = 0.0;
   if (SEP)
   {
#pragma unroll
      for (int q = 0; q < (Q + 1) / 2; q++) { ws[q] = uniform_f64(a.w1[q]); }
      wxe = a.w1[qx] * se;
   }
   __syncthreads();
   const double *sI = sIn + eb * CS;
   // the 1-D table in scalar registers, in half: the Bernstein basis at Gauss-Legendre points is mirror symmetric,
   // B[q,l] = B[Q-1-q, L-1-l] (checked by lgh_create; the column form runs otherwise)
   constexpr int QL = Q * L, HB = (QL + 1) / 2;
   double Bh[HB];
#pragma unroll
   for (int i = 0; i < HB; i++) { Bh[i] = uniform_f64(a.B[i]); }
   auto Bt = [&](const int idx) -> double { return idx < HB ? Bh[idx] : Bh[QL - 1 - idx]; };
   // forward x: t[dy,dz] = sum_dx B[qx,dx] d[dx,dy,dz]   (the HY threads of a plane each form it)
   double t[LL];
   {
      double bx[L];
#pragma unroll
      for (int dx = 0; dx < L; dx++) { bx[dx] = sB[qx + Q * dx]; }
#pragma unroll
      for (int k = 0; k < LL; k++)
      {
         double u = 0.0;
#pragma unroll
         for (int dx = 0; dx < L; dx++) { u = fma(bx[dx], sI[dx + L * k], u); }
         t[k] = u;
      }
   }
   // this thread's rows: forward y, then per row forward z, scaling, backward z; partial backward y on the fly
   double acc[LL]; // [dy + L*dz]: sum over this thread's qy rows of B[qy,dy] w[qy][dz]
#pragma unroll
   for (int k = 0; k < LL; k++) { acc[k] = 0.0; }
#pragma unroll
   for (int r = 0; r < QH; r++)
   {
      double dn[SEP ? 1 : Q];
      if (!SEP && r + 1 < QH)
      {
#pragma unroll
         for (int qz = 0; qz < Q; qz++) { dn[qz] = Dp[Q * (r + 1) + Q * Q * qz] * se; }
      }
      double wrow = 0.0; // SEP: se w[qx] w[qy] of this row
      if (SEP)
      {
         double wy = wq(r);
#pragma unroll
         for (int hh = 1; hh < HY; hh++) { wy = (h == hh) ? wq(hh * QH + r) : wy; }
         wrow = wxe * wy;
      }
      // the table row of qy = h*QH + r (h differs between lanes: select, not index)
      double by[L];
#pragma unroll
      for (int dy = 0; dy < L; dy++)
      {
         double v = Bt(r + Q * dy);
#pragma unroll
         for (int hh = 1; hh < HY; hh++) { v = (h == hh) ? Bt(hh * QH + r + Q * dy) : v; }
         by[dy] = v;
      }
      double wr[L];
#pragma unroll
      for (int dz = 0; dz < L; dz++)
      {
         double u = 0.0;
#pragma unroll
         for (int dy = 0; dy < L; dy++) { u = fma(by[dy], t[dy + L * dz], u); }
         wr[dz] = u;
      }
      double cz[Q];
#pragma unroll
      for (int qz = 0; qz < Q; qz++)
      {
         double u = 0.0;
#pragma unroll
         for (int dz = 0; dz < L; dz++) { u = fma(Bt(qz + Q * dz), wr[dz], u); }
         cz[qz] = SEP ? (u * wrow) * wq(qz) : u * dq[SEP ? 0 : qz];
      }
#pragma unroll
      for (int dz = 0; dz < L; dz++)
      {
         double u = 0.0;
#pragma unroll
         for (int qz = 0; qz < Q; qz++) { u = fma(Bt(qz + Q * dz), cz[qz], u); }
#pragma unroll
         for (int dy = 0; dy < L; dy++) { acc[dy + L * dz] = fma(by[dy], u, acc[dy + L * dz]); }
      }
      if (!SEP && r + 1 < QH)
      {
#pragma unroll
         for (int qz = 0; qz < Q; qz++) { dq[qz] = dn[qz]; }
      }
   }
   // hand the plane over; HY = 2: lane h of a pair delivers the (dy,dz) entries k = h (mod 2), summed over both lanes
   double *sEe = sE + eb * CE;
   if (HY == 1)
   {
#pragma unroll
      for (int k = 0; k < LL; k++) { sEe[qx + Q * k] = acc[k]; }
   }
   else
   {
#pragma unroll
      for (int k = 0; k + 1 < LL; k += 2)
      {
         const double give = h ? acc[k] : acc[k + 1];
         const double keep = h ? acc[k + 1] : acc[k];
         sEe[qx + Q * (k + h)] = keep + pair_swap(give);
      }
      if (LL & 1)
      {
         const double tot = acc[LL - 1] + pair_swap(acc[LL - 1]);
         if (h == 0) { sEe[qx + Q * (LL - 1)] = tot; }
      }
   }
   __syncthreads();
   // backward x: thread (lx = qx < L, h) sums the Q planes of both halves for its share of the (dy,dz) pairs
   double dot = 0.0;
   if (qx < L && active)
   {
      const double *sEa = sE + eb * CE;
      double bt[Q];
#pragma unroll
      for (int q = 0; q < Q; q++) { bt[q] = sB[q + Q * qx]; }
      constexpr int KH = (LL + HY - 1) / HY;
#pragma unroll
      for (int kk = 0; kk < KH; kk++)
      {
         const int k = h * KH + kk;
         if (k < LL)
         {
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < Q; q++) { u = fma(bt[q], sEa[q + Q * k], u); }
            a.y[(size_t)e * NL + qx + L * k] = u;
            if (MODE == 3) { dot = fma(sI[qx + L * k], u, dot); }
         }
      }
   }
   if (MODE == 3)
   {
      const double bsum = block_sum(dot, red);
      double total;
      if (grid_sum_last_block(bsum, a.partials, a.ticket, red, total))
      {
         if (tid == 0)
         {
            a.cgs->den = total;
            if (a.cgs->first) { a.cgs->first = 0; }
            if (total == 0.0 && !a.multi) { a.cgs->done = 1; } // breakdown (den == 0): stop, as upstream
         }
      }
   }
}

// 2D: one thread per quadrature point; same MODE semantics.
template <int D, int Q, int NEB, int MODE>
__global__ void __launch_bounds__(Q *Q *NEB)
mass_apply_2d(const MassArgs a)
{
   constexpr int NQ = Q * Q, ND = D * D;
   constexpr int PER = ND + D * Q + NQ + 1;
   __shared__ double smem[NEB * PER];
   __shared__ double red[16];
   const int tid = threadIdx.x;
   const int tx = tid % Q, ty = (tid / Q) % Q, eb = tid / (Q * Q);
   const int e0 = blockIdx.x * NEB;
   const int e = e0 + eb;
   const bool active = (e < a.NE);
   double *sX = smem + eb * PER; // [dy][dx]
   double *sA = sX + ND;         // [dy][qx] / [qy][dx]
   double *sQ = sA + D * Q;      // [qy][qx]
   double beta = 0.0;
   bool first = false;
   if (MODE >= 2)
   {
      if (a.cgs->done) { return; }
      first = a.cgs->first != 0;
      if (a.multi && !first && cg_pending_update(a.cgs, a.iter, blockIdx.x == 0 && threadIdx.x == 0)) { return; }
      beta = first ? 0.0 : a.cgs->rz / a.cgs->rz_prev;
   }
   {
      const int nthr = Q * Q * NEB;
      const int nel = min(NEB, a.NE - e0);
      for (int i = tid; i < nel * ND; i += nthr)
      {
         const int el = i / ND, d = i - el * ND;
         const size_t p = (size_t)(e0 + el) * ND + d;
         double val;
         if (MODE == 0) { val = a.x[p]; }
         else if (MODE == 1) { val = a.x[a.map[p]]; }
         else if (MODE == 2)
         {
            const int n = a.map[p];
            val = a.x[n];
            if (!first) { val += beta * a.d[n]; }
         }
         else
         {
            val = a.x[p];
            if (!first) { val += beta * a.d[p]; }
            a.d[p] = val;
         }
         smem[el * PER + d] = val;
      }
   }
   __syncthreads();
   if (ty < D)
   {
      double u = 0.0;
      for (int dx = 0; dx < D; dx++) { u += a.B[tx + Q * dx] * sX[dx + D * ty]; }
      sA[tx + Q * ty] = u;
   }
   __syncthreads();
   {
      double u = 0.0;
      for (int dy = 0; dy < D; dy++) { u += a.B[ty + Q * dy] * sA[tx + Q * dy]; }
      sQ[tx + Q * ty] = active ? u * a.Dq[(size_t)e * NQ + tx + Q * ty] : 0.0;
   }
   __syncthreads();
   if (tx < D)
   {
      double u = 0.0;
      for (int qx = 0; qx < Q; qx++) { u += a.B[qx + Q * tx] * sQ[qx + Q * ty]; }
      sA[tx + D * ty] = u;
   }
   __syncthreads();
   double dot = 0.0;
   if (tx < D && ty < D && active)
   {
      double u = 0.0;
      for (int qy = 0; qy < Q; qy++) { u += a.B[qy + Q * ty] * sA[tx + D * qy]; }
      const size_t po = tx + D * ty + (size_t)ND * e;
      a.y[po] = u;
      if (MODE >= 2) { dot = sX[tx + D * ty] * u; }
   }
   if (MODE >= 2)
   {
      const double bsum = block_sum(dot, red);
      double total;
      if (grid_sum_last_block(bsum, a.partials, a.ticket, red, total))
      {
         if (tid == 0)
         {
            a.cgs->den = total;
            if (a.cgs->first) { a.cgs->first = 0; }
            if (total == 0.0 && !a.multi) { a.cgs->done = 1; }
         }
      }
   }
}

static int unknown_kernel(int id)
{
   set_error("Unknown kernel 0x%x", id);
   return LGH_ERR_UNSUPPORTED;
}

// ---- L2 mass apply, Kronecker form (3D; compact mass data on a tensor-product rule: lgh_create's 1-D mass tile M1l)
// B^T diag(s_e w (x) w (x) w) B = s_e M1 (x) M1 (x) M1 with M1 = B^T diag(w) B (L x L): three contractions of L on the
// element's L^3 dofs - 3 L^4 FMAs instead of ~2 (L^3 Q + L^2 Q^2 + L Q^3) + Q^3 through the quadrature points (L = 5,
// Q = 10: 1 875 against 18 500) and no quadrature-point values at all: the kernel moves its vectors and nothing else.
// Same operator in exact arithmetic (the compact form itself is accepted to 1e-12, mass_data); rounding differs.
// MODE 0: y = M x.  MODE 3: CG K1 of the L2 solve (d = r + beta d stored in place, den = (d, M d)), as the plane form; with
// a.no_y the product itself is not stored.  MODE 4 (round 6): the UPDATE of the same iteration, for the launch after it -
// M d is formed again from the stored d (same code, same bits) and x += alpha d, r -= alpha M d, (r, r) follow with the
// expressions, the scalars and the convergence logic of cg_update_k: the product never travels through memory.
// One workgroup of 256 threads takes NEB consecutive elements (their dofs are one contiguous run of the L2 vector);
// a thread owns one row of L values per stage: x rows (e, lz, ly), y rows (e, lz, lx), z rows (e, ly, lx).
template <int L, int NEB, int MODE>
__global__ void __launch_bounds__(256)
mass_apply_l2_kron(const MassArgs a)
{
   constexpr int NL = L * L * L, LL = L * L, NT = 256;
   __shared__ double sIn[NEB * NL], sT1[NEB * NL], sT2[NEB * NL];
   __shared__ double red[16];
   const int tid = threadIdx.x;
   const int e0 = blockIdx.x * NEB;
   const int nel = min(NEB, a.NE - e0);
   double beta = 0.0, alpha = 0.0;
   bool first = false;
   if (MODE == 3)
   {
      if (a.cgs->done) { return; }
      first = a.cgs->first != 0;
      if (a.ls_on && !first)
      {
         const double rz = lockstep_rz(a.ls);
         if (lockstep_pending_update(a.cgs, a.iter, rz, blockIdx.x == 0 && threadIdx.x == 0)) { return; }
         beta = rz / a.cgs->rz_prev;
      }
      else
      {
         if (a.multi && !first && cg_pending_update(a.cgs, a.iter, blockIdx.x == 0 && threadIdx.x == 0)) { return; }
         beta = first ? 0.0 : a.cgs->rz / a.cgs->rz_prev;
      }
   }
   if (MODE == 4)
   {
      if (a.cgs->done) { return; }
      if (a.ls_on)
      {
         const double den = *a.ls_den_src; // (summed over the ranks by the halo exchange of this iteration)
         if (den == 0.0) // breakdown, as cg_pending_den
         {
            if (blockIdx.x == 0 && threadIdx.x == 0)
            {
               __hip_atomic_store(&a.cgs->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               __hip_atomic_store(&a.cgs->rz, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
         }
         alpha = a.cgs->rz / den;
      }
      else
      {
         if (a.multi && cg_pending_den(a.cgs, blockIdx.x == 0 && threadIdx.x == 0)) { return; } // (as cg_update_k)
         alpha = a.cgs->rz / a.cgs->den;
      }
   }
   double M[LL]; // M[i + L j], symmetric, in scalar registers
#pragma unroll
   for (int i = 0; i < LL; i++) { M[i] = uniform_f64(a.M1[i]); }
   // MODE 4: the r and x values of the places this thread updates in the last stage, asked for now
   constexpr int ZR = (NEB * LL + NT - 1) / NT; // rows (e, ly, lx) per thread in the z stage
   double ro[MODE == 4 ? ZR : 1][L], xo[MODE == 4 ? ZR : 1][L];
   if (MODE == 4)
   {
#pragma unroll
      for (int k = 0; k < ZR; k++)
      {
         const int r = tid + k * NT;
         const int yx = r % LL, el = r / LL;
         const bool ok = r < nel * LL;
#pragma unroll
         for (int i = 0; i < L; i++)
         {
            const size_t p = (size_t)e0 * NL + (ok ? el * NL + yx + LL * i : 0);
            ro[k][i] = a.ur[p];
            xo[k][i] = a.ux[p];
         }
      }
   }
   for (int i = tid; i < nel * NL; i += NT)
   {
      const size_t p = (size_t)e0 * NL + i;
      double val = (MODE == 4) ? a.d[p] : a.x[p];
      if (MODE == 3)
      {
         if (!first) { val += beta * a.d[p]; }
         a.d[p] = val;
      }
      sIn[i] = val;
   }
   __syncthreads();
   // x: rows of L consecutive values
   for (int r = tid; r < nel * LL; r += NT)
   {
      double u[L];
#pragma unroll
      for (int j = 0; j < L; j++) { u[j] = sIn[r * L + j]; }
#pragma unroll
      for (int i = 0; i < L; i++)
      {
         double s = M[i] * u[0];
#pragma unroll
         for (int j = 1; j < L; j++) { s = fma(M[i + L * j], u[j], s); }
         sT1[r * L + i] = s;
      }
   }
   __syncthreads();
   // y: rows (e, lz, lx), stride L
   for (int r = tid; r < nel * LL; r += NT)
   {
      const int lx = r % L, ez = r / L; // ez = lz + L e
      const int base = ez * LL + lx;
      double u[L];
#pragma unroll
      for (int j = 0; j < L; j++) { u[j] = sT1[base + L * j]; }
#pragma unroll
      for (int i = 0; i < L; i++)
      {
         double s = M[i] * u[0];
#pragma unroll
         for (int j = 1; j < L; j++) { s = fma(M[i + L * j], u[j], s); }
         sT2[base + L * i] = s;
      }
   }
   __syncthreads();
   // z: rows (e, ly, lx), stride L^2; the element factor; out (and the partial of (d, M d))
   double dot = 0.0;
#pragma unroll
   for (int k = 0; k < ZR; k++)
   {
      const int r = tid + k * NT;
      if (r >= nel * LL) { break; }
      const int yx = r % LL, el = r / LL;
      const int base = el * NL + yx;
      const double se = a.Se[e0 + el];
      double u[L];
#pragma unroll
      for (int j = 0; j < L; j++) { u[j] = sT2[base + LL * j]; }
#pragma unroll
      for (int i = 0; i < L; i++)
      {
         double s = M[i] * u[0];
#pragma unroll
         for (int j = 1; j < L; j++) { s = fma(M[i + L * j], u[j], s); }
         s *= se;
         if (MODE == 4)
         {
            // cg_update_k's expressions: x += alpha d, r -= alpha (M d), (r, r)
            const double dv = sIn[base + LL * i];
            const double xv = xo[MODE == 4 ? k : 0][i] + alpha * dv;
            const double rv = ro[MODE == 4 ? k : 0][i] - alpha * s;
            a.ux[(size_t)e0 * NL + base + LL * i] = xv;
            a.ur[(size_t)e0 * NL + base + LL * i] = rv;
            dot += rv * rv;
            continue;
         }
         if (!(MODE == 3 && a.no_y)) { a.y[(size_t)e0 * NL + base + LL * i] = s; }
         if (MODE == 3) { dot = fma(sIn[base + LL * i], s, dot); }
      }
   }
   if (MODE == 4)
   {
      const double bsum = block_sum(dot, red);
      double total;
      if (grid_sum_last_block(bsum, a.partials, a.ticket, red, total))
      {
         if (tid == 0)
         {
            CgScalars *sc = a.cgs;
            sc->rz_prev = sc->rz;
            sc->rz = total; // betanom
            if (a.ls_on) { a.ls_word_out[kLsWord] = __double_as_longlong(total); } // (its sum over the ranks: the next apply, or l2_lockstep_fold)
            if (!a.multi)
            {
               sc->iters = a.iter;
               if (total < 0.0 || total <= sc->r0) { sc->done = 1; }
            }
         }
      }
   }
   if (MODE == 3)
   {
      const double bsum = block_sum(dot, red);
      double total;
      if (grid_sum_last_block(bsum, a.partials, a.ticket, red, total))
      {
         if (tid == 0)
         {
            a.cgs->den = total;
            if (a.ls_on && a.ls_den_mirror) { *a.ls_den_mirror = total; } // (lockstep: behind the halo messages of the velocity iteration)
            if (a.cgs->first) { a.cgs->first = 0; }
            if (total == 0.0 && !a.multi) { a.cgs->done = 1; } // breakdown (den == 0): stop, as upstream
         }
      }
   }
}

template <int Q> constexpr int neb_for() { return (256 / (Q * Q)) > 0 ? (256 / (Q * Q)) : 1; }
static bool l2_one_round(const lgh_ctx *c)
{
   const char *e = getenv("LGH_L2_NEB");
   return c->L1D >= 5 && !(e && e[0] == '0');
}

// Which kernel the L2 mass apply (modes 0 / 3: the energy CG) launches, decided in ONE place for the dispatch below
// and for lgh_l2_mass_form(): 2 = mass_apply_l2_kron (needs compact data on a tensor-product rule), 1 = the plane form,
// 0 = the column form.  `compact`: whether the kernel reads one factor per element instead of the NQ-entry table.
static int l2_mass_kernel(lgh_ctx *c, int *form, int *compact)
{
   const int id = (c->dim << 8) | (c->L1D << 4) | c->Q1D;
   const double *Dq, *Se;
   int dqs = 1;
   *form = 0;
   *compact = 0;
   const bool kron_ok = c->dim == 3 && c->M1l && c->w1d && c->L1D <= 5;
   const char *penv = getenv("LGH_L2_PLANE"); // A/B: 0 = column form
   const bool plane_ok = c->b_l2_sym && (id == 0x336 || id == 0x348 || id == 0x35A) && !(penv && penv[0] == '0');
   if (kron_ok || plane_ok)
   {
      const int rc = mass_data(c, &Dq, &dqs, &Se);
      if (rc) { return rc; }
   }
   if (kron_ok && dqs == 0) { *form = 2; *compact = 1; }
   else if (plane_ok) { *form = 1; *compact = (dqs == 0); }
   return LGH_OK;
}
int l2_mass_form(lgh_ctx *c, int *form, int *compact) { return l2_mass_kernel(c, form, compact); }

template <int MODE> static int launch_mass(lgh_ctx *c, int space, const MassArgs &a0)
{
   const MassArgs &a = a0;
   const int n1d = (space == LGH_SPACE_H1) ? c->D1D : c->L1D;
   const int id = (c->dim << 8) | (n1d << 4) | c->Q1D;
   int l2form = 0, l2compact = 0;
   if ((MODE == 0 || MODE == 3) && space == LGH_SPACE_L2)
   {
      const int rc_f = l2_mass_kernel(c, &l2form, &l2compact);
      if (rc_f) { return rc_f; }
   }
#define LGH_MASS_CASE(DIMK, D_, Q_)                                                           \
   {                                                                                          \
      constexpr int NEB_ = neb_for<Q_>();                                                     \
      hipLaunchKernelGGL((DIMK<D_, Q_, NEB_, MODE>), dim3(ceil_div(c->NE, NEB_)),             \
                         dim3(Q_ * Q_ * NEB_), 0, c->stream, a);                              \
   }                                                                                          \
   break
   if (l2form == 2)
   {
      // compact mass data on a tensor-product rule: the Kronecker form (LGH_MASS_KRON=0: no tiles, see lgh_create)
      MassArgs a = a0;
      const int rc_md = mass_data(c, &a.Dq, &a.dqs, &a.Se);
      if (rc_md) { return rc_md; }
      {
         constexpr int M = (MODE == 3 ? 3 : 0);
         a.M1 = c->M1l;
#define LGH_L2K(L_, NEB_) hipLaunchKernelGGL((mass_apply_l2_kron<L_, NEB_, M>), dim3(ceil_div(c->NE, NEB_)), dim3(256), 0, c->stream, a)
         // zones per workgroup.  L = 5 (round 6): ten, so that every thread has ONE row of a stage (25 rows per zone) and five
         // workgroups fit a CU's LDS, instead of sixteen (400 rows: a second round with 144 of 256 threads, three workgroups per
         // CU): 53 -> 46 us per apply at config 5, the leg 501 -> 534 (profiles/r6_l2_neb.txt); at L = 3 the same choice (28
         // zones instead of 64) changes nothing.  LGH_L2_NEB=0: sixteen.
         if (l2_one_round(c)) { LGH_L2K(5, 10); }
         else
         {
            switch (c->L1D)
            {
               case 1: LGH_L2K(1, 256); break;
               case 2: LGH_L2K(2, 128); break;
               case 3: LGH_L2K(3, 64); break;
               case 4: LGH_L2K(4, 32); break;
               default: LGH_L2K(5, 16); break;
            }
         }
#undef LGH_L2K
         LGH_HIP_CHECK(hipGetLastError());
         return LGH_OK;
      }
   }
   if (l2form == 1)
   {
      constexpr int M = (MODE == 3 ? 3 : 0);
      {
         MassArgs a = a0; // (the plane form takes the mass data in its compact form where it has one)
         const int rc_md = mass_data(c, &a.Dq, &a.dqs, &a.Se);
         if (rc_md) { return rc_md; }
         a.w1 = (a.dqs == 0) ? c->w1d : nullptr;
#define LGH_L2P(L_, Q_, HY_, NEB_) do { if (a.w1) { hipLaunchKernelGGL((mass_apply_l2_plane<L_, Q_, HY_, NEB_, M, true>), dim3(ceil_div(c->NE, NEB_)), dim3(Q_ * HY_ * NEB_), 0, c->stream, a); } \
                                        else { hipLaunchKernelGGL((mass_apply_l2_plane<L_, Q_, HY_, NEB_, M, false>), dim3(ceil_div(c->NE, NEB_)), dim3(Q_ * HY_ * NEB_), 0, c->stream, a); } } while (0)
         if (id == 0x336) { LGH_L2P(3, 6, 1, 42); }
         else if (id == 0x348) { LGH_L2P(4, 8, 1, 32); }
         else { LGH_L2P(5, 10, 2, 12); }
#undef LGH_L2P
         LGH_HIP_CHECK(hipGetLastError());
         return LGH_OK;
      }
   }
   switch (id)
   {
      case 0x212: LGH_MASS_CASE(mass_apply_2d, 1, 2);
      case 0x222: LGH_MASS_CASE(mass_apply_2d, 2, 2);
      case 0x224: LGH_MASS_CASE(mass_apply_2d, 2, 4);
      case 0x234: LGH_MASS_CASE(mass_apply_2d, 3, 4);
      case 0x236: LGH_MASS_CASE(mass_apply_2d, 3, 6);
      case 0x246: LGH_MASS_CASE(mass_apply_2d, 4, 6);
      case 0x248: LGH_MASS_CASE(mass_apply_2d, 4, 8);
      case 0x258: LGH_MASS_CASE(mass_apply_2d, 5, 8);
      case 0x25A: LGH_MASS_CASE(mass_apply_2d, 5, 10);
      case 0x26A: LGH_MASS_CASE(mass_apply_2d, 6, 10);
      case 0x312: LGH_MASS_CASE(mass_apply_3d, 1, 2);
      case 0x322: LGH_MASS_CASE(mass_apply_3d, 2, 2);
      case 0x324: LGH_MASS_CASE(mass_apply_3d, 2, 4);
      case 0x334: LGH_MASS_CASE(mass_apply_3d, 3, 4);
      case 0x336: LGH_MASS_CASE(mass_apply_3d, 3, 6);
      case 0x346: LGH_MASS_CASE(mass_apply_3d, 4, 6);
      case 0x348: LGH_MASS_CASE(mass_apply_3d, 4, 8);
      case 0x358: LGH_MASS_CASE(mass_apply_3d, 5, 8);
      case 0x35A: LGH_MASS_CASE(mass_apply_3d, 5, 10);
      case 0x36A: LGH_MASS_CASE(mass_apply_3d, 6, 10);
      default: return unknown_kernel(id);
   }
#undef LGH_MASS_CASE
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

// The energy CG's update as a second launch of the Kronecker kernel (MODE 4) instead of cg_update_k: when the L2 apply runs in
// its Kronecker form.  LGH_L2_FUSED=0: cg_update_k reads the stored product (rounds 1-5).
static bool l2_fused_update(lgh_ctx *c)
{
   const bool on = !(getenv("LGH_L2_FUSED") && getenv("LGH_L2_FUSED")[0] == '0'); // (looked up per call: the switch tests flip it inside one process)
   int form = 0, compact = 0;
   return on && l2_mass_kernel(c, &form, &compact) == LGH_OK && form == 2;
}
// D[q, e] = W[q] s_e ?  In exact arithmetic the data of laghos_assembly.cpp:92-95, w_q detJ0(x_q) rho0(x_q), has this
// form whenever the initial element is affine and rho0 constant in it; the stored entries carry the rounding of the
// Jacobian and density evaluation at each point (measured: 220 ulp at Q3Q2 on the 8^3 box mesh, it grows like 1 / h).
// One thread per element: s_e = mean of D / W over the points (the centre of that rounding cloud), every point within
// tol (relative; kMassRank1Tol unless LGH_MASS_RANK1_TOL) or the stored table stays in use.
constexpr double kMassRank1Tol = 1e-12;
__global__ void __launch_bounds__(256)
mass_rank1_k(const int NE, const int NQ, const double tol, const double *__restrict__ D, const double *__restrict__ W, double *__restrict__ S,
             double *__restrict__ ones, int *flag)
{
   const int e = blockIdx.x * blockDim.x + threadIdx.x;
   if (e >= NE) { return; }
   const double *De = D + (size_t)e * NQ;
   double s = 0.0;
   for (int q = 0; q < NQ; q++) { s += De[q] / W[q]; }
   s /= NQ;
   bool ok = isfinite(s);
   for (int q = 0; q < NQ; q++) { ok = ok && fabs(De[q] - W[q] * s) <= tol * fabs(De[q]); }
   S[e] = s;
   ones[e] = 1.0;
   if (!ok) { *flag = 1; }
}
int mass_data(lgh_ctx *c, const double **Dq, int *dqs, const double **Se)
{
   if (c->mass_rank1 < 0)
   {
      if (!c->massS)
      {
         LGH_HIP_CHECK(hipMalloc((void **)&c->massS, sizeof(double) * (size_t)c->NE));
         LGH_HIP_CHECK(hipMalloc((void **)&c->ones_ne, sizeof(double) * (size_t)c->NE));
      }
      int *flag = c->dev_flags + 3, h = 1;
      LGH_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
      const char *tenv = getenv("LGH_MASS_RANK1_TOL");
      const double tol = tenv ? atof(tenv) : kMassRank1Tol;
      hipLaunchKernelGGL(mass_rank1_k, dim3(ceil_div(c->NE, 256)), dim3(256), 0, c->stream, c->NE, c->NQ, tol, c->massD, c->W, c->massS, c->ones_ne, flag);
      LGH_HIP_CHECK(hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      const char *env = getenv("LGH_MASS_RANK1");
      c->mass_rank1 = (h == 0 && !(env && env[0] == '0')) ? 1 : 0;
   }
   if (c->mass_rank1 == 1) { *Dq = c->W; *dqs = 0; *Se = c->massS; }
   else { *Dq = c->massD; *dqs = c->NQ; *Se = c->ones_ne; }
   return LGH_OK;
}

static MassArgs base_args(lgh_ctx *c, int space)
{
   MassArgs a;
   memset(&a, 0, sizeof(a));
   a.NE = c->NE;
   a.B = (space == LGH_SPACE_H1) ? c->B : c->Bl;
   a.Dq = c->massD; // (every kernel but mass_apply_l2_plane reads the stored table; launch_mass fills Se / dqs)
   return a;
}

int mass_apply_E(lgh_ctx *c, int space, const double *xE, double *yE)
{
   MassArgs a = base_args(c, space);
   a.x = xE;
   a.y = yE;
   return launch_mass<0>(c, space, a);
}

// ---- node kernels -----------------------------------------------------------
// y[n] = sum of element contributions; optional essential-row elimination.
// ELL transpose: t_ell[k*N + n] = E-vector position of the k-th contribution to
// node n (ascending element order), -1 when the node has fewer than k+1.
__device__ __forceinline__ double ell_gather(const int n, const int N, const int deg,
                                             const int *__restrict__ ell, const double *__restrict__ YE)
{
   double s = 0.0;
#pragma unroll 8
   for (int k = 0; k < deg; k++)
   {
      const int p = ell[(size_t)k * N + n];
      if (p >= 0) { s += YE[p]; }
   }
   return s;
}

__global__ void __launch_bounds__(256)
mass_gather_k(const int N, const int deg, const int *__restrict__ ell,
              const double *__restrict__ YE, const uint8_t *__restrict__ ess, double *__restrict__ y)
{
   const int n = blockIdx.x * blockDim.x + threadIdx.x;
   if (n >= N) { return; }
   double s = ell_gather(n, N, deg, ell, YE);
   if (ess && ess[n]) { s = 0.0; }
   y[n] = s;
}

int mass_apply_h1(lgh_ctx *c, const double *x, double *y, bool eliminate)
{
   MassArgs a = base_args(c, LGH_SPACE_H1);
   a.x = x;
   a.map = c->h1map;
   a.y = c->YE;
   int rc = launch_mass<1>(c, LGH_SPACE_H1, a);
   if (rc) { return rc; }
   const bool multi = (c->multi != 0);
   const uint8_t *ess = (eliminate && c->cur_ess >= 0 && !multi) ? c->essmask[c->cur_ess] : nullptr;
   hipLaunchKernelGGL(mass_gather_k, dim3(ceil_div(c->N, 256)), dim3(256), 0, c->stream, c->N,
                      c->t_deg, c->t_ell, c->YE, ess, y);
   LGH_HIP_CHECK(hipGetLastError());
   if (multi)
   {
      rc = halo_sum(c, y, 1);
      if (rc) { return rc; }
      if (eliminate && c->cur_ess >= 0 && c->ess_count[c->cur_ess] > 0)
      {
         rc = vec_zero_list(c, y, c->ess[c->cur_ess], c->ess_count[c->cur_ess]);
      }
   }
   return rc;
}

int mass_apply_l2(lgh_ctx *c, const double *x, double *y)
{
   MassArgs a = base_args(c, LGH_SPACE_L2);
   a.x = x;
   a.y = y;
   return launch_mass<0>(c, LGH_SPACE_L2, a);
}

// ---- Jacobi diagonal: diag_e[d] = sum_q prod_a B(q_a,d_a)^2 D_e[q] ---------------
template <int DIM>
__global__ void __launch_bounds__(256)
mass_diag_k(const int NE, const int D, const int Q, const double *__restrict__ B,
            const double *__restrict__ Dq, double *__restrict__ yE)
{
   const int ND = (DIM == 2) ? D * D : D * D * D;
   const int NQ = (DIM == 2) ? Q * Q : Q * Q * Q;
   const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= (size_t)NE * ND) { return; }
   const int e = (int)(i / ND), d = (int)(i - (size_t)e * ND);
   const int dx = d % D, dy = (d / D) % D, dz = d / (D * D);
   const double *De = Dq + (size_t)e * NQ;
   double s = 0.0;
   if (DIM == 2)
   {
      for (int qy = 0; qy < Q; qy++)
         for (int qx = 0; qx < Q; qx++)
         {
            const double bxv = B[qx + Q * dx], byv = B[qy + Q * dy];
            s += bxv * bxv * byv * byv * De[qx + Q * qy];
         }
   }
   else
   {
      for (int qz = 0; qz < Q; qz++)
         for (int qy = 0; qy < Q; qy++)
            for (int qx = 0; qx < Q; qx++)
            {
               const double bxv = B[qx + Q * dx], byv = B[qy + Q * dy], bzv = B[qz + Q * dz];
               s += bxv * bxv * byv * byv * bzv * bzv * De[qx + Q * (qy + Q * qz)];
            }
   }
   yE[i] = s;
}

__global__ void __launch_bounds__(256)
reciprocal_k(const int n, const double *__restrict__ x, double *__restrict__ y)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { y[i] = 1.0 / x[i]; }
}

int mass_assemble_diag(lgh_ctx *c)
{
   const size_t tot = (size_t)c->NE * c->ND;
   if (c->dim == 2)
   {
      hipLaunchKernelGGL(mass_diag_k<2>, dim3(ceil_div(tot, 256)), dim3(256), 0, c->stream, c->NE,
                         c->D1D, c->Q1D, c->B, c->massD, c->YE);
   }
   else
   {
      hipLaunchKernelGGL(mass_diag_k<3>, dim3(ceil_div(tot, 256)), dim3(256), 0, c->stream, c->NE,
                         c->D1D, c->Q1D, c->B, c->massD, c->YE);
   }
   LGH_HIP_CHECK(hipGetLastError());
   hipLaunchKernelGGL(mass_gather_k, dim3(ceil_div(c->N, 256)), dim3(256), 0, c->stream, c->N,
                      c->t_deg, c->t_ell, c->YE, (const uint8_t *)nullptr, c->diagV);
   LGH_HIP_CHECK(hipGetLastError());
   if (c->multi != 0)
   {
      int rc = halo_sum(c, c->diagV, 1);
      if (rc) { return rc; }
   }
   hipLaunchKernelGGL(reciprocal_k, dim3(ceil_div(c->N, 256)), dim3(256), 0, c->stream, c->N,
                      c->diagV, c->dinvV);
   LGH_HIP_CHECK(hipGetLastError());
   c->mass_gen++; // (the velocity solve keeps 1/diag in its own node numbering: lgh_vcg.hip)
   return LGH_OK;
}

// ---------------------------------------------------------------------------
// CG node / vector kernels
// ---------------------------------------------------------------------------
struct CgVecArgs
{
   int n;              // N (H1) or L2V
   const int *ell;     // H1 fused gather (ELL transpose; null: y already an L-vector)
   int deg;
   const double *YE;   // E-vector from K1 (H1) ...
   double *yL;         // ... or the operator result as an L-vector (L2, multi-GPU H1, atomics)
   int d_in_place;     // 1: d already holds the new direction (L2); 0: form z + beta d here
   const uint8_t *ess; // essential mask or null
   const double *dinv; // Jacobi (null: no preconditioner)
   const double *owner;// ownership weights or null
   const double *b;
   double *d;
   double *x, *r, *z;
   CgScalars *cgs;
   double *partials;
   unsigned int *ticket;
   int iter;
   int allreduce_pending; // multi-GPU: the scalar fold is finished by a later kernel
};

// init: r = b - A x (A x supplied in yL/YE, or absent when x == 0), z = M^-1 r,
// nom = (z, r); sets r0, done, rz.
template <bool HAVE_AX>
__global__ void __launch_bounds__(256)
cg_init_k(const CgVecArgs a)
{
   __shared__ double red[16];
   const int n = blockIdx.x * blockDim.x + threadIdx.x;
   double part = 0.0;
   if (n < a.n)
   {
      double rv = a.b[n];
      if (HAVE_AX) { rv -= a.yL[n]; }
      a.r[n] = rv;
      double zv = rv;
      if (a.dinv)
      {
         zv = rv * a.dinv[n];
         a.z[n] = zv;
      }
      part = (a.owner ? a.owner[n] : 1.0) * zv * rv;
   }
   const double bsum = block_sum(part, red);
   double total;
   if (grid_sum_last_block(bsum, a.partials, a.ticket, red, total))
   {
      if (threadIdx.x == 0)
      {
         CgScalars *s = a.cgs;
         s->rz = total;
         s->rz_prev = total;
         s->nom0 = total;
         s->iters = 0;
         s->first = 1;
         if (!a.allreduce_pending)
         {
            s->r0 = fmax(total * s->rel_tol2, 0.0);
            s->done = (total < 0.0 || total <= s->r0) ? 1 : 0;
         }
      }
   }
}

// multi-GPU: after the all-reduce of rz (init) finish the scalar bookkeeping
__global__ void cg_init_finish_k(CgScalars *s)
{
   s->rz_prev = s->rz;
   s->nom0 = s->rz;
   s->r0 = fmax(s->rz * s->rel_tol2, 0.0);
   s->done = (s->rz < 0.0 || s->rz <= s->r0) ? 1 : 0;
}

// K2: see file header.
// U nodes per thread with every load of all U nodes issued before first use: the
// kernel is pure memory latency (PMC: 90 % of wave cycles waiting), so the number
// of independent loads in flight per lane is what sets its speed.
constexpr int kUpdU = 2;
template <bool FUSED_GATHER, int DEG>
__global__ void __launch_bounds__(256)
cg_update_k(const CgVecArgs a)
{
   __shared__ double red[16];
   if (a.cgs->done) { return; }
   if (a.allreduce_pending && cg_pending_den(a.cgs, blockIdx.x == 0 && threadIdx.x == 0)) { return; }
   const double alpha = a.cgs->rz / a.cgs->den;
   // `first` was cleared by K1 of this iteration: iteration 1 is recognised by iter
   const bool it1 = (a.iter == 1);
   const double beta = it1 ? 0.0 : a.cgs->rz / a.cgs->rz_prev;
   const int base = blockIdx.x * (256 * kUpdU) + threadIdx.x;
   int n[kUpdU];
   bool ok[kUpdU];
#pragma unroll
   for (int k = 0; k < kUpdU; k++)
   {
      n[k] = base + 256 * k;
      ok[k] = n[k] < a.n;
      if (!ok[k]) { n[k] = a.n - 1; }
   }
   // ---- phase 1: issue all independent loads
   int pidx[kUpdU][DEG > 0 ? DEG : 1];
   if (FUSED_GATHER)
   {
#pragma unroll
      for (int k = 0; k < kUpdU; k++)
#pragma unroll
         for (int j = 0; j < DEG; j++) { pidx[k][j] = (j < a.deg) ? a.ell[(size_t)j * a.n + n[k]] : -1; }
   }
   double zold[kUpdU], dold[kUpdU], xo[kUpdU], ro[kUpdU], di[kUpdU], ow[kUpdU], zv[kUpdU];
   bool es[kUpdU];
#pragma unroll
   for (int k = 0; k < kUpdU; k++)
   {
      ro[k] = a.r[n[k]];
      xo[k] = a.x[n[k]];
      di[k] = a.dinv ? a.dinv[n[k]] : 1.0;
      zold[k] = a.dinv ? a.z[n[k]] : ro[k];
      dold[k] = (a.d_in_place || !it1) ? a.d[n[k]] : 0.0;
      ow[k] = a.owner ? a.owner[n[k]] : 1.0;
      es[k] = a.ess ? (a.ess[n[k]] != 0) : false;
      if (!FUSED_GATHER) { zv[k] = a.yL[n[k]]; }
   }
   // ---- phase 2: dependent E-vector reads
   if (FUSED_GATHER)
   {
      double ye[kUpdU][DEG > 0 ? DEG : 1];
#pragma unroll
      for (int k = 0; k < kUpdU; k++)
#pragma unroll
         for (int j = 0; j < DEG; j++) { ye[k][j] = (pidx[k][j] >= 0) ? a.YE[pidx[k][j]] : 0.0; }
#pragma unroll
      for (int k = 0; k < kUpdU; k++)
      {
         // ascending contribution order, skipping absent slots (same sum as the CSR loop)
         double s = 0.0;
#pragma unroll
         for (int j = 0; j < DEG; j++) { if (pidx[k][j] >= 0) { s += ye[k][j]; } }
         zv[k] = s;
      }
   }
   // ---- phase 3: updates
   double part = 0.0;
#pragma unroll
   for (int k = 0; k < kUpdU; k++)
   {
      if (!ok[k]) { continue; }
      const double z_ = es[k] ? 0.0 : zv[k];
      double dv;
      if (a.d_in_place) { dv = dold[k]; }
      else
      {
         // same expression as K1's gather: d = z_old + beta * d_old
         dv = zold[k];
         if (!it1) { dv += beta * dold[k]; }
         a.d[n[k]] = dv;
      }
      const double xv = xo[k] + alpha * dv;
      const double rv = ro[k] - alpha * z_;
      a.x[n[k]] = xv;
      a.r[n[k]] = rv;
      double pz = rv;
      if (a.dinv)
      {
         pz = rv * di[k];
         a.z[n[k]] = pz;
      }
      part += ow[k] * rv * pz;
   }
   const double bsum = block_sum(part, red);
   double total;
   if (grid_sum_last_block(bsum, a.partials, a.ticket, red, total))
   {
      if (threadIdx.x == 0)
      {
         CgScalars *s = a.cgs;
         s->rz_prev = s->rz;
         s->rz = total; // betanom
         if (!a.allreduce_pending)
         {
            s->iters = a.iter;
            if (total < 0.0 || total <= s->r0) { s->done = 1; }
         }
      }
   }
}
// (several ranks) a finished solve leaves rz = den = 0: the exchanges of the launches the host enqueued past
// convergence then sum zeros instead of multiplying the global values by the number of ranks each time
__global__ void cg_update_finish_k(CgScalars *s, int iter)
{
   if (s->done) { s->rz = 0.0; s->den = 0.0; return; }
   s->iters = iter;
   if (s->rz < 0.0 || s->rz <= s->r0) { s->done = 1; s->rz = 0.0; s->den = 0.0; }
}

__global__ void cg_set_tol_k(CgScalars *s, double rel_tol2)
{
   s->rel_tol2 = rel_tol2;
   s->done = 0;
   s->first = 1;
   s->iters = 0;
}

static int launch_l2_update(lgh_ctx *c, const MassArgs &m, const CgVecArgs &v)
{
   MassArgs a = m;
   const int rc = mass_data(c, &a.Dq, &a.dqs, &a.Se);
   if (rc) { return rc; }
   a.M1 = c->M1l;
   a.ux = v.x;
   a.ur = v.r;
   a.partials = v.partials;
   a.ticket = v.ticket;
   a.iter = v.iter;
#define LGH_L2U(L_, NEB_) hipLaunchKernelGGL((mass_apply_l2_kron<L_, NEB_, 4>), dim3(ceil_div(c->NE, NEB_)), dim3(256), 0, c->stream, a)
   if (l2_one_round(c)) { LGH_L2U(5, 10); }
   else
   {
      switch (c->L1D)
      {
         case 1: LGH_L2U(1, 256); break;
         case 2: LGH_L2U(2, 128); break;
         case 3: LGH_L2U(3, 64); break;
         case 4: LGH_L2U(4, 32); break;
         default: LGH_L2U(5, 16); break;
      }
   }
#undef LGH_L2U
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

int cg_solve(lgh_ctx *c, int space, const double *b, double *x, double rel_tol, int max_iter,
             int *iters, bool x_is_zero)
{
   const bool h1 = (space == LGH_SPACE_H1);
   const int n = h1 ? c->N : c->L2V;
   const bool multi = (c->multi != 0);
   int rc;
   hipLaunchKernelGGL(cg_set_tol_k, dim3(1), dim3(1), 0, c->stream, c->cgs, rel_tol * rel_tol);

   CgVecArgs v;
   memset(&v, 0, sizeof(v));
   v.n = n;
   v.b = b;
   v.x = x;
   v.r = c->cg_r;
   v.z = h1 ? c->cg_z : c->cg_r; // no preconditioner: z aliases r
   v.dinv = h1 ? c->dinvV : nullptr;
   v.owner = (h1 && multi) ? c->owner : nullptr;
   v.cgs = c->cgs;
   v.partials = c->partials;
   v.ticket = c->tickets;
   v.allreduce_pending = multi ? 1 : 0;
   const uint8_t *ess = (h1 && c->cur_ess >= 0) ? c->essmask[c->cur_ess] : nullptr;
   const int nb = ceil_div(n, 256);
   const int nbu = ceil_div(n, 256 * kUpdU);

   // --- r = b - A x  (iterative_mode for H1; L2 starts from x = 0: solvers.cpp)
   if (!h1) { rc = vec_set(c, x, 0.0, n); if (rc) { return rc; } x_is_zero = true; }
   if (!x_is_zero)
   {
      rc = mass_apply_h1(c, x, c->cg_y, true);
      if (rc) { return rc; }
      v.yL = c->cg_y;
      hipLaunchKernelGGL(cg_init_k<true>, dim3(nb), dim3(256), 0, c->stream, v);
   }
   else { hipLaunchKernelGGL(cg_init_k<false>, dim3(nb), dim3(256), 0, c->stream, v); }
   LGH_HIP_CHECK(hipGetLastError());
   if (multi)
   {
      rc = allreduce_dev(c, &c->cgs->rz, 1, 0);
      if (rc) { return rc; }
      hipLaunchKernelGGL(cg_init_finish_k, dim3(1), dim3(1), 0, c->stream, c->cgs);
   }

   MassArgs m = base_args(c, space);
   m.x = h1 ? c->cg_z : c->cg_r;
   m.map = h1 ? c->h1map : nullptr;
   m.y = h1 ? c->YE : c->cg_y;
   m.cgs = c->cgs;
   m.partials = c->partials + c->part_stride;
   m.ticket = c->tickets + 1 * kTicketSlot;
   m.multi = multi ? 1 : 0;

   m.d = c->cg_d0;
   v.d = c->cg_d0;
   v.d_in_place = h1 ? 0 : 1;
   m.no_y = (!h1 && l2_fused_update(c)) ? 1 : 0;
   int it = 0;
   CgScalars *hs = (CgScalars *)c->host_pinned;
   // chunk = iterations enqueued between two looks at the convergence flag.  The
   // iteration count of a mass solve barely changes from one RK stage to the next,
   // so the first chunk is the previous count for this (space, component): the
   // common case costs two host round trips and no wasted launches; kernels
   // enqueued past convergence exit on the device-side `done` flag.
   int &last = c->cg_last_iters[h1 ? 0 : 1][(h1 && c->cur_ess >= 0) ? c->cur_ess : 0];
   int chunk = last > 0 ? last : 8;
   bool done = false;
   bool first_look = true;
   int looks = 0;
   while (!done)
   {
      // several ranks: the outcome of the last enqueued update is still pending (cg_pending_update) - commit it
      if (multi && it > 0) { hipLaunchKernelGGL(cg_update_finish_k, dim3(1), dim3(1), 0, c->stream, c->cgs, it); }
      LGH_HIP_CHECK(hipMemcpyAsync(hs, c->cgs, sizeof(CgScalars), hipMemcpyDeviceToHost, c->stream));
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      if (hs->done || it >= max_iter) { break; }
      // after the first chunk (the previous solve's count): two more iterations per look while the count only creeps, then
      // doubling - a solve whose count jumps (the unpreconditioned L2 CG of the first steps at order 4: 225 -> 300) must not pay
      // a host look every two iterations (round 6: 149 looks per RK step at config 5); launches past convergence return at once
      if (!first_look) { chunk = (++looks <= 2) ? 2 : std::min(64, 2 * chunk); }
      first_look = false;
      const int upto = std::min(max_iter, it + chunk);
      for (; it < upto;)
      {
         ++it;
         m.iter = it;
         kt_begin(c, h1 ? LGH_KERNEL_MASS_CG_H1 : LGH_KERNEL_MASS_CG_L2);
         rc = h1 ? launch_mass<2>(c, space, m) : launch_mass<3>(c, space, m);
         kt_end(c, h1 ? LGH_KERNEL_MASS_CG_H1 : LGH_KERNEL_MASS_CG_L2);
         if (rc) { return rc; }
         v.iter = it;
         v.ess = ess;
         if (h1 && !multi)
         {
            v.ell = c->t_ell;
            v.deg = c->t_deg;
            v.YE = c->YE;
            kt_begin(c, LGH_KERNEL_CG_UPDATE_H1);
            if (c->t_deg <= 4) { hipLaunchKernelGGL((cg_update_k<true, 4>), dim3(nbu), dim3(256), 0, c->stream, v); }
            else if (c->t_deg <= 8) { hipLaunchKernelGGL((cg_update_k<true, 8>), dim3(nbu), dim3(256), 0, c->stream, v); }
            else
            {
               // unusual valence (not a tensor-product mesh): unfused gather, then update
               hipLaunchKernelGGL(mass_gather_k, dim3(nb), dim3(256), 0, c->stream, c->N, c->t_deg,
                                  c->t_ell, c->YE, (const uint8_t *)nullptr, c->cg_y);
               v.yL = c->cg_y;
               hipLaunchKernelGGL((cg_update_k<false, 0>), dim3(nbu), dim3(256), 0, c->stream, v);
            }
            kt_end(c, LGH_KERNEL_CG_UPDATE_H1);
         }
         else
         {
            if (h1)
            {
               // multi-GPU: assemble the L-vector, sum shared nodes, reduce den
               hipLaunchKernelGGL(mass_gather_k, dim3(nb), dim3(256), 0, c->stream, c->N, c->t_deg,
                                  c->t_ell, c->YE, (const uint8_t *)nullptr, c->cg_y);
               rc = halo_sum(c, c->cg_y, 1);
               if (rc) { return rc; }
            }
            if (multi)
            {
               rc = allreduce_dev(c, &c->cgs->den, 1, 0); // breakdown is looked at by cg_update_k (cg_pending_den)
               if (rc) { return rc; }
            }
            v.yL = c->cg_y;
            if (!h1 && m.no_y) { rc = launch_l2_update(c, m, v); if (rc) { return rc; } } // (the update forms M d of its zones itself)
            else { hipLaunchKernelGGL((cg_update_k<false, 0>), dim3(nbu), dim3(256), 0, c->stream, v); }
            if (multi)
            {
               rc = allreduce_dev(c, &c->cgs->rz, 1, 0); // convergence is looked at by the next K1 (cg_pending_update)
               if (rc) { return rc; }
            }
         }
         LGH_HIP_CHECK(hipGetLastError());
      }
   }
   // upstream: final_iter = max_iter when the loop runs out without converging
   int fin = hs->iters;
   if (!hs->done && it >= max_iter) { fin = max_iter; }
   last = fin;
   if (iters) { *iters = fin; }
   return LGH_OK;
}

// ---- the L2 (energy) solve split into an enqueue-only half and a completing half -----
// SolveEnergy (laghos_solver.cpp:400-493) needs only the quadrature data and the
// velocity block of S - not the result of SolveVelocity.  lgh_solve_energy_begin
// enqueues F^T v and the first chunk of CG iterations on the context's second stream
// (the caller swaps the stream in), the velocity solve then runs on the main stream
// with its host round trips, and lgh_solve_energy_end looks at the convergence flag.
// Same kernels, same order and same scalars as cg_solve(LGH_SPACE_L2): the iterates
// are identical, only the initial look (rhs == 0) is folded into the first one.
struct L2Run
{
   CgVecArgs v;
   MassArgs m;
   int it, max_iter, nbu;
   bool active;
};

static int l2_enqueue(lgh_ctx *c, L2Run *r, int upto)
{
   for (; r->it < upto;)
   {
      ++r->it;
      r->m.iter = r->it;
      int rc = launch_mass<3>(c, LGH_SPACE_L2, r->m);
      if (rc) { return rc; }
      if (r->m.multi) { rc = allreduce_dev(c, &c->cgs->den, 1, 0); if (rc) { return rc; } } // as cg_solve
      r->v.iter = r->it;
      if (r->m.no_y) { rc = launch_l2_update(c, r->m, r->v); if (rc) { return rc; } }
      else { hipLaunchKernelGGL((cg_update_k<false, 0>), dim3(r->nbu), dim3(256), 0, c->stream, r->v); }
      LGH_HIP_CHECK(hipGetLastError());
      if (r->m.multi) { rc = allreduce_dev(c, &c->cgs->rz, 1, 0); if (rc) { return rc; } }
   }
   return LGH_OK;
}

static int cg_l2_begin_impl(lgh_ctx *c, const double *b, double *x, double rel_tol, int max_iter, const bool enqueue)
{
   const bool multi = c->multi != 0; // several ranks: the caller has checked that the second stream has a communicator
   if (!c->l2run) { c->l2run = new L2Run(); }
   L2Run *r = (L2Run *)c->l2run;
   const int n = c->L2V;
   hipLaunchKernelGGL(cg_set_tol_k, dim3(1), dim3(1), 0, c->stream, c->cgs, rel_tol * rel_tol);
   CgVecArgs &v = r->v;
   memset(&v, 0, sizeof(v));
   v.n = n;
   v.b = b;
   v.x = x;
   v.r = c->cg_r;
   v.z = c->cg_r; // no preconditioner: z aliases r
   v.cgs = c->cgs;
   v.partials = c->partials;
   v.ticket = c->tickets;
   v.allreduce_pending = multi ? 1 : 0;
   int rc = vec_set(c, x, 0.0, n); // the L2 solve starts from x = 0
   if (rc) { return rc; }
   hipLaunchKernelGGL(cg_init_k<false>, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, v);
   LGH_HIP_CHECK(hipGetLastError());
   if (multi)
   {
      rc = allreduce_dev(c, &c->cgs->rz, 1, 0);
      if (rc) { return rc; }
      hipLaunchKernelGGL(cg_init_finish_k, dim3(1), dim3(1), 0, c->stream, c->cgs);
   }
   MassArgs &m = r->m;
   m = base_args(c, LGH_SPACE_L2);
   m.x = c->cg_r;
   m.map = nullptr;
   m.y = c->cg_y;
   m.cgs = c->cgs;
   m.partials = c->partials + c->part_stride;
   m.ticket = c->tickets + 1 * kTicketSlot;
   m.multi = multi ? 1 : 0;
   m.d = c->cg_d0;
   v.d = c->cg_d0;
   v.d_in_place = 1;
   v.yL = c->cg_y;
   m.no_y = l2_fused_update(c) ? 1 : 0;
   r->it = 0;
   r->max_iter = max_iter;
   r->nbu = ceil_div(n, 256 * kUpdU);
   r->active = true;
   const int last = c->cg_last_iters[1][0];
   if (!enqueue) { return LGH_OK; } // (lockstep: the velocity solve interleaves the iterations)
   return l2_enqueue(c, r, std::min(max_iter, last > 0 ? last : 8));
}
int cg_l2_begin(lgh_ctx *c, const double *b, double *x, double rel_tol, int max_iter) { return cg_l2_begin_impl(c, b, x, rel_tol, max_iter, true); }

// ---- lockstep with the velocity CG (lgh_common.hpp): set-up and initial residual here, one iteration per velocity
// iteration from inside vcg_solve (apply behind K1, update behind K2), the rest - if any - by cg_l2_end as before
bool l2_lockstep_possible(lgh_ctx *c) { return l2_fused_update(c); }
int l2_lockstep_limit(lgh_ctx *c)
{
   const L2Run *r = (const L2Run *)c->l2run;
   const int last = c->cg_last_iters[1][0];
   const int want = last > 0 ? last : 8;
   return (r && r->active) ? std::min(want, r->max_iter) : 0; // (never past the cap this solve was given)
}
int cg_l2_begin_lockstep(lgh_ctx *c, const double *b, double *x, double rel_tol, int max_iter) { return cg_l2_begin_impl(c, b, x, rel_tol, max_iter, false); }
int l2_lockstep_apply(lgh_ctx *c, int it, const LockstepWords &prev, double *den_mirror)
{
   L2Run *r = (L2Run *)c->l2run;
   if (!r || !r->active || !r->m.no_y || it != r->it + 1 || it > r->max_iter) { set_error("l2_lockstep_apply: no energy solve set up for iteration %d", it); return LGH_ERR_ARG; }
   ++r->it;
   r->m.iter = r->it;
   r->m.ls = prev;
   r->m.ls_on = 1;
   r->m.ls_den_mirror = den_mirror;
   return launch_mass<3>(c, LGH_SPACE_L2, r->m);
}
int l2_lockstep_update(lgh_ctx *c, int it, const double *den_src, long long *word_out)
{
   L2Run *r = (L2Run *)c->l2run;
   if (!r || !r->active || it != r->it) { set_error("l2_lockstep_update: iteration %d has no apply", it); return LGH_ERR_ARG; }
   r->v.iter = r->it;
   MassArgs m = r->m;
   m.ls_on = 1;
   m.ls_den_src = den_src;
   m.ls_word_out = word_out;
   return launch_l2_update(c, m, r->v);
}
__global__ void l2_lockstep_fold_k(CgScalars *s, const int it, const LockstepWords w)
{
   if (s->done) { return; }
   (void)lockstep_pending_update(s, it + 1, lockstep_rz(w), true); // what the apply of iteration it + 1 would commit
}
int l2_lockstep_fold(lgh_ctx *c, int it, const LockstepWords &last)
{
   L2Run *r = (L2Run *)c->l2run;
   if (!r || !r->active || it != r->it) { set_error("l2_lockstep_fold: iteration %d is not the last one enqueued", it); return LGH_ERR_ARG; }
   hipLaunchKernelGGL(l2_lockstep_fold_k, dim3(1), dim3(1), 0, c->stream, c->cgs, it, last);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int cg_l2_end(lgh_ctx *c, int *iters);
int cg_l2_end_lockstep(lgh_ctx *c, int *iters)
{
   L2Run *r = (L2Run *)c->l2run;
   if (!r || !r->active) { return LGH_ERR_ARG; }
   r->m.ls_on = 0; // (from here on the solve reduces its scalars itself: the sequence of cg_solve)
   r->m.ls_den_mirror = nullptr;
   const int inside = r->it;
   const int rc = cg_l2_end(c, iters);
   c->ls_stats[0] += 1;
   c->ls_stats[1] += inside;
   c->ls_stats[2] += r->it - inside;
   return rc;
}

int cg_l2_end(lgh_ctx *c, int *iters)
{
   L2Run *r = (L2Run *)c->l2run;
   if (!r || !r->active) { return LGH_ERR_ARG; }
   CgScalars *hs = (CgScalars *)c->host_pinned;
   int looks = 0, chunk = 2;
   while (true)
   {
      if (r->m.multi && r->it > 0) { hipLaunchKernelGGL(cg_update_finish_k, dim3(1), dim3(1), 0, c->stream, c->cgs, r->it); } // as cg_solve
      LGH_HIP_CHECK(hipMemcpyAsync(hs, c->cgs, sizeof(CgScalars), hipMemcpyDeviceToHost, c->stream));
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      if (hs->done || r->it >= r->max_iter) { break; }
      chunk = (++looks <= 2) ? 2 : std::min(64, 2 * chunk); // (as cg_solve: a count that jumps is not chased two iterations at a time)
      const int rc = l2_enqueue(c, r, std::min(r->max_iter, r->it + chunk));
      if (rc) { return rc; }
   }
   int fin = hs->iters;
   if (!hs->done && r->it >= r->max_iter) { fin = r->max_iter; }
   c->cg_last_iters[1][0] = fin;
   if (iters) { *iters = fin; }
   r->active = false;
   return LGH_OK;
}
void cg_l2_free(lgh_ctx *c)
{
   delete (L2Run *)c->l2run;
   c->l2run = nullptr;
}


} // namespace lgh
