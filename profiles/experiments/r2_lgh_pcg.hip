// lgh_pcg.hip — the mass solves of one RK stage as ONE persistent kernel.
//
// Reference: /root/reference/laghos_solver.cpp:363-398 (SolveVelocity: CG_VMass.Mult once per
// velocity component); the recurrence is upstream CGSolver::Mult.  lgh_vcg.hip runs
// the velocity components in lockstep with two launches per iteration (K1: element batches, K2:
// nodes).  A third of each of those launches is not work: launch, the spread of the workgroups over
// the chip, and the ticketed reduction at the tail (profiles/r1_k1_block_timestamps.txt), paid ~180
// times per RK4 step.
//
// Here a solve is one launch.  The grid is one resident workgroup per CU; the two dot products of an
// iteration are folded at two grid-wide barriers, and every workgroup derives the same scalars (alpha,
// beta, convergence) from the same partial sums in the same order, so there is no "last block" and no
// scalar round trip through memory:
//
//   init     b = -(H1R^T F.1) (essential rows zeroed), r = b, x = 0, nom_c = (r_c, r_c/diag)
//   barrier  nom, r0 = max(nom rel_tol^2, 0), converged components drop out
//   loop it = 1, 2, ...
//     phase A (element batches, plane-per-thread contraction as vcg_apply_plane):
//              d = r/diag + beta d formed in the gather, Y_E = B^T D B d_E, partial (d, A d)
//     barrier  den; breakdown check (den == 0)
//     phase B (nodes): A d = ELL gather of Y_E (fixed order), essential rows, d stored,
//              x += alpha d (every second iteration, both terms), r -= alpha A d, partial (r, r/diag)
//     barrier  betanom; convergence check; iteration count
//
// Every component performs exactly the operations of its stand-alone CGSolver::Mult; the fold order
// of the dot products is fixed, so results are bit-reproducible run to run.
//
// Code structure: the phases are inlined into one kernel body, which needs care: left alone the
// register allocator lets the phases interfere (address arithmetic of the node phase hoisted out of
// the solve loop and spilled across the element phase; loaded values spilled to scratch behind
// s_waitcnt vmcnt(0) in branchy bodies).  Each phase therefore starts from opaque copies of the thread
// index and of the argument pointer, reads its arguments from a device-memory block instead of ~50
// kernel-argument SGPRs, and the node phase is straight-line code.  Real function calls
// (__noinline__) isolate the phases perfectly but cost ~8 us per call in callee-saved register
// traffic through scratch memory (measured).
//
// Inter-workgroup visibility (MI355X: per-CU L1 never refreshed by other CUs' stores, write-back L2
// per XCD): before arriving at a barrier every wave drains its stores, lane 0 issues an agent-scope
// release (L2 write-back) and arrives; after the barrier wave 0 issues an agent-scope acquire (L1
// invalidate) and the workgroup synchronises (cdna_hip_programming.md Guideline 16).  Spins are
// bounded: a barrier that does not complete within kSpinLimit ticks of the 100 MHz clock sets
// res->error and every workgroup leaves (the host reports LGH_ERR_HIP) - a missing co-resident
// workgroup cannot hang the GPU.
#include "lgh_common.hpp"

#include <type_traits>

namespace lgh
{

constexpr int kPV = 3; // velocity components
constexpr int kPC = 4; // room for a fourth lockstep component
constexpr unsigned long long kSpinLimit = 300000000ull; // 3 s of the 100 MHz wall clock
constexpr int kCtrStride = 32;                          // uints: counters 128 B apart
constexpr int kCtrShards = 8;
constexpr int kSweepMax = 8;                            // blocks per lane of the sweeping wave: grids up to 512
constexpr int kTrIter = 24, kTrPer = 22;                // LGH_PCG_TRACE: stamps per workgroup and iteration

struct PcgResult
{
   int iters[kPC], done[kPC];
   int error, iterations; // iterations = loop trips executed by the kernel
   double rz[kPC], den[kPC];
   unsigned long long shader_clk, wall_clk; // block 0: shader cycles and 100 MHz ticks spent in the kernel
};

// Lives in device memory; the kernel gets a pointer (the phases read what they need through it).
struct PcgArgs
{
   int NE, N, nv;
   const double *B, *Dq;
   const int *map;
   const unsigned *ellz;   // ELL transpose as BYTE offsets into a Y_E plane; absent entries point at the zero slot NE*ND
   const uint8_t *ess[kPV];
   const uint8_t *essbits; // bit k: node is essential for component k
   const int *nstart;      // node range of worker w: [nstart[w], nstart[w+1]) (balanced by cost, multiples of 16)
   const double *dinv;
   const double *FE; // force E-vector (D1D^3, dim, NE): b = -(H1R^T FE) with essential rows zeroed; or nullptr
   double *b;        // kPV*N right-hand sides: output when FE, input otherwise
   double *x;        // kPV*N solutions (zero initial guess)
   double *r, *d;    // kPV*N work vectors
   double *YE;       // kPV planes of NE*ND + pad
   size_t ye_stride;
   double rel_tol2;
   int max_iter;
   unsigned long long *gran; // 2 slots of G records of kPC doubles (room for twice that), zero at launch
   unsigned int *ctr;        // (kCtrShards + 1) arrival counters, kCtrStride apart, zero at launch
   PcgResult *res;
   unsigned long long *trace; // debug (LGH_PCG_TRACE=file)
};

struct PcgState
{
   double rz[kPC], rz_prev[kPC], den[kPC], r0[kPC], tot[kPC];
   double alpha_last[kPC]; // alpha of the latest completed update of x, r
   int done[kPC], iters[kPC], nupd[kPC];
   int all_done, abort;
};

struct Vals4
{
   double v[kPC];
};

__device__ __forceinline__ double pcg_uniform(const double v)
{
   const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
   const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
   return __hiloint2double(hi, lo);
}
// function arguments arrive in vector registers; a pointer every lane agrees on goes back to scalar
// registers (loads through it are then known to be uniform)
// Opaque copy of a uniform pointer: what is loaded through it cannot be hoisted out of the solve loop
// and kept live across the other phase.
__device__ __forceinline__ const struct PcgArgs *uniform_ptr(const struct PcgArgs *p)
{
   asm volatile("" : "+s"(p));
   return p;
}
template <typename T> __device__ __forceinline__ T *uniform_ptr(T *p) { return p; }
__device__ __forceinline__ int uniform_int(const int v) { return v; }
// opaque copy of the thread index, per phase and iteration: thread-index arithmetic and the addresses
// derived from it would otherwise be hoisted out of the solve loop and spilled across the other phase
__device__ __forceinline__ int opaque_tid()
{
   int t = threadIdx.x;
   asm volatile("" : "+v"(t));
   return t;
}

// base + 32-bit byte offset: the form the compiler turns into global_load ... v_off, s[base] (no 64-bit
// address arithmetic per lane).  Valid while a vector set / a Y_E plane is smaller than 4 GiB.
template <typename T> __device__ __forceinline__ T ld_off(const T *base, const unsigned byte_off)
{
   return *(const T *)((const char *)base + byte_off);
}
__device__ __forceinline__ double *ptr_off(double *base, const unsigned byte_off)
{
   return (double *)((char *)base + byte_off);
}

// work distribution: the workgroups of XCD x (observed: block b runs on XCD b % 8; used for locality
// only) sweep the x-th contiguous eighth of the element batches and of the node ranges
struct PcgWork
{
   int per, worker, w; // workgroups per XCD, this one takes part, its index xcd*per + j
   int xcd, jx;
};
__device__ __forceinline__ PcgWork pcg_work()
{
   PcgWork k;
   const int G = gridDim.x, bid = blockIdx.x;
   k.per = G >> 3;
   k.worker = bid < (k.per << 3);
   k.xcd = bid & 7;
   k.jx = bid >> 3;
   k.w = k.xcd * k.per + k.jx;
   return k;
}

// ---- grid-wide barrier + sum --------------------------------------------------------------------
// vals: this thread's contributions.  On return (thread 0) st->tot[0..kPC) holds the grid totals - every
// workgroup computes identical values; returns false when the barrier timed out (same answer in every
// thread).
//
// A workgroup folds its contributions (wave shuffles, then the waves in order); lane 0 releases the
// workgroup's stores (agent scope), publishes the partial sums as one 32-byte record (write-through
// 8-byte stores) and arrives: sharded counters (block index mod 8; the last arriver of a shard arrives on
// the top counter), polled by that one lane with s_sleep in between - a sweep of tagged records by every
// waiting workgroup instead of the counter was measured to cost the workgroups still working 25 % of
// their speed.  Once the top counter is complete lane 0 acquires, and thread t of every workgroup reads
// the record of block t (+ NT, ...), all loads in flight at once; the fold is lane-wise in block order,
// the shuffle tree across a wave, the waves in order: fixed, bit-reproducible.  Record slots alternate
// between consecutive barriers, so a fast workgroup never overwrites what a slow one still has to read.
__device__ __forceinline__ bool pcg_barrier(const PcgArgs *ap_, const Vals4 vals, const unsigned epoch, double *red,
                                         PcgState *st, unsigned long long *tr)
{
   const PcgArgs *ap = uniform_ptr(ap_);
   const int NT = blockDim.x, NW = (NT + 63) >> 6;
   const int tid = opaque_tid(), lane = tid & 63, wid = tid >> 6;
   const int nact = (NT - (wid << 6)) < 64 ? (NT - (wid << 6)) : 64;
   const unsigned G = gridDim.x, bid = blockIdx.x;
   double *rec = (double *)ap->gran + (size_t)(epoch & 1u) * G * kPC;
   PcgResult *res = ap->res;
   double v[kPC];
#pragma unroll
   for (int i = 0; i < kPC; i++) { v[i] = wave_sum(vals.v[i], lane, nact); }
   if (lane == 0)
   {
#pragma unroll
      for (int i = 0; i < kPC; i++) { red[wid * kPC + i] = v[i]; }
   }
   asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's global stores have left
   __syncthreads();
   if (tid == 0)
   {
      int ok = 1;
      if (tr) { tr[0] = wall_clock64(); }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tr) { tr[1] = wall_clock64(); }
#pragma unroll
      for (int i = 0; i < kPC; i++)
      {
         double s = 0.0;
         for (int w = 0; w < NW; w++) { s += red[w * kPC + i]; }
         __hip_atomic_store(&rec[(size_t)bid * kPC + i], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // record out before the arrival
      const unsigned shard = bid % kCtrShards;
      const unsigned members = G / kCtrShards + ((shard < G % kCtrShards) ? 1u : 0u);
      const unsigned nsh = G < (unsigned)kCtrShards ? G : (unsigned)kCtrShards;
      unsigned int *ctr = ap->ctr;
      unsigned int *top = ctr + kCtrShards * kCtrStride;
      const unsigned arr = __hip_atomic_fetch_add(ctr + shard * kCtrStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (arr + 1 == members * epoch) { __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      const unsigned target = nsh * epoch;
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
      {
         __builtin_amdgcn_s_sleep(2);
         if (wall_clock64() - t0 > kSpinLimit || __hip_atomic_load(&res->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
         {
            __hip_atomic_store(&res->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            break;
         }
      }
      if (tr) { tr[2] = wall_clock64(); }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (tr) { tr[3] = wall_clock64(); }
      st->abort = ok ? 0 : 1;
   }
   __syncthreads();
   if (st->abort) { return false; }
   double acc[kPC];
#pragma unroll
   for (int i = 0; i < kPC; i++) { acc[i] = 0.0; }
   for (unsigned k0 = (unsigned)tid; k0 < G; k0 += 2u * (unsigned)NT) // one trip for grids up to 2*NT blocks
   {
      double pl[2][kPC];
#pragma unroll
      for (int q = 0; q < 2; q++)
      {
         const unsigned k = k0 + (unsigned)q * (unsigned)NT;
         const unsigned kk = k < G ? k : 0u;
#pragma unroll
         for (int i = 0; i < kPC; i++)
         {
            pl[q][i] = __hip_atomic_load(&rec[(size_t)kk * kPC + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         }
      }
#pragma unroll
      for (int q = 0; q < 2; q++)
      {
         const bool in = (k0 + (unsigned)q * (unsigned)NT) < G;
#pragma unroll
         for (int i = 0; i < kPC; i++) { acc[i] += in ? pl[q][i] : 0.0; }
      }
   }
#pragma unroll
   for (int i = 0; i < kPC; i++) { acc[i] = wave_sum(acc[i], lane, nact); }
   if (lane == 0)
   {
#pragma unroll
      for (int i = 0; i < kPC; i++) { red[wid * kPC + i] = acc[i]; }
   }
   __syncthreads();
   if (tid == 0)
   {
#pragma unroll
      for (int i = 0; i < kPC; i++)
      {
         double s = 0.0;
         for (int w = 0; w < NW; w++) { s += red[w * kPC + i]; }
         st->tot[i] = s;
      }
   }
   return true; // the caller's thread 0 updates *st, then __syncthreads()
}

// ---- init: b, r, x = 0, partial nom ---------------------------------------------------------------
template <int ND>
__device__ __forceinline__ Vals4 pcg_phase_init(const PcgArgs *ap_)
{
   const PcgArgs *ap = uniform_ptr(ap_);
   const PcgWork wk = pcg_work();
   const int NT = blockDim.x, tid = threadIdx.x;
   const size_t N = (size_t)ap->N;
   const int n0 = wk.worker ? ap->nstart[wk.w] : 0;
   const int n1 = wk.worker ? ap->nstart[wk.w + 1] : 0;
   const double *FE = ap->FE;
   const int NE = ap->NE;
   Vals4 pv;
#pragma unroll
   for (int i = 0; i < kPC; i++) { pv.v[i] = 0.0; }
   for (int n = n0 + tid; n < n1; n += NT)
   {
      const double di = ap->dinv[n];
      if (FE)
      {
         long pos[8];
#pragma unroll
         for (int j = 0; j < 8; j++)
         {
            const int p = (int)(ap->ellz[(size_t)j * N + n] >> 3); // e*ND + d, or NE*ND: no j-th contribution
            const int e = p / ND;
            pos[j] = (e < NE) ? (long)kPV * ND * e + (p - e * ND) : -1;
         }
#pragma unroll
         for (int k = 0; k < kPV; k++)
         {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 8; j++) { if (pos[j] >= 0) { s += FE[pos[j] + (long)ND * k]; } }
            double bv = -s;
            if (ap->ess[k] && ap->ess[k][n]) { bv = 0.0; }
            const size_t i = (size_t)k * N + n;
            ap->b[i] = bv;
            ap->r[i] = bv;
            ap->x[i] = 0.0;
            pv.v[k] += __dmul_rn(bv, di) * bv;
         }
      }
      else
      {
#pragma unroll
         for (int k = 0; k < kPV; k++)
         {
            const size_t i = (size_t)k * N + n;
            const double bv = ap->b[i];
            ap->r[i] = bv;
            ap->x[i] = 0.0;
            pv.v[k] += __dmul_rn(bv, di) * bv;
         }
      }
   }
   return pv;
}

// ---- phase A: Y_E = B^T D B d_E over this workgroup's element batches ---------------------------
// Plane-per-thread contraction of vcg_apply_plane (lgh_vcg.hip): thread (c, qx, eb) owns one x-index of
// component c of element eb and keeps the whole (y, z) plane of that index in registers; only the two x
// contractions exchange data through LDS; the 1-D table sits in scalar registers.  Software pipeline: the
// element->node map runs two batches ahead, the gathers and the quadrature data one batch ahead.
// Returns this thread's part of (d, A d) of its component.
template <int D, int Q, int NEB>
__device__ __forceinline__ double pcg_phase_a(const PcgArgs *ap_, const PcgState *st_, double *smem_, const int it_,
                                           unsigned long long *tr_)
{
   constexpr int NQ = Q * Q * Q, ND = D * D * D, DD = D * D;
   constexpr int TE = kPV * Q; // threads per element
   constexpr int NT = TE * NEB;
   // LDS strides: consecutive (element, component) groups of a wave are skewed by 16 B modulo 128 B
   constexpr int CS = (ND + 3) & ~1;
   constexpr int CE = (DD * Q + 3) & ~1;
   constexpr int PER0 = kPV * (CS + CE);
   constexpr int PER = PER0 + ((6 - PER0 % 16) + 16) % 16;
   constexpr int GPT = (NEB * ND + NT - 1) / NT;
   constexpr int DPT = (NQ + TE - 1) / TE;
   constexpr int DSTR = (NQ + 7) & ~1;
   const PcgArgs *ap = uniform_ptr(ap_);
   const PcgState *st = uniform_ptr(st_);
   double *smem = uniform_ptr(smem_);
   unsigned long long *tr = uniform_ptr(tr_);
   const bool first = uniform_int(it_) == 1;
   const PcgWork wk = pcg_work();
   const int per = wk.per;
   const int NE = ap->NE;
   const size_t N = (size_t)ap->N;
   const int nbatch = (NE + NEB - 1) / NEB;
   const int nbx = (nbatch + 7) >> 3;
   const int b_lo = wk.worker ? wk.xcd * nbx + wk.jx : nbatch;
   const int b_hi = min(nbatch, (wk.xcd + 1) * nbx);
   const int *map = ap->map;
   const double *dinv = ap->dinv, *Dq = ap->Dq;
   const double *rv = ap->r, *dv = ap->d;
   double *YE = ap->YE;
   const size_t ye_stride = ap->ye_stride;

   const int tid = opaque_tid();
   const int eb = tid / TE, lt = tid - eb * TE;
   const int c = lt / Q, qx = lt - c * Q;
   double *sIn = smem + eb * PER + c * CS;
   double *sE = smem + eb * PER + kPV * CS + c * CE;
   double *sD = smem + NEB * PER + eb * DSTR;
   double beta[kPV];
#pragma unroll
   for (int k = 0; k < kPV; k++)
   {
      beta[k] = (first || st->done[k]) ? 0.0 : pcg_uniform(st->rz[k] / st->rz_prev[k]);
   }
   const bool mine = st->done[c] == 0;
   // 1-D table: scalar registers for the register-resident contractions, this thread's row / column for
   // the two x contractions
   double Bs[Q * D];
#pragma unroll
   for (int i = 0; i < Q * D; i++) { Bs[i] = pcg_uniform(ap->B[i]); }
   double bx[D], bt[Q];
#pragma unroll
   for (int dx = 0; dx < D; dx++) { bx[dx] = ap->B[qx + Q * dx]; }
#pragma unroll
   for (int q = 0; q < Q; q++) { bt[q] = ap->B[q + Q * (qx < D ? qx : 0)]; }

   int mi[GPT];
   auto load_map = [&](const int b) {
      const int e0 = b * NEB, nel = min(NEB, NE - e0);
#pragma unroll
      for (int k = 0; k < GPT; k++)
      {
         const int i = tid + k * NT;
         mi[k] = (i < nel * ND) ? map[(size_t)e0 * ND + i] : 0; // entries beyond the batch are not used
      }
   };
   double gz[kPV][GPT], gd[kPV][GPT], gi[GPT], dq[DPT];
   auto load_gather = [&]() {
#pragma unroll
      for (int k = 0; k < GPT; k++) { gi[k] = dinv[mi[k]]; }
#pragma unroll
      for (int k2 = 0; k2 < kPV; k2++)
      {
#pragma unroll
         for (int k = 0; k < GPT; k++)
         {
            gz[k2][k] = rv[(size_t)k2 * N + mi[k]];
            gd[k2][k] = first ? 0.0 : dv[(size_t)k2 * N + mi[k]];
         }
      }
   };
   auto load_dq = [&](const int bb) {
      const int e = bb * NEB + eb;
#pragma unroll
      for (int k = 0; k < DPT; k++)
      {
         const int j = lt + k * TE;
         dq[k] = (e < NE && j < NQ) ? Dq[(size_t)e * NQ + j] : 0.0;
      }
   };
   double dot = 0.0;
   int b = b_lo;
   if (b < b_hi) { load_map(b); }
   // Everything loaded so far (tables, map) is complete before the pipelined loop starts: the compiler's
   // wait-count analysis is loop-conservative and would otherwise put a wait for the NEXT batch's gathers
   // ahead of the first FMA of each batch (see vcg_apply_plane).
   __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
   if (b < b_hi)
   {
      load_gather();
      load_dq(b);
      if (b + per < b_hi) { load_map(b + per); }
   }
   for (; b < b_hi; b += per)
   {
      const int e = b * NEB + eb;
      const bool active = (e < NE) && mine;
      const int nit = min(NEB, NE - b * NEB) * ND;
      __syncthreads(); // previous batch finished with the LDS buffers
#pragma unroll
      for (int k = 0; k < GPT; k++)
      {
         const int i = tid + k * NT;
         if (i < nit)
         {
            const int el = i / ND, dd = i - el * ND;
#pragma unroll
            for (int k2 = 0; k2 < kPV; k2++)
            {
               smem[el * PER + k2 * CS + dd] = fma(beta[k2], gd[k2][k], __dmul_rn(gz[k2][k], gi[k]));
            }
         }
      }
#pragma unroll
      for (int k = 0; k < DPT; k++)
      {
         const int j = lt + k * TE;
         if (j < NQ) { sD[j] = dq[k]; }
      }
      const int bn = b + per;
      if (bn < b_hi)
      {
         load_gather();
         load_dq(bn);
         if (bn + per < b_hi) { load_map(bn + per); }
      }
      __syncthreads();
      // forward x
      double t[DD];
#pragma unroll
      for (int k = 0; k < DD; k++)
      {
         double u = 0.0;
#pragma unroll
         for (int dx = 0; dx < D; dx++) { u = fma(bx[dx], sIn[dx + D * k], u); }
         t[k] = u;
      }
      // forward y
      double w[Q][D];
#pragma unroll
      for (int qy = 0; qy < Q; qy++)
      {
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            double u = 0.0;
#pragma unroll
            for (int dy = 0; dy < D; dy++) { u = fma(Bs[qy + Q * dy], t[dy + D * dz], u); }
            w[qy][dz] = u;
         }
      }
      // per qy row: forward z, scale by the quadrature data, backward z
#pragma unroll
      for (int qy = 0; qy < Q; qy++)
      {
         double cz[Q];
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            double u = 0.0;
#pragma unroll
            for (int dz = 0; dz < D; dz++) { u = fma(Bs[qz + Q * dz], w[qy][dz], u); }
            cz[qz] = u * sD[qx + Q * (qy + Q * qz)];
         }
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            double u = 0.0;
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { u = fma(Bs[qz + Q * dz], cz[qz], u); }
            w[qy][dz] = u;
         }
      }
      // backward y; hand the plane over
#pragma unroll
      for (int dz = 0; dz < D; dz++)
      {
#pragma unroll
         for (int dy = 0; dy < D; dy++)
         {
            double u = 0.0;
#pragma unroll
            for (int qy = 0; qy < Q; qy++) { u = fma(Bs[qy + Q * dy], w[qy][dz], u); }
            sE[qx + Q * (dy + D * dz)] = u;
         }
      }
      __syncthreads();
      // backward x: thread dx = qx < D sums over the Q planes
      if (qx < D && active)
      {
         double *yc = YE + (size_t)c * ye_stride + (size_t)ND * e;
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
      if (tr && tid == 0)
      {
         const int kb = (b - b_lo) / per;
         if (kb < 8) { tr[14 + kb] = wall_clock64(); }
      }
   }
   return dot;
}

// ---- phase B: nodes ---------------------------------------------------------------------------------
// Pass q handles nodes n0 + (U q + u) NT + tid, u < U: all loads of a pass are issued before the first use.
// The pass body is straight-line code (uniform conditions are selects / store predicates, not branches:
// across the many blocks of a branchy body the register allocator spilled loaded values behind full
// waits).  A component that has converged is still loaded - it stops at most a few iterations before the
// others - and its stores are masked.
// x is only needed at the end: it is updated every second iteration with both terms,
//   x = (x + alpha_{it-1} d_{it-1}) + alpha_it d_it      (same roundings as two single updates),
// and a component that stops after an odd number of updates gets its last term after the loop.
__device__ __forceinline__ Vals4 pcg_phase_b(const PcgArgs *ap_, const PcgState *st_, const int it_)
{
   constexpr int U = 2;
   const PcgArgs *ap = uniform_ptr(ap_);
   const PcgState *st = uniform_ptr(st_);
   const int it = uniform_int(it_);
   const bool first = it == 1;
   const PcgWork wk = pcg_work();
   const int NT = blockDim.x, tid = opaque_tid();
   const int n0 = wk.worker ? ap->nstart[wk.w] : 0;
   const int n1 = wk.worker ? ap->nstart[wk.w + 1] : 0;
   const unsigned *ellz = ap->ellz;
   const uint8_t *essbits = ap->essbits;
   const double *dinv = ap->dinv;
   double *r = ap->r, *d = ap->d, *x = ap->x;
   const double *YE = ap->YE;
   const size_t ye_stride = ap->ye_stride;
   Vals4 pv;
#pragma unroll
   for (int i = 0; i < kPC; i++) { pv.v[i] = 0.0; }
   bool todo[kPV];
   double alpha[kPV], alpha_prev[kPV], beta[kPV];
#pragma unroll
   for (int k = 0; k < kPV; k++)
   {
      todo[k] = st->done[k] == 0;
      alpha[k] = todo[k] ? pcg_uniform(st->rz[k] / st->den[k]) : 0.0;
      alpha_prev[k] = todo[k] ? pcg_uniform(st->alpha_last[k]) : 0.0;
      beta[k] = (first || !todo[k]) ? 0.0 : pcg_uniform(st->rz[k] / st->rz_prev[k]);
   }
   const bool xupd = (it & 1) == 0, xload = xupd && it > 2;
   const unsigned rowb = 4u * (unsigned)ap->N, compb = 8u * (unsigned)ap->N;
   auto pass = [&](const int base, auto xu_tag) {
      constexpr bool XU = decltype(xu_tag)::value;
      unsigned nn[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; u++)
      {
         const int n = base + tid + u * NT;
         ok[u] = n < n1;
         nn[u] = (unsigned)(ok[u] ? n : n0);
      }
      unsigned ix[U][8];
#pragma unroll
      for (int u = 0; u < U; u++)
      {
#pragma unroll
         for (int j = 0; j < 8; j++) { ix[u][j] = ld_off(ellz, 4u * nn[u] + (unsigned)j * rowb); }
      }
      double di[U], ro[U][kPV], dol[U][kPV], xo[U][kPV];
      unsigned es[U];
#pragma unroll
      for (int u = 0; u < U; u++)
      {
         di[u] = ld_off(dinv, 8u * nn[u]);
         es[u] = ld_off(essbits, nn[u]);
#pragma unroll
         for (int k = 0; k < kPV; k++)
         {
            const unsigned vb = 8u * nn[u] + (unsigned)k * compb;
            ro[u][k] = ld_off((const double *)r, vb);
            dol[u][k] = ld_off((const double *)d, vb);
            xo[u][k] = 0.0;
            if (XU) { xo[u][k] = ld_off((const double *)x, vb); }
         }
      }
      double ye[U][kPV][8];
#pragma unroll
      for (int k = 0; k < kPV; k++)
      {
         const double *yc = YE + (size_t)k * ye_stride;
#pragma unroll
         for (int u = 0; u < U; u++)
         {
#pragma unroll
            for (int j = 0; j < 8; j++) { ye[u][k][j] = ld_off(yc, ix[u][j]); }
         }
      }
#pragma unroll
      for (int u = 0; u < U; u++)
      {
#pragma unroll
         for (int k = 0; k < kPV; k++)
         {
            const unsigned vb = 8u * nn[u] + (unsigned)k * compb;
            double zs = 0.0; // ascending contribution order; absent slots add 0.0 at the end
#pragma unroll
            for (int j = 0; j < 8; j++) { zs += ye[u][k][j]; }
            const double z_ = ((es[u] >> k) & 1u) ? 0.0 : zs;
            const double zold = __dmul_rn(ro[u][k], di[u]); // z of the previous iterate, not stored
            const double dnew = first ? zold : fma(beta[k], dol[u][k], zold);
            const double rnew = ro[u][k] - alpha[k] * z_;
            if (ok[u] && todo[k])
            {
               *ptr_off(d, vb) = dnew;
               *ptr_off(r, vb) = rnew;
               if (XU) // read back by this thread only
               {
                  const double x0 = xload ? xo[u][k] : 0.0;
                  *ptr_off(x, vb) = fma(alpha[k], dnew, fma(alpha_prev[k], first ? 0.0 : dol[u][k], x0));
               }
               pv.v[k] += rnew * __dmul_rn(rnew, di[u]);
            }
         }
      }
   };
   if (xupd)
   {
      for (int base = n0; base < n1; base += NT * U) { pass(base, std::true_type()); }
   }
   else
   {
      for (int base = n0; base < n1; base += NT * U) { pass(base, std::false_type()); }
   }
   return pv;
}

// ---- after the loop: components whose last update of x is still pending (odd number of updates) ------
__device__ __forceinline__ void pcg_phase_finish(const PcgArgs *ap_, const PcgState *st_)
{
   const PcgArgs *ap = uniform_ptr(ap_);
   const PcgState *st = uniform_ptr(st_);
   const PcgWork wk = pcg_work();
   const int NT = blockDim.x, tid = threadIdx.x;
   const size_t N = (size_t)ap->N;
   const int n0 = wk.worker ? ap->nstart[wk.w] : 0;
   const int n1 = wk.worker ? ap->nstart[wk.w + 1] : 0;
   for (int k = 0; k < kPV; k++)
   {
      if ((st->nupd[k] & 1) == 0) { continue; }
      const double al = pcg_uniform(st->alpha_last[k]);
      for (int n = n0 + tid; n < n1; n += NT)
      {
         const size_t i = (size_t)k * N + n;
         ap->x[i] = fma(al, ap->d[i], ap->x[i]);
      }
   }
}

template <int D, int Q, int NEB>
__global__ void __launch_bounds__(kPV *Q *NEB, 2)
pcg_solve_k(const PcgArgs *ap)
{
   constexpr int ND = D * D * D, DD = D * D, NQ = Q * Q * Q;
   constexpr int TE = kPV * Q, NT = TE * NEB, NW = (NT + 63) / 64;
   constexpr int CS = (ND + 3) & ~1;
   constexpr int CE = (DD * Q + 3) & ~1;
   constexpr int PER0 = kPV * (CS + CE);
   constexpr int PER = PER0 + ((6 - PER0 % 16) + 16) % 16;
   constexpr int DSTR = (NQ + 7) & ~1;
   __shared__ double smem[NEB * (PER + DSTR)];
   __shared__ double red[NW * kPC];
   __shared__ PcgState st;

   const int tid = threadIdx.x, bid = blockIdx.x;
   const unsigned long long clk0 = clock64(), wall0 = wall_clock64();
   const int nv = ap->nv;
   const int max_iter = ap->max_iter;
   const double rel_tol2 = ap->rel_tol2;
   unsigned long long *trace = ap->trace;
   unsigned epoch = 0;

   {
      const Vals4 pv = pcg_phase_init<ND>(ap);
      if (!pcg_barrier(ap, pv, ++epoch, red, &st, nullptr)) { return; }
      if (tid == 0)
      {
         int all = 1;
         for (int k = 0; k < kPC; k++)
         {
            const double nom = (k < nv) ? st.tot[k] : 0.0;
            st.rz[k] = st.rz_prev[k] = nom;
            st.den[k] = 0.0;
            st.alpha_last[k] = 0.0;
            st.nupd[k] = 0;
            st.iters[k] = 0;
            st.r0[k] = fmax(nom * rel_tol2, 0.0);
            st.done[k] = (k >= nv || nom < 0.0 || nom <= st.r0[k] || max_iter <= 0) ? 1 : 0;
            all = all && st.done[k];
         }
         st.all_done = all;
      }
      __syncthreads();
   }

   int it = 0;
   while (!st.all_done)
   {
      ++it;
      const bool first = (it == 1);
      unsigned long long *tr = (trace && tid == 0 && it <= kTrIter) ? trace + ((size_t)bid * kTrIter + (it - 1)) * kTrPer : nullptr;
      if (tr) { tr[0] = wall_clock64(); }
      {
         const double dot = pcg_phase_a<D, Q, NEB>(ap, &st, smem, it, tr);
         const int c = (tid % TE) / Q;
         Vals4 pv;
#pragma unroll
         for (int k = 0; k < kPC; k++) { pv.v[k] = (c == k) ? dot : 0.0; }
         if (tr) { tr[1] = wall_clock64(); }
         if (!pcg_barrier(ap, pv, ++epoch, red, &st, tr ? tr + 2 : nullptr)) { return; }
         if (tr) { tr[6] = wall_clock64(); }
         if (tid == 0)
         {
            int all = 1;
            for (int k = 0; k < kPC; k++)
            {
               if (!st.done[k])
               {
                  st.den[k] = st.tot[k];
                  if (st.tot[k] == 0.0) // breakdown, as upstream: final_iter = 0 before the loop, i inside it
                  {
                     st.done[k] = 1;
                     st.iters[k] = first ? 0 : it;
                  }
               }
               all = all && st.done[k];
            }
            st.all_done = all;
         }
         __syncthreads();
      }
      if (st.all_done) { break; }
      {
         const Vals4 pv = pcg_phase_b(ap, &st, it);
         if (tr) { tr[7] = wall_clock64(); }
         if (!pcg_barrier(ap, pv, ++epoch, red, &st, tr ? tr + 8 : nullptr)) { return; }
         if (tr) { tr[12] = wall_clock64(); }
         if (tid == 0)
         {
            int all = 1;
            for (int k = 0; k < kPC; k++)
            {
               if (!st.done[k])
               {
                  const double bn = st.tot[k];
                  st.alpha_last[k] = st.rz[k] / st.den[k]; // the alpha phase B used
                  st.nupd[k] = it;
                  st.rz_prev[k] = st.rz[k];
                  st.rz[k] = bn;
                  st.iters[k] = it;
                  if (bn < 0.0 || bn <= st.r0[k]) { st.done[k] = 1; }
                  else if (it >= max_iter) { st.done[k] = 2; } // ran out: final_iter = max_iter
               }
               all = all && st.done[k];
            }
            st.all_done = all;
         }
         __syncthreads();
      }
   }
   pcg_phase_finish(ap, &st);
   if (bid == 0 && tid == 0)
   {
      PcgResult *res = ap->res;
      for (int k = 0; k < kPC; k++)
      {
         res->iters[k] = st.iters[k];
         res->done[k] = st.done[k];
         res->rz[k] = st.rz[k];
         res->den[k] = st.den[k];
      }
      res->iterations = it;
      res->shader_clk = clock64() - clk0;
      res->wall_clk = wall_clock64() - wall0;
   }
}

// ---- host side -------------------------------------------------------------------------
constexpr size_t kGranWords = (size_t)2 * 64 * kSweepMax * 2 * kPC;
constexpr size_t kSyncWords = kGranWords + (kCtrShards + 1) * kCtrStride / 2; // granules, then the counters
struct PcgHost
{
   PcgResult *res_dev = nullptr;
   PcgArgs *args_dev = nullptr;
   unsigned long long *gran = nullptr;
   double *ye = nullptr;     // kPV planes of NE*ND + kYePad doubles; slot NE*ND of each plane stays 0.0
   unsigned *ellz = nullptr; // ELL transpose as byte offsets, absent entries -> 8*NE*ND
   uint8_t *essbits = nullptr;
   int *nstart = nullptr;    // node partition of the node phase for `grid` workgroups
   PcgArgs *args_host = nullptr; // pinned
   int grid = 0;
};

__global__ void __launch_bounds__(256)
pcg_essbits_k(const uint8_t *e0, const uint8_t *e1, const uint8_t *e2, const uint8_t *hmask, const double *owner, uint8_t *bits,
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
pcg_ellz_k(const int *__restrict__ ell, unsigned *__restrict__ ellz, const size_t n_have, const size_t n_all, const int zslot)
{
   const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_all) { return; }
   const int p = (i < n_have) ? ell[i] : -1; // rows beyond the mesh's valence: absent
   ellz[i] = 8u * (unsigned)(p < 0 ? zslot : p);
}

static bool pcg_kid_ok(const lgh_ctx *c)
{
   switch (c->kid)
   {
      case 0x322: case 0x334: case 0x346: return true;
   }
   return false;
}

bool pcg_available(const lgh_ctx *c)
{
   // 32-bit byte offsets into a vector set / a Y_E plane
   if ((size_t)c->N * 8 * kPV >= 0xffffffffull || ((size_t)c->NE * c->ND + kYePad) * 8 >= 0xffffffffull) { return false; }
   // Opt-in (LGH_PCG=1).  Measured at C2 (profiles/r2_pcg_*): one fused iteration takes 114-124 us against
   // 106 us for the K1 + K2 launches of lgh_vcg.hip - at two waves per SIMD the node phase and the element
   // phase each run slower inside the persistent kernel than as kernels of their own with their own
   // occupancy, which costs more than the two launch/tail overheads the fusion removes; and the
   // persistent grid leaves no room for the energy solve that otherwise overlaps the velocity solve.
   static const char *env = getenv("LGH_PCG");
   if (!(env && env[0] == '1')) { return false; }
   return c->dim == 3 && pcg_kid_ok(c) && c->multi == 0 && c->t_deg <= 8;
}

// Node phase: a node costs a fixed part (its vectors, the ELL row) plus a part per element contribution
// (measured: t = 4.2..5.3 ns + 1.7..1.8 ns * valence per node and workgroup, LGH_PCG_TRACE); with equal
// node counts the workgroups whose range covers element-boundary planes take 40 % longer and every barrier
// waits for them.  Ranges of equal cost instead, boundaries rounded to 16 nodes (128 B).
int partition_nodes_by_cost(lgh_ctx *c, const int W, int **out)
{
   const int N = c->N;
   std::vector<int> off((size_t)N + 1);
   LGH_HIP_CHECK(hipMemcpy(off.data(), c->t_off, off.size() * sizeof(int), hipMemcpyDeviceToHost));
   static const char *wenv = getenv("LGH_PCG_NODE_WEIGHT"); // fixed part in units of half a contribution; <0: equal counts
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
static int pcg_partition_nodes(lgh_ctx *c, PcgHost *h) { return partition_nodes_by_cost(c, (h->grid >> 3) << 3, &h->nstart); }

// ELL transpose of the restriction as byte offsets into a Y_E plane of NE*ND + pad doubles whose slot NE*ND is
// zero: absent contributions point there, so the gather needs no predicate
int make_ellz(lgh_ctx *c, unsigned **out)
{
   const size_t ell_n = (size_t)8 * c->N;
   LGH_HIP_CHECK(hipMalloc((void **)out, ell_n * sizeof(unsigned)));
   hipLaunchKernelGGL(pcg_ellz_k, dim3((unsigned)((ell_n + 255) / 256)), dim3(256), 0, nullptr, c->t_ell, *out,
                      (size_t)c->t_deg * c->N, ell_n, c->NE * c->ND);
   LGH_HIP_CHECK(hipGetLastError());
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
   hipLaunchKernelGGL(pcg_essbits_k, dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, nullptr, c->essmask[0], c->essmask[1],
                      c->essmask[2], hmask, c->multi ? c->owner : nullptr, *out, c->N);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

template <int D, int Q, int NEB> static int pcg_launch_neb(lgh_ctx *c, PcgHost *h, PcgArgs &a, const int wg_per_cu)
{
   auto kern = pcg_solve_k<D, Q, NEB>;
   if (h->grid <= 0)
   {
      int per_cu = 0, ncu = 256;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) { ncu = prop.multiProcessorCount; }
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kPV * Q * NEB, 0) != hipSuccess || per_cu <= 0)
      {
         set_error("pcg: occupancy query failed");
         return LGH_ERR_HIP;
      }
      // every workgroup must be resident (grid barrier): never more than the kernel was built for
      h->grid = std::min(std::min(per_cu, wg_per_cu) * ncu, 64 * kSweepMax);
      static const char *genv = getenv("LGH_PCG_GRID");
      if (genv && atoi(genv) > 0) { h->grid = std::min(h->grid, atoi(genv)); }
      h->grid = std::max(h->grid, 8);
   }
   if (!h->nstart)
   {
      const int rc = pcg_partition_nodes(c, h);
      if (rc) { return rc; }
   }
   a.nstart = h->nstart;
   *h->args_host = a;
   LGH_HIP_CHECK(hipMemcpyAsync(h->args_dev, h->args_host, sizeof(PcgArgs), hipMemcpyHostToDevice, c->stream));
   LGH_HIP_CHECK(hipMemsetAsync(h->gran, 0, kSyncWords * sizeof(unsigned long long), c->stream));
   hipLaunchKernelGGL(kern, dim3(h->grid), dim3(kPV * Q * NEB), 0, c->stream, (const PcgArgs *)h->args_dev);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

template <int D, int Q> static int pcg_launch(lgh_ctx *c, PcgHost *h, PcgArgs &a)
{
   constexpr int NEB0 = (256 / (kPV * Q)) > 0 ? (256 / (kPV * Q)) : 1;
   constexpr int NEB = (Q == 6) ? NEB0 - 1 : NEB0; // as launch_vcg_plane: two workgroups per CU
   // One workgroup of twice that size per CU: two workgroups of a CU do not share it evenly (the older
   // waves win the issue arbitration; the younger workgroup finishes a phase 40 % later, LGH_PCG_TRACE) and
   // every barrier waits for the slower one.  LGH_PCG_WIDE=0: two workgroups per CU (A/B).
   static const char *wenv = getenv("LGH_PCG_WIDE");
   if (wenv && wenv[0] == '0') { return pcg_launch_neb<D, Q, NEB>(c, h, a, 2); }
   return pcg_launch_neb<D, Q, 2 * NEB>(c, h, a, 1);
}

// B, X: dim*N (byNODES).  force_E != nullptr: B and X are outputs (see vcg_init_force_k); otherwise B holds
// the eliminated right-hand sides and X is overwritten (zero initial guess, laghos_solver.cpp:338, :382).
int pcg_solve(lgh_ctx *c, double *B, double *X, double rel_tol, int max_iter, int iters[3], const double *force_E)
{
   if (!pcg_available(c)) { return LGH_ERR_UNSUPPORTED; }
   const size_t N = (size_t)c->N;
   if (!c->pcg)
   {
      PcgHost *h = new PcgHost();
      c->pcg = h;
      LGH_HIP_CHECK(hipMalloc((void **)&h->res_dev, sizeof(PcgResult)));
      LGH_HIP_CHECK(hipMemset(h->res_dev, 0, sizeof(PcgResult)));
      LGH_HIP_CHECK(hipMalloc((void **)&h->args_dev, sizeof(PcgArgs)));
      LGH_HIP_CHECK(hipHostMalloc((void **)&h->args_host, sizeof(PcgArgs), hipHostMallocDefault));
      LGH_HIP_CHECK(hipMalloc((void **)&h->gran, kSyncWords * sizeof(unsigned long long)));
      const size_t ye_n = (size_t)kPV * ((size_t)c->NE * c->ND + kYePad);
      LGH_HIP_CHECK(hipMalloc((void **)&h->ye, ye_n * sizeof(double)));
      LGH_HIP_CHECK(hipMemset(h->ye, 0, ye_n * sizeof(double)));
      int rcb = make_ellz(c, &h->ellz);
      if (rcb) { return rcb; }
      rcb = make_essbits(c, &h->essbits);
      if (rcb) { return rcb; }
      if (!c->vcg_vec)
      {
         LGH_HIP_CHECK(hipMalloc((void **)&c->vcg_vec, 3 * kPV * N * sizeof(double))); // r, d (, yL of the multi-rank path)
         LGH_HIP_CHECK(hipMemset(c->vcg_vec, 0, 3 * kPV * N * sizeof(double)));
      }
      LGH_HIP_CHECK(hipStreamSynchronize(nullptr));
   }
   PcgHost *h = (PcgHost *)c->pcg;
   PcgArgs a;
   memset(&a, 0, sizeof(a));
   a.NE = c->NE;
   a.N = c->N;
   a.nv = kPV;
   a.B = c->B;
   a.Dq = c->massD;
   a.map = c->h1map;
   a.ellz = h->ellz;
   a.essbits = h->essbits;
   for (int k = 0; k < kPV; k++) { a.ess[k] = c->essmask[k]; }
   a.dinv = c->dinvV;
   a.FE = force_E;
   a.b = B;
   a.x = X;
   a.r = c->vcg_vec;
   a.d = c->vcg_vec + kPV * N;
   a.YE = h->ye;
   a.ye_stride = (size_t)c->NE * c->ND + kYePad;
   a.rel_tol2 = rel_tol * rel_tol;
   a.max_iter = max_iter;
   a.gran = h->gran;
   a.ctr = (unsigned int *)(h->gran + kGranWords);
   a.res = h->res_dev;
   LGH_HIP_CHECK(hipMemsetAsync(h->res_dev, 0, sizeof(PcgResult), c->stream));
   // debug: LGH_PCG_TRACE=<file> dumps the per-workgroup phase time stamps of the last solve (10 ns ticks)
   static const char *trace_path = getenv("LGH_PCG_TRACE");
   static unsigned long long *trace_dev = nullptr;
   const size_t trace_n = (size_t)64 * kSweepMax * kTrIter * kTrPer;
   if (trace_path && !trace_dev) { (void)hipMalloc((void **)&trace_dev, trace_n * sizeof(unsigned long long)); }
   if (trace_dev) { (void)hipMemsetAsync(trace_dev, 0, trace_n * sizeof(unsigned long long), c->stream); }
   a.trace = trace_dev;
   // the pinned argument block is reused: the previous solve has completed (its result was read synchronously)
   int rc = LGH_ERR_UNSUPPORTED;
   kt_begin(c, LGH_KERNEL_PCG);
   switch (c->kid)
   {
      case 0x322: rc = pcg_launch<2, 2>(c, h, a); break;
      case 0x334: rc = pcg_launch<3, 4>(c, h, a); break;
      case 0x346: rc = pcg_launch<4, 6>(c, h, a); break;
   }
   kt_end(c, LGH_KERNEL_PCG);
   if (rc) { return rc; }
   PcgResult *hr = (PcgResult *)(c->host_pinned + 32);
   static_assert(sizeof(PcgResult) <= 32 * sizeof(double), "pinned staging too small");
   LGH_HIP_CHECK(hipMemcpyAsync(hr, h->res_dev, sizeof(PcgResult), hipMemcpyDeviceToHost, c->stream));
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   if (hr->error)
   {
      set_error("pcg: grid barrier timed out (a workgroup of the persistent grid was not resident)");
      return LGH_ERR_HIP;
   }
   static const char *clk_env = getenv("LGH_PCG_CLOCK");
   if (clk_env && hr->wall_clk)
   {
      fprintf(stderr, "pcg: %d iterations, %.1f us, shader clock %.0f MHz\n", hr->iterations, 0.01 * hr->wall_clk,
              100.0 * (double)hr->shader_clk / (double)hr->wall_clk);
   }
   if (trace_dev)
   {
      std::vector<unsigned long long> t(trace_n);
      (void)hipMemcpy(t.data(), trace_dev, trace_n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      FILE *f = fopen(trace_path, "w");
      if (f)
      {
         for (int b = 0; b < h->grid; b++)
         {
            for (int i = 0; i < kTrIter && i < hr->iterations; i++)
            {
               fprintf(f, "%d %d", b, i + 1);
               for (int k = 0; k < kTrPer; k++) { fprintf(f, " %llu", t[((size_t)b * kTrIter + i) * kTrPer + k]); }
               fprintf(f, "\n");
            }
         }
         fclose(f);
      }
   }
   int mx = 0;
   for (int k = 0; k < kPV; k++)
   {
      iters[k] = hr->iters[k];
      mx = std::max(mx, iters[k]);
   }
   c->vcg_last = mx;
   c->pcg_iterations += hr->iterations;
   return LGH_OK;
}

void pcg_free(lgh_ctx *c)
{
   PcgHost *h = (PcgHost *)c->pcg;
   if (!h) { return; }
   (void)hipFree(h->res_dev);
   (void)hipFree(h->args_dev);
   (void)hipFree(h->gran);
   (void)hipFree(h->ye);
   (void)hipFree(h->ellz);
   (void)hipFree(h->essbits);
   (void)hipFree(h->nstart);
   if (h->args_host) { (void)hipHostFree(h->args_host); }
   delete h;
   c->pcg = nullptr;
}

} // namespace lgh
