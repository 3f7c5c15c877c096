// lgh_vcg.hpp — state, argument block and grid reductions shared by the kernels of the lockstep velocity solve
// (lgh_vcg.hip: column / plane forms and the node kernel K2; lgh_vcg_mfma.hip: the matrix-core form of K1).
#pragma once
#include "lgh_common.hpp"

namespace lgh
{

constexpr int kVC = 3; // velocity components handled in lockstep
constexpr int kTraceRec = 16; // debug (LGH_VCG_TRACE): 64-bit words per workgroup record of K1

struct VcgScalars
{
   double rz[kVC], rz_prev[kVC], den[kVC], r0[kVC];
   double rel_tol2;
   double alpha_last[kVC]; // vcg_update_p_k: alpha of the latest completed update of component c
   int done[kVC], iters[kVC], first;
   int all_done, pad;
   int nupd[kVC], pad2;    // vcg_update_p_k: iteration of that update (x lags one update behind when it is odd)
};

// Several ranks (see cg_pending_update, lgh_mass.hip): the sums of den and (r, z) over the ranks complete between
// the kernels, and the decisions they feed are taken by the next kernel of the sequence - every workgroup evaluates
// the same predicate on the same reduced values, thread 0 of workgroup 0 also commits the outcome (write-through
// stores; a commit only writes values under which the predicate stays true).
// K1 of iteration iter >= 2: outcome of the update of iteration iter - 1.  live[c]: component c still iterates.
// Returns false when none does.
__device__ __forceinline__ bool vcg_pending_update(VcgScalars *s, const int iter, const bool commit, bool live[kVC])
{
   bool any = false;
#pragma unroll
   for (int c = 0; c < kVC; c++)
   {
      bool dn = s->done[c] != 0;
      if (!dn)
      {
         const double rz = s->rz[c];
         dn = rz < 0.0 || rz <= s->r0[c];
         if (commit)
         {
            __hip_atomic_store(&s->iters[c], iter - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (dn)
            {
               __hip_atomic_store(&s->done[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               __hip_atomic_store(&s->rz[c], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               __hip_atomic_store(&s->den[c], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
         }
      }
      live[c] = !dn;
      any = any || !dn;
   }
   if (!any && commit) { __hip_atomic_store(&s->all_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
   return any;
}
// K2: breakdown of component c (den == 0 after the sum over the ranks), as upstream
__device__ __forceinline__ bool vcg_pending_den(VcgScalars *s, const int c, const bool commit)
{
   const bool brk = s->den[c] == 0.0;
   if (brk && commit)
   {
      __hip_atomic_store(&s->done[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&s->rz[c], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   }
   return brk;
}

// three-value variant of grid_reduce_last_block (sum): partials[v*stride + i]
__device__ __forceinline__ bool grid_sum3_last_block(const double bp[kVC], double *partials,
                                                     const unsigned stride, unsigned int *ticket,
                                                     double *red, double total[kVC])
{
   const int tid = threadIdx.x;
   const int nthr = blockDim.x;
   const unsigned int nblk = gridDim.x, bid = blockIdx.x;
   const unsigned s = bid % kShards;
   const unsigned cnt = nblk / kShards + ((s < nblk % kShards) ? 1u : 0u);
   const unsigned nsh = nblk < kShards ? nblk : kShards;
   unsigned int *t1 = ticket + s * kTicketStride;
   unsigned int *t2 = ticket + kShards * kTicketStride;
   __shared__ unsigned int s_flag;
   if (tid == 0)
   {
#pragma unroll
      for (int v = 0; v < kVC; v++)
      {
         __hip_atomic_store(&partials[(size_t)v * stride + bid], bp[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned a = __hip_atomic_fetch_add(t1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_flag = (a == cnt - 1) ? 1u : 0u;
   }
   __syncthreads();
   if (!s_flag) { return false; }
   double ssum[kVC];
#pragma unroll
   for (int v = 0; v < kVC; v++)
   {
      double acc = 0.0;
      for (unsigned int i = tid; i < cnt; i += nthr)
      {
         acc += __hip_atomic_load(&partials[(size_t)v * stride + s + kShards * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ssum[v] = block_sum(acc, red);
      __syncthreads();
   }
   const unsigned shard_off = nblk; // shard sums follow the block partials of each value
   if (tid == 0)
   {
#pragma unroll
      for (int v = 0; v < kVC; v++)
      {
         __hip_atomic_store(&partials[(size_t)v * stride + shard_off + s], ssum[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(t1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned b = __hip_atomic_fetch_add(t2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_flag = (b == nsh - 1) ? 1u : 0u;
      if (s_flag) { __hip_atomic_store(t2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
   }
   __syncthreads();
   if (!s_flag) { return false; }
#pragma unroll
   for (int v = 0; v < kVC; v++)
   {
      double acc = 0.0;
      for (unsigned int i = tid; i < nsh; i += nthr)
      {
         acc += __hip_atomic_load(&partials[(size_t)v * stride + shard_off + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      total[v] = block_sum(acc, red);
      __syncthreads();
   }
   return true;
}

// Single-level variant for the persistent K1: its <= 1024 workgroups leave their
// loops spread over ~15 us (profiles/r1_k1_block_timestamps.txt), so one ticket word
// is not contended, and the last arrival sums all partials in one pass.  Saves the
// second store / ticket / reload round trip of the sharded form at the kernel's tail.
__device__ __forceinline__ bool grid_sum3_last_block_flat(const double bp[kVC], double *partials,
                                                          const unsigned stride, unsigned int *ticket,
                                                          double *red, double total[kVC])
{
   const int tid = threadIdx.x;
   const int nthr = blockDim.x;
   const unsigned int nblk = gridDim.x, bid = blockIdx.x;
   unsigned int *t2 = ticket + kShards * kTicketStride; // the top word of the slot
   __shared__ unsigned int s_flag;
   if (tid == 0)
   {
#pragma unroll
      for (int v = 0; v < kVC; v++)
      {
         __hip_atomic_store(&partials[(size_t)v * stride + bid], bp[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned a = __hip_atomic_fetch_add(t2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_flag = (a == nblk - 1) ? 1u : 0u;
      if (s_flag) { __hip_atomic_store(t2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
   }
   __syncthreads();
   if (!s_flag) { return false; }
   double acc[kVC] = {0.0, 0.0, 0.0};
   for (unsigned int i = tid; i < nblk; i += nthr)
   {
#pragma unroll
      for (int v = 0; v < kVC; v++)
      {
         acc[v] += __hip_atomic_load(&partials[(size_t)v * stride + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
   }
#pragma unroll
   for (int v = 0; v < kVC; v++)
   {
      total[v] = block_sum(acc[v], red);
      __syncthreads();
   }
   return true;
}

struct VcgArgs
{
   int NE, N;
   const double *B, *Dq;
   const int *map;
   const unsigned *mapb;  // map as byte offsets into a node vector (8 * node): the matrix-core K1 (lgh_vcg_mfma.hip)
   int map_xrows;         // 1: the D1D nodes of every x-row of every element are consecutive node numbers (checked at set-up)
   const int *ell;
   int deg;
   const uint8_t *ess[kVC];
   const double *dinv, *owner;
   const double *b;       // kVC*N right-hand sides (byNODES)
   double *x;             // kVC*N solutions
   double *r, *d;         // kVC*N each (z = r/diag is recomputed where it is used)
   double *YE;            // kVC * NE*ND
   size_t ye_stride;      // NE*ND
   double *yL;            // kVC*N (unfused path)
   VcgScalars *s;
   double *partials;
   unsigned stride;
   unsigned int *ticket;
   int iter, multi;
   unsigned long long *trace; // debug (LGH_VCG_TRACE=file): per-workgroup time stamps of K1
   // multi-rank: nodes shared with other ranks take their (halo-summed) A d from yL,
   // all others gather it from the E-vector as on one rank
   const uint8_t *hmask;      // N flags, or nullptr
   const int *sh_node;        // the shared nodes
   int n_shared;
   // vcg_update_p_k
   const unsigned *ellz;      // ell as byte offsets into a Y_E plane, absent entries -> its zero slot NE*ND
   const uint8_t *essbits;    // bit k: node essential for component k
   const int *nstart;         // node range of block w: [nstart[w], nstart[w+1]), balanced by cost
   int k2_skip;               // vcg_update_p_k: skip ELL slots no node of the wavefront uses (LGH_K2_SKIP=0: fetch all 8)
};

// lgh_vcg_mfma.hip
bool vcg_mfma_available(lgh_ctx *c);
void launch_vcg_mfma(lgh_ctx *c, const VcgArgs &a);
// lgh_vcg_slab.hip
bool vcg_slab_available(lgh_ctx *c);
void launch_vcg_slab(lgh_ctx *c, const VcgArgs &a);

} // namespace lgh
