// lgh_vcg.hpp — state, argument block and grid reductions shared by the kernels of the lockstep velocity solve
// (lgh_vcg.hip: column / plane / Kronecker forms of K1 and the node kernel K2; lgh_vcg_slab.hip: the slab form of K1).
#pragma once
#include "lgh_common.hpp"

namespace lgh
{

constexpr int kVC = 3; // velocity components handled in lockstep
constexpr int kSlabMinElements = 20000;  // default dispatch of the slab-form K1 (vcg_k1_form): 39.3 against 48.5 us at 32^3, 248 against 383 at 64^3
constexpr int kTraceRec = 16; // debug (LGH_VCG_TRACE): 64-bit words per workgroup record of K1

struct VcgScalars
{
   double rz[kVC], rz_prev[kVC], den[kVC];
   double den_e;   // lockstep energy CG (lgh_mass.hip): its (d, M d), a fourth scalar right behind den[] - it rides on the halo messages with them
   double r0[kVC];
   double rel_tol2;
   double alpha_last[kVC]; // vcg_update_p_k: alpha of the latest completed update of component c
   int done[kVC], iters[kVC], first;
   int all_done, pad;
   int nupd[kVC], pad2;    // vcg_update_p_k: iteration of that update (x lags one update behind when it is odd)
   double alpha_hist[2][kVC]; // rz_limbs mode: alpha of iteration it in [it & 1] - written by workgroup 0 of K2(it) while other
                              // workgroups of the same launch may still be reading the alpha of it - 1 out of the other row
   double rzh[2][kVC];        // rz_limbs mode: (r, z) after iteration j in [j & 1], written by workgroup 0 of K1(j + 1)
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
   const double *B, *Dq;  // Dq: quadrature data of the mass operator, value(q, e) = Dq[q + dqs e] * Se[e] (mass_data, lgh_mass.hip) in the
   const double *Se;      // plane and slab forms of K1; the column and matrix-core forms read DqFull[q + NQ e]
   const double *M1;      // 1-D mass tile B^T diag(w) B (D1D x D1D) when the mass data is compact AND the rule a tensor product: the Kronecker form of K1 (lgh_vcg_slab.hip), nullptr otherwise
   const double *w1;      // vcg_apply_plane_ho<.., SEP = true>: one-dimensional weights of a tensor-product rule (compact data only), nullptr otherwise
   const double *DqFull;
   int dqs;
   const int *map;
   const unsigned *mapb;  // map as byte offsets into a node vector (8 * node): the slab K1 (lgh_vcg_slab.hip)
   int map_xrows;         // 1: the D1D nodes of every x-row of every element are consecutive node numbers (checked at set-up)
   unsigned *queue;       // slab-form K1, dynamic schedule: one set counter per XCD range, 128 bytes apart (zero between launches)
   long long *rzl;        // rz_limbs mode: three sets of kLimbWords words, exact accumulators of (r, z) (see vcg_rz_commit), or nullptr
   const long long *rzl_peers; // several ranks, rz_limbs mode: the other ranks' words of the set K2 just added into (n_rz_peers blocks of
   int n_rz_peers;             // kLimbWords, exchange_words) - whoever folds a set adds them to the own words: an exact all-reduce
   long long *limbs;      // exact accumulators of (d, A d): two sets of kLimbWords words (slab-form K1), or nullptr (ticketed fold of workgroup partials)
   int den_limbs;         // 1: K1 only adds into set (iter & 1) of limbs - no ticket, no last workgroup; K2 folds the set itself (exact_den) and clears the other one
   const int *ell;
   int deg;
   const uint8_t *ess[kVC];
   const double *dinv, *owner;
   int reset;             // the first kernel of a solve (vcg_init_*) also does what vcg_set_tol_k did in front of it: clears the exact accumulators
   double reset_tol2;     // and the set counters (workgroup 0) and resets the scalars of the solve with rel_tol^2 = reset_tol2 (the thread that writes them)
   const int *ncaller;    // the solve runs in the library's own node numbering (lgh_order.hip): internal node m is the caller's node ncaller[m] - the
                          // caller's vectors (b in, the right-hand side and x out) are indexed through it; nullptr: the caller's numbering is the solve's
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
   int store_wait;        // slab-form K1 (A/B, LGH_SLAB_STORE_WAIT): every wavefront waits for the stores of a pass before it starts the next one
   int ye_wide;           // slab-form K1: the three planes of Y_E exceed 4 GB - per-set 64-bit store base instead of 32-bit offsets from Y_E
   // slab-form K1, merged E-vector layout (round 5; LGH_SLAB_MERGE=0: nullptr): one word per set of five zones - bits 0..30 the
   // set's slice of a Y_E plane in units of 64 doubles, bit 31 set when the five zones are x-neighbours ("x-chain": zone i + 1's
   // dx = 0 nodes are zone i's dx = 3 nodes, checked on the map) and the set is stored as 16 rows (dz, dy) of 16 x-nodes with the
   // shared x-faces already summed (256 doubles) instead of 5 x 64 element-local values; ell / ellz / nstart then describe THAT layout
   const unsigned *settab;
   // several ranks: kernels that fill the send buffer themselves (one launch less per exchange)
   HaloPackTables hp;
   const int *sh_off;     // (= hp.sh_off: CSR of the unique shared nodes over the entries of the neighbour lists)
   int pack_halo;         // vcg_gather_list_k also writes every shared-node sum to its places in the send buffer
   int nx_den;            // ... and its workgroup 0 the local (d, A d) behind every neighbour's block (piggy-backed sums)
   int pack_rz;           // the last workgroup of vcg_update_p_k puts the local (r, z) into the send buffer of the scalar exchange
};

// ---- exact, order-independent sums of doubles (the (d, A d) of the slab-form K1) -------------------------------
// Why: a ticketed last-workgroup fold puts ~4 us of dependent memory round trips behind the slowest workgroup of
// K1, and a sum whose bits must not depend on which wavefront took which set of elements rules out scheduling the
// sets dynamically.  Integer addition is associative: every addend v, known to lie in (-2^(E-1), 2^(E-1)), is split
// into kLimbs signed pieces of 32 bits - limb j weighs 2^(E - 32 (j + 1)) - which are added into 64-bit integer
// accumulators (registers, LDS, finally fire-and-forget device atomics: no ticket, no fold, nobody is "last").
// The window of 128 bits below 2^E loses at most 2^(E-128) per addend; up to 2^31 addends fit the accumulators.
// E comes from a quantity every workgroup of the producing and of the consuming kernel reads alike (rz of the
// iteration: (d, A d) <= lambda_max(D^-1 M) (r, z) <= 64 (r, z) for the Jacobi-preconditioned mass matrix), a
// non-finite or out-of-window addend sets a sticky flag that turns the sum into NaN.
constexpr int kLimbs = 4;      // 64-bit accumulators per sum
constexpr int kLimbShards = 4; // copies of the accumulators (workgroup b adds to shard b % kLimbShards): ~64 atomics per word and launch
constexpr int kLimbWords = kLimbShards * kVC * kLimbs + 8; // the accumulators, then the flag word (padded)
static_assert(kLimbWords == kLsWordsPerSet && kLsWord > kLimbShards * kVC * kLimbs && kLsWord < kLimbWords, "the lockstep energy CG keeps its (r, r) in a padding word of a set");
__device__ __forceinline__ int exact_scale(const double rz) // E for sums bounded by 64 rz (margin 2^5)
{
   int e;
   (void)frexp(rz, &e); // rz = m 2^e, m in [0.5, 1)
   return e + 12;
}
// returns false when v does not fit the window (|v| >= 2^(E-1), NaN, inf); acc is then left alone
__device__ __forceinline__ bool exact_add(long long (&acc)[kLimbs], const double v, const int E)
{
   double x = ldexp(v, 32 - E); // exact; |x| < 2^31 required
   if (!(fabs(x) < 2147483648.0)) { return false; }
   double f = floor(x);
   acc[0] += (long long)(int)f; // signed top limb
   x = (x - f) * 4294967296.0;  // exact: fractional part, scaled by 2^32
#pragma unroll
   for (int j = 1; j < kLimbs; j++)
   {
      f = floor(x);
      acc[j] += (long long)(unsigned)f;
      x = (x - f) * 4294967296.0;
   }
   return true;
}
// the value of the accumulators: carries first (integers), then one deterministic chain of fp operations
__device__ __forceinline__ double exact_value(const long long (&L)[kLimbs], const int E)
{
   long long l[kLimbs];
#pragma unroll
   for (int j = 0; j < kLimbs; j++) { l[j] = L[j]; }
#pragma unroll
   for (int j = kLimbs - 1; j > 0; j--)
   {
      const long long carry = l[j] >> 32; // arithmetic shift: floor division by 2^32
      l[j] -= carry * 4294967296LL;       // now in [0, 2^32)
      l[j - 1] += carry;
   }
   // l[0] may need more than 53 bits on very large meshes: split it so that every piece converts exactly
   const long long h = l[0] >> 26;
   const long long m = l[0] - h * 67108864LL; // [0, 2^26)
   double s = 0.0;
#pragma unroll
   for (int j = kLimbs - 1; j > 0; j--) { s += ldexp((double)l[j], E - 32 * (j + 1)); }
   s += ldexp((double)m, E - 32);
   s += ldexp((double)h, E - 32 + 26);
   return s;
}

// (d, A d) of component k out of one set of accumulators (den_limbs: every workgroup of K2 folds the dozen words itself
// - integers, so all of them get the same bits - instead of one last workgroup of K1 doing it while the chip waits).
__device__ __forceinline__ double exact_den(const long long *__restrict__ set, const int k, const double rz)
{
   long long l4[kLimbs];
#pragma unroll
   for (int j = 0; j < kLimbs; j++) { l4[j] = 0; }
#pragma unroll
   for (int sh = 0; sh < kLimbShards; sh++)
   {
#pragma unroll
      for (int j = 0; j < kLimbs; j++) { l4[j] += set[sh * (kVC * kLimbs) + kLimbs * k + j]; }
   }
   const long long bad = set[kLimbShards * kVC * kLimbs];
   return bad ? __builtin_nan("") : exact_value(l4, exact_scale(rz));
}
// the same with the scale given (the (r, z) sets of the rz_limbs mode use one scale per solve)
__device__ __forceinline__ double exact_fold(const long long *__restrict__ set, const int k, const int E)
{
   long long l4[kLimbs];
#pragma unroll
   for (int j = 0; j < kLimbs; j++) { l4[j] = 0; }
#pragma unroll
   for (int sh = 0; sh < kLimbShards; sh++)
   {
#pragma unroll
      for (int j = 0; j < kLimbs; j++) { l4[j] += set[sh * (kVC * kLimbs) + kLimbs * k + j]; }
   }
   const long long bad = set[kLimbShards * kVC * kLimbs];
   return bad ? __builtin_nan("") : exact_value(l4, E);
}

// the same out of a wavefront's registers: lane l holds word l of the set (l <= kLimbShards * kVC * kLimbs: the flag word)
__device__ __forceinline__ long long lane_word(const long long w, const int l)
{
   const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)w, l);
   const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)w >> 32), l);
   return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double exact_fold_lanes(const long long w, const int k, const int E)
{
   static_assert(kLimbShards * kVC * kLimbs < 64, "one word per lane");
   long long l4[kLimbs];
#pragma unroll
   for (int j = 0; j < kLimbs; j++) { l4[j] = 0; }
#pragma unroll
   for (int sh = 0; sh < kLimbShards; sh++)
   {
#pragma unroll
      for (int j = 0; j < kLimbs; j++) { l4[j] += lane_word(w, sh * (kVC * kLimbs) + kLimbs * k + j); }
   }
   const long long bad = lane_word(w, kLimbShards * kVC * kLimbs);
   return bad ? __builtin_nan("") : exact_value(l4, E);
}

// ---- rz_limbs mode (one rank, slab K1 + bounded-grid K2, exact accumulators): K2 has no last workgroup either.
// Round 4 measured what the ticketed fold of (r, z) at the end of K2 costs: two dependent atomic round trips and two
// dependent reads behind the slowest workgroup (profiles/r4_k2_tail.txt).  Now every workgroup of K2(i) adds its
// share of R_i = (r, z) after iteration i into set i % 3 of `rzl` (exact integer limbs, fire-and-forget atomics; scale
// exact_scale(R_(i-1)), which both sides know), and K1(i + 1) folds that set: every wavefront out of its own registers
// (one vector load in its prologue batch, v_readlane; four wavefronts per CU) - integers, so every wavefront of every
// workgroup gets the same bits and takes the same decisions, whatever it sees of the flags workgroup 0 is committing:
//   K1(i) folds set (i-1) % 3 -> R_(i-1); R_(i-2) comes from VcgScalars::rzh[(i-2) & 1]; workgroup 0 writes
//   rzh[(i-1) & 1] = R_(i-1) and commits what K2(i) and the host read (done, iters, all_done) - values under which the
//   predicate the other workgroups evaluate stays true; it also clears set i % 3 (last read by K1(i - 2));
//   K2(i) reads R_(i-1), R_(i-2) and the flags from the scalars (complete: K1(i) has ended) and folds nothing;
//   convergence of component k after iteration j = R_j <= r0[k];
//   a one-thread kernel does what K1(last + 1) would do before the host looks (vcg_rz_finish_k).
__device__ __forceinline__ bool vcg_rz_converged(const int it, const double cur, const double r0)
{
   return it > 1 && (cur < 0.0 || cur <= r0); // (as the last workgroup of the ticketed K2 decides)
}
// what workgroup 0 of K1(it) leaves for K2(it), the next K1 and the host; cur[k] = R_(it-1)
__device__ __forceinline__ void vcg_rz_commit(VcgScalars *s, const int it, const double (&cur)[kVC], const bool (&done)[kVC])
{
   int all = 1;
#pragma unroll
   for (int k = 0; k < kVC; k++)
   {
      s->rzh[(it - 1) & 1][k] = cur[k];
      if (!s->done[k])
      {
         __hip_atomic_store(&s->iters[k], it - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         if (done[k]) { __hip_atomic_store(&s->done[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
      all = all && done[k];
   }
   if (all) { __hip_atomic_store(&s->all_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

// 64-bit integer sum over the 64 lanes of a full wavefront (DPP, as wave_sum); the total is returned in every lane
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#define LGH_I64_STEP(CTRL_, MASK_)                                                                             \
   {                                                                                                            \
      const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v & 0xffffffffLL), CTRL_, MASK_, 0xF, false); \
      const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL_, MASK_, 0xF, false);                   \
      v += (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);            \
   }
   LGH_I64_STEP(0x111, 0xF) // row_shr:1
   LGH_I64_STEP(0x112, 0xF) // row_shr:2
   LGH_I64_STEP(0x114, 0xF) // row_shr:4
   LGH_I64_STEP(0x118, 0xF) // row_shr:8
   LGH_I64_STEP(0x142, 0xA) // row_bcast:15 into rows 1 and 3
   LGH_I64_STEP(0x143, 0xC) // row_bcast:31 into rows 2 and 3
#undef LGH_I64_STEP
   const int lo = __builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffLL), 63);
   const int hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
   return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

// lgh_vcg_slab.hip
bool vcg_slab_available(lgh_ctx *c);
void launch_vcg_slab(lgh_ctx *c, const VcgArgs &a);

} // namespace lgh
