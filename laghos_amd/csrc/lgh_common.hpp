// lgh_common.hpp — context, error handling and device reduction helpers shared by
// the HIP translation units of liblaghos_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/laghos_hip.h"

namespace lgh
{

void set_error(const char *fmt, ...);

#define LGH_HIP_CHECK(expr)                                                              \
   do                                                                                    \
   {                                                                                     \
      hipError_t e_ = (expr);                                                            \
      if (e_ != hipSuccess)                                                              \
      {                                                                                  \
         lgh::set_error("HIP error %s at %s:%d: %s", hipGetErrorName(e_), __FILE__,      \
                        __LINE__, #expr);                                                \
         return LGH_ERR_HIP;                                                             \
      }                                                                                  \
   } while (0)

#define LGH_CHECK_ARG(cond)                                                              \
   do                                                                                    \
   {                                                                                     \
      if (!(cond))                                                                       \
      {                                                                                  \
         lgh::set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__);          \
         return LGH_ERR_ARG;                                                             \
      }                                                                                  \
   } while (0)

constexpr int kWave = 64;          // gfx950 wavefront

// Device-resident scalars of one CG solve (SURVEY §3.2 names).
struct CgScalars
{
   double rz;       // (r, z) current: nom, later betanom
   double rz_prev;  // previous (r, z): denominator of beta
   double den;      // (d, A d)
   double r0;       // max(nom*rel_tol^2, abs_tol^2)
   double nom0;     // initial nom
   double rel_tol2; // rel_tol^2 (set by host before the init kernel)
   int done;        // 1: converged / breakdown, later kernels become no-ops
   int iters;       // final_iter
   int first;       // 1 until the first direction has been formed (beta = 0)
   int pad;
};

struct Timers
{
   bool enabled = true;
   hipEvent_t ev[2] = {nullptr, nullptr};
   double t[4] = {0, 0, 0, 0}; // cgH1, cgL2, force, qdata (seconds)
   long c[3] = {0, 0, 0};      // H1iter, L2iter, quad_tstep
};

// event pairs around the launches of one kernel (lgh_ktime_begin / _end)
struct KTime
{
   int which = -1, max = 0, n = 0;
   std::vector<hipEvent_t> ev; // 2 per sample
};

struct Comm; // RCCL state (lgh_comm.hip)

// The library's own order of the zones and numbering of the nodes (lgh_order.hip): found from the element -> node map at
// lgh_create by face adjacency, nothing about the caller's numbering is assumed.  identity: the caller's order already is
// that order (or no structured block was found, or LGH_ORDER=0) - nothing is permuted anywhere.
struct MeshOrder
{
   bool structured = false, identity = true;
   int components = 0, extent[3] = {0, 0, 0};
   std::vector<int> zorder;   // internal zone i is the caller's zone zorder[i]
   std::vector<int> nnum;     // the caller's node n is internal node nnum[n]
   std::vector<int> ncaller;  // internal node m is the caller's node ncaller[m]
   int *zorder_d = nullptr, *ncaller_d = nullptr, *nnum_d = nullptr; // the same on the device (nullptr when identity)
};

} // namespace lgh

struct lgh_ctx
{
   int dim, NE, D1D, Q1D, L1D, ND, NQ, NL, N, H1V, L2V;
   int kid; // (dim<<8)|(D1D<<4)|Q1D, the reference's kernel id
   bool visc, vort;
   double cfl, h0, h1order;
   double q_tiny_grad;   // QUpdate: threshold of the wave-uniform eigen-decomposition shortcut (lgh_qupdate.hip)
   int device;
   hipStream_t stream;
   // second stream + fork/join events: the energy solve overlaps the velocity solve (lgh_api.hip)
   hipStream_t stream2;
   int on_stream2;       // 1 while `stream` holds the second stream (energy solve beside the velocity solve): reductions use the communicator's second channel
   hipEvent_t ev_fork, ev_join;
   void *l2run;          // state of a split L2 solve (lgh_mass.hip)
   const double *accel_src; // dim*N acceleration source of SolveVelocity (source_type 2) or nullptr
   int e_async;          // 1: lgh_solve_energy_begin enqueued the solve, 2: deferred to _end
   int e_lockstep;       // lgh_solve_energy_begin has set the energy CG up for LOCKSTEP with the velocity CG on the one stream and communicator
   int ls_side = 0;      // ... with its kernels on the second stream (they exchange nothing themselves), ordered against the velocity iteration by
   hipEvent_t ls_ev[4][4] = {}; // events: [0] main -> side after the word exchange, [1] side -> main after the apply, [2] main -> side after the
                         // halo sums, [3] side -> main after the update; rings of four (iteration & 3).  LGH_LOCKSTEP_STREAM2=1 (default: one stream)
   long ls_stats[3] = {0, 0, 0}; // lockstep energy solves; their iterations enqueued inside the velocity solve; ... and after it (lgh_energy_lockstep_stats)
                         // (several ranks without a second channel; lgh_mass.hip "lockstep"): lgh_solve_velocity interleaves its iterations
   int e_polled, e_iters; // the enqueued solve has already been completed (energy_overlap_poll, from inside the velocity solve): its iteration count
   struct { const double *S, *v; double *dS, *e_rhs; const double *src; double tol; int maxit; } e_args;
   bool own_stream;

   // tables (device): B_h1/G_h1 [q + Q*d], B_l2 [q + Q*l], weights [NQ]
   double *B, *G, *Bl, *W, *gamma;
   // element restriction: gather map, and its transpose in CSR form
   int *h1map;   // NE*ND
   int *t_off;   // N+1   (CSR transpose, used by the multi-component force gather)
   int *t_idx;   // NE*ND: E-vector positions (e*ND + d) contributing to node
   int *t_ell;   // t_deg*N: the same in ELL format, [k*N + n], -1 = none
   int t_deg;    // max contributions per node (8 for hexes, 4 for quads)
   uint8_t *essmask[3]; // N each (0/1)
   int *ess[3];
   int ess_count[3];
   double *owner; // N or nullptr
   int cur_ess;

   // QuadratureData + mass PA data
   double *stressJinvT, *Jac0inv, *rho0DetJ0w, *massD, *diagV, *dinvV;
   double *Jac0inv_soa;  // plane-major copy of Jac0inv for coalesced reads in QUpdate
   double *Jac0inv_e;    // dim*dim per zone: Jac0inv where it is the same at every point of a zone (an affine initial zone; jac0_compact)
   int jac0_compact;     // 1: the row-form update reads Jac0inv_e (checked at lgh_setup_rho0detj0), 0: the point values
   // Rank-1 form of the mass data: where detJ0 and rho0 are constant inside every element (affine elements, piecewise
   // constant density - every mesh and problem of data/), massD[q + NQ e] = W[q] * massS[e] to the last bit or two;
   // the mass kernels then read 8 bytes per element instead of 8 NQ.  -1: not looked at yet (set-up, lgh_mass_D handed
   // out), 0: no (massD as stored), 1: yes (lgh_mass.hip mass_data).  LGH_MASS_RANK1=0 keeps massD.
   double *massS, *ones_ne;
   double *M1h = nullptr, *M1l = nullptr; // 1-D mass tiles B^T diag(w1d) B of the H1 (D1D x D1D) and L2 (L1D x L1D) bases: the Kronecker form of the
                                          // mass operators for compact mass data on a tensor-product rule (nullptr: no such rule, or LGH_MASS_KRON=0)
   double *w1d = nullptr;   // one-dimensional weights when the rule is a tensor product, W[qx + Q (qy + Q qz)] = w[qx] w[qy] w[qz] (checked by lgh_create; nullptr otherwise)
   int mass_rank1;
   double *dt_est_dev;   // running min of the point-wise estimate: [0] the estimate, then kDtSlots partial minima 128 bytes apart (the row-form
                         // update sends its candidates there by non-returning atomic min; lgh_get_dt_est folds them into [0])
   // Force products formed inside the fused QUpdate.  Validity is by construction, not by address: `fused_*_valid` says
   // that the product belongs to the quadrature data as it stands (set by lgh_qupdate, cleared by everything that can
   // change stressJinvT: lgh_reset_quadrature_data, lgh_qdata_stressJinvT, lgh_set_fused_forces, the set-up), and
   // F^T v is only used for a velocity that compares equal, element for element, to the one it was formed from.
   double *erhs_q;       // L2V: F^T v of ...
   double *v_snap;       // ... H1V: the velocity block of the state of the last lgh_qupdate (copy)
   double *force_e_q;    // NE*ND*dim: F.1 as E-vector (3D)
   int fused_ftv_valid, fused_f1_valid;
   unsigned long qgen;   // counts lgh_qupdate / invalidations (lgh_quadrature_generation)
   int *dev_flags;       // 8 device ints, one owner each: [0] / [2] "v differs from v_snap" of the current lgh_solve_energy (main /
                         // second stream), [1] "x differs from ones" of lgh_force_mult, [3] "mass table is not W[q]*s_e" (mass_data),
                         // [4] "an energy right-hand side was poisoned" (erhs_take_or_poison_k; read and cleared by lgh_get_dt_est),
                         // [5] "Jac0inv varies inside a zone" (jac0_compact_k, lgh_setup_rho0detj0)
   double *ones_l2;      // L2V ones: the operator's own `one` (laghos_solver.cpp:170-171), allocated on first use
   int stress_store;            // lgh_qupdate_store_stress: 1 (default) = lgh_qupdate writes the nine stressJinvT planes; 0 = the stress stays in registers
   int stress_current;          // stressJinvT holds the stress of the current quadrature data (consumers refuse it otherwise)
   int fused_forces_off;        // lgh_set_fused_forces(ctx, 0): the update forms no force products (measurement / A-B)
   // scratch
   double *XE;           // max(L2V, NE*ND*dim)
   double *YE;           // NE*ND*dim
   double *cg_r, *cg_z, *cg_d0, *cg_d1, *cg_y; // max(N, L2V)
   double *partials;     // 4 reduction slots of part_stride block partials each
   int part_stride;      // slot size: >= max #blocks of any reducing launch + kShards
   unsigned int *tickets;// 4 reduction slots of lgh::kTicketSlot counters (sharded tickets)
   lgh::CgScalars *cgs;  // device
   double *scal;         // small device scalar pool (16 doubles)
   double *host_pinned;  // pinned host staging, 64 doubles: [0..7] scalar CG / misc, [8] dt, [32..] lockstep CG scalars
   unsigned long long look_token; // counts the host looks that wait for a token in pinned memory (host_wait_token)
   double *host_pinned_dev; // the same memory as the device addresses it: scalars a host look needs are written there by the kernel that
                            // finishes them (no copy kernel between it and the stream synchronisation)

   // lockstep velocity CG (lgh_vcg.hip), allocated on first use
   void *vcg_s;
   double *vcg_vec, *vcg_partials;
   unsigned int *vcg_tickets;
   unsigned vcg_stride;
   int vcg_last;
   int vcg_grid;         // persistent grid size of the K1 kernel (one resident wave)
   int b_h1_sym, b_l2_sym; // the 1-D H1 / L2 table is mirror symmetric, B[q,d] = B[Q-1-q, D-1-d] (to 1e-14: the round-off of its evaluation): kernels may hold half of it
   int ncu;              // CUs of this context's device (0: not asked yet)
   int vcg_variant;      // LGH_VCG_VARIANT: which K1 form vcg_solve launches (lgh_vcg.hip); -1: by kernel id and mesh size
   int slab_wps, slab_wide, slab_exact, slab_dyn; // A/B switches of the slab-form K1 (LGH_SLAB_WPS / _WIDE / _EXACT / _DYN, read by lgh_create)
   void *vcg_aux;        // tables of the node kernel K2 (lgh_vcg.hip VcgAux), allocated on first use

   lgh::Timers timers;
   lgh::KTime *ktime;
   int cg_last_iters[2][3]; // iteration count of the previous solve per (space, component)
   lgh::Comm *comm;
   int nranks, rank;
   int multi;            // 1: run the multi-rank code path (nranks > 1, or LGH_FORCE_MULTI=1 for testing on one GPU)
   unsigned long long *q_trace_dev; // debug (LGH_Q_TRACE): stage stamps of the quadrature update's workgroups, NE records
   int q_trace_n, q_trace_calls;
   void *order;          // lgh::MeshOrder (lgh_order.hip)
   unsigned long mass_gen; // counts changes of the mass data / Jacobi diagonal (the velocity solve keeps copies in its own numbering)
};

namespace lgh
{

// ---- wave / block reductions (wave64, DPP) ----------------------------------
// Workgroups here are Q*Q*NEB threads, not always a multiple of 64, so the last wave may be partial (`nact` live
// lanes, wave-uniform).  The reduction runs on the DPP data path: row_shr 1, 2, 4, 8 inside the rows of 16 lanes,
// then row_bcast 15 / 31 across them - 3 instructions per step instead of the ~13 of a ds_bpermute shuffle with
// its lane guard (these trees are ~80 of the QUpdate's ~2100 vector instructions per wave, and the tail of every
// CG kernel).  A lane whose source lies outside its row, in a masked row or in the dead part of a partial wave takes
// the neutral `old` operand instead, so no guard is needed; the total forms in the last live lane and is read
// from there into every lane.  Fixed order: bit-reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(const double old, const double src)
{
   const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, 0xF, false);
   const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, 0xF, false);
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_bcast_f64(const double v, const int src_lane /* uniform */)
{
   const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
   const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
   return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v, const int lane, const int nact)
{
   (void)lane;
   v += dpp_f64<0x111, 0xF>(0.0, v); // row_shr:1
   v += dpp_f64<0x112, 0xF>(0.0, v); // row_shr:2
   v += dpp_f64<0x114, 0xF>(0.0, v); // row_shr:4
   v += dpp_f64<0x118, 0xF>(0.0, v); // row_shr:8
   v += dpp_f64<0x142, 0xA>(0.0, v); // row_bcast:15 into rows 1 and 3
   v += dpp_f64<0x143, 0xC>(0.0, v); // row_bcast:31 into rows 2 and 3
   return wave_bcast_f64(v, nact - 1);
}
__device__ __forceinline__ double wave_min(double v, const int lane, const int nact)
{
   (void)lane;
   v = fmin(v, dpp_f64<0x111, 0xF>(v, v));
   v = fmin(v, dpp_f64<0x112, 0xF>(v, v));
   v = fmin(v, dpp_f64<0x114, 0xF>(v, v));
   v = fmin(v, dpp_f64<0x118, 0xF>(v, v));
   v = fmin(v, dpp_f64<0x142, 0xA>(v, v));
   v = fmin(v, dpp_f64<0x143, 0xC>(v, v));
   return wave_bcast_f64(v, nact - 1);
}

// a value that is the same in every lane, moved to scalar registers (an FMA can take it as its SGPR operand)
__device__ __forceinline__ double uniform_f64(const double v)
{
   const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
   const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
   return __hiloint2double(hi, lo);
}

// Sum over the block; result valid in thread 0.  `red` = LDS scratch of >= 16 doubles.
__device__ __forceinline__ double block_sum(double v, double *red)
{
   const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
   const int nthr = blockDim.x * blockDim.y * blockDim.z;
   const int lane = tid & 63, wid = tid >> 6, nw = (nthr + 63) >> 6;
   const int nact = min(64, nthr - (wid << 6));
   v = wave_sum(v, lane, nact);
   __syncthreads();
   if (lane == 0) { red[wid] = v; }
   __syncthreads();
   double s = 0.0;
   if (tid == 0)
   {
      for (int w = 0; w < nw; w++) { s += red[w]; }
   }
   return s;
}
// Three sums at once (the three velocity components of the lockstep CG): one pair of barriers instead of three
// pairs plus separators; the same wave sums added in the same order as three block_sum calls - the same bits.
// `red` = LDS scratch of >= 48 doubles; results valid in thread 0.
__device__ __forceinline__ void block_sum3(const double v0, const double v1, const double v2, double *red, double out[3])
{
   const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
   const int nthr = blockDim.x * blockDim.y * blockDim.z;
   const int lane = tid & 63, wid = tid >> 6, nw = (nthr + 63) >> 6;
   const int nact = min(64, nthr - (wid << 6));
   const double a0 = wave_sum(v0, lane, nact), a1 = wave_sum(v1, lane, nact), a2 = wave_sum(v2, lane, nact);
   __syncthreads();
   if (lane == 0)
   {
      red[3 * wid + 0] = a0;
      red[3 * wid + 1] = a1;
      red[3 * wid + 2] = a2;
   }
   __syncthreads();
   out[0] = out[1] = out[2] = 0.0;
   if (tid == 0)
   {
      for (int w = 0; w < nw; w++)
      {
         out[0] += red[3 * w + 0];
         out[1] += red[3 * w + 1];
         out[2] += red[3 * w + 2];
      }
   }
   __syncthreads(); // red is free again (the grid reductions that follow reuse it)
}
__device__ __forceinline__ double block_min(double v, double *red)
{
   const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
   const int nthr = blockDim.x * blockDim.y * blockDim.z;
   const int lane = tid & 63, wid = tid >> 6, nw = (nthr + 63) >> 6;
   const int nact = min(64, nthr - (wid << 6));
   v = wave_min(v, lane, nact);
   __syncthreads();
   if (lane == 0) { red[wid] = v; }
   __syncthreads();
   double s = v;
   if (tid == 0)
   {
      s = red[0];
      for (int w = 1; w < nw; w++) { s = fmin(s, red[w]); }
   }
   return s;
}

// ---- deterministic grid-wide reductions without a second launch -----------------
// Every block publishes its partial with an agent-scope (write-through) atomic
// store, drains, and takes a ticket; the last block to arrive re-reads all
// partials with agent-scope loads in block order (fixed summation tree ->
// bit-reproducible).  Both sides use 8-byte agent atomics (MI355X guide,
// Guideline 16), so no L2 write-back fence is needed.
//
// One device-scope counter serialises at ~12 ns per arrival (guide: "fanin"), so
// thousands of blocks on ONE counter cost tens of microseconds - measured here as
// the whole duration of the first CG kernels.  The ticket is therefore sharded:
// block b arrives on shard b % kShards (one cache line each); the last arriver of
// a shard arrives on the top counter; the last of those owns the final fold.
// Counters are reset by their last arrivers, so a slot is reusable by the next
// launch on the stream.
//
// Memory ordering.  The partial is published with a relaxed agent-scope 8-byte atomic store (sc1:
// write-through, nothing left in this XCD's L2), then `s_waitcnt vmcnt(0)` completes it before the ticket
// is taken; the last arriver reads the partials with relaxed agent-scope 8-byte atomic loads (sc1: served
// past the L1, and never stale in an L2 - the L2s snoop write-through traffic).  That is the "8-byte agent
// atomics on both sides" form of MI355X_MICROARCH.md / cdna_hip_programming.md Guideline 16; it holds by
// the behaviour of gfx9-family vmcnt (which counts stores) and sc1, not by the C++ memory model.  The
// model-conforming form - a RELEASE fetch_add by the producers and an ACQUIRE one by the last block - costs
// an L2 write-back (buffer_wbl2) per workgroup on gfx950, microseconds for each of the thousands of blocks
// of the node kernels whose vector stores are still dirty in the L2; it is not used.  The guard below keeps
// this file from being built for a target where the argument does not hold (vmcnt does not count stores from
// gfx10 on).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "lgh_common.hpp: the grid reductions rely on gfx9-family vmcnt / sc1 semantics (see comment)"
#endif
constexpr unsigned kShards = 64;
constexpr unsigned kTicketStride = 32;                               // uints: 128 B apart
constexpr unsigned kTicketSlot = (kShards + 1) * kTicketStride;      // uints per reduction slot

// Returns true in ALL threads of the globally last block; `total` is then valid in
// thread 0.  `partials` must hold gridsize + kShards doubles: block partials, then
// (at offset `shard_off`) one folded partial per shard.  Fold order is fixed:
// shard s folds blocks s, s+kShards, ... through the block tree, the last block
// folds the shard sums in shard order -> bit-reproducible, and the serial tail
// after the last arrival is two memory round trips, not gridsize/blocksize.
template <bool IS_MIN>
__device__ __forceinline__ bool grid_reduce_last_block(double block_partial /* thread 0 */,
                                                       double *partials, const unsigned shard_off,
                                                       unsigned int *ticket, double *red, double &total)
{
   const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
   const int nthr = blockDim.x * blockDim.y * blockDim.z;
   const unsigned int nblk = gridDim.x * gridDim.y * gridDim.z;
   const unsigned int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
   const unsigned s = bid % kShards;
   const unsigned cnt = nblk / kShards + ((s < nblk % kShards) ? 1u : 0u);
   const unsigned nsh = nblk < kShards ? nblk : kShards;
   unsigned int *t1 = ticket + s * kTicketStride;
   unsigned int *t2 = ticket + kShards * kTicketStride;
   const double ident = IS_MIN ? INFINITY : 0.0;
   __shared__ unsigned int s_flag;
   if (tid == 0)
   {
      __hip_atomic_store(&partials[bid], block_partial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned a = __hip_atomic_fetch_add(t1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_flag = (a == cnt - 1) ? 1u : 0u;
   }
   __syncthreads();
   if (!s_flag) { return false; }
   // last arriver of shard s: fold the shard's block partials
   double v = ident;
   for (unsigned int i = tid; i < cnt; i += nthr)
   {
      const double p = __hip_atomic_load(&partials[s + kShards * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = IS_MIN ? fmin(v, p) : v + p;
   }
   const double ssum = IS_MIN ? block_min(v, red) : block_sum(v, red);
   __syncthreads();
   if (tid == 0)
   {
      __hip_atomic_store(&partials[shard_off + s], ssum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(t1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned b = __hip_atomic_fetch_add(t2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_flag = (b == nsh - 1) ? 1u : 0u;
      if (s_flag) { __hip_atomic_store(t2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
   }
   __syncthreads();
   if (!s_flag) { return false; }
   v = ident;
   for (unsigned int i = tid; i < nsh; i += nthr)
   {
      const double p = __hip_atomic_load(&partials[shard_off + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = IS_MIN ? fmin(v, p) : v + p;
   }
   total = IS_MIN ? block_min(v, red) : block_sum(v, red);
   return true;
}
// The shard sums live right after the block partials of a slot: every reduction
// slot of the context is part_stride + kShards doubles (lgh_create).
__device__ __forceinline__ bool grid_sum_last_block(double bp, double *partials, unsigned int *ticket,
                                                    double *red, double &total)
{
   const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
   return grid_reduce_last_block<false>(bp, partials, nblk, ticket, red, total);
}
__device__ __forceinline__ bool grid_min_last_block(double bp, double *partials, unsigned int *ticket,
                                                    double *red, double &total)
{
   const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
   return grid_reduce_last_block<true>(bp, partials, nblk, ticket, red, total);
}

// block index -> work chunk so that blocks resident on one XCD (observed:
// block b runs on XCD b % 8) sweep a contiguous range: neighbouring elements
// share H1 nodes, so their gathers hit the same XCD-private L2.
__device__ __forceinline__ int xcd_swizzle(const int b, const int nblocks)
{
   const int per = nblocks >> 3; // blocks per XCD (floor)
   const int main = per << 3;
   if (b >= main) { return b; } // tail stays in place
   return (b & 7) * per + (b >> 3);
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// ---- lgh_order.hip
int mesh_order_build(lgh_ctx *c, const int *map_host);
void mesh_order_free(lgh_ctx *c);
// the internal order when it differs from the caller's, else nullptr
inline const MeshOrder *mesh_order(const lgh_ctx *c)
{
   const MeshOrder *o = (const MeshOrder *)c->order;
   return (o && !o->identity) ? o : nullptr;
}
int order_gather_nodes(lgh_ctx *c, const double *in, double *out, int ncomp);  // out[k N + m] = in[k N + ncaller[m]]
int order_scatter_nodes(lgh_ctx *c, const double *in, double *out, int ncomp); // out[k N + ncaller[m]] = in[k N + m]
int order_gather_bytes(lgh_ctx *c, const uint8_t *in, uint8_t *out);
int order_zone_blocks(lgh_ctx *c, const double *in, double *out, int per, bool to_caller); // blocks of `per` doubles per zone

// ---- cross-TU launch helpers (implemented in the .hip files) ------------------
int force_mult_E(lgh_ctx *c, const double *sJit, const double *xE, double *yE);
int force_mult_t_L(lgh_ctx *c, const double *sJit, const double *v_h1, double *y_l2, const int *guard = nullptr);
int force_mult_t_E(lgh_ctx *c, const double *sJit, const double *vE, double *y_l2);
int h1_transpose_gather(lgh_ctx *c, int ncomp, const double *YE, double *yL);
int mass_apply_h1(lgh_ctx *c, const double *x, double *y, bool eliminate);
int mass_apply_l2(lgh_ctx *c, const double *x, double *y);
int mass_apply_E(lgh_ctx *c, int space, const double *xE, double *yE);
int mass_assemble_diag(lgh_ctx *c);
int qupdate_form(lgh_ctx *c); // 1: row form of the 3D quadrature update (lgh_qrows.hpp), 0: point form
// the quadrature data of the mass operators as (table, element stride in doubles, per-element factor): value(q, e) = Dq[q + dqs e] * Se[e]
int mass_data(lgh_ctx *c, const double **Dq, int *dqs, const double **Se);
int l2_mass_form(lgh_ctx *c, int *form, int *compact); // kernel of the L2 mass apply: 2 Kronecker, 1 plane, 0 column (lgh_l2_mass_form)
int vcg_solve(lgh_ctx *c, double *B, double *X, double rel_tol, int max_iter, int iters[3],
              const double *force_E = nullptr);
bool vcg_fused_init_ok(const lgh_ctx *c);
// tables of the node kernel K2 (lgh_vcg.hip)
int partition_nodes_by_cost(lgh_ctx *c, int W, int **out, const std::vector<int> *valence = nullptr);
int make_ellz(lgh_ctx *c, unsigned **out, const int *ell = nullptr, int deg = 0);
int vcg_test_merged_faces(lgh_ctx *c, unsigned char *mask, long *n_merged); // lgh_test_vcg_merged_faces
int vcg_layout_stats(lgh_ctx *c, long out[4]); // lgh_vcg_layout_stats
int make_essbits(lgh_ctx *c, uint8_t **out);
void vcg_free(lgh_ctx *c);
constexpr int kDtSlots = 256, kDtSlotStride = 16; // doubles; slot s at dt_est_dev[kDtSlotStride * (1 + s)]
constexpr int kYePad = 16; // doubles behind every Y_E plane of the CG; the first one (slot NE*ND) stays 0.0
bool vcg_available(const lgh_ctx *c);
int vcg_test_k1(lgh_ctx *c, const double *r, const double *d_old, const double rz[3], const double rz_prev[3], int first,
                double *YE_out, double den_out[3]); // one launch of K1 (test hook)
int vcg_test_k2(lgh_ctx *c, int it, const double *YE_in, double *r, double *d, double *x, const double den[3], const double rz[3],
                const double rz_prev[3], const double alpha_prev[3], double rz_out[3], int *deferred_x); // one launch of K2 (test hook)
int vcg_k1_form(lgh_ctx *c); // 0 column, 2 plane, 3 matrix cores, 4 slab, -1 none
// multi-rank: flags / list of the nodes shared with other ranks (nullptr / 0 without neighbours)
void comm_shared_nodes(const lgh_ctx *c, const uint8_t **hmask, const int **sh_node, int *n_shared);
// Completes the energy solve that lgh_solve_energy_begin enqueued on the second stream - its host looks and, if it has not
// converged, its further iterations - from INSIDE the velocity solve, while the main stream still has the velocity
// solve's first chunk of iterations in front of it: lgh_solve_energy_end then only joins the streams (round 6: the looks of
// the energy solve used to sit between the end of the velocity solve and the RK combination, with the GPU idle).
int energy_overlap_poll(lgh_ctx *c);
// Host look without a copy kernel and without the wake-up latency of a blocking stream synchronisation: the one-thread
// kernel that finishes the scalars writes them, then `token`, into pinned host memory (a.k.a. host_pinned_dev); the host
// spins on the token word (falls back to hipStreamSynchronize after a while; LGH_SPIN=0: always synchronise).
int host_wait_token(lgh_ctx *c, volatile unsigned long long *word, unsigned long long token);
// ---- energy CG in lockstep with the velocity CG (several ranks, one stream, one communicator; round 6).  Its two dot products
// per iteration ride on exchanges the velocity iteration makes anyway: (d, M d) as a fourth scalar behind the halo messages
// (VcgScalars::den_e), (r, r) as a double in word kLsWord of the accumulator-word exchange.  No exchange of its own.
constexpr int kLsWord = 50;             // (a padding word of an accumulator set: lgh_vcg.hpp checks it is one)
constexpr int kLsWordsPerSet = 56;      // = kLimbWords (checked there): stride of the peers' sets
struct LockstepWords { const long long *own; const long long *peers; int n_peers, before; }; // before: peers of lower rank
bool l2_lockstep_possible(lgh_ctx *c);  // the L2 apply runs as the Kronecker kernel with the fused update (lgh_mass.hip)
int cg_l2_begin_lockstep(lgh_ctx *c, const double *b, double *x, double rel_tol, int max_iter); // set-up + initial residual, no iteration
int l2_lockstep_limit(lgh_ctx *c);      // iterations worth interleaving: the previous solve's count
int l2_lockstep_apply(lgh_ctx *c, int it, const LockstepWords &prev, double *den_mirror);
int l2_lockstep_update(lgh_ctx *c, int it, const double *den_src, long long *word_out);
int l2_lockstep_fold(lgh_ctx *c, int it, const LockstepWords &last); // commits the outcome of iteration it (what the next apply would)
int cg_l2_end_lockstep(lgh_ctx *c, int *iters); // the rest of the solve on its own (exchanges of its own), the host look
int comm_ranks_before(const lgh_ctx *c); // neighbours whose rank is lower than this rank's
bool vcg_lockstep_ready(const lgh_ctx *c); // the velocity solve of this context exchanges accumulator words and packs its halo itself
int cg_l2_begin(lgh_ctx *c, const double *b, double *x, double rel_tol, int max_iter);
int cg_l2_end(lgh_ctx *c, int *iters);
void cg_l2_free(lgh_ctx *c);
int cg_solve(lgh_ctx *c, int space, const double *b, double *x, double rel_tol, int max_iter,
             int *iters, bool x_is_zero);
int qupdate(lgh_ctx *c, const double *S);
int setup_rho0detj0(lgh_ctx *c, const double *x0, const double *rho0_l2, const double *rho0_q,
                    double *volume);
int interp_energy(lgh_ctx *c, int which, const double *vec, double *result);
int tg_source_2d(lgh_ctx *c, const double *S, double *out);
int test_eig(lgh_ctx *c, int dim, int n, const double *A, double *lambda, double *vec);
int test_singular(lgh_ctx *c, int dim, int n, const double *A, double *sv);
int test_sqrt(lgh_ctx *c, int n, const double *x, double *y);
int vec_set(lgh_ctx *c, double *y, double a, long n);
int vec_axpby(lgh_ctx *c, double *z, double a, const double *x, double b, const double *y, long n);
int vec_axpby_pair(lgh_ctx *c, double *z1, double a1, const double *x1, double b1, double *z2, double a2, const double *x2, double b2,
                   const double *y, long n);
int vec_neg_inplace(lgh_ctx *c, double *y, long n);
int vec_zero_list(lgh_ctx *c, double *y, const int *list, int n);
int vec_dot(lgh_ctx *c, const double *x, const double *y, const double *w, long n, double *dev_out);
// packed: the caller's own kernel has already put the node values / scalars into the send buffer of the main channel
// (HaloPackTables: what it needs for that) - the exchange then starts without a pack kernel
struct HaloNodeAlias { const int *nodes, *sh_node; }; // the exchange's node lists in another numbering of the SAME nodes (the velocity solve's own)
int halo_sum(lgh_ctx *c, double *v, int ncomp, double *extra = nullptr, int nextra = 0, bool packed = false, const HaloNodeAlias *alias = nullptr);
void comm_node_lists(const lgh_ctx *c, const int **nodes, int *total, const int **sh_node, int *n_shared);
struct HaloPackTables
{
   const int *sh_off, *sh_src; // CSR over the unique shared nodes: entries of the concatenated neighbour lists (-1: own value)
   const int *pos, *cnt;       // entry j -> offset of its node value in the buffer (component c at + c * cnt[j])
   const int *base, *ncnt;     // neighbour k -> start of its block, nodes in it (scalars sit behind ncomp * ncnt[k] values)
   double *sbuf;
   int n_nbr;
};
bool comm_pack_tables(const lgh_ctx *c, HaloPackTables *t); // false: no neighbours / no tables
bool halo_can_piggyback(const lgh_ctx *c);
bool comm_second_channel(const lgh_ctx *c); // reductions may run on the context's second stream as well
// Exact all-reduce of integer accumulator words (lgh_vcg.hpp: the (r, z) of the lockstep solve) in ONE message round and
// without a kernel: every rank sends its `nwords` 64-bit words to every peer straight out of the accumulators and
// receives theirs into its peer buffer (comm_word_peers: block k = neighbour k); whoever folds the sum adds own and
// peer words - integers, so every rank gets the same bits whatever the arrival order.  All-pairs partitions only.
int exchange_words(lgh_ctx *c, const long long *src, int nwords);
int comm_word_peers(lgh_ctx *c, int nwords, const long long **peers, int *n_peers); // (allocates the peer buffer on first use)
int allreduce_dev(lgh_ctx *c, double *dev, int count, int op, bool packed = false); // packed: as halo_sum (sums that travel as one exchange with every peer only)

// bracket one launch of kernel `id` with an event pair when sampling is on
inline void kt_begin(lgh_ctx *c, int id)
{
   KTime *k = c->ktime;
   if (k && k->which == id && k->n < k->max) { (void)hipEventRecord(k->ev[2 * k->n], c->stream); }
}
inline void kt_end(lgh_ctx *c, int id)
{
   KTime *k = c->ktime;
   if (k && k->which == id && k->n < k->max) { (void)hipEventRecord(k->ev[2 * k->n + 1], c->stream); k->n++; }
}

// one sample around a scope: closed on every way out (error returns included)
struct KtScope
{
   lgh_ctx *c;
   int id;
   KtScope(lgh_ctx *c_, int id_) : c(c_), id(id_) { kt_begin(c, id); }
   ~KtScope() { kt_end(c, id); }
   KtScope(const KtScope &) = delete;
   KtScope &operator=(const KtScope &) = delete;
};

void timer_start(lgh_ctx *c);
void timer_stop(lgh_ctx *c, int which);
// roctx range around a region of the host timeline, named after the reference's Caliper regions
// (/root/reference/laghos_solver.cpp:353-356 "SolveVelocity-ForcePA", :387-390 "SolveVelocity-CGVMass", :472-475
// "SolveEnergy-ForcePA", :480-483 "SolveEnergy-CGEMass", :1358 "QUpdate-UpdateQuadratureData"): a `rocprofv3 --marker-trace
// --kernel-trace` run then attributes the kernels the way the reference's own profile does.  LGH_ROCTX=1 switches them on
// (librocprofiler-sdk-roctx / libroctx64 resolved with dlopen: no link dependency, nothing happens without the variable).
struct RoctxRange
{
   bool on;
   explicit RoctxRange(const char *name);
   ~RoctxRange();
   RoctxRange(const RoctxRange &) = delete;
   RoctxRange &operator=(const RoctxRange &) = delete;
};

} // namespace lgh
