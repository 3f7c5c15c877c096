// lgh_vcg_slab.hip — K1 of the lockstep velocity solve (y_e = B^T D_e B d_e for the three velocity components) at
// Q3Q2 (kernel id 0x346: D1D = 4, Q1D = 6) with the whole sum factorisation in registers: "slab form".
//
// Reference math: MassPAOperator::Mult, /root/reference/laghos_assembly.cpp:117-121 (the contraction is upstream
// MFEM's MassIntegrator::AddMultPA; restated in amr/laghos_assembly.cpp:878-963).
//
// Where it comes from.  The plane form (vcg_apply_plane, lgh_vcg.hip) exchanges the operands of the two x contractions
// through LDS behind workgroup barriers: ~19 of the ~33 us of its loop are LDS issue.  The matrix-core form
// (lgh_vcg_mfma.hip) removed that exchange - one tile column per (element, component) item, everything else in
// registers - but v_mfma_f64_16x16x4_f64 runs on the fp64 FMA pipe itself (tools/ubench_f64.hip: 64 cycles per
// instruction = 16 FMA issue slots, no overlap with v_fma_f64 of the same wave) and the 6x4 / 4x6 tables fill only 35 %
// / 18 % of a tile, so its x contractions cost four times their flops (66.8 vs 50.0 us at C2).  This kernel keeps the
// register-only data flow and does the one unavoidable exchange with gfx950's v_permlane32_swap / v_permlane16_swap:
//   * a wavefront works on a SET of 5 consecutive elements = 15 (element, component) items, item n = lane & 15
//     (column 15 idles); lane (g, n), g = lane >> 4, owns the z-slab dz = g of item n: it gathers the 16 dofs
//     d[dx][dy][dz = g] from the node vectors, contracts x and y on registers (24 + 36 outputs), and holds the 36
//     values w[qx + 6 qy] of its slab;
//   * the z contraction needs all four slabs of an item: a 4 x 4 block transpose over the lane groups (two stages of
//     18 register-pair swaps each) leaves lane group g with the nine (qx, qy) pairs 9 g .. 9 g + 8 of all four slabs;
//     forward z, scaling by the quadrature data, (d, A d) at the quadrature points and backward z follow on registers;
//   * the inverse transpose returns the slabs; backward y and x on registers; every lane stores the 16 contiguous
//     doubles (128 bytes, one cache line) of its slab of the E-vector.
//   All 60 active lanes of a wave carry the same work; the tensor data never touches LDS; there is no barrier in
//   the loop.  LDS only stages the quadrature data (coalesced global read, wave-private region).
// 1 164 vector instructions per lane and set (912 FMAs, 108 for the quadrature data and the dot product, 144 swaps).
#include "lgh_vcg.hpp"

namespace lgh
{

typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef v2d v2d_a8 __attribute__((aligned(8))); // 16-byte loads at 8-byte aligned addresses (global_load_dwordx4 takes them)

__device__ __forceinline__ double slab_ld(const double *base, const unsigned off) { return *(const double *)((const char *)base + off); }

// lane i of every row of 16 lanes receives the value of lane i - K of the same row (DPP row_shr:K, the direction the wave
// reductions of lgh_common.hpp use; a lane whose source lies outside the row receives 0.0)
template <int K> __device__ __forceinline__ double dpp_row_shr(const double v)
{
   const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + K, 0xF, 0xF, true);
   const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + K, 0xF, 0xF, true);
   return __hiloint2double(hi, lo);
}

// After the call (H = 32 or 16, "side" = bit 5 or bit 4 of the lane number): a = the a or b of the side-0 lane of the
// pair, b = that of the side-1 lane - each lane keeps the register that carries its side's number and receives the
// same register of its partner:  side 0: (a, b) = (own a, partner's a);  side 1: (a, b) = (partner's b, own b).
// v_permlane32_swap vdst, src0 swaps lanes [32, 63] of vdst with lanes [0, 31] of src0 (v_permlane16_swap: [16, 31]
// with [0, 15] and [48, 63] with [32, 47]); slab_swap_probe_k checks exactly this contract on the device.
template <int H> __device__ __forceinline__ void slab_swap(double &a, double &b)
{
   unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
   unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
   if (H == 32)
   {
      const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
      const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
      alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
   }
   else
   {
      const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
      const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
      alo = l[0]; blo = l[1]; ahi = h[0]; bhi = h[1];
   }
   a = __hiloint2double((int)ahi, (int)alo);
   b = __hiloint2double((int)bhi, (int)blo);
}

// w[9 b + i]: block b = 2 b1 + b0 of nine doubles.  slab_to_pairs: lane group s holds block b = "for lane group b";
// afterwards register block sigma of lane group g holds what lane group sigma had for g.  Its own inverse when the two
// stages run in the opposite order (slab_to_slabs).
__device__ __forceinline__ void slab_to_pairs(double (&w)[36])
{
#pragma unroll
   for (int b0 = 0; b0 < 2; b0++)
   {
#pragma unroll
      for (int i = 0; i < 9; i++) { slab_swap<32>(w[9 * b0 + i], w[9 * (2 + b0) + i]); }
   }
#pragma unroll
   for (int s1 = 0; s1 < 2; s1++)
   {
#pragma unroll
      for (int i = 0; i < 9; i++) { slab_swap<16>(w[9 * (2 * s1) + i], w[9 * (2 * s1 + 1) + i]); }
   }
}
__device__ __forceinline__ void slab_to_slabs(double (&w)[36])
{
#pragma unroll
   for (int s1 = 0; s1 < 2; s1++)
   {
#pragma unroll
      for (int i = 0; i < 9; i++) { slab_swap<16>(w[9 * (2 * s1) + i], w[9 * (2 * s1 + 1) + i]); }
   }
#pragma unroll
   for (int b0 = 0; b0 < 2; b0++)
   {
#pragma unroll
      for (int i = 0; i < 9; i++) { slab_swap<32>(w[9 * b0 + i], w[9 * (2 + b0) + i]); }
   }
}

// The same 4 x 4 block transpose for a slab of 16 values (the Kronecker form of the operator, below): w[4 b + i], block b
// of four doubles "for lane group b"; afterwards register block sigma of lane group g holds what lane group sigma had for g.
__device__ __forceinline__ void slab16_to_pairs(double (&w)[16])
{
#pragma unroll
   for (int b0 = 0; b0 < 2; b0++)
   {
#pragma unroll
      for (int i = 0; i < 4; i++) { slab_swap<32>(w[4 * b0 + i], w[4 * (2 + b0) + i]); }
   }
#pragma unroll
   for (int s1 = 0; s1 < 2; s1++)
   {
#pragma unroll
      for (int i = 0; i < 4; i++) { slab_swap<16>(w[4 * (2 * s1) + i], w[4 * (2 * s1 + 1) + i]); }
   }
}
__device__ __forceinline__ void slab16_to_slabs(double (&w)[16])
{
#pragma unroll
   for (int s1 = 0; s1 < 2; s1++)
   {
#pragma unroll
      for (int i = 0; i < 4; i++) { slab_swap<16>(w[4 * (2 * s1) + i], w[4 * (2 * s1 + 1) + i]); }
   }
#pragma unroll
   for (int b0 = 0; b0 < 2; b0++)
   {
#pragma unroll
      for (int i = 0; i < 4; i++) { slab_swap<32>(w[4 * b0 + i], w[4 * (2 + b0) + i]); }
   }
}

// The contract of slab_swap and of the two transposes, checked on the device once per process: lane group s fills
// block b, entry i with 1000 s + 10 b + i (+ 0.5 to use both register halves); after slab_to_pairs lane group g must
// hold 1000 sigma + 10 g + i in block sigma, after slab_to_slabs the original again.
__global__ void slab_swap_probe_k(double *out)
{
   const int lane = threadIdx.x, s = lane >> 4;
   double w[36];
#pragma unroll
   for (int k = 0; k < 36; k++) { w[k] = 1000.0 * s + 10.0 * (k / 9) + (k % 9) + 0.5 + 1e-3 * (lane & 15); }
   slab_to_pairs(w);
#pragma unroll
   for (int k = 0; k < 36; k++) { out[36 * lane + k] = w[k]; }
   slab_to_slabs(w);
#pragma unroll
   for (int k = 0; k < 36; k++) { out[36 * 64 + 36 * lane + k] = w[k]; }
}
static bool slab_swaps_probe(lgh_ctx *c);
static bool slab_swaps_ok(lgh_ctx *c)
{
   // once per process, and thread-safe: the ranks of the in-process loop-back communicator are host threads, and a rank that
   // asked while another one's probe was still running used to be told "unavailable" - its solve then took another form of K1
   // than its peers' and the exchanges of the ranks no longer matched
   static const bool ok = slab_swaps_probe(c);
   return ok;
}
static bool slab_swaps_probe(lgh_ctx *c)
{
   double *dev = nullptr;
   if (hipMalloc((void **)&dev, 2 * 36 * 64 * sizeof(double)) != hipSuccess) { return false; }
   hipLaunchKernelGGL(slab_swap_probe_k, dim3(1), dim3(64), 0, c->stream, dev);
   std::vector<double> h(2 * 36 * 64);
   const bool ok = hipMemcpyAsync(h.data(), dev, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                   hipStreamSynchronize(c->stream) == hipSuccess;
   (void)hipFree(dev);
   if (!ok) { return false; }
   bool good = true;
   for (int lane = 0; lane < 64 && good; lane++)
   {
      const int g = lane >> 4;
      for (int k = 0; k < 36; k++)
      {
         const int sg = k / 9, i = k % 9;
         const double want1 = 1000.0 * sg + 10.0 * g + i + 0.5 + 1e-3 * (lane & 15);
         const double want2 = 1000.0 * g + 10.0 * sg + i + 0.5 + 1e-3 * (lane & 15);
         if (h[36 * lane + k] != want1 || h[36 * 64 + 36 * lane + k] != want2)
         {
            fprintf(stderr, "lgh: v_permlane swap probe: lane %d register %d holds %g / %g, expected %g / %g - slab form of K1 unavailable\n",
                    lane, k, h[36 * lane + k], h[36 * 64 + 36 * lane + k], want1, want2);
            good = false;
            break;
         }
      }
   }
   return good;
}

// KRON (needs RANK1 and a tensor-product rule: a.M1 != nullptr): the element matrix of compact, separable mass data is a
// Kronecker product, B^T diag(s_e w (x) w (x) w) B = s_e M1 (x) M1 (x) M1 with the 1-D mass tile M1 = B^T diag(w) B
// (D x D, assembled once by lgh_create).  The pass then contracts x and y with M1 on the slab (2 x 64 FMAs), transposes
// 16 instead of 36 values over the lane groups, contracts z (64 FMAs) and transposes back: 192 FMAs + 64 lane swaps
// where the general form has 1 020 + 144, no quadrature-point values at all, and (d, A d) = sum over the element's
// nodes of d y.  Same operator in exact arithmetic; rounding differs (tests: tests/test_gpu_k1.py, 2e-12 like every
// compact-data case).  What a pass then costs is its memory instructions: the kernel becomes bandwidth-bound.
template <bool SYM, int WPS, int TRACE, bool WIDE, bool EXACT, bool DYN, bool RANK1, bool KRON = false>
__global__ void __launch_bounds__(256 * WPS, WPS)
vcg_apply_slab346(const VcgArgs a, const int nset)
{
   static_assert(!KRON || RANK1, "the Kronecker form needs compact mass data");
   constexpr int D = 4, Q = 6, NQ = Q * Q * Q, ND = D * D * D, QD = Q * D, HB = SYM ? (QD + 1) / 2 : QD;
   constexpr int ES = 5;                     // elements of a set: 15 items on the 16 lanes of a lane group
   constexpr int NW = 4 * WPS;               // wavefronts of the workgroup (one workgroup per CU), each on its own sets: WPS per SIMD
   constexpr int NDMA = (ES * NQ * 8 + 1023) / 1024; // 1 KB LDS-DMA pieces per set of quadrature data (9)
   constexpr int SBUF = NDMA * 128;          // doubles per LDS buffer (the last piece runs past the set's 1080 values)
   // per wave two buffers: the set being contracted, the set in flight.  The loop body takes them as __restrict__
   // pointers: that is what tells the compiler that the LDS-DMA in flight does not write what the body reads - without
   // it, it waits for ALL outstanding loads (the gathers of the next set) before the first LDS read of an iteration
   __shared__ double sDa[NW * SBUF], sDb[NW * SBUF];
   __shared__ double red[48];

   const int tid = threadIdx.x, lane = tid & 63;
   const unsigned long long t_enter = TRACE ? wall_clock64() : 0ull; // debug: the workgroup is on its CU
   const int wid = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform: set indices and their base addresses stay in scalar registers
   const int g = lane >> 4, n = lane & 15;
   const int ni = min(n, 14), el = ni / 3, c = ni - 3 * el;
   const int W = gridDim.x * NW;
   // The sets of a wavefront.  Static: s, s + W, ... from its place in the grid.  DYN (needs EXACT: the sum must not
   // depend on who contracts what): the workgroup owns one contiguous range of sets (neighbouring sets share nodes) and
   // its wavefronts draw them one at a time from a counter in LDS.  Why: the two wavefronts of a SIMD do not run at the
   // same speed - the older one wins the arbitration for issue slots and memory requests - and with equal shares the
   // first wavefront of a workgroup leaves its loop at 20 us, the last at 35 us (profiles/r3_k1_slab_*).
   // Pipeline: set s0 is contracted while the gathers of s1 and the map of s2 are in flight; -1 = no set.
   __shared__ unsigned s_next;
   const int wg = xcd_swizzle(blockIdx.x, gridDim.x);
   const int wg_base = (int)(((long)nset * wg) / gridDim.x), wg_len = (int)(((long)nset * (wg + 1)) / gridDim.x) - wg_base;
   auto draw = [&]() -> int {
      unsigned t = 0;
      if (lane == 0) { t = atomicAdd(&s_next, 1u); }
      const int ti = (int)__builtin_amdgcn_readfirstlane(t);
      return (ti < wg_len) ? wg_base + ti : -1;
   };
   int s0, s1, s2;
   if (DYN)
   {
      // the first three sets of a wavefront are fixed (no LDS round trips and no barrier in front of the first load of
      // the kernel); the queue takes over from there - its counter is in place long before anybody draws (barrier below)
      if (tid == 0) { s_next = 3u * NW; }
      s0 = (wid < wg_len) ? wg_base + wid : -1;
      s1 = (NW + wid < wg_len) ? wg_base + NW + wid : -1;
      s2 = (2 * NW + wid < wg_len) ? wg_base + 2 * NW + wid : -1;
   }
   else
   {
      s0 = xcd_swizzle(blockIdx.x, gridDim.x) * NW + wid;
      s1 = s0 + W;
      s2 = s0 + 2 * W;
      if (s0 >= nset) { s0 = -1; }
   }
   const int s = (s0 >= 0) ? s0 : 0; // (first set, for the prologue loads)

   // No predicates on the loads of the pipeline (as in vcg_apply_plane): sets past the end re-read the last set, elements
   // past the end the last element, and the quadrature data is padded by one set behind its last element (lgh_create);
   // nothing of that is stored or summed.  The loop body is straight-line code.
   unsigned mo[16]; // byte offsets of this lane's 16 nodes (dx + 4 dy; dz = g) into a node vector (WIDE: of the first node of each x-row, mo[4 dy])
   unsigned mn[16]; // ... of the set after: the gathers of a set go out in four parts during the pass before it (see the loop body)
   auto load_map = [&](const int ss, unsigned (&m_)[16]) {
      const int e = min(ES * min(ss, nset - 1) + el, a.NE - 1);
      if (WIDE)
      {
         const unsigned *p = a.mapb + (size_t)e * ND + 16 * g;
#pragma unroll
         for (int dy = 0; dy < 4; dy++) { m_[4 * dy] = p[4 * dy]; }
         return;
      }
      const v4u *p = (const v4u *)(a.mapb + (size_t)e * ND + 16 * g);
#pragma unroll
      for (int dy = 0; dy < 4; dy++)
      {
         const v4u m = p[dy];
         m_[4 * dy + 0] = m[0]; m_[4 * dy + 1] = m[1]; m_[4 * dy + 2] = m[2]; m_[4 * dy + 3] = m[3];
      }
   };
   load_map(s, mo);
   // merged E-vector layout (a.settab, lgh_vcg.hpp): word of the set being contracted - its slice of a Y_E plane in units
   // of 64 doubles, bit 31: the set is an x-chain whose rows are stored merged; the word of the next set is read a pass ahead
   unsigned tw_cur = a.settab ? a.settab[min(s, nset - 1)] : 0u;
   // Everything else the prologue needs from memory goes out WITH the map, ahead of the first wait: the scalars of the
   // solve and the 1-D table.  (Round 3 read them one after the other behind the gathers - convergence flag, then the
   // done flags and (r, z), then the table: four dependent round trips of ~1.2 us between a workgroup's arrival and its
   // first pass, profiles/r4_k1_kron_trace.txt; now two: this batch, then the gathers.)
   const VcgScalars *const sc0 = a.s;
   const int sc_all_done = sc0->all_done;
   int sc_done[kVC];
   double sc_rz[kVC], sc_rzp[kVC], sc_r0[kVC];
#pragma unroll
   for (int k = 0; k < kVC; k++) { sc_done[k] = sc0->done[k]; sc_rz[k] = sc0->rz[k]; sc_rzp[k] = sc0->rz_prev[k]; sc_r0[k] = sc0->r0[k]; }
   // rz_limbs mode (lgh_vcg.hpp): (r, z) of the iteration before the last comes from the scalars (first iteration: both
   // are the initial one), the last one out of the exact accumulators K2 added into - folded below
   if (a.rzl && a.iter > 1)
   {
#pragma unroll
      for (int k = 0; k < kVC; k++) { sc_rzp[k] = sc0->rzh[a.iter & 1][k]; } // (= [(iter - 2) & 1])
   }
   // (the words of that set: one per lane, one load instruction in this batch - not a round trip of its own later)
   long long rz_word = 0;
   if (a.rzl && a.iter > 1 && lane <= kLimbShards * kVC * kLimbs)
   {
      rz_word = a.rzl[((a.iter - 1) % 3) * kLimbWords + lane];
      // several ranks: the words of the other ranks, as the exchange after K2 left them - own + peers' is the sum over
      // the ranks, the same integers on every rank (the flag word adds up to "some rank's sum is bad")
      for (int p = 0; p < a.n_rz_peers; p++) { rz_word += a.rzl_peers[p * kLimbWords + lane]; }
   }
   constexpr int NTAB = KRON ? D * D : HB;
   double tabv[NTAB];
#pragma unroll
   for (int i = 0; i < NTAB; i++) { tabv[i] = KRON ? a.M1[i] : a.B[i]; }

   // node vectors: one scalar base + 32-bit byte offsets (vcg_slab_available checks kVC * N * 8 < 2^32).  In the first
   // iteration (beta = 0) the old direction is not defined: r is read in its place and multiplied by zero.
   const bool first = (a.iter == 1); // (= a.s->first, without a memory round trip in front of the gathers)
   const unsigned coff = 8u * (unsigned)c * (unsigned)a.N;
   const double *dsrc = first ? a.r : a.d;
   __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the one-time loads are complete before the pipelined loop (see vcg_apply_plane)

   // WIDE: the four nodes of every x-row of an element are consecutive in the node vectors (checked on the map at set-up:
   // true for any tensor-product numbering with x fastest) - a row is two 16-byte loads per vector instead of four
   // 8-byte gathers, and every cache line is asked for by one or two instructions instead of four
   double gz[16], gd[16], gv[16];
   auto load_gather_part = [&](const int dy, const bool pin = false) __attribute__((always_inline)) {
      if (WIDE)
      {
         // (pin: the offset passes through an empty volatile asm - the loads are read-only and the optimiser would
         //  otherwise collect all parts in one place, whatever fences stand between them)
         unsigned off = mo[4 * dy];
         if (pin) { asm volatile("" : "+v"(off)); }
         const v2d_a8 *pv = (const v2d_a8 *)((const char *)a.dinv + off);
         const v2d_a8 *pz = (const v2d_a8 *)((const char *)a.r + (off + coff));
         const v2d_a8 *pd = (const v2d_a8 *)((const char *)dsrc + (off + coff));
         const v2d va = pv[0], vb = pv[1], za = pz[0], zb = pz[1], da = pd[0], db = pd[1];
         gv[4 * dy] = va[0]; gv[4 * dy + 1] = va[1]; gv[4 * dy + 2] = vb[0]; gv[4 * dy + 3] = vb[1];
         gz[4 * dy] = za[0]; gz[4 * dy + 1] = za[1]; gz[4 * dy + 2] = zb[0]; gz[4 * dy + 3] = zb[1];
         gd[4 * dy] = da[0]; gd[4 * dy + 1] = da[1]; gd[4 * dy + 2] = db[0]; gd[4 * dy + 3] = db[1];
         return;
      }
#pragma unroll
      for (int j = 4 * dy; j < 4 * dy + 4; j++)
      {
         unsigned off = mo[j];
         if (pin) { asm volatile("" : "+v"(off)); }
         gv[j] = slab_ld(a.dinv, off);
         gz[j] = slab_ld(a.r, off + coff);
         gd[j] = slab_ld(dsrc, off + coff);
      }
   };
   auto load_gather = [&]() {
#pragma unroll
      for (int dy = 0; dy < 4; dy++) { load_gather_part(dy); }
   };
   // quadrature data of a set: LDS-DMA, 16 bytes per lane and piece, straight into the wave's buffer `buf` (no registers)
   // Quadrature data of a set -> the wave's LDS buffer by LDS-DMA (16 bytes per lane and piece, no registers).
   // RANK1 (mass_data, lgh_mass.hip: D[q, e] = W[q] s_e): both buffers take the point weights of five elements once,
   // before the loop, and a set only brings its five factors s_e - 8 instead of 1728 bytes per element.
   auto load_dq = [&](const int ss, double *__restrict__ l) {
      if (RANK1) { return; }
      const double *p = a.Dq + (size_t)min(ss, nset - 1) * (ES * NQ) + 2 * lane;
#pragma unroll
      for (int k = 0; k < NDMA; k++)
      {
         __builtin_amdgcn_global_load_lds(p + 128 * k, (__attribute__((address_space(3))) void *)(l + 128 * k), 16, 0, 0);
      }
   };
   if (RANK1 && !KRON)
   {
#pragma unroll
      for (int k = 0; k < NDMA; k++)
      {
         const int t = (128 * k + 2 * lane) % NQ; // (NQ is even: a 16-byte piece never straddles two elements)
         __builtin_amdgcn_global_load_lds(a.Dq + t, (__attribute__((address_space(3))) void *)(sDa + wid * SBUF + 128 * k), 16, 0, 0);
      }
   }
   load_gather();
   load_dq(s, sDa + wid * SBUF);
   load_map(DYN ? max(s1, 0) : s1, mo);
   // (the scalars of the solve are looked at with the first loads of the kernel already in flight)
   if (sc_all_done) { __builtin_amdgcn_s_waitcnt(0x0F70); return; } // (nothing in flight into the LDS of a workgroup that is gone)
   bool todo[kVC];
   double beta[kVC];
#pragma unroll
   for (int k = 0; k < kVC; k++) { todo[k] = sc_done[k] == 0; }
   if (a.rzl)
   {
      // (r, z) after the last iteration: every wavefront folds the set K2 added into out of its own registers (the scale
      // is that of the value before; one vector load in the prologue batch, v_readlane, integer adds - four wavefronts
      // per CU: the scalar unit has the room, and no wavefront depends on another one's view of the flags).  Every
      // wavefront of every workgroup decides alike (integers); workgroup 0 commits what K2 and the host read and clears
      // the set K2 of this iteration adds into.
      if (a.iter > 1)
      {
#pragma unroll
         for (int k = 0; k < kVC; k++) { sc_rz[k] = exact_fold_lanes(rz_word, k, exact_scale(sc_rzp[k])); }
      }
      bool dn[kVC];
#pragma unroll
      for (int k = 0; k < kVC; k++)
      {
         dn[k] = sc_done[k] != 0 || vcg_rz_converged(a.iter, sc_rz[k], sc_r0[k]);
         todo[k] = !dn[k];
      }
      if (blockIdx.x == 0)
      {
         if (tid == 0)
         {
            vcg_rz_commit(a.s, a.iter, sc_rz, dn);
            // several ranks: the exchanges of launches enqueued past a component's convergence go on summing its (d, A d)
            // over the ranks - they must sum zeros (see vcg_update_finish_k)
            if (a.multi)
            {
#pragma unroll
               for (int k = 0; k < kVC; k++) { if (dn[k]) { __hip_atomic_store(&a.s->den[k], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } }
            }
         }
         if (tid < kLimbWords) { a.rzl[(a.iter % 3) * kLimbWords + tid] = 0; }
      }
      if (!(todo[0] || todo[1] || todo[2])) { __builtin_amdgcn_s_waitcnt(0x0F70); return; }
   }
   if (a.multi && !a.rzl && !first && !vcg_pending_update(a.s, a.iter, blockIdx.x == 0 && tid == 0, todo)) { __builtin_amdgcn_s_waitcnt(0x0F70); return; }
#pragma unroll
   for (int k = 0; k < kVC; k++) { beta[k] = (first || !todo[k]) ? 0.0 : sc_rz[k] / sc_rzp[k]; }
   const bool mine = (c == 0) ? todo[0] : (c == 1) ? todo[1] : todo[2];
   const double betac = (c == 0) ? beta[0] : (c == 1) ? beta[1] : beta[2];

   double Bsr[HB];
#pragma unroll
   for (int i = 0; i < HB; i++) { Bsr[i] = KRON ? 0.0 : uniform_f64(tabv[KRON ? 0 : i]); }
   double M1[D * D]; // KRON: the 1-D mass tile, M1[i + D j] (symmetric), in scalar registers
#pragma unroll
   for (int i = 0; i < D * D; i++) { M1[i] = KRON ? uniform_f64(tabv[KRON ? i : 0]) : 0.0; }
   auto Bs = [&](const int idx) -> double { return (SYM && idx >= HB) ? Bsr[QD - 1 - idx] : Bsr[idx]; };
   // (RANK1: A d = s_e (B^T W B d) - the element factor multiplies the 16 outputs and the partial of (d, A d) at the
   //  end of the pass, not the 54 point values; loaded with the gathers of the pass, two registers)
   // direction d = z + beta d (K2 stores the same values)
   double dd[16];
   auto convert = [&]() {
#pragma unroll
      for (int j = 0; j < 16; j++) { dd[j] = fma(betac, gd[j], __dmul_rn(gz[j], gv[j])); }
   };

   // The E-vector stores.  The 16 outputs of a lane are one 128-byte line of Y_E, and a store instruction that writes
   // 16 bytes of 64 different lines costs the memory pipeline as much as 64 full lines would (measured: a quarter of the
   // pass spent behind them, profiles/README.md).  The lines of a group of eight lanes are therefore exchanged through
   // the wavefront's LDS buffer (144-byte pitch: every access at the bank limit) so that instruction k stores the
   // complete line of lane (lane & ~7) + k, 16 bytes per lane.  Per lane and k: the byte offset of that piece inside the
   // set's slice of Y_E, and whether the owner is an item at all (n < 15) whose component still iterates.
   constexpr int TP = 18; // doubles per lane in the exchange buffer (16 + 2: pitch 144 bytes)
   static_assert(64 * TP <= SBUF, "the exchange buffer is a quadrature-data buffer of the wavefront");
   unsigned st_off[8];
   unsigned st_ok = 0, st_el = 0; // bit k / bits 3k..3k+2: owner k is live / its element within the set
#pragma unroll
   for (int k = 0; k < 8; k++)
   {
      const int no = (n & 8) + k, nio = min(no, 14), elo = nio / 3, co = nio - 3 * elo;
      st_off[k] = 8u * ((unsigned)co * (unsigned)a.ye_stride + (unsigned)(ND * elo + 16 * g)) + 16u * (unsigned)(lane & 7);
      const bool live = (no < 15) && ((co == 0) ? todo[0] : (co == 1) ? todo[1] : todo[2]);
      st_ok |= live ? (1u << k) : 0u;
      st_el |= (unsigned)elo << (3 * k);
   }
   double dot = 0.0;
   // EXACT: (d, A d) in integer accumulators (lgh_vcg.hpp): no ticket, no last workgroup; vcg_update_p_k forms the value
   long long acc[kLimbs] = {0, 0, 0, 0};
   bool acc_bad = false;
   const int accE = exact_scale((c == 0) ? sc_rz[0] : (c == 1) ? sc_rz[1] : sc_rz[2]);
   __builtin_amdgcn_s_waitcnt(0x0F70);
   convert();
   if (DYN) { __syncthreads(); } // s_next is in place (the first draw is a pass away)
   // debug (LGH_VCG_TRACE): wall-clock stamps of wave 0 and the shader cycles it spends waiting for the loads of a set
   unsigned long long t_start = 0, t_loop = 0, c_wait = 0, c_loop = 0;
   unsigned long long c_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c_prev = 0; // shader cycles per phase of the loop body (wave 0)
   if (TRACE) { t_start = wall_clock64(); }
   if (TRACE == 2) { c_loop = clock64(); c_prev = c_loop; }
#define LGH_SLAB_STAMP(K_) do { if (TRACE == 2) { const unsigned long long c_now = clock64(); c_ph[K_] += c_now - c_prev; c_prev = c_now; } } while (0)
   auto body = [&](const double *__restrict__ sDcur, double *__restrict__ sDnxt) __attribute__((always_inline)) {
      const int e = ES * max(s0, 0) + el;
      const bool act = (n < 15) && (e < a.NE) && mine;
      const double actf = act ? 1.0 : 0.0;
      const double *sDp = sDcur + el * NQ + 9 * g; // this lane's nine (qx, qy) pairs: sDp[i + 36 qz]
      // next set: gathers and quadrature data now, the map of the one after
      // (DYN: an empty pipeline slot - at most three per wavefront, at the end - re-reads set 0: no branch around the
      // LDS-DMA, or the compiler loses track of what it writes and drains every load before the first LDS read)
      double se = 1.0;
      if (RANK1) { se = a.Se[min(e, a.NE - 1)]; }
      // The gathers go out in four parts (one per y-row: 6 wide loads, or 12 single ones), a part at the top and the
      // others between the phases below: issued all at once they queue up behind the address unit of the CU and the
      // wavefront - the only one on its SIMD, or one of two - stands still for a quarter of the pass.
      load_gather_part(0);
      load_dq(DYN ? max(s1, 0) : s1, sDnxt);
      load_map(DYN ? max(s2, 0) : s2, mn);
      unsigned tw_nxt = 0u;
      if (a.settab) { tw_nxt = a.settab[__builtin_amdgcn_readfirstlane(min(max(s1, 0), nset - 1))]; } // (wave-uniform: a scalar load, a pass ahead of its use)
      LGH_SLAB_STAMP(0); // issue of the loads
      double o[16], dset = 0.0;
      // (The body only runs with a set to contract: s0 >= 0 - the sets of a wavefront are a run of valid ones followed by
      //  empty slots, see more().  Round 4: a `s0 < 0` path that only issued the loads was dead code, but the compiler
      //  merged its pending loads into the wait counters of the live path and drained the gathers just issued - vmcnt(0) -
      //  in front of the first contraction of every pass, profiles/r4_k1_spurious_waits.txt.)
      if (KRON)
      {
         {
            __builtin_amdgcn_sched_barrier(0);
            // x, then y with the 1-D mass tile on this lane's slab dz = g: w[dx + 4 dy]
            double w[16];
#pragma unroll
            for (int dy = 0; dy < D; dy++)
            {
#pragma unroll
               for (int i = 0; i < D; i++)
               {
                  double u = M1[i] * dd[D * dy];
#pragma unroll
                  for (int dx = 1; dx < D; dx++) { u = fma(M1[i + D * dx], dd[dx + D * dy], u); }
                  w[i + D * dy] = u;
               }
            }
            double v[16];
#pragma unroll
            for (int i = 0; i < D; i++)
            {
#pragma unroll
               for (int j = 0; j < D; j++)
               {
                  double u = M1[j] * w[i];
#pragma unroll
                  for (int dy = 1; dy < D; dy++) { u = fma(M1[j + D * dy], w[i + D * dy], u); }
                  v[i + D * j] = u;
               }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_gather_part(1, true);
            __builtin_amdgcn_sched_barrier(0);
            LGH_SLAB_STAMP(1);
            // slabs -> pairs: v[4 dz + i] = slab dz of the pair (i, dy = g); z with the tile; back
            slab16_to_pairs(v);
            __builtin_amdgcn_sched_barrier(0);
            LGH_SLAB_STAMP(2);
#pragma unroll
            for (int i = 0; i < D; i++)
            {
               double t[D];
#pragma unroll
               for (int k = 0; k < D; k++)
               {
                  double u = M1[k] * v[i];
#pragma unroll
                  for (int dz = 1; dz < D; dz++) { u = fma(M1[k + D * dz], v[D * dz + i], u); }
                  t[k] = u;
               }
#pragma unroll
               for (int k = 0; k < D; k++) { v[D * k + i] = t[k]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_gather_part(2, true);
            __builtin_amdgcn_sched_barrier(0);
            LGH_SLAB_STAMP(3);
            slab16_to_slabs(v);
            __builtin_amdgcn_sched_barrier(0);
            load_gather_part(3, true);
            __builtin_amdgcn_sched_barrier(0);
            LGH_SLAB_STAMP(4);
            // (d, A d) of this lane's slab: sum over its 16 nodes (the element factor s_e multiplies it below, with the outputs)
#pragma unroll
            for (int j = 0; j < 16; j++)
            {
               o[j] = v[j];
               dset = fma(dd[j], v[j], dset);
            }
            __builtin_amdgcn_sched_barrier(0);
            LGH_SLAB_STAMP(5);
         }
      }
      else
      {
      // Phase order is pinned (sched_barrier): with few wavefronts per SIMD the compiler would otherwise hoist every LDS
      // read and half the next phase above the current one and pay for it in register moves.
      __builtin_amdgcn_sched_barrier(0);
      // forward x and y, one x-index at a time: t[dy] = sum_dx B[qx,dx] d[dx + 4 dy]; w[qx + 6 qy] = sum_dy B[qy,dy] t[dy]
      double w[36];
#pragma unroll
      for (int qx = 0; qx < Q; qx++)
      {
         double t[D];
#pragma unroll
         for (int dy = 0; dy < D; dy++)
         {
            double u = 0.0;
#pragma unroll
            for (int dx = 0; dx < D; dx++) { u = fma(Bs(qx + Q * dx), dd[dx + D * dy], u); }
            t[dy] = u;
         }
#pragma unroll
         for (int qy = 0; qy < Q; qy++)
         {
            double u = 0.0;
#pragma unroll
            for (int dy = 0; dy < D; dy++) { u = fma(Bs(qy + Q * dy), t[dy], u); }
            w[qx + Q * qy] = u;
         }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_gather_part(1, true);
      __builtin_amdgcn_sched_barrier(0);
      LGH_SLAB_STAMP(1); // forward x, y
      // slabs -> pairs: w[9 dz + i] = slab dz, pair 9 g + i
      slab_to_pairs(w);
      __builtin_amdgcn_sched_barrier(0);
      LGH_SLAB_STAMP(2); // transpose
      // forward z, quadrature data, (d, A d), backward z; the quadrature data of pair i + 1 is read while pair i is contracted
      // (one wavefront per SIMD: the data of pair i + 1 is read while pair i is contracted; with two the other wavefront
      // covers the LDS latency and the 12 registers of the look-ahead are worth more)
      constexpr bool AHEAD = (WPS == 1);
      double dcur[Q], dnxt[Q];
#pragma unroll
      for (int qz = 0; qz < Q; qz++) { dcur[qz] = sDp[36 * qz]; }
#pragma unroll
      for (int i = 0; i < 9; i++)
      {
         if (AHEAD && i < 8)
         {
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { dnxt[qz] = sDp[i + 1 + 36 * qz]; }
         }
         double cz[Q];
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            double u = 0.0;
#pragma unroll
            for (int dz = 0; dz < D; dz++) { u = fma(Bs(qz + Q * dz), w[9 * dz + i], u); }
            cz[qz] = u * dcur[qz];
            dset = fma(u, cz[qz], dset);
         }
#pragma unroll
         for (int dz = 0; dz < D; dz++)
         {
            double u = 0.0;
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { u = fma(Bs(qz + Q * dz), cz[qz], u); }
            w[9 * dz + i] = u;
         }
         if (AHEAD)
         {
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { dcur[qz] = dnxt[qz]; }
         }
         else if (i < 8)
         {
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { dcur[qz] = sDp[i + 1 + 36 * qz]; }
         }
         if (i == 2) { load_gather_part(2, true); }
         if (i == 5) { load_gather_part(3, true); }
         asm volatile("" : "+v"(dset)); // the partial sum exists HERE: left alone, the compiler keeps all 54 factor pairs alive (216 registers) and forms the sum after the loop body
         __builtin_amdgcn_sched_barrier(0);
      }
      LGH_SLAB_STAMP(3); // z
      // pairs -> slabs
      slab_to_slabs(w);
      __builtin_amdgcn_sched_barrier(0);
      LGH_SLAB_STAMP(4); // transpose back
      // backward y and x, one x-index at a time: t[dy] = sum_qy B[qy,dy] w[qx + 6 qy]; out[dx + 4 dy] += B[qx,dx] t[dy]
#pragma unroll
      for (int qx = 0; qx < Q; qx++)
      {
         double t[D];
#pragma unroll
         for (int dy = 0; dy < D; dy++)
         {
            double u = 0.0;
#pragma unroll
            for (int qy = 0; qy < Q; qy++) { u = fma(Bs(qy + Q * dy), w[qx + Q * qy], u); }
            t[dy] = u;
         }
#pragma unroll
         for (int dy = 0; dy < D; dy++)
         {
#pragma unroll
            for (int dx = 0; dx < D; dx++) { o[dx + D * dy] = (qx == 0) ? Bs(qx + Q * dx) * t[dy] : fma(Bs(qx + Q * dx), t[dy], o[dx + D * dy]); }
         }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the loads of the next set are complete by now (wave 0 counts what is left of their latency): its direction;
      // the stores of this set go out behind them, so that no wait ever covers a store that has just been issued
      LGH_SLAB_STAMP(5); // backward y, x
      }
      if (TRACE == 2)
      {
         const unsigned long long c0 = clock64();
         __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
         c_wait += clock64() - c0;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): gathers, LDS-DMA and map of the next set (and the ticket)
      convert(); // (an empty slot s1 < 0 has re-read set 0: finite values nobody uses)
#pragma unroll
      for (int j = 0; j < 16; j += (WIDE ? 4 : 1)) { mo[j] = mn[j]; }
      __builtin_amdgcn_sched_barrier(0);
      LGH_SLAB_STAMP(6); // wait + direction of the next set
      // the slab of the E-vector: 16 contiguous doubles per lane, stored as whole lines by groups of eight lanes (above);
      // an x-chain set of the merged layout: 48 rows of 16 x-nodes, the shared x-faces summed here (below)
      {
         double *tl = RANK1 ? (sDb + wid * SBUF) : const_cast<double *>(sDcur); // (not RANK1: the data of this set has been used)
         const int s0c = max(s0, 0);
         const int nel = min(ES, a.NE - ES * s0c); // elements of this set (the last one may be short)
         // one scalar base + 32-bit byte offsets while the three planes of Y_E fit 4 GB (every mesh up to 140^3 zones);
         // beyond that (a.ye_wide) the set's offset goes into the base and only the per-lane part has to fit
         const bool tabled = a.settab != nullptr;
         const bool chain = tabled && (tw_cur >> 31) != 0u;
         const unsigned set_off = tabled ? (tw_cur & 0x7fffffffu) * 512u : 8u * (unsigned)ND * (unsigned)(ES * s0c);
         char *const set_base = (char *)a.YE + (tabled ? (size_t)(tw_cur & 0x7fffffffu) * 512 : (size_t)8 * ND * ES * (size_t)s0c);
         if (chain)
         {
            // Merged rows (round 5).  The five zones of this set are x-neighbours (checked on the map at set-up): zone el
            // and zone el + 1 share the 16 nodes of an x-face, and the two contributions to such a node are added HERE -
            // a lane shift by three items (DPP row_shr:3: item n - 3 is the same component of the zone to the left) - instead
            // of travelling to K2 as two values behind two table entries.  A component's slice of the set is then 16 rows
            // (dz, dy) of 16 x-nodes: 256 instead of 320 doubles, every row one 128-byte line, and the node kernel's
            // wavefronts - consecutive nodes of an x-line - read whole lines of it.  The rows are assembled in the
            // wavefront's LDS buffer (lane (g, el, c) owns x = 3 el .. 3 el + 2 of the rows (c, dz = g, dy), zone 4 also
            // x = 15) and leave as six instructions of 1 KB: eight complete lines each.
            constexpr int RB = 66; // doubles per (component, dz) block of four rows in the exchange buffer (64 + 2: blocks start on different banks)
            static_assert(12 * RB <= SBUF, "the rows of a set fit the wavefront's exchange buffer");
            double os[16];
#pragma unroll
            for (int j = 0; j < 16; j++) { os[j] = o[j] * se; }
#pragma unroll
            for (int dy = 0; dy < 4; dy++)
            {
               // x = 3 el is this zone's dx = 0 AND the left zone's dx = 3: the owner of the place (this lane) takes the sum
               const double t = dpp_row_shr<3>(os[4 * dy + 3]); // (dx = 3 column of the zone to the left: three items down)
               os[4 * dy] += (el > 0) ? t : 0.0;
            }
            if (n < 15)
            {
               double *rw = tl + (4 * c + g) * RB + 3 * el;
#pragma unroll
               for (int dy = 0; dy < 4; dy++)
               {
                  rw[16 * dy + 0] = os[4 * dy + 0];
                  rw[16 * dy + 1] = os[4 * dy + 1];
                  rw[16 * dy + 2] = os[4 * dy + 2];
                  if (el == ES - 1) { rw[16 * dy + 3] = os[4 * dy + 3]; }
               }
            }
            __builtin_amdgcn_wave_barrier(); // (one wavefront: its LDS instructions execute in order; this only pins the compiler)
            // piece q = 64 k + lane of the 384 16-byte pieces: row q >> 3 = 4 (block) + dy, piece q & 7 of the row
            const v2d *rp = (const v2d *)(tl + (lane >> 5) * RB + 16 * ((lane >> 3) & 3) + 2 * (lane & 7));
            double vlo[6], vhi[6];
#pragma unroll
            for (int k = 0; k < 6; k++)
            {
               const v2d val = rp[k * RB]; // (8 rows = 2 blocks further per instruction: 2 RB doubles = RB pieces)
               vlo[k] = val[0];
               vhi[k] = val[1];
            }
#pragma unroll
            for (int k = 0; k < 6; k++) { asm volatile("" : "+v"(vlo[k]), "+v"(vhi[k])); }
#pragma unroll
            for (int k = 0; k < 6; k++)
            {
               const bool live = (k < 2) ? todo[0] : (k < 4) ? todo[1] : todo[2];
               if (live)
               {
                  const unsigned off = 16u * (unsigned)lane + 1024u * (unsigned)(k & 1) + 8u * (unsigned)(k >> 1) * (unsigned)a.ye_stride;
                  if (a.ye_wide) { *(v2d *)(set_base + off) = v2d{vlo[k], vhi[k]}; }
                  else { *(v2d *)((char *)a.YE + (off + set_off)) = v2d{vlo[k], vhi[k]}; }
               }
            }
         }
         else
         {
         v2d *wp = (v2d *)(tl + lane * TP);
#pragma unroll
         for (int j = 0; j < 8; j++) { wp[j] = v2d{o[2 * j] * se, o[2 * j + 1] * se}; }
         __builtin_amdgcn_wave_barrier(); // (one wavefront: its LDS instructions execute in order; this only pins the compiler)
         const v2d *rp = (const v2d *)(tl + (lane & ~7) * TP) + (lane & 7);
         // (all eight reads first, pinned: inside the predicated blocks each of them would be followed by a wait for
         //  the LDS - eight round trips in a row, a seventh of the pass)
         double vlo[8], vhi[8];
#pragma unroll
         for (int k = 0; k < 8; k++)
         {
            const v2d val = rp[k * (TP / 2)];
            vlo[k] = val[0];
            vhi[k] = val[1];
         }
#pragma unroll
         for (int k = 0; k < 8; k++) { asm volatile("" : "+v"(vlo[k]), "+v"(vhi[k])); }
#pragma unroll
         for (int k = 0; k < 8; k++)
         {
            if (((st_ok >> k) & 1u) && (int)((st_el >> (3 * k)) & 7u) < nel) 
            {
               if (a.ye_wide) { *(v2d *)(set_base + st_off[k]) = v2d{vlo[k], vhi[k]}; }
               else { *(v2d *)((char *)a.YE + (st_off[k] + set_off)) = v2d{vlo[k], vhi[k]}; }
            }
         }
         }
         tw_cur = tw_nxt;
      }
      if (a.store_wait) { __builtin_amdgcn_s_waitcnt(0x0F70); }
      LGH_SLAB_STAMP(7); // stores
      // (a select would let the compiler sink all 54 products of dset behind the branch: 216 live registers)
      if (EXACT) { acc_bad = acc_bad || !exact_add(acc, dset * (actf * se), accE); }
      else { dot = fma(dset, actf * se, dot); }
      // the pipeline moves on
      if (DYN)
      {
         s0 = s1;
         s1 = s2;
         s2 = draw();
      }
      else
      {
         s0 = (s1 < nset) ? s1 : -1;
         s1 = s2;
         s2 += W;
      }
   };
   // (DYN: the initial slots and the draws are monotone - once a slot is empty all later ones are - so s0 < 0 ends the loop)
   auto more = [&]() -> bool { return s0 >= 0; };
   double *bcur = sDa + wid * SBUF, *bnxt = RANK1 ? bcur : sDb + wid * SBUF; // (RANK1: the weights stay in sDa, sDb is the exchange buffer of the stores)
   int n_pass = 0;
   while (more())
   {
      if (TRACE == 1) { n_pass += (s0 >= 0) ? 1 : 0; }
      body(bcur, bnxt);
      double *const tmp = bcur;
      bcur = bnxt;
      bnxt = tmp;
   }
   if (TRACE) { t_loop = wall_clock64(); }
   if (TRACE == 2) { c_loop = clock64() - c_loop; }
   __shared__ unsigned long long s_twave[NW]; // debug (TRACE 1): when every wavefront left the loop, and after how many passes
   if (TRACE == 1 && lane == 0) { s_twave[wid] = (t_loop << 8) | (unsigned long long)(n_pass & 0xff); }
   if (EXACT)
   {
      // integer sums over the workgroup, then kVC * kLimbs fire-and-forget atomics into this workgroup's shard
      __shared__ long long redi[NW][kVC * kLimbs];
      __shared__ int redbad[NW];
#pragma unroll
      for (int k = 0; k < kVC; k++)
      {
#pragma unroll
         for (int j = 0; j < kLimbs; j++)
         {
            const long long tot = wave_sum_i64((c == k && n < 15) ? acc[j] : 0LL);
            if (lane == 0) { redi[wid][kLimbs * k + j] = tot; }
         }
      }
      const bool anybad = __any(acc_bad && n < 15 && mine);
      if (lane == 0) { redbad[wid] = anybad ? 1 : 0; }
      __syncthreads();
      long long *L = a.limbs + (a.den_limbs ? (a.iter & 1) * kLimbWords : 0);
      __shared__ unsigned int s_last;
      if (tid < kVC * kLimbs)
      {
         long long sum = 0;
#pragma unroll
         for (int w = 0; w < NW; w++) { sum += redi[w][tid]; }
         if (sum != 0) { (void)__hip_atomic_fetch_add(&L[(blockIdx.x % kLimbShards) * (kVC * kLimbs) + tid], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
      if (tid == kVC * kLimbs)
      {
         int bad = 0;
#pragma unroll
         for (int w = 0; w < NW; w++) { bad |= redbad[w]; }
         if (bad) { (void)__hip_atomic_fetch_or(&L[kLimbShards * kVC * kLimbs], 1LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
      if (a.den_limbs)
      {
         // K2 folds the set: nothing returns to this kernel, the workgroup is gone as soon as its atomics are on their way
         if (TRACE && tid == 0)
         {
            a.trace[kTraceRec * blockIdx.x + 0] = t_start;
            a.trace[kTraceRec * blockIdx.x + 1] = t_loop;
            a.trace[kTraceRec * blockIdx.x + 2] = wall_clock64();
            a.trace[kTraceRec * blockIdx.x + 12] = t_enter;
            a.trace[kTraceRec * blockIdx.x + 3] = (c_wait << 32) | (c_loop & 0xffffffffull);
            for (int k = 0; k < 8; k++) { a.trace[kTraceRec * blockIdx.x + 4 + k] = (TRACE == 1) ? (k < NW ? s_twave[k] : 0ull) : c_ph[k]; }
         }
         return;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the atomics have been performed (gfx9: vmcnt counts them) ...
      __syncthreads();
      if (tid == 0)
      {
         // ... before this workgroup is counted.  Only the count needs a returning atomic; whoever is last reads the
         // accumulators (a dozen words, not one partial per workgroup), leaves them cleared and commits the result.
         unsigned int *tk = a.ticket + kShards * kTicketStride;
         const unsigned arrived = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         s_last = (arrived == gridDim.x - 1) ? 1u : 0u;
         if (s_last) { __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
      __syncthreads();
      if (s_last && tid < kVC)
      {
         long long l4[kLimbs] = {0, 0, 0, 0};
#pragma unroll
         for (int sh = 0; sh < kLimbShards; sh++)
         {
#pragma unroll
            for (int j = 0; j < kLimbs; j++)
            {
               long long *w = &L[sh * (kVC * kLimbs) + kLimbs * tid + j];
               l4[j] += __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               __hip_atomic_store(w, 0LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
         }
         const long long bad = __hip_atomic_load(&L[kLimbShards * kVC * kLimbs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
         VcgScalars *sc = a.s;
         const bool td = (tid == 0) ? todo[0] : (tid == 1) ? todo[1] : todo[2];
         if (td)
         {
            const double den = bad ? __builtin_nan("") : exact_value(l4, exact_scale(sc->rz[tid]));
            sc->den[tid] = den;
            if (den == 0.0 && !a.multi) { sc->done[tid] = 1; } // breakdown, as upstream
         }
         if (tid == 0)
         {
            sc->first = 0;
            if (bad) { __hip_atomic_store(&L[kLimbShards * kVC * kLimbs], 0LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
         }
      }
      if (TRACE && tid == 0)
      {
         a.trace[kTraceRec * blockIdx.x + 0] = t_start;
         a.trace[kTraceRec * blockIdx.x + 1] = t_loop;
         a.trace[kTraceRec * blockIdx.x + 2] = wall_clock64();
            a.trace[kTraceRec * blockIdx.x + 12] = t_enter;
         a.trace[kTraceRec * blockIdx.x + 3] = (c_wait << 32) | (c_loop & 0xffffffffull);
         for (int k = 0; k < 8; k++) { a.trace[kTraceRec * blockIdx.x + 4 + k] = c_ph[k]; }
      }
      return;
   }
   double bp[kVC];
   block_sum3(c == 0 ? dot : 0.0, c == 1 ? dot : 0.0, c == 2 ? dot : 0.0, red, bp);
   double total[kVC];
   const bool last = grid_sum3_last_block_flat(bp, a.partials, a.stride, a.ticket, red, total);
   if (TRACE && tid == 0)
   {
      a.trace[kTraceRec * blockIdx.x + 0] = t_start;
      a.trace[kTraceRec * blockIdx.x + 1] = t_loop;
      a.trace[kTraceRec * blockIdx.x + 2] = wall_clock64();
            a.trace[kTraceRec * blockIdx.x + 12] = t_enter;
      a.trace[kTraceRec * blockIdx.x + 3] = (c_wait << 32) | (c_loop & 0xffffffffull); // cycles waiting | cycles in the loop
      for (int k = 0; k < 8; k++) { a.trace[kTraceRec * blockIdx.x + 4 + k] = c_ph[k]; }
   }
   if (last)
   {
      if (tid == 0)
      {
         VcgScalars *sc = a.s;
         for (int k = 0; k < kVC; k++)
         {
            if (!todo[k]) { continue; }
            sc->den[k] = total[k];
            if (total[k] == 0.0 && !a.multi) { sc->done[k] = 1; } // breakdown, as upstream
         }
         sc->first = 0;
      }
   }
}

bool vcg_slab_available(lgh_ctx *c)
{
   // (node vectors are addressed by one scalar base + a 32-bit byte offset)
   return c->dim == 3 && c->kid == 0x346 && (size_t)c->N * 8 * kVC < 0xffffffffull && ((size_t)c->NE * c->ND + kYePad) * 8 * (kVC - 1) + 65536 < 0xffffffffull && slab_swaps_ok(c); // (per-lane store offsets: the last component's plane + a set)
}

template <bool SYM, int WPS, int TR, bool WIDE, bool EX, bool DYN>
static void slab_launch(lgh_ctx *c, const VcgArgs &a, const int grid, const int nset)
{
   if (a.dqs == 0)
   {
      // compact mass data; with a tensor-product rule as well (a.M1): the Kronecker form
      if constexpr (SYM)
      {
         if (a.M1) { hipLaunchKernelGGL((vcg_apply_slab346<SYM, WPS, TR, WIDE, EX, DYN, true, true>), dim3(grid), dim3(256 * WPS), 0, c->stream, a, nset); return; }
      }
      hipLaunchKernelGGL((vcg_apply_slab346<SYM, WPS, TR, WIDE, EX, DYN, true>), dim3(grid), dim3(256 * WPS), 0, c->stream, a, nset);
   }
   else { hipLaunchKernelGGL((vcg_apply_slab346<SYM, WPS, TR, WIDE, EX, DYN, false>), dim3(grid), dim3(256 * WPS), 0, c->stream, a, nset); }
}

void launch_vcg_slab(lgh_ctx *c, const VcgArgs &a)
{
   if (c->ncu <= 0) // (per context: contexts of one process may sit on different devices)
   {
      hipDeviceProp_t prop;
      c->ncu = (hipGetDeviceProperties(&prop, c->device) == hipSuccess) ? prop.multiProcessorCount : 256;
   }
   const int ncu = c->ncu;
   const int wps = c->slab_wps; // wavefronts per SIMD (workgroup of 256 or 512 threads, one per CU)
   const int nset = ceil_div(c->NE, 5);
   const int grid = std::min(ceil_div(nset, 4 * wps), ncu); // one workgroup of 4 wps wavefronts per CU
   const bool wide = c->slab_wide && a.map_xrows != 0;
   const bool exact = a.limbs != nullptr;
   // Sets drawn from the workgroup's queue even out the wavefronts of a workgroup, which pays while a wavefront has few
   // passes (32^3: 6.4 each, 30.6 against 31.6 us); with many, the static interleaved schedule - the whole grid sweeps
   // through the mesh together - is ahead (64^3: 51 passes each, 210.8 against 222 us, profiles/r4_k1_forms.txt).
   const bool dyn = exact && (c->slab_dyn < 0 ? nset <= 16 * grid * 4 * wps : c->slab_dyn != 0);
#define LGH_SLAB_LAUNCH(SYM_, WPS_, TR_, WIDE_, EX_, DYN_) slab_launch<SYM_, WPS_, TR_, WIDE_, EX_, DYN_>(c, a, grid, nset)
#define LGH_SLAB_LAUNCH2(SYM_, WPS_, TR_) do { if (wide && dyn) { LGH_SLAB_LAUNCH(SYM_, WPS_, TR_, true, true, true); } else if (wide && exact) { LGH_SLAB_LAUNCH(SYM_, WPS_, TR_, true, true, false); } \
                                               else if (wide) { LGH_SLAB_LAUNCH(SYM_, WPS_, TR_, true, false, false); } \
                                               else if (exact) { LGH_SLAB_LAUNCH(SYM_, WPS_, TR_, false, true, false); } else { LGH_SLAB_LAUNCH(SYM_, WPS_, TR_, false, false, false); } } while (0)
   static const bool trace_full = getenv("LGH_VCG_TRACE_PHASES") != nullptr; // per-phase cycle counters as well (more registers: not the shipped schedule)
   // (the traced instantiations exist for the mirror-symmetric table only: a context without the symmetry runs untraced)
   const bool tr = a.trace != nullptr && c->b_h1_sym;
   if (tr && trace_full) { if (wps == 2) { LGH_SLAB_LAUNCH2(true, 2, 2); } else { LGH_SLAB_LAUNCH2(true, 1, 2); } }
   else if (tr) { if (wps == 2) { LGH_SLAB_LAUNCH2(true, 2, 1); } else { LGH_SLAB_LAUNCH2(true, 1, 1); } } // debug (LGH_VCG_TRACE): wall-clock stamps per workgroup
   else if (wps == 2) { if (c->b_h1_sym) { LGH_SLAB_LAUNCH2(true, 2, 0); } else { LGH_SLAB_LAUNCH2(false, 2, 0); } }
   else { if (c->b_h1_sym) { LGH_SLAB_LAUNCH2(true, 1, 0); } else { LGH_SLAB_LAUNCH2(false, 1, 0); } }
#undef LGH_SLAB_LAUNCH2
#undef LGH_SLAB_LAUNCH
}

} // namespace lgh
