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
static bool slab_swaps_ok(lgh_ctx *c)
{
   static int cached = -1;
   if (cached >= 0) { return cached == 1; }
   cached = 0;
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
   cached = good ? 1 : 0;
   return good;
}

template <bool SYM, int WPS, bool TRACE, bool WIDE>
__global__ void __launch_bounds__(256, WPS)
vcg_apply_slab346(const VcgArgs a, const int nset)
{
   constexpr int D = 4, Q = 6, NQ = Q * Q * Q, ND = D * D * D, QD = Q * D, HB = SYM ? (QD + 1) / 2 : QD;
   constexpr int ES = 5;                     // elements of a set: 15 items on the 16 lanes of a lane group
   constexpr int NW = 4;                     // wavefronts of a workgroup, each on its own sets
   constexpr int NDMA = (ES * NQ * 8 + 1023) / 1024; // 1 KB LDS-DMA pieces per set of quadrature data (9)
   constexpr int SBUF = NDMA * 128;          // doubles per LDS buffer (the last piece runs past the set's 1080 values)
   __shared__ double sDall[NW * 2 * SBUF];   // per wave: two buffers (the set being contracted, the set in flight)
   __shared__ double red[48];

   const int tid = threadIdx.x, lane = tid & 63;
   const int wid = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform: set indices and their base addresses stay in scalar registers
   const int g = lane >> 4, n = lane & 15;
   const int ni = min(n, 14), el = ni / 3, c = ni - 3 * el;
   double *sD = sDall + wid * (2 * SBUF);
   const int W = gridDim.x * NW;
   int s = xcd_swizzle(blockIdx.x, gridDim.x) * NW + wid;

   // No predicates on the loads of the pipeline (as in vcg_apply_plane): sets past the end re-read the last set, elements
   // past the end the last element, and the quadrature data is padded by one set behind its last element (lgh_create);
   // nothing of that is stored or summed.  The loop body is straight-line code.
   unsigned mo[16]; // byte offsets of this lane's 16 nodes (dx + 4 dy; dz = g) into a node vector
   auto load_map = [&](const int ss) {
      const int e = min(ES * min(ss, nset - 1) + el, a.NE - 1);
      const v4u *p = (const v4u *)(a.mapb + (size_t)e * ND + 16 * g);
#pragma unroll
      for (int dy = 0; dy < 4; dy++)
      {
         const v4u m = p[dy];
         mo[4 * dy + 0] = m[0]; mo[4 * dy + 1] = m[1]; mo[4 * dy + 2] = m[2]; mo[4 * dy + 3] = m[3];
      }
   };
   load_map(s); // in flight while the scalars are read

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
   const double betac = (c == 0) ? beta[0] : (c == 1) ? beta[1] : beta[2];

   double Bsr[HB];
#pragma unroll
   for (int i = 0; i < HB; i++) { Bsr[i] = uniform_f64(a.B[i]); }
   auto Bs = [&](const int idx) -> double { return (SYM && idx >= HB) ? Bsr[QD - 1 - idx] : Bsr[idx]; };
   // node vectors: one scalar base + 32-bit byte offsets (vcg_slab_available checks kVC * N * 8 < 2^32).  In the first
   // iteration (beta = 0) the old direction is not defined: r is read in its place and multiplied by zero.
   const unsigned coff = 8u * (unsigned)c * (unsigned)a.N;
   const double *dsrc = first ? a.r : a.d;
   __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the one-time loads are complete before the pipelined loop (see vcg_apply_plane)

   // WIDE: the four nodes of every x-row of an element are consecutive in the node vectors (checked on the map at set-up:
   // true for any tensor-product numbering with x fastest) - a row is two 16-byte loads per vector instead of four
   // 8-byte gathers, and every cache line is asked for by one or two instructions instead of four
   double gz[16], gd[16], gv[16];
   auto load_gather = [&]() {
      if (WIDE)
      {
#pragma unroll
         for (int dy = 0; dy < 4; dy++)
         {
            const v2d_a8 *pv = (const v2d_a8 *)((const char *)a.dinv + mo[4 * dy]);
            const v2d va = pv[0], vb = pv[1];
            gv[4 * dy] = va[0]; gv[4 * dy + 1] = va[1]; gv[4 * dy + 2] = vb[0]; gv[4 * dy + 3] = vb[1];
         }
#pragma unroll
         for (int dy = 0; dy < 4; dy++)
         {
            const v2d_a8 *pz = (const v2d_a8 *)((const char *)a.r + (mo[4 * dy] + coff));
            const v2d_a8 *pd = (const v2d_a8 *)((const char *)dsrc + (mo[4 * dy] + coff));
            const v2d za = pz[0], zb = pz[1], da = pd[0], db = pd[1];
            gz[4 * dy] = za[0]; gz[4 * dy + 1] = za[1]; gz[4 * dy + 2] = zb[0]; gz[4 * dy + 3] = zb[1];
            gd[4 * dy] = da[0]; gd[4 * dy + 1] = da[1]; gd[4 * dy + 2] = db[0]; gd[4 * dy + 3] = db[1];
         }
         return;
      }
#pragma unroll
      for (int j = 0; j < 16; j++) { gv[j] = slab_ld(a.dinv, mo[j]); }
#pragma unroll
      for (int j = 0; j < 16; j++)
      {
         gz[j] = slab_ld(a.r, mo[j] + coff);
         gd[j] = slab_ld(dsrc, mo[j] + coff);
      }
   };
   // quadrature data of a set: LDS-DMA, 16 bytes per lane and piece, straight into the wave's buffer `buf` (no registers)
   auto load_dq = [&](const int ss, const int buf) {
      const double *p = a.Dq + (size_t)min(ss, nset - 1) * (ES * NQ) + 2 * lane;
      double *l = sD + buf * SBUF;
#pragma unroll
      for (int k = 0; k < NDMA; k++)
      {
         __builtin_amdgcn_global_load_lds(p + 128 * k, (__attribute__((address_space(3))) void *)(l + 128 * k), 16, 0, 0);
      }
   };
   // direction d = z + beta d (K2 stores the same values)
   double dd[16];
   auto convert = [&]() {
#pragma unroll
      for (int j = 0; j < 16; j++) { dd[j] = fma(betac, gd[j], __dmul_rn(gz[j], gv[j])); }
   };

   double dot = 0.0;
   load_gather();
   load_dq(s, 0);
   load_map(s + W);
   __builtin_amdgcn_s_waitcnt(0x0F70);
   convert();
   // debug (LGH_VCG_TRACE): wall-clock stamps of wave 0 and the shader cycles it spends waiting for the loads of a set
   unsigned long long t_start = 0, t_loop = 0, c_wait = 0, c_loop = 0;
   unsigned long long c_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c_prev = 0; // shader cycles per phase of the loop body (wave 0)
   if (TRACE) { t_start = wall_clock64(); c_loop = clock64(); c_prev = c_loop; }
#define LGH_SLAB_STAMP(K_) do { if (TRACE) { const unsigned long long c_now = clock64(); c_ph[K_] += c_now - c_prev; c_prev = c_now; } } while (0)
   int buf = 0;
   for (; s < nset; s += W, buf ^= 1)
   {
      const int e = ES * s + el;
      const bool act = (n < 15) && (e < a.NE) && mine;
      const double actf = act ? 1.0 : 0.0;
      const double *sDp = sD + buf * SBUF + el * NQ + 9 * g; // this lane's nine (qx, qy) pairs: sDp[i + 36 qz]
      // next set: gathers and quadrature data now, the map of the one after
      load_gather();
      load_dq(s + W, buf ^ 1);
      load_map(s + 2 * W);
      LGH_SLAB_STAMP(0); // issue of the loads
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
      LGH_SLAB_STAMP(1); // forward x, y
      // slabs -> pairs: w[9 dz + i] = slab dz, pair 9 g + i
      slab_to_pairs(w);
      __builtin_amdgcn_sched_barrier(0);
      LGH_SLAB_STAMP(2); // transpose
      // forward z, quadrature data, (d, A d), backward z; the quadrature data of pair i + 1 is read while pair i is contracted
      double dset = 0.0;
      double dcur[Q], dnxt[Q];
#pragma unroll
      for (int qz = 0; qz < Q; qz++) { dcur[qz] = sDp[36 * qz]; }
#pragma unroll
      for (int i = 0; i < 9; i++)
      {
         if (i < 8)
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
#pragma unroll
         for (int qz = 0; qz < Q; qz++) { dcur[qz] = dnxt[qz]; }
         asm volatile("" : "+v"(dset)); // the partial sum exists HERE: left alone, the compiler keeps all 54 factor pairs alive (216 registers) and forms the sum after the loop body
         __builtin_amdgcn_sched_barrier(0);
      }
      LGH_SLAB_STAMP(3); // z
      // pairs -> slabs
      slab_to_slabs(w);
      __builtin_amdgcn_sched_barrier(0);
      LGH_SLAB_STAMP(4); // transpose back
      // backward y and x, one x-index at a time: t[dy] = sum_qy B[qy,dy] w[qx + 6 qy]; out[dx + 4 dy] += B[qx,dx] t[dy]
      double o[16];
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
      if (TRACE)
      {
         const unsigned long long c0 = clock64();
         __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
         c_wait += clock64() - c0;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): gathers, LDS-DMA and map of the next set
      convert();
      __builtin_amdgcn_sched_barrier(0);
      LGH_SLAB_STAMP(6); // wait + direction of the next set
      // the slab of the E-vector: 16 contiguous doubles
      if (act)
      {
         double *yc = a.YE + (size_t)c * a.ye_stride + (size_t)ND * e + 16 * g;
#pragma unroll
         for (int dy = 0; dy < D; dy++) { *(v4d *)(yc + 4 * dy) = v4d{o[4 * dy], o[4 * dy + 1], o[4 * dy + 2], o[4 * dy + 3]}; }
      }
      LGH_SLAB_STAMP(7); // stores
      dot = fma(dset, actf, dot); // (a select would let the compiler sink all 54 products of dset behind the branch: 216 live registers)
   }
   if (TRACE) { t_loop = wall_clock64(); c_loop = clock64() - c_loop; }
   double bp[kVC];
   block_sum3(c == 0 ? dot : 0.0, c == 1 ? dot : 0.0, c == 2 ? dot : 0.0, red, bp);
   double total[kVC];
   const bool last = grid_sum3_last_block_flat(bp, a.partials, a.stride, a.ticket, red, total);
   if (TRACE && tid == 0)
   {
      a.trace[kTraceRec * blockIdx.x + 0] = t_start;
      a.trace[kTraceRec * blockIdx.x + 1] = t_loop;
      a.trace[kTraceRec * blockIdx.x + 2] = wall_clock64();
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
   return c->dim == 3 && c->kid == 0x346 && (size_t)c->N * 8 * kVC < 0xffffffffull && slab_swaps_ok(c);
}

void launch_vcg_slab(lgh_ctx *c, const VcgArgs &a)
{
   static int ncu = 0, wps = 0;
   if (ncu == 0)
   {
      hipDeviceProp_t prop;
      ncu = (hipGetDeviceProperties(&prop, c->device) == hipSuccess) ? prop.multiProcessorCount : 256;
      const char *env = getenv("LGH_SLAB_WPS"); // A/B: workgroups (of four wavefronts) per CU
      wps = (env && env[0] == '2') ? 2 : 1;
   }
   const int nset = ceil_div(c->NE, 5);
   const int grid = std::min(ceil_div(nset, 4), wps * ncu);
   static const bool wide_env = !(getenv("LGH_SLAB_WIDE") && getenv("LGH_SLAB_WIDE")[0] == '0'); // A/B
   const bool wide = wide_env && a.map_xrows != 0;
#define LGH_SLAB_LAUNCH(SYM_, WPS_, TR_, WIDE_) hipLaunchKernelGGL((vcg_apply_slab346<SYM_, WPS_, TR_, WIDE_>), dim3(grid), dim3(256), 0, c->stream, a, nset)
#define LGH_SLAB_LAUNCH2(SYM_, WPS_, TR_) do { if (wide) { LGH_SLAB_LAUNCH(SYM_, WPS_, TR_, true); } else { LGH_SLAB_LAUNCH(SYM_, WPS_, TR_, false); } } while (0)
   if (a.trace) { if (wps == 2) { LGH_SLAB_LAUNCH2(true, 2, true); } else { LGH_SLAB_LAUNCH2(true, 1, true); } } // debug (LGH_VCG_TRACE): per-phase cycle counters
   else if (wps == 2) { if (c->b_h1_sym) { LGH_SLAB_LAUNCH2(true, 2, false); } else { LGH_SLAB_LAUNCH2(false, 2, false); } }
   else { if (c->b_h1_sym) { LGH_SLAB_LAUNCH2(true, 1, false); } else { LGH_SLAB_LAUNCH2(false, 1, false); } }
#undef LGH_SLAB_LAUNCH2
#undef LGH_SLAB_LAUNCH
}

} // namespace lgh
