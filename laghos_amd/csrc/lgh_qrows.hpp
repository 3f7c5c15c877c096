#pragma once
// lgh_qrows.hpp — the 3D quadrature-data update with ROW-OWNED contraction stages (round 4; Q1Q0 .. Q4Q3).
//
// Replaces the same reference code as the update mode of qpoint_kernel (lgh_qpoint.hpp): the five passes of
// QUpdate::UpdateQuadratureData (/root/reference/laghos_solver.cpp:1365-1373), QKernel / QUpdateBody (:1263-1352,
// :1042-1168) and, fused, ForcePA->Mult(one) and ForcePA->MultTranspose(v) of the same state
// (/root/reference/laghos_assembly.cpp:296-514, :715-924).  The point-wise body is the same function (qpoint_body).
//
// Why a second form.  In qpoint_kernel every thread owns ONE OUTPUT of each sum-factorisation stage: an output costs
// D (or Q) LDS reads of data, as many of the 1-D table, index arithmetic per item and 2..3 FMAs per read - at Q3Q2
// ~950 of the kernel's ~2 100 vector instructions per wavefront are spent on 235 FMAs of contraction per point
// (profiles/r3: the kernel is bound by fp64 VALU issue, not by bytes).  Here a thread owns one INPUT ROW of a stage:
// it reads the D (or Q) values of the row with wide LDS loads (the rows are contiguous and 16-byte aligned by
// layout), keeps the whole 1-D table in registers (one base pointer per thread selects B or G: every item of a stage
// runs the same instructions, no divergence between the B and G parts) and produces ALL outputs of the row with
// compile-time table indices: per row D*Q FMAs, (D + Q*D)/2 wide loads, Q stores, a dozen address instructions.
// Stages (one workgroup = one element, Q^3 threads, thread (qx, qy, qz) owns a point in the body):
//   gather x, v (6 fields), e          -> sU  [f][dz][dy][dx], sE
//   X   rows (B|G, f, dz, dy)          -> sX  [B|G][f][qx][dz][dy]
//   Y   rows (GB|BG|BB, f, qx, dz)     -> sY  [part][f][qy][qx][dz]
//   Z   per point: 18 dot products of D (rows of sY, table row qz), the point body, stress out (9 coalesced planes)
//   F.1: stress -> sF [c][gd][qy][qx][qz];  BZ rows (c, gd, qy, qx) -> sA [k][dz][qx][qy];  BY rows (k, dz, qx)
//        -> sW [k][dz][dy][qx];  BX rows (c, dz, dy): sum over gd and qx -> E-vector
//   F^T v: the point integrand tested with the L2 basis, z -> y -> x, rows likewise (a few dozen rows per stage)
// 6 barriers instead of 9-11; the LDS slice is 41 KB at Q3Q2 (three workgroups per CU as before).
// The element a workgroup takes is XCD-swizzled (xcd_swizzle): x-neighbours share node lines, and workgroups of
// consecutive index sit on different XCDs, i.e. behind different L2s (profiles/r3_pmc_traffic.json: 1.30 x the
// algorithmic bytes fetched by the unswizzled kernel).
#include "lgh_qpoint.hpp"

namespace lgh
{

template <int D, int Q, int L> struct QRows
{
   static constexpr int ND = D * D * D, NQ = Q * Q * Q, NL = L * L * L, NF = 6;
   static constexpr int SU = NF * ND;             // gathered fields
   static constexpr int SX = NF * Q * D * D;      // one of the two x-contracted arrays
   static constexpr int SFS = 9 * NQ + NQ;        // stress (9 planes) + integrand of F^T v
   static constexpr int R1 = (SU + 2 * SX > SFS) ? SU + 2 * SX : SFS; // also sW: 9*D*D*Q
   static constexpr int SY = 3 * NF * Q * Q * D;  // also sA: 9*D*Q*Q
   static constexpr int SE = NL + L * L * Q + L * Q * Q;
   static constexpr int ST = 2 * Q * D + Q * L;   // tables, q-major
   static constexpr int TOTAL = R1 + SY + SE + ST;
};

// out[q] = sum_d T[q*NIN + d] * in[d]: one row of a forward stage (NIN inputs -> NOUT outputs)
template <int NIN, int NOUT>
__device__ __forceinline__ void row_fwd(const double *__restrict__ T, const double (&in)[NIN], double (&out)[NOUT])
{
#pragma unroll
   for (int q = 0; q < NOUT; q++)
   {
      double s = T[q * NIN] * in[0];
#pragma unroll
      for (int d = 1; d < NIN; d++) { s = fma(T[q * NIN + d], in[d], s); }
      out[q] = s;
   }
}
// out[d] = sum_q T[q*NOUT + d] * in[q]: one row of a transposed stage (NIN = Q inputs -> NOUT = D outputs)
template <int NIN, int NOUT>
__device__ __forceinline__ void row_bwd(const double *__restrict__ T, const double (&in)[NIN], double (&out)[NOUT])
{
#pragma unroll
   for (int d = 0; d < NOUT; d++)
   {
      double s = T[d] * in[0];
#pragma unroll
      for (int q = 1; q < NIN; q++) { s = fma(T[q * NOUT + d], in[q], s); }
      out[d] = s;
   }
}
template <int N> __device__ __forceinline__ void row_load(const double *__restrict__ p, double (&v)[N])
{
#pragma unroll
   for (int i = 0; i < N; i++) { v[i] = p[i]; }
}

// MINW = 4: the same code under a cap of 128 registers (four workgroups per CU at Q3Q2 instead of three), launched for
// contexts that run WITHOUT artificial viscosity (3D Taylor-Green).  The cap costs the full body 27 spilled registers,
// and the register allocator places them in and around the viscosity branch - the one part such a context never
// executes: 64^3 TG 3.42 -> 3.10 ms per call.  With viscosity the same build loses (614 -> 694 us at C2), and an
// instantiation WITHOUT the branch needs more registers, not fewer (177 uncapped; capped it spills 49 into code that
// runs: 5.10 ms) - profiles/r4_q_occupancy.txt.  So: one source, two register budgets, chosen by a.visc.
// TRACE (debug, LGH_Q_TRACE=<file>): thread 0 of every workgroup stamps the 100 MHz wall clock at the stage boundaries
// (16 words per workgroup: 12 stamps, then xcc << 32 | hw_id) - where a workgroup's life goes, and what shares a CU
// with what (tools/q_trace_summary.py).  An instantiation of its own: the production kernel carries none of it.
constexpr int kQTraceRec = 16;
// J0C (round 5): Jac0inv is the same at every point of a zone (checked at lgh_setup_rho0detj0; a.Jac0inv_e): nine doubles per zone
// in scalar registers instead of nine per point in vector registers - 15.5 of the 20.5 KB a zone streams at Q3Q2, and the
// first thing a workgroup waits for (profiles/r5_q_map_formula_negative.txt: its load phase is bound by this stream).
template <int D, int Q, int L, int MINW, bool TRACE = false, bool J0C = false>
__global__ void __launch_bounds__(Q *Q *Q, MINW)
qrows_kernel(const QArgs a, unsigned long long *trace = nullptr)
{
   unsigned long long tstamp[12];
#define LGH_QSTAMP(K_) do { if (TRACE) { tstamp[K_] = wall_clock64(); } } while (0)
   LGH_QSTAMP(0);
   using S = QRows<D, Q, L>;
   constexpr int ND = S::ND, NQ = S::NQ, NL = S::NL, NF = S::NF, NT = NQ;
   constexpr int DD = D * D, QQ = Q * Q;
   __shared__ __attribute__((aligned(16))) double smem[S::TOTAL];
   double *const sR1 = smem;
   double *const sU = sR1;                       // [f][dz][dy][dx]
   double *const sX = sR1 + S::SU;               // [B|G][f][qx][dz][dy]
   double *const sF = sR1;                       // [k = gd + 3c][qy][qx][qz]   (after the Y stage)
   double *const sS = sR1 + 9 * NQ;              // [qy][qx][qz]: integrand of F^T v
   double *const sW = sR1;                       // [k][dz][dy][qx]             (after the BZ stage)
   double *const sY = smem + S::R1;              // [part][f][qy][qx][dz]
   double *const sA = sY;                        // [k][dz][qx][qy]             (after the body)
   double *const sTB = sY + S::SY;               // B[q][d]   (tables first: their offsets are even, rows of D = 4 load as 16-byte pairs)
   double *const sTG = sTB + Q * D;              // G[q][d]
   double *const sTL = sTG + Q * D;              // Bl[q][l]
   double *const sE = sTL + Q * L;               // [lz][ly][lx]
   double *const sE1 = sE + NL;                  // [lz][ly][qx]   (F^T v: [lz][ly][qx] again, on the way back)
   double *const sE2 = sE1 + L * L * Q;          // [lz][qy][qx]

   const int lt = threadIdx.x;
   const int tx = lt % Q, ty = (lt / Q) % Q, tz = lt / QQ;
   // element of this workgroup: workgroup b runs on XCD b % 8 (observed); swz = log2 of the run of consecutive elements
   // that stay on one XCD (runs are dealt to the XCDs in turn); swz < 0: one contiguous eighth of the mesh per XCD
   int e = blockIdx.x;
   if (a.q_swz < 0) { e = xcd_swizzle(blockIdx.x, gridDim.x); }
   else if (a.q_swz > 0)
   {
      const int R = 1 << a.q_swz, span = 8 * R, b = blockIdx.x;
      if (b < (int)(gridDim.x / span) * span) { e = (b / span) * span + (b & 7) * R + ((b >> 3) & (R - 1)); }
   }
   // (the zones are walked in the library's own order; every per-zone array is the caller's and is read / written at the caller's place)
   if (a.zorder) { e = __builtin_amdgcn_readfirstlane(a.zorder[min(e, a.NE - 1)]); }
   const size_t eq = (size_t)e * NQ + lt;
   const size_t plane = (size_t)a.NE * NQ;

   // ---- P0: tables (q-major), gathers, point data: every global read of the element before the first barrier
   for (int i = lt; i < Q * D; i += NT)
   {
      const int q = i / D, d = i - q * D;
      sTB[i] = a.B[q + Q * d];
      sTG[i] = a.G[q + Q * d];
   }
   for (int i = lt; i < Q * L; i += NT)
   {
      const int q = i / L, l = i - q * L;
      sTL[i] = a.Bl[q + Q * l];
   }
   for (int i = lt; i < 3 * ND; i += NT)
   {
      const int c = i / ND, d = i - c * ND;
      const size_t n = (size_t)c * a.N + a.map[(size_t)e * ND + d];
      sU[i] = a.x[n];
      sU[i + 3 * ND] = a.v[n];
   }
   for (int i = lt; i < NL; i += NT) { sE[i] = a.e[(size_t)e * NL + i]; }
   double J0i[9];
   if (J0C)
   {
#pragma unroll
      for (int k = 0; k < 9; k++) { J0i[k] = uniform_f64(a.Jac0inv_e[(size_t)9 * e + k]); } // (e is wave-uniform: scalar registers)
   }
   else
   {
#pragma unroll
      for (int k = 0; k < 9; k++) { J0i[k] = a.Jac0inv_soa[eq + plane * k]; }
   }
   const double rdw = a.rho0DetJ0w_in[eq];
   const double weight = a.W[lt];
   LGH_QSTAMP(1); // loads issued
   __syncthreads();
   LGH_QSTAMP(2); // gathers, tables in LDS

   // ---- P1: X stage, rows (which, f, dz, dy); the L2 field's x stage on the last threads
   for (int i = lt; i < 2 * NF * DD; i += NT)
   {
      const int which = i / (NF * DD), r = i - which * (NF * DD); // r = dy + D*(dz + D*f)
      const double *T = sTB + which * (Q * D); // (one address, not a select between two loaded tables)
      double u[D], o[Q];
      row_load<D>(sU + D * r, u);
      row_fwd<D, Q>(T, u, o);
      const int f = r / DD, zy = r - f * DD;
      double *dst = sX + which * S::SX + f * (Q * DD) + zy;
#pragma unroll
      for (int q = 0; q < Q; q++) { dst[q * DD] = o[q]; }
   }
   {
      const int j = NT - 1 - lt;
      if (j < L * L)
      {
         double u[L], o[Q];
         row_load<L>(sE + L * j, u);
         row_fwd<L, Q>(sTL, u, o);
#pragma unroll
         for (int q = 0; q < Q; q++) { sE1[j * Q + q] = o[q]; }
      }
   }
   __syncthreads();
   LGH_QSTAMP(3); // X stage

   // ---- P2: Y stage, rows (part, f, qx, dz): part 0 = B on the G array (d/dx), 1 = G on the B array (d/dy), 2 = B on B (for d/dz)
   for (int i = lt; i < 3 * NF * Q * D; i += NT)
   {
      const int part = i / (NF * Q * D), r = i - part * (NF * Q * D); // r = dz + D*(qx + Q*f)
      const double *T = sTB + ((part == 1) ? Q * D : 0);
      double u[D], o[Q];
      row_load<D>(sX + (part == 0 ? S::SX : 0) + D * r, u);
      row_fwd<D, Q>(T, u, o);
      const int f = r / (Q * D), xz = r - f * (Q * D);
      double *dst = sY + (part * NF + f) * (QQ * D) + xz;
#pragma unroll
      for (int q = 0; q < Q; q++) { dst[q * (Q * D)] = o[q]; }
   }
   {
      const int j = NT - 1 - lt;
      if (j < L * Q)
      {
         const int qx = j % Q, lz = j / Q;
         double u[L], o[Q];
#pragma unroll
         for (int ly = 0; ly < L; ly++) { u[ly] = sE1[(lz * L + ly) * Q + qx]; }
         row_fwd<L, Q>(sTL, u, o);
#pragma unroll
         for (int q = 0; q < Q; q++) { sE2[(lz * Q + q) * Q + qx] = o[q]; }
      }
   }
   __syncthreads();
   LGH_QSTAMP(4); // Y stage

   // ---- P3: Z stage of this thread's point and the point-wise body
   double J[9], dV[9], e_val = 0.0;
   {
      double tb[D], tg[D];
      row_load<D>(sTB + D * tz, tb);
      row_load<D>(sTG + D * tz, tg);
      const double *col = sY + (ty * Q + tx) * D;
#pragma unroll
      for (int f = 0; f < NF; f++)
      {
         double gb[D], bg[D], bb[D];
         row_load<D>(col + (0 * NF + f) * (QQ * D), gb);
         row_load<D>(col + (1 * NF + f) * (QQ * D), bg);
         row_load<D>(col + (2 * NF + f) * (QQ * D), bb);
         double d0 = tb[0] * gb[0], d1 = tb[0] * bg[0], d2 = tg[0] * bb[0];
#pragma unroll
         for (int dz = 1; dz < D; dz++)
         {
            d0 = fma(tb[dz], gb[dz], d0);
            d1 = fma(tb[dz], bg[dz], d1);
            d2 = fma(tg[dz], bb[dz], d2);
         }
         // column-major [c + 3*d] = d u_c / d xi_d; fields 0..2 = x, 3..5 = v
         double *M = (f < 3) ? J : dV;
         const int c = (f < 3) ? f : f - 3;
         M[c] = d0;
         M[c + 3] = d1;
         M[c + 6] = d2;
      }
#pragma unroll
      for (int lz = 0; lz < L; lz++) { e_val = fma(sTL[tz * L + lz], sE2[(lz * Q + ty) * Q + tx], e_val); }
   }
   LGH_QSTAMP(5); // Z stage (this wavefront)
   double ftv = 0.0, sjw[9];
   const double cand = qpoint_body<3>(a, e, eq, weight, J, dV, e_val, plane, J0i, rdw, ftv, sjw);
   LGH_QSTAMP(6); // point body (this wavefront)
   tstamp[7] = tstamp[8] = tstamp[9] = tstamp[10] = 0;

   const bool do_f = (a.force_e != nullptr), do_t = (a.erhs_q != nullptr);
   if (do_f || do_t)
   {
      // the x-contracted arrays are dead since the barrier above: the stress goes straight to its place
      const int pq = (ty * Q + tx) * Q + tz;
      if (do_f)
      {
#pragma unroll
         for (int k = 0; k < 9; k++) { sF[k * NQ + pq] = sjw[k]; } // sjw[gd + 3*c]
      }
      if (do_t) { sS[pq] = ftv; }
      __syncthreads();
      LGH_QSTAMP(7); // every wavefront's body done, stress in LDS

      // ---- P5: contraction over qz, rows (k, qy, qx); F^T v: rows (qy, qx)
      if (do_f)
      {
         for (int i = lt; i < 9 * QQ; i += NT)
         {
            const int k = i / QQ, r = i - k * QQ; // r = qx + Q*qy
            const double *T = sTB + (((k % 3) == 2) ? Q * D : 0);
            double u[Q], o[D];
            row_load<Q>(sF + (size_t)i * Q, u);
            row_bwd<Q, D>(T, u, o);
            const int qx = r % Q, qy = r / Q;
            double *dst = sA + k * (D * QQ) + qx * Q + qy;
#pragma unroll
            for (int d = 0; d < D; d++) { dst[d * QQ] = o[d]; }
         }
      }
      if (do_t)
      {
         const int j = NT - 1 - lt;
         if (j < QQ)
         {
            double u[Q], o[L];
            row_load<Q>(sS + j * Q, u);
            row_bwd<Q, L>(sTL, u, o);
#pragma unroll
            for (int l = 0; l < L; l++) { sE2[l * QQ + j] = o[l]; } // [lz][qy][qx]
         }
      }
      __syncthreads();
      LGH_QSTAMP(8); // qz contraction

      // ---- P6: contraction over qy, rows (k, dz, qx); F^T v: rows (lz, qx)
      if (do_f)
      {
         for (int i = lt; i < 9 * D * Q; i += NT)
         {
            const int k = i / (D * Q);
            const double *T = sTB + (((k % 3) == 1) ? Q * D : 0);
            double u[Q], o[D];
            row_load<Q>(sA + (size_t)i * Q, u);
            row_bwd<Q, D>(T, u, o);
            const int r = i - k * (D * Q), qx = r % Q, dz = r / Q;
            double *dst = sW + ((k * D + dz) * D) * Q + qx;
#pragma unroll
            for (int d = 0; d < D; d++) { dst[d * Q] = o[d]; }
         }
      }
      if (do_t)
      {
         const int j = NT - 1 - lt;
         if (j < L * Q)
         {
            const int qx = j % Q, lz = j / Q;
            double u[Q], o[L];
#pragma unroll
            for (int qy = 0; qy < Q; qy++) { u[qy] = sE2[(lz * Q + qy) * Q + qx]; }
            row_bwd<Q, L>(sTL, u, o);
#pragma unroll
            for (int l = 0; l < L; l++) { sE1[(lz * L + l) * Q + qx] = o[l]; }
         }
      }
      __syncthreads();
      LGH_QSTAMP(9); // qy contraction

      // ---- P7: contraction over qx and the sum over the three reference directions, rows (c, dz, dy) -> E-vector;
      //          F^T v: rows (lz, ly) -> L2 vector
      if (do_f)
      {
         const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;
         for (int i = lt; i < 3 * DD; i += NT)
         {
            const int c = i / DD, r = i - c * DD; // r = dy + D*dz
            double wg[Q], w1[Q], w2[Q], og[D], ob[D];
            row_load<Q>(sW + ((3 * c + 0) * DD + r) * Q, wg);
            row_load<Q>(sW + ((3 * c + 1) * DD + r) * Q, w1);
            row_load<Q>(sW + ((3 * c + 2) * DD + r) * Q, w2);
#pragma unroll
            for (int q = 0; q < Q; q++) { w1[q] += w2[q]; }
            row_bwd<Q, D>(sTG, wg, og);
            row_bwd<Q, D>(sTB, w1, ob);
            double *dst = a.force_e + (size_t)ND * (c + 3 * (size_t)e) + D * r;
#pragma unroll
            for (int d = 0; d < D; d++)
            {
               double v = og[d] + ob[d];
               if (fabs(v) < eps2) { v = 0.0; } // laghos_assembly.cpp:495-512
               dst[d] = v;
            }
         }
      }
      if (do_t)
      {
         const int j = NT - 1 - lt;
         if (j < L * L)
         {
            double u[Q], o[L];
            row_load<Q>(sE1 + j * Q, u);
            row_bwd<Q, L>(sTL, u, o);
#pragma unroll
            for (int l = 0; l < L; l++) { a.erhs_q[(size_t)e * NL + j * L + l] = o[l]; }
         }
      }
   }
   LGH_QSTAMP(10); // qx contraction, outputs stored
   // q_dt_est = qdata.dt_est; Min() (:1374, :1406).  Round 5: no workgroup reduction, no ticket, no last workgroup - the
   // stage trace (profiles/r5_q_stage_trace.txt) showed a workgroup spending the last tenth of its life (1.2 us) in the
   // two dependent atomic round trips of the ticketed fold while it held its third of the CU.  The candidates are
   // non-negative doubles (or +inf), whose order is the order of their bit patterns as unsigned integers: every
   // wavefront folds its lanes on the DPP path and sends ONE non-returning 64-bit unsigned atomic min to one of kDtSlots
   // partial minima (by workgroup: ~130 000 atomics per call on ONE word cost more than the fold they replaced - 620 ->
   // 850 us, profiles/r5_q_stage_trace.txt; spread over 256 lines they cost nothing); lgh_get_dt_est folds the slots.
   // A minimum does not depend on the order of its operands: the same bits as the ordered fold.
   {
      const int lane = lt & 63, nact = min(64, NT - (lt & ~63));
      const double wmin = wave_min(cand, lane, nact);
      if (lane == 0 && wmin < __builtin_inf())
      {
         // (unsigned order = numeric order only from +0.0 up: lgh_create refuses cfl <= 0, so a candidate is cfl / inv_dt >= 0 or
         //  the 0 of an inverted zone; a -0.0 goes in as +0.0 - round-5 advisor)
         const double w0 = (wmin > 0.0) ? wmin : 0.0;
         unsigned long long *slot = (unsigned long long *)(a.result + kDtSlotStride * (1 + (blockIdx.x % kDtSlots)));
         (void)__hip_atomic_fetch_min(slot, (unsigned long long)__double_as_longlong(w0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
   }
   if (TRACE && lt == 0 && trace)
   {
      LGH_QSTAMP(11);
      unsigned xcc = 0, hwid = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      unsigned long long *rec = trace + (size_t)kQTraceRec * blockIdx.x;
      for (int k = 0; k < 12; k++) { rec[k] = tstamp[k]; }
      rec[12] = ((unsigned long long)xcc << 32) | hwid;
      rec[13] = (unsigned long long)e;
   }
#undef LGH_QSTAMP
}

// kernel ids the row form is instantiated for (3D; Q5Q4 keeps the point form with two points per thread: the slices of
// a 1000-point element do not fit the LDS of a CU twice over)
static bool qrows_available(const lgh_ctx *c)
{
   if (c->dim != 3) { return false; }
   switch (c->kid)
   {
      case 0x322: case 0x334: case 0x346: case 0x358: return true;
   }
   return false;
}
// debug: LGH_Q_TRACE=<file>: the Q3Q2 update through the traced instantiation; the stamps of call number LGH_Q_TRACE_CALL
// (default 40: inside the timed window of a bench run) are written to the file (qrows_trace_dump)
// (the buffer and the call counter live in the context - round-5 advisor: file-scope statics were shared by every context)
static void qrows_trace_dump(lgh_ctx *c)
{
   const char *path = getenv("LGH_Q_TRACE");
   if (!path || !c->q_trace_dev || c->q_trace_n <= 0) { return; }
   (void)hipStreamSynchronize(c->stream);
   std::vector<unsigned long long> h((size_t)kQTraceRec * c->q_trace_n);
   if (hipMemcpy(h.data(), c->q_trace_dev, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { return; }
   FILE *f = fopen(path, "w");
   if (!f) { return; }
   for (int i = 0; i < c->q_trace_n; i++)
   {
      fprintf(f, "%d", i);
      for (int k = 0; k < 14; k++) { fprintf(f, " %llu", h[(size_t)kQTraceRec * i + k]); }
      fprintf(f, "\n");
   }
   fclose(f);
}
template <int MINW6> static int launch_qrows_w(lgh_ctx *c, const QArgs &a)
{
   if (c->kid == 0x346 && getenv("LGH_Q_TRACE"))
   {
      if (c->q_trace_n != c->NE)
      {
         if (c->q_trace_dev) { (void)hipFree(c->q_trace_dev); c->q_trace_dev = nullptr; }
         LGH_HIP_CHECK(hipMalloc((void **)&c->q_trace_dev, (size_t)kQTraceRec * c->NE * 8));
         c->q_trace_n = c->NE;
      }
      if (a.Jac0inv_e) { hipLaunchKernelGGL((qrows_kernel<4, 6, 3, MINW6, true, true>), dim3(c->NE), dim3(216), 0, c->stream, a, c->q_trace_dev); }
      else { hipLaunchKernelGGL((qrows_kernel<4, 6, 3, MINW6, true, false>), dim3(c->NE), dim3(216), 0, c->stream, a, c->q_trace_dev); }
      LGH_HIP_CHECK(hipGetLastError());
      const char *nenv = getenv("LGH_Q_TRACE_CALL"); // which call to dump (default 40: a developed bench window)
      if (++c->q_trace_calls == (nenv ? atoi(nenv) : 40)) { qrows_trace_dump(c); }
      return LGH_OK;
   }
#define LGH_QROWS(D_, Q_, L_, W_, NT_) do { if (a.Jac0inv_e) { hipLaunchKernelGGL((qrows_kernel<D_, Q_, L_, W_, false, true>), dim3(c->NE), dim3(NT_), 0, c->stream, a, (unsigned long long *)nullptr); } \
                                           else { hipLaunchKernelGGL((qrows_kernel<D_, Q_, L_, W_, false, false>), dim3(c->NE), dim3(NT_), 0, c->stream, a, (unsigned long long *)nullptr); } } while (0)
   switch (c->kid)
   {
      case 0x322: LGH_QROWS(2, 2, 1, 1, 8); break;
      case 0x334: LGH_QROWS(3, 4, 2, 1, 64); break;
      case 0x346: LGH_QROWS(4, 6, 3, MINW6, 216); break;
      case 0x358: LGH_QROWS(5, 8, 4, 1, 512); break;
      default: return unknown_kernel(c->kid);
   }
#undef LGH_QROWS
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
static int launch_qrows(lgh_ctx *c, const QArgs &a)
{
   const char *oenv = getenv("LGH_Q_OCC4"); // A/B: 0 = the three-wavefront build for every context, 1 = the four-wavefront build
   // (round 5: with one Jac0inv per zone in scalar registers the 128-register build spills 11 instead of 26 and wins with
   //  viscosity as well - 591 -> 584 us at C2, 3.96 -> 3.75 ms at 64^3 Sedov, profiles/r5_q_jac0inv_ab.txt)
   const bool w4 = oenv ? oenv[0] == '1' : (!a.visc || a.Jac0inv_e != nullptr);
   return w4 ? launch_qrows_w<4>(c, a) : launch_qrows_w<1>(c, a);
}

} // namespace lgh
