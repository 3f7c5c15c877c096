#pragma once
// lgh_qpoint.hpp — the quadrature-point kernel (update, set-up and energy-integral modes) shared by lgh_qupdate.hip
// (update / integrals, compiled with the relaxed fp64 division of the Makefile) and lgh_qsetup.hip (Rho0DetJ0Vol: IEEE
// division - the set-up data is computed once and read by every kernel afterwards).  Quadrature-data update, initial
// geometric data and energy integrals for gfx950.
//
// Replaces QUpdate::UpdateQuadratureData + QKernel/QUpdateBody
// (/root/reference/laghos_solver.cpp:1354-1411, :1263-1352, :1042-1168),
// Rho0DetJ0Vol (:1170-1261) and InternalEnergy/KineticEnergy (:640-697).
//
// MI355X design: the reference makes five global-memory round trips before the
// physics kernel starts (E-vector of x, 9*NQ Jacobians, E-vector of v, 9*NQ
// velocity gradients, NQ energies: laghos_solver.cpp:1365-1373).  Here one
// workgroup per element gathers x, v, e straight from the state L-vector into
// LDS, evaluates the reference gradients by sum factorisation in LDS, runs the
// point-wise EOS / viscosity / time-step body in registers (one thread per
// quadrature point) and streams the nine stressJinvT planes out coalesced.  The
// running dt estimate is folded on the device (block min -> last-block fold).
// Algorithmic traffic 36 KB/element instead of >= 100 KB.
#include "lgh_common.hpp"
#include "lgh_smallmat.hpp"

namespace lgh
{

enum { QMODE_UPDATE = 0, QMODE_SETUP = 1, QMODE_IE = 2, QMODE_KE = 3 };

struct QArgs
{
   int NE, N;
   const double *B, *G, *Bl, *W;
   const int *map;
   const double *x, *v, *e; // L-vectors (byNODES for x, v)
   const double *gamma;
   const double *rho0DetJ0w_in;
   const double *Jac0inv_in;
   const double *Jac0inv_soa; // [q + NQ*e + NE*NQ*k], k = i + dim*j (internal copy)
   const double *Jac0inv_e;   // row form of the update: [k + dim*dim*e] where Jac0inv is the same at every point of a zone (else nullptr)
   double *Jac0inv_soa_out;
   double *stressJinvT;
   // setup outputs
   double *Jac0inv_out, *rho0DetJ0w_out, *massD_out;
   const double *rho0_q;
   // reductions
   double *partials;
   unsigned int *ticket;
   double *result; // dt_est (min-folded) or sum
   double h0, h1order, cfl;
   int visc, vort;
   double *force_e;  // update mode (3D): F.1 as E-vector (D1D^3, dim, NE), or nullptr (see below)
   double *erhs_q;   // update mode: F^T v of the SAME state (the velocity block of S), L2 vector, or nullptr (see below)
   int q_swz;        // row form of the update (lgh_qrows.hpp): element -> XCD mapping
   const int *zorder; // row form of the update: workgroup i takes the caller's zone zorder[i] - the library's own zone order (lgh_order.hip:
                      // neighbours close on one XCD whatever order the caller numbers its zones in); nullptr: zone i
   double tiny_grad; // wave-uniform shortcut of the eigen-decomposition below this |sym grad v| (see qpoint_body); < 0: off
};

// Smooth transition between 0 and 1 for x in [-eps, eps] (laghos_solver.cpp:799-805)
__device__ __forceinline__ double smooth_step_01(double x, double eps)
{
   const double y = (x + eps) / (2.0 * eps);
   if (y < 0.0) { return 0.0; }
   if (y > 1.0) { return 1.0; }
   return (3.0 - 2.0 * y) * y * y;
}

// The point-wise body: QUpdateBody (laghos_solver.cpp:1042-1168).  J and dV are
// column-major [c + DIM*d] = d u_c / d xi_d.  Returns this point's dt candidate.
// VISC = false: instantiation without the artificial-viscosity branch (problems that run with visc off: 3D
// Taylor-Green) - no eigen-decomposition in the code, a register budget of its own (lgh_qrows.hpp)
template <int DIM, bool VISC = true>
__device__ __forceinline__ double qpoint_body(const QArgs &a, const int e, const size_t eq,
                                              const double weight, const double *J, const double *dV,
                                              const double e_val, const size_t plane,
                                              const double *J0i, const double rho0DetJ0w, double &ftv, double *sjw)
{
   constexpr int DIM2 = DIM * DIM;
   double Jinv[DIM2], stress[DIM2], sgrad_v[DIM2], stressJiT[DIM2];
   const double gamma = a.gamma[e];
   const double inv_weight = 1. / weight;
   const double detJ = sm::det<DIM>(J);
   sm::inverse<DIM>(J, detJ, Jinv);
   const double R = inv_weight * rho0DetJ0w / detJ;
   const double E = fmax(0.0, e_val);
   const double P = (gamma - 1.0) * R * E;
   const double S = sm::fsqrt(gamma * (gamma - 1.0) * E);
#pragma unroll
   for (int k = 0; k < DIM2; k++) { stress[k] = 0.0; }
#pragma unroll
   for (int d = 0; d < DIM; d++) { stress[d * DIM + d] = -P; }
   double visc_coeff = 0.0;
   // stress : grad v, the integrand of ForceMultTranspose at this point (laghos_assembly.cpp:859-872: sum over (c, gd) of
   // stressJinvT(q, gd, c) d v_c / d xi_gd = (stress Jinv^T) : dV = stress : (dV Jinv) - and the stress is symmetric, so the
   // symmetrised gradient the viscosity branch holds anyway serves).  Formed where that gradient exists (round 5): dV is
   // dead from there on - 18 registers less across the eigen-decomposition; the value differs from the other association
   // of the same sum by round-off only.
   double s_dot_g;
   if (VISC && a.visc)
   {
      sm::matmul<DIM>(dV, Jinv, sgrad_v);
      double vorticity_coeff = 1.0;
      if (a.vort)
      {
         const double grad_norm = sm::fnorm<DIM>(sgrad_v);
         const double div_v = fabs(sm::trace<DIM>(sgrad_v));
         vorticity_coeff = (grad_norm > 0.0) ? div_v / grad_norm : 1.0;
      }
      sm::symmetrize<DIM>(sgrad_v);
      double mu, compr_dir[DIM], Jpi[DIM2], ph_dir[DIM];
      // Zones the flow has not reached carry velocities that are exact zeros or the exponentially small
      // tails the CG iterations spread from the active region (1e-40 and below), yet every one of their
      // points runs the full eigen-decomposition - a third of this kernel's instructions.  When EVERY point
      // of a wavefront has |sym grad v|_max <= tiny_grad (1e-30 by default, in 1/time) the wave takes the
      // result the decomposition has for a tensor with no deviatoric part: mu = tr/3, direction e_x - for an
      // exactly zero tensor bit for bit what min_eigenpair returns (its `triple` branch).  For a non-zero
      // tensor below the threshold only the DIRECTION differs, and it enters through h = h0 |Jpi dir|/|dir|
      // alone: where the mesh has not deformed (Jpi = I to round-off, which is where such points are) h does
      // not depend on it, and the stress changes by visc_coeff * 1e-30 at most.  Wave-uniform: no divergence.
      bool shortcut = false;
      if (DIM == 3 && a.tiny_grad >= 0.0)
      {
         double m = 0.0;
#pragma unroll
         for (int k = 0; k < DIM2; k++) { m = fmax(m, fabs(sgrad_v[k])); }
         shortcut = __all(m <= a.tiny_grad ? 1 : 0) != 0;
      }
      if (shortcut)
      {
         mu = sm::trace<DIM>(sgrad_v) / 3;
#pragma unroll
         for (int d = 0; d < DIM; d++) { compr_dir[d] = (d == 0) ? 1.0 : 0.0; }
      }
      else { sm::min_eigenpair<DIM>(sgrad_v, mu, compr_dir); }
      // ph_dir = (J J0inv) compr_dir as two matrix-vector products (the reference forms the matrix Jpi first, :1117-1119:
      // 18 instead of 36 multiply-adds in 3D, the same value to round-off)
      sm::matvec<DIM>(J0i, compr_dir, Jpi);
      sm::matvec<DIM>(J, Jpi, ph_dir);
      const double ph_dir_nl2 = sm::norml2<DIM>(ph_dir);
      const double compr_dir_nl2 = sm::norml2<DIM>(compr_dir);
      const double H = a.h0 * ph_dir_nl2 / compr_dir_nl2;
      visc_coeff = 2.0 * R * H * H * fabs(mu);
      const double eps = 1e-12;
      visc_coeff += 0.5 * R * H * S * vorticity_coeff * (1.0 - smooth_step_01(mu - 2.0 * eps, eps));
#pragma unroll
      for (int k = 0; k < DIM2; k++) { stress[k] += visc_coeff * sgrad_v[k]; }
      s_dot_g = 0.0;
#pragma unroll
      for (int k = 0; k < DIM2; k++) { s_dot_g += stress[k] * sgrad_v[k]; }
   }
   else
   {
      // no viscosity: stress = -P I, stress : grad v = -P div v, div v = trace(dV Jinv)
      double divv = 0.0;
#pragma unroll
      for (int i = 0; i < DIM; i++)
#pragma unroll
         for (int d = 0; d < DIM; d++) { divv += dV[i + DIM * d] * Jinv[d + DIM * i]; }
      s_dot_g = -P * divv;
   }
   const double sv = sm::min_singular<DIM>(J);
   const double h_min = sv / a.h1order;
   const double ih_min = 1. / h_min;
   const double irho_ih_min_sq = ih_min * ih_min / R;
   const double idt = S * ih_min + 2.5 * visc_coeff * irho_ih_min_sq;
   double dt_cand = INFINITY;
   if (detJ < 0.0) { dt_cand = 0.0; }
   else if (idt > 0.0) { dt_cand = a.cfl / idt; }
   sm::matmul_abt<DIM>(stress, Jinv, stressJiT);
   const double wd = weight * detJ;
   ftv = s_dot_g * wd;
#pragma unroll
   for (int vd = 0; vd < DIM; vd++)
#pragma unroll
      for (int gd = 0; gd < DIM; gd++)
      {
         const double sv_ = stressJiT[vd + gd * DIM] * wd;
         sjw[gd + vd * DIM] = sv_;
         if (a.stressJinvT) { a.stressJinvT[eq + plane * (gd + vd * DIM)] = sv_; } // (nullptr: the stress stays in registers, lgh_qupdate_store_stress)
      }
   return dt_cand;
}

// One workgroup = NEB elements, one thread per quadrature point.
//   3D: blockDim = Q^3 (NEB = 1); 2D: blockDim = Q^2 * NEB.
// NFMAX = H1 fields (components of x and v) interpolated per LDS pass; 6 = one
// pass for the update (3 barriers in total), smaller when LDS is short.
// All global reads of an element (x, v, e gathers and the point-wise
// Jac0inv / rho0DetJ0w) are issued before the first barrier, so a workgroup pays
// one global-memory latency, not five; the L2 (energy) interpolation shares the
// barrier intervals of the H1 stages.
// PPT = 2 (3D update mode, all fields in one LDS pass): two quadrature points per thread, tz and tz + Q/2 - the
// 1000-thread workgroups of Q5Q4 are capped at 128 registers per thread (16 wavefronts on one CU) and spill 41; with
// 500 threads the cap is 256.  The second point's z stage and body run after the first one's, on the same registers.
template <int DIM, int D, int Q, int L, int NEB, int NFMAX, int MODE, int PPT = 1>
__global__ void __launch_bounds__((DIM == 3 ? Q * Q * Q : Q * Q) * NEB / PPT)
qpoint_kernel(const QArgs a)
{
   constexpr int ND = (DIM == 3) ? D * D * D : D * D;
   constexpr int NQ = (DIM == 3) ? Q * Q * Q : Q * Q;
   constexpr int NL = (DIM == 3) ? L * L * L : L * L;
   constexpr int NTE = NQ / PPT; // threads per element
   static_assert(PPT == 1 || (PPT == 2 && DIM == 3 && MODE == QMODE_UPDATE && Q % 2 == 0), "two points per thread: 3D update only");
   constexpr bool HOIST = (NQ * NEB <= 256) && PPT == 1; // per-thread-constant table rows kept in registers (below)
   constexpr bool NEED_X = (MODE == QMODE_UPDATE || MODE == QMODE_SETUP);
   constexpr bool NEED_V = (MODE == QMODE_UPDATE || MODE == QMODE_KE);
   constexpr bool NEED_E = (MODE != QMODE_KE);
   constexpr int NFIELD = (NEED_X ? DIM : 0) + (NEED_V ? DIM : 0);
   constexpr int NF = (NFIELD == 0) ? 1 : (NFMAX < NFIELD ? NFMAX : NFIELD);
   static_assert(PPT == 1 || NF == NFIELD, "two points per thread: all fields in one LDS pass");
   // LDS per element
   constexpr int SU = NF * ND;
   constexpr int SXs = (DIM == 3) ? 2 * NF * D * D * Q : 2 * NF * D * Q; // B,G applied in x
   constexpr int SYs = (DIM == 3) ? 3 * NF * D * Q * Q : 0;              // BB,GB,BG (3D only)
   constexpr int SEs = NL + ((DIM == 3) ? (L * L * Q + L * Q * Q) : (L * Q));
   constexpr int PER0 = SU + SXs + SYs + SEs + 1;
   // F.1 fused into the update (3D): the point values of CPR components x 3 reference directions and their
   // z-contracted arrays live in this element's interpolation buffers; all 3 components per round where that
   // fits, one per round otherwise (the slice is enlarged if even that does not fit: Q5Q4)
   constexpr int FTN = NQ + ((DIM == 3) ? L * Q * Q : 0); // point values + z-contracted array of F^T v
   constexpr int FNEED3 = 9 * NQ + 9 * D * Q * Q + FTN, FNEED1 = 3 * NQ + 3 * D * Q * Q + FTN;
   constexpr int AVAIL = SU + SXs + SYs + ((DIM == 3) ? L * L * L : L * L); // the slice up to the array the y stage of F^T v writes
   constexpr int CPR = (DIM == 3 && MODE == QMODE_UPDATE) ? ((AVAIL >= FNEED3) ? 3 : 1) : 0;
   // (one component per round where three do not fit; the slice is enlarged if even that does not: Q5Q4)
   constexpr int GROW = (CPR == 1 && AVAIL < FNEED1) ? FNEED1 - AVAIL : 0;
   constexpr int PER = PER0 + GROW;
   __shared__ double smem[NEB * PER];
   __shared__ double sB[Q * D], sG[Q * D], sBl[Q * L];
   __shared__ double red[16];

   const int tid = threadIdx.x;
   const int lt = tid % NTE, eb = tid / NTE;
   const int tx = lt % Q, ty = (lt / Q) % Q, tz = lt / (Q * Q);
   const int e = blockIdx.x * NEB + eb;
   const bool active = (e < a.NE);
   const int ec = active ? e : a.NE - 1; // clamp: inactive threads compute on a valid element
   double *sU = smem + eb * PER;
   double *sX = sU + SU;
   double *sY = sX + SXs;
   double *sE = sY + SYs + GROW; // (GROW: room for the force contractions of the update mode, see above)
   double *sE1 = sE + NL;                            // 3D [lz][ly][qx]; 2D [ly][qx]
   double *sE2 = sE1 + ((DIM == 3) ? L * L * Q : 0); // 3D [lz][qy][qx]

   const size_t eq = (size_t)ec * NQ + lt;

   // ---- issue every global read of this element up front
   for (int i = tid; i < Q * D; i += NTE * NEB) { sB[i] = a.B[i]; sG[i] = a.G[i]; }
   for (int i = tid; i < Q * L; i += NTE * NEB) { sBl[i] = a.Bl[i]; }
   auto gather_fields = [&](const int f0) {
      for (int i = lt; i < NF * ND; i += NTE)
      {
         const int fl = i / ND, d = i - fl * ND;
         const int f = f0 + fl;
         double u = 0.0;
         if (f < NFIELD)
         {
            const bool isx = NEED_X && (f < DIM);
            const int comp = isx ? f : (f - (NEED_X ? DIM : 0));
            const double *src = isx ? a.x : a.v;
            u = src[(size_t)comp * a.N + a.map[(size_t)ec * ND + d]];
         }
         sU[i] = u;
      }
   };
   if (NFIELD > 0) { gather_fields(0); }
   if (NEED_E)
   {
      for (int i = lt; i < NL; i += NTE) { sE[i] = a.e[(size_t)ec * NL + i]; }
   }
   double J0i[DIM * DIM];
   double rdw = 0.0;
   double J0i_b[PPT > 1 ? DIM * DIM : 1]; // (PPT = 2: the point lt + NTE)
   double rdw_b = 0.0;
   if (MODE == QMODE_UPDATE)
   {
#pragma unroll
      for (int k = 0; k < DIM * DIM; k++) { J0i[k] = a.Jac0inv_soa[eq + (size_t)a.NE * NQ * k]; } // plane-major copy: coalesced
      rdw = a.rho0DetJ0w_in[eq];
      if constexpr (PPT > 1)
      {
#pragma unroll
         for (int k = 0; k < DIM * DIM; k++) { J0i_b[k] = a.Jac0inv_soa[eq + NTE + (size_t)a.NE * NQ * k]; }
         rdw_b = a.rho0DetJ0w_in[eq + NTE];
      }
   }
   else if (MODE == QMODE_IE || MODE == QMODE_KE) { rdw = a.rho0DetJ0w_in[eq]; }

   double grad[NFIELD > 0 ? NFIELD * DIM : 1]; // [field][d]
   double val[NFIELD > 0 ? NFIELD : 1];
   (void)grad;
   (void)val;
   double e_val = 0.0;
   // 3D z stage of the point (tx, ty, tzp): values and gradients of the fields of the pass, and e
   auto zstage3 = [&](const int f0, const bool first_pass, const int tzp, double *grad_, double *val_, double &ev_) __attribute__((always_inline)) {
#pragma unroll
      for (int fl = 0; fl < NF; fl++)
      {
         if (NFIELD > 0 && f0 + fl < NFIELD)
         {
            double vv = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int dz = 0; dz < D; dz++)
            {
               const int j = tx + Q * (ty + Q * (dz + D * fl));
               const double b = sB[tzp + Q * dz], g = sG[tzp + Q * dz];
               const double bb = sY[j];
               vv += b * bb;
               d0 += b * sY[j + NF * D * Q * Q];
               d1 += b * sY[j + 2 * NF * D * Q * Q];
               d2 += g * bb;
            }
            val_[f0 + fl] = vv;
            grad_[(f0 + fl) * DIM + 0] = d0;
            grad_[(f0 + fl) * DIM + 1] = d1;
            grad_[(f0 + fl) * DIM + (DIM - 1)] = d2;
         }
      }
      if (NEED_E && first_pass)
      {
#pragma unroll
         for (int lz = 0; lz < L; lz++) { ev_ += sBl[tzp + Q * lz] * sE2[tx + Q * (ty + Q * lz)]; }
      }
   };

   for (int f0 = 0; f0 < (NFIELD > 0 ? NFIELD : 1); f0 += NF)
   {
      const bool first_pass = (f0 == 0);
      if (!first_pass)
      {
         __syncthreads(); // previous pass done with sU/sX/sY
         gather_fields(f0);
      }
      __syncthreads();
      if (DIM == 3)
      {
         // x stage: [k][fl][dz][dy][qx]
         if (NFIELD > 0)
         {
            // qx = i % Q is the same for every item of a thread (the stride Q^3 is a multiple of Q): its table rows
            // are read once, not per item - LDS issue, shared by the four SIMDs, is scarcer than FMA issue
            // (only where registers are to spare: the 512- and 1000-thread workgroups of Q4Q3 / Q5Q4 are capped)
            double bxr[HOIST ? D : 1], gxr[HOIST ? D : 1];
            if constexpr (HOIST)
            {
#pragma unroll
               for (int dx = 0; dx < D; dx++) { bxr[dx] = sB[tx + Q * dx]; gxr[dx] = sG[tx + Q * dx]; }
            }
            auto xitem = [&](const int i, const int qx, const int dy, const int dz, const int fl) {
               double u = 0.0, w = 0.0;
#pragma unroll
               for (int dx = 0; dx < D; dx++)
               {
                  const double s = sU[dx + D * (dy + D * dz) + ND * fl];
                  u += (HOIST ? bxr[HOIST ? dx : 0] : sB[qx + Q * dx]) * s;
                  w += (HOIST ? gxr[HOIST ? dx : 0] : sG[qx + Q * dx]) * s;
               }
               sX[i] = u;
               sX[i + NF * D * D * Q] = w;
            };
            if constexpr (HOIST)
            {
               // the items of a thread are i = tx + Q * r, r = ty + Q * tz + k * Q^2: no division by Q per item
               for (int r = ty + Q * tz; r < NF * D * D; r += Q * Q) { xitem(tx + Q * r, tx, r % D, (r / D) % D, r / (D * D)); }
            }
            else
            {
               for (int i = lt; i < NF * D * D * Q; i += NTE) { xitem(i, i % Q, (i / Q) % D, (i / (Q * D)) % D, i / (Q * D * D)); }
            }
         }
         if (NEED_E && first_pass)
         {
            for (int i = lt; i < L * L * Q; i += NTE)
            {
               const int qx = i % Q, ly = (i / Q) % L, lz = i / (Q * L);
               double u = 0.0;
#pragma unroll
               for (int lx = 0; lx < L; lx++) { u += sBl[qx + Q * lx] * sE[lx + L * (ly + L * lz)]; }
               sE1[i] = u;
            }
         }
         __syncthreads();
         // y stage: BB, GB, BG [k][fl][dz][qy][qx]
         if (NFIELD > 0)
         {
            double byr[HOIST ? D : 1], gyr[HOIST ? D : 1]; // (qx, qy) = (tx, ty) for every item of a thread: the stride is a multiple of Q^2
            if constexpr (HOIST)
            {
#pragma unroll
               for (int dy = 0; dy < D; dy++) { byr[dy] = sB[ty + Q * dy]; gyr[dy] = sG[ty + Q * dy]; }
            }
            auto yitem = [&](const int i, const int qx, const int qy, const int dz, const int fl) {
               double bb = 0.0, gb = 0.0, bg = 0.0;
#pragma unroll
               for (int dy = 0; dy < D; dy++)
               {
                  const int j = qx + Q * (dy + D * (dz + D * fl));
                  const double vb = sX[j], vg = sX[j + NF * D * D * Q];
                  const double tb = HOIST ? byr[HOIST ? dy : 0] : sB[qy + Q * dy], tg = HOIST ? gyr[HOIST ? dy : 0] : sG[qy + Q * dy];
                  bb += tb * vb;
                  gb += tb * vg;
                  bg += tg * vb;
               }
               sY[i] = bb;
               sY[i + NF * D * Q * Q] = gb;
               sY[i + 2 * NF * D * Q * Q] = bg;
            };
            if constexpr (HOIST)
            {
               // i = tx + Q * ty + Q^2 * m, m = dz + D * fl = tz + k * Q
               for (int m = tz; m < NF * D; m += Q) { yitem(tx + Q * (ty + Q * m), tx, ty, m % D, m / D); }
            }
            else
            {
               for (int i = lt; i < NF * D * Q * Q; i += NTE) { yitem(i, i % Q, (i / Q) % Q, (i / (Q * Q)) % D, i / (Q * Q * D)); }
            }
         }
         if (NEED_E && first_pass)
         {
            for (int i = lt; i < L * Q * Q; i += NTE)
            {
               const int qx = i % Q, qy = (i / Q) % Q, lz = i / (Q * Q);
               double u = 0.0;
#pragma unroll
               for (int ly = 0; ly < L; ly++) { u += sBl[qy + Q * ly] * sE1[qx + Q * (ly + L * lz)]; }
               sE2[i] = u;
            }
         }
         __syncthreads();
         // z stage: this thread's point (PPT = 2: both points after the loop, one after the other - update section)
         if constexpr (PPT == 1) { zstage3(f0, first_pass, tz, grad, val, e_val); }
      }
      else
      {
         if (NFIELD > 0)
         {
            for (int i = lt; i < NF * D * Q; i += NTE)
            {
               const int qx = i % Q, dy = (i / Q) % D, fl = i / (Q * D);
               double u = 0.0, w = 0.0;
#pragma unroll
               for (int dx = 0; dx < D; dx++)
               {
                  const double s = sU[dx + D * dy + ND * fl];
                  u += sB[qx + Q * dx] * s;
                  w += sG[qx + Q * dx] * s;
               }
               sX[i] = u;
               sX[i + NF * D * Q] = w;
            }
         }
         if (NEED_E && first_pass)
         {
            for (int i = lt; i < L * Q; i += NTE)
            {
               const int qx = i % Q, ly = i / Q;
               double u = 0.0;
#pragma unroll
               for (int lx = 0; lx < L; lx++) { u += sBl[qx + Q * lx] * sE[lx + L * ly]; }
               sE1[i] = u;
            }
         }
         __syncthreads();
#pragma unroll
         for (int fl = 0; fl < NF; fl++)
         {
            if (NFIELD > 0 && f0 + fl < NFIELD)
            {
               double vv = 0.0, d0 = 0.0, d1 = 0.0;
#pragma unroll
               for (int dy = 0; dy < D; dy++)
               {
                  const int j = tx + Q * (dy + D * fl);
                  const double vb = sX[j], vg = sX[j + NF * D * Q];
                  vv += sB[ty + Q * dy] * vb;
                  d0 += sB[ty + Q * dy] * vg;
                  d1 += sG[ty + Q * dy] * vb;
               }
               val[f0 + fl] = vv;
               grad[(f0 + fl) * DIM + 0] = d0;
               grad[(f0 + fl) * DIM + 1] = d1;
            }
         }
         if (NEED_E && first_pass)
         {
#pragma unroll
            for (int ly = 0; ly < L; ly++) { e_val += sBl[ty + Q * ly] * sE1[tx + Q * ly]; }
         }
      }
   }

   const size_t plane = (size_t)a.NE * NQ;
   const double weight = a.W[lt];

   if (MODE == QMODE_UPDATE)
   {
      // grad[(c)*DIM + d] for x then v  ->  column-major J[c + DIM*d]
      double J[DIM * DIM], dV[DIM * DIM];
#pragma unroll
      for (int c = 0; c < DIM; c++)
#pragma unroll
         for (int d = 0; d < DIM; d++)
         {
            J[c + DIM * d] = grad[c * DIM + d];
            dV[c + DIM * d] = grad[(DIM + c) * DIM + d];
         }
      double cand = INFINITY, ftv = 0.0, sjw[DIM * DIM];
      double ftv_b = 0.0, sjw_b[PPT > 1 ? DIM * DIM : 1]; // (PPT = 2: the point lt + NTE)
#pragma unroll
      for (int k = 0; k < DIM * DIM; k++) { sjw[k] = 0.0; }
      if constexpr (PPT == 1)
      {
         if (active) { cand = qpoint_body<DIM>(a, e, eq, weight, J, dV, e_val, plane, J0i, rdw, ftv, sjw); }
      }
      else
      {
#pragma unroll
         for (int k = 0; k < DIM * DIM; k++) { sjw_b[k] = 0.0; }
#pragma unroll
         for (int p = 0; p < PPT; p++)
         {
            double g2[NFIELD * DIM], v2[NFIELD], ev = 0.0;
            zstage3(0, true, tz + p * (Q / PPT), g2, v2, ev);
#pragma unroll
            for (int c = 0; c < DIM; c++)
#pragma unroll
               for (int d = 0; d < DIM; d++)
               {
                  J[c + DIM * d] = g2[c * DIM + d];
                  dV[c + DIM * d] = g2[(DIM + c) * DIM + d];
               }
            if (active)
            {
               const double c_p = (p == 0) ? qpoint_body<DIM>(a, e, eq, weight, J, dV, ev, plane, J0i, rdw, ftv, sjw)
                                           : qpoint_body<DIM>(a, e, eq + NTE, a.W[lt + NTE], J, dV, ev, plane, J0i_b, rdw_b, ftv_b, sjw_b);
               cand = fmin(cand, c_p);
            }
            __builtin_amdgcn_sched_barrier(0); // (one point after the other: interleaved, the two bodies need twice the registers)
         }
      }
      // ---- the two force products of this state, from the values still in registers -------------------
      // F^T v (ForcePAOperator::MultTranspose, laghos_assembly.cpp:859-921: the point integrand above tested
      // with the L2 basis) is SolveEnergy's right-hand side for the velocity block of THIS state, and F.1
      // (ForcePA->Mult(one), :296-514 with x = 1 - the Bernstein functions sum to one, so the interpolated
      // `one` is 1 at every point: Y(d, c) = sum_q sum_gd stressJinvT(q, gd, c) d_gd phi_d(q)) is
      // SolveVelocity's.  Formed here they cost a few LDS contractions (transposed sum factorisation z -> y -> x)
      // instead of two more passes over the 9 stressJinvT planes (2 x 510 MB at C2).  The stages of both share
      // their barriers.  lgh_solve_velocity / lgh_solve_energy use the results when they are called for the
      // state of the last update (and, for F^T v, with its own velocity).
      const bool do_f = (DIM == 3) && (CPR > 0) && (a.force_e != nullptr);
      const bool do_t = (a.erhs_q != nullptr);
      if (do_f || do_t)
      {
         const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;
         constexpr int CP = (CPR > 0) ? CPR : 1;
         constexpr bool F3 = (DIM == 3) && (CPR > 0);
         constexpr int OFF_A = F3 ? CP * 3 * NQ : 0;                   // sF: [cc][gd][q]
         constexpr int OFF_S = OFF_A + (F3 ? CP * 3 * D * Q * Q : 0);  // sA: [cc][gd][dz][qy][qx]
         constexpr int OFF_T = OFF_S + NQ;                                     // sS: point values of F^T v
         static_assert(OFF_T + ((DIM == 3) ? L * Q * Q : 0) <= PER - 1 - SEs + NL, "LDS: no room for the force contractions");
         double *sF = sU, *sA = sU + OFF_A, *sW = sU; // sW [cc][which][dz][dy][qx] over sF (dead by then)
         double *sS = sU + OFF_S, *sT = sU + OFF_T;   // sT [lz][qy][qx] (3D)
#pragma unroll 1
         for (int c0 = 0; c0 < (do_f ? 3 : 1); c0 += CP)
         {
            const bool t_now = do_t && (c0 == 0);
            __syncthreads(); // the buffers are free (interpolation / previous round done)
            if constexpr (F3)
            {
               if (do_f)
               {
#pragma unroll
                  for (int cc = 0; cc < CP; cc++)
                  {
#pragma unroll
                     for (int gd = 0; gd < 3; gd++)
                     {
                        sF[lt + NQ * (gd + 3 * cc)] = sjw[gd + 3 * (c0 + cc)];
                        if constexpr (PPT > 1) { sF[lt + NTE + NQ * (gd + 3 * cc)] = sjw_b[gd + 3 * (c0 + cc)]; }
                     }
                  }
               }
            }
            if (t_now)
            {
               sS[lt] = active ? ftv : 0.0;
               if constexpr (PPT > 1) { sS[lt + NTE] = active ? ftv_b : 0.0; }
            }
            __syncthreads();
            if constexpr (DIM == 3)
            {
               // ---- z
               if (do_f)
               {
                  auto fzitem = [&](const int i, const int qx, const int qy, const int dz, const int k) { // k = gd + 3 cc
                     const double *tab = ((k % 3) == 2) ? sG : sB;
                     double u = 0.0;
#pragma unroll
                     for (int qz = 0; qz < Q; qz++) { u += tab[qz + Q * dz] * sF[qx + Q * (qy + Q * qz) + NQ * k]; }
                     sA[i] = u;
                  };
                  if constexpr (HOIST)
                  {
                     for (int m = tz; m < CP * 3 * D; m += Q) { fzitem(tx + Q * (ty + Q * m), tx, ty, m % D, m / D); } // as the y stage above
                  }
                  else
                  {
                     for (int i = lt; i < CP * 3 * D * Q * Q; i += NTE) { fzitem(i, i % Q, (i / Q) % Q, (i / (Q * Q)) % D, i / (Q * Q * D)); }
                  }
               }
               if (t_now)
               {
                  for (int i = lt; i < L * Q * Q; i += NTE)
                  {
                     const int qx = i % Q, qy = (i / Q) % Q, lz = i / (Q * Q);
                     double u = 0.0;
#pragma unroll
                     for (int qz = 0; qz < Q; qz++) { u += sBl[qz + Q * lz] * sS[qx + Q * (qy + Q * qz)]; }
                     sT[i] = u;
                  }
               }
               __syncthreads();
               // ---- y
               if (do_f)
               {
                  // (dy is the same for every item of a thread where the stride is a multiple of Q*D: table columns in registers)
                  constexpr bool YINV = HOIST && (NTE % (Q * D) == 0);
                  double byf[YINV ? Q : 1], gyf[YINV ? Q : 1];
                  if constexpr (YINV)
                  {
                     const int dy0 = (lt / Q) % D;
#pragma unroll
                     for (int qy = 0; qy < Q; qy++) { byf[qy] = sB[qy + Q * dy0]; gyf[qy] = sG[qy + Q * dy0]; }
                  }
                  auto fyitem = [&](const int i, const int qx, const int dy, const int dz, const int wh, const int cc) {
                     const double *a0 = sA + D * Q * Q * (0 + 3 * cc) + Q * Q * dz + qx;
                     const double *a1 = sA + D * Q * Q * (1 + 3 * cc) + Q * Q * dz + qx;
                     const double *a2 = sA + D * Q * Q * (2 + 3 * cc) + Q * Q * dz + qx;
                     double u = 0.0;
                     if (wh == 0)
                     {
#pragma unroll
                        for (int qy = 0; qy < Q; qy++) { u += (YINV ? byf[qy] : sB[qy + Q * dy]) * a0[Q * qy]; } // gd 0: G in x below
                     }
                     else
                     {
#pragma unroll
                        for (int qy = 0; qy < Q; qy++)
                        {
                           u += (YINV ? gyf[qy] : sG[qy + Q * dy]) * a1[Q * qy] + (YINV ? byf[qy] : sB[qy + Q * dy]) * a2[Q * qy];
                        }
                     }
                     sW[i] = u;
                  };
                  if constexpr (HOIST)
                  {
                     for (int r = ty + Q * tz; r < CP * 2 * D * D; r += Q * Q) // as the x stage above
                     {
                        fyitem(tx + Q * r, tx, r % D, (r / D) % D, (r / (D * D)) % 2, r / (D * D * 2));
                     }
                  }
                  else
                  {
                     for (int i = lt; i < CP * 2 * D * D * Q; i += NTE)
                     {
                        fyitem(i, i % Q, (i / Q) % D, (i / (Q * D)) % D, (i / (Q * D * D)) % 2, i / (Q * D * D * 2));
                     }
                  }
               }
               if (t_now)
               {
                  for (int i = lt; i < L * L * Q; i += NTE)
                  {
                     const int qx = i % Q, ly = (i / Q) % L, lz = i / (Q * L);
                     double u = 0.0;
#pragma unroll
                     for (int qy = 0; qy < Q; qy++) { u += sBl[qy + Q * ly] * sT[qx + Q * (qy + Q * lz)]; }
                     sE1[i] = u;
                  }
               }
               __syncthreads();
               // ---- x
               if (do_f)
               {
                  constexpr bool XINV = HOIST && (NTE % D == 0); // dx = i % D is then the same for every item of a thread
                  double bxf[XINV ? Q : 1], gxf[XINV ? Q : 1];
                  if constexpr (XINV)
                  {
#pragma unroll
                     for (int qx = 0; qx < Q; qx++) { bxf[qx] = sB[qx + Q * (lt % D)]; gxf[qx] = sG[qx + Q * (lt % D)]; }
                  }
                  for (int i = lt; i < CP * ND; i += NTE)
                  {
                     const int dx = i % D, dy = (i / D) % D, dz = (i / (D * D)) % D, cc = i / ND;
                     const double *wg = sW + Q * (dy + D * (dz + D * (0 + 2 * cc)));
                     const double *wb = sW + Q * (dy + D * (dz + D * (1 + 2 * cc)));
                     double r = 0.0;
#pragma unroll
                     for (int qx = 0; qx < Q; qx++) { r += (XINV ? gxf[qx] : sG[qx + Q * dx]) * wg[qx] + (XINV ? bxf[qx] : sB[qx + Q * dx]) * wb[qx]; }
                     if (fabs(r) < eps2) { r = 0.0; } // laghos_assembly.cpp:495-512
                     if (active) { a.force_e[dx + D * (dy + D * dz) + (size_t)ND * ((c0 + cc) + 3 * (size_t)e)] = r; }
                  }
               }
               if (t_now)
               {
                  for (int i = lt; i < NL; i += NTE)
                  {
                     const int lx = i % L, ly = (i / L) % L, lz = i / (L * L);
                     double u = 0.0;
#pragma unroll
                     for (int qx = 0; qx < Q; qx++) { u += sBl[qx + Q * lx] * sE1[qx + Q * (ly + L * lz)]; }
                     if (active) { a.erhs_q[(size_t)e * NL + i] = u; }
                  }
               }
            }
            else if (t_now)
            {
               for (int i = lt; i < L * Q; i += NTE)
               {
                  const int qx = i % Q, ly = i / Q;
                  double u = 0.0;
#pragma unroll
                  for (int qy = 0; qy < Q; qy++) { u += sBl[qy + Q * ly] * sS[qx + Q * qy]; }
                  sE1[i] = u;
               }
               __syncthreads();
               for (int i = lt; i < NL; i += NTE)
               {
                  const int lx = i % L, ly = i / L;
                  double u = 0.0;
#pragma unroll
                  for (int qx = 0; qx < Q; qx++) { u += sBl[qx + Q * lx] * sE1[qx + Q * ly]; }
                  if (active) { a.erhs_q[(size_t)e * NL + i] = u; }
               }
            }
         }
      }
      const double bmin = block_min(cand, red);
      double total;
      if (grid_min_last_block(bmin, a.partials, a.ticket, red, total))
      {
         if (tid == 0) { *a.result = fmin(*a.result, total); } // q_dt_est = qdata.dt_est; Min() (:1374, :1406)
      }
   }
   else if (MODE == QMODE_SETUP)
   {
      // Rho0DetJ0Vol: Jac0inv with the reference's index convention (:1209-1251)
      double J[DIM * DIM];
#pragma unroll
      for (int c = 0; c < DIM; c++)
#pragma unroll
         for (int d = 0; d < DIM; d++) { J[c + DIM * d] = grad[c * DIM + d]; }
      const double det = sm::det<DIM>(J);
      double part = 0.0;
      if (active)
      {
         double *Ji = a.Jac0inv_out + eq * DIM * DIM;
         const double r = 1.0 / det;
         if (DIM == 2)
         {
            Ji[0] = J[3] * r;
            Ji[1] = -J[1] * r;
            Ji[2] = -J[2] * r;
            Ji[3] = J[0] * r;
         }
         else
         {
            // J11..J33 named as in the reference: Jab = J(q,a-1,b-1,e)
            const double J11 = J[0], J12 = J[3], J13 = J[6];
            const double J21 = J[1], J22 = J[4], J23 = J[7];
            const double J31 = J[2], J32 = J[5], J33 = J[8];
            Ji[0] = r * ((J22 * J33) - (J23 * J32));
            Ji[1] = r * ((J32 * J13) - (J33 * J12));
            Ji[2] = r * ((J12 * J23) - (J13 * J22));
            Ji[3] = r * ((J23 * J31) - (J21 * J33));
            Ji[4] = r * ((J33 * J11) - (J31 * J13));
            Ji[5] = r * ((J13 * J21) - (J11 * J23));
            Ji[6] = r * ((J21 * J32) - (J22 * J31));
            Ji[7] = r * ((J31 * J12) - (J32 * J11));
            Ji[8] = r * ((J11 * J22) - (J12 * J21));
         }
         for (int k = 0; k < DIM * DIM; k++) { a.Jac0inv_soa_out[eq + plane * k] = Ji[k]; }
         a.rho0DetJ0w_out[eq] = weight * e_val * det; // e_val = rho0 grid function here
         a.massD_out[eq] = weight * det * a.rho0_q[eq];
         part = weight * det;
      }
      const double bsum = block_sum(part, red);
      double total;
      if (grid_sum_last_block(bsum, a.partials, a.ticket, red, total))
      {
         if (tid == 0) { *a.result = total; }
      }
   }
   else
   {
      // ComputeVolumeIntegral (laghos_solver.cpp:565-639): sum_q f(q) * rho0DetJ0w
      double part = 0.0;
      if (active)
      {
         double f;
         if (MODE == QMODE_IE) { f = e_val; }
         else
         {
            f = 0.0;
#pragma unroll
            for (int c = 0; c < DIM; c++) { f += val[c] * val[c]; }
         }
         part = f * rdw;
      }
      const double bsum = block_sum(part, red);
      double total;
      if (grid_sum_last_block(bsum, a.partials, a.ticket, red, total))
      {
         if (tid == 0) { *a.result = total; }
      }
   }
}

static int unknown_kernel(int id)
{
   set_error("Unknown kernel 0x%x", id);
   return LGH_ERR_UNSUPPORTED;
}

template <int MODE> static int launch_q(lgh_ctx *c, const QArgs &a)
{
#define LGH_Q3(D_, Q_, L_, NF_)                                                                      \
   hipLaunchKernelGGL((qpoint_kernel<3, D_, Q_, L_, 1, NF_, MODE>), dim3(c->NE), dim3(Q_ * Q_ * Q_), \
                      0, c->stream, a);                                                              \
   break
#define LGH_Q2(D_, Q_, L_)                                                                           \
   {                                                                                                 \
      constexpr int NEB_ = (256 / (Q_ * Q_)) > 0 ? (256 / (Q_ * Q_)) : 1;                            \
      hipLaunchKernelGGL((qpoint_kernel<2, D_, Q_, L_, NEB_, 4, MODE>), dim3(ceil_div(c->NE, NEB_)), \
                         dim3(Q_ * Q_ * NEB_), 0, c->stream, a);                                     \
   }                                                                                                 \
   break
   switch (c->kid)
   {
      // ids follow the reference table (laghos_solver.cpp:1387-1396) joined with D1D
      case 0x222: LGH_Q2(2, 2, 1);
      case 0x234: LGH_Q2(3, 4, 2);
      case 0x246: LGH_Q2(4, 6, 3);
      case 0x258: LGH_Q2(5, 8, 4);
      case 0x26A: LGH_Q2(6, 10, 5);
      case 0x322: LGH_Q3(2, 2, 1, 6);
      case 0x334: LGH_Q3(3, 4, 2, 6);
      case 0x346: LGH_Q3(4, 6, 3, 6);
      // all six H1 fields in one LDS pass at every order: only one workgroup of 512 / 1000 threads fits a CU at Q4Q3 /
      // Q5Q4 anyway, so its 76 / 140 KB of LDS cost nothing, and one pass needs fewer registers than several (the
      // gradients do not have to survive across passes): 234 -> 158 VGPRs at Q4Q3, 70 -> 41 spilled registers under
      // the 128-register cap of the 1000-thread workgroup at Q5Q4, where the update takes 10.4 instead of 17 ms
      case 0x358: LGH_Q3(5, 8, 4, 6);
      case 0x36A: // extension: not in the reference table
         if constexpr (MODE == QMODE_UPDATE)
         {
            // two points per thread: 500-thread workgroups under a cap of 256 registers instead of 1000 under 128 (41 spilled)
            const char *penv = getenv("LGH_Q_PPT"); // A/B: 1 = one point per thread (read per launch: tests switch it)
            if (!(penv && penv[0] == '1'))
            {
               hipLaunchKernelGGL((qpoint_kernel<3, 6, 10, 5, 1, 6, MODE, 2>), dim3(c->NE), dim3(500), 0, c->stream, a);
               break;
            }
         }
         LGH_Q3(6, 10, 5, 6);
      default: return unknown_kernel(c->kid);
   }
#undef LGH_Q3
#undef LGH_Q2
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

static QArgs q_base(lgh_ctx *c)
{
   QArgs a;
   memset(&a, 0, sizeof(a));
   a.NE = c->NE;
   a.N = c->N;
   a.B = c->B;
   a.G = c->G;
   a.Bl = c->Bl;
   a.W = c->W;
   a.map = c->h1map;
   a.gamma = c->gamma;
   a.rho0DetJ0w_in = c->rho0DetJ0w;
   a.Jac0inv_in = c->Jac0inv;
   a.Jac0inv_soa = c->Jac0inv_soa;
   a.stressJinvT = c->stressJinvT;
   a.partials = c->partials + 2 * (size_t)c->part_stride;
   a.ticket = c->tickets + 2 * kTicketSlot;
   a.h0 = c->h0;
   a.h1order = c->h1order;
   a.cfl = c->cfl;
   a.visc = c->visc;
   a.vort = c->vort;
   a.tiny_grad = c->q_tiny_grad;
   {
      const char *senv = getenv("LGH_Q_SWZ"); // A/B: -1 contiguous eighths, 0 none, n: runs of 2^n elements
      a.q_swz = senv ? atoi(senv) : -1;
   }
   a.erhs_q = c->fused_forces_off ? nullptr : c->erhs_q;
   a.force_e = (c->dim == 3 && !c->fused_forces_off) ? c->force_e_q : nullptr;
   {
      const MeshOrder *o = mesh_order(c);
      a.zorder = o ? o->zorder_d : nullptr;
   }
   return a;
}

} // namespace lgh
