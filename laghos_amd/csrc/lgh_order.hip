// lgh_order.hip — the order in which the library walks the zones and numbers the nodes of ITS OWN vectors.
//
// The reference hands its operators whatever numbering the mesh library has: `H1.GetElementRestriction(LEXICOGRAPHIC)`
// of an MFEM space (/root/reference/laghos_assembly.cpp:133-134) numbers vertex dofs first, then edge, face and interior
// dofs, and the zones come in the order `UniformRefinement` leaves them (/root/reference/laghos.cpp:391: the children of a
// zone consecutive - a tree order).  Only the element-local dof order is fixed (lexicographic: the interface).  The fast
// paths of the velocity solve need more than that: zones that follow each other along their local x-axis (sets of five
// as x-chains: merged E-vector), node numbers that run along those rows (row loads of the slab K1, whole cache lines in
// K2), neighbouring zones close in time on one XCD.  Rounds 1-5 FOUND that structure in the caller's map when it was
// there - and it was there only because every bench mesh came out of this repository's own generator (round-5 verdict,
// item 1; profiles/r6_numbering_before.txt: 0.58 x under an MFEM-like numbering, 0.23 x under a random one).
//
// Now the library orders things itself, from the map alone:
//   1. zone f is the +a neighbour of zone e (a = x, y, z of the element-local frame) iff the D x D nodes of e's high
//      a-face ARE the nodes of f's low a-face, position by position (same orientation) - found by sorting the faces'
//      corner nodes, verified on every node; no index arithmetic on zone or node numbers anywhere;
//   2. a flood fill over those relations gives every zone integer coordinates (i, j, k) inside its connected component;
//      the relations must be consistent with them (a mesh that wraps around, or whose blocks are rotated against each
//      other where they meet, is not a structured block: the library then keeps the caller's order - correct, general
//      path, as before);
//   3. internal zone order = components in order of their first zone, zones by (k, j, i): rows along x, rows of rows;
//   4. internal node number = rank of the node under (component, z, y, x) of its integer coordinates
//      (i p + dx, j p + dy, k p + dz), taken from the first zone in internal order that holds it.
// What is kept in the internal numbering is everything the library owns in the velocity solve: r, d, x, 1/diag, flag
// bytes, the E-vector and its transpose tables, the K1 schedule; the caller's vectors are read (right-hand side) and
// written (dv/dt) once per solve through the permutation.  The quadrature update walks the zones in internal order and
// reads / writes every per-zone array at the caller's position.  On a mesh that already comes in this order (this
// repository's generator) both permutations are the identity and nothing changes - not even a bit.
// LGH_ORDER=0: keep the caller's numbering (A/B; the general-mesh path of rounds 1-5).
#include "lgh_common.hpp"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <numeric>

namespace lgh
{

static void free_order(MeshOrder *o)
{
   if (!o) { return; }
   (void)hipFree(o->zorder_d);
   (void)hipFree(o->ncaller_d);
   (void)hipFree(o->nnum_d);
   delete o;
}
void mesh_order_free(lgh_ctx *c)
{
   free_order((MeshOrder *)c->order);
   c->order = nullptr;
}

// host part: fills zorder / nnum / ncaller and the statistics; returns false when the map is not a set of structured
// blocks (the permutations are then left empty = identity)
static bool analyse(const int *map, const int NE, const int N, const int D, MeshOrder &o)
{
   const int ND = D * D * D, p = D - 1;
   std::vector<int> nxt[3], prv[3];
   for (int a = 0; a < 3; a++)
   {
      // local dofs of the low / high a-face, (u, v) over the other two axes in ascending axis order
      std::vector<int> lo((size_t)D * D), hi((size_t)D * D);
      const int st[3] = {1, D, D * D};
      const int au = (a == 0) ? 1 : 0, av = (a == 2) ? 1 : 2;
      for (int v = 0; v < D; v++)
         for (int u = 0; u < D; u++)
         {
            lo[(size_t)u + D * v] = u * st[au] + v * st[av];
            hi[(size_t)u + D * v] = u * st[au] + v * st[av] + p * st[a];
         }
      const int corner[4] = {0, D - 1, D * (D - 1), D * D - 1};
      struct Key { int c[4]; int e; };
      std::vector<Key> keys((size_t)NE);
      for (int e = 0; e < NE; e++)
      {
         for (int k = 0; k < 4; k++) { keys[e].c[k] = map[(size_t)e * ND + lo[corner[k]]]; }
         keys[e].e = e;
      }
      auto less = [](const Key &x, const Key &y) { return std::lexicographical_compare(x.c, x.c + 4, y.c, y.c + 4); };
      std::sort(keys.begin(), keys.end(), [&](const Key &x, const Key &y) { return less(x, y) || (!less(y, x) && x.e < y.e); });
      nxt[a].assign((size_t)NE, -1);
      prv[a].assign((size_t)NE, -1);
      for (int e = 0; e < NE; e++)
      {
         Key q;
         for (int k = 0; k < 4; k++) { q.c[k] = map[(size_t)e * ND + hi[corner[k]]]; }
         q.e = -1;
         auto it = std::lower_bound(keys.begin(), keys.end(), q, less);
         for (; it != keys.end() && !less(q, *it); ++it)
         {
            const int f = it->e;
            if (f == e) { continue; }
            bool same = true;
            for (int k = 0; same && k < D * D; k++) { same = map[(size_t)e * ND + hi[k]] == map[(size_t)f * ND + lo[k]]; }
            if (!same) { continue; }
            if (nxt[a][e] >= 0 || prv[a][f] >= 0) { return false; } // (a face shared by more than two zones: not a manifold mesh)
            nxt[a][e] = f;
            prv[a][f] = e;
         }
      }
   }
   // flood fill: integer coordinates inside each connected component
   std::vector<int> comp((size_t)NE, -1);
   std::vector<std::array<int, 3>> ijk((size_t)NE);
   std::vector<int> queue;
   queue.reserve((size_t)NE);
   int ncomp = 0;
   std::vector<std::array<int, 3>> cmin, cmax;
   for (int seed = 0; seed < NE; seed++)
   {
      if (comp[seed] >= 0) { continue; }
      const int cc = ncomp++;
      comp[seed] = cc;
      ijk[seed] = {0, 0, 0};
      std::array<int, 3> mn{0, 0, 0}, mx{0, 0, 0};
      size_t head = queue.size();
      queue.push_back(seed);
      while (head < queue.size())
      {
         const int e = queue[head++];
         for (int a = 0; a < 3; a++)
            for (int s = -1; s <= 1; s += 2)
            {
               const int f = (s > 0) ? nxt[a][e] : prv[a][e];
               if (f < 0) { continue; }
               std::array<int, 3> want = ijk[e];
               want[a] += s;
               if (comp[f] < 0)
               {
                  comp[f] = cc;
                  ijk[f] = want;
                  for (int b = 0; b < 3; b++) { mn[b] = std::min(mn[b], want[b]); mx[b] = std::max(mx[b], want[b]); }
                  queue.push_back(f);
               }
               else if (comp[f] != cc || ijk[f] != want) { return false; } // the block wraps around / is twisted
            }
      }
      cmin.push_back(mn);
      cmax.push_back(mx);
   }
   // two zones of a component at the same place (cannot happen on a manifold block; cheap to exclude)
   {
      std::vector<std::array<long, 2>> where((size_t)NE);
      for (int e = 0; e < NE; e++)
      {
         const int cc = comp[e];
         const long ex = cmax[cc][0] - cmin[cc][0] + 1, ey = cmax[cc][1] - cmin[cc][1] + 1;
         where[e] = {cc, (ijk[e][0] - cmin[cc][0]) + ex * ((ijk[e][1] - cmin[cc][1]) + ey * (long)(ijk[e][2] - cmin[cc][2]))};
      }
      std::sort(where.begin(), where.end());
      for (int e = 1; e < NE; e++) { if (where[e] == where[e - 1]) { return false; } }
   }
   // zones: components in the order of their first zone, then (k, j, i)
   o.zorder.resize((size_t)NE);
   std::iota(o.zorder.begin(), o.zorder.end(), 0);
   auto zkey = [&](const int e) {
      const int cc = comp[e];
      return std::array<long, 4>{cc, ijk[e][2] - cmin[cc][2], ijk[e][1] - cmin[cc][1], ijk[e][0] - cmin[cc][0]};
   };
   std::sort(o.zorder.begin(), o.zorder.end(), [&](const int x, const int y) { return zkey(x) < zkey(y); });
   // nodes: integer coordinates from the first zone (internal order) that holds the node
   std::vector<std::array<long, 4>> nkey((size_t)N, std::array<long, 4>{-1, 0, 0, 0});
   for (int i = 0; i < NE; i++)
   {
      const int e = o.zorder[i];
      const auto zk = zkey(e);
      for (int d = 0; d < ND; d++)
      {
         const int n = map[(size_t)e * ND + d];
         if (nkey[n][0] >= 0) { continue; }
         const int dx = d % D, dy = (d / D) % D, dz = d / (D * D);
         nkey[n] = {zk[0], zk[1] * p + dz, zk[2] * p + dy, zk[3] * p + dx};
      }
   }
   o.ncaller.resize((size_t)N);
   std::iota(o.ncaller.begin(), o.ncaller.end(), 0);
   // (a node no zone holds sorts behind everything, by its own number)
   auto nk = [&](const int n) { return nkey[n][0] >= 0 ? nkey[n] : std::array<long, 4>{(long)ncomp, 0, 0, (long)n}; };
   std::sort(o.ncaller.begin(), o.ncaller.end(), [&](const int x, const int y) {
      const auto kx = nk(x), ky = nk(y);
      return kx < ky || (kx == ky && x < y);
   });
   o.nnum.resize((size_t)N);
   for (int m = 0; m < N; m++) { o.nnum[o.ncaller[m]] = m; }
   o.components = ncomp;
   for (int b = 0; b < 3; b++) { o.extent[b] = cmax[0][b] - cmin[0][b] + 1; }
   return true;
}

int mesh_order_build(lgh_ctx *c, const int *map)
{
   mesh_order_free(c);
   MeshOrder *o = new MeshOrder();
   c->order = o;
   const char *env = getenv("LGH_ORDER"); // A/B: 0 = the caller's numbering everywhere (rounds 1-5)
   if (c->dim != 3 || (env && env[0] == '0')) { return LGH_OK; }
   o->structured = analyse(map, c->NE, c->N, c->D1D, *o);
   if (!o->structured)
   {
      o->zorder.clear();
      o->nnum.clear();
      o->ncaller.clear();
      return LGH_OK;
   }
   bool ident = true;
   for (int e = 0; ident && e < c->NE; e++) { ident = o->zorder[e] == e; }
   for (int n = 0; ident && n < c->N; n++) { ident = o->ncaller[n] == n; }
   o->identity = ident;
   if (ident) { return LGH_OK; }
   LGH_HIP_CHECK(hipMalloc((void **)&o->zorder_d, (size_t)c->NE * sizeof(int)));
   LGH_HIP_CHECK(hipMalloc((void **)&o->ncaller_d, (size_t)c->N * sizeof(int)));
   LGH_HIP_CHECK(hipMalloc((void **)&o->nnum_d, (size_t)c->N * sizeof(int)));
   LGH_HIP_CHECK(hipMemcpy(o->zorder_d, o->zorder.data(), (size_t)c->NE * sizeof(int), hipMemcpyHostToDevice));
   LGH_HIP_CHECK(hipMemcpy(o->ncaller_d, o->ncaller.data(), (size_t)c->N * sizeof(int), hipMemcpyHostToDevice));
   LGH_HIP_CHECK(hipMemcpy(o->nnum_d, o->nnum.data(), (size_t)c->N * sizeof(int), hipMemcpyHostToDevice));
   return LGH_OK;
}

// ---- permutation kernels -------------------------------------------------------------------------------------------
// out[c * N + m] = in[c * N + idx[m]]  (caller -> internal with idx = ncaller)
__global__ void __launch_bounds__(256) order_gather_k(const double *__restrict__ in, const int *__restrict__ idx, double *__restrict__ out, const int N, const int ncomp)
{
   const int m = blockIdx.x * blockDim.x + threadIdx.x;
   if (m >= N) { return; }
   const int n = idx[m];
   for (int c = 0; c < ncomp; c++) { out[(size_t)c * N + m] = in[(size_t)c * N + n]; }
}
// out[c * N + idx[m]] = in[c * N + m]  (internal -> caller with idx = ncaller)
__global__ void __launch_bounds__(256) order_scatter_k(const double *__restrict__ in, const int *__restrict__ idx, double *__restrict__ out, const int N, const int ncomp)
{
   const int m = blockIdx.x * blockDim.x + threadIdx.x;
   if (m >= N) { return; }
   const int n = idx[m];
   for (int c = 0; c < ncomp; c++) { out[(size_t)c * N + n] = in[(size_t)c * N + m]; }
}
__global__ void __launch_bounds__(256) order_gather_u8_k(const uint8_t *__restrict__ in, const int *__restrict__ idx, uint8_t *__restrict__ out, const int N)
{
   const int m = blockIdx.x * blockDim.x + threadIdx.x;
   if (m < N) { out[m] = in[idx[m]]; }
}
// blocks of `per` doubles: out block i = in block idx[i]  (zone data, caller -> internal with idx = zorder), or the other way round
__global__ void __launch_bounds__(256) order_blocks_k(const double *__restrict__ in, const int *__restrict__ idx, double *__restrict__ out, const size_t nblk, const int per,
                                                      const int scatter)
{
   const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
   if (t >= nblk * per) { return; }
   const size_t i = t / per, k = t - i * per;
   const size_t j = (size_t)idx[i];
   if (scatter) { out[j * per + k] = in[t]; }
   else { out[t] = in[j * per + k]; }
}

int order_gather_nodes(lgh_ctx *c, const double *in, double *out, int ncomp)
{
   const MeshOrder *o = (const MeshOrder *)c->order;
   hipLaunchKernelGGL(order_gather_k, dim3(ceil_div(c->N, 256)), dim3(256), 0, c->stream, in, o->ncaller_d, out, c->N, ncomp);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int order_scatter_nodes(lgh_ctx *c, const double *in, double *out, int ncomp)
{
   const MeshOrder *o = (const MeshOrder *)c->order;
   hipLaunchKernelGGL(order_scatter_k, dim3(ceil_div(c->N, 256)), dim3(256), 0, c->stream, in, o->ncaller_d, out, c->N, ncomp);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int order_gather_bytes(lgh_ctx *c, const uint8_t *in, uint8_t *out)
{
   const MeshOrder *o = (const MeshOrder *)c->order;
   hipLaunchKernelGGL(order_gather_u8_k, dim3(ceil_div(c->N, 256)), dim3(256), 0, c->stream, in, o->ncaller_d, out, c->N);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int order_zone_blocks(lgh_ctx *c, const double *in, double *out, int per, bool to_caller)
{
   const MeshOrder *o = (const MeshOrder *)c->order;
   const size_t n = (size_t)c->NE * per;
   hipLaunchKernelGGL(order_blocks_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, in, o->zorder_d, out, (size_t)c->NE, per, to_caller ? 1 : 0);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

} // namespace lgh

extern "C" int lgh_mesh_order_host(int dim, int NE, int N, int D1D, const int *h1_map, int *zorder, int *nnum, long out[8])
{
   LGH_CHECK_ARG(h1_map && zorder && nnum && out && NE > 0 && N > 0 && D1D >= 2);
   for (int i = 0; i < 8; i++) { out[i] = 0; }
   for (int e = 0; e < NE; e++) { zorder[e] = e; }
   for (int n = 0; n < N; n++) { nnum[n] = n; }
   out[1] = 1;
   if (dim != 3) { return LGH_OK; }
   const size_t nmap = (size_t)NE * D1D * D1D * D1D;
   for (size_t i = 0; i < nmap; i++) { LGH_CHECK_ARG(h1_map[i] >= 0 && h1_map[i] < N); }
   lgh::MeshOrder o;
   if (!lgh::analyse(h1_map, NE, N, D1D, o)) { return LGH_OK; }
   bool ident = true;
   for (int e = 0; e < NE; e++) { zorder[e] = o.zorder[e]; ident = ident && o.zorder[e] == e; }
   for (int n = 0; n < N; n++) { nnum[n] = o.nnum[n]; ident = ident && o.nnum[n] == n; }
   out[0] = 1;
   out[1] = ident ? 1 : 0;
   out[2] = o.components;
   for (int b = 0; b < 3; b++) { out[3 + b] = o.extent[b]; }
   return LGH_OK;
}

extern "C" int lgh_mesh_order(lgh_ctx *c, long out[8])
{
   LGH_CHECK_ARG(c && out);
   const lgh::MeshOrder *o = (const lgh::MeshOrder *)c->order;
   for (int i = 0; i < 8; i++) { out[i] = 0; }
   if (!o) { return LGH_OK; }
   out[0] = o->structured ? 1 : 0;
   out[1] = o->identity ? 1 : 0;
   out[2] = o->components;
   out[3] = o->extent[0];
   out[4] = o->extent[1];
   out[5] = o->extent[2];
   return LGH_OK;
}
