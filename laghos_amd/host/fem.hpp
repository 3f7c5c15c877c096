// fem.hpp — the minimal finite-element substrate the Laghos hot path needs on
// the host, standing in for the parts of MFEM the reference driver uses
// (/root/reference/laghos.cpp:378-632): axis-aligned tensor-product meshes with
// uniform refinement and block partitioning, the H1 (Gauss-Lobatto) and L2
// (Bernstein) tensor bases with their 1-D DofToQuad tables, lexicographic
// element restrictions, boundary-attribute essential dofs, and the initial
// conditions of problems 0-7 (laghos.cpp:1094-1275).  Host-only, no GPU code.
//
// This is deliberately not a general FE library: just enough to
// drive and verify the partial-assembly path.
#pragma once
#include <array>
#include <string>
#include <vector>

namespace laghos
{

// ---- 1-D rules and bases on [0,1] -------------------------------------------------
void GaussLegendre(int n, std::vector<double> &x, std::vector<double> &w);
void GaussLobatto(int n, std::vector<double> &x);
// B[q + Q*d] = l_d(pts[q]), G likewise (Lagrange basis on `nodes`)
void LagrangeTables(const std::vector<double> &nodes, const std::vector<double> &pts,
                    std::vector<double> &B, std::vector<double> &G);
// B[q + Q*l] = C(p,l) x^l (1-x)^(p-l)
void BernsteinTable(int p, const std::vector<double> &pts, std::vector<double> &B);

// DofToQuad::TENSOR data for one (order_v, order_e) pair (laghos_assembly.cpp:141-142)
struct Tables
{
   int order_v, order_e, D1D, L1D, Q1D;
   std::vector<double> qpts, qwts, gll;
   std::vector<double> B, G; // H1, [q + Q1D*d]
   std::vector<double> Bl;   // L2, [q + Q1D*l]
   // integration rule order 3*ok+ot-1 unless oq > 0 (laghos_solver.cpp:145-147)
   Tables(int order_v, int order_e, int order_q = -1);
};

// ---- mesh ---------------------------------------------------------------------------
// Tensor-product mesh given by per-axis break points; element e = ex + nx*(ey + ny*ez).
struct CartMesh
{
   int dim = 0;
   std::array<std::vector<double>, 3> brk;
   // the structured meshes shipped in /root/reference/data
   static CartMesh Named(const std::string &name);
   static CartMesh Cartesian(int dim, int nx, int ny, int nz, double sx, double sy, double sz);
   void UniformRefinement();
   int ne(int a) const { return (int)brk[a].size() - 1; }
   long NE() const;
};

// Block partition of the element grid over `nranks` processes (one per GPU).
struct Partition
{
   int dim = 0, nranks = 1, rank = 0;
   std::array<int, 3> pgrid{1, 1, 1}, rcoord{0, 0, 0};
   std::array<int, 3> ne{1, 1, 1}, eoff{0, 0, 0}; // local element counts / offsets
   Partition() {}
   Partition(const CartMesh &mesh, int nranks, int rank);
};

// ---- discretisation of one rank's block ---------------------------------------------------
struct Discretization
{
   int dim, problem;
   Tables tab;
   CartMesh mesh;   // global
   Partition part;
   std::array<std::vector<double>, 3> brk; // local break points
   int NE, ND, NQ, NL;
   std::array<int, 3> nn{1, 1, 1};         // local H1 nodes per axis
   int N;                                  // local scalar H1 nodes
   long global_N, global_NE;
   int H1V, L2V;
   std::vector<double> W;                  // NQ tensor weights
   std::vector<int> h1map;                 // NE*ND
   std::array<std::vector<int>, 3> ess;    // essential scalar nodes per component
   std::vector<double> owner;              // N (1 = owned by this rank)
   // neighbours (ranks sharing H1 nodes) and the shared local node lists
   std::vector<int> nbr_rank;
   std::vector<std::vector<int>> nbr_nodes;
   double blast_energy = 1.0;
   // `-renumber mfem|random` (Renumber below): the numbering a general mesh library hands the operators instead of this
   // generator's own lexicographic one.  Empty = identity.
   std::vector<int> node_perm;             // structured node i is node node_perm[i]
   std::vector<int> elem_perm;             // zone j is the structured zone elem_perm[j]
   std::string numbering = "lexicographic";

   Discretization(const CartMesh &mesh, int order_v, int order_e, int problem, int nranks = 1,
                  int rank = 0, int order_q = -1, double blast_energy = 1.0);

   // S = [x | v | e]; rho0 grid function (L2 dofs), gamma per element, rho0 at qpts
   void InitialState(std::vector<double> &S, std::vector<double> &rho0_l2,
                     std::vector<double> &gamma, std::vector<double> &rho0_q) const;
   // Renumber the H1 nodes and reorder the zones of this rank's block (h1map, ess, owner, nbr_nodes; InitialState follows):
   //   "mfem":   the numbering MFEM gives `levels` uniform refinements of the lexicographic base mesh (what upstream Laghos
   //             hands its operators: laghos.cpp:391, laghos_assembly.cpp:133-134) - zones in refinement-tree order (the
   //             2^dim children of a zone consecutive, in the order of the parent's vertices), H1 dofs as vertices, then
   //             edge, face (3D) and interior dofs, each class in the order the zones first meet the entity;
   //   "random": a seeded random permutation of both.
   // The element-local dof order stays lexicographic: that is the interface (ElementDofOrdering::LEXICOGRAPHIC).
   void Renumber(const std::string &mode, int levels, unsigned seed = 1);
   bool impose_visc = false; // -iv (laghos.cpp:648)
   bool UseViscosity() const { return impose_visc || (problem != 0 && problem != 4); } // laghos.cpp:636-648
   int SourceType() const { return problem == 7 ? 2 : ((problem == 0 && dim == 2) ? 1 : 0); } // laghos.cpp:636-647
   bool UseVorticity() const { return problem == 7; }

   // problem definitions (laghos.cpp:1094-1275)
   double rho0(const double *x) const;
   double gamma_func(const double *x) const;
   void v0(const double *x, double *v) const;
   double e0(const double *x) const;

private:
   void ElemPoint(int e, const double *ref, double *x) const;
   void NodalToBernstein(std::vector<double> &vals) const;
};

} // namespace laghos
