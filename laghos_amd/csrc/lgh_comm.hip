// lgh_comm.hip — multi-GPU support: element blocks per rank, shared H1 nodes
// summed across ranks and scalar all-reduces, over RCCL on the context stream.
//
// Replaces what the reference gets from MFEM's ParFiniteElementSpace /
// GroupCommunicator over MPI (call-site inventory in SURVEY §2): the conforming
// prolongation P^T (sum of shared dofs, inside mass->Mult laghos_assembly.cpp:119
// and Pconf->MultTranspose laghos_solver.cpp:368), CGSolver::Dot's MPI_Allreduce
// and the MIN reduction of the time-step estimate (laghos_solver.cpp:533).
//
// MI355X design: xGMI is point-to-point, and a 2x2x2 block partition gives every
// GPU 7 neighbours = its 7 links, so the halo is one grouped ncclSend/ncclRecv
// per neighbour (<= 75 KB each at 32^3 Q3 elements) rather than a ring
// collective; pack and unpack-add are single kernels over all neighbours.
// RCCL is resolved with dlopen at lgh_comm_init time so that the same library
// loads on a host without RCCL (and reuses torch's copy when already loaded).
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdlib>
#include <cmath>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>

#include "lgh_common.hpp"

namespace lgh
{

// minimal NCCL ABI (rccl.h): opaque handles + the enums used here
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat64 = 8 };
enum { ncclSum = 0, ncclMin = 3 };

struct NcclApi
{
   void *lib = nullptr;
   int (*GetUniqueId)(ncclUniqueId *) = nullptr;
   int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
   int (*CommDestroy)(ncclComm_t) = nullptr;
   int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
   int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr; // optional
   int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
   int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
   int (*GroupStart)() = nullptr;
   int (*GroupEnd)() = nullptr;
   const char *(*GetErrorString)(int) = nullptr;
};
static NcclApi g_nccl;

static int load_nccl()
{
   if (g_nccl.lib) { return LGH_OK; }
   const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
   void *h = nullptr;
   for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) { break; } }
   if (!h) { for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) { break; } } }
   if (!h)
   {
      set_error("cannot load RCCL: %s", dlerror());
      return LGH_ERR_COMM;
   }
   g_nccl.lib = h;
#define LGH_SYM(field, name)                                                        \
   *(void **)(&g_nccl.field) = dlsym(h, name);                                     \
   if (!g_nccl.field) { set_error("RCCL symbol %s missing", name); g_nccl.lib = nullptr; return LGH_ERR_COMM; }
   LGH_SYM(GetUniqueId, "ncclGetUniqueId");
   LGH_SYM(CommInitRank, "ncclCommInitRank");
   LGH_SYM(CommDestroy, "ncclCommDestroy");
   LGH_SYM(AllReduce, "ncclAllReduce");
   LGH_SYM(Send, "ncclSend");
   LGH_SYM(Recv, "ncclRecv");
   LGH_SYM(GroupStart, "ncclGroupStart");
   LGH_SYM(GroupEnd, "ncclGroupEnd");
   LGH_SYM(GetErrorString, "ncclGetErrorString");
#undef LGH_SYM
   *(void **)(&g_nccl.Broadcast) = dlsym(h, "ncclBroadcast");
   return LGH_OK;
}

#define LGH_NCCL_CHECK(expr)                                                             \
   do                                                                                    \
   {                                                                                     \
      int r_ = (expr);                                                                   \
      if (r_ != ncclSuccess)                                                             \
      {                                                                                  \
         set_error("RCCL error %s at %s:%d", g_nccl.GetErrorString(r_), __FILE__, __LINE__); \
         return LGH_ERR_COMM;                                                            \
      }                                                                                  \
   } while (0)

// ---- in-process loopback communicator (test vehicle) -------------------------------
// A unique id that starts with "LGHLOCAL" selects it: the ranks are contexts of ONE
// process (one host thread each, any devices that can copy to each other - in the
// tests all on the single GPU of the box) and the two collectives are carried out
// with stream synchronisation, host barriers and device-to-device copies.  Everything
// else - owner masks, pack / canonical combine kernels, the separate-gather CG
// sequencing, finish kernels, dt and norm reductions - is the code that runs over RCCL
// on a node, so a multi-rank run can be checked against a single-rank one without a
// multi-GPU machine.  Not a product path: RCCL over xGMI is.
struct LocalGroup
{
   int n = 0;
   std::mutex m;
   std::condition_variable cv;
   int arrived = 0;
   long gen = 0;
   bool broken = false;
   std::vector<lgh_ctx *> ctx;   // by rank
   std::vector<double> slots;    // n * 8 doubles of all-reduce staging
   bool barrier()
   {
      std::unique_lock<std::mutex> lk(m);
      if (broken) { return false; }
      const long g = gen;
      if (++arrived == n)
      {
         arrived = 0;
         gen++;
         cv.notify_all();
         return true;
      }
      // a rank that never arrives (diverged control flow) must fail the test, not hang it
      if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return gen != g || broken; }))
      {
         broken = true;
         cv.notify_all();
         return false;
      }
      return !broken;
   }
};
static std::mutex g_local_m;
static std::map<std::string, std::shared_ptr<LocalGroup>> g_local;

// ---- cross-process loopback communicator (test vehicle, bench.py --transport shm) ---------------
// A unique id that starts with "LGHSHM" names a POSIX shared-memory segment: the ranks are PROCESSES (what torchrun
// starts: one per rank, as on a multi-GPU node) that may all sit on one GPU - RCCL refuses two ranks on one device, so
// the one-GPU box cannot run the product transport with N > 1.  The exchanges are staged through the segment (device ->
// segment -> device) between process-shared barriers with a time-out.  Everything above the transport - torchrun
// rendezvous, broadcast of the id, lgh_comm_init / lgh_comm_set_neighbors, partition, owner masks, pack / canonical
// combine, piggy-backed sums, the sequencing of the solves, bench.py's rank-0 line - is the code a node runs.
// Not a product path: RCCL over xGMI is.
struct ShmHeader
{
   std::atomic<unsigned> magic;    // kShmMagic once rank 0 has initialised the segment
   std::atomic<int> arrived;       // sense-reversing barrier
   std::atomic<int> gen;
   std::atomic<int> broken;
   int nranks;
   size_t cap;                     // bytes of message area per rank and channel
   size_t off_slots, off_tables, off_msg;
};
constexpr unsigned kShmMagic = 0x4C474853u; // "LGHS"
static std::chrono::seconds shm_timeout() // how long a rank waits for its peers (LGH_SHM_TIMEOUT seconds, default 120)
{
   const char *e = getenv("LGH_SHM_TIMEOUT");
   return std::chrono::seconds((e && atoi(e) > 0) ? atoi(e) : 120);
}
constexpr int kShmMaxNbr = 32;
struct ShmTable // what a rank publishes about its send buffer: where the block of each neighbour starts
{
   int n_nbr;
   int nbr_rank[kShmMaxNbr], nbr_base[kShmMaxNbr], nbr_count[kShmMaxNbr];
};
struct ShmGroup
{
   std::string name;
   void *base = nullptr;
   size_t bytes = 0;
   int n = 0, rank = 0;
   ShmHeader *hdr() const { return (ShmHeader *)base; }
   double *slots(int r) const { return (double *)((char *)base + hdr()->off_slots) + (size_t)r * 8; }
   ShmTable *table(int r) const { return (ShmTable *)((char *)base + hdr()->off_tables) + r; }
   double *msg(int r, int ch) const { return (double *)((char *)base + hdr()->off_msg + ((size_t)r * 2 + ch) * hdr()->cap); }
   bool barrier()
   {
      ShmHeader *h = hdr();
      if (h->broken.load()) { return false; }
      const int g = h->gen.load();
      if (h->arrived.fetch_add(1) + 1 == n)
      {
         h->arrived.store(0);
         h->gen.fetch_add(1);
         return true;
      }
      const auto t0 = std::chrono::steady_clock::now();
      long spins = 0;
      while (h->gen.load() == g)
      {
         if (h->broken.load()) { return false; }
         if ((++spins & 1023) == 0)
         {
            // a rank that never arrives (diverged control flow, dead process) must fail the run, not hang it
            if (std::chrono::steady_clock::now() - t0 > shm_timeout()) { h->broken.store(1); return false; }
            usleep(50);
         }
      }
      return !h->broken.load();
   }
   ~ShmGroup()
   {
      if (base) { munmap(base, bytes); }
      if (rank == 0 && !name.empty()) { shm_unlink(name.c_str()); }
   }
};
static int shm_open_group(const char unique_id[128], int nranks, int rank, std::shared_ptr<ShmGroup> &out)
{
   auto g = std::make_shared<ShmGroup>();
   g->name = std::string("/") + std::string(unique_id, strnlen(unique_id, 64));
   g->n = nranks;
   g->rank = rank;
   const char *menv = getenv("LGH_SHM_MB"); // message area per rank and channel (a 32^3 Q3Q2 block sends 1.4 MB per exchange)
   const size_t cap = (size_t)std::max(1, menv ? atoi(menv) : 16) << 20;
   const size_t off_slots = 4096, off_tables = off_slots + ((size_t)nranks * 8 * sizeof(double) + 4095) / 4096 * 4096;
   const size_t off_msg = off_tables + ((size_t)nranks * sizeof(ShmTable) + 4095) / 4096 * 4096;
   g->bytes = off_msg + (size_t)nranks * 2 * cap;
   int fd = -1;
   if (rank == 0)
   {
      shm_unlink(g->name.c_str());
      fd = shm_open(g->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)g->bytes) != 0) { set_error("shm transport: cannot create %s", g->name.c_str()); if (fd >= 0) { close(fd); } return LGH_ERR_COMM; }
   }
   else
   {
      const auto t0 = std::chrono::steady_clock::now();
      while (true)
      {
         fd = shm_open(g->name.c_str(), O_RDWR, 0600);
         struct stat st;
         // (the size is rank 0's: its LGH_SHM_MB decides, whatever this rank's environment says)
         if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= off_msg) { g->bytes = (size_t)st.st_size; break; }
         if (fd >= 0) { close(fd); fd = -1; }
         if (std::chrono::steady_clock::now() - t0 > shm_timeout()) { set_error("shm transport: %s did not appear (rank 0 missing?)", g->name.c_str()); return LGH_ERR_COMM; }
         usleep(2000);
      }
   }
   g->base = mmap(nullptr, g->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
   close(fd);
   if (g->base == MAP_FAILED) { g->base = nullptr; set_error("shm transport: mmap failed"); return LGH_ERR_COMM; }
   ShmHeader *h = g->hdr();
   if (rank == 0)
   {
      h->arrived.store(0); h->gen.store(0); h->broken.store(0);
      h->nranks = nranks; h->cap = cap; h->off_slots = off_slots; h->off_tables = off_tables; h->off_msg = off_msg;
      h->magic.store(kShmMagic);
   }
   else
   {
      const auto t0 = std::chrono::steady_clock::now();
      while (h->magic.load() != kShmMagic)
      {
         if (std::chrono::steady_clock::now() - t0 > shm_timeout()) { set_error("shm transport: segment never initialised"); return LGH_ERR_COMM; }
         usleep(1000);
      }
      if (h->nranks != nranks) { set_error("shm transport: segment is for %d ranks, not %d", h->nranks, nranks); return LGH_ERR_COMM; }
   }
   out = g;
   return LGH_OK;
}

struct Comm
{
   ncclComm_t comm = nullptr;
   // Second channel: a communicator and message buffers of its own for the stream the energy solve runs on
   // beside the velocity solve (lgh_solve_energy_begin/_end) - one communicator must not be driven from two
   // streams.  It only ever carries sums of scalars.  Present on every rank or on none (decided collectively).
   ncclComm_t comm2 = nullptr;
   bool channel2 = false;
   double *sendbuf2 = nullptr, *recvbuf2 = nullptr;
   std::shared_ptr<LocalGroup> local;
   std::shared_ptr<ShmGroup> shm;
   std::vector<double> stage; // shm transport: host staging of one receive buffer
   int n_nbr = 0;
   std::vector<int> nbr_rank, nbr_count, nbr_off, nbr_base; // node offsets / buffer bases of the neighbours
   int total = 0;        // sum of nbr_count
   int *nodes = nullptr; // device: concatenated neighbour node lists
   double *sendbuf = nullptr, *recvbuf = nullptr; // device: total * 3 doubles
   int n_shared = 0;                               // unique shared nodes
   int *sh_node = nullptr, *sh_off = nullptr, *sh_src = nullptr; // device CSR (see halo_combine_k)
   int *pos = nullptr, *cnt = nullptr; // per concatenated entry: buffer position, neighbour count
   uint8_t *hmask = nullptr;           // device: 1 for every shared node
   // piggy-backed scalars: when every other rank is a neighbour (block partitions of up
   // to 2x2x2) a few doubles ride on each halo message and are summed in rank order by
   // the combine kernel - one collective less per CG iteration
   bool allpairs = false;
   int *d_base = nullptr, *d_cnt = nullptr; // device, per neighbour: block base, node count
   int *rank_src = nullptr;                 // device, per rank: neighbour index or -1 (self)
   int bufsize = 0;                         // doubles in sendbuf / recvbuf
   // exchange_words: the peers' accumulator words (n_nbr blocks of wordcap words), and where this rank's own words are
   // (the in-process transport copies device to device out of the peer's accumulators)
   long long *wordbuf = nullptr;
   int wordcap = 0, wordnbr = 0; // words per peer and peers the buffer was allocated for
   const long long *word_src = nullptr;
   std::vector<long long> wstage;
};
constexpr int kHaloSlack = 4; // doubles of room behind every neighbour's block (<= 4 scalars: (d, A d) of the three velocity components and, in lockstep, of the energy CG)

// Buffers: neighbour k owns the contiguous block starting at base_k = 3*off_k +
// kHaloSlack*k: ncomp*cnt_k node values, component-major, then (optionally) nextra
// piggy-backed scalars - so every neighbour is ONE send and ONE recv.
// pos[j] = base_k + i for entry j = off_k + i of the concatenated node lists;
// cnt[j] = cnt_k.
// nextra > 0: the first n_nbr*nextra threads also put the piggy-backed scalars behind every
// neighbour's node block.
__global__ void __launch_bounds__(256)
halo_pack_k(const int total, const int ncomp, const int N, const int *__restrict__ nodes,
            const int *__restrict__ pos, const int *__restrict__ cnt, const double *__restrict__ v,
            double *__restrict__ buf, const int n_nbr, const int nextra, const int *__restrict__ base,
            const int *__restrict__ ncnt, const double *__restrict__ extra)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n_nbr * nextra)
   {
      const int k = i / nextra, e = i - k * nextra;
      buf[(size_t)base[k] + (size_t)ncomp * ncnt[k] + e] = extra[e];
   }
   if (i >= total * ncomp) { return; }
   const int c = i / total, j = i - c * total;
   buf[(size_t)pos[j] + (size_t)c * cnt[j]] = v[(size_t)c * N + nodes[j]];
}
// Canonical sum of a shared node: contributions are added in ascending rank
// order (own value at its rank's position), so every rank holding the node
// computes bit-identical results, as MFEM's GroupCommunicator does.  CSR over the
// unique shared nodes; src >= 0: entry j of the concatenated lists, -1: own value.
// nextra > 0: threads e < nextra of the launch also sum the piggy-backed scalars over all ranks in
// ascending rank order (own value at its rank's position): bit-identical on every rank.
__global__ void __launch_bounds__(256)
halo_combine_k(const int n_shared, const int ncomp, const int N, const int *__restrict__ sh_node,
               const int *__restrict__ sh_off, const int *__restrict__ sh_src,
               const int *__restrict__ pos, const int *__restrict__ cnt,
               const double *__restrict__ buf, double *__restrict__ v, const int nranks, const int nextra,
               const int *__restrict__ rank_src, const int *__restrict__ base, const int *__restrict__ ncnt,
               double *__restrict__ extra)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < nextra)
   {
      double s = 0.0;
      for (int r = 0; r < nranks; r++)
      {
         const int k = rank_src[r];
         const double val = (k < 0) ? extra[i] : buf[(size_t)base[k] + (size_t)ncomp * ncnt[k] + i];
         s = (r == 0) ? val : s + val;
      }
      extra[i] = s;
   }
   if (i >= n_shared * ncomp) { return; }
   const int c = i / n_shared, u = i - c * n_shared;
   const int node = sh_node[u];
   double s = 0.0;
   for (int k = sh_off[u]; k < sh_off[u + 1]; k++)
   {
      const int j = sh_src[k];
      const double val = (j < 0) ? v[(size_t)c * N + node] : buf[(size_t)pos[j] + (size_t)c * cnt[j]];
      s = (k == sh_off[u]) ? val : s + val;
   }
   v[(size_t)c * N + node] = s;
}

bool halo_can_piggyback(const lgh_ctx *c)
{
   const bool on = !(getenv("LGH_HALO_PIGGYBACK") && getenv("LGH_HALO_PIGGYBACK")[0] == '0');
   return on && c->comm && c->comm->allpairs;
}

// extra != nullptr (device, nextra <= 3 doubles, requires halo_can_piggyback): replaced
// by its sum over all ranks, carried by the same messages
// alias (optional): v is numbered otherwise than the caller's vectors (the velocity solve's own node numbering,
// lgh_order.hip) - the same node lists in THAT numbering; everything else of the exchange does not know about numbers
int halo_sum(lgh_ctx *c, double *v, int ncomp, double *extra, int nextra, bool packed, const HaloNodeAlias *alias)
{
   Comm *cm = c->comm;
   if (!cm || cm->n_nbr == 0) { return LGH_OK; }
   const int *const l_nodes = alias ? alias->nodes : cm->nodes, *const l_sh_node = alias ? alias->sh_node : cm->sh_node;
   if (ncomp > 3 || ncomp < 0 || (ncomp == 0 && !extra)) { set_error("halo_sum: bad ncomp"); return LGH_ERR_ARG; }
   if (extra && (!cm->allpairs || nextra < 1 || nextra > kHaloSlack)) { set_error("halo_sum: cannot piggy-back"); return LGH_ERR_ARG; }
   const int nx = extra ? nextra : 0;
   const int tot = cm->total;
   const bool ch2 = c->on_stream2 != 0;
   if (ch2 && (!cm->channel2 || ncomp != 0)) { set_error("halo_sum: the second channel carries scalars only"); return LGH_ERR_ARG; }
   double *const sbuf = ch2 ? cm->sendbuf2 : cm->sendbuf;
   double *const rbuf = ch2 ? cm->recvbuf2 : cm->recvbuf;
   const ncclComm_t ncomm = ch2 ? cm->comm2 : cm->comm;
   if (packed && ch2) { set_error("halo_sum: pre-packed messages use the main channel"); return LGH_ERR_ARG; }
   KtScope sample(c, LGH_KERNEL_HALO); // (sampling on: one event pair around pack + exchange + combine, closed on error returns too)
   // ncomp == 0 (v unused): the messages carry the scalars only - a sum over the ranks as one
   // exchange with every peer (allreduce_dev uses it in all-pairs partitions)
   const long npack = std::max((long)tot * ncomp, (long)cm->n_nbr * nx);
   if (!packed)
   {
      hipLaunchKernelGGL(halo_pack_k, dim3(ceil_div(npack, 256)), dim3(256), 0, c->stream, tot,
                         ncomp, c->N, l_nodes, cm->pos, cm->cnt, v, sbuf, cm->n_nbr, nx, cm->d_base, cm->d_cnt,
                         extra);
      LGH_HIP_CHECK(hipGetLastError());
   }
   auto combine = [&]() {
      const long ncomb = std::max((long)cm->n_shared * ncomp, (long)nx);
      hipLaunchKernelGGL(halo_combine_k, dim3(ceil_div(ncomb, 256)), dim3(256), 0,
                         c->stream, cm->n_shared, ncomp, c->N, l_sh_node, cm->sh_off, cm->sh_src, cm->pos,
                         cm->cnt, rbuf, v, c->nranks, nx, cm->rank_src, cm->d_base, cm->d_cnt, extra);
   };
   if (cm->local)
   {
      LocalGroup *g = cm->local.get();
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream)); // my send buffer is packed
      if (!g->barrier()) { set_error("local communicator: barrier timed out (halo)"); return LGH_ERR_COMM; }
      for (int k = 0; k < cm->n_nbr; k++)
      {
         const Comm *pc = g->ctx[cm->nbr_rank[k]]->comm;
         int kk = -1;
         for (int j = 0; j < pc->n_nbr; j++) { if (pc->nbr_rank[j] == c->rank) { kk = j; } }
         if (kk < 0 || pc->nbr_count[kk] != cm->nbr_count[k])
         {
            set_error("local communicator: neighbour lists of ranks %d and %d do not match", c->rank, cm->nbr_rank[k]);
            return LGH_ERR_COMM;
         }
         LGH_HIP_CHECK(hipMemcpyAsync(rbuf + cm->nbr_base[k], (ch2 ? pc->sendbuf2 : pc->sendbuf) + pc->nbr_base[kk],
                                      ((size_t)ncomp * cm->nbr_count[k] + nx) * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      }
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      if (!g->barrier()) { set_error("local communicator: barrier timed out (halo)"); return LGH_ERR_COMM; } // peers may repack
      combine();
      LGH_HIP_CHECK(hipGetLastError());
      return LGH_OK;
   }
   if (cm->shm)
   {
      ShmGroup *g = cm->shm.get();
      const int ch = ch2 ? 1 : 0;
      if ((size_t)cm->bufsize * sizeof(double) > g->hdr()->cap) { set_error("shm transport: message of %zu bytes, room for %zu (LGH_SHM_MB)", (size_t)cm->bufsize * sizeof(double), g->hdr()->cap); return LGH_ERR_COMM; }
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream)); // my send buffer is packed
      LGH_HIP_CHECK(hipMemcpy(g->msg(c->rank, ch), sbuf, (size_t)cm->bufsize * sizeof(double), hipMemcpyDeviceToHost));
      if (!g->barrier()) { set_error("shm transport: barrier timed out (halo): a rank is missing or the ranks do not call the same exchanges"); return LGH_ERR_COMM; }
      cm->stage.resize((size_t)cm->bufsize);
      for (int k = 0; k < cm->n_nbr; k++)
      {
         const ShmTable *pt = g->table(cm->nbr_rank[k]);
         int kk = -1;
         for (int j = 0; j < pt->n_nbr; j++) { if (pt->nbr_rank[j] == c->rank) { kk = j; } }
         if (kk < 0 || pt->nbr_count[kk] != cm->nbr_count[k])
         {
            set_error("shm transport: neighbour lists of ranks %d and %d do not match", c->rank, cm->nbr_rank[k]);
            return LGH_ERR_COMM;
         }
         memcpy(cm->stage.data() + cm->nbr_base[k], g->msg(cm->nbr_rank[k], ch) + pt->nbr_base[kk], ((size_t)ncomp * cm->nbr_count[k] + nx) * sizeof(double));
      }
      LGH_HIP_CHECK(hipMemcpy(rbuf, cm->stage.data(), (size_t)cm->bufsize * sizeof(double), hipMemcpyHostToDevice));
      if (!g->barrier()) { set_error("shm transport: barrier timed out (halo)"); return LGH_ERR_COMM; } // peers may repack
      combine();
      LGH_HIP_CHECK(hipGetLastError());
      return LGH_OK;
   }
   LGH_NCCL_CHECK(g_nccl.GroupStart());
   for (int k = 0; k < cm->n_nbr; k++)
   {
      const size_t o = (size_t)cm->nbr_base[k];
      const size_t n = (size_t)ncomp * cm->nbr_count[k] + nx;
      LGH_NCCL_CHECK(g_nccl.Send(sbuf + o, n, ncclFloat64, cm->nbr_rank[k], ncomm, c->stream));
      LGH_NCCL_CHECK(g_nccl.Recv(rbuf + o, n, ncclFloat64, cm->nbr_rank[k], ncomm, c->stream));
   }
   LGH_NCCL_CHECK(g_nccl.GroupEnd());
   combine();
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

// may the energy solve run its reductions on the context's second stream?
bool comm_second_channel(const lgh_ctx *c) { return c->comm && c->comm->channel2; }

bool comm_pack_tables(const lgh_ctx *c, HaloPackTables *t)
{
   const Comm *cm = c->comm;
   if (!cm || cm->n_nbr == 0 || !cm->sh_off || !cm->sendbuf) { return false; }
   t->sh_off = cm->sh_off; t->sh_src = cm->sh_src; t->pos = cm->pos; t->cnt = cm->cnt;
   t->base = cm->d_base; t->ncnt = cm->d_cnt; t->sbuf = cm->sendbuf; t->n_nbr = cm->n_nbr;
   return true;
}

// the node lists of the exchanges (device, the caller's numbering): concatenated neighbour lists, unique shared nodes
void comm_node_lists(const lgh_ctx *c, const int **nodes, int *total, const int **sh_node, int *n_shared)
{
   const Comm *cm = c->comm;
   *nodes = cm ? cm->nodes : nullptr;
   *total = cm ? cm->total : 0;
   *sh_node = cm ? cm->sh_node : nullptr;
   *n_shared = (cm && cm->sh_node) ? cm->n_shared : 0;
}
// neighbours of lower rank, or -1 when the neighbour list is not in ascending rank order (a sum "own value between the
// lower and the higher peers' blocks" is then not the rank-ordered sum every rank must form alike)
int comm_ranks_before(const lgh_ctx *c)
{
   const Comm *cm = c->comm;
   int n = 0;
   if (cm)
   {
      for (int k = 0; k < cm->n_nbr; k++)
      {
         if (k > 0 && cm->nbr_rank[k] <= cm->nbr_rank[k - 1]) { return -1; }
         n += (cm->nbr_rank[k] < c->rank) ? 1 : 0;
      }
   }
   return n;
}
void comm_shared_nodes(const lgh_ctx *c, const uint8_t **hmask, const int **sh_node, int *n_shared)
{
   const Comm *cm = c->comm;
   *hmask = (cm && cm->hmask) ? cm->hmask : nullptr;
   *sh_node = cm ? cm->sh_node : nullptr;
   *n_shared = (cm && cm->hmask) ? cm->n_shared : 0;
}

int comm_word_peers(lgh_ctx *c, int nwords, const long long **peers, int *n_peers)
{
   Comm *cm = c->comm;
   *peers = nullptr;
   *n_peers = 0;
   if (!cm || cm->n_nbr == 0) { return LGH_OK; } // (a communicator of size 1: nobody to hear from)
   if (!cm->allpairs) { set_error("comm_word_peers: all-pairs partitions only"); return LGH_ERR_ARG; }
   if (cm->wordcap != nwords || cm->wordnbr != cm->n_nbr) // (sized by BOTH factors)
   {
      if (cm->wordbuf) { (void)hipFree(cm->wordbuf); cm->wordbuf = nullptr; }
      cm->wordcap = 0;
      LGH_HIP_CHECK(hipMalloc((void **)&cm->wordbuf, (size_t)cm->n_nbr * nwords * sizeof(long long)));
      LGH_HIP_CHECK(hipMemset(cm->wordbuf, 0, (size_t)cm->n_nbr * nwords * sizeof(long long)));
      cm->wordcap = nwords;
      cm->wordnbr = cm->n_nbr;
   }
   *peers = cm->wordbuf;
   *n_peers = cm->n_nbr;
   return LGH_OK;
}

int exchange_words(lgh_ctx *c, const long long *src, int nwords)
{
   Comm *cm = c->comm;
   if (!cm || cm->n_nbr == 0) { return LGH_OK; }
   if (!cm->allpairs || !cm->wordbuf || cm->wordcap != nwords || cm->wordnbr != cm->n_nbr || c->on_stream2) { set_error("exchange_words: all-pairs partition, main channel, peer buffer of %d words", nwords); return LGH_ERR_ARG; }
   KtScope sample(c, LGH_KERNEL_ALLREDUCE); // (counted with the small sums it replaces)
   const size_t bytes = (size_t)nwords * sizeof(long long);
   if (cm->local)
   {
      LocalGroup *g = cm->local.get();
      cm->word_src = src;
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream)); // my words are complete
      if (!g->barrier()) { set_error("local communicator: barrier timed out (words)"); return LGH_ERR_COMM; }
      for (int k = 0; k < cm->n_nbr; k++)
      {
         const Comm *pc = g->ctx[cm->nbr_rank[k]]->comm;
         if (!pc->word_src) { set_error("local communicator: rank %d exchanges no words", cm->nbr_rank[k]); return LGH_ERR_COMM; }
         LGH_HIP_CHECK(hipMemcpyAsync(cm->wordbuf + (size_t)k * nwords, pc->word_src, bytes, hipMemcpyDeviceToDevice, c->stream));
      }
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      if (!g->barrier()) { set_error("local communicator: barrier timed out (words)"); return LGH_ERR_COMM; } // peers may go on adding
      return LGH_OK;
   }
   if (cm->shm)
   {
      ShmGroup *g = cm->shm.get();
      if (bytes > g->hdr()->cap) { set_error("shm transport: %zu bytes of words, room for %zu", bytes, g->hdr()->cap); return LGH_ERR_COMM; }
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      LGH_HIP_CHECK(hipMemcpy(g->msg(c->rank, 0), src, bytes, hipMemcpyDeviceToHost));
      if (!g->barrier()) { set_error("shm transport: barrier timed out (words): a rank is missing or the ranks do not call the same exchanges"); return LGH_ERR_COMM; }
      cm->wstage.resize((size_t)cm->n_nbr * nwords);
      for (int k = 0; k < cm->n_nbr; k++) { memcpy(cm->wstage.data() + (size_t)k * nwords, g->msg(cm->nbr_rank[k], 0), bytes); }
      LGH_HIP_CHECK(hipMemcpy(cm->wordbuf, cm->wstage.data(), (size_t)cm->n_nbr * bytes, hipMemcpyHostToDevice));
      if (!g->barrier()) { set_error("shm transport: barrier timed out (words)"); return LGH_ERR_COMM; } // peers may reuse their message area
      return LGH_OK;
   }
   LGH_NCCL_CHECK(g_nccl.GroupStart());
   for (int k = 0; k < cm->n_nbr; k++)
   {
      // (send / recv move bytes: the words travel under the 8-byte type the halo messages already use)
      LGH_NCCL_CHECK(g_nccl.Send(src, (size_t)nwords, ncclFloat64, cm->nbr_rank[k], cm->comm, c->stream));
      LGH_NCCL_CHECK(g_nccl.Recv(cm->wordbuf + (size_t)k * nwords, (size_t)nwords, ncclFloat64, cm->nbr_rank[k], cm->comm, c->stream));
   }
   LGH_NCCL_CHECK(g_nccl.GroupEnd());
   return LGH_OK;
}

int allreduce_dev(lgh_ctx *c, double *dev, int count, int op, bool packed)
{
   Comm *cm = c->comm;
   // Sums of up to three scalars in an all-pairs partition (the CG dot products on <= 2x2x2 ranks):
   // one grouped exchange with every peer and a sum in rank order, the transport of the halo
   // messages, instead of a ring / tree all-reduce - one hop, bit-identical on every rank.
   if (op == 0 && count >= 1 && count <= 3 && cm && cm->n_nbr > 0 && halo_can_piggyback(c))
   {
      return halo_sum(c, nullptr, 0, dev, count, packed && !c->on_stream2);
   }
   if (packed) { set_error("allreduce_dev: pre-packed sums need the all-pairs exchange"); return LGH_ERR_ARG; }
   if (cm && cm->local)
   {
      LocalGroup *g = cm->local.get();
      if (count > 8) { set_error("local communicator: count > 8"); return LGH_ERR_ARG; }
      // `mine` / `res` are pageable: an async copy to or from pageable memory is not
      // ordered with the kernels of the stream (observed: stale sums, run-to-run different
      // results), so drain the stream first and copy synchronously
      double mine[8];
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      LGH_HIP_CHECK(hipMemcpy(mine, dev, count * sizeof(double), hipMemcpyDeviceToHost));
      for (int i = 0; i < count; i++) { g->slots[(size_t)c->rank * 8 + i] = mine[i]; }
      if (!g->barrier()) { set_error("local communicator: barrier timed out (all-reduce)"); return LGH_ERR_COMM; }
      double res[8];
      for (int i = 0; i < count; i++)
      {
         double r = g->slots[i];
         for (int k = 1; k < g->n; k++)
         {
            const double x = g->slots[(size_t)k * 8 + i];
            r = (op == 0) ? r + x : std::min(r, x); // rank order: identical on every rank
         }
         res[i] = r;
      }
      if (!g->barrier()) { set_error("local communicator: barrier timed out (all-reduce)"); return LGH_ERR_COMM; }
      LGH_HIP_CHECK(hipMemcpy(dev, res, count * sizeof(double), hipMemcpyHostToDevice));
      return LGH_OK;
   }
   if (cm && cm->shm)
   {
      ShmGroup *g = cm->shm.get();
      if (count > 8) { set_error("shm transport: count > 8"); return LGH_ERR_ARG; }
      double mine[8], res[8];
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      LGH_HIP_CHECK(hipMemcpy(mine, dev, count * sizeof(double), hipMemcpyDeviceToHost));
      for (int i = 0; i < count; i++) { g->slots(c->rank)[i] = mine[i]; }
      if (!g->barrier()) { set_error("shm transport: barrier timed out (all-reduce)"); return LGH_ERR_COMM; }
      for (int i = 0; i < count; i++)
      {
         double r = g->slots(0)[i];
         for (int k = 1; k < g->n; k++) { const double x = g->slots(k)[i]; r = (op == 0) ? r + x : std::min(r, x); } // rank order: identical on every rank
         res[i] = r;
      }
      if (!g->barrier()) { set_error("shm transport: barrier timed out (all-reduce)"); return LGH_ERR_COMM; }
      LGH_HIP_CHECK(hipMemcpy(dev, res, count * sizeof(double), hipMemcpyHostToDevice));
      return LGH_OK;
   }
   if (!cm || !cm->comm) { return LGH_OK; }
   if (c->on_stream2 && !cm->comm2) { set_error("all-reduce on the second stream without a second communicator"); return LGH_ERR_COMM; }
   kt_begin(c, LGH_KERNEL_ALLREDUCE);
   LGH_NCCL_CHECK(g_nccl.AllReduce(dev, dev, (size_t)count, ncclFloat64, op == 0 ? ncclSum : ncclMin,
                                   c->on_stream2 ? cm->comm2 : cm->comm, c->stream));
   kt_end(c, LGH_KERNEL_ALLREDUCE);
   return LGH_OK;
}

} // namespace lgh

using namespace lgh;

extern "C"
{

void lgh_comm_free(lgh_ctx *c)
{
   if (!c || !c->comm) { return; }
   Comm *cm = c->comm;
   if (cm->local)
   {
      std::lock_guard<std::mutex> lk(g_local_m);
      for (auto it = g_local.begin(); it != g_local.end();)
      {
         if (it->second == cm->local)
         {
            it->second->ctx[c->rank] = nullptr;
            bool any = false;
            for (lgh_ctx *p : it->second->ctx) { any = any || p; }
            it = any ? std::next(it) : g_local.erase(it);
         }
         else { ++it; }
      }
   }
   cm->shm.reset();
   if (cm->comm2 && g_nccl.CommDestroy) { g_nccl.CommDestroy(cm->comm2); }
   if (cm->comm && g_nccl.CommDestroy) { g_nccl.CommDestroy(cm->comm); }
   void *ptrs[] = {cm->nodes, cm->sendbuf, cm->recvbuf, cm->sendbuf2, cm->recvbuf2, cm->sh_node, cm->sh_off, cm->sh_src, cm->pos, cm->cnt, cm->hmask,
                   cm->d_base, cm->d_cnt, cm->rank_src, cm->wordbuf};
   for (void *p : ptrs) { if (p) { (void)hipFree(p); } }
   delete cm;
   c->comm = nullptr;
}

int lgh_comm_unique_id(char id_out[128])
{
   LGH_CHECK_ARG(id_out);
   int rc = load_nccl();
   if (rc) { return rc; }
   ncclUniqueId id;
   LGH_NCCL_CHECK(g_nccl.GetUniqueId(&id));
   memcpy(id_out, id.internal, 128);
   return LGH_OK;
}

int lgh_comm_unique_id_shm(char id_out[128])
{
   LGH_CHECK_ARG(id_out);
   memset(id_out, 0, 128);
   unsigned long long r = (unsigned long long)getpid() * 0x9E3779B97F4A7C15ull ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
   snprintf(id_out, 64, "LGHSHM_%d_%016llx", (int)getpid(), r);
   return LGH_OK;
}

int lgh_comm_init(lgh_ctx *c, int nranks, int rank, const char unique_id[128])
{
   LGH_CHECK_ARG(c && nranks >= 1 && rank >= 0 && rank < nranks && unique_id);
   if (memcmp(unique_id, "LGHSHM", 6) == 0) // cross-process loopback communicator (bench.py --transport shm, tests)
   {
      if (!c->comm) { c->comm = new Comm(); }
      int rc = shm_open_group(unique_id, nranks, rank, c->comm->shm);
      if (rc) { return rc; }
      // (the exchanges are synchronous on the host: the energy solve's sums may use the second channel's buffers safely)
      c->comm->channel2 = !(getenv("LGH_COMM2") && getenv("LGH_COMM2")[0] == '0');
      c->nranks = nranks;
      c->rank = rank;
      c->multi = 1;
      vcg_free(c);
      if (!c->comm->shm->barrier()) { set_error("shm transport: not all %d ranks arrived", nranks); return LGH_ERR_COMM; }
      return LGH_OK;
   }
   if (memcmp(unique_id, "LGHLOCAL", 8) == 0) // in-process loopback communicator (tests)
   {
      if (!c->comm) { c->comm = new Comm(); }
      std::shared_ptr<LocalGroup> g;
      {
         std::lock_guard<std::mutex> lk(g_local_m);
         const std::string key(unique_id, 128);
         auto &slot = g_local[key];
         if (!slot)
         {
            slot = std::make_shared<LocalGroup>();
            slot->n = nranks;
            slot->ctx.assign(nranks, nullptr);
            slot->slots.assign((size_t)nranks * 8, 0.0);
         }
         if (slot->n != nranks || slot->ctx[rank]) { set_error("local communicator: rank %d registered twice", rank); return LGH_ERR_ARG; }
         slot->ctx[rank] = c;
         g = slot;
      }
      c->comm->local = g;
      c->comm->channel2 = !(getenv("LGH_COMM2") && getenv("LGH_COMM2")[0] == '0');
      c->nranks = nranks;
      c->rank = rank;
      c->multi = 1;
      vcg_free(c); // (tables of K2 built for another communicator)
      if (!g->barrier()) { set_error("local communicator: not all %d ranks arrived", nranks); return LGH_ERR_COMM; }
      return LGH_OK;
   }
   int rc = load_nccl();
   if (rc) { return rc; }
   LGH_HIP_CHECK(hipSetDevice(c->device));
   if (!c->comm) { c->comm = new Comm(); }
   ncclUniqueId id;
   memcpy(id.internal, unique_id, 128);
   LGH_NCCL_CHECK(g_nccl.CommInitRank(&c->comm->comm, nranks, id, rank));
   c->nranks = nranks;
   c->rank = rank;
   {
      const char *env = getenv("LGH_FORCE_MULTI");
      c->multi = (nranks > 1 || (env && env[0] == '1')) ? 1 : 0;
   }
   // Second communicator (same ranks) for the second stream, so that the energy solve can run beside the velocity solve
   // on several ranks too.  Its id is drawn by rank 0 and travels over the first one TOGETHER with rank 0's verdict (a
   // rank that got no id must not leave the others blocked in CommInitRank); the outcome is then agreed on by a MIN over
   // the ranks, so either every rank has the channel or none does.
   // Two communicators driven from two streams of one device have only ever run on a communicator of size 1 and on the
   // in-process loopback here (no multi-GPU box): collectives of different communicators that the ranks happen to
   // schedule in different orders can block each other.  Until a real multi-GPU run has validated it, the channel is
   // therefore opt-in on more than one rank: LGH_COMM2=1 on every rank (LGH_COMM2=0 switches it off everywhere).
   const char *env2 = getenv("LGH_COMM2");
   const bool want2 = env2 ? env2[0] == '1' : (nranks == 1);
   if (c->multi && g_nccl.Broadcast && want2)
   {
      Comm *cm = c->comm;
      char *dbuf = nullptr;
      LGH_HIP_CHECK(hipMalloc((void **)&dbuf, 128 + 2 * sizeof(double)));
      char hbuf[128 + sizeof(double)];
      ncclUniqueId id2;
      memset(&id2, 0, sizeof(id2));
      double ok = 1.0;
      if (rank == 0 && g_nccl.GetUniqueId(&id2) != ncclSuccess) { ok = 0.0; }
      memcpy(hbuf, id2.internal, 128);
      memcpy(hbuf + 128, &ok, sizeof(double));
      LGH_HIP_CHECK(hipMemcpy(dbuf, hbuf, sizeof(hbuf), hipMemcpyHostToDevice));
      LGH_NCCL_CHECK(g_nccl.Broadcast(dbuf, dbuf, sizeof(hbuf), /* ncclChar */ 0, 0, cm->comm, c->stream));
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      LGH_HIP_CHECK(hipMemcpy(hbuf, dbuf, sizeof(hbuf), hipMemcpyDeviceToHost));
      memcpy(id2.internal, hbuf, 128);
      memcpy(&ok, hbuf + 128, sizeof(double)); // rank 0's verdict: nobody calls CommInitRank when it is 0
      if (ok != 0.0 && g_nccl.CommInitRank(&cm->comm2, nranks, id2, rank) != ncclSuccess) { ok = 0.0; cm->comm2 = nullptr; }
      double *dok = (double *)(dbuf + 128 + sizeof(double));
      LGH_HIP_CHECK(hipMemcpy(dok, &ok, sizeof(double), hipMemcpyHostToDevice));
      LGH_NCCL_CHECK(g_nccl.AllReduce(dok, dok, 1, ncclFloat64, ncclMin, cm->comm, c->stream));
      LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
      LGH_HIP_CHECK(hipMemcpy(&ok, dok, sizeof(double), hipMemcpyDeviceToHost));
      (void)hipFree(dbuf);
      cm->channel2 = (ok != 0.0);
      if (!cm->channel2 && cm->comm2) { g_nccl.CommDestroy(cm->comm2); cm->comm2 = nullptr; }
   }
   vcg_free(c); // the tables of the node kernel K2 carry the shared-node / owner flags of the previous communicator
   return LGH_OK;
}

int lgh_comm_stats(lgh_ctx *c, int *n_neighbours, long *max_nodes_per_neighbour, long *shared_nodes, int *all_pairs, int *second_channel)
{
   LGH_CHECK_ARG(c && n_neighbours && max_nodes_per_neighbour && shared_nodes && all_pairs && second_channel);
   const Comm *cm = c->comm;
   *n_neighbours = cm ? cm->n_nbr : 0;
   long mx = 0;
   if (cm) { for (int k = 0; k < cm->n_nbr; k++) { mx = std::max(mx, (long)cm->nbr_count[k]); } }
   *max_nodes_per_neighbour = mx;
   *shared_nodes = cm ? cm->n_shared : 0;
   *all_pairs = (cm && cm->allpairs) ? 1 : 0;
   *second_channel = (cm && cm->channel2) ? 1 : 0;
   return LGH_OK;
}

int lgh_comm_set_neighbors(lgh_ctx *c, int n_nbr, const int *nbr_rank, const int *nbr_count,
                           const int *const *nbr_nodes)
{
   LGH_CHECK_ARG(c && n_nbr >= 0);
   if (!c->comm) { c->comm = new Comm(); }
   Comm *cm = c->comm;
   std::vector<int> rs((size_t)std::max(c->nranks, 1), -2);
   // Everything that can fail locally (argument checks, allocations) runs first and its status is folded
   // into the collective below: a rank that fails must not leave its peers blocked in the all-reduce.
   const int rc_local = [&]() -> int {
   LGH_CHECK_ARG(n_nbr == 0 || (nbr_rank && nbr_count && nbr_nodes));
   if (n_nbr == 0 && c->nranks > 1)
   {
      // every exchange is a rendezvous of all ranks (process-wide barriers on the loopback transports): a rank of a
      // connected partition always has a neighbour, and one without would return from halo_sum before the others arrive
      set_error("lgh_comm_set_neighbors: rank %d of %d has no neighbour (disconnected partition)", c->rank, c->nranks);
      return LGH_ERR_ARG;
   }
   for (int k = 0; k < n_nbr; k++)
   {
      LGH_CHECK_ARG(nbr_count[k] >= 0 && (nbr_count[k] == 0 || nbr_nodes[k]));
      for (int i = 0; i < nbr_count[k]; i++)
      {
         if (nbr_nodes[k][i] < 0 || nbr_nodes[k][i] >= c->N) { set_error("neighbour node out of range"); return LGH_ERR_ARG; }
      }
   }
   // (the peer buffer of exchange_words is sized by the neighbour count: round-5 advisor - a second call with more
   //  neighbours would otherwise leave the exchange writing behind a buffer of the old size)
   if (cm->wordbuf) { (void)hipFree(cm->wordbuf); cm->wordbuf = nullptr; }
   cm->wordcap = 0;
   cm->word_src = nullptr;
   cm->n_nbr = n_nbr;
   cm->nbr_rank.assign(nbr_rank, nbr_rank + n_nbr);
   cm->nbr_count.assign(nbr_count, nbr_count + n_nbr);
   cm->nbr_off.resize(n_nbr);
   cm->nbr_base.resize(n_nbr);
   std::vector<int> all;
   int tot = 0;
   for (int k = 0; k < n_nbr; k++)
   {
      cm->nbr_off[k] = tot;
      cm->nbr_base[k] = 3 * tot + kHaloSlack * k;
      for (int i = 0; i < nbr_count[k]; i++)
      {
         const int n = nbr_nodes[k][i];
         if (n < 0 || n >= c->N) { set_error("neighbour node out of range"); return LGH_ERR_ARG; }
         all.push_back(n);
      }
      tot += nbr_count[k];
   }
   cm->total = tot;
   // canonical-order combine lists (requires the rank: call lgh_comm_init first)
   {
      std::vector<std::vector<std::pair<int, int>>> per_node; // (rank, recv index)
      std::vector<int> uniq;
      std::vector<int> slot((size_t)c->N, -1);
      for (int k = 0; k < n_nbr; k++)
      {
         for (int i = 0; i < nbr_count[k]; i++)
         {
            const int n = nbr_nodes[k][i];
            if (slot[n] < 0)
            {
               slot[n] = (int)uniq.size();
               uniq.push_back(n);
               per_node.emplace_back();
               per_node.back().push_back({c->rank, -1});
            }
            per_node[slot[n]].push_back({nbr_rank[k], cm->nbr_off[k] + i});
         }
      }
      std::vector<int> off(uniq.size() + 1, 0), src;
      for (size_t u = 0; u < uniq.size(); u++)
      {
         std::sort(per_node[u].begin(), per_node[u].end());
         for (auto &pr : per_node[u]) { src.push_back(pr.second); }
         off[u + 1] = (int)src.size();
      }
      void *old[] = {cm->sh_node, cm->sh_off, cm->sh_src};
      for (void *p : old) { if (p) { (void)hipFree(p); } }
      cm->n_shared = (int)uniq.size();
      LGH_HIP_CHECK(hipMalloc((void **)&cm->sh_node, std::max<size_t>(uniq.size(), 1) * sizeof(int)));
      LGH_HIP_CHECK(hipMalloc((void **)&cm->sh_off, off.size() * sizeof(int)));
      LGH_HIP_CHECK(hipMalloc((void **)&cm->sh_src, std::max<size_t>(src.size(), 1) * sizeof(int)));
      if (!uniq.empty()) { LGH_HIP_CHECK(hipMemcpy(cm->sh_node, uniq.data(), uniq.size() * sizeof(int), hipMemcpyHostToDevice)); }
      LGH_HIP_CHECK(hipMemcpy(cm->sh_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
      if (!src.empty()) { LGH_HIP_CHECK(hipMemcpy(cm->sh_src, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice)); }
      std::vector<uint8_t> hm((size_t)c->N, 0);
      for (int n : uniq) { hm[n] = 1; }
      if (cm->hmask) { (void)hipFree(cm->hmask); }
      vcg_free(c); // (the flag bytes of the node kernel K2 were built from the previous neighbour lists)
      LGH_HIP_CHECK(hipMalloc((void **)&cm->hmask, std::max<size_t>(hm.size(), 1)));
      LGH_HIP_CHECK(hipMemcpy(cm->hmask, hm.data(), hm.size(), hipMemcpyHostToDevice));
   }
   if (cm->nodes) { (void)hipFree(cm->nodes); (void)hipFree(cm->sendbuf); (void)hipFree(cm->recvbuf); (void)hipFree(cm->pos); (void)hipFree(cm->cnt); }
   {
      std::vector<int> pos(std::max(tot, 1)), cnt(std::max(tot, 1));
      for (int k = 0; k < n_nbr; k++)
         for (int i = 0; i < nbr_count[k]; i++)
         {
            pos[cm->nbr_off[k] + i] = cm->nbr_base[k] + i;
            cnt[cm->nbr_off[k] + i] = nbr_count[k];
         }
      LGH_HIP_CHECK(hipMalloc((void **)&cm->pos, pos.size() * sizeof(int)));
      LGH_HIP_CHECK(hipMalloc((void **)&cm->cnt, cnt.size() * sizeof(int)));
      LGH_HIP_CHECK(hipMemcpy(cm->pos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice));
      LGH_HIP_CHECK(hipMemcpy(cm->cnt, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice));
   }
   LGH_HIP_CHECK(hipMalloc((void **)&cm->nodes, std::max<size_t>(tot, 1) * sizeof(int)));
   if (tot) { LGH_HIP_CHECK(hipMemcpy(cm->nodes, all.data(), (size_t)tot * sizeof(int), hipMemcpyHostToDevice)); }
   cm->bufsize = 3 * std::max(tot, 1) + kHaloSlack * n_nbr;
   LGH_HIP_CHECK(hipMalloc((void **)&cm->sendbuf, (size_t)cm->bufsize * sizeof(double)));
   LGH_HIP_CHECK(hipMalloc((void **)&cm->recvbuf, (size_t)cm->bufsize * sizeof(double)));
   LGH_HIP_CHECK(hipMemset(cm->sendbuf, 0, (size_t)cm->bufsize * sizeof(double)));
   LGH_HIP_CHECK(hipMemset(cm->recvbuf, 0, (size_t)cm->bufsize * sizeof(double)));
   if (cm->channel2)
   {
      (void)hipFree(cm->sendbuf2);
      (void)hipFree(cm->recvbuf2);
      LGH_HIP_CHECK(hipMalloc((void **)&cm->sendbuf2, (size_t)cm->bufsize * sizeof(double)));
      LGH_HIP_CHECK(hipMalloc((void **)&cm->recvbuf2, (size_t)cm->bufsize * sizeof(double)));
      LGH_HIP_CHECK(hipMemset(cm->sendbuf2, 0, (size_t)cm->bufsize * sizeof(double)));
      LGH_HIP_CHECK(hipMemset(cm->recvbuf2, 0, (size_t)cm->bufsize * sizeof(double)));
   }
   LGH_HIP_CHECK(hipStreamSynchronize(nullptr)); // the fills run asynchronously on the null stream
   {
      // per-neighbour tables and the rank -> neighbour map of the piggy-backed scalars
      void *old[] = {cm->d_base, cm->d_cnt, cm->rank_src};
      for (void *p : old) { if (p) { (void)hipFree(p); } }
      cm->d_base = cm->d_cnt = cm->rank_src = nullptr;
      rs[c->rank] = -1;
      for (int k = 0; k < n_nbr; k++)
      {
         if (nbr_rank[k] >= 0 && nbr_rank[k] < c->nranks) { rs[nbr_rank[k]] = k; }
      }
      cm->allpairs = (n_nbr > 0 && n_nbr == c->nranks - 1);
      for (int r : rs) { if (r == -2) { cm->allpairs = false; } }
   }
   return LGH_OK;
   }();
   // the message sizes depend on it: every rank must come to the same decision (in a 3x1x1 partition
   // only the middle rank sees all others); a local failure travels in the same MIN-reduction
   if (cm->shm)
   {
      if (n_nbr > kShmMaxNbr) { set_error("shm transport: more than %d neighbours", kShmMaxNbr); return LGH_ERR_ARG; }
      ShmTable *t = cm->shm->table(c->rank);
      t->n_nbr = rc_local ? 0 : n_nbr;
      for (int k = 0; k < t->n_nbr; k++) { t->nbr_rank[k] = cm->nbr_rank[k]; t->nbr_base[k] = cm->nbr_base[k]; t->nbr_count[k] = cm->nbr_count[k]; }
   }
   if (c->multi != 0 && (cm->comm || cm->local || cm->shm))
   {
      double flag = rc_local ? -1.0 : (cm->allpairs ? 1.0 : 0.0);
      const int rc = lgh_allreduce(c, &flag, 1);
      if (rc) { return rc; }
      if (flag < 0.0)
      {
         if (!rc_local) { set_error("lgh_comm_set_neighbors failed on another rank"); }
         return rc_local ? rc_local : LGH_ERR_COMM;
      }
      cm->allpairs = (flag > 0.5);
   }
   else if (rc_local) { return rc_local; }
   {
      LGH_HIP_CHECK(hipMalloc((void **)&cm->d_base, std::max<size_t>(n_nbr, 1) * sizeof(int)));
      LGH_HIP_CHECK(hipMalloc((void **)&cm->d_cnt, std::max<size_t>(n_nbr, 1) * sizeof(int)));
      LGH_HIP_CHECK(hipMalloc((void **)&cm->rank_src, rs.size() * sizeof(int)));
      if (n_nbr)
      {
         LGH_HIP_CHECK(hipMemcpy(cm->d_base, cm->nbr_base.data(), n_nbr * sizeof(int), hipMemcpyHostToDevice));
         LGH_HIP_CHECK(hipMemcpy(cm->d_cnt, cm->nbr_count.data(), n_nbr * sizeof(int), hipMemcpyHostToDevice));
      }
      LGH_HIP_CHECK(hipMemcpy(cm->rank_src, rs.data(), rs.size() * sizeof(int), hipMemcpyHostToDevice));
   }
   return LGH_OK;
}

int lgh_test_set_rank(lgh_ctx *c, int nranks, int rank)
{
   LGH_CHECK_ARG(c && nranks >= 1 && rank >= 0 && rank < nranks);
   c->nranks = nranks;
   c->rank = rank;
   return LGH_OK;
}
int lgh_test_word_peers(lgh_ctx *c, int nwords, long *capacity_words)
{
   LGH_CHECK_ARG(c && nwords > 0 && capacity_words);
   const long long *peers = nullptr;
   int n_peers = 0;
   const int rc = comm_word_peers(c, nwords, &peers, &n_peers);
   if (rc) { return rc; }
   const Comm *cm = c->comm;
   *capacity_words = (cm && cm->wordbuf) ? (long)cm->wordnbr * cm->wordcap : 0;
   if (peers) { LGH_HIP_CHECK(hipMemset((void *)peers, 0, (size_t)n_peers * nwords * sizeof(long long))); } // (writes the whole buffer: a short one faults here)
   return (cm && cm->wordbuf && cm->wordnbr == n_peers && cm->wordcap == nwords) || n_peers == 0 ? LGH_OK : LGH_ERR_COMM;
}
int lgh_test_halo_pack(lgh_ctx *c, const double *v, int ncomp, double *out)
{
   LGH_CHECK_ARG(c && v && out && c->comm && ncomp >= 1 && ncomp <= 3);
   Comm *cm = c->comm;
   if (cm->total == 0) { return LGH_OK; }
   hipLaunchKernelGGL(halo_pack_k, dim3(ceil_div((long)cm->total * ncomp, 256)), dim3(256), 0, c->stream,
                      cm->total, ncomp, c->N, cm->nodes, cm->pos, cm->cnt, v, cm->sendbuf, 0, 0, nullptr, nullptr, nullptr);
   LGH_HIP_CHECK(hipGetLastError());
   LGH_HIP_CHECK(hipMemcpyAsync(out, cm->sendbuf, (size_t)cm->bufsize * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
   return LGH_OK;
}
int lgh_test_halo_combine(lgh_ctx *c, const double *in, double *v, int ncomp)
{
   LGH_CHECK_ARG(c && v && in && c->comm && ncomp >= 1 && ncomp <= 3);
   Comm *cm = c->comm;
   if (cm->total == 0) { return LGH_OK; }
   LGH_HIP_CHECK(hipMemcpyAsync(cm->recvbuf, in, (size_t)cm->bufsize * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
   hipLaunchKernelGGL(halo_combine_k, dim3(ceil_div((long)cm->n_shared * ncomp, 256)), dim3(256), 0, c->stream,
                      cm->n_shared, ncomp, c->N, cm->sh_node, cm->sh_off, cm->sh_src, cm->pos, cm->cnt,
                      cm->recvbuf, v, c->nranks, 0, nullptr, nullptr, nullptr, nullptr);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

// Probe of the transport used by halo_sum on the REAL communicator (not the loopback one):
// a grouped ncclSend / ncclRecv of n doubles from this rank to itself, the only
// peer a one-GPU box offers.  *max_abs_diff = max |received - sent|.
int lgh_test_rccl_self_sendrecv(lgh_ctx *c, int n, double *max_abs_diff)
{
   LGH_CHECK_ARG(c && c->comm && c->comm->comm && !c->comm->local && n > 0 && max_abs_diff);
   double *buf = nullptr;
   LGH_HIP_CHECK(hipMalloc((void **)&buf, 2 * (size_t)n * sizeof(double)));
   std::vector<double> h(2 * (size_t)n, 0.0);
   for (int i = 0; i < n; i++) { h[i] = 1.0 + 0.5 * i; }
   LGH_HIP_CHECK(hipMemcpy(buf, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
   int rc = LGH_OK;
   if (g_nccl.GroupStart() != ncclSuccess || g_nccl.Send(buf, (size_t)n, ncclFloat64, c->rank, c->comm->comm, c->stream) != ncclSuccess ||
       g_nccl.Recv(buf + n, (size_t)n, ncclFloat64, c->rank, c->comm->comm, c->stream) != ncclSuccess ||
       g_nccl.GroupEnd() != ncclSuccess)
   {
      set_error("RCCL self send/recv failed");
      rc = LGH_ERR_COMM;
   }
   if (rc == LGH_OK && hipStreamSynchronize(c->stream) != hipSuccess) { rc = LGH_ERR_HIP; }
   if (rc == LGH_OK && hipMemcpy(h.data(), buf, h.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { rc = LGH_ERR_HIP; }
   (void)hipFree(buf);
   if (rc) { return rc; }
   double d = 0.0;
   for (int i = 0; i < n; i++) { d = std::max(d, std::fabs(h[(size_t)n + i] - h[i])); }
   *max_abs_diff = d;
   return LGH_OK;
}

int lgh_groups_to_neighbors(int my_rank, int N, int n_groups, const int *group_off, const int *group_ranks,
                            const int *group_master, const int *ldof_off, const int *ldofs, double *owner,
                            int *n_nbr, int *nbr_rank, int *nbr_count, int cap_nbr, int *nbr_nodes, long cap_nodes)
{
   LGH_CHECK_ARG(my_rank >= 0 && N >= 0 && n_groups >= 0 && owner && n_nbr);
   LGH_CHECK_ARG(n_groups == 0 || (group_off && group_ranks && ldof_off && ldofs));
   for (int i = 0; i < N; i++) { owner[i] = 1.0; }
   // groups ordered by their sorted rank sets: the same sequence on every member
   std::vector<std::vector<int>> sets((size_t)n_groups);
   std::vector<int> order((size_t)n_groups);
   for (int g = 0; g < n_groups; g++)
   {
      LGH_CHECK_ARG(group_off[g + 1] - group_off[g] >= 2 && ldof_off[g + 1] >= ldof_off[g]);
      sets[g].assign(group_ranks + group_off[g], group_ranks + group_off[g + 1]);
      std::sort(sets[g].begin(), sets[g].end());
      if (std::adjacent_find(sets[g].begin(), sets[g].end()) != sets[g].end() ||
          !std::binary_search(sets[g].begin(), sets[g].end(), my_rank) || sets[g].front() < 0)
      {
         set_error("lgh_groups_to_neighbors: group %d must list distinct ranks including this one", g);
         return LGH_ERR_ARG;
      }
      order[g] = g;
      const int master = group_master ? group_master[g] : sets[g].front();
      if (!std::binary_search(sets[g].begin(), sets[g].end(), master)) { set_error("lgh_groups_to_neighbors: master of group %d is not a member", g); return LGH_ERR_ARG; }
      for (int k = ldof_off[g]; k < ldof_off[g + 1]; k++)
      {
         if (ldofs[k] < 0 || ldofs[k] >= N) { set_error("lgh_groups_to_neighbors: dof out of range in group %d", g); return LGH_ERR_ARG; }
         if (master != my_rank) { owner[ldofs[k]] = 0.0; }
      }
   }
   std::sort(order.begin(), order.end(), [&](int a, int b) { return sets[a] < sets[b]; });
   for (int i = 0; i + 1 < n_groups; i++)
   {
      if (sets[order[i]] == sets[order[i + 1]]) { set_error("lgh_groups_to_neighbors: two groups with the same rank set"); return LGH_ERR_ARG; }
   }
   std::vector<int> peers;
   for (int g = 0; g < n_groups; g++) { for (int r : sets[g]) { if (r != my_rank) { peers.push_back(r); } } }
   std::sort(peers.begin(), peers.end());
   peers.erase(std::unique(peers.begin(), peers.end()), peers.end());
   if ((int)peers.size() > cap_nbr) { set_error("lgh_groups_to_neighbors: %d peers, room for %d", (int)peers.size(), cap_nbr); return LGH_ERR_ARG; }
   LGH_CHECK_ARG(peers.empty() || (nbr_rank && nbr_count && nbr_nodes));
   long pos = 0;
   for (size_t k = 0; k < peers.size(); k++)
   {
      nbr_rank[k] = peers[k];
      int cnt = 0;
      for (int i = 0; i < n_groups; i++)
      {
         const int g = order[i];
         if (!std::binary_search(sets[g].begin(), sets[g].end(), peers[k])) { continue; }
         const int nd = ldof_off[g + 1] - ldof_off[g];
         if (pos + nd > cap_nodes) { set_error("lgh_groups_to_neighbors: node list capacity %ld too small", cap_nodes); return LGH_ERR_ARG; }
         for (int j = 0; j < nd; j++) { nbr_nodes[pos + j] = ldofs[ldof_off[g] + j]; }
         pos += nd;
         cnt += nd;
      }
      nbr_count[k] = cnt;
   }
   *n_nbr = (int)peers.size();
   return LGH_OK;
}

int lgh_halo_sum(lgh_ctx *c, double *v_h1, int ncomp)
{
   LGH_CHECK_ARG(c && v_h1 && ncomp >= 1 && ncomp <= 3);
   return halo_sum(c, v_h1, ncomp, nullptr, 0);
}

int lgh_allreduce(lgh_ctx *c, double *value, int op)
{
   LGH_CHECK_ARG(c && value);
   if (!c->multi || !c->comm) { return LGH_OK; }
   // through the pinned staging area: an async copy from pageable memory is not ordered
   // with the stream
   c->host_pinned[3] = *value;
   LGH_HIP_CHECK(hipMemcpyAsync(c->scal + 2, c->host_pinned + 3, sizeof(double), hipMemcpyHostToDevice, c->stream));
   int rc = allreduce_dev(c, c->scal + 2, 1, op);
   if (rc) { return rc; }
   LGH_HIP_CHECK(hipMemcpyAsync(c->host_pinned + 2, c->scal + 2, sizeof(double), hipMemcpyDeviceToHost, c->stream));
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   *value = c->host_pinned[2];
   return LGH_OK;
}

} // extern "C"
