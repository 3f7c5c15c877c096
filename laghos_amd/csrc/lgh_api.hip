// lgh_api.hip — C ABI of liblaghos_hip.so: context life cycle, vector helpers,
// timers, and the operator-level entry points declared in include/laghos_hip.h.
#include <cstdarg>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <limits>

#include "lgh_common.hpp"

#include <chrono>
#include <dlfcn.h>

namespace lgh
{

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(g_err, sizeof(g_err), fmt, ap);
   va_end(ap);
}

// ---- small vector kernels ------------------------------------------------------
__global__ void __launch_bounds__(256) vec_set_k(double *y, double a, long n)
{
   for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { y[i] = a; }
}
__global__ void __launch_bounds__(256)
vec_axpby_k(double *z, double a, const double *x, double b, const double *y, long n)
{
   for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
   {
      z[i] = fma(a, x[i], __dmul_rn(b, y[i])); // (one form in both kernels: same bits whatever the alignment)
   }
}
// the same with 16-byte accesses and two of them in flight per thread (all three pointers 16-byte aligned; the
// scalar kernel takes the odd tail): 3.6-3.9 -> ~5 TB/s on the RK stage updates
__global__ void __launch_bounds__(256)
vec_axpby2_k(double2 *z, double a, const double2 *x, double b, const double2 *y, long n2)
{
   const long stride = (long)gridDim.x * blockDim.x;
   long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
   for (; i + stride < n2; i += 2 * stride)
   {
      const double2 x0 = x[i], y0 = y[i], x1 = x[i + stride], y1 = y[i + stride];
      z[i] = make_double2(fma(a, x0.x, __dmul_rn(b, y0.x)), fma(a, x0.y, __dmul_rn(b, y0.y)));
      z[i + stride] = make_double2(fma(a, x1.x, __dmul_rn(b, y1.x)), fma(a, x1.y, __dmul_rn(b, y1.y)));
   }
   if (i < n2)
   {
      const double2 x0 = x[i], y0 = y[i];
      z[i] = make_double2(fma(a, x0.x, __dmul_rn(b, y0.x)), fma(a, x0.y, __dmul_rn(b, y0.y)));
   }
}
// Two RK stage combinations that share their increment y in one pass (round 5): z1 = a1 x1 + b1 y, z2 = a2 x2 + b2 y with the
// expressions of vec_axpby2_k - the same bits as two calls; z2 may alias x2 (z += b k), z1 and z2 must not overlap.
__global__ void __launch_bounds__(256)
vec_axpby_pair_k(double2 *z1, double a1, const double2 *x1, double b1, double2 *z2, double a2, const double2 *x2, double b2, const double2 *y, long n2)
{
   const long stride = (long)gridDim.x * blockDim.x;
   for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n2; i += stride)
   {
      const double2 y0 = y[i], p = x1[i], q = x2[i];
      z1[i] = make_double2(fma(a1, p.x, __dmul_rn(b1, y0.x)), fma(a1, p.y, __dmul_rn(b1, y0.y)));
      z2[i] = make_double2(fma(a2, q.x, __dmul_rn(b2, y0.x)), fma(a2, q.y, __dmul_rn(b2, y0.y)));
   }
}
__global__ void __launch_bounds__(256) vec_neg_k(double *y, long n)
{
   for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { y[i] = -y[i]; }
}
__global__ void __launch_bounds__(256) vec_zero_list_k(double *y, const int *list, int n)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { y[list[i]] = 0.0; }
}
__global__ void __launch_bounds__(256)
vec_dot_k(const double *x, const double *y, const double *w, long n, double *partials,
          unsigned int *ticket, double *out)
{
   __shared__ double red[16];
   double s = 0.0;
   for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
   {
      s += (w ? w[i] : 1.0) * x[i] * y[i];
   }
   const double bsum = block_sum(s, red);
   double total;
   if (grid_sum_last_block(bsum, partials, ticket, red, total))
   {
      if (threadIdx.x == 0) { *out = total; }
   }
}
__global__ void set_double_k(double *p, double v) { *p = v; }
// qdata.dt_est = v: the estimate itself and, behind it, the partial minima the row-form update folds its candidates into
__global__ void __launch_bounds__(256) dt_est_set_k(double *p, double v)
{
   if (threadIdx.x == 0) { p[0] = v; }
   for (int s = threadIdx.x; s < kDtSlots; s += blockDim.x) { p[kDtSlotStride * (1 + s)] = __builtin_inf(); }
}
// ... and the fold in front of every read: estimate = min(estimate, partial minima) in a fixed tree (a minimum does not
// depend on the order anyway); the slots go back to +inf
// host_out (pinned host memory as the device sees it): [0] the estimate, [1] the error word `err` - the host reads them
// after the stream synchronisation it needs anyway, without two copy kernels in front of it
__global__ void __launch_bounds__(256) dt_est_fold_k(double *p, const int *err, double *host_out, const unsigned long long token)
{
   __shared__ double red[16];
   double m = __builtin_inf();
   for (int s = threadIdx.x; s < kDtSlots; s += blockDim.x)
   {
      m = fmin(m, p[kDtSlotStride * (1 + s)]);
      p[kDtSlotStride * (1 + s)] = __builtin_inf();
   }
   m = block_min(m, red);
   if (threadIdx.x == 0)
   {
      const double est = fmin(p[0], m);
      p[0] = est;
      host_out[0] = est;
      ((int *)(host_out + 1))[0] = *err;
      __threadfence_system();
      __hip_atomic_store((unsigned long long *)(host_out + 2), token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); // (host_wait_token)
   }
}

static int grid_for(long n) { return (int)std::min<long>(std::max<long>((n + 255) / 256, 1), 2048); }

int vec_set(lgh_ctx *c, double *y, double a, long n)
{
   if (n <= 0) { return LGH_OK; }
   hipLaunchKernelGGL(vec_set_k, dim3(grid_for(n)), dim3(256), 0, c->stream, y, a, n);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int vec_axpby(lgh_ctx *c, double *z, double a, const double *x, double b, const double *y, long n)
{
   if (n <= 0) { return LGH_OK; }
   if (n >= 4096 && (((uintptr_t)z | (uintptr_t)x | (uintptr_t)y) & 15u) == 0)
   {
      const long n2 = n / 2;
      hipLaunchKernelGGL(vec_axpby2_k, dim3(grid_for(n2)), dim3(256), 0, c->stream, (double2 *)z, a, (const double2 *)x, b,
                         (const double2 *)y, n2);
      if (n & 1) { hipLaunchKernelGGL(vec_axpby_k, dim3(1), dim3(64), 0, c->stream, z + 2 * n2, a, x + 2 * n2, b, y + 2 * n2, 1L); }
   }
   else { hipLaunchKernelGGL(vec_axpby_k, dim3(grid_for(n)), dim3(256), 0, c->stream, z, a, x, b, y, n); }
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int vec_axpby_pair(lgh_ctx *c, double *z1, double a1, const double *x1, double b1, double *z2, double a2, const double *x2, double b2,
                   const double *y, long n)
{
   if (n <= 0) { return LGH_OK; }
   if (n >= 4096 && (((uintptr_t)z1 | (uintptr_t)x1 | (uintptr_t)z2 | (uintptr_t)x2 | (uintptr_t)y) & 15u) == 0)
   {
      const long n2 = n / 2;
      hipLaunchKernelGGL(vec_axpby_pair_k, dim3(grid_for(n2)), dim3(256), 0, c->stream, (double2 *)z1, a1, (const double2 *)x1, b1, (double2 *)z2, a2,
                         (const double2 *)x2, b2, (const double2 *)y, n2);
      LGH_HIP_CHECK(hipGetLastError());
      if (n & 1)
      {
         hipLaunchKernelGGL(vec_axpby_k, dim3(1), dim3(64), 0, c->stream, z1 + 2 * n2, a1, x1 + 2 * n2, b1, y + 2 * n2, 1L);
         hipLaunchKernelGGL(vec_axpby_k, dim3(1), dim3(64), 0, c->stream, z2 + 2 * n2, a2, x2 + 2 * n2, b2, y + 2 * n2, 1L);
      }
      LGH_HIP_CHECK(hipGetLastError());
      return LGH_OK;
   }
   int rc = vec_axpby(c, z1, a1, x1, b1, y, n);
   if (rc) { return rc; }
   return vec_axpby(c, z2, a2, x2, b2, y, n);
}
int vec_neg_inplace(lgh_ctx *c, double *y, long n)
{
   if (n <= 0) { return LGH_OK; }
   hipLaunchKernelGGL(vec_neg_k, dim3(grid_for(n)), dim3(256), 0, c->stream, y, n);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int vec_zero_list(lgh_ctx *c, double *y, const int *list, int n)
{
   if (n <= 0) { return LGH_OK; }
   hipLaunchKernelGGL(vec_zero_list_k, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, y, list, n);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int vec_dot(lgh_ctx *c, const double *x, const double *y, const double *w, long n, double *dev_out)
{
   hipLaunchKernelGGL(vec_dot_k, dim3(grid_for(n)), dim3(256), 0, c->stream, x, y, w, n,
                      c->partials + 3 * (size_t)c->part_stride, c->tickets + 3 * kTicketSlot, dev_out);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}

// ---- timers: HIP events around the reference's stopwatch regions -----------------
void timer_start(lgh_ctx *c)
{
   if (!c->timers.enabled) { return; }
   (void)hipEventRecord(c->timers.ev[0], c->stream);
}
void timer_stop(lgh_ctx *c, int which)
{
   if (!c->timers.enabled) { return; }
   (void)hipEventRecord(c->timers.ev[1], c->stream);
   (void)hipEventSynchronize(c->timers.ev[1]);
   float ms = 0.f;
   (void)hipEventElapsedTime(&ms, c->timers.ev[0], c->timers.ev[1]);
   c->timers.t[which] += 1e-3 * ms;
}

// ---- roctx ranges (LGH_ROCTX=1) ----------------------------------------------------------------
namespace
{
struct Roctx
{
   int (*push)(const char *) = nullptr;
   int (*pop)() = nullptr;
   Roctx()
   {
      const char *env = getenv("LGH_ROCTX");
      if (!(env && env[0] == '1')) { return; }
      void *h = nullptr;
      for (const char *n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
      {
         h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
         if (h) { break; }
      }
      if (!h) { return; }
      push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (!push || !pop) { push = nullptr; pop = nullptr; }
   }
};
const Roctx &roctx()
{
   static const Roctx r;
   return r;
}
} // namespace
RoctxRange::RoctxRange(const char *name) : on(roctx().push != nullptr)
{
   if (on) { (void)roctx().push(name); }
}
RoctxRange::~RoctxRange()
{
   if (on) { (void)roctx().pop(); }
}

template <typename T> static int dev_alloc_copy(T **dst, const T *src, size_t n)
{
   LGH_HIP_CHECK(hipMalloc((void **)dst, std::max<size_t>(n, 1) * sizeof(T)));
   if (src && n) { LGH_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice)); }
   return LGH_OK;
}
template <typename T> static int dev_alloc_zero(T **dst, size_t n)
{
   LGH_HIP_CHECK(hipMalloc((void **)dst, std::max<size_t>(n, 1) * sizeof(T)));
   LGH_HIP_CHECK(hipMemset(*dst, 0, std::max<size_t>(n, 1) * sizeof(T)));
   // hipMemset of device memory returns before the fill has run (null stream) and the
   // context's stream does not synchronise with the null stream
   LGH_HIP_CHECK(hipStreamSynchronize(nullptr));
   return LGH_OK;
}

static bool kernel_id_supported(int id)
{
   switch (id)
   {
      case 0x222: case 0x234: case 0x246: case 0x258: case 0x26A:
      case 0x322: case 0x334: case 0x346: case 0x358: case 0x36A:
         return true;
   }
   return false;
}

} // namespace lgh

namespace lgh
{
int host_wait_token(lgh_ctx *c, volatile unsigned long long *word, const unsigned long long token)
{
   const bool spin = !(getenv("LGH_SPIN") && getenv("LGH_SPIN")[0] == '0'); // (per call: the switch tests flip it inside one process)
   if (spin)
   {
      // polling for at most ~50 ms (a look normally returns within the tail of the last enqueued kernels: a chunk of
      // iterations, 1-20 ms); a host that would wait longer than that sleeps in the stream synchronisation below instead
      const auto t0 = std::chrono::steady_clock::now();
      for (long i = 0;; i++)
      {
         if (__atomic_load_n((const unsigned long long *)word, __ATOMIC_ACQUIRE) == token) { return LGH_OK; }
         __builtin_ia32_pause();
         if ((i & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) { break; }
      }
   }
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   if (__atomic_load_n((const unsigned long long *)word, __ATOMIC_ACQUIRE) != token) { set_error("host look: the finishing kernel did not report"); return LGH_ERR_HIP; }
   return LGH_OK;
}

int energy_overlap_poll(lgh_ctx *c)
{
   if (c->e_async != 1 || c->e_polled) { return LGH_OK; }
   int it = 0;
   std::swap(c->stream, c->stream2);
   c->on_stream2 = 1;
   int rc = cg_l2_end(c, &it);
   if (rc == LGH_OK) { rc = (hipEventRecord(c->ev_join, c->stream) == hipSuccess) ? LGH_OK : LGH_ERR_HIP; }
   std::swap(c->stream, c->stream2);
   c->on_stream2 = 0;
   if (rc) { return rc; }
   c->e_polled = 1;
   c->e_iters = it;
   return LGH_OK;
}

} // namespace lgh

using namespace lgh;

extern "C"
{

const char *lgh_last_error(void) { return g_err; }
const char *lgh_version(void) { return "laghos_hip 0.1 (gfx950)"; }

static int create_impl(const lgh_config *cfg, lgh_ctx *c, const int kid);

int lgh_create(const lgh_config *cfg, lgh_ctx **out)
{
   LGH_CHECK_ARG(cfg && out);
   LGH_CHECK_ARG(cfg->dim == 2 || cfg->dim == 3);
   LGH_CHECK_ARG(cfg->NE > 0 && cfg->N > 0);
   LGH_CHECK_ARG(cfg->h1_map && cfg->B_h1 && cfg->G_h1 && cfg->B_l2 && cfg->weights && cfg->gamma);
   // MFEM_VERIFY(L1D==D1D-1) (laghos_assembly.cpp:533, :943)
   if (cfg->L1D != cfg->D1D - 1)
   {
      set_error("L1D!=D1D-1");
      return LGH_ERR_UNSUPPORTED;
   }
   const int kid = (cfg->dim << 8) | (cfg->D1D << 4) | cfg->Q1D;
   if (!kernel_id_supported(kid))
   {
      set_error("Unknown kernel 0x%x", kid);
      return LGH_ERR_UNSUPPORTED;
   }
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
   {
      set_error("no HIP device available: the laghos_hip path requires an MI355X (gfx950) GPU");
      return LGH_ERR_HIP;
   }
   LGH_HIP_CHECK(hipSetDevice(cfg->device));
   // argument errors a caller can trigger are found before anything is allocated
   LGH_CHECK_ARG(cfg->cfl > 0.0); // (the time-step estimate is a minimum over cfl / inv_dt >= 0, folded as such: lgh_qrows.hpp)
   for (size_t i = 0, n = (size_t)cfg->NE * (cfg->dim == 2 ? cfg->D1D * cfg->D1D : cfg->D1D * cfg->D1D * cfg->D1D); i < n; i++)
   {
      if (cfg->h1_map[i] < 0 || cfg->h1_map[i] >= cfg->N) { set_error("h1_map entry out of range"); return LGH_ERR_ARG; }
   }
   for (int k = 0; k < cfg->dim; k++)
   {
      LGH_CHECK_ARG(cfg->ess_count[k] >= 0 && (cfg->ess_count[k] == 0 || cfg->ess[k]));
      for (int i = 0; i < cfg->ess_count[k]; i++)
      {
         if (cfg->ess[k][i] < 0 || cfg->ess[k][i] >= cfg->N) { set_error("essential dof out of range"); return LGH_ERR_ARG; }
      }
   }

   lgh_ctx *c = new lgh_ctx();
   memset((void *)c, 0, sizeof(lgh_ctx));
   new (&c->timers) Timers();
   c->device = cfg->device;
   // every failure below (HIP errors, out of memory) releases what was built so far
   const int rc_create = create_impl(cfg, c, kid);
   if (rc_create != LGH_OK)
   {
      lgh_destroy(c);
      return rc_create;
   }
   *out = c;
   return LGH_OK;
}

static int create_impl(const lgh_config *cfg, lgh_ctx *c, const int kid)
{
   c->dim = cfg->dim; c->NE = cfg->NE; c->D1D = cfg->D1D; c->Q1D = cfg->Q1D; c->L1D = cfg->L1D;
   const int dim = c->dim;
   c->ND = dim == 2 ? c->D1D * c->D1D : c->D1D * c->D1D * c->D1D;
   c->NQ = dim == 2 ? c->Q1D * c->Q1D : c->Q1D * c->Q1D * c->Q1D;
   c->NL = dim == 2 ? c->L1D * c->L1D : c->L1D * c->L1D * c->L1D;
   c->N = cfg->N;
   c->H1V = dim * c->N;
   c->L2V = c->NE * c->NL;
   c->kid = kid;
   c->visc = cfg->use_viscosity != 0;
   c->vort = cfg->use_vorticity != 0;
   c->cfl = cfg->cfl;
   c->h1order = (double)cfg->order_v;
   {
      // LGH_Q_TINY_GRAD=<value>: default threshold of the QUpdate shortcut; 0: exact zeros only; -1: off
      const char *env = getenv("LGH_Q_TINY_GRAD");
      c->q_tiny_grad = env ? atof(env) : 1e-30;
      c->stress_store = 1; // (the reference's behaviour: UpdateQuadratureData leaves stressJinvT in memory)
      c->stress_current = 0;
   }
   c->h0 = 0.0;
   c->device = cfg->device;
   c->cur_ess = -1;
   c->nranks = 1;
   c->rank = 0;
   if (cfg->stream) { c->stream = (hipStream_t)cfg->stream; c->own_stream = false; }
   else { LGH_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }

   int rc;
#define LGH_TRY(x) do { rc = (x); if (rc) { return rc; } } while (0)
   LGH_TRY(dev_alloc_copy(&c->B, cfg->B_h1, (size_t)c->Q1D * c->D1D));
   {
      // mirror symmetry of the tables (every nodal or Bernstein basis on symmetric points has it); LGH_B_SYM=0: assume not
      const char *env = getenv("LGH_B_SYM");
      auto mirror = [&](const double *B, const int n) {
         if (env && env[0] == '0') { return 0; }
         for (int i = 0; i < n; i++)
         {
            const double u = B[i], v = B[n - 1 - i];
            if (std::fabs(u - v) > 1e-14 * std::max(1.0, std::max(std::fabs(u), std::fabs(v)))) { return 0; } // (1e-15 at order 5)
         }
         return 1;
      };
      c->b_h1_sym = mirror(cfg->B_h1, c->Q1D * c->D1D);
      c->b_l2_sym = mirror(cfg->B_l2, c->Q1D * c->L1D);
   }
   LGH_TRY(dev_alloc_copy(&c->G, cfg->G_h1, (size_t)c->Q1D * c->D1D));
   LGH_TRY(dev_alloc_copy(&c->Bl, cfg->B_l2, (size_t)c->Q1D * c->L1D));
   LGH_TRY(dev_alloc_copy(&c->W, cfg->weights, (size_t)c->NQ));
   {
      // Tensor-product rule?  w[i] = W[i, 0, 0] / w[0]^(dim - 1) with w[0] = W[0]^(1/dim); every entry of W must be the
      // product of its three factors to 8 ulp.  The plane-form L2 mass apply then takes the weights from Q scalars
      // instead of NQ loads per element batch (compact mass data only: value(q, e) = s_e w[qx] w[qy] w[qz]).
      const int Q = c->Q1D, dim = c->dim;
      std::vector<double> w1((size_t)Q);
      const double w0 = (dim == 3) ? cbrt(cfg->weights[0]) : sqrt(cfg->weights[0]);
      bool ok = std::isfinite(w0) && w0 > 0.0;
      for (int i = 0; ok && i < Q; i++)
      {
         w1[i] = cfg->weights[i] / ((dim == 3) ? w0 * w0 : w0);
         ok = std::isfinite(w1[i]) && w1[i] > 0.0;
      }
      // (the kernels keep half of them in scalar registers: w[i] = w[Q - 1 - i], as every rule on symmetric points has it)
      for (int i = 0; ok && i < Q / 2; i++)
      {
         ok = std::fabs(w1[i] - w1[Q - 1 - i]) <= 8.9e-16 * w1[i];
         w1[Q - 1 - i] = w1[i];
      }
      for (int q = 0; ok && q < c->NQ; q++)
      {
         const int qx = q % Q, qy = (q / Q) % Q, qz = q / (Q * Q);
         const double prod = (dim == 3) ? w1[qx] * w1[qy] * w1[qz] : w1[qx] * w1[qy];
         ok = std::fabs(prod - cfg->weights[q]) <= 1.8e-15 * std::fabs(cfg->weights[q]);
      }
      const char *senv = getenv("LGH_MASS_SEP"); // A/B: 0 = weights from the NQ-entry table
      if (ok && !(senv && senv[0] == '0')) { LGH_TRY(dev_alloc_copy(&c->w1d, w1.data(), (size_t)Q)); }
      // 1-D mass tiles of the two bases on this rule, M1[i + n j] = sum_q B[q, i] w[q] B[q, j] (long double sums, rounded
      // once): with compact mass data the element matrices are s_e M1 (x) M1 (x) M1 (lgh_vcg_slab.hip, lgh_mass.hip)
      const char *kenv = getenv("LGH_MASS_KRON"); // A/B: 0 = always contract through the quadrature points
      if (c->w1d && !(kenv && kenv[0] == '0'))
      {
         auto tile = [&](const double *Bt, const int n, double **out) -> int {
            std::vector<double> M((size_t)n * n);
            for (int i = 0; i < n; i++)
            {
               for (int j = 0; j < n; j++)
               {
                  long double s = 0.0L;
                  for (int q = 0; q < Q; q++) { s += (long double)Bt[q + Q * i] * (long double)w1[q] * (long double)Bt[q + Q * j]; }
                  M[(size_t)i + (size_t)n * j] = (double)s;
               }
            }
            return dev_alloc_copy(out, M.data(), M.size());
         };
         LGH_TRY(tile(cfg->B_h1, c->D1D, &c->M1h));
         LGH_TRY(tile(cfg->B_l2, c->L1D, &c->M1l));
      }
   }
   LGH_TRY(dev_alloc_copy(&c->gamma, cfg->gamma, (size_t)c->NE));
   const size_t nmap = (size_t)c->NE * c->ND;
   LGH_TRY(dev_alloc_copy(&c->h1map, cfg->h1_map, nmap));
   LGH_TRY(mesh_order_build(c, cfg->h1_map)); // the library's own zone order and node numbering (lgh_order.hip)
   // transpose of the restriction in CSR form (ascending element order per node)
   {
      std::vector<int> off((size_t)c->N + 1, 0), idx(nmap);
      for (size_t i = 0; i < nmap; i++)
      {
         const int n = cfg->h1_map[i];
         if (n < 0 || n >= c->N) { set_error("h1_map entry out of range"); return LGH_ERR_ARG; }
         off[(size_t)n + 1]++;
      }
      for (int n = 0; n < c->N; n++) { off[(size_t)n + 1] += off[n]; }
      std::vector<int> pos(off.begin(), off.end() - 1);
      for (size_t i = 0; i < nmap; i++) { idx[pos[cfg->h1_map[i]]++] = (int)i; }
      LGH_TRY(dev_alloc_copy(&c->t_off, off.data(), off.size()));
      LGH_TRY(dev_alloc_copy(&c->t_idx, idx.data(), idx.size()));
      int deg = 0;
      for (int n = 0; n < c->N; n++) { deg = std::max(deg, off[(size_t)n + 1] - off[n]); }
      std::vector<int> ell((size_t)deg * c->N, -1);
      for (int n = 0; n < c->N; n++)
         for (int k = off[n]; k < off[(size_t)n + 1]; k++) { ell[(size_t)(k - off[n]) * c->N + n] = idx[k]; }
      c->t_deg = deg;
      LGH_TRY(dev_alloc_copy(&c->t_ell, ell.data(), ell.size()));
      const char *env = getenv("LGH_VCG_VARIANT"); // A/B switch of the lockstep K1 (lgh_vcg.hip)
      c->vcg_variant = (env && env[0] >= '0' && env[0] <= '9') ? env[0] - '0' : -1; // -1: by kernel id and mesh size (vcg_k1_form)
      // slab-form K1 (lgh_vcg_slab.hip), A/B: wavefronts per SIMD (default 1), row loads, exact sum of (d, A d), sets drawn from a workgroup queue
      env = getenv("LGH_SLAB_WPS");
      c->slab_wps = (env && env[0] == '2') ? 2 : 1;
      env = getenv("LGH_SLAB_WIDE");
      c->slab_wide = (env && env[0] == '0') ? 0 : 1;
      env = getenv("LGH_SLAB_EXACT");
      c->slab_exact = (env && env[0] == '0') ? 0 : 1;
      env = getenv("LGH_SLAB_DYN");
      c->slab_dyn = (env && env[0] == '0') ? 0 : (env && env[0] == '1') ? 1 : -1; // -1: by passes per wavefront (launch_vcg_slab)
   }
   for (int k = 0; k < 3; k++)
   {
      c->ess_count[k] = (k < dim) ? cfg->ess_count[k] : 0;
      std::vector<uint8_t> mask((size_t)c->N, 0);
      for (int i = 0; i < c->ess_count[k]; i++)
      {
         const int n = cfg->ess[k][i];
         if (n < 0 || n >= c->N) { set_error("essential dof out of range"); return LGH_ERR_ARG; }
         mask[n] = 1;
      }
      LGH_TRY(dev_alloc_copy(&c->essmask[k], mask.data(), mask.size()));
      LGH_TRY(dev_alloc_copy(&c->ess[k], c->ess_count[k] ? cfg->ess[k] : nullptr, (size_t)c->ess_count[k]));
   }
   if (cfg->owner) { LGH_TRY(dev_alloc_copy(&c->owner, cfg->owner, (size_t)c->N)); }

   const size_t nq = (size_t)c->NE * c->NQ;
   LGH_TRY(dev_alloc_zero(&c->stressJinvT, nq * dim * dim));
   LGH_TRY(dev_alloc_zero(&c->Jac0inv, nq * dim * dim));
   LGH_TRY(dev_alloc_zero(&c->Jac0inv_soa, nq * dim * dim));
   LGH_TRY(dev_alloc_zero(&c->Jac0inv_e, (size_t)cfg->NE * dim * dim));
   LGH_TRY(dev_alloc_zero(&c->rho0DetJ0w, nq));
   LGH_TRY(dev_alloc_zero(&c->massD, nq + 2048)); // (one set of the matrix-core K1 behind the last element: its pipeline prefetches without predicates)
   LGH_TRY(dev_alloc_zero(&c->diagV, (size_t)c->N));
   LGH_TRY(dev_alloc_zero(&c->dinvV, (size_t)c->N));
   LGH_TRY(dev_alloc_zero(&c->dt_est_dev, (size_t)kDtSlotStride * (1 + kDtSlots)));
   {
      const char *env = getenv("LGH_FUSED_FTV"); // A/B: 0 = F^T v always by its own kernel
      if (!(env && env[0] == '0'))
      {
         LGH_TRY(dev_alloc_zero(&c->erhs_q, (size_t)c->L2V));
         LGH_TRY(dev_alloc_zero(&c->v_snap, (size_t)c->H1V));
      }
      LGH_TRY(dev_alloc_zero(&c->dev_flags, (size_t)8));
      c->mass_rank1 = -1;
      env = getenv("LGH_FUSED_F1");              // A/B: 0 = F.1 always by its own kernel
      if (c->dim == 3 && !(env && env[0] == '0')) { LGH_TRY(dev_alloc_zero(&c->force_e_q, nmap * dim + (size_t)dim * c->ND)); } // (+ a zero element: vcg_init_force_z_k)
   }
   const size_t ne_nd = nmap * dim;
   LGH_TRY(dev_alloc_zero(&c->XE, std::max<size_t>(ne_nd, (size_t)c->L2V)));
   LGH_TRY(dev_alloc_zero(&c->YE, ne_nd + (size_t)dim * c->ND)); // (+ a zero element: vcg_init_force_z_k)
   const size_t nv = std::max<size_t>((size_t)c->N, (size_t)c->L2V);
   LGH_TRY(dev_alloc_zero(&c->cg_r, nv));
   LGH_TRY(dev_alloc_zero(&c->cg_z, nv));
   LGH_TRY(dev_alloc_zero(&c->cg_d0, nv));
   LGH_TRY(dev_alloc_zero(&c->cg_d1, nv));
   LGH_TRY(dev_alloc_zero(&c->cg_y, nv));
   c->part_stride = (int)std::max<size_t>(std::max<size_t>((size_t)c->NE, (nv + 255) / 256), 2048) + (int)kShards;
   LGH_TRY(dev_alloc_zero(&c->partials, 4 * (size_t)c->part_stride));
   LGH_TRY(dev_alloc_zero(&c->tickets, 4 * (size_t)kTicketSlot));
   LGH_TRY(dev_alloc_zero(&c->cgs, 1));
   LGH_TRY(dev_alloc_zero(&c->scal, 16));
   LGH_HIP_CHECK(hipHostMalloc((void **)&c->host_pinned, 96 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
   memset(c->host_pinned, 0, 96 * sizeof(double));
   LGH_HIP_CHECK(hipHostGetDevicePointer((void **)&c->host_pinned_dev, c->host_pinned, 0)); // (the one-thread kernels in front of a host look write there)
   LGH_HIP_CHECK(hipEventCreate(&c->timers.ev[0]));
   LGH_HIP_CHECK(hipEventCreate(&c->timers.ev[1]));
   {
      const std::vector<double> infs((size_t)kDtSlotStride * (1 + kDtSlots), std::numeric_limits<double>::infinity());
      LGH_HIP_CHECK(hipMemcpy(c->dt_est_dev, infs.data(), infs.size() * sizeof(double), hipMemcpyHostToDevice));
   }
#undef LGH_TRY
   return LGH_OK;
}

int lgh_destroy(lgh_ctx *c)
{
   if (!c) { return LGH_OK; }
   (void)hipSetDevice(c->device);
   (void)hipStreamSynchronize(c->stream);
   void *ptrs[] = {c->B, c->G, c->Bl, c->W, c->w1d, c->M1h, c->M1l, c->gamma, c->h1map, c->t_off, c->t_idx, c->t_ell, c->essmask[0],
                   c->essmask[1], c->essmask[2], c->ess[0], c->ess[1], c->ess[2], c->owner,
                   c->stressJinvT, c->Jac0inv, c->Jac0inv_soa, c->Jac0inv_e, c->rho0DetJ0w, c->massD, c->diagV, c->dinvV,
                   c->dt_est_dev, c->erhs_q, c->v_snap, c->dev_flags, c->ones_l2, c->massS, c->ones_ne, c->force_e_q, c->XE, c->YE, c->cg_r, c->cg_z, c->cg_d0, c->cg_d1, c->cg_y,
                   c->partials, c->tickets, c->cgs, c->scal, c->vcg_s, c->vcg_vec, c->vcg_partials,
                   c->vcg_tickets};
   for (void *p : ptrs) { if (p) { (void)hipFree(p); } }
   if (c->host_pinned) { (void)hipHostFree(c->host_pinned); }
   if (c->timers.ev[0]) { (void)hipEventDestroy(c->timers.ev[0]); }
   if (c->timers.ev[1]) { (void)hipEventDestroy(c->timers.ev[1]); }
   if (c->ktime)
   {
      for (hipEvent_t e : c->ktime->ev) { (void)hipEventDestroy(e); }
      delete c->ktime;
   }
   cg_l2_free(c);
   vcg_free(c);
   mesh_order_free(c);
   if (c->q_trace_dev) { (void)hipFree(c->q_trace_dev); }
   if (c->stream2)
   {
      (void)hipStreamSynchronize(c->stream2);
      (void)hipStreamDestroy(c->stream2);
      (void)hipEventDestroy(c->ev_fork);
      (void)hipEventDestroy(c->ev_join);
   }
   for (auto &ring : c->ls_ev) { for (hipEvent_t e : ring) { if (e) { (void)hipEventDestroy(e); } } }
   extern void lgh_comm_free(lgh_ctx *);
   lgh_comm_free(c);
   if (c->own_stream) { (void)hipStreamDestroy(c->stream); }
   delete c;
   return LGH_OK;
}

int lgh_sync(lgh_ctx *c)
{
   LGH_CHECK_ARG(c);
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   return LGH_OK;
}
void *lgh_stream(lgh_ctx *c) { return c ? (void *)c->stream : nullptr; }

static void invalidate_fused(lgh_ctx *c)
{
   c->fused_ftv_valid = 0;
   c->fused_f1_valid = 0;
   c->qgen++;
}
double *lgh_qdata_stressJinvT(lgh_ctx *c)
{
   // the caller may write through this pointer: F^T v and F.1 of the fused update no longer belong to it
   invalidate_fused(c);
   // default mode: whatever the caller puts there is what the force kernels read from now on.  With the stress kept in
   // registers (explicit opt-in) a request for the pointer does not turn planes nobody wrote into current data: the
   // readers keep refusing until lgh_qupdate_store_stress(ctx, 1) and an update
   if (c->stress_store) { c->stress_current = 1; }
   return c->stressJinvT;
}
// The stress of the current quadrature data is in memory (readers of stressJinvT call this first).
static int stress_on_hand(lgh_ctx *c, const char *who)
{
   if (c->stress_current) { return LGH_OK; }
   set_error("%s needs stressJinvT, which the last lgh_qupdate kept in registers (lgh_qupdate_store_stress(ctx, 0)): "
             "only F.1 and F^T v of the state's own velocity are on hand; store the stress (..., 1) and update again", who);
   return LGH_ERR_ARG;
}
int lgh_qupdate_form(lgh_ctx *c, int *form)
{
   LGH_CHECK_ARG(c && form);
   *form = qupdate_form(c);
   return LGH_OK;
}
int lgh_qupdate_stores_stress(lgh_ctx *c, int *on)
{
   LGH_CHECK_ARG(c && on);
   // (what the next lgh_qupdate will do: the planes are only left out when both products come out of the update kernel)
   *on = (!c->stress_store && c->erhs_q && c->force_e_q && c->v_snap && !c->fused_forces_off && c->dim == 3) ? 0 : 1;
   return LGH_OK;
}
int lgh_qupdate_store_stress(lgh_ctx *c, int on)
{
   LGH_CHECK_ARG(c);
   if ((on != 0) != (c->stress_store != 0))
   {
      c->stress_store = on ? 1 : 0;
      invalidate_fused(c); // the quadrature data has to be updated again before anything reads it
      c->stress_current = 0;
   }
   return LGH_OK;
}
int lgh_reset_quadrature_data(lgh_ctx *c)
{
   LGH_CHECK_ARG(c);
   invalidate_fused(c);
   return LGH_OK;
}
// The force products of the last lgh_qupdate as vectors: F.1 summed to the H1 L-vector (what SolveVelocity negates into
// its right-hand side), F^T v_state as L2 vector.  LGH_ERR_ARG when the product is not on hand (fusion off, 2D for
// F.1, quadrature data changed since).
int lgh_fused_force_mult(lgh_ctx *c, double *y_h1)
{
   LGH_CHECK_ARG(c && y_h1);
   if (!(c->force_e_q && c->fused_f1_valid)) { set_error("lgh_fused_force_mult: no fused F.1 on hand"); return LGH_ERR_ARG; }
   int rc = h1_transpose_gather(c, c->dim, c->force_e_q, y_h1);
   if (rc == LGH_OK && c->multi != 0) { rc = halo_sum(c, y_h1, c->dim); }
   return rc;
}
int lgh_fused_force_mult_transpose(lgh_ctx *c, double *y_l2)
{
   LGH_CHECK_ARG(c && y_l2);
   if (!(c->erhs_q && c->fused_ftv_valid)) { set_error("lgh_fused_force_mult_transpose: no fused F^T v on hand"); return LGH_ERR_ARG; }
   LGH_HIP_CHECK(hipMemcpyAsync(y_l2, c->erhs_q, sizeof(double) * (size_t)c->L2V, hipMemcpyDeviceToDevice, c->stream));
   return LGH_OK;
}
int lgh_quadrature_generation(lgh_ctx *c, unsigned long *gen, int *f1_valid, int *ftv_valid)
{
   LGH_CHECK_ARG(c && gen && f1_valid && ftv_valid);
   *gen = c->qgen;
   *f1_valid = c->fused_f1_valid;
   *ftv_valid = c->fused_ftv_valid;
   return LGH_OK;
}
double *lgh_qdata_Jac0inv(lgh_ctx *c) { return c->Jac0inv; }
double *lgh_qdata_rho0DetJ0w(lgh_ctx *c) { return c->rho0DetJ0w; }
double *lgh_mass_D(lgh_ctx *c)
{
   c->mass_rank1 = -1; // (the caller may write through this pointer: the compact form of the mass data is looked for again)
   c->mass_gen++;
   return c->massD;
}
int lgh_mass_data_form(lgh_ctx *c, int *form)
{
   LGH_CHECK_ARG(c && form);
   const double *Dq, *Se;
   int dqs;
   int rc = mass_data(c, &Dq, &dqs, &Se);
   if (rc) { return rc; }
   *form = (c->mass_rank1 == 1) ? 1 : 0;
   return LGH_OK;
}
int lgh_mass_data_changed(lgh_ctx *c)
{
   LGH_CHECK_ARG(c);
   c->mass_rank1 = -1;            // the compact form is looked for again at the next mass apply
   c->mass_gen++;
   return mass_assemble_diag(c);  // operator and Jacobi preconditioner stay consistent (laghos_solver.cpp:266-270)
}
double *lgh_mass_diag(lgh_ctx *c) { return c->diagV; }
int lgh_set_h0(lgh_ctx *c, double h0) { LGH_CHECK_ARG(c); c->h0 = h0; return LGH_OK; }
int lgh_get_h0(lgh_ctx *c, double *h0) { LGH_CHECK_ARG(c && h0); *h0 = c->h0; return LGH_OK; }

int lgh_set_dt_est(lgh_ctx *c, double v)
{
   LGH_CHECK_ARG(c);
   hipLaunchKernelGGL(dt_est_set_k, dim3(1), dim3(256), 0, c->stream, c->dt_est_dev, v);
   LGH_HIP_CHECK(hipGetLastError());
   return LGH_OK;
}
int lgh_get_dt_est(lgh_ctx *c, double *v)
{
   LGH_CHECK_ARG(c && v);
   const unsigned long long token = ++c->look_token;
   hipLaunchKernelGGL(dt_est_fold_k, dim3(1), dim3(256), 0, c->stream, c->dt_est_dev, c->dev_flags + 4, c->host_pinned_dev + 8, token);
   LGH_HIP_CHECK(hipGetLastError());
   {
      const int rc = host_wait_token(c, (volatile unsigned long long *)(c->host_pinned + 10), token);
      if (rc) { return rc; }
   }
   *v = c->host_pinned[8];
   if (*(const int *)(c->host_pinned + 9) != 0)
   {
      LGH_HIP_CHECK(hipMemsetAsync(c->dev_flags + 4, 0, sizeof(int), c->stream));
      set_error("lgh_solve_energy was given a velocity other than the state's while the stress was kept in registers "
                "(lgh_qupdate_store_stress(ctx, 0)): its right-hand side is NaN");
      return LGH_ERR_ARG;
   }
   return LGH_OK;
}

int lgh_setup_rho0detj0(lgh_ctx *c, const double *x0, const double *rho0_l2, const double *rho0_q,
                        double *volume)
{
   LGH_CHECK_ARG(c && x0 && rho0_l2 && rho0_q && volume);
   invalidate_fused(c);
   c->mass_rank1 = -1; // (new mass data)
   c->mass_gen++;
   int rc = setup_rho0detj0(c, x0, rho0_l2, rho0_q, volume);
   if (rc) { return rc; }
   return mass_assemble_diag(c);
}

int lgh_force_mult(lgh_ctx *c, const double *x_l2, double *y_h1)
{
   LGH_CHECK_ARG(c && x_l2 && y_h1);
   // L2R->Mult is the identity for the lexicographic L2 space (assembly.cpp:559-560)
   int rc = stress_on_hand(c, "lgh_force_mult"); // (a refusal is taken before the timing sample opens)
   if (rc) { return rc; }
   kt_begin(c, LGH_KERNEL_FORCE_MULT);
   rc = force_mult_E(c, c->stressJinvT, x_l2, c->YE);
   kt_end(c, LGH_KERNEL_FORCE_MULT);
   if (rc) { return rc; }
   rc = h1_transpose_gather(c, c->dim, c->YE, y_h1); // H1R->MultTranspose (:564)
   if (rc) { return rc; }
   if (c->multi != 0) { rc = halo_sum(c, y_h1, c->dim); }
   return rc;
}
int lgh_force_mult_transpose(lgh_ctx *c, const double *v_h1, double *y_l2)
{
   LGH_CHECK_ARG(c && v_h1 && y_l2);
   const int rc0 = stress_on_hand(c, "lgh_force_mult_transpose");
   if (rc0) { return rc0; }
   kt_begin(c, LGH_KERNEL_FORCE_MULT_T);
   const int rc = force_mult_t_L(c, c->stressJinvT, v_h1, y_l2);
   kt_end(c, LGH_KERNEL_FORCE_MULT_T);
   return rc;
}

int lgh_mass_set_essential_tdofs(lgh_ctx *c, int comp)
{
   LGH_CHECK_ARG(c && comp >= -1 && comp < c->dim);
   c->cur_ess = comp;
   return LGH_OK;
}
int lgh_mass_eliminate_rhs(lgh_ctx *c, double *b)
{
   LGH_CHECK_ARG(c && b);
   if (c->cur_ess < 0) { return LGH_OK; }
   return vec_zero_list(c, b, c->ess[c->cur_ess], c->ess_count[c->cur_ess]);
}
int lgh_mass_mult(lgh_ctx *c, int space, const double *x, double *y)
{
   LGH_CHECK_ARG(c && x && y && (space == LGH_SPACE_H1 || space == LGH_SPACE_L2));
   return space == LGH_SPACE_H1 ? mass_apply_h1(c, x, y, true) : mass_apply_l2(c, x, y);
}
int lgh_mass_mult_full(lgh_ctx *c, int space, const double *x, double *y)
{
   LGH_CHECK_ARG(c && x && y && (space == LGH_SPACE_H1 || space == LGH_SPACE_L2));
   return space == LGH_SPACE_H1 ? mass_apply_h1(c, x, y, false) : mass_apply_l2(c, x, y);
}

int lgh_cg_solve(lgh_ctx *c, int space, const double *b, double *x, double rel_tol, int max_iter,
                 int *iters)
{
   LGH_CHECK_ARG(c && b && x && (space == LGH_SPACE_H1 || space == LGH_SPACE_L2));
   return cg_solve(c, space, b, x, rel_tol, max_iter, iters, false);
}

int lgh_qupdate(lgh_ctx *c, const double *S)
{
   LGH_CHECK_ARG(c && S);
   RoctxRange range("QUpdate-UpdateQuadratureData"); // laghos_solver.cpp:1358
   timer_start(c);
   kt_begin(c, LGH_KERNEL_QUPDATE);
   int rc = qupdate(c, S);
   kt_end(c, LGH_KERNEL_QUPDATE);
   timer_stop(c, 3);
   c->timers.c[2] += c->NE;
   return rc;
}

// The fused update forms F.1 for the constant-one L2 function, which is what SolveVelocity passes
// (laghos_solver.cpp:170-171, :354).  one_l2 == NULL means exactly that - the operator's own `one`, as in the
// reference, whose SolveVelocity takes no such argument - and costs nothing.  A vector passed explicitly is checked
// on every call (a pass over it and a host look): nothing is remembered about an address.
__global__ void __launch_bounds__(256) not_all_ones_k(const double *x, const long n, int *flag)
{
   const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n && x[i] != 1.0) { *flag = 1; }
}
static bool is_the_one_vector(lgh_ctx *c, const double *one_l2)
{
   if (!one_l2) { return true; }
   int *flag = c->dev_flags + 1;
   if (hipMemsetAsync(flag, 0, sizeof(int), c->stream) != hipSuccess) { return false; }
   hipLaunchKernelGGL(not_all_ones_k, dim3(ceil_div(c->L2V, 256)), dim3(256), 0, c->stream, one_l2, (long)c->L2V, flag);
   int h = 1;
   if (hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) { return false; }
   if (hipStreamSynchronize(c->stream) != hipSuccess) { return false; }
   return h == 0;
}
// the vector ForcePA->Mult is applied to when the kernel has to run: the caller's, or the context's own ones
static int one_vector(lgh_ctx *c, const double *one_l2, const double **out)
{
   if (one_l2) { *out = one_l2; return LGH_OK; }
   if (!c->ones_l2)
   {
      LGH_HIP_CHECK(hipMalloc((void **)&c->ones_l2, sizeof(double) * (size_t)c->L2V));
      const int rc = vec_set(c, c->ones_l2, 1.0, c->L2V);
      if (rc) { return rc; }
   }
   *out = c->ones_l2;
   return LGH_OK;
}

// SolveVelocity, PA branch without acceleration source (laghos_solver.cpp:329-399).
// The caller has already run UpdateQuadratureData(S) if the data was stale (:332).
int lgh_solve_velocity(lgh_ctx *c, const double *S, double *dS_dt, const double *one_l2,
                       double *rhs_h1, double *work_B, double rel_tol, int max_iter, int *h1_iters)
{
   LGH_CHECK_ARG(c && S && dS_dt && rhs_h1 && work_B);
   const int dim = c->dim, N = c->N;
   double *dv = dS_dt + c->H1V;
   int rc;
   // One rank, region timers off, no acceleration source: the E->L sum of F.1, the negation,
   // EliminateRHS, dv = 0 and the CG initialisation are one kernel (vcg_init_force_k), with
   // bit-identical results.  With the timers on the reference's regions are kept apart.
   if (!c->timers.enabled && !c->accel_src && vcg_fused_init_ok(c))
   {
      // F.1 up to the E-vector (:354): formed by the fused update for this very state, or by the force kernel
      const double *force_E = c->YE;
      if (c->force_e_q && c->fused_f1_valid && is_the_one_vector(c, one_l2)) { force_E = c->force_e_q; }
      else
      {
         const double *ones = nullptr;
         rc = one_vector(c, one_l2, &ones);
         if (rc) { return rc; }
         rc = stress_on_hand(c, "lgh_solve_velocity (F.1 of a vector other than one, or of changed quadrature data)");
         if (rc) { return rc; }
         kt_begin(c, LGH_KERNEL_FORCE_MULT);
         rc = force_mult_E(c, c->stressJinvT, ones, c->YE);
         kt_end(c, LGH_KERNEL_FORCE_MULT);
         if (rc) { return rc; }
      }
      int its[3] = {0, 0, 0};
      RoctxRange range("SolveVelocity-CGVMass"); // laghos_solver.cpp:387-390 (here: with the E->L sum of F.1 and EliminateRHS in its first kernel)
      rc = vcg_solve(c, rhs_h1, dv, rel_tol, max_iter, its, force_E); // :358-388
      if (rc) { return rc; }
      for (int cc = 0; cc < dim; cc++)
      {
         c->timers.c[0] += its[cc]; // :392
         if (h1_iters) { *h1_iters += its[cc]; }
      }
      c->cur_ess = dim - 1;
      return LGH_OK;
   }
   rc = vec_set(c, dv, 0.0, c->H1V); // dv = 0.0 (:338)
   if (rc) { return rc; }
   {
   RoctxRange range("SolveVelocity-ForcePA"); // laghos_solver.cpp:353-356
   timer_start(c);
   if (c->force_e_q && c->fused_f1_valid && is_the_one_vector(c, one_l2))
   {
      // F.1 of this state from the fused update: only H1R^T (and the sum over the ranks) is left of :354, and
      // the right-hand side has the bits of the fused-init path above
      rc = h1_transpose_gather(c, c->dim, c->force_e_q, rhs_h1);
      if (rc == LGH_OK && c->multi != 0) { rc = halo_sum(c, rhs_h1, c->dim); }
   }
   else
   {
      const double *ones = nullptr;
      rc = one_vector(c, one_l2, &ones);
      if (rc == LGH_OK) { rc = lgh_force_mult(c, ones, rhs_h1); } // :354
   }
   timer_stop(c, 2);
   }
   if (rc) { return rc; }
   rc = vec_neg_inplace(c, rhs_h1, c->H1V); // :358
   if (rc) { return rc; }
   if (c->accel_src) // source_type == 2: B_c += VMassPA->MultFull(accel_c) (:371-380), before EliminateRHS
   {
      for (int cc = 0; cc < dim; cc++)
      {
         rc = mass_apply_h1(c, c->accel_src + (size_t)cc * N, work_B, false);
         if (rc) { return rc; }
         rc = vec_axpby(c, rhs_h1 + (size_t)cc * N, 1.0, rhs_h1 + (size_t)cc * N, 1.0, work_B, N);
         if (rc) { return rc; }
      }
   }
   // the dim component solves in lockstep (lgh_vcg.hip) when the kernel id has it
   {
      for (int cc = 0; cc < dim; cc++)
      {
         // EliminateRHS with c_tdofs[cc] (:383-384); rhs_h1 is caller scratch
         rc = vec_zero_list(c, rhs_h1 + (size_t)cc * N, c->ess[cc], c->ess_count[cc]);
         if (rc) { return rc; }
      }
      int its[3] = {0, 0, 0};
      RoctxRange range("SolveVelocity-CGVMass"); // laghos_solver.cpp:387-390
      timer_start(c);
      rc = vcg_solve(c, rhs_h1, dv, rel_tol, max_iter, its); // :388 for all components
      if (rc == LGH_OK)
      {
         timer_stop(c, 0);
         for (int cc = 0; cc < dim; cc++)
         {
            c->timers.c[0] += its[cc]; // :392
            if (h1_iters) { *h1_iters += its[cc]; }
         }
         c->cur_ess = dim - 1;
         return LGH_OK;
      }
      if (rc != LGH_ERR_UNSUPPORTED) { return rc; }
   }
   for (int cc = 0; cc < dim; cc++)
   {
      // B = rhs_c (Pconf is the identity on a conforming mesh; shared-node sums
      // were already applied by lgh_force_mult) (:368-369)
      LGH_HIP_CHECK(hipMemcpyAsync(work_B, rhs_h1 + (size_t)cc * N, sizeof(double) * N,
                                   hipMemcpyDeviceToDevice, c->stream));
      rc = lgh_mass_set_essential_tdofs(c, cc); // :383
      if (rc) { return rc; }
      rc = lgh_mass_eliminate_rhs(c, work_B); // :384
      if (rc) { return rc; }
      int it = 0;
      timer_start(c);
      rc = cg_solve(c, LGH_SPACE_H1, work_B, dv + (size_t)cc * N, rel_tol, max_iter, &it, true); // :388
      timer_stop(c, 0);
      if (rc) { return rc; }
      c->timers.c[0] += it; // :392
      if (h1_iters) { *h1_iters += it; }
   }
   return LGH_OK;
}

// ForcePA->MultTranspose(v, e_rhs) of SolveEnergy (laghos_solver.cpp:473).  The fused QUpdate has formed F^T v for the
// velocity of the state it was called for; it stands in for the kernel exactly when the quadrature data is still that
// of this update AND the v passed now equals that velocity element for element - compared on the device, whatever the
// address (RK2Avg's averaged velocity, or a state changed in place since, go through the kernel, as in the reference).
// No host look: the comparison leaves a flag, the force kernel returns at once when it is 0, the copy when it is not.
__global__ void __launch_bounds__(256) vec_differs_k(const double *__restrict__ a, const double *__restrict__ b, const long n, int *flag)
{
   // (compared as bit patterns: -0.0 vs 0.0 or a NaN payload count as different - the kernel then simply runs)
   // grid-stride, 16-byte loads where both vectors allow them (round 5: one element per thread reached 2 TB/s at 64^3)
   const long stride = (long)gridDim.x * blockDim.x, t = (long)blockIdx.x * blockDim.x + threadIdx.x;
   bool diff = false;
   if (((((uintptr_t)a) | ((uintptr_t)b)) & 15u) == 0)
   {
      const longlong2 *a2 = (const longlong2 *)a, *b2 = (const longlong2 *)b;
      for (long i = t; i < n / 2; i += stride)
      {
         const longlong2 p = a2[i], q = b2[i];
         diff = diff || p.x != q.x || p.y != q.y;
      }
      if ((n & 1) && t == 0) { diff = diff || __double_as_longlong(a[n - 1]) != __double_as_longlong(b[n - 1]); }
   }
   else
   {
      for (long i = t; i < n; i += stride) { diff = diff || __double_as_longlong(a[i]) != __double_as_longlong(b[i]); }
   }
   if (diff) { *flag = 1; }
}
// e_rhs = F^T v of the quadrature update unless the velocity differs from the one it was formed for; then (the stress being in
// registers only) NaN and the error word (rounds 3-4: two launches, poison and copy)
__global__ void __launch_bounds__(256) erhs_take_or_poison_k(double *__restrict__ y, const double *__restrict__ x, const long n, const int *__restrict__ flag, int *err)
{
   const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
   if (*flag != 0)
   {
      if (i < n) { y[i] = __builtin_nan(""); }
      if (i == 0) { *err = 1; }
      return;
   }
   if (i < n) { y[i] = x[i]; }
}
__global__ void __launch_bounds__(256) copy_unless_k(double *__restrict__ y, const double *__restrict__ x, const long n, const int *__restrict__ flag)
{
   if (*flag != 0) { return; }
   const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { y[i] = x[i]; }
}
static int energy_rhs(lgh_ctx *c, const double *v_h1, double *e_rhs)
{
   RoctxRange range("SolveEnergy-ForcePA"); // laghos_solver.cpp:472-475
   if (c->erhs_q && c->fused_ftv_valid)
   {
      int *flag = c->dev_flags + (c->on_stream2 ? 2 : 0);
      LGH_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
      hipLaunchKernelGGL(vec_differs_k, dim3(grid_for((c->H1V + 1) / 2)), dim3(256), 0, c->stream, v_h1, c->v_snap, (long)c->H1V, flag);
      LGH_HIP_CHECK(hipGetLastError());
      if (c->stress_current)
      {
         const int rc = force_mult_t_L(c, c->stressJinvT, v_h1, e_rhs, flag); // (runs only for a different v)
         if (rc) { return rc; }
         hipLaunchKernelGGL(copy_unless_k, dim3(ceil_div(c->L2V, 256)), dim3(256), 0, c->stream, e_rhs, c->erhs_q, (long)c->L2V, flag);
      }
      else
      {
         // the stress was kept in registers: a velocity other than the state's cannot be served - the right-hand side
         // becomes NaN on the device (nothing downstream can look right) and the next lgh_get_dt_est reports it
         hipLaunchKernelGGL(erhs_take_or_poison_k, dim3(ceil_div(c->L2V, 256)), dim3(256), 0, c->stream, e_rhs, c->erhs_q, (long)c->L2V, flag, c->dev_flags + 4);
      }
      LGH_HIP_CHECK(hipGetLastError());
      return LGH_OK;
   }
   return lgh_force_mult_transpose(c, v_h1, e_rhs);
}

// SolveEnergy, PA branch (laghos_solver.cpp:442-490)
int lgh_solve_energy(lgh_ctx *c, const double *S, const double *v_h1, double *dS_dt, double *e_rhs,
                     const double *e_source, double rel_tol, int max_iter, int *l2_iters)
{
   LGH_CHECK_ARG(c && S && v_h1 && dS_dt && e_rhs);
   (void)S;
   double *de = dS_dt + 2 * (size_t)c->H1V;
   timer_start(c);
   int rc = energy_rhs(c, v_h1, e_rhs); // :473
   timer_stop(c, 2);
   if (rc) { return rc; }
   if (e_source)
   {
      rc = vec_axpby(c, e_rhs, 1.0, e_rhs, 1.0, e_source, c->L2V); // :477
      if (rc) { return rc; }
   }
   int it = 0;
   RoctxRange range("SolveEnergy-CGEMass"); // laghos_solver.cpp:480-483
   timer_start(c);
   rc = cg_solve(c, LGH_SPACE_L2, e_rhs, de, rel_tol, max_iter, &it, true); // :481
   timer_stop(c, 1);
   if (rc) { return rc; }
   const int counted = (it == 0) ? 1 : it; // :486
   c->timers.c[1] += counted;
   if (l2_iters) { *l2_iters += counted; }
   return LGH_OK;
}

// SolveEnergy does not depend on SolveVelocity's result (it takes v from S), so
// LagrangianHydroOperator::Mult may run the two concurrently: _begin enqueues F^T v
// and the first chunk of the L2 CG on the second stream behind a fork event, _end
// completes the solve and joins.  Falls back to the sequential lgh_solve_energy
// inside _end when region timers / kernel timing are on (their semantics are
// sequential, as in the reference), on several ranks without a second communicator
// (one RCCL communicator must not be driven from two streams: lgh_comm_init creates a
// second one for this), or when the velocity solve would use the scalar CG (shared
// scratch).  LGH_OVERLAP=0 switches it off.
// The second stream, for the energy solve beside the velocity solve - at the LOWEST priority the device offers (round 6): the
// velocity solve is the longer of the two at every order up to Q4Q3 and its kernels fill the device; the energy solve's small
// kernels then take what is left instead of an equal share of the dispatch slots: 8.81 -> 8.75 ms per step at C2 on one box,
// 8.85 -> 8.78 on another; config 5, where the energy CG is the longer one, does not lose (372 -> 367 ms), the highest priority
// gains nothing anywhere (profiles/r6_stream_priority.txt).  LGH_STREAM2_PRIORITY=d / h: the default / the highest priority.
static int create_stream2(lgh_ctx *c)
{
   const char *penv = getenv("LGH_STREAM2_PRIORITY");
   const char want = (penv && (penv[0] == 'd' || penv[0] == 'h')) ? penv[0] : 'l';
   int least = 0, greatest = 0;
   if (want != 'd' && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
   {
      LGH_HIP_CHECK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, want == 'h' ? greatest : least));
   }
   else { LGH_HIP_CHECK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)); }
   LGH_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
   LGH_HIP_CHECK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
   return LGH_OK;
}
static bool energy_overlap_ok(const lgh_ctx *c)
{
   const bool on = !(getenv("LGH_OVERLAP") && getenv("LGH_OVERLAP")[0] == '0');
   return on && (c->multi == 0 || comm_second_channel(c)) && !c->timers.enabled && !(c->ktime && c->ktime->which >= 0) && vcg_available(c);
}
int lgh_solve_energy_begin(lgh_ctx *c, const double *S, const double *v_h1, double *dS_dt, double *e_rhs,
                           const double *e_source, double rel_tol, int max_iter)
{
   LGH_CHECK_ARG(c && S && v_h1 && dS_dt && e_rhs);
   c->e_args = {S, v_h1, dS_dt, e_rhs, e_source, rel_tol, max_iter};
   c->e_lockstep = 0;
   if (!energy_overlap_ok(c))
   {
      // Several ranks without a second channel (the default over real RCCL): the energy CG runs in LOCKSTEP with the velocity
      // CG on this one stream and communicator instead of after it - its two dot products per iteration ride on exchanges the
      // velocity iteration makes anyway.  Right-hand side and initial residual here, the iterations inside
      // lgh_solve_velocity, what is left of them in _end.  LGH_ENERGY_LOCKSTEP=0: the solve after the velocity solve.
      const bool ls_on = !(getenv("LGH_ENERGY_LOCKSTEP") && getenv("LGH_ENERGY_LOCKSTEP")[0] == '0');
      const bool overlap_wanted = !(getenv("LGH_OVERLAP") && getenv("LGH_OVERLAP")[0] == '0');
      if (ls_on && overlap_wanted && c->multi != 0 && !comm_second_channel(c) && !c->timers.enabled && !(c->ktime && c->ktime->which >= 0) && vcg_available(c) &&
          l2_lockstep_possible(c) && vcg_lockstep_ready(c))
      {
         int rc = energy_rhs(c, v_h1, e_rhs); // :473
         if (rc == LGH_OK && e_source) { rc = vec_axpby(c, e_rhs, 1.0, e_rhs, 1.0, e_source, c->L2V); } // :477
         if (rc == LGH_OK) { rc = cg_l2_begin_lockstep(c, e_rhs, dS_dt + 2 * (size_t)c->H1V, rel_tol, max_iter); }
         if (rc) { return rc; }
         // its iterations exchange nothing themselves, so their kernels may run BESIDE K1 / K2 on the second stream (no second
         // communicator needed), ordered by four events per iteration (vcg_solve): LGH_LOCKSTEP_STREAM2=1.  Not the default: the
         // cross-stream waits cost more than the overlap hides (11.0-11.2 against 10.1 ms per step through the N-rank path on one
         // rank, profiles/r6_lockstep.txt)
         c->ls_side = (getenv("LGH_LOCKSTEP_STREAM2") && getenv("LGH_LOCKSTEP_STREAM2")[0] == '1') ? 1 : 0;
         if (c->ls_side)
         {
            if (!c->stream2)
            {
               const int rc2 = create_stream2(c);
               if (rc2) { return rc2; }
            }
            for (auto &ring : c->ls_ev) { for (hipEvent_t &e : ring) { if (!e) { LGH_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); } } }
         }
         c->e_async = 3;
         c->e_lockstep = 1;
         return LGH_OK;
      }
      c->e_async = 2;
      return LGH_OK;
   }
   if (!c->stream2)
   {
      const int rc2 = create_stream2(c);
      if (rc2) { return rc2; }
   }
   LGH_HIP_CHECK(hipEventRecord(c->ev_fork, c->stream));
   LGH_HIP_CHECK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
   std::swap(c->stream, c->stream2); // everything below is enqueued on the second stream
   c->on_stream2 = 1;
   int rc = energy_rhs(c, v_h1, e_rhs); // :473
   if (rc == LGH_OK && e_source) { rc = vec_axpby(c, e_rhs, 1.0, e_rhs, 1.0, e_source, c->L2V); } // :477
   if (rc == LGH_OK)
   {
      RoctxRange range("SolveEnergy-CGEMass"); // laghos_solver.cpp:480-483 (first chunk, second stream)
      rc = cg_l2_begin(c, e_rhs, dS_dt + 2 * (size_t)c->H1V, rel_tol, max_iter); // :481
   }
   std::swap(c->stream, c->stream2);
   c->on_stream2 = 0;
   if (rc) { return rc; }
   c->e_async = 1;
   c->e_polled = 0;
   return LGH_OK;
}
int lgh_energy_lockstep_stats(lgh_ctx *c, long out[4])
{
   LGH_CHECK_ARG(c && out);
   out[0] = c->ls_stats[0]; out[1] = c->ls_stats[1]; out[2] = c->ls_stats[2];
   out[3] = (c->multi != 0 && !comm_second_channel(c) && vcg_available(c) && l2_lockstep_possible(c) && vcg_lockstep_ready(c)) ? 1 : 0;
   return LGH_OK;
}
int lgh_solve_energy_end(lgh_ctx *c, int *l2_iters)
{
   LGH_CHECK_ARG(c && (c->e_async == 1 || c->e_async == 2 || c->e_async == 3));
   const int mode = c->e_async;
   c->e_async = 0;
   if (mode == 3)
   {
      // lockstep: whatever the velocity solve did not interleave (a solve that needs more iterations than last time, or than the
      // velocity solve took) runs now, reducing its scalars itself; then the look
      c->e_lockstep = 0;
      int it = 0;
      const int rc = cg_l2_end_lockstep(c, &it);
      if (rc) { return rc; }
      const int counted = (it == 0) ? 1 : it; // :486
      c->timers.c[1] += counted;
      if (l2_iters) { *l2_iters += counted; }
      return LGH_OK;
   }
   if (mode == 1 && c->e_polled)
   {
      // completed from inside the velocity solve (energy_overlap_poll): only the join is left
      c->e_polled = 0;
      LGH_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
      const int counted = (c->e_iters == 0) ? 1 : c->e_iters; // :486
      c->timers.c[1] += counted;
      if (l2_iters) { *l2_iters += counted; }
      return LGH_OK;
   }
   if (mode == 2)
   {
      return lgh_solve_energy(c, c->e_args.S, c->e_args.v, c->e_args.dS, c->e_args.e_rhs, c->e_args.src,
                              c->e_args.tol, c->e_args.maxit, l2_iters);
   }
   int it = 0;
   std::swap(c->stream, c->stream2);
   c->on_stream2 = 1;
   int rc = cg_l2_end(c, &it);
   if (rc == LGH_OK) { rc = (hipEventRecord(c->ev_join, c->stream) == hipSuccess) ? LGH_OK : LGH_ERR_HIP; }
   std::swap(c->stream, c->stream2);
   c->on_stream2 = 0;
   if (rc) { return rc; }
   LGH_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
   const int counted = (it == 0) ? 1 : it; // :486
   c->timers.c[1] += counted;
   if (l2_iters) { *l2_iters += counted; }
   return LGH_OK;
}

int lgh_vec_set(lgh_ctx *c, double *y, double a, long n) { LGH_CHECK_ARG(c && y); return vec_set(c, y, a, n); }
int lgh_vec_copy(lgh_ctx *c, double *y, const double *x, long n)
{
   LGH_CHECK_ARG(c && y && x);
   LGH_HIP_CHECK(hipMemcpyAsync(y, x, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
   return LGH_OK;
}
int lgh_vec_axpby(lgh_ctx *c, double *z, double a, const double *x, double b, const double *y, long n)
{
   LGH_CHECK_ARG(c && z && x && y);
   return vec_axpby(c, z, a, x, b, y, n);
}
int lgh_vec_axpby_pair(lgh_ctx *c, double *z1, double a1, const double *x1, double b1, double *z2, double a2, const double *x2, double b2,
                       const double *y, long n)
{
   LGH_CHECK_ARG(c && z1 && x1 && z2 && x2 && y && n >= 0);
   // "the same bits as two lgh_vec_axpby calls" only holds when the first result is not an operand of the second: z1 must not
   // overlap z2, x2 or y (round-5 advisor; z1 may be x1, z2 may be x2 - exact aliases, element i is read before it is written)
   {
      auto overlap = [n](const double *p, const double *q) { return p < q + n && q < p + n; };
      LGH_CHECK_ARG(!overlap(z1, z2) && !overlap(z1, x2) && !overlap(z1, y));
      LGH_CHECK_ARG(!overlap(z2, y) && !overlap(z2, x1) && (z2 == x2 || !overlap(z2, x2)) && (z1 == x1 || !overlap(z1, x1)));
   }
   return vec_axpby_pair(c, z1, a1, x1, b1, z2, a2, x2, b2, y, n);
}
int lgh_vec_dot(lgh_ctx *c, const double *x, const double *y, long n, double *result)
{
   LGH_CHECK_ARG(c && x && y && result);
   int rc = vec_dot(c, x, y, nullptr, n, c->scal + 1);
   if (rc) { return rc; }
   LGH_HIP_CHECK(hipMemcpyAsync(c->host_pinned + 1, c->scal + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   *result = c->host_pinned[1];
   return LGH_OK;
}

int lgh_set_velocity_source(lgh_ctx *c, const double *accel_h1)
{
   LGH_CHECK_ARG(c);
   c->accel_src = accel_h1;
   return LGH_OK;
}

int lgh_tg_source_2d(lgh_ctx *c, const double *S, double *e_source)
{
   LGH_CHECK_ARG(c && S && e_source);
   return tg_source_2d(c, S, e_source);
}

int lgh_internal_energy(lgh_ctx *c, const double *e_l2, double *result)
{
   LGH_CHECK_ARG(c && e_l2 && result);
   return interp_energy(c, 0, e_l2, result);
}
int lgh_kinetic_energy(lgh_ctx *c, const double *v_h1, double *result)
{
   LGH_CHECK_ARG(c && v_h1 && result);
   return interp_energy(c, 1, v_h1, result);
}

int lgh_get_timers(lgh_ctx *c, double t[4], long n[3])
{
   LGH_CHECK_ARG(c && t && n);
   for (int i = 0; i < 4; i++) { t[i] = c->timers.t[i]; }
   for (int i = 0; i < 3; i++) { n[i] = c->timers.c[i]; }
   return LGH_OK;
}
int lgh_reset_timers(lgh_ctx *c)
{
   LGH_CHECK_ARG(c);
   for (int i = 0; i < 4; i++) { c->timers.t[i] = 0; }
   for (int i = 0; i < 3; i++) { c->timers.c[i] = 0; }
   return LGH_OK;
}
int lgh_enable_timers(lgh_ctx *c, int on)
{
   LGH_CHECK_ARG(c);
   c->timers.enabled = on != 0;
   return LGH_OK;
}

int lgh_ktime_begin(lgh_ctx *c, int which, int max_samples)
{
   LGH_CHECK_ARG(c && which >= 0 && max_samples > 0);
   if (!c->ktime) { c->ktime = new KTime(); }
   KTime *k = c->ktime;
   while ((int)k->ev.size() < 2 * max_samples)
   {
      hipEvent_t e;
      LGH_HIP_CHECK(hipEventCreate(&e));
      k->ev.push_back(e);
   }
   k->which = which;
   k->max = max_samples;
   k->n = 0;
   return LGH_OK;
}
int lgh_ktime_end(lgh_ctx *c, int *launches, double *mean_seconds)
{
   LGH_CHECK_ARG(c && launches && mean_seconds && c->ktime);
   KTime *k = c->ktime;
   LGH_HIP_CHECK(hipStreamSynchronize(c->stream));
   double tot = 0.0;
   for (int i = 0; i < k->n; i++)
   {
      float ms = 0.f;
      LGH_HIP_CHECK(hipEventElapsedTime(&ms, k->ev[2 * i], k->ev[2 * i + 1]));
      tot += 1e-3 * ms;
   }
   *launches = k->n;
   *mean_seconds = k->n ? tot / k->n : 0.0;
   k->which = -1;
   return LGH_OK;
}

int lgh_set_fused_forces(lgh_ctx *c, int on)
{
   LGH_CHECK_ARG(c);
   if (c->fused_forces_off == (on ? 0 : 1)) { return LGH_OK; } // (nothing changes: the products on hand stay)
   c->fused_forces_off = on ? 0 : 1;
   invalidate_fused(c);
   return LGH_OK;
}

int lgh_get_fused_forces(lgh_ctx *c, int *f1, int *ftv)
{
   LGH_CHECK_ARG(c && f1 && ftv);
   *f1 = (c->force_e_q && !c->fused_forces_off) ? 1 : 0;
   *ftv = (c->erhs_q && !c->fused_forces_off) ? 1 : 0;
   return LGH_OK;
}

int lgh_qupdate_set_tiny_grad(lgh_ctx *c, double tiny_grad)
{
   LGH_CHECK_ARG(c);
   c->q_tiny_grad = tiny_grad;
   return LGH_OK;
}

int lgh_table_symmetry(lgh_ctx *c, int *h1, int *l2)
{
   LGH_CHECK_ARG(c && h1 && l2);
   *h1 = c->b_h1_sym;
   *l2 = c->b_l2_sym;
   return LGH_OK;
}
int lgh_l2_mass_form(lgh_ctx *c, int *form, int *compact)
{
   LGH_CHECK_ARG(c && form && compact);
   return l2_mass_form(c, form, compact);
}
int lgh_k1_form(lgh_ctx *c, int *form)
{
   LGH_CHECK_ARG(c && form);
   *form = vcg_k1_form(c);
   return LGH_OK;
}

int lgh_test_vcg_k1(lgh_ctx *c, const double *r, const double *d_old, const double rz[3], const double rz_prev[3], int first,
                    double *y_E, double den[3])
{
   LGH_CHECK_ARG(c && r && (first || d_old) && rz && rz_prev && y_E && den);
   return vcg_test_k1(c, r, d_old, rz, rz_prev, first, y_E, den);
}
int lgh_jac0inv_form(lgh_ctx *c, int *compact)
{
   LGH_CHECK_ARG(c && compact);
   *compact = (c->jac0_compact == 1) ? 1 : 0;
   return LGH_OK;
}
int lgh_vcg_layout_stats(lgh_ctx *c, long out[4])
{
   LGH_CHECK_ARG(c && out);
   return vcg_layout_stats(c, out);
}
int lgh_test_vcg_merged_faces(lgh_ctx *c, unsigned char *mask, long *n_merged)
{
   LGH_CHECK_ARG(c && mask && n_merged);
   return vcg_test_merged_faces(c, mask, n_merged);
}
int lgh_test_vcg_k2(lgh_ctx *c, int it, const double *y_E, double *r, double *d, double *x, const double den[3], const double rz[3],
                    const double rz_prev[3], const double alpha_prev[3], double rz_out[3], int *deferred_x)
{
   LGH_CHECK_ARG(c && it >= 1 && y_E && r && d && x && den && rz && rz_prev && alpha_prev && rz_out && deferred_x);
   return vcg_test_k2(c, it, y_E, r, d, x, den, rz, rz_prev, alpha_prev, rz_out, deferred_x);
}
int lgh_force_mult_E(lgh_ctx *c, const double *sJit, const double *x_E, double *y_E)
{
   LGH_CHECK_ARG(c && sJit && x_E && y_E);
   return force_mult_E(c, sJit, x_E, y_E);
}
int lgh_force_mult_transpose_E(lgh_ctx *c, const double *sJit, const double *v_E, double *y_E)
{
   LGH_CHECK_ARG(c && sJit && v_E && y_E);
   return force_mult_t_E(c, sJit, v_E, y_E);
}
int lgh_mass_apply_E(lgh_ctx *c, int space, const double *x_E, double *y_E)
{
   LGH_CHECK_ARG(c && x_E && y_E);
   return mass_apply_E(c, space, x_E, y_E);
}
int lgh_test_eig(lgh_ctx *c, int dim, int n, const double *A, double *lambda, double *vec)
{
   LGH_CHECK_ARG(c && A && lambda && vec && (dim == 2 || dim == 3));
   return test_eig(c, dim, n, A, lambda, vec);
}
int lgh_test_sqrt(lgh_ctx *c, int n, const double *x, double *y)
{
   LGH_CHECK_ARG(c && x && y && n >= 0);
   return test_sqrt(c, n, x, y);
}
int lgh_test_singular(lgh_ctx *c, int dim, int n, const double *A, double *sv)
{
   LGH_CHECK_ARG(c && A && sv && (dim == 2 || dim == 3));
   return test_singular(c, dim, n, A, sv);
}

} // extern "C"
