// Round 4 micro-benchmark: what one host "look" at a device flag costs between two dependent kernels.
//   A: kernel -> hipMemcpyAsync D2H (pinned) -> hipStreamSynchronize -> next kernel       (what the CG loops do)
//   B: kernel writes the flag into mapped pinned host memory, host spins on it -> next kernel
//   C: no look: kernel -> kernel (the floor)
// Reported: wall time per (work kernel + look + next kernel) pair minus nothing; compare the three.
// build: hipcc --offload-arch=gfx950 -O2 look_latency.hip -o look_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void work_k(double *p, int n, int *flag, int seq)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) { p[i] = p[i] * 1.0000001 + 1e-9; }
   if (i == 0) { *flag = seq; }
}
__global__ void publish_k(const int *flag, volatile int *host_flag)
{
   *host_flag = *flag;
   __threadfence_system();
}
int main()
{
   const int n = 1 << 22; // ~10 us of work
   double *p; int *flag; CK(hipMalloc(&p, n * 8)); CK(hipMalloc(&flag, 64)); CK(hipMemset(p, 0, n * 8));
   int *hp; CK(hipHostMalloc(&hp, 64, hipHostMallocDefault));
   volatile int *hm; CK(hipHostMalloc((void **)&hm, 64, hipHostMallocMapped)); int *hm_dev; CK(hipHostGetDevicePointer((void **)&hm_dev, (void *)hm, 0));
   hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
   const int reps = 2000;
   auto run = [&](int mode) {
      CK(hipStreamSynchronize(s));
      const auto t0 = std::chrono::steady_clock::now();
      for (int r = 1; r <= reps; r++)
      {
         hipLaunchKernelGGL(work_k, dim3(n / 256), dim3(256), 0, s, p, n, flag, r);
         if (mode == 0) { CK(hipMemcpyAsync(hp, flag, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); if (*hp != r) { printf("bad\n"); } }
         if (mode == 1) { hipLaunchKernelGGL(publish_k, dim3(1), dim3(1), 0, s, flag, hm_dev); while (*hm != r) { } }
         if (mode == 3) { CK(hipStreamSynchronize(s)); }
      }
      CK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      return us;
   };
   for (int rep = 0; rep < 3; rep++)
   {
      const double c = run(2), a = run(0), b = run(1), d = run(3);
      printf("per kernel: no look %.2f us | memcpy+sync %.2f us (+%.2f) | mapped flag + spin %.2f us (+%.2f) | sync only %.2f us (+%.2f)\n", c, a, a - c, b, b - c, d, d - c);
   }
   return 0;
}
