// Micro-benchmark: what the data movement of the lockstep mass apply K1 costs on gfx950 without any arithmetic.
// Mesh of n^3 Q3 elements on a (3n+1)^3 lexicographic node grid (the numbering of the box meshes of this repo), three
// velocity components in three arrays.  Per element: gather 3 x 64 node values (16 rows of 4 consecutive doubles per
// component), write 3 x 64 element values (contiguous), optionally read NQ = 216 quadrature values.
//   mode 0: gather + write     mode 1: gather only     mode 2: write only     mode 3: gather + write + quadrature data
// One wave handles one element per pass: lane = node (64 nodes), loop over the 3 components.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench_k1_floor.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE, int UNROLL>
__global__ void __launch_bounds__(256) floor_k(const int n, const int NE, const double *__restrict__ d, const size_t N, double *__restrict__ ye,
                                               const double *__restrict__ dq, double *sink)
{
   const int lane = threadIdx.x & 63;
   const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = (gridDim.x * blockDim.x) >> 6;
   const int g = 3 * n + 1;
   const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
   double acc = 0.0;
   for (int e0 = wave * UNROLL; e0 < NE; e0 += nwave * UNROLL)
   {
      double v[UNROLL][3];
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
      {
         const int e = min(e0 + u, NE - 1);
         const int ex = e % n, ey = (e / n) % n, ez = e / (n * n);
         const size_t node = (size_t)(3 * ex + lx) + (size_t)g * ((3 * ey + ly) + (size_t)g * (3 * ez + lz));
#pragma unroll
         for (int c = 0; c < 3; c++) { v[u][c] = (MODE == 2) ? (double)lane : d[c * N + node]; }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
      {
         const int e = min(e0 + u, NE - 1);
         if (MODE == 3)
         {
            const double *p = dq + (size_t)e * 216;
            acc += p[lane] + p[64 + lane] + p[128 + lane] + ((lane < 24) ? p[192 + lane] : 0.0);
         }
#pragma unroll
         for (int c = 0; c < 3; c++)
         {
            if (MODE == 1) { acc += v[u][c]; }
            else { ye[((size_t)e * 3 + c) * 64 + lane] = v[u][c] + acc; }
         }
      }
   }
   if (MODE == 1 || MODE == 3) { if (acc == 12345.678) { sink[0] = acc; } }
}

int main(int argc, char **argv)
{
   const int n = argc > 1 ? atoi(argv[1]) : 32;
   const int NE = n * n * n;
   const size_t g = 3 * n + 1, N = g * g * g;
   double *d, *ye, *dq, *sink;
   (void)hipMalloc((void **)&d, 3 * N * 8);
   (void)hipMalloc((void **)&ye, (size_t)NE * 192 * 8);
   (void)hipMalloc((void **)&dq, (size_t)NE * 216 * 8);
   (void)hipMalloc((void **)&sink, 8);
   (void)hipMemset(d, 0, 3 * N * 8);
   (void)hipMemset(dq, 0, (size_t)NE * 216 * 8);
   hipEvent_t a, b;
   (void)hipEventCreate(&a);
   (void)hipEventCreate(&b);
   const char *name[] = {"gather + write", "gather only", "write only", "gather + write + quadrature data"};
   const double bytes[] = {8.0 * 384, 8.0 * 192, 8.0 * 192, 8.0 * 600};
   for (int wgs_per_cu : {4, 8})
   {
      for (int mode = 0; mode < 4; mode++)
      {
         for (int unroll : {1, 4})
         {
            const int grid = 256 * wgs_per_cu;
            float best = 1e30f;
            for (int rep = 0; rep < 12; rep++)
            {
               (void)hipEventRecord(a, 0);
#define LAUNCH(M, U) hipLaunchKernelGGL((floor_k<M, U>), dim3(grid), dim3(256), 0, 0, n, NE, d, N, ye, dq, sink)
               if (unroll == 1) { if (mode == 0) { LAUNCH(0, 1); } else if (mode == 1) { LAUNCH(1, 1); } else if (mode == 2) { LAUNCH(2, 1); } else { LAUNCH(3, 1); } }
               else { if (mode == 0) { LAUNCH(0, 4); } else if (mode == 1) { LAUNCH(1, 4); } else if (mode == 2) { LAUNCH(2, 4); } else { LAUNCH(3, 4); } }
               (void)hipEventRecord(b, 0);
               (void)hipEventSynchronize(b);
               float ms;
               (void)hipEventElapsedTime(&ms, a, b);
               if (rep >= 2 && ms < best) { best = ms; }
            }
            printf("n=%d  %-34s wgs/cu=%d unroll=%d  %8.1f us   %7.1f GB/s (element-level bytes)\n", n, name[mode], wgs_per_cu, unroll, best * 1e3,
                   bytes[mode] * NE / (best * 1e-3) * 1e-9);
         }
      }
   }
   return 0;
}
