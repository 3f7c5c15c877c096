// Micro-benchmark (round 6): does a VALU-bound kernel whose workgroups are 216 threads = 3 full wavefronts + 24 lanes (the
// quadrature update: one zone per workgroup, four workgroups per CU) get faster when two zones share a workgroup
// (432 threads = 6 full wavefronts + 48 lanes: 7 wavefronts per two zones instead of 8)?  Only if the dispatcher spreads the
// 14 wavefronts of two resident workgroups evenly over the four SIMDs of a CU.  Both shapes run the same dependent fp64 FMA
// work per thread, hold the registers and the LDS slice of the real kernel (128 VGPRs; 40 KB per zone), and separate their
// phases by workgroup barriers like its stages.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/ubench_wgshape.hip -o /tmp/ubench_wgshape && /tmp/ubench_wgshape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ZONES>
__global__ void __launch_bounds__(216 * ZONES, 4) shape_k(double *out, const double a, const double b, const int phases, const int per_phase)
{
   __shared__ double lds[5007 * ZONES]; // the real kernel's slice per zone
   const int t = threadIdx.x;
   double x[12];
#pragma unroll
   for (int i = 0; i < 12; i++) { x[i] = t + i; }
   lds[t] = x[0];
   for (int p = 0; p < phases; p++)
   {
      for (int k = 0; k < per_phase; k++)
      {
#pragma unroll
         for (int i = 0; i < 12; i++) { x[i] = fma(x[i], a, b); }
      }
      lds[(t * 7 + p) % (5007 * ZONES)] = x[p % 12];
      __syncthreads();
      x[0] += lds[(t * 11 + p) % (5007 * ZONES)];
   }
   double s = 0.0;
#pragma unroll
   for (int i = 0; i < 12; i++) { s += x[i]; }
   if (s == 123.456) { out[blockIdx.x] = s; }
}

template <int ZONES> static float run(const int nzones, double *out, const int phases, const int per_phase)
{
   hipEvent_t e0, e1;
   hipEventCreate(&e0);
   hipEventCreate(&e1);
   for (int w = 0; w < 2; w++) { hipLaunchKernelGGL(shape_k<ZONES>, dim3(nzones / ZONES), dim3(216 * ZONES), 0, 0, out, 0.999, 1e-3, phases, per_phase); }
   hipEventRecord(e0, 0);
   for (int r = 0; r < 5; r++) { hipLaunchKernelGGL(shape_k<ZONES>, dim3(nzones / ZONES), dim3(216 * ZONES), 0, 0, out, 0.999, 1e-3, phases, per_phase); }
   hipEventRecord(e1, 0);
   hipEventSynchronize(e1);
   float ms = 0.f;
   hipEventElapsedTime(&ms, e0, e1);
   return ms / 5.f;
}

int main()
{
   double *out = nullptr;
   hipMalloc((void **)&out, 65536 * sizeof(double));
   const int nzones = 32768;
   for (int per_phase : {8, 16, 32})
   {
      const float t1 = run<1>(nzones, out, 6, per_phase), t2 = run<2>(nzones, out, 6, per_phase);
      int occ1 = 0, occ2 = 0;
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ1, shape_k<1>, 216, 0);
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ2, shape_k<2>, 432, 0);
      printf("6 phases x %2d x 12 FMAs per thread: one zone per workgroup (%d per CU) %.1f us, two zones per workgroup (%d per CU) %.1f us: ratio %.3f (7/8 = 0.875)\n",
             per_phase, occ1, 1e3 * t1, occ2, 1e3 * t2, t2 / t1);
   }
   return 0;
}
