// Micro-benchmark: cost of one wave-wide 8-byte (or 16-byte) gather instruction on gfx950 as a function of how the
// 64 lane addresses are laid out - what the texture-address unit of a CU coalesces and what it does not.
// One workgroup of 256 threads per CU, every wave issues NLOAD loads per pass with 16 in flight; data resident in L2.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/ubench_gather.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// pattern p: byte offset of lane l within a 16 KB window
__device__ __forceinline__ unsigned lane_off(int p, int l)
{
   switch (p)
   {
      case 0: return 8u * l;                                   // 64 consecutive doubles (512 B)
      case 1: return 256u * (l >> 2) + 8u * (l & 3);           // quads of 4 consecutive doubles, one quad per 256 B
      case 2: return 256u * l;                                 // every lane its own line
      case 3: return 24u * (l & 15) + 2048u * (l >> 4);        // 16 adjacent lanes 24 B apart (x-neighbour elements), 4 far groups
      case 4: return 24u * ((l & 15) % 5) + 4096u * ((l & 15) / 5) + 1024u * (l >> 4); // slab form: 5 elements x 3 far arrays x 4 far groups
      case 5: return 8u * (l >> 4) + 24u * ((l & 15) % 5) + 4096u * ((l & 15) / 5);    // matrix-core form: dx in lane >> 4
      case 6: return 128u * (l >> 2) + 8u * (l & 3);           // quads, one quad per 128 B line
      case 7: return 64u * (l >> 2) + 8u * (l & 3);            // quads, one per 64 B
      case 8: return 32u * (l >> 2) + 8u * (l & 3);            // quads back to back = consecutive (as 0)
      case 9: return 128u * (l >> 1) + 8u * (l & 1);           // pairs, one pair per line
      case 10: return 8u * ((l & 15) % 5 * 3) + 4096u * ((l & 15) / 5) + 1024u * (l >> 4); // as 4, written differently
      case 11: return 16u * l;                                 // (16-byte loads) consecutive
      case 12: return 128u * l;                                // (16-byte loads) one line per lane, 16 B each
      default: return 0;
   }
}

template <int BYTES>
__global__ void __launch_bounds__(256) gather_k(const char *base, int p, int npass, unsigned long long *out, double *sink)
{
   const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
   const unsigned lo = lane_off(p, lane);
   const char *b = base + (size_t)blockIdx.x * (1u << 20) + (size_t)wid * (1u << 18);
   double acc = 0.0;
   unsigned long long t0 = clock64();
   for (int pass = 0; pass < npass; pass++)
   {
#pragma unroll
      for (int k = 0; k < 16; k++)
      {
         const char *q = b + ((pass * 16 + k) & 15) * 16384u + lo;
         if (BYTES == 8) { acc += *(const double *)q; }
         else { const double2 v = *(const double2 *)q; acc += v.x + v.y; }
      }
   }
   unsigned long long t1 = clock64();
   if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
   sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
   const int nb = 256;
   char *buf;
   unsigned long long *out;
   double *sink;
   (void)hipMalloc((void **)&buf, (size_t)nb << 20);
   (void)hipMemset(buf, 0, (size_t)nb << 20);
   (void)hipMalloc((void **)&out, nb * 8);
   (void)hipMalloc((void **)&sink, nb * 256 * 8);
   const char *name[] = {"64 consecutive doubles", "quads (32 B), one per 256 B", "one line per lane", "16 lanes 24 B apart x 4 groups",
                         "slab form (5 elem x 3 arrays x 4 slabs)", "matrix-core form (dx = lane >> 4)", "quads, one per 128 B", "quads, one per 64 B",
                         "quads back to back", "pairs, one per line", "slab form (b)", "16-byte loads, consecutive", "16-byte loads, one line per lane"};
   const int npass = 64;
   for (int p = 0; p < 13; p++)
   {
      std::vector<unsigned long long> h(nb);
      for (int rep = 0; rep < 2; rep++)
      {
         if (p >= 11) { hipLaunchKernelGGL(gather_k<16>, dim3(nb), dim3(256), 0, 0, buf, p, npass, out, sink); }
         else { hipLaunchKernelGGL(gather_k<8>, dim3(nb), dim3(256), 0, 0, buf, p, npass, out, sink); }
      }
      (void)hipMemcpy(h.data(), out, nb * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (int i = 0; i < nb; i++) { s += (double)h[i]; }
      printf("%-44s %8.1f cycles per wave-load (4 waves per CU issuing: %6.1f per CU-load)\n", name[p], s / nb / (npass * 16), s / nb / (npass * 16) / 4);
   }
   return 0;
}
