// Micro-benchmarks behind the design of lgh_vcg_mfma.hip (one wavefront per SIMD): issue and dependent latency of
// fp64 FMA and of v_mfma_f64_16x16x4_f64 on gfx950, their overlap, and the cost of AGPR <-> VGPR moves.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/ubench_f64.hip -o /tmp/ubench_f64 && /tmp/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__device__ __forceinline__ unsigned long long clk() { return clock64(); }

// which: test id; out[0] = cycles, out[1] = instructions of the kind measured
__global__ void ub(int which, unsigned long long *out, double *sink, double a, double b)
{
   double x0 = threadIdx.x, x1 = 1.0 + threadIdx.x, x2 = 2.0, x3 = 3.0, x4 = 4.0, x5 = 5.0, x6 = 6.0, x7 = 7.0;
   v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
   unsigned long long t0 = 0, t1 = 0;
   long n = 0;
   if (which == 0) // one dependent chain of v_fma_f64
   {
      t0 = clk();
      for (int i = 0; i < 16; i++) { asm volatile(REP64("v_fma_f64 %0, %1, %2, %0\n") : "+v"(x0) : "v"(a), "v"(b)); }
      t1 = clk(); n = 16 * 64;
   }
   else if (which == 1) // two independent chains
   {
      t0 = clk();
      for (int i = 0; i < 16; i++) { asm volatile(REP64("v_fma_f64 %0, %2, %3, %0\nv_fma_f64 %1, %2, %3, %1\n") : "+v"(x0), "+v"(x1) : "v"(a), "v"(b)); }
      t1 = clk(); n = 16 * 128;
   }
   else if (which == 2) // four independent chains
   {
      t0 = clk();
      for (int i = 0; i < 16; i++)
      {
         asm volatile(REP64("v_fma_f64 %0, %4, %5, %0\nv_fma_f64 %1, %4, %5, %1\nv_fma_f64 %2, %4, %5, %2\nv_fma_f64 %3, %4, %5, %3\n")
                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
      }
      t1 = clk(); n = 16 * 256;
   }
   else if (which == 3) // three independent chains
   {
      t0 = clk();
      for (int i = 0; i < 16; i++)
      {
         asm volatile(REP64("v_fma_f64 %0, %3, %4, %0\nv_fma_f64 %1, %3, %4, %1\nv_fma_f64 %2, %3, %4, %2\n")
                      : "+v"(x0), "+v"(x1), "+v"(x2) : "v"(a), "v"(b));
      }
      t1 = clk(); n = 16 * 192;
   }
   else if (which == 4) // independent MFMAs (four accumulators round-robin: each is dependent on itself 4 issues back)
   {
      t0 = clk();
      for (int i = 0; i < 16; i++)
      {
         asm volatile(REP16("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\nv_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n"
                            "v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\nv_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n")
                      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
      }
      t1 = clk(); n = 16 * 64;
   }
   else if (which == 5) // one dependent MFMA chain
   {
      t0 = clk();
      for (int i = 0; i < 16; i++) { asm volatile(REP64("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n") : "+v"(c0) : "v"(a), "v"(b)); }
      t1 = clk(); n = 16 * 64;
   }
   else if (which == 6 || which == 7 || which == 8) // one MFMA + K independent FMAs (4 chains): overlap
   {
      t0 = clk();
      for (int i = 0; i < 16; i++)
      {
         if (which == 6)
         {
            asm volatile(REP16("v_mfma_f64_16x16x4_f64 %4, %6, %7, %4\n" REP4("v_fma_f64 %0, %6, %7, %0\nv_fma_f64 %1, %6, %7, %1\nv_fma_f64 %2, %6, %7, %2\nv_fma_f64 %3, %6, %7, %3\n")
                               "v_mfma_f64_16x16x4_f64 %5, %6, %7, %5\n" REP4("v_fma_f64 %0, %6, %7, %0\nv_fma_f64 %1, %6, %7, %1\nv_fma_f64 %2, %6, %7, %2\nv_fma_f64 %3, %6, %7, %3\n"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
         }
         else if (which == 7)
         {
            asm volatile(REP16("v_mfma_f64_16x16x4_f64 %4, %6, %7, %4\n" REP4(REP4("v_fma_f64 %0, %6, %7, %0\nv_fma_f64 %1, %6, %7, %1\n"))
                               "v_mfma_f64_16x16x4_f64 %5, %6, %7, %5\n" REP4(REP4("v_fma_f64 %2, %6, %7, %2\nv_fma_f64 %3, %6, %7, %3\n")))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
         }
         else
         {
            asm volatile(REP16("v_mfma_f64_16x16x4_f64 %4, %6, %7, %4\n" REP4("v_fma_f64 %0, %6, %7, %0\nv_fma_f64 %1, %6, %7, %1\n")
                               "v_mfma_f64_16x16x4_f64 %5, %6, %7, %5\n" REP4("v_fma_f64 %2, %6, %7, %2\nv_fma_f64 %3, %6, %7, %3\n"))
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
         }
      }
      t1 = clk(); n = 16 * 32; // MFMAs; FMAs per MFMA: 16 (which 6), 32 (7), 8 (8)
   }
   else if (which == 9) // v_accvgpr_write + v_accvgpr_read pairs
   {
      int r = threadIdx.x;
      t0 = clk();
      for (int i = 0; i < 16; i++) { asm volatile(REP64("v_accvgpr_write_b32 a0, %0\nv_accvgpr_read_b32 %0, a0\n") : "+v"(r) : : "a0"); }
      t1 = clk(); n = 16 * 128;
      x0 += r;
   }
   else if (which == 10) // independent v_mov_b32 (8 registers)
   {
      int r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3;
      t0 = clk();
      for (int i = 0; i < 16; i++) { asm volatile(REP64("v_mov_b32 %0, %2\nv_mov_b32 %1, %3\nv_mov_b32 %2, %0\nv_mov_b32 %3, %1\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)); }
      t1 = clk(); n = 16 * 256;
      x0 += r0 + r1 + r2 + r3;
   }
   else if (which == 11) // FMA with an SGPR operand, four chains
   {
      const double sa = __builtin_bit_cast(double, (long)__builtin_amdgcn_readfirstlane((int)threadIdx.x) + 0x3ff0000000000000L);
      t0 = clk();
      for (int i = 0; i < 16; i++)
      {
         asm volatile(REP64("v_fma_f64 %0, %4, %5, %0\nv_fma_f64 %1, %4, %5, %1\nv_fma_f64 %2, %4, %5, %2\nv_fma_f64 %3, %4, %5, %3\n")
                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(sa), "v"(b));
      }
      t1 = clk(); n = 16 * 256;
   }
   else if (which == 12 || which == 13) // v_permlane32_swap / v_permlane16_swap, four independent register pairs
   {
      unsigned r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
      t0 = clk();
      for (int i = 0; i < 16; i++)
      {
         if (which == 12) { asm volatile(REP64("v_permlane32_swap_b32 %0, %1\nv_permlane32_swap_b32 %2, %3\nv_permlane32_swap_b32 %4, %5\nv_permlane32_swap_b32 %6, %7\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)); }
         else { asm volatile(REP64("v_permlane16_swap_b32 %0, %1\nv_permlane16_swap_b32 %2, %3\nv_permlane16_swap_b32 %4, %5\nv_permlane16_swap_b32 %6, %7\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)); }
      }
      t1 = clk(); n = 16 * 256;
      x0 += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
   }
   else if (which == 14) // swap followed by a dependent fma on the swapped register (hazard / latency)
   {
      unsigned p0 = threadIdx.x, p1 = 7;
      t0 = clk();
      for (int i = 0; i < 16; i++)
      {
         asm volatile(REP64("v_permlane32_swap_b32 %0, %1\nv_fma_f64 %2, %4, %5, %2\nv_add_u32 %0, %0, %1\nv_fma_f64 %3, %4, %5, %3\n") : "+v"(p0), "+v"(p1), "+v"(x0), "+v"(x1) : "v"(a), "v"(b));
      }
      t1 = clk(); n = 16 * 64;
      x0 += p0 + p1;
   }
   else if (which == 15) // dependent ds_read_b64 chain (LDS latency of one wave)
   {
      __shared__ double lds[512];
      for (int i = threadIdx.x; i < 512; i += blockDim.x) { lds[i] = 0.0; }
      __syncthreads();
      unsigned addr = 8 * (threadIdx.x & 63);
      unsigned v = 0;
      t0 = clk();
      for (int i = 0; i < 16; i++)
      {
         asm volatile(REP16("ds_read_b32 %1, %0\ns_waitcnt lgkmcnt(0)\nv_add_u32 %0, %0, %1\n") : "+v"(addr), "+v"(v));
      }
      t1 = clk(); n = 16 * 16;
      x0 += v + addr;
      if (threadIdx.x == 1000) { lds[0] = v; }
   }
   if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = n; }
   sink[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0[0] + c1[1] + c2[2] + c3[3];
}

int main()
{
   unsigned long long *out;
   double *sink;
   (void)hipMalloc((void **)&out, 16);
   (void)hipMalloc((void **)&sink, 8 * 1024 * 1024);
   const char *name[] = {"fma_f64 1 chain", "fma_f64 2 chains", "fma_f64 4 chains", "fma_f64 3 chains", "mfma_f64 4 accumulators", "mfma_f64 1 chain",
                         "mfma + 16 fma (4 chains)", "mfma + 32 fma (2 chains)", "mfma + 8 fma", "accvgpr write+read", "v_mov_b32 x4 indep",
                         "fma_f64 sgpr operand 4 chains", "permlane32_swap x4", "permlane16_swap x4", "swap + fma + add + fma (per group)", "ds_read_b64 dependent (latency)"};
   for (int blocks = 1; blocks <= 1024; blocks *= 1024)
   {
      for (int w = (blocks == 1 ? 0 : 12); w < 16; w++)
      {
         for (int threads = 64; threads <= 512; threads *= 2)
         {
            if (threads == 128) { continue; }
            unsigned long long h[2] = {0, 0};
            hipLaunchKernelGGL(ub, dim3(blocks), dim3(threads), 0, 0, w, out, sink, 1.0000001, 1e-9);
            hipLaunchKernelGGL(ub, dim3(blocks), dim3(threads), 0, 0, w, out, sink, 1.0000001, 1e-9);
            (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("blocks %4d threads %3d  %-42s %8llu cycles / %6llu = %6.2f cycles each\n", blocks, threads, name[w], h[0], h[1], (double)h[0] / (double)h[1]);
         }
      }
   }
   return 0;
}
