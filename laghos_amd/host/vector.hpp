// vector.hpp — device-resident Vector, the stand-in for mfem::Vector in the C++
// host layer.  Data lives in HBM (hipMalloc); the reference's Vector::Read/Write
// device pointers (e.g. laghos_assembly.cpp:304-312) become Vector::Read()/Write().
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace laghos
{

#define LAGHOS_HIP(expr)                                                                   \
   do                                                                                      \
   {                                                                                       \
      hipError_t e_ = (expr);                                                              \
      if (e_ != hipSuccess)                                                                \
      {                                                                                    \
         std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
         std::abort();                                                                     \
      }                                                                                    \
   } while (0)

class Vector
{
   double *d_ = nullptr;
   long n_ = 0;
   bool own_ = false;

public:
   Vector() {}
   explicit Vector(long n) { SetSize(n); }
   Vector(const Vector &) = delete;
   Vector &operator=(const Vector &) = delete;
   Vector(Vector &&o) noexcept : d_(o.d_), n_(o.n_), own_(o.own_) { o.d_ = nullptr; o.own_ = false; o.n_ = 0; }
   ~Vector() { Destroy(); }
   void Destroy()
   {
      if (own_ && d_) { (void)hipFree(d_); }
      d_ = nullptr;
      n_ = 0;
      own_ = false;
   }
   void SetSize(long n)
   {
      Destroy();
      n_ = n;
      own_ = true;
      LAGHOS_HIP(hipMalloc((void **)&d_, (n > 0 ? n : 1) * sizeof(double)));
      LAGHOS_HIP(hipMemset(d_, 0, (n > 0 ? n : 1) * sizeof(double)));
      // the fill runs asynchronously on the null stream; the library's stream is non-blocking
      LAGHOS_HIP(hipStreamSynchronize(nullptr));
   }
   // non-owning view of a sub-range (ParGridFunction::MakeRef, laghos_solver.cpp:319)
   void MakeRef(Vector &base, long offset, long n)
   {
      Destroy();
      d_ = base.d_ + offset;
      n_ = n;
      own_ = false;
   }
   void MakeRef(double *p, long n)
   {
      Destroy();
      d_ = p;
      n_ = n;
      own_ = false;
   }
   long Size() const { return n_; }
   const double *Read() const { return d_; }
   double *Write() { return d_; }
   double *ReadWrite() { return d_; }
   void FromHost(const std::vector<double> &h)
   {
      if ((long)h.size() != n_) { SetSize((long)h.size()); }
      LAGHOS_HIP(hipMemcpy(d_, h.data(), n_ * sizeof(double), hipMemcpyHostToDevice));
   }
   void ToHost(std::vector<double> &h) const
   {
      h.resize(n_);
      LAGHOS_HIP(hipMemcpy(h.data(), d_, n_ * sizeof(double), hipMemcpyDeviceToHost));
   }
};

} // namespace laghos
