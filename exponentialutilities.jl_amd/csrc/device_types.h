// device_types.h -- scalar types shared by host drivers and gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace expv_mi {

// complex-fp64 as the C ABI lays it out: interleaved (re, im)
struct __attribute__((aligned(16))) cplx {
  double re, im;
};

__host__ __device__ inline cplx make_cplx(double r, double i) { cplx c; c.re = r; c.im = i; return c; }

template <class T> struct ST;  // scalar traits
template <> struct ST<double> {
  static constexpr bool is_complex = false;
  static constexpr int nreal = 1;
  __host__ __device__ static inline double zero() { return 0.0; }
  __host__ __device__ static inline double from_real(double r) { return r; }
  __host__ __device__ static inline double conj(double a) { return a; }
  __host__ __device__ static inline double real(double a) { return a; }
  __host__ __device__ static inline double abs2(double a) { return a * a; }
  // acc += conj(a) * b
  __host__ __device__ static inline void cfma(double &acc, double a, double b) { acc = fma(a, b, acc); }
  // y -= h * v
  __host__ __device__ static inline void nfma(double &y, double h, double v) { y = fma(-h, v, y); }
  // acc += a * b
  __host__ __device__ static inline void fma_(double &acc, double a, double b) { acc = fma(a, b, acc); }
  __host__ __device__ static inline double mul_real(double a, double r) { return a * r; }
  __host__ __device__ static inline double div_real(double a, double r) { return a / r; }
  __host__ __device__ static inline double add(double a, double b) { return a + b; }
  __host__ __device__ static inline double sub(double a, double b) { return a - b; }
  __host__ __device__ static inline double mul(double a, double b) { return a * b; }
  __host__ __device__ static inline double real_only(double a) { return a; }
};
template <> struct ST<cplx> {
  static constexpr bool is_complex = true;
  static constexpr int nreal = 2;
  __host__ __device__ static inline cplx zero() { return make_cplx(0.0, 0.0); }
  __host__ __device__ static inline cplx from_real(double r) { return make_cplx(r, 0.0); }
  __host__ __device__ static inline cplx conj(cplx a) { return make_cplx(a.re, -a.im); }
  __host__ __device__ static inline double real(cplx a) { return a.re; }
  __host__ __device__ static inline double abs2(cplx a) { return fma(a.re, a.re, a.im * a.im); }
  __host__ __device__ static inline void cfma(cplx &acc, cplx a, cplx b) {  // acc += conj(a)*b
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(-a.im, b.re, acc.im);
  }
  __host__ __device__ static inline void nfma(cplx &y, cplx h, cplx v) {  // y -= h*v
    y.re = fma(-h.re, v.re, y.re);
    y.re = fma(h.im, v.im, y.re);
    y.im = fma(-h.re, v.im, y.im);
    y.im = fma(-h.im, v.re, y.im);
  }
  __host__ __device__ static inline void fma_(cplx &acc, cplx a, cplx b) {  // acc += a*b
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(a.im, b.re, acc.im);
  }
  __host__ __device__ static inline cplx mul_real(cplx a, double r) { return make_cplx(a.re * r, a.im * r); }
  __host__ __device__ static inline cplx div_real(cplx a, double r) { return make_cplx(a.re / r, a.im / r); }
  __host__ __device__ static inline cplx add(cplx a, cplx b) { return make_cplx(a.re + b.re, a.im + b.im); }
  __host__ __device__ static inline cplx sub(cplx a, cplx b) { return make_cplx(a.re - b.re, a.im - b.im); }
  __host__ __device__ static inline cplx mul(cplx a, cplx b) {
    return make_cplx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
  }
  __host__ __device__ static inline cplx real_only(cplx a) { return make_cplx(a.re, 0.0); }
};

// device-resident per-subspace step state (one per KrylovSubspace handle)
constexpr int MAX_TICKET_GROUPS = 32, TICKET_STRIDE = 32;   // groups per grid reduction (kernels.h: MAX_GROUPS), words between counters
struct StepState {
  double hnorm;          // beta_j of the last finished step (H[j+1,j])
  double sumsq;          // scratch: last reduced sum of squares
  double beta0sq;        // ||b||^2 of the first step (Ks.beta^2), kept for the host
  double inv;            // 1 / beta of the vector being normalised lazily (single-reduction path)
  int32_t breakdown;     // 1: beta_j < tol (arnoldi.jl:370-374); 2: zero starting vector (arnoldi.jl:366)
  int32_t m_done;        // last step whose column of H is complete
  uint32_t ticket;       // arrival counter of the group reducers (stage 2 of the grid reduction)
  uint32_t pad;
  uint32_t pad1[20];     // -> 128
  // arrival counters of the workgroup groups (stage 1), one per 128-byte line: up to 64 workgroups of a group hit
  // their counter within a few microseconds, and same-line atomics of different groups would queue behind each other
  uint32_t gticket[MAX_TICKET_GROUPS * TICKET_STRIDE];
};
static_assert(sizeof(StepState) == 128 + 4 * MAX_TICKET_GROUPS * TICKET_STRIDE, "StepState layout");

}  // namespace expv_mi
