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

// complex-fp32, interleaved (re, im): ComplexF32
struct __attribute__((aligned(8))) cplx32 {
  float re, im;
};
__host__ __device__ inline cplx32 make_cplx32(float r, float i) { cplx32 c; c.re = r; c.im = i; return c; }

// Scalar traits.  acc_t is the type projection sums are accumulated in: the 32-bit element types (half the HBM traffic of the
// 64-bit ones: BlasFloat of ExponentialUtilities.jl:19) store and update in fp32 but sum their dot products in fp64 -- the
// products of two fp32 numbers are exact in fp64, so a sum over 1e6 rows carries fp64 rounding only.
template <class T> struct ST;  // scalar traits
template <> struct ST<double> {
  using acc_t = double;
  using real_t = double;
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
  using acc_t = cplx;
  using real_t = double;
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

template <> struct ST<float> {
  using acc_t = double;
  using real_t = float;
  static constexpr bool is_complex = false;
  static constexpr int nreal = 1;
  __host__ __device__ static inline float zero() { return 0.0f; }
  __host__ __device__ static inline float from_real(double r) { return (float)r; }
  __host__ __device__ static inline float conj(float a) { return a; }
  __host__ __device__ static inline double real(float a) { return (double)a; }
  __host__ __device__ static inline double abs2(float a) { return (double)a * (double)a; }
  __host__ __device__ static inline void cfma(double &acc, float a, float b) { acc = fma((double)a, (double)b, acc); }
  __host__ __device__ static inline void nfma(float &y, float h, float v) { y = fmaf(-h, v, y); }
  __host__ __device__ static inline void fma_(float &acc, float a, float b) { acc = fmaf(a, b, acc); }
  __host__ __device__ static inline float mul_real(float a, double r) { return a * (float)r; }
  __host__ __device__ static inline float div_real(float a, double r) { return a / (float)r; }
  __host__ __device__ static inline float add(float a, float b) { return a + b; }
  __host__ __device__ static inline float sub(float a, float b) { return a - b; }
  __host__ __device__ static inline float mul(float a, float b) { return a * b; }
  __host__ __device__ static inline float real_only(float a) { return a; }
};
template <> struct ST<cplx32> {
  using acc_t = cplx;
  using real_t = float;
  static constexpr bool is_complex = true;
  static constexpr int nreal = 2;
  __host__ __device__ static inline cplx32 zero() { return make_cplx32(0.0f, 0.0f); }
  __host__ __device__ static inline cplx32 from_real(double r) { return make_cplx32((float)r, 0.0f); }
  __host__ __device__ static inline cplx32 conj(cplx32 a) { return make_cplx32(a.re, -a.im); }
  __host__ __device__ static inline double real(cplx32 a) { return (double)a.re; }
  __host__ __device__ static inline double abs2(cplx32 a) { return fma((double)a.re, (double)a.re, (double)a.im * (double)a.im); }
  __host__ __device__ static inline void cfma(cplx &acc, cplx32 a, cplx32 b) {  // acc += conj(a)*b, summed in fp64
    acc.re = fma((double)a.re, (double)b.re, acc.re);
    acc.re = fma((double)a.im, (double)b.im, acc.re);
    acc.im = fma((double)a.re, (double)b.im, acc.im);
    acc.im = fma(-(double)a.im, (double)b.re, acc.im);
  }
  __host__ __device__ static inline void nfma(cplx32 &y, cplx32 h, cplx32 v) {  // y -= h*v
    y.re = fmaf(-h.re, v.re, y.re);
    y.re = fmaf(h.im, v.im, y.re);
    y.im = fmaf(-h.re, v.im, y.im);
    y.im = fmaf(-h.im, v.re, y.im);
  }
  __host__ __device__ static inline void fma_(cplx32 &acc, cplx32 a, cplx32 b) {  // acc += a*b
    acc.re = fmaf(a.re, b.re, acc.re);
    acc.re = fmaf(-a.im, b.im, acc.re);
    acc.im = fmaf(a.re, b.im, acc.im);
    acc.im = fmaf(a.im, b.re, acc.im);
  }
  __host__ __device__ static inline cplx32 mul_real(cplx32 a, double r) { return make_cplx32(a.re * (float)r, a.im * (float)r); }
  __host__ __device__ static inline cplx32 div_real(cplx32 a, double r) { return make_cplx32(a.re / (float)r, a.im / (float)r); }
  __host__ __device__ static inline cplx32 add(cplx32 a, cplx32 b) { return make_cplx32(a.re + b.re, a.im + b.im); }
  __host__ __device__ static inline cplx32 sub(cplx32 a, cplx32 b) { return make_cplx32(a.re - b.re, a.im - b.im); }
  __host__ __device__ static inline cplx32 mul(cplx32 a, cplx32 b) {
    return make_cplx32(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
  }
  __host__ __device__ static inline cplx32 real_only(cplx32 a) { return make_cplx32(a.re, 0.0f); }
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
