// A compiled matrix-free operator for the library's callback contract (include/expv_mi.h: expv_mi_matvec_fn; the reference's operator
// interface, docs/src/interfaces.md:7-36, test/basictests.jl:786-816): y = A x for a constant-coefficient banded stencil as ONE HIP
// kernel on the stream the library hands over.  Test / bench infrastructure: it stands for the user's own mul! (a Julia host would pass
// a @cfunction that launches its kernel); nothing in exponentialutilities.jl_amd/ loads it.  bench.py's `matrix_free_compiled` entry
// uses it to show the LIBRARY's share of a matrix-free Krylov step (the Python callback of `matrix_free_callback` costs more than the step).
#include <hip/hip_runtime.h>
#include <stdint.h>

struct StencilOp {      // `user` of the callback
  int64_t n;
  int ndiag;
  int off[8];
  double coef[8];
};

__global__ __launch_bounds__(256) void k_stencil(StencilOp op, const double *__restrict__ x, double *__restrict__ y) {
  // two rows per lane, 16-byte stores; the neighbours come from L1 / L2 (every x element is read once from HBM)
  const int64_t i = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= op.n) return;
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    if (d < op.ndiag) {
      const int64_t j0 = i + op.off[d], j1 = j0 + 1;
      if (j0 >= 0 && j0 < op.n) a0 = fma(op.coef[d], x[j0], a0);
      if (j1 >= 0 && j1 < op.n && i + 1 < op.n) a1 = fma(op.coef[d], x[j1], a1);
    }
  }
  if (i + 1 < op.n) *reinterpret_cast<double2 *>(y + i) = make_double2(a0, a1);
  else y[i] = a0;
}

extern "C" int stencil_matvec(void *user, const void *x_dev, void *y_dev, void *hip_stream) {
  const StencilOp *op = static_cast<const StencilOp *>(user);
  const int64_t pairs = (op->n + 1) / 2;
  const int grid = (int)((pairs + 255) / 256);
  hipLaunchKernelGGL(k_stencil, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(hip_stream), *op,
                     static_cast<const double *>(x_dev), static_cast<double *>(y_dev));
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
extern "C" int stencil_op_sizeof(void) { return (int)sizeof(StencilOp); }
