// debug_kernels.hip -- test hook: the wavefront sums of paint_device.h /
// exact_sum.h applied to caller-supplied arrays (one wave per array), so that
// tests can drive sum_exact_fast with adversarial inputs (ties, terms spanning
// many binades, totals next to powers of two) and compare with a literal
// left-to-right sum.
#include "paint_device.h"
#include "exact_sum.h"
#include "launch.h"
#include "common.h"

namespace rl {

template <int S, int MODE>
__global__ void __launch_bounds__(64) sum_kernel(const double *__restrict__ x, int n, double *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float stage[16 * 64];
  (void)stage;
  const int lane = threadIdx.x & 63, q = n / 64, rem = n % 64;  // the kernels' layout over n terms
  const int start = lane * q + (lane < rem ? lane : rem), len = q + (lane < rem ? 1 : 0);
  const double *xb = x + (size_t)blockIdx.x * n;
  double a[S];
#pragma unroll
  for (int i = 0; i < S; i++) a[i] = (i < len) ? xb[start + i] : 0.0;
  const RegTerm<S> t{a};
  const double r = wave_sum<MODE, S>(t, local_sum<S>(t));
  if (lane == 0) out[blockIdx.x] = r;
}

template <int MODE>
static hipError_t launch_sum(const double *x, int n, int batch, double *out, int S, hipStream_t st) {
  switch (S) {
#define RL_CASE(s, t)                                                                         \
  case s:                                                                                     \
    hipLaunchKernelGGL((sum_kernel<s, MODE>), dim3(batch), dim3(64), 0, st, x, n, out); \
    return hipGetLastError();
    RL_FOR_EACH_S(RL_CASE)
#undef RL_CASE
  }
  return hipErrorInvalidValue;
}

}  // namespace rl

extern "C" int rl_debug_wave_sum(const double *x, int n, int batch, int sum_mode, double *out) {
  using namespace rl;
  if (!x || !out || n < 1 || batch < 1) return RL_EINVAL;
  const Layout lay = make_layout(n);
  const int S = choose_S(lay);
  if (!S) {
    set_error("rl_debug_wave_sum: n too large");
    return RL_EINVAL;
  }
  DevBuf dx, dout;
  int rc;
  if ((rc = dx.alloc(sizeof(double) * (size_t)n * batch))) return rc;
  if ((rc = dout.alloc(sizeof(double) * batch))) return rc;
  RL_HIP(hipMemcpy(dx.p, x, sizeof(double) * (size_t)n * batch, hipMemcpyHostToDevice));
  hipError_t e;
  switch (kernel_mode(sum_mode)) {
    case 0: e = launch_sum<0>(dx.as<double>(), n, batch, dout.as<double>(), S, nullptr); break;
    case 1: e = launch_sum<1>(dx.as<double>(), n, batch, dout.as<double>(), S, nullptr); break;
    default: e = launch_sum<2>(dx.as<double>(), n, batch, dout.as<double>(), S, nullptr); break;
  }
  RL_HIP(e);
  RL_HIP(hipDeviceSynchronize());
  RL_HIP(hipMemcpy(out, dout.p, sizeof(double) * batch, hipMemcpyDeviceToHost));
  return RL_OK;
}
