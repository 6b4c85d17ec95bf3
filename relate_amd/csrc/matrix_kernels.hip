// matrix_kernels.hip -- K3: the N x N distance matrix at one SNP.
//
// Replaces the row loop of DistanceMeasure::GetMatrix (anc_builder.cpp:116-194).
// One 256-thread block per target row n: reads one (derived at snp) or two
// (interpolated) posterior rows of target n in the rows' register-major order, applies fast_log (fast_log.hpp),
// reduces the row minimum over all N computed entries (the diagonal's
// fast_log(0) value included), writes d[n][j] - min (diagonal 0).
// Traffic: <= 8N^2 B read + 4N^2 B written per call: HBM-bound.
#include "paint_device.h"
#include "launch.h"

namespace rl {

__global__ void __launch_bounds__(256) matrix_kernel(const MatrixParams p, const Layout lay, int S, int waves) {
  extern __shared__ float vals[];  // N floats
  __shared__ float red[256];
  const int t = blockIdx.x;   // row of this context
  const int n = p.k0 + t;     // its target
  const int N = p.N;
  const int64_t stride = (int64_t)S * 64 * waves;
  const MatrixArg arg = p.args[t];
  const int64_t r0 = p.top_off[t] + arg.v_snp_prev;
  const float *__restrict__ tp = p.topology + (p.slab_base[t] + arg.v_snp_prev) * stride;
  const float *__restrict__ tn = tp + stride;
  const float ls_prev = p.logscales[r0];
  const bool direct = arg.direct != 0;
  const float ls_next = direct ? 0.0f : p.logscales[r0 + 1];
  const double wl = arg.wl, wr = arg.wr;
  const float e_pn = arg.e_pn, e_np = arg.e_np;
  const float scale = -1.0f;

  // The posterior rows are read in their own (register-major) order: thread = lane, four registers per pass, every
  // wavefront 256 contiguous bytes; the entry of donor j = start(lane) + register lands in LDS in donor order.
  float mn = INFINITY;
  const int lane = threadIdx.x & 63;
  for (int w = 0; w < waves; w++) {
    const int l = w * 64 + lane;                             // virtual lane: a run of consecutive donors
    const int len = lay.q + (l < lay.rem ? 1 : 0);
    const int start = l * lay.q + (l < lay.rem ? l : lay.rem);
    for (int i = threadIdx.x >> 6; i < len; i += blockDim.x >> 6) {
      const int idx = (w * S + i) * 64 + lane;
      float v;
      if (direct) {
        const float x = tp[idx];
        v = (fast_log_dev(x) + ls_prev) * scale;  // :128
      } else {
        const float xp = tp[idx], xn = tn[idx];
        if (ls_prev <= ls_next) {  // :172-178
          const float x = (float)(wl * xp * e_pn + wr * xn);
          v = (fast_log_dev(x) + ls_next) * scale;
        } else {
          const float x = (float)(wl * xp + wr * xn * e_np);
          v = (fast_log_dev(x) + ls_prev) * scale;
        }
      }
      vals[start + i] = v;
      if (v < mn) mn = v;
    }
  }
  red[threadIdx.x] = mn;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float o = red[threadIdx.x + s];
      if (o < red[threadIdx.x]) red[threadIdx.x] = o;
    }
    __syncthreads();
  }
  mn = red[0];
  float *__restrict__ out = p.matrix + (size_t)t * N;
  if (!p.member && !p.rowmin) {
    for (int j = threadIdx.x; j < N; j += blockDim.x) out[j] = (j == n) ? 0.0f : vals[j] - mn;  // :190-192
    return;
  }
  // ... and, while the row is here, the carrier penalty (the same two operations per entry, in the same order, as the
  // loop of anc_builder.cpp:567-574 performs on the finished matrix) and the row's minimum off the diagonal
  const bool carrier = p.member && p.member[n];
  float rm = INFINITY;
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    float x = (j == n) ? 0.0f : vals[j] - mn;
    if (carrier) {
      x = x + p.val;
      if (p.member[j]) x -= p.val;
    }
    out[j] = x;
    if (j != n) rm = fminf(rm, x);
  }
  if (!p.rowmin) return;
  __syncthreads();  // (red is read above by every thread)
  red[threadIdx.x] = rm;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fminf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) p.rowmin[t] = red[0];
}

// The per-target records come from a pinned host block.  Read there by the matrix kernel itself -- one 32-byte
// scalar load per workgroup -- they cost 0.4 ms per matrix (5000 dependent reads across PCIe, a few at a time), as
// much as a copy engine's hop; so a few wide, coalesced reads bring them to HBM first (160 KB: microseconds).
__global__ void __launch_bounds__(256) stage_args_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

hipError_t launch_matrix(const MatrixParams &p, const Layout &lay, int S, int waves, hipStream_t stream) {
  if (p.host_args) {
    // (the carrier flags, when there are any, lie behind the records in both blocks)
    const int n16 = (int)(((size_t)p.nloc * sizeof(MatrixArg) + (p.member ? (size_t)p.N : 0) + 15) / 16);
    hipLaunchKernelGGL(stage_args_kernel, dim3((n16 + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<const uint4 *>(p.host_args), reinterpret_cast<uint4 *>(const_cast<MatrixArg *>(p.args)), n16);
  }
  hipLaunchKernelGGL(matrix_kernel, dim3(p.nloc), dim3(256), (size_t)p.N * sizeof(float), stream, p, lay, S, waves);
  return hipGetLastError();
}

}  // namespace rl
