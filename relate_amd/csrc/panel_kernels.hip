// panel_kernels.hip -- bit-packed panel -> lane-mask panel of the stepping-stone
// kernel (paint_device.h "lane-mask panel"): masks[s][j] bit l = the donor lane
// l holds in register j (layout over all N donors) is ancestral at site s, 0
// past the end of a lane's run; row L is all zero; row L+1 holds the validity
// masks (bit l: register j lies inside lane l's run).  With `waves` = 2 (K1 at
// N > 5120) a site has two consecutive rows, one per wave of the painting
// workgroup (virtual lanes 0..63 / 64..127).  One workgroup per site, once per chunk.
#include "paint_device.h"
#include "launch.h"

namespace rl {

__global__ void __launch_bounds__(128) lane_mask_kernel(const uint32_t *__restrict__ bits, int row_words, int L, Layout lay,
                                                        int S, int waves, unsigned long long *__restrict__ masks) {
  const int s = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;  // one wave per wave of the painting workgroup
  const int vl = threadIdx.x;                                   // virtual lane
  const uint32_t *__restrict__ row = bits + (size_t)s * row_words;
  const int start = vl * lay.q + (vl < lay.rem ? vl : lay.rem);
  const int len = lay.q + (vl < lay.rem ? 1 : 0);
  unsigned long long *__restrict__ out = masks + ((size_t)s * waves + wave) * S;
  for (int j0 = 0; j0 < S; j0 += 64) {
    unsigned long long mine = 0;
    const int jn = S - j0 < 64 ? S - j0 : 64;
    for (int jj = 0; jj < jn; jj++) {
      const int j = j0 + jj, n = start + j;
      const bool bit = s < L ? (j < len ? !((row[n >> 5] >> (n & 31)) & 1u) : false) : (s == L + 1 && j < len);
      const unsigned long long m = __ballot(bit);
      if (lane == jj) mine = m;
    }
    if (lane < jn) out[j0 + lane] = mine;
  }
}

hipError_t launch_lane_masks(const uint32_t *bits, int row_words, int L, const Layout &lay, int S, int waves,
                             unsigned long long *masks, hipStream_t stream) {
  hipLaunchKernelGGL(lane_mask_kernel, dim3(L + 2), dim3(64 * waves), 0, stream, bits, row_words, L, lay, S, waves,
                     masks);
  return hipGetLastError();
}

// The paint file's quantisation of a stepping stone (CollapsedMatrix::DumpToFile / ReadFromFile,
// collapsed_matrix.hpp:228-296) without the file: a value joins the current run -- and is replaced by the run's FIRST
// value -- while |first - v| < 1e-3 * min(first, v) (float difference, double product), otherwise it starts the next
// run.  The rule is sequential along a stone, so a lane walks one stone; the 64 stones of a wave move through LDS in
// tiles of 64 x 64 so that HBM sees whole 256-byte segments.
__global__ void __launch_bounds__(64) quantise_kernel(float *__restrict__ stones, int rows, int N) {
  __shared__ float tile[64][65];
  const int lane = threadIdx.x, r0 = blockIdx.x * 64;
  const int nr = rows - r0 < 64 ? rows - r0 : 64;
  float current = 0.0f;
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int nc = N - c0 < 64 ? N - c0 : 64;
    for (int r = 0; r < nr; r++)
      if (lane < nc) tile[r][lane] = stones[(size_t)(r0 + r) * N + c0 + lane];
    __syncthreads();
    if (lane < nr) {
      for (int c = 0; c < nc; c++) {
        const float v = tile[lane][c];
        if (c0 + c == 0) {
          current = v;
        } else {
          const float diff = fabsf(current - v);
          const float mn = v < current ? v : current;
          if ((double)diff < 1e-3 * (double)mn)
            tile[lane][c] = current;
          else
            current = v;
        }
      }
    }
    __syncthreads();
    for (int r = 0; r < nr; r++)
      if (lane < nc) stones[(size_t)(r0 + r) * N + c0 + lane] = tile[r][lane];
    __syncthreads();
  }
}

hipError_t launch_quantise(float *stones, int rows, int N, hipStream_t stream) {
  hipLaunchKernelGGL(quantise_kernel, dim3((rows + 63) / 64), dim3(64), 0, stream, stones, rows, N);
  return hipGetLastError();
}

}  // namespace rl
