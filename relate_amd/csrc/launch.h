// launch.h -- host-visible launchers of the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace rl {

// Register-tile sizes the kernels are instantiated for: S doubles per lane,
// with the last TAIL registers carrying a per-lane validity test.
#ifdef RL_ONLY_S
#define RL_FOR_EACH_S(X) X(RL_ONLY_S, 16)
#else
#define RL_FOR_EACH_S(X) \
  X(8, 8) X(16, 8) X(32, 16) X(48, 16) X(64, 16) X(80, 16)
#endif

// smallest instantiated S with S >= q + (rem > 0); 0 if N is too large
inline int choose_S(const Layout &lay) {
  const int need = lay.q + (lay.rem > 0 ? 1 : 0);
  static const int sizes[] = {8, 16, 32, 48, 64, 80};
  for (int s : sizes)
    if (s >= need) return s;
  return 0;
}

// the layout of all kernels: all N donors, the target keeps a slot that is pinned to +0.0;
// cut into 64*waves balanced runs (target_waves)
inline Layout make_layout(int N, int waves = 1) {
  Layout l;
  l.N = N;
  l.P = N;
  l.q = N / (64 * waves);
  l.rem = N % (64 * waves);
  return l;
}
// K1 and K2 give a target to a workgroup of two waves once one wave would need more than 80
// registers per lane (two waves per SIMD need the kernels to stay within 256 VGPRs)
inline int target_waves(int N) { return N > 80 * 64 ? 2 : 1; }
#define RL_FOR_EACH_S_2WAVES(X) X(48, 16) X(64, 16) X(80, 16)
hipError_t launch_lane_masks(const uint32_t *bits, int row_words, int L, const Layout &lay, int S, int waves,
                             unsigned long long *masks, hipStream_t stream);

// kernel summation modes (template parameter MODE of the kernels)
//   0 = lanes (RL_SUM_LANES), 1 = exact, parallel (RL_SUM_EXACT), 2 = exact, literal serial (RL_SUM_EXACT_SERIAL),
//   3 = lanes on a packed-FP32 state (RL_SUM_LANES32; K1 only, paint32_kernels.hip -- K2 then runs its `lanes` kernels)
// dir: 0 forward pass, 1 backward pass, 2 both in one launch of 2 * nloc workgroups
template <int MODE>
hipError_t launch_paint_mode(const PaintParams &p, int S, int waves, int dir, hipStream_t stream);
template <int MODE>
hipError_t launch_repaint_mode(const RepaintParams &p, int S, int waves, hipStream_t stream);
template <> hipError_t launch_paint_mode<0>(const PaintParams &, int, int, int, hipStream_t);
template <> hipError_t launch_paint_mode<1>(const PaintParams &, int, int, int, hipStream_t);
template <> hipError_t launch_paint_mode<2>(const PaintParams &, int, int, int, hipStream_t);
template <> hipError_t launch_paint_mode<3>(const PaintParams &, int, int, int, hipStream_t);
template <> hipError_t launch_repaint_mode<0>(const RepaintParams &, int, int, hipStream_t);
template <> hipError_t launch_repaint_mode<1>(const RepaintParams &, int, int, hipStream_t);
template <> hipError_t launch_repaint_mode<2>(const RepaintParams &, int, int, hipStream_t);

inline int kernel_mode(int sum_mode) { return sum_mode == 0 ? 1 : (sum_mode == 1 || sum_mode == 3 ? 0 : 2); }
inline hipError_t launch_paint(const PaintParams &p, int S, int waves, int dir, hipStream_t stream) {
  if (p.sum_mode == 3) return launch_paint_mode<3>(p, S, waves, dir, stream);  // RL_SUM_LANES32
  switch (kernel_mode(p.sum_mode)) {
    case 0: return launch_paint_mode<0>(p, S, waves, dir, stream);
    case 1: return launch_paint_mode<1>(p, S, waves, dir, stream);
    default: return launch_paint_mode<2>(p, S, waves, dir, stream);
  }
}
// K2: the forward kernel (checkpoint rows + side records of every target), then the backward kernel, one workgroup
// per target each
inline hipError_t launch_repaint(const RepaintParams &p, int S, int waves, hipStream_t stream) {
  switch (kernel_mode(p.sum_mode)) {
    case 0: return launch_repaint_mode<0>(p, S, waves, stream);
    case 1: return launch_repaint_mode<1>(p, S, waves, stream);
    default: return launch_repaint_mode<2>(p, S, waves, stream);
  }
}
// the paint file's run-length quantisation of `rows` stones of N floats, in place (panel_kernels.hip)
hipError_t launch_quantise(float *stones, int rows, int N, hipStream_t stream);
hipError_t launch_matrix(const MatrixParams &p, const Layout &lay, int S, int waves, hipStream_t stream);

}  // namespace rl
