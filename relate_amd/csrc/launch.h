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
  X(8, 8) X(16, 8) X(32, 16) X(48, 16) X(64, 16) X(80, 16) X(96, 16) X(128, 32) X(160, 32)
#endif

// smallest instantiated S with S >= q + (rem > 0); 0 if N is too large
inline int choose_S(const Layout &lay) {
  const int need = lay.q + (lay.rem > 0 ? 1 : 0);
  static const int sizes[] = {8, 16, 32, 48, 64, 80, 96, 128, 160};
  for (int s : sizes)
    if (s >= need) return s;
  return 0;
}

inline Layout make_layout(int N) {
  Layout l;
  l.N = N;
  l.P = N - 1;
  l.q = l.P / 64;
  l.rem = l.P % 64;
  return l;
}

hipError_t launch_paint(const PaintParams &p, int S, int backward, hipStream_t stream);
hipError_t launch_repaint(const RepaintParams &p, int S, int nblocks, int *counter, hipStream_t stream);
hipError_t launch_matrix(const MatrixParams &p, const Layout &lay, int S, hipStream_t stream);

}  // namespace rl
