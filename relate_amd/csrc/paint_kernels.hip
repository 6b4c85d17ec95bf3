// paint_kernels.hip -- K1: stepping-stone painting of all targets.
//
// Replaces FastPainting::PaintSteppingStones (fast_painting.cpp:18-618).
// Two launches of N wavefronts each (forward, backward: independent in this
// stage); block b paints target order[b] (longest target first).
#include "paint_device.h"
#include "exact_sum.h"
#include "launch.h"

#ifndef RL_MODE
#error "compile with -DRL_MODE=0 (lanes), 1 (exact, parallel), 2 (exact, literal serial)"
#endif

namespace rl {

// Write the lane's registers as one stepping stone in donor order.  Stones
// are rare (W per target against D_k steps); to keep S per-register store
// addresses out of the hot loop's register budget the registers are staged,
// 16 at a time, through a 4 KiB LDS strip private to the wave and written by a
// rolled loop (each lane reads back only what it wrote: no barrier needed).
template <int S>
RL_DEV void emit_stone(const LaneCtx<S> &lc, const double (&v)[S], float *__restrict__ out,
                       float self_value, float *stage) {
  static_assert(S % 8 == 0, "S must be a multiple of 8");
  constexpr int R = S % 16 == 0 ? 16 : 8;
#pragma unroll
  for (int c = 0; c < S / R; c++) {
#pragma unroll
    for (int ii = 0; ii < R; ii++) {
      // pin the conversion to its chunk: hoisted, all S floats would be live at once
      double x = v[c * R + ii];
      asm volatile("" : "+v"(x) : : "memory");
      stage[ii * 64 + lc.lane] = (float)x;
    }
#pragma clang loop unroll(disable)
    for (int ii = 0; ii < R; ii++) {
      const int i = c * R + ii;
      if (i < lc.len) out[lc.donor(i)] = stage[ii * 64 + lc.lane];
    }
  }
  if (lc.lane == 0) out[lc.k] = self_value;
}

template <int S, int TAIL, int MODE>
RL_DEV void paint_forward(const PaintParams &p, int k, float *stage) {
  LaneCtx<S> lc;
  lc.init(p.lay, k);
  const PaintConsts &c = p.c;
  const int N = p.lay.N, W = p.W;
  const int64_t off = p.plan_off[k];
  const int D = (int)(p.plan_off[k + 1] - off);
  const int32_t *__restrict__ st = p.sites + off;
  const double *__restrict__ cfp = p.cf + off;
  const double *__restrict__ nx = p.nxt + off;
  const int32_t *__restrict__ ia = p.stone_ia + (size_t)k * W;

  double a[S];
  RawBits<S> raw;
  LaneBits<S> mb;

  // ---- SNP 0 (fast_painting.cpp:207-253)
  int sv = st[0];
  raw.load(p.bits + (size_t)(sv & 0x7fffffff) * p.row_words, lc.w0);
  mb.from_raw(raw, lc);
  mb.to_mismatch(sv < 0);
#pragma unroll
  for (int i = 0; i < S; i++) {
    double v = mb.get(i) ? c.init1 : c.init0;
    a[i] = (i < lc.len) ? v : 0.0;
  }
  double ssum = wave_sum<MODE, S>(RegTerm<S>{a, 0.0, 0.0, p.stats}, local_sum<S>(RegTerm<S>{a}));
  double ls = 0.0;
  int wa = 0;
  int next_stone = ia[0];
  while (next_stone == 0) {
    emit_stone<S>(lc, a, p.alpha + ((size_t)wa * N + k) * N, 0.0f, stage);
    if (lc.lane == 0) p.ls_alpha[(size_t)wa * N + k] = (float)ls;
    wa++;
    next_stone = wa < W ? ia[wa] : -1;
  }
  double cfac = cfp[0] * ssum;  // :260

  // prefetch row of step 1
  if (D > 1) raw.load(p.bits + (size_t)(st[1] & 0x7fffffff) * p.row_words, lc.w0);
  int sv_next = D > 1 ? st[1] : 0;

  for (int i = 1; i < D; i++) {
    sv = sv_next;
    mb.from_raw(raw, lc);
    mb.to_mismatch(sv < 0);
    if (i + 1 < D) {
      sv_next = st[i + 1];
      raw.load(p.bits + (size_t)(sv_next & 0x7fffffff) * p.row_words, lc.w0);
    }
    ls += nx[i - 1];  // :281-282
    double lsum = 0.0;
#pragma unroll
    for (int j = 0; j < S; j++) {  // :288-295
      double v = a[j] + cfac;
      masked_mul(v, __ballot(mb.get(j)), c.K1);  // v *= (mismatch ? K1 : 1.0)
      if (j >= S - TAIL) v = (j < lc.len) ? v : 0.0;
      a[j] = v;
      lsum += v;  // the lane's share of the serial sum (:300-303)
    }
    ssum = wave_sum<MODE, S>(RegTerm<S>{a, 0.0, 0.0, p.stats}, lsum);
    cfac = ssum;
    if (cfac < c.lower || cfac > c.upper) {  // :334-347
#pragma unroll
      for (int j = 0; j < S; j++) a[j] /= ssum;
      ls += log(ssum);
      cfac = 1.0;
    }
    cfac *= cfp[i];  // :349-352
    while (next_stone == i) {  // :354-374
      emit_stone<S>(lc, a, p.alpha + ((size_t)wa * N + k) * N, 0.0f, stage);
      if (lc.lane == 0) p.ls_alpha[(size_t)wa * N + k] = (float)ls;
      wa++;
      next_stone = wa < W ? ia[wa] : -1;
    }
  }
}

template <int S, int TAIL, int MODE>
RL_DEV void paint_backward(const PaintParams &p, int k, float *stage) {
  LaneCtx<S> lc;
  lc.init(p.lay, k);
  const PaintConsts &c = p.c;
  const int N = p.lay.N, W = p.W;
  const int64_t off = p.plan_off[k];
  const int D = (int)(p.plan_off[k + 1] - off);
  const int32_t *__restrict__ st = p.sites + off;
  const double *__restrict__ cfp = p.cf + off;
  const double *__restrict__ nx = p.nxt + off;
  const int32_t *__restrict__ ie = p.stone_ie + (size_t)k * W;

  double b[S];
  RawBits<S> raw;
  LaneBits<S> m_next, m_here;

  // ---- last SNP (:396-448)
  double ls = c.log_Nm1 - D * c.log_ntheta;  // normalizing_constant :399
#pragma unroll
  for (int i = 0; i < S; i++) b[i] = (i < lc.len) ? 1.0 : 0.0;
  double bsum = p.binit[k];  // serial sum of theta/ntheta minus ntheta (:421-431)
  int we = W - 1;
  int next_stone = ie[we];
  while (next_stone == D - 1) {
    emit_stone<S>(lc, b, p.beta + ((size_t)we * N + k) * N, 1.0f, stage);  // beta[k] = 1 here
    if (lc.lane == 0) p.ls_beta[(size_t)we * N + k] = (float)ls;
    we--;
    next_stone = we >= 0 ? ie[we] : -2;
  }
  double cfac = cfp[D - 1] * bsum;  // :454-455

  int sv = st[D - 1];
  raw.load(p.bits + (size_t)(sv & 0x7fffffff) * p.row_words, lc.w0);
  m_here.from_raw(raw, lc);
  m_here.to_mismatch(sv < 0);
  int sv_prev = D > 1 ? st[D - 2] : 0;
  if (D > 1) raw.load(p.bits + (size_t)(sv_prev & 0x7fffffff) * p.row_words, lc.w0);

  for (int j = D - 2; j >= 0; j--) {
    m_next = m_here;  // the later site's mismatches drive the update (:481-488)
    sv = sv_prev;
    m_here.from_raw(raw, lc);
    m_here.to_mismatch(sv < 0);
    if (j > 0) {
      sv_prev = st[j - 1];
      raw.load(p.bits + (size_t)(sv_prev & 0x7fffffff) * p.row_words, lc.w0);
    }
    ls += nx[j + 1];                       // :471-472
    const double b1 = cfac / c.ntheta;     // :474
    const double bt = cfac / c.theta - b1; // :475
    double lsum = 0.0;
#pragma unroll
    for (int i = 0; i < S; i++) {
      const unsigned long long mn = __ballot(m_next.get(i));
      double v = b[i];
      masked_add(v, mn, bt);      // b + mis*bt  (b + 0.0 == b)
      v = v + b1;
      masked_mul(v, mn, c.K1);    // *(mis ? K1 : 1.0)
      if (i >= S - TAIL) v = (i < lc.len) ? v : 0.0;
      b[i] = v;
      lsum += (m_here.get(i) ? c.theta : c.ntheta) * v;  // the lane's share of :495-503
    }
    const WeightedTerm<S> term{m_here, b, c.theta, c.ntheta, p.stats ? p.stats + 8 : nullptr};
    bsum = wave_sum<MODE, S>(term, lsum);  // :495-503
    cfac = bsum;
    if (cfac < c.lower || cfac > c.upper) {  // :538-551
#pragma unroll
      for (int i = 0; i < S; i++) b[i] /= bsum;
      ls += fast_log_dev((float)bsum);
      cfac = 1.0;
    }
    cfac *= cfp[j];  // :553-556
    while (next_stone == j) {  // :559-578
      emit_stone<S>(lc, b, p.beta + ((size_t)we * N + k) * N, 0.0f, stage);
      if (lc.lane == 0) p.ls_beta[(size_t)we * N + k] = (float)ls;
      we--;
      next_stone = we >= 0 ? ie[we] : -2;
    }
  }
}

// S <= 80: hold the kernel to 256 registers so that two waves share a SIMD
template <int S, int TAIL, int MODE, bool BACKWARD>
__global__ void __launch_bounds__(64, (S <= 80 ? 2 : 1)) paint_kernel(const PaintParams p) {
  __shared__ float stage[16 * 64];
  const int k = p.order[blockIdx.x];
  if (BACKWARD)
    paint_backward<S, TAIL, MODE>(p, k, stage);
  else
    paint_forward<S, TAIL, MODE>(p, k, stage);
}

template <int S, int TAIL>
static hipError_t launch_paint_t(const PaintParams &p, int backward, hipStream_t stream) {
  const dim3 grid(p.lay.N), block(64);
  if (backward)
    hipLaunchKernelGGL((paint_kernel<S, TAIL, RL_MODE, true>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((paint_kernel<S, TAIL, RL_MODE, false>), grid, block, 0, stream, p);
  return hipGetLastError();
}

template <>
hipError_t launch_paint_mode<RL_MODE>(const PaintParams &p, int S, int backward, hipStream_t stream) {
  switch (S) {
#define RL_CASE(s, t) \
  case s:             \
    return launch_paint_t<s, t>(p, backward, stream);
    RL_FOR_EACH_S(RL_CASE)
#undef RL_CASE
  }
  return hipErrorInvalidValue;
}

}  // namespace rl
