// repaint_kernels.hip -- K2: RePaintSection for all targets of one window.
//
// Replaces FastPainting::RePaintSection (fast_painting.cpp:621-1092) as it is
// driven by DistanceMeasure::GetTopologyWithRepaint (anc_builder.cpp:49-106).
// One wavefront per target; persistent blocks pull targets from an atomic
// counter (longest first).  The forward pass keeps every alpha row (double) in
// a per-block HBM scratch strip laid out [row][register][lane] so that every
// store/load instruction moves 512 contiguous bytes; the backward pass reads
// it back and writes the posterior rows `topology = float(alpha*beta)` in the
// same register-major layout (4 B per donor per visited site: the kernel is
// HBM-write-bound).
#include "paint_device.h"
#include "exact_sum.h"
#include "launch.h"

#ifndef RL_MODE
#error "compile with -DRL_MODE=0|1|2"
#endif

namespace rl {

// Load N floats in donor order into the lane's registers (as doubles),
// 16 registers at a time through the wave-private LDS strip.
template <int S>
RL_DEV void load_stone(const LaneCtx<S> &lc, const float *__restrict__ in, double (&v)[S], float *stage) {
  constexpr int R = S % 16 == 0 ? 16 : 8;
#pragma unroll
  for (int c = 0; c < S / R; c++) {
#pragma clang loop unroll(disable)
    for (int ii = 0; ii < R; ii++) {
      const int i = c * R + ii;
      stage[ii * 64 + lc.lane] = (i < lc.len) ? in[lc.donor(i)] : 0.0f;
    }
#pragma unroll
    for (int ii = 0; ii < R; ii++) v[c * R + ii] = (double)stage[ii * 64 + lc.lane];
  }
}

template <int S, int TAIL, int MODE>
RL_DEV void repaint_target(const RepaintParams &p, int n, double *__restrict__ scratch, float *stage) {
  LaneCtx<S> lc;
  lc.init(p.lay, n);
  const PaintConsts &c = p.c;
  const int N = p.lay.N;
  const int t = n - p.k0;  // index into the per-target arrays of this context
  const int ib = p.ib[t], ie = p.ie[t];
  const int D = ie - ib + 1;
  const int64_t off = p.plan_off[n] + ib;
  const int32_t *__restrict__ st = p.sites + off;
  const double *__restrict__ cfp = p.cf + off;
  const double *__restrict__ nx = p.nxt + off;
  const double cf_last = p.cf_last[t], nxt_last = p.nxt_last[t];
  constexpr int ROW = (S + 1) * 64;  // doubles per scratch row (+64: per-lane logscale copy)
  const int64_t trow0 = p.top_off[t];
  float *__restrict__ top = p.topology + trow0 * (int64_t)(S * 64);
  float *__restrict__ lsout = p.logscales + trow0;

  double a[S];
  RawBits<S> raw;
  LaneBits<S> mb;

  // ---------------- forward (fast_painting.cpp:769-885)
  load_stone<S>(lc, p.alpha_begin + (size_t)t * N, a, stage);
  double ssum = wave_sum<MODE, S>(RegTerm<S>{a}, local_sum<S>(RegTerm<S>{a}));
  float lsf = p.ls_alpha[t];
  double prev_ls = (double)lsf;
  {
    double *row = scratch;
#pragma unroll
    for (int i = 0; i < S; i++) row[i * 64 + lc.lane] = a[i];
    row[S * 64 + lc.lane] = (double)lsf;
  }
  double cfac = (D == 1 ? cf_last : cfp[0]) * ssum;
  int sv_next = D > 1 ? st[1] : 0;
  if (D > 1) raw.load(p.bits + (size_t)(sv_next & 0x7fffffff) * p.row_words, lc.w0);
  for (int i = 1; i < D; i++) {
    const int sv = sv_next;
    mb.from_raw(raw, lc);
    mb.to_mismatch(sv < 0);
    if (i + 1 < D) {
      sv_next = st[i + 1];
      raw.load(p.bits + (size_t)(sv_next & 0x7fffffff) * p.row_words, lc.w0);
    }
    prev_ls += nx[i - 1];
    lsf = (float)prev_ls;  // :806-807
    double lsum = 0.0;
#pragma unroll
    for (int j = 0; j < S; j++) {
      double v = a[j] + cfac;
      masked_mul(v, __ballot(mb.get(j)), c.K1);  // v *= (mismatch ? K1 : 1.0)
      if (j >= S - TAIL) v = (j < lc.len) ? v : 0.0;
      a[j] = v;
      lsum += v;
    }
    ssum = wave_sum<MODE, S>(RegTerm<S>{a}, lsum);
    cfac = ssum;
    if (cfac < c.lower || cfac > c.upper) {  // :865-877
#pragma unroll
      for (int j = 0; j < S; j++) a[j] /= ssum;
      const double lg = log(ssum);
      prev_ls += lg;
      lsf = (float)((double)lsf + lg);
      cfac = 1.0;
    }
    cfac *= (i == D - 1 ? cf_last : cfp[i]);
    double *row = scratch + (int64_t)i * ROW;
#pragma unroll
    for (int j = 0; j < S; j++) row[j * 64 + lc.lane] = a[j];
    row[S * 64 + lc.lane] = (double)lsf;
  }

  // ---------------- backward (:887-1073)
  double b[S];
  LaneBits<S> m_next, m_here;
  lsf = lsf + p.ls_beta[t];  // float += float (:895)
  load_stone<S>(lc, p.beta_end + (size_t)t * N, b, stage);
  int sv = st[D - 1];
  raw.load(p.bits + (size_t)(sv & 0x7fffffff) * p.row_words, lc.w0);
  m_here.from_raw(raw, lc);
  m_here.to_mismatch(sv < 0);
  const WeightedTerm<S> term{m_here, b, c.theta, c.ntheta};
  double bsum = wave_sum<MODE, S>(term, local_sum<S>(term));
  {
    // `a` still holds row D-1 (:930)
    float *trow = top + (int64_t)(D - 1) * (S * 64);
#pragma unroll
    for (int i = 0; i < S; i++) trow[i * 64 + lc.lane] = (float)(a[i] * b[i]);
    if (lc.lane == 0) lsout[D - 1] = lsf;
  }
  cfac = cf_last * bsum;
  prev_ls = (double)p.ls_beta[t];  // :951
  int sv_prev = D > 1 ? st[D - 2] : 0;
  if (D > 1) raw.load(p.bits + (size_t)(sv_prev & 0x7fffffff) * p.row_words, lc.w0);
  for (int j = D - 2; j >= 0; j--) {
    m_next = m_here;
    sv = sv_prev;
    m_here.from_raw(raw, lc);
    m_here.to_mismatch(sv < 0);
    if (j > 0) {
      sv_prev = st[j - 1];
      raw.load(p.bits + (size_t)(sv_prev & 0x7fffffff) * p.row_words, lc.w0);
    }
    const double *row = scratch + (int64_t)j * ROW;
    prev_ls += (j + 1 == D - 1 ? nxt_last : nx[j + 1]);
    lsf = (float)(row[S * 64 + lc.lane] + prev_ls);  // :962-963
    const double b1 = cfac / c.ntheta;
    const double bt = cfac / c.theta - b1;
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
    bsum = wave_sum<MODE, S>(term, lsum);
    cfac = bsum;
    float *trow = top + (int64_t)j * (S * 64);
#pragma unroll
    for (int i = 0; i < S; i++) trow[i * 64 + lc.lane] = (float)(row[i * 64 + lc.lane] * b[i]);  // :1039
    if (cfac < c.lower || cfac > c.upper) {  // :1047-1061
#pragma unroll
      for (int i = 0; i < S; i++) b[i] /= bsum;
      const double lg = log(bsum);
      prev_ls += lg;
      lsf = (float)((double)lsf + lg);
      cfac = 1.0;
    }
    cfac *= cfp[j];
    if (lc.lane == 0) lsout[j] = lsf;
  }
}

template <int S, int TAIL, int MODE>
__global__ void __launch_bounds__(64) repaint_kernel(const RepaintParams p, int *counter) {
  __shared__ float stage[16 * 64];
  __shared__ int s_t;
  double *scratch = p.scratch + (int64_t)blockIdx.x * p.scratch_stride;
  for (;;) {
    if (threadIdx.x == 0) s_t = atomicAdd(counter, 1);
    __syncthreads();
    const int t = s_t;
    __syncthreads();
    if (t >= p.nloc) break;
    repaint_target<S, TAIL, MODE>(p, p.order[t], scratch, stage);
  }
}

template <>
hipError_t launch_repaint_mode<RL_MODE>(const RepaintParams &p, int S, int nblocks, int *counter, hipStream_t stream) {
  switch (S) {
#define RL_CASE(s, t)                                                                                        \
  case s:                                                                                                    \
    hipLaunchKernelGGL((repaint_kernel<s, t, RL_MODE>), dim3(nblocks), dim3(64), 0, stream, p, counter); \
    return hipGetLastError();
    RL_FOR_EACH_S(RL_CASE)
#undef RL_CASE
  }
  return hipErrorInvalidValue;
}

}  // namespace rl
