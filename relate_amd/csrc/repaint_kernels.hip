// repaint_kernels.hip -- K2: RePaintSection for all targets of one window.
//
// Replaces FastPainting::RePaintSection (fast_painting.cpp:621-1092) as it is
// driven by DistanceMeasure::GetTopologyWithRepaint (anc_builder.cpp:49-106).
// One wavefront per target; persistent blocks pull targets from an atomic
// counter (longest first).  Same register layout and lane-mask panel as the
// stepping-stone kernel (paint_device.h).  The forward pass keeps every alpha
// row (double) in a per-block HBM scratch strip laid out [row][register][lane]
// so that every store/load instruction moves 512 contiguous bytes; the backward
// pass reads it back, two chunks of registers ahead of their use, and writes the
// posterior rows `topology = float(alpha*beta)` in the same register-major
// layout (4 B per donor per visited site: the kernel is HBM-bound).
#include "paint_device.h"
#include "exact_sum.h"
#include "launch.h"

#ifndef RL_MODE
#error "compile with -DRL_MODE=0|1|2"
#endif

namespace rl {

typedef const __attribute__((address_space(4))) RepaintParams *ColdRepaint;

// Load N floats in donor order into the lane's registers (as doubles),
// 16 registers at a time through the wave-private LDS strip.
template <int S>
RL_DEV void load_stone(const PaintLane<S> &pl, const float *__restrict__ in, double (&v)[S], float *stage) {
  constexpr int R = S % 16 == 0 ? 16 : 8;
#pragma unroll
  for (int c = 0; c < S / R; c++) {
#pragma clang loop unroll(disable)
    for (int ii = 0; ii < R; ii++) {
      const int i = c * R + ii;
      stage[ii * 64 + pl.lane] = (i < pl.len) ? in[pl.start + i] : 0.0f;
    }
#pragma unroll
    for (int ii = 0; ii < R; ii++) v[c * R + ii] = (double)stage[ii * 64 + pl.lane];
  }
}

template <int S, int TAIL, int MODE, int WAVES>
RL_DEV void repaint_target(const RepaintParams &p, int n, double *__restrict__ scratch, float *stage,
                           WaveLink<WAVES> &lk) {
  const int wv = lk.w;  // this wave of the target's workgroup (wave-uniform)
  PaintLane<S> pl;
  pl.init(p.lay, n, wv);
  const PaintConsts &c = p.c;
  const int t = n - p.k0;  // index into the per-target arrays of this context
  const int ib = p.ib[t], ie = p.ie[t];
  const int D = ie - ib + 1;
  const int64_t off = p.plan_off[n] + ib;
  const int32_t *__restrict__ st = p.sites + off;
  const double *__restrict__ cfp = p.cf + off;
  const double *__restrict__ nx = p.nxt + off;
  const double cf_last = p.cf_last[t], nxt_last = p.nxt_last[t];
  constexpr int CK = REPAINT_CHECKPOINT;
  constexpr int WROW = S * 64;          // doubles per wave and scratch row
  constexpr int ROW = WROW * WAVES;     // a scratch row: [wave][register][lane]
  constexpr int TROW = S * 64 * WAVES;  // a posterior row, same order
  double *__restrict__ side = scratch + p.side_offset;  // [step][cfac, rescaling divisor or 0, logscale]
  scratch += (size_t)wv * WROW;
  const bool scribe = pl.lane == 0 && wv == 0;  // writes the side records
  constexpr int CH = S % 16 == 0 ? 16 : 8;  // registers per chunk of masks (forward)
  typedef typename MaskChunk<CH>::type Chunk;
  const double K1 = in_vgpr(c.K1);

  double a[S];

  // ---------------- forward (fast_painting.cpp:769-885)
  {
    const ColdRepaint cp = cold_params<RepaintParams>();
    load_stone<S>(pl, cp->alpha_begin + (size_t)t * cp->lay.N, a, stage);
  }
  set_slot<S>(a, pl.jk, pl.kbit, 0.0);  // alpha[n] = 0 for the target itself (:781)
  double ssum = wave_sum<MODE, S, WAVES>(RegTerm<S>{a}, local_sum<S>(RegTerm<S>{a}), lk);
  float lsf = p.ls_alpha[t];
  double prev_ls = (double)lsf;
  {
    double *row = scratch;  // row 0 is a checkpoint
#pragma unroll
    for (int i = 0; i < S; i++) row[i * 64 + pl.lane] = a[i];
    if (scribe) {
      side[0] = 0.0;
      side[1] = 0.0;
      side[2] = (double)lsf;
    }
  }
  double cfac = (D == 1 ? cf_last : cfp[0]) * ssum;
  {
    int s1 = D > 1 ? st[1] : 0, s2 = D > 2 ? st[2] : 0;  // row pipeline as in paint_forward
    uint32_t touched = 0;
    MaskRow row = site_row(p.masks, S, p.L, s1, WAVES, wv);
    Chunk first = load_masks<CH>(row, 0);
    for (int i = 1; i < D; i++) {
      retire_touch(touched);
      if (i + 1 < D) touched = touch_row(p.masks, S, s2, pl.lane, WAVES, wv);
      s1 = s2;
      if (i + 2 < D) s2 = st[i + 2];
      const double nx_i = nx[i - 1], cf_i = (i == D - 1 ? cf_last : cfp[i]);
      const double cfac_used = cfac;
      double divisor = 0.0;
      set_slot<S>(a, pl.jk, pl.kbit, -cfac);  // the target's own slot: (-c) + c = +0.0
      double lsum = 0.0;
      for_each_chunk_from<S, CH>(row, first, [&](int j0, const Chunk &m) {
        double v[CH];
#pragma unroll
        for (int jj = 0; jj < CH; jj++) {
          v[jj] = a[j0 + jj];
          if (j0 + jj < S - TAIL)
            v[jj] = v[jj] + cfac;
          else
            tail_add(v[jj], pl.len, j0 + jj, cfac);
        }
        masked_mul8<0>(v, m, K1);  // v *= (mismatch ? K1 : 1.0)
        if constexpr (CH == 16) masked_mul8<8>(v + 8, m, K1);
#pragma unroll
        for (int jj = 0; jj < CH; jj++) {
          a[j0 + jj] = v[jj];
          lsum += v[jj];
        }
      });
      row = site_row(p.masks, S, p.L, s1, WAVES, wv);
      first = load_masks<CH>(row, 0);
      ssum = wave_sum<MODE, S, WAVES>(RegTerm<S>{a}, lsum, lk);
      prev_ls += nx_i;
      lsf = (float)prev_ls;  // :806-807
      cfac = ssum;
      if (cfac < c.lower || cfac > c.upper) {  // :865-877
#pragma unroll
        for (int j = 0; j < S; j++) a[j] /= ssum;
        divisor = ssum;
        const double lg = log(ssum);
        prev_ls += lg;
        lsf = (float)((double)lsf + lg);
        cfac = 1.0;
      }
      cfac *= cf_i;
      if (i % CK == 0) {  // checkpoint row
        double *srow = scratch + (int64_t)(i / CK) * ROW;
#pragma unroll
        for (int j = 0; j < S; j++) srow[j * 64 + pl.lane] = a[j];
      }
      if (scribe) {  // what the backward pass needs to redo this step from the previous row
        side[(size_t)i * REPAINT_SIDE + 0] = cfac_used;
        side[(size_t)i * REPAINT_SIDE + 1] = divisor;
        side[(size_t)i * REPAINT_SIDE + 2] = (double)lsf;
      }
    }
    retire_touch(touched);
  }

  // ---------------- backward (:887-1073)
  const int row_lo = p.row_lo[t], row_hi = p.row_hi[t];  // posterior rows that are kept
  float *__restrict__ top = p.topology + p.slab_off[t] * (int64_t)TROW + (size_t)wv * (S * 64);
  float *__restrict__ lsout = p.logscales + p.top_off[t];
  const double theta = in_vgpr(c.theta), ntheta = in_vgpr(c.ntheta);
  double b[S];
  lsf = lsf + p.ls_beta[t];  // float += float (:895)
  {
    const ColdRepaint cp = cold_params<RepaintParams>();
    load_stone<S>(pl, cp->beta_end + (size_t)t * cp->lay.N, b, stage);
  }
  if constexpr (WAVES > 1) __syncthreads();  // the side records of wave 0 are visible to wave 1
  __threadfence_block();
  set_slot<S>(b, pl.jk, pl.kbit, 0.0);  // the target's own slot of beta is +0.0 (fast_painting.cpp: b[k] = 0)
  // topology row j = float(alpha_j * beta_j).  alpha_j is rebuilt chunk by chunk from the nearest checkpoint row
  // at or before j: up to CK-1 forward steps (the same operations on the same operands: same bits).  Its slots of
  // the target itself and past a lane's run come out as finite garbage there; beta is +0.0 in both.
  auto product_row = [&](int j) {
    if (j < row_lo || j >= row_hi) return;  // (uniform over the workgroup)
    const int cp = j - j % CK, r = j - cp;
    const double *__restrict__ arow = scratch + (int64_t)(cp / CK) * ROW;
    float *__restrict__ trow = top + (int64_t)(j - row_lo) * TROW;
    MaskRow rows[CK];
    double cfs[CK], dvs[CK];
#pragma unroll
    for (int q = 1; q < CK; q++) {
      const int i = cp + q <= j ? cp + q : j;  // (unused beyond r)
      rows[q] = site_row(p.masks, S, p.L, st[i], WAVES, wv);
      cfs[q] = side[(size_t)i * REPAINT_SIDE + 0];
      dvs[q] = side[(size_t)i * REPAINT_SIDE + 1];
    }
#pragma unroll
    for (int c0 = 0; c0 < S / 8; c0++) {
      double a8[8];
#pragma unroll
      for (int jj = 0; jj < 8; jj++) a8[jj] = arow[(c0 * 8 + jj) * 64 + pl.lane];
#pragma unroll
      for (int q = 1; q < CK; q++) {
        if (q <= r) {  // wave-uniform
          const u64x8 m = load_masks<8>(rows[q], c0);
#pragma unroll
          for (int jj = 0; jj < 8; jj++) a8[jj] = a8[jj] + cfs[q];
          masked_mul8<0>(a8, m, K1);
          if (dvs[q] != 0.0) {
#pragma unroll
            for (int jj = 0; jj < 8; jj++) a8[jj] /= dvs[q];
          }
        }
      }
#pragma unroll
      for (int jj = 0; jj < 8; jj++) trow[(c0 * 8 + jj) * 64 + pl.lane] = (float)(a8[jj] * b[c0 * 8 + jj]);
    }
  };
  product_row(D - 1);  // :930
  if (pl.lane == 0 && wv == 0) lsout[D - 1] = lsf;
  int s0 = st[D - 1], s1 = D > 1 ? st[D - 2] : 0, s2 = D > 2 ? st[D - 3] : 0;
  MaskRow rown = site_row(p.masks, S, p.L, s0, WAVES, wv);
  double bsum;
  {
    const MaskTerm<S> term{rown, b, theta, ntheta};
    bsum = wave_sum<MODE, S, WAVES>(term, local_sum<S>(term), lk);
  }
  cfac = cf_last * bsum;
  prev_ls = (double)p.ls_beta[t];  // :951
  MaskRow rowh = site_row(p.masks, S, p.L, s1, WAVES, wv);
  u64x4 firstn = load_masks<4>(rown, 0), firsth = load_masks<4>(rowh, 0);
  uint32_t touched = 0;
  for (int j = D - 2; j >= 0; j--) {
    retire_touch(touched);
    if (j > 0) touched = touch_row(p.masks, S, s2, pl.lane, WAVES, wv);
    s0 = s1;
    s1 = s2;
    if (j > 1) s2 = st[j - 2];
    const double nx_j = (j + 1 == D - 1 ? nxt_last : nx[j + 1]), cf_j = cfp[j];
    const double als = side[(size_t)j * REPAINT_SIDE + 2];  // the forward logscale of this site
    const double b1 = cfac / ntheta;
    const double bt = cfac / theta - b1;
    set_slot<S>(b, pl.jk, pl.kbit, -b1);
    double lsum = 0.0;
    MaskRow vrow = (MaskRow)(p.masks + ((size_t)(p.L + 1) * WAVES + wv) * S);
    asm volatile("" : "+s"(vrow));
    for_each_chunk2_tail<S, 4, TAIL>(rown, rowh, vrow, firstn, firsth,
                                     [&](int j0, const u64x4 &mn, const u64x4 &mh, const u64x4 &va) {
      double v[4], x[4];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) v[jj] = b[j0 + jj];
      if (j0 + 4 <= S - TAIL)
        backward4(v, x, mn, mh, bt, b1, K1, theta, ntheta);
      else
        backward4_tail(v, x, mn, mh, va, bt, b1, K1, theta, ntheta);
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        b[j0 + jj] = v[jj];
        lsum += x[jj];
      }
    });
    const MaskTerm<S> term{rowh, b, theta, ntheta};
    rown = rowh;
    rowh = site_row(p.masks, S, p.L, s1, WAVES, wv);
    if (MODE == 0) {
      firstn = load_masks<4>(rown, 0);
      firsth = load_masks<4>(rowh, 0);
    }
    bsum = wave_sum<MODE, S, WAVES>(term, lsum, lk);
    if (MODE != 0) {
      firstn = load_masks<4>(rown, 0);
      firsth = load_masks<4>(rowh, 0);
    }
    prev_ls += nx_j;
    lsf = (float)(als + prev_ls);  // :962-963
    cfac = bsum;
    product_row(j);  // topology = float(alpha * beta) before the rescale of this step (:1039 vs :1047)
    if (cfac < c.lower || cfac > c.upper) {  // :1047-1061
#pragma unroll
      for (int i = 0; i < S; i++) b[i] /= bsum;
      const double lg = log(bsum);
      prev_ls += lg;
      lsf = (float)((double)lsf + lg);
      cfac = 1.0;
    }
    cfac *= cf_j;
    if (pl.lane == 0 && wv == 0) lsout[j] = lsf;
  }
  retire_touch(touched);
}

template <int S, int TAIL, int MODE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, (S <= 80 ? 2 : 1)) repaint_kernel(const RepaintParams p, int *counter) {
  __shared__ float stage[WAVES][16 * 64];
  __shared__ WaveLinkStorage link;
  __shared__ int s_t;
  WaveLink<WAVES> lk;
  lk.s = &link;
  lk.w = WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  double *scratch = p.scratch + (int64_t)blockIdx.x * p.scratch_stride;
  for (;;) {
    if (threadIdx.x == 0) s_t = atomicAdd(counter, 1);
    __syncthreads();
    const int t = s_t;
    __syncthreads();
    if (t >= p.nloc) break;
    repaint_target<S, TAIL, MODE, WAVES>(p, p.order[t], scratch, stage[lk.w], lk);
  }
}

template <>
hipError_t launch_repaint_mode<RL_MODE>(const RepaintParams &p, int S, int waves, int nblocks, int *counter,
                                        hipStream_t stream) {
  if (waves == 1) {
    switch (S) {
#define RL_CASE(s, t)                                                                                           \
  case s:                                                                                                       \
    hipLaunchKernelGGL((repaint_kernel<s, t, RL_MODE, 1>), dim3(nblocks), dim3(64), 0, stream, p, counter); \
    return hipGetLastError();
      RL_FOR_EACH_S(RL_CASE)
#undef RL_CASE
    }
  } else if (waves == 2) {
#ifndef RL_ONLY_S
    switch (S) {
#define RL_CASE(s, t)                                                                                            \
  case s:                                                                                                        \
    hipLaunchKernelGGL((repaint_kernel<s, t, RL_MODE, 2>), dim3(nblocks), dim3(128), 0, stream, p, counter); \
    return hipGetLastError();
      RL_FOR_EACH_S_2WAVES(RL_CASE)
#undef RL_CASE
    }
#endif
  }
  return hipErrorInvalidValue;
}

}  // namespace rl
