// paint_kernels.hip -- K1: stepping-stone painting of all targets.
//
// Replaces FastPainting::PaintSteppingStones (fast_painting.cpp:18-618).
// The two directions are independent in this stage: by default ONE launch of
// 2N workgroups paints both (block 2i the backward pass of target order[i],
// block 2i+1 its forward pass; longest target first), so that the last round
// of workgroups of one direction does not leave the chip half empty while the
// other direction waits.  DIR = 0 / 1 launch one direction alone (profiling).
#include <cstdlib>
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
// The slot of donor k itself (held at +0.0) is written as self_value.
template <int S>
RL_DEV void emit_stone(const PaintLane<S> &pl, const double (&v)[S], float *__restrict__ out,
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
      stage[ii * 64 + pl.lane] = (float)x;
    }
#pragma clang loop unroll(disable)
    for (int ii = 0; ii < R; ii++) {
      const int i = c * R + ii;
      const int n = pl.start + i;
      if (i < pl.len) out[n] = (n == pl.k) ? self_value : stage[ii * 64 + pl.lane];
    }
  }
}

// experiment builds (-DRL_STATS): per-segment cycle counters of the forward step, lanes mode
#ifdef RL_STATS
#define RL_TICK(n) const unsigned long long tick##n = __builtin_readcyclecounter()
#define RL_TOCK(acc, a, b) acc += tick##b - tick##a
#else
#define RL_TICK(n) do { } while (0)
#define RL_TOCK(acc, a, b) do { } while (0)
#endif

typedef const __attribute__((address_space(4))) PaintParams *ColdParams;

template <int S, int TAIL, int MODE, int WAVES>
RL_DEV void paint_forward(const PaintParams &p, int k, float *stage, WaveLink<WAVES> &lk) {
  const int wv = lk.w;  // this wave of the target's workgroup (wave-uniform)
  PaintLane<S> pl;
  pl.init(p.lay, k, wv);
  const PaintConsts &c = p.c;
  const int64_t off = p.plan_off[k];
  const int D = (int)(p.plan_off[k + 1] - off);
  // the plan arrays come in by vector loads (vmcnt), requested a step ahead
  const int32_t *__restrict__ st = p.sites + off;
  const double *__restrict__ cfp = p.cf + off;
  const double *__restrict__ nx = p.nxt + off;

  double a[S];

  // ---- SNP 0 (fast_painting.cpp:207-253)
  for_each_chunk<S, 8>(site_row(p.masks, S, p.L, st[0], WAVES, wv), [&](int j0, const u64x8 &m) {
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
      double v = c.init0;
      masked_mov(v, m[jj], c.init1);
      if (j0 + jj >= S - TAIL) masked_mov(v, ~pl.valid(j0 + jj), 0.0);
      a[j0 + jj] = v;
    }
  });
  set_slot<S>(a, pl.jk, pl.kbit, 0.0);
  double ssum = wave_sum<MODE, S, WAVES>(RegTerm<S>{a, 0.0, 0.0, p.stats}, local_sum<S>(RegTerm<S>{a}), lk);
  double ls = 0.0;
  int wa = 0;
  // stone wa of the target: written when the visited index reaches stone_ia[k][wa] (:354-374)
  auto stone_index = [&](int w) {
    const ColdParams cp = cold_params<PaintParams>();
    return w < cp->W ? cp->stone_ia[(size_t)k * cp->W + w] : -1;
  };
  auto write_stone = [&]() {
    const ColdParams cp = cold_params<PaintParams>();
    const size_t N = cp->lay.N, row = (size_t)wa * cp->nloc + (k - cp->k0);
    emit_stone<S>(pl, a, cp->alpha + row * N, 0.0f, stage);
    if (pl.lane == 0 && wv == 0) cp->ls_alpha[row] = (float)ls;
    wa++;
  };
  int next_stone = stone_index(0);
  while (next_stone == 0) {
    write_stone();
    next_stone = stone_index(wa);
  }
  double cfac = cfp[0] * ssum;  // :260

  // row pipeline: step i reads the row of s1 with scalar loads, its first
  // chunk requested before the previous step's sum; the row of s2 (step i+1) is
  // pulled into L2 by a vector load during step i
  int s1 = D > 1 ? st[1] : 0, s2 = D > 2 ? st[2] : 0;
  uint32_t touched = 0;
  const double K1 = in_vgpr(c.K1);
  constexpr int CH = S % 16 == 0 ? 16 : 8;  // registers per chunk of masks
  typedef typename MaskChunk<CH>::type Chunk;
  MaskRow row = site_row(p.masks, S, p.L, s1, WAVES, wv);
  Chunk first = load_masks<CH>(row, 0);
  unsigned long long seg1 = 0, seg2 = 0, seg3 = 0, seg4 = 0, seg5 = 0;
  (void)seg1; (void)seg2; (void)seg3; (void)seg4; (void)seg5;
  for (int i = 1; i < D; i++) {
    RL_TICK(0);
    retire_touch(touched);
    RL_TICK(1);
    if (i + 1 < D) touched = touch_row(p.masks, S, s2, pl.lane, WAVES, wv);
    s1 = s2;
    if (i + 2 < D) s2 = st[i + 2];
    // requested here, used after the sum: the chunk loop's waits cover the latency
    const double nx_i = nx[i - 1], cf_i = cfp[i];
    set_slot<S>(a, pl.jk, pl.kbit, -cfac);  // donor k: (-c) + c = +0.0
    double lsum = 0.0;
    RL_TICK(2);
    for_each_chunk_from<S, CH>(row, first, [&](int j0, const Chunk &m) {  // :288-295
      double v[CH];
#pragma unroll
      for (int jj = 0; jj < CH; jj++) {
        v[jj] = a[j0 + jj];
        if (j0 + jj < S - TAIL)
          v[jj] = v[jj] + cfac;
        else
          tail_add(v[jj], pl.len, j0 + jj, cfac);  // slots past the lane's run stay +0.0
      }
      masked_mul8<0>(v, m, K1);  // v *= (mismatch ? K1 : 1.0)
      if constexpr (CH == 16) masked_mul8<8>(v + 8, m, K1);
#pragma unroll
      for (int jj = 0; jj < CH; jj++) {
        a[j0 + jj] = v[jj];
        lsum += v[jj];  // the lane's share of the serial sum (:300-303)
      }
    });
    RL_TICK(3);
    row = site_row(p.masks, S, p.L, s1, WAVES, wv);
    first = load_masks<CH>(row, 0);
    ssum = wave_sum<MODE, S, WAVES>(RegTerm<S>{a, 0.0, 0.0, p.stats}, lsum, lk);
    RL_TICK(4);
    ls += nx_i;  // :281-282
    cfac = ssum;
    if (cfac < c.lower || cfac > c.upper) {  // :334-347
#pragma unroll
      for (int j = 0; j < S; j++) a[j] /= ssum;
      ls += log(ssum);
      cfac = 1.0;
    }
    cfac *= cf_i;  // :349-352
    RL_TICK(5);
    RL_TOCK(seg1, 0, 1); RL_TOCK(seg2, 1, 2); RL_TOCK(seg3, 2, 3); RL_TOCK(seg4, 3, 4); RL_TOCK(seg5, 4, 5);
    while (next_stone == i) {  // :354-374
      write_stone();
      next_stone = stone_index(wa);
    }
  }
  retire_touch(touched);
#ifdef RL_STATS
  if (MODE != 0 && p.stats && pl.lane == 0 && wv == 0) {  // whole step | chunk loop | sum | rescale test + factor
    atomicAdd(&p.stats[16], seg1 + seg2 + seg3 + seg4 + seg5);
    atomicAdd(&p.stats[17], seg3); atomicAdd(&p.stats[18], seg4); atomicAdd(&p.stats[19], seg1 + seg2 + seg5);
  }
  if (MODE == 0 && p.stats && pl.lane == 0 && wv == 0) {  // wait for prefetch | loads + slot | chunk loop | sum | rescale test
    atomicAdd(&p.stats[0], (unsigned long long)(D - 1));
    atomicAdd(&p.stats[1], seg1); atomicAdd(&p.stats[2], seg2); atomicAdd(&p.stats[3], seg3);
    atomicAdd(&p.stats[4], seg4); atomicAdd(&p.stats[5], seg5);
  }
#endif
}

template <int S, int TAIL, int MODE, int WAVES>
RL_DEV void paint_backward(const PaintParams &p, int k, float *stage, WaveLink<WAVES> &lk) {
  const int wv = lk.w;
  PaintLane<S> pl;
  pl.init(p.lay, k, wv);
  const PaintConsts &c = p.c;
  const int64_t off = p.plan_off[k];
  const int D = (int)(p.plan_off[k + 1] - off);
  const int32_t *__restrict__ st = p.sites + off;
  const double *__restrict__ cfp = p.cf + off;
  const double *__restrict__ nx = p.nxt + off;

  double b[S];

  // ---- last SNP (:396-448)
  double ls = c.log_Nm1 - D * c.log_ntheta;  // normalizing_constant :399
#pragma unroll
  for (int i = 0; i < S; i++) {
    double v = 1.0;
    if (i >= S - TAIL) masked_mov(v, ~pl.valid(i), 0.0);
    b[i] = v;
  }
  set_slot<S>(b, pl.jk, pl.kbit, 0.0);  // written as beta[k] = 1 below, +0.0 from then on
  double bsum = p.binit[k];  // serial sum of theta/ntheta minus ntheta (:421-431)
  int we = p.W - 1;
  auto stone_index = [&](int w) {
    const ColdParams cp = cold_params<PaintParams>();
    return w >= 0 ? cp->stone_ie[(size_t)k * cp->W + w] : -2;
  };
  auto write_stone = [&](float self_value) {
    const ColdParams cp = cold_params<PaintParams>();
    const size_t N = cp->lay.N, row = (size_t)we * cp->nloc + (k - cp->k0);
    emit_stone<S>(pl, b, cp->beta + row * N, self_value, stage);
    if (pl.lane == 0 && wv == 0) cp->ls_beta[row] = (float)ls;
    we--;
  };
  int next_stone = stone_index(we);
  while (next_stone == D - 1) {
    write_stone(1.0f);  // beta[k] = 1 at the last SNP
    next_stone = stone_index(we);
  }
  double cfac = cfp[D - 1] * bsum;  // :454-455

  // row pipeline as in paint_forward: step j reads the rows of s0 (site j+1)
  // and s1 (site j); the row of s2 (site j-1) goes to L2 during the step
  int s0 = st[D - 1], s1 = D > 1 ? st[D - 2] : 0, s2 = D > 2 ? st[D - 3] : 0;
  uint32_t touched = 0;
  MaskRow rown = site_row(p.masks, S, p.L, s0, WAVES, wv);  // the later site's mismatches drive the update (:481-488)
  MaskRow rowh = site_row(p.masks, S, p.L, s1, WAVES, wv);
  u64x4 firstn = load_masks<4>(rown, 0), firsth = load_masks<4>(rowh, 0);
  const double K1 = in_vgpr(c.K1), theta = in_vgpr(c.theta), ntheta = in_vgpr(c.ntheta);
  unsigned long long bseg1 = 0, bseg2 = 0, bseg3 = 0, bseg4 = 0;
  (void)bseg1; (void)bseg2; (void)bseg3; (void)bseg4;
  for (int j = D - 2; j >= 0; j--) {
    retire_touch(touched);
    if (j > 0) touched = touch_row(p.masks, S, s2, pl.lane, WAVES, wv);
    s0 = s1;
    s1 = s2;
    if (j > 1) s2 = st[j - 2];
    RL_TICK(0);
    const double nx_j = nx[j + 1], cf_j = cfp[j];  // used after the sum (see paint_forward)
    const double b1 = div_by_const(cfac, ntheta, c.inv_ntheta);     // cfac / ntheta, :474
    const double bt = div_by_const(cfac, theta, c.inv_theta) - b1;  // cfac / theta - b1, :475
    set_slot<S>(b, pl.jk, pl.kbit, -b1);   // donor k: (-b1) + b1 = +0.0 (never a mismatch with itself)
    double lsum = 0.0;
    MaskRow vrow = (MaskRow)(p.masks + ((size_t)(p.L + 1) * WAVES + wv) * S);
    asm volatile("" : "+s"(vrow));
    RL_TICK(1);
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
        lsum += x[jj];  // the lane's share of :495-503
      }
    });
    RL_TICK(2);
    const MaskTerm<S> term{rowh, b, theta, ntheta, p.stats ? p.stats + 8 : nullptr};
    rown = rowh;
    rowh = site_row(p.masks, S, p.L, s1, WAVES, wv);
    if (MODE == 0) {  // lanes: the sum reads no masks, request the next step's first chunks across it
      firstn = load_masks<4>(rown, 0);
      firsth = load_masks<4>(rowh, 0);
    }
    bsum = wave_sum<MODE, S, WAVES>(term, lsum, lk);  // :495-503
    if (MODE != 0) {
      firstn = load_masks<4>(rown, 0);
      firsth = load_masks<4>(rowh, 0);
    }
    RL_TICK(3);
    ls += nx_j;  // :471-472
    cfac = bsum;
    if (cfac < c.lower || cfac > c.upper) {  // :538-551
#pragma unroll
      for (int i = 0; i < S; i++) b[i] /= bsum;
      ls += fast_log_dev((float)bsum);
      cfac = 1.0;
    }
    cfac *= cf_j;  // :553-556
    RL_TICK(4);
    RL_TOCK(bseg1, 0, 1); RL_TOCK(bseg2, 1, 2); RL_TOCK(bseg3, 2, 3); RL_TOCK(bseg4, 3, 4);
    while (next_stone == j) {  // :559-578
      write_stone(0.0f);
      next_stone = stone_index(we);
    }
  }
  retire_touch(touched);
#ifdef RL_STATS
  if (p.stats && pl.lane == 0 && wv == 0) {  // whole step | divisions + slot + loads | chunk loop | sum | rescale test + factor
    atomicAdd(&p.stats[20], bseg1 + bseg2 + bseg3 + bseg4);
    atomicAdd(&p.stats[21], bseg1); atomicAdd(&p.stats[22], bseg2); atomicAdd(&p.stats[23], bseg3);
    atomicAdd(&p.stats[24], bseg4);
  }
#endif
}

// S <= 80: hold the kernel to 256 registers so that two waves share a SIMD.
// WAVES = 2: a workgroup of two waves paints one target (N > 5120).
template <int S, int TAIL, int MODE, int WAVES, int DIR>
__global__ void __launch_bounds__(64 * WAVES, 2) paint_kernel(const PaintParams p) {
  __shared__ float stage[WAVES][16 * 64];
  __shared__ WaveLinkStorage link;
  WaveLink<WAVES> lk;
  lk.s = &link;
  lk.w = WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  int b = blockIdx.x;
  bool backward = DIR == 1;
  if (DIR == 2) {
    if (p.merge_order == 1) {
      backward = !(b & 1);
      b >>= 1;
    } else {
      backward = b < p.nloc;
      if (!backward) b -= p.nloc;
    }
  }
  const int k = p.order[b];
#ifdef RL_STATS
  // experiment builds: the counters are gathered in LDS (an atomic to global memory per sum and counter made the
  // kernel 25 x slower and the cycle counts meaningless) and added to the global ones once per workgroup
  __shared__ unsigned long long lstats[32];
  if (threadIdx.x < 32) lstats[threadIdx.x] = 0;
  __syncthreads();
  PaintParams q = p;
  if (p.stats) q.stats = lstats;
  if (backward)
    paint_backward<S, TAIL, MODE, WAVES>(q, k, stage[lk.w], lk);
  else
    paint_forward<S, TAIL, MODE, WAVES>(q, k, stage[lk.w], lk);
  __syncthreads();
  if (p.stats && threadIdx.x < 32 && lstats[threadIdx.x]) atomicAdd(&p.stats[threadIdx.x], lstats[threadIdx.x]);
#else
  if (backward)
    paint_backward<S, TAIL, MODE, WAVES>(p, k, stage[lk.w], lk);
  else
    paint_forward<S, TAIL, MODE, WAVES>(p, k, stage[lk.w], lk);
#endif
}

template <int S, int TAIL, int WAVES>
static hipError_t launch_paint_t(const PaintParams &p, int dir, hipStream_t stream) {
  const dim3 grid(dir == 2 ? 2 * p.nloc : p.nloc), block(64 * WAVES);
  const int lds = 0;
  if (dir == 2)
    hipLaunchKernelGGL((paint_kernel<S, TAIL, RL_MODE, WAVES, 2>), grid, block, lds, stream, p);
  else if (dir == 1)
    hipLaunchKernelGGL((paint_kernel<S, TAIL, RL_MODE, WAVES, 1>), grid, block, lds, stream, p);
  else
    hipLaunchKernelGGL((paint_kernel<S, TAIL, RL_MODE, WAVES, 0>), grid, block, lds, stream, p);
  return hipGetLastError();
}

template <>
hipError_t launch_paint_mode<RL_MODE>(const PaintParams &p, int S, int waves, int dir, hipStream_t stream) {
  if (waves == 1) {
    switch (S) {
#define RL_CASE(s, t) \
  case s:             \
    return launch_paint_t<s, t, 1>(p, dir, stream);
      RL_FOR_EACH_S(RL_CASE)
#undef RL_CASE
    }
  } else if (waves == 2) {
#ifndef RL_ONLY_S
    switch (S) {
#define RL_CASE(s, t) \
  case s:             \
    return launch_paint_t<s, t, 2>(p, dir, stream);
      RL_FOR_EACH_S_2WAVES(RL_CASE)
#undef RL_CASE
    }
#endif
  }
  return hipErrorInvalidValue;
}

}  // namespace rl
