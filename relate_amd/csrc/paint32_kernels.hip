// paint32_kernels.hip -- K1 in the FAST mode RL_SUM_LANES32: stepping-stone painting with the per-donor state in
// packed FP32 (fast_painting.cpp:288-303, 481-503 restated, NOT bit-identical to the reference).
//
// Why: the exact kernels are bound by FP64 issue -- 3 (forward) + 6 (backward) instructions per donor and site before
// any summation order is paid for, and FP64 / non-packed FP32 instructions issue at one per lane and cycle.  Only the
// PACKED FP32 forms (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32: two donors per lane and instruction) go faster.  The
// stepping stones are floats anyway (fast_painting.cpp:241-245, 566-571); what the state loses against doubles is
// far inside the tolerance on the distance matrix (tools/exp_fp32_state.py: 0.02 of 1e-5 max(|d|, |logscale|) at
// N = 5000; asserted against the reference's matrices in tests/test_n5000_gpu.py).
//
// Layout: the lane's run of donors as in the FP64 kernels (paint_device.h), donors 2p / 2p+1 of the run in the halves
// of register pair p; the lane-mask panel is used as it is (word 2p masks the low halves, word 2p+1 the high ones).
// The mismatch factor is applied by narrowing EXEC for one non-packed instruction per half -- per donor and site:
//   forward   a = (a + c) * (mis ? K : 1); sum += a          pk_add, 2 masked v_mul_f32, pk_add      = 2 per donor
//   backward  b = (b + mis bt + b1) * (mis ? K : 1)          pk_add, 2 masked v_fma_f32 (t K + bt K)
//             sum += (mis' ? theta : ntheta) * b              pk_add (all), 2 masked v_add_f32 (mis') = 3 per donor
// (the weighted sum as ntheta * sum_all + (theta - ntheta) * sum_mis'), against 3 + 6 in the `lanes` order on doubles.
// The lanes' sums are FP32 partials (two chains per half), widened and reduced over the wave in double as in `lanes`;
// the step's factor, logscale and rescaling tests stay double.  80 state registers instead of 160: four waves a SIMD.
#include <cstdlib>
#include "paint_device.h"
#include "exact_sum.h"
#include "launch.h"

// experiment knobs (tools/build_paint_variant.sh): waves per SIMD the kernel is held to, mask words per chunk
#ifndef RL32_WAVES_PER_SIMD
#define RL32_WAVES_PER_SIMD 4
#endif
#ifndef RL32_FWD_CH
#define RL32_FWD_CH 8  // (16: 61.6 ms, 8: 59.7 ms per Paint of the L = 100k cut of C3)
#endif
#ifndef RL32_BWD_CH
#define RL32_BWD_CH 4
#endif

namespace rl {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) PaintParams *ColdParams32;

RL_DEV f32x2 splat(float v) { return f32x2{v, v}; }
// (clang selects v_pk_add_f32 / v_pk_mul_f32 for <2 x float> arithmetic on gfx950; -ffp-contract=off keeps them apart.
//  The masked instructions work on ONE half of a register pair: the halves are taken out of the vector, go through
//  the asm as 32-bit operands and are put back -- sub-register copies the register coalescer removes.)
// p0, p1 *= k in the lanes of m0 (p0.x), m1 (p0.y), m2 (p1.x), m3 (p1.y), under one return to exec = -1
RL_DEV void masked_mul_f32x4(f32x2 &p0, f32x2 &p1, u64 m0, u64 m1, u64 m2, u64 m3, float k) {
  float x0 = p0.x, x1 = p0.y, x2 = p1.x, x3 = p1.y;
  asm volatile(
      "s_mov_b64 exec, %4\n\tv_mul_f32 %0, %0, %8\n\t"
      "s_mov_b64 exec, %5\n\tv_mul_f32 %1, %1, %8\n\t"
      "s_mov_b64 exec, %6\n\tv_mul_f32 %2, %2, %8\n\t"
      "s_mov_b64 exec, %7\n\tv_mul_f32 %3, %3, %8\n\t"
      "s_mov_b64 exec, -1"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)
      : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "v"(k));
  p0 = f32x2{x0, x1};
  p1 = f32x2{x2, x3};
}
RL_DEV void masked_mul_f32x8(f32x2 &p0, f32x2 &p1, f32x2 &p2, f32x2 &p3, u64 m0, u64 m1, u64 m2, u64 m3, u64 m4, u64 m5,
                             u64 m6, u64 m7, float k) {
  float x0 = p0.x, x1 = p0.y, x2 = p1.x, x3 = p1.y, x4 = p2.x, x5 = p2.y, x6 = p3.x, x7 = p3.y;
  asm volatile(
      "s_mov_b64 exec, %8\n\tv_mul_f32 %0, %0, %16\n\t"
      "s_mov_b64 exec, %9\n\tv_mul_f32 %1, %1, %16\n\t"
      "s_mov_b64 exec, %10\n\tv_mul_f32 %2, %2, %16\n\t"
      "s_mov_b64 exec, %11\n\tv_mul_f32 %3, %3, %16\n\t"
      "s_mov_b64 exec, %12\n\tv_mul_f32 %4, %4, %16\n\t"
      "s_mov_b64 exec, %13\n\tv_mul_f32 %5, %5, %16\n\t"
      "s_mov_b64 exec, %14\n\tv_mul_f32 %6, %6, %16\n\t"
      "s_mov_b64 exec, %15\n\tv_mul_f32 %7, %7, %16\n\t"
      "s_mov_b64 exec, -1"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
      : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4), "s"(m5), "s"(m6), "s"(m7), "v"(k));
  p0 = f32x2{x0, x1};
  p1 = f32x2{x2, x3};
  p2 = f32x2{x4, x5};
  p3 = f32x2{x6, x7};
}
// x = x * k + a in the lanes of m
RL_DEV void masked_fma_f32x4(f32x2 &p0, f32x2 &p1, u64 m0, u64 m1, u64 m2, u64 m3, float k, float a) {
  float x0 = p0.x, x1 = p0.y, x2 = p1.x, x3 = p1.y;
  asm volatile(
      "s_mov_b64 exec, %4\n\tv_fma_f32 %0, %0, %8, %9\n\t"
      "s_mov_b64 exec, %5\n\tv_fma_f32 %1, %1, %8, %9\n\t"
      "s_mov_b64 exec, %6\n\tv_fma_f32 %2, %2, %8, %9\n\t"
      "s_mov_b64 exec, %7\n\tv_fma_f32 %3, %3, %8, %9\n\t"
      "s_mov_b64 exec, -1"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)
      : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "v"(k), "v"(a));
  p0 = f32x2{x0, x1};
  p1 = f32x2{x2, x3};
}
// the two above for one chunk of two pairs under ONE return to exec = -1: p = p * k + a in the lanes of mn, then
// s0 / s1 += the new halves in the lanes of mh
RL_DEV void masked_fma_acc_f32x4(f32x2 &p0, f32x2 &p1, float &s0, float &s1, u64 n0, u64 n1, u64 n2, u64 n3, u64 h0,
                                 u64 h1, u64 h2, u64 h3, float k, float a) {
  float x0 = p0.x, x1 = p0.y, x2 = p1.x, x3 = p1.y;
  asm volatile(
      "s_mov_b64 exec, %6\n\tv_fma_f32 %0, %0, %14, %15\n\t"
      "s_mov_b64 exec, %7\n\tv_fma_f32 %1, %1, %14, %15\n\t"
      "s_mov_b64 exec, %8\n\tv_fma_f32 %2, %2, %14, %15\n\t"
      "s_mov_b64 exec, %9\n\tv_fma_f32 %3, %3, %14, %15\n\t"
      "s_mov_b64 exec, %10\n\tv_add_f32 %4, %4, %0\n\t"
      "s_mov_b64 exec, %11\n\tv_add_f32 %5, %5, %1\n\t"
      "s_mov_b64 exec, %12\n\tv_add_f32 %4, %4, %2\n\t"
      "s_mov_b64 exec, %13\n\tv_add_f32 %5, %5, %3\n\t"
      "s_mov_b64 exec, -1"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(s0), "+v"(s1)
      : "s"(n0), "s"(n1), "s"(n2), "s"(n3), "s"(h0), "s"(h1), "s"(h2), "s"(h3), "v"(k), "v"(a));
  p0 = f32x2{x0, x1};
  p1 = f32x2{x2, x3};
}
// s0 += p0.x (lanes of m0), s1 += p0.y (m1), s0 += p1.x (m2), s1 += p1.y (m3)
RL_DEV void masked_acc_f32x4(float &s0, float &s1, f32x2 p0, f32x2 p1, u64 m0, u64 m1, u64 m2, u64 m3) {
  const float x0 = p0.x, x1 = p0.y, x2 = p1.x, x3 = p1.y;
  asm volatile(
      "s_mov_b64 exec, %6\n\tv_add_f32 %0, %0, %2\n\t"
      "s_mov_b64 exec, %7\n\tv_add_f32 %1, %1, %3\n\t"
      "s_mov_b64 exec, %8\n\tv_add_f32 %0, %0, %4\n\t"
      "s_mov_b64 exec, %9\n\tv_add_f32 %1, %1, %5\n\t"
      "s_mov_b64 exec, -1"
      : "+v"(s0), "+v"(s1)
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(m0), "s"(m1), "s"(m2), "s"(m3));
}
// p += k in the lanes of m0 (p.x) / m1 (p.y)
RL_DEV void masked_add_f32x2(f32x2 &p, u64 m0, u64 m1, float k) {
  float x0 = p.x, x1 = p.y;
  asm volatile("s_mov_b64 exec, %2\n\tv_add_f32 %0, %0, %4\n\ts_mov_b64 exec, %3\n\tv_add_f32 %1, %1, %4\n\ts_mov_b64 exec, -1"
               : "+v"(x0), "+v"(x1)
               : "s"(m0), "s"(m1), "v"(k));
  p = f32x2{x0, x1};
}
RL_DEV void masked_mov_f32(float &x, u64 m, float v) {
  asm volatile("s_mov_b64 exec, %1\n\tv_mov_b32 %0, %2\n\ts_mov_b64 exec, -1" : "+v"(x) : "s"(m), "v"(v));
}

// a[j] = v in the lanes of `bit`, j wave-uniform but dynamic (paint_device.h set_slot, for the float halves)
// (the pair is an INPUT whose register is overwritten behind the compiler's back, see paint_device.h poke_slot; the
//  half is named in the instruction: an operand of 64 bits prints as v[n:n+1])
template <int HALF>
RL_DEV void poke_slot32(const f32x2 &t, u64 mask, float v) {
  float h = HALF ? t.y : t.x;
  asm volatile("s_mov_b64 exec, %1\n\tv_mov_b32 %0, %2\n\ts_mov_b64 exec, -1" : : "v"(h), "s"(mask), "v"(v));
}
template <int S>
RL_DEV void pin_registers32(f32x2 (&a)[S / 2]) {
#pragma unroll
  for (int i = 0; i < S / 2; i += 4) asm volatile("" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]));
}
template <int S>
RL_DEV void set_slot32(f32x2 (&a)[S / 2], int j, u64 bit, float v) {
  static_assert(S <= 80 && S % 8 == 0, "extend the case list");
  pin_registers32<S>(a);
#define RL_SLOT(J)                                                                   \
  case J:                                                                            \
    if constexpr ((J) < S) poke_slot32<(J) & 1>(a[((J) < S ? (J) : 0) / 2], bit, v); \
    break;
#define RL_SLOT8(B) RL_SLOT(B) RL_SLOT(B + 1) RL_SLOT(B + 2) RL_SLOT(B + 3) RL_SLOT(B + 4) RL_SLOT(B + 5) RL_SLOT(B + 6) RL_SLOT(B + 7)
  switch (j) {
    RL_SLOT8(0) RL_SLOT8(8) RL_SLOT8(16) RL_SLOT8(24) RL_SLOT8(32) RL_SLOT8(40) RL_SLOT8(48) RL_SLOT8(56)
    RL_SLOT8(64) RL_SLOT8(72)
    default: break;
  }
#undef RL_SLOT8
#undef RL_SLOT
  pin_registers32<S>(a);
}

// one stepping stone, donor order (paint_kernels.hip emit_stone, the state being floats already)
template <int S>
RL_DEV void emit_stone32(const PaintLane<S> &pl, const f32x2 (&v)[S / 2], float *__restrict__ out, float self_value,
                         float *stage) {
  constexpr int R = S % 16 == 0 ? 16 : 8;
#pragma unroll
  for (int c = 0; c < S / R; c++) {
#pragma unroll
    for (int ii = 0; ii < R; ii += 2) {
      f32x2 x = v[(c * R + ii) / 2];
      asm volatile("" : "+v"(x) : : "memory");
      stage[ii * 64 + pl.lane] = x.x;
      stage[(ii + 1) * 64 + pl.lane] = x.y;
    }
#pragma clang loop unroll(disable)
    for (int ii = 0; ii < R; ii++) {
      const int i = c * R + ii;
      const int n = pl.start + i;
      if (i < pl.len) out[n] = (n == pl.k) ? self_value : stage[ii * 64 + pl.lane];
    }
  }
}

// the wave's (workgroup's) sum of per-lane FP32 partials, in double: the `lanes` reduction (exact_sum.h wave_sum<0>)
template <int S, int WAVES>
RL_DEV double lanes_total(double lane_sum, WaveLink<WAVES> &lk) {
  lk.phase++;
  const double t = wave_sum_butterfly(lane_sum);
  if constexpr (WAVES == 1) {
    return t;
  } else {
    const unsigned ph = lk.phase & 1u;
    if ((threadIdx.x & 63) == 0) lk.s->tot[ph][lk.w] = t;
    lk.barrier();
    return lk.s->tot[ph][0] + lk.s->tot[ph][1];
  }
}

template <int S, int TAIL, int WAVES>
RL_DEV void paint32_forward(const PaintParams &p, int k, float *stage, WaveLink<WAVES> &lk) {
  static_assert(TAIL % 2 == 0 && S % 8 == 0, "pairs");
  const int wv = lk.w;
  PaintLane<S> pl;
  pl.init(p.lay, k, wv);
  const PaintConsts &c = p.c;
  const int64_t off = p.plan_off[k];
  const int D = (int)(p.plan_off[k + 1] - off);
  const int32_t *__restrict__ st = p.sites + off;
  const double *__restrict__ cfp = p.cf + off;
  const double *__restrict__ nx = p.nxt + off;
  constexpr int P = S / 2;
  f32x2 a[P];

  // ---- SNP 0 (fast_painting.cpp:207-253)
  const float init0 = (float)c.init0, init1 = (float)c.init1;
  for_each_chunk<S, 8>(site_row(p.masks, S, p.L, st[0], WAVES, wv), [&](int j0, const u64x8 &m) {
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
      float v = init0;
      masked_mov_f32(v, m[jj], init1);
      if (j0 + jj >= S - TAIL) masked_mov_f32(v, ~pl.valid(j0 + jj), 0.0f);
      if (jj & 1)
        a[(j0 + jj) / 2].y = v;
      else
        a[(j0 + jj) / 2].x = v;
    }
  });
  set_slot32<S>(a, pl.jk, pl.kbit, 0.0f);
  auto lane_sum = [&]() {
    f32x2 s0 = splat(0.f), s1 = splat(0.f);
#pragma unroll
    for (int q = 0; q < P; q += 2) {
      s0 += a[q];
      s1 += a[q + 1];
    }
    s0 += s1;
    return (double)s0.x + (double)s0.y;
  };
  double ssum = lanes_total<S, WAVES>(lane_sum(), lk);
  double ls = 0.0;
  int wa = 0;
  auto stone_index = [&](int w) {
    const ColdParams32 cp = cold_params<PaintParams>();
    return w < cp->W ? cp->stone_ia[(size_t)k * cp->W + w] : -1;
  };
  auto write_stone = [&]() {
    const ColdParams32 cp = cold_params<PaintParams>();
    const size_t N = cp->lay.N, row = (size_t)wa * cp->nloc + (k - cp->k0);
    emit_stone32<S>(pl, a, cp->alpha + row * N, 0.0f, stage);
    if (pl.lane == 0 && wv == 0) cp->ls_alpha[row] = (float)ls;
    wa++;
  };
  int next_stone = stone_index(0);
  while (next_stone == 0) {
    write_stone();
    next_stone = stone_index(wa);
  }
  double cfac = cfp[0] * ssum;  // :260

  int s1 = D > 1 ? st[1] : 0, s2 = D > 2 ? st[2] : 0;
  uint32_t touched = 0;
  float K1 = (float)c.K1;
  asm volatile("" : "+v"(K1));
  constexpr int CH = (S % 16 == 0 && RL32_FWD_CH == 16) ? 16 : 8;  // mask words (= donors per lane) per chunk
  typedef typename MaskChunk<CH>::type Chunk;
  MaskRow row = site_row(p.masks, S, p.L, s1, WAVES, wv);
  Chunk first = load_masks<CH>(row, 0);
  for (int i = 1; i < D; i++) {
    retire_touch(touched);
    if (i + 1 < D) touched = touch_row(p.masks, S, s2, pl.lane, WAVES, wv);
    s1 = s2;
    if (i + 2 < D) s2 = st[i + 2];
    const double nx_i = nx[i - 1], cf_i = cfp[i];
    const float cf32 = (float)cfac;
    set_slot32<S>(a, pl.jk, pl.kbit, -cf32);  // donor k: (-c) + c = +0.0
    const f32x2 c2 = splat(cf32);
    f32x2 sa = splat(0.f), sb = splat(0.f);
    for_each_chunk_from<S, CH>(row, first, [&](int j0, const Chunk &m) {  // :288-295
      f32x2 v[CH / 2];
#pragma unroll
      for (int q = 0; q < CH / 2; q++) {
        v[q] = a[j0 / 2 + q];
        if (j0 + 2 * q + 1 < S - TAIL) {
          v[q] += c2;
        } else {  // slots past the lane's run stay +0.0
          masked_add_f32x2(v[q], pl.valid(j0 + 2 * q), pl.valid(j0 + 2 * q + 1), cf32);
        }
      }
#pragma unroll
      for (int q = 0; q < CH / 2; q += 4)
        masked_mul_f32x8(v[q], v[q + 1], v[q + 2], v[q + 3], m[2 * q], m[2 * q + 1], m[2 * q + 2], m[2 * q + 3], m[2 * q + 4],
                         m[2 * q + 5], m[2 * q + 6], m[2 * q + 7], K1);
#pragma unroll
      for (int q = 0; q < CH / 2; q += 2) {
        a[j0 / 2 + q] = v[q];
        a[j0 / 2 + q + 1] = v[q + 1];
        sa += v[q];
        sb += v[q + 1];
      }
    });
    row = site_row(p.masks, S, p.L, s1, WAVES, wv);
    first = load_masks<CH>(row, 0);
    sa += sb;
    ssum = lanes_total<S, WAVES>((double)sa.x + (double)sa.y, lk);
    ls += nx_i;  // :281-282
    cfac = ssum;
    if (cfac < c.lower || cfac > c.upper) {  // :334-347
      const f32x2 inv = splat((float)(1.0 / ssum));
#pragma unroll
      for (int q = 0; q < P; q++) a[q] *= inv;
      ls += log(ssum);
      cfac = 1.0;
    }
    cfac *= cf_i;  // :349-352
    while (next_stone == i) {  // :354-374
      write_stone();
      next_stone = stone_index(wa);
    }
  }
  retire_touch(touched);
}

template <int S, int TAIL, int WAVES>
RL_DEV void paint32_backward(const PaintParams &p, int k, float *stage, WaveLink<WAVES> &lk) {
  const int wv = lk.w;
  PaintLane<S> pl;
  pl.init(p.lay, k, wv);
  const PaintConsts &c = p.c;
  const int64_t off = p.plan_off[k];
  const int D = (int)(p.plan_off[k + 1] - off);
  const int32_t *__restrict__ st = p.sites + off;
  const double *__restrict__ cfp = p.cf + off;
  const double *__restrict__ nx = p.nxt + off;
  constexpr int P = S / 2;
  f32x2 b[P];

  // ---- last SNP (:396-448)
  double ls = c.log_Nm1 - D * c.log_ntheta;
#pragma unroll
  for (int i = 0; i < S; i++) {
    float v = 1.0f;
    if (i >= S - TAIL) masked_mov_f32(v, ~pl.valid(i), 0.0f);
    if (i & 1)
      b[i / 2].y = v;
    else
      b[i / 2].x = v;
  }
  set_slot32<S>(b, pl.jk, pl.kbit, 0.0f);
  double bsum = p.binit[k];
  int we = p.W - 1;
  auto stone_index = [&](int w) {
    const ColdParams32 cp = cold_params<PaintParams>();
    return w >= 0 ? cp->stone_ie[(size_t)k * cp->W + w] : -2;
  };
  auto write_stone = [&](float self_value) {
    const ColdParams32 cp = cold_params<PaintParams>();
    const size_t N = cp->lay.N, row = (size_t)we * cp->nloc + (k - cp->k0);
    emit_stone32<S>(pl, b, cp->beta + row * N, self_value, stage);
    if (pl.lane == 0 && wv == 0) cp->ls_beta[row] = (float)ls;
    we--;
  };
  int next_stone = stone_index(we);
  while (next_stone == D - 1) {
    write_stone(1.0f);
    next_stone = stone_index(we);
  }
  double cfac = cfp[D - 1] * bsum;  // :454-455

  int s0 = st[D - 1], s1 = D > 1 ? st[D - 2] : 0, s2 = D > 2 ? st[D - 3] : 0;
  uint32_t touched = 0;
  MaskRow rown = site_row(p.masks, S, p.L, s0, WAVES, wv);
  MaskRow rowh = site_row(p.masks, S, p.L, s1, WAVES, wv);
  constexpr int BCH = (S % 8 == 0 && RL32_BWD_CH == 8 && TAIL % 8 == 0) ? 8 : 4;
  typedef typename MaskChunk<BCH>::type BChunk;
  BChunk firstn = load_masks<BCH>(rown, 0), firsth = load_masks<BCH>(rowh, 0);
  float K1 = (float)c.K1;
  asm volatile("" : "+v"(K1));
  const double theta = c.theta, ntheta = c.ntheta;
  for (int j = D - 2; j >= 0; j--) {
    retire_touch(touched);
    if (j > 0) touched = touch_row(p.masks, S, s2, pl.lane, WAVES, wv);
    s0 = s1;
    s1 = s2;
    if (j > 1) s2 = st[j - 2];
    const double nx_j = nx[j + 1], cf_j = cfp[j];
    const double b1d = cfac * c.inv_ntheta;        // cfac / ntheta, :474 (fast mode: the reciprocal)
    const double btd = cfac * c.inv_theta - b1d;   // :475
    const float b1 = (float)b1d, btK = (float)(btd * c.K1);
    set_slot32<S>(b, pl.jk, pl.kbit, -b1);  // donor k: (-b1) + b1 = +0.0 (never a mismatch with itself)
    const f32x2 b12 = splat(b1);
    f32x2 sall = splat(0.f);
    float smis0 = 0.f, smis1 = 0.f;
    MaskRow vrow = (MaskRow)(p.masks + ((size_t)(p.L + 1) * WAVES + wv) * S);
    asm volatile("" : "+s"(vrow));
    for_each_chunk2_tail<S, BCH, TAIL>(rown, rowh, vrow, firstn, firsth,
                                       [&](int j0, const BChunk &mn, const BChunk &mh, const BChunk &va) {
      f32x2 v[BCH / 2];
#pragma unroll
      for (int q = 0; q < BCH / 2; q++) {
        v[q] = b[j0 / 2 + q];
        if (j0 + 2 * q + 2 <= S - TAIL)
          v[q] += b12;
        else
          masked_add_f32x2(v[q], va[2 * q], va[2 * q + 1], b1);
      }
      // mismatch at the later site: ((b + b1) + bt) K as (b + b1) K + bt K
#pragma unroll
      for (int q = 0; q < BCH / 2; q += 2)  // ... and the lanes that mismatch at this site into their sum (:495-503)
        masked_fma_acc_f32x4(v[q], v[q + 1], smis0, smis1, mn[2 * q], mn[2 * q + 1], mn[2 * q + 2], mn[2 * q + 3], mh[2 * q],
                             mh[2 * q + 1], mh[2 * q + 2], mh[2 * q + 3], K1, btK);
#pragma unroll
      for (int q = 0; q < BCH / 2; q++) {
        b[j0 / 2 + q] = v[q];
        sall += v[q];
      }
    });
    rown = rowh;
    rowh = site_row(p.masks, S, p.L, s1, WAVES, wv);
    firstn = load_masks<BCH>(rown, 0);
    firsth = load_masks<BCH>(rowh, 0);
    const double lane_all = (double)sall.x + (double)sall.y, lane_mis = (double)smis0 + (double)smis1;
    bsum = lanes_total<S, WAVES>(ntheta * lane_all + (theta - ntheta) * lane_mis, lk);
    ls += nx_j;  // :471-472
    cfac = bsum;
    if (cfac < c.lower || cfac > c.upper) {  // :538-551
      const f32x2 inv = splat((float)(1.0 / bsum));
#pragma unroll
      for (int q = 0; q < P; q++) b[q] *= inv;
      ls += fast_log_dev((float)bsum);
      cfac = 1.0;
    }
    cfac *= cf_j;  // :553-556
    while (next_stone == j) {  // :559-578
      write_stone(0.0f);
      next_stone = stone_index(we);
    }
  }
  retire_touch(touched);
}

// 80 state registers per lane: held to 128 VGPRs, four waves share a SIMD
template <int S, int TAIL, int WAVES, int DIR>
__global__ void __launch_bounds__(64 * WAVES, WAVES == 1 ? RL32_WAVES_PER_SIMD : RL32_WAVES_PER_SIMD / 2) paint32_kernel(const PaintParams p) {
  __shared__ float stage[WAVES][16 * 64];
  __shared__ WaveLinkStorage link;
  WaveLink<WAVES> lk;
  lk.s = &link;
  lk.w = WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  int b = blockIdx.x;
  bool backward = DIR == 1;
  if (DIR == 2) {
    backward = b < p.nloc;
    if (!backward) b -= p.nloc;
  }
  const int k = p.order[b];
  if (backward)
    paint32_backward<S, TAIL, WAVES>(p, k, stage[lk.w], lk);
  else
    paint32_forward<S, TAIL, WAVES>(p, k, stage[lk.w], lk);
}

template <int S, int TAIL, int WAVES>
static hipError_t launch_paint32_t(const PaintParams &p, int dir, hipStream_t stream) {
  const dim3 grid(dir == 2 ? 2 * p.nloc : p.nloc), block(64 * WAVES);
  if (dir == 2)
    hipLaunchKernelGGL((paint32_kernel<S, TAIL, WAVES, 2>), grid, block, 0, stream, p);
  else if (dir == 1)
    hipLaunchKernelGGL((paint32_kernel<S, TAIL, WAVES, 1>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((paint32_kernel<S, TAIL, WAVES, 0>), grid, block, 0, stream, p);
  return hipGetLastError();
}

template <>
hipError_t launch_paint_mode<3>(const PaintParams &p, int S, int waves, int dir, hipStream_t stream) {
  if (waves == 1) {
    switch (S) {
#define RL_CASE(s, t) \
  case s:             \
    return launch_paint32_t<s, t, 1>(p, dir, stream);
      RL_FOR_EACH_S(RL_CASE)
#undef RL_CASE
    }
  } else if (waves == 2) {
#ifndef RL_ONLY_S
    switch (S) {
#define RL_CASE(s, t) \
  case s:             \
    return launch_paint32_t<s, t, 2>(p, dir, stream);
      RL_FOR_EACH_S_2WAVES(RL_CASE)
#undef RL_CASE
    }
#endif
  }
  return hipErrorInvalidValue;
}

}  // namespace rl
