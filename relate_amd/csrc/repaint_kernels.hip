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
#include <algorithm>
#include <atomic>

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

// ---- rows of the scratch strip / of the posterior slab: [register][lane], one address per chunk of 8 registers
// (the registers of a chunk are immediates from it), so that no table of addresses is kept
typedef __attribute__((address_space(1))) double *GlobalF64;
typedef const __attribute__((address_space(1))) double *GlobalF64In;
typedef __attribute__((address_space(1))) float *GlobalF32;
template <int S>
RL_DEV void store_row(double *row_lane, const double (&a)[S]) {
#pragma unroll
  for (int c = 0; c < S / 8; c++) {
    GlobalF64 q = (GlobalF64)(row_lane + c * 8 * 64);
    asm volatile("" : "+v"(q));
#pragma unroll
    for (int jj = 0; jj < 8; jj++) q[jj * 64] = a[c * 8 + jj];
  }
}
template <int S>
RL_DEV void load_row(const double *row_lane, double (&a)[S]) {
#pragma unroll
  for (int c = 0; c < S / 8; c++) {
    GlobalF64In q = (GlobalF64In)(row_lane + c * 8 * 64);
    asm volatile("" : "+v"(q));
#pragma unroll
    for (int jj = 0; jj < 8; jj++) a[c * 8 + jj] = q[jj * 64];
  }
}
// the double that lane `l` holds (l wave-uniform), as a uniform value
RL_DEV double lane_value(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// ---------------- K2a: the forward pass of one target (fast_painting.cpp:769-885)
// Leaves every CK-th alpha row (double) in the target's strip of checkpoint rows and one side record per step.
template <int S, int TAIL, int MODE, int WAVES>
RL_DEV void repaint_forward(const RepaintParams &p, int n, float *stage, WaveLink<WAVES> &lk) {
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
  const double cf_last = p.cf_last[t];
  constexpr int CK = REPAINT_CHECKPOINT;
  constexpr int ROW = S * 64 * WAVES;  // a checkpoint row: [wave][register][lane]
  double *__restrict__ side = p.side + p.top_off[t] * REPAINT_SIDE;  // [step][cfac, rescaling divisor or 0, logscale]
  double *__restrict__ ckrows = p.scratch + p.ck_off[t] * (int64_t)ROW + (size_t)wv * (S * 64) + pl.lane;
  const bool scribe = pl.lane == 0 && wv == 0;  // writes the side records
  constexpr int CH = S % 16 == 0 ? 16 : 8;  // registers per chunk of masks
  typedef typename MaskChunk<CH>::type Chunk;
  const double K1 = in_vgpr(c.K1);

  double a[S];
  // (a later launch of a bounded window starts from the state an earlier one left behind row f0, a checkpoint row)
  const int f0 = (p.fstate && p.partial) ? max(0, __builtin_amdgcn_readfirstlane(p.fstart_row[t])) : 0;
  const int fsave = p.fstate ? __builtin_amdgcn_readfirstlane(p.fsave_row[t]) : -1;
  double *__restrict__ fst = p.fstate ? p.fstate + ((size_t)t * WAVES + wv) * (S * 64) + pl.lane : nullptr;
  double ssum, cfac, prev_ls;
  float lsf;
  if (f0 == 0) {
    const ColdRepaint cp = cold_params<RepaintParams>();
    load_stone<S>(pl, cp->alpha_begin + (size_t)t * cp->lay.N, a, stage);
    set_slot<S>(a, pl.jk, pl.kbit, 0.0);  // alpha[n] = 0 for the target itself (:781)
    ssum = wave_sum<MODE, S, WAVES>(RegTerm<S>{a}, local_sum<S>(RegTerm<S>{a}), lk);
    lsf = p.ls_alpha[t];
    prev_ls = (double)lsf;
    cfac = (D == 1 ? cf_last : cfp[0]) * ssum;
  } else {
    load_row<S>(fst, a);
    cfac = p.fscal[(size_t)t * 4];
    prev_ls = p.fscal[(size_t)t * 4 + 1];
    lsf = (float)p.fscal[(size_t)t * 4 + 2];
    ssum = 0.0;
  }
  // the row the pass starts from is a checkpoint (row 0 always is) -- kept if the backward pass rebuilds rows from its
  // block (a later launch of a bounded window addresses the strips compactly, from the block of row_lo on: window.cpp)
  if (!p.partial || f0 + CK > (int)p.row_lo[t]) store_row<S>(ckrows + (int64_t)(f0 / CK) * ROW, a);
  if (scribe) {
    side[(size_t)f0 * REPAINT_SIDE + 0] = 0.0;
    side[(size_t)f0 * REPAINT_SIDE + 1] = 0.0;
    side[(size_t)f0 * REPAINT_SIDE + 2] = (double)lsf;
  }
  int s1 = D > f0 + 1 ? st[f0 + 1] : 0, s2 = D > f0 + 2 ? st[f0 + 2] : 0;  // row pipeline as in paint_forward
  uint32_t touched = 0;
  MaskRow row = site_row(p.masks, S, p.L, s1, WAVES, wv);
  Chunk first = load_masks<CH>(row, 0);
  // (a later launch of a bounded window: rows from row_hi on are of no use to the backward pass)
  const int Dfwd = p.partial ? min(D, max(1, (int)p.row_hi[t])) : D;
  const int ck_from = p.partial ? (int)p.row_lo[t] : 0;
  for (int i = f0 + 1; i < Dfwd; i++) {
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
    // checkpoint row (a later launch of a bounded window: only the blocks the backward pass rebuilds rows from)
    if (i % CK == 0 && (!p.partial || i + CK > ck_from)) store_row<S>(ckrows + (int64_t)(i / CK) * ROW, a);
    if (scribe) {  // what the backward pass needs to redo this step from the previous row
      side[(size_t)i * REPAINT_SIDE + 0] = cfac_used;
      side[(size_t)i * REPAINT_SIDE + 1] = divisor;
      side[(size_t)i * REPAINT_SIDE + 2] = (double)lsf;
    }
    if (i == fsave) {  // the state behind row i, for the launch of the next part (window.cpp: place_rows)
      store_row<S>(fst, a);
      if (scribe) {
        p.fscal[(size_t)t * 4] = cfac;
        p.fscal[(size_t)t * 4 + 1] = prev_ls;
        p.fscal[(size_t)t * 4 + 2] = (double)lsf;
      }
    }
  }
  retire_touch(touched);
}

// ---------------- K2b: the backward pass of one target (:887-1073) and its posterior rows
// The block's checkpoint row is held on chip while the pass walks down the block: registers 0 .. LREG-1 in the
// wave's LDS strip [register][lane] (filled by global->LDS loads that bypass the VGPRs), the rest in VGPRs.
// NOSTRIP (a bounded window's part launches): nothing is held -- a product row reads the block's checkpoint row
// where it lies, a chunk of 8 registers at a time (41 KB per target and row, through the L2: the forward kernel of the
// same launch has just written it, and a part is ~6 rows per target) -- so the kernel needs no strip and TWO waves
// share a SIMD: a part launch is mostly the beta-only way down to the part, and those loops run 1.77 x slower with
// one wave a SIMD.  The whole-window pass (200 rows per target) keeps the strip: there the re-reads would be traffic.
template <int S, bool NOSTRIP = false>
struct HeldRow {
  static constexpr int LREG = NOSTRIP ? 0 : (S < 76 ? S : 76);  // 76 * 512 B = 38 KB: four waves of a CU fit their strips in LDS
  static constexpr int VREG = NOSTRIP ? 0 : S - LREG;
  double *lds;  // the wave's strip (wave-uniform)
  double v[VREG > 0 ? VREG : 1];
  const __attribute__((address_space(1))) double *grow = nullptr;  // (NOSTRIP) the row, [register][lane], at this lane
  RL_DEV double get(int i, int lane) const {
    if constexpr (NOSTRIP) return grow[i * 64];
    return i < LREG ? lds[i * 64 + lane] : v[i - LREG < 0 ? 0 : i - LREG];
  }
  // request the [register][lane] row of doubles at `row` (wave-uniform)
  RL_DEV void request(const double *row, int lane) {
    if constexpr (NOSTRIP) {
      grow = (const __attribute__((address_space(1))) double *)(row + lane);
      return;
    }
    typedef const __attribute__((address_space(1))) void *GP;
    typedef __attribute__((address_space(3))) void *LP;
    // one scalar base per 4 KB (four instructions with immediate offsets 0 .. 3 KB), the lane's 16 bytes as a
    // 32-bit offset: the address registers are not rewritten between the requests
    const uint64_t r64 = (uint64_t)row;
    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(r64 >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)r64);
    const uint32_t voff = (uint32_t)lane * 16u;
#ifdef RL_K2_NO_DMA
    (void)base; (void)voff;
#pragma unroll
    for (int i = 0; i < LREG; i++) lds[i * 64 + lane] = row[i * 64 + lane];
#else
    static_assert(LREG % 2 == 0, "the strip is requested a KB (two registers) at a time, 4 KB per address");
#pragma unroll
    for (int g = 0; g < (LREG + 7) / 8; g++) {
      GP src = (GP)((const char *)base + g * 4096 + voff);
      LP dst = (LP)(lds + g * 512);
      __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
      if (g * 8 + 2 < LREG) __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
      if (g * 8 + 4 < LREG) __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0);
      if (g * 8 + 6 < LREG) __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0);
    }
#endif
    if constexpr (VREG > 0) {
      GlobalF64In q = (GlobalF64In)(row + LREG * 64 + lane);
      asm volatile("" : "+v"(q));
#pragma unroll
      for (int i = 0; i < VREG; i++) v[i] = q[i * 64];
    }
  }
};
// Posterior row of the block's checkpoint row itself (no forward step): trow = float(ck * b)
template <int S, bool NS>
RL_DEV void product_last(const HeldRow<S, NS> &ck, int lane, const double (&b)[S], float *trow_lane) {
#pragma unroll
  for (int c = 0; c < S / 8; c++) {
    GlobalF32 q = (GlobalF32)(trow_lane + c * 8 * 64);
    asm volatile("" : "+v"(q));
#pragma unroll
    for (int jj = 0; jj < 8; jj++) q[jj * 64] = (float)(ck.get(c * 8 + jj, lane) * b[c * 8 + jj]);
  }
}
// Posterior row r steps above the checkpoint row: every chunk of 8 registers walks the steps cp+1 .. cp+r (add the
// step's constant, multiply the mismatching donors by K1, divide where the step rescaled) and is stored as
// float(alpha * b).  !DIV: exactly R steps, no divisions, straight-line; DIV: r <= R steps, each tested.
// The masks of the (chunk, step) pairs come in one behind the other (scalar loads, one pair ahead).
template <int S, int R, bool DIV, int CKN, bool NS>
RL_DEV void product_steps(const HeldRow<S, NS> &ck, int lane, const double (&b)[S], float *trow_lane,
                          const MaskRow (&rows)[CKN], const double (&cfs)[CKN], const double (&dvs)[CKN], int r,
                          double K1) {
  constexpr int NC = S / 8;
  u64x8 m = load_masks<8>(rows[1], 0);
#pragma unroll
  for (int c = 0; c < NC; c++) {
    double v[8];
#pragma unroll
    for (int jj = 0; jj < 8; jj++) v[jj] = ck.get(c * 8 + jj, lane);
#pragma unroll
    for (int q = 1; q <= R; q++) {
      if (DIV && q > r) break;  // (wave-uniform)
      const u64x8 mc = m;
      const bool last_step = DIV ? (q == r) : (q == R);
      if (!(last_step && c + 1 == NC)) {
        MaskRow nr = last_step ? rows[1] : rows[q < R ? q + 1 : 1];
        const int nc = last_step ? c + 1 : c;
        asm volatile("" : "+s"(nr) : "s"(mc[0]));
        m = load_masks<8>(nr, nc);
      }
#pragma unroll
      for (int jj = 0; jj < 8; jj++) v[jj] = v[jj] + cfs[q];
      masked_mul8<0>(v, mc, K1);
      if (DIV && dvs[q] != 0.0) {
#pragma unroll
        for (int jj = 0; jj < 8; jj++) v[jj] /= dvs[q];
      }
    }
    GlobalF32 qo = (GlobalF32)(trow_lane + c * 8 * 64);
    asm volatile("" : "+v"(qo));
#pragma unroll
    for (int jj = 0; jj < 8; jj++) qo[jj * 64] = (float)(v[jj] * b[c * 8 + jj]);
  }
}

// (The beta-only descent of a part launch as a kernel of its own, two waves a SIMD, was built and measured in round 5:
//  waits halved, same wall-clock -- removed, DESIGN_NOTES.md 11.)
template <int S, int TAIL, int MODE, int WAVES, bool NOSTRIP = false>
RL_DEV void repaint_backward(const RepaintParams &p, int n, float *stage, double *strip, WaveLink<WAVES> &lk) {
  const int wv = lk.w;
  PaintLane<S> pl;
  pl.init(p.lay, n, wv);
  const PaintConsts &c = p.c;
  const int t = n - p.k0;
  const int ib = p.ib[t], ie = p.ie[t];
  const int D = ie - ib + 1;
  const int64_t off = p.plan_off[n] + ib;
  const int32_t *__restrict__ st = p.sites + off;
  const double *__restrict__ cfp = p.cf + off;
  const double *__restrict__ nx = p.nxt + off;
  const double cf_last = p.cf_last[t], nxt_last = p.nxt_last[t];
  constexpr int CK = REPAINT_CHECKPOINT;
  constexpr int ROW = S * 64 * WAVES;
  constexpr int TROW = S * 64 * WAVES;  // a posterior row, same order
  const double *__restrict__ side = p.side + p.top_off[t] * REPAINT_SIDE;
  const double *__restrict__ ckrows = p.scratch + p.ck_off[t] * (int64_t)ROW + (size_t)wv * (S * 64) + pl.lane;
  const double K1 = in_vgpr(c.K1);
  const int row_lo = p.row_lo[t], row_hi = p.row_hi[t];  // posterior rows that are kept
  float *__restrict__ top = p.topology + p.slab_off[t] * (int64_t)TROW + (size_t)wv * (S * 64) + pl.lane;
  float *__restrict__ lsout = p.logscales + p.top_off[t];
  const double theta = in_vgpr(c.theta), ntheta = in_vgpr(c.ntheta);

  // The rows go by blocks of CK: the block's checkpoint row alpha_cp (cp = j - j % CK) is read ONCE and held on
  // chip while the pass walks down the block; row j is rebuilt from it chunk by chunk with j - cp forward steps
  // (the same operations on the same operands: same bits) and topology row j = float(alpha_j * beta_j).  The
  // alpha slots of the target itself and past a lane's run come out as finite garbage there; beta is +0.0 in
  // both.  The next block's checkpoint row is requested as soon as the block's last product (row cp) has let
  // go of the strip, so those loads land during the following beta step.
  int cpb = (D - 1) - (D - 1) % CK;  // checkpoint row of the current block
  auto block_kept = [&](int c0) { return c0 >= 0 && c0 < row_hi && c0 + CK > row_lo; };
  // a block's side records and sites in one vector load each: lane l holds double l of side[cp ...], site cp + l
  auto block_records = [&](int c0, double &sd, int &sw) {
    const int l = pl.lane;
    const bool in = c0 >= 0;
    sd = (in && l < CK * REPAINT_SIDE && (c0 + l / REPAINT_SIDE) < D) ? side[(size_t)c0 * REPAINT_SIDE + l] : 0.0;
    sw = (in && l < CK && c0 + l < D) ? st[c0 + l] : 0;
  };
  double rec, rec_next;
  int sit, sit_next;
  block_records(cpb, rec, sit);
  block_records(cpb - CK, rec_next, sit_next);
  double b[S];
  // (a later launch of a bounded window may start from the state an earlier one left at row jstart: nothing above it
  //  is asked for -- the rows kept lie below -- and the pass goes on exactly as it would have)
  int jstart = (p.bstate && p.partial) ? __builtin_amdgcn_readfirstlane(p.start_row[t]) : -1;
  const int jsave = p.bstate ? __builtin_amdgcn_readfirstlane(p.save_row[t]) : -1;
  double *__restrict__ bst = p.bstate ? p.bstate + ((size_t)t * WAVES + wv) * (S * 64) + pl.lane : nullptr;
  const double *__restrict__ state_from = bst;          // where a start from a kept state loads from ...
  const double *__restrict__ scal_from = p.bscal ? p.bscal + (size_t)t * 2 : nullptr;  // ... and its two scalars
  float lsf = 0.0f;
  if (jstart < 0) {
    lsf = (float)lane_value(rec, (D - 1 - cpb) * REPAINT_SIDE + 2);  // the forward pass's last logscale
    lsf = lsf + p.ls_beta[t];  // float += float (:895)
    const ColdRepaint cp = cold_params<RepaintParams>();
    load_stone<S>(pl, cp->beta_end + (size_t)t * cp->lay.N, b, stage);
    set_slot<S>(b, pl.jk, pl.kbit, 0.0);  // the target's own slot of beta is +0.0 (fast_painting.cpp: b[k] = 0)
  } else {
    load_row<S>(state_from, b);
    cpb = (jstart + 1) - (jstart + 1) % CK;  // the block of the row above: the loop steps down from there
    block_records(cpb, rec, sit);
    block_records(cpb - CK, rec_next, sit_next);
  }
  HeldRow<S, NOSTRIP> ck;  // (the strip shares its LDS with `stage`, which is done with by now)
  ck.lds = strip;
#pragma unroll
  for (int i = 0; i < (HeldRow<S, NOSTRIP>::VREG > 0 ? HeldRow<S, NOSTRIP>::VREG : 1); i++) ck.v[i] = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  {  // the checkpoint row of the block the first row to be done belongs to
    const int first_row = jstart < 0 ? D - 1 : jstart;
    const int blk = first_row - first_row % CK;
    if (block_kept(blk)) ck.request(ckrows + (int64_t)(blk / CK) * ROW - pl.lane, pl.lane);
  }
  MaskRow rows[CK];
  double cfs[CK], dvs[CK];
  auto open_block = [&]() {
#pragma unroll
    for (int q = 1; q < CK; q++) {
      rows[q] = site_row(p.masks, S, p.L, __builtin_amdgcn_readlane(sit, q), WAVES, wv);
      cfs[q] = lane_value(rec, q * REPAINT_SIDE + 0);
      dvs[q] = lane_value(rec, q * REPAINT_SIDE + 1);
    }
  };
  open_block();
  auto product_row = [&](int j, bool first) {
    const int r = j - cpb;
    const bool kept = j >= row_lo && j < row_hi;  // (uniform over the workgroup)
    float *__restrict__ trow = top + (int64_t)(j - row_lo) * TROW;
    if (r == 0) {
      if (kept) product_last(ck, pl.lane, b, trow);
      if (block_kept(cpb - CK)) ck.request(ckrows + (int64_t)(cpb / CK - 1) * ROW - pl.lane, pl.lane);
      return;
    }
    if (!kept) return;
    bool div = first;  // (the window's last row goes the tested way: one copy of the straight-line variants)
#pragma unroll
    for (int q = 1; q < CK; q++) div = div || (q <= r && dvs[q] != 0.0);
    if (div) {
      product_steps<S, CK - 1, true>(ck, pl.lane, b, trow, rows, cfs, dvs, r, K1);
      return;
    }
    static_assert(CK <= 8, "one straight-line variant per distance from the checkpoint");
    switch (r) {
      case 1: product_steps<S, 1, false>(ck, pl.lane, b, trow, rows, cfs, dvs, r, K1); break;
      case 2: if constexpr (CK > 2) product_steps<S, 2, false>(ck, pl.lane, b, trow, rows, cfs, dvs, r, K1); break;
      case 3: if constexpr (CK > 3) product_steps<S, 3, false>(ck, pl.lane, b, trow, rows, cfs, dvs, r, K1); break;
      case 4: if constexpr (CK > 4) product_steps<S, 4, false>(ck, pl.lane, b, trow, rows, cfs, dvs, r, K1); break;
      case 5: if constexpr (CK > 5) product_steps<S, 5, false>(ck, pl.lane, b, trow, rows, cfs, dvs, r, K1); break;
      case 6: if constexpr (CK > 6) product_steps<S, 6, false>(ck, pl.lane, b, trow, rows, cfs, dvs, r, K1); break;
      default: if constexpr (CK > 7) product_steps<S, 7, false>(ck, pl.lane, b, trow, rows, cfs, dvs, r, K1); break;
    }
  };
  // (a later launch of a bounded window: the logscales are in place, the forward pass stopped below row_hi -- its
  //  records above are not there -- and nothing below row_lo is asked for)
  const int jstop = p.partial ? max(0, row_lo) : 0, jls = p.partial ? row_hi : D;
  const int jtop = jstart < 0 ? D - 2 : jstart;  // the first row the loop does
  // at the top of row j: s0 = site j + 1, s1 = site j, s2 = site j - 1
  int s0 = st[jtop + 1], s1 = jtop >= 0 ? st[jtop] : 0, s2 = jtop >= 1 ? st[jtop - 1] : 0;
  MaskRow rown = site_row(p.masks, S, p.L, s0, WAVES, wv);
  double bsum, cfac, prev_ls;
  if (jstart < 0) {
    {  // the window's last row: beta is the stone (:930)
      const MaskTerm<S> term{rown, b, theta, ntheta};
      bsum = wave_sum<MODE, S, WAVES>(term, local_sum<S>(term), lk);
    }
    product_row(D - 1, true);
    if (pl.lane == 0 && wv == 0 && D - 1 < jls) lsout[D - 1] = lsf;
    cfac = cf_last * bsum;
    prev_ls = (double)p.ls_beta[t];  // :951
  } else {
    cfac = scal_from[0];
    prev_ls = scal_from[1];
  }
  MaskRow rowh = site_row(p.masks, S, p.L, s1, WAVES, wv);
  u64x4 firstn = load_masks<4>(rown, 0), firsth = load_masks<4>(rowh, 0);
  uint32_t touched = 0;
  for (int j = jtop; j >= jstop; j--) {
    if (j == jsave) {  // the state before row j, for the launches to come (window.cpp: place_rows)
      store_row<S>(bst, b);
      if (pl.lane == 0 && wv == 0) {
        p.bscal[(size_t)t * 2] = cfac;
        p.bscal[(size_t)t * 2 + 1] = prev_ls;
      }
    }
    if (j < cpb) {  // the pass enters the block below: its records were requested a block ago
      cpb -= CK;
      rec = rec_next;
      sit = sit_next;
      block_records(cpb - CK, rec_next, sit_next);
      open_block();
    }
    retire_touch(touched);
    if (j > 0) touched = touch_row(p.masks, S, s2, pl.lane, WAVES, wv);
    s0 = s1;
    s1 = s2;
    if (j > 1) s2 = st[j - 2];
    const double nx_j = (j + 1 == D - 1 ? nxt_last : nx[j + 1]), cf_j = cfp[j];
    const double b1 = div_by_const(cfac, ntheta, c.inv_ntheta);     // cfac / ntheta
    const double bt = div_by_const(cfac, theta, c.inv_theta) - b1;  // cfac / theta - b1
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
    const double als = lane_value(rec, (j - cpb) * REPAINT_SIDE + 2);  // the forward logscale of this site
    prev_ls += nx_j;
    lsf = (float)(als + prev_ls);  // :962-963
    cfac = bsum;
    product_row(j, false);  // topology = float(alpha * beta) before the rescale of this step (:1039 vs :1047)
    if (cfac < c.lower || cfac > c.upper) {  // :1047-1061
#pragma unroll
      for (int i = 0; i < S; i++) b[i] /= bsum;
      const double lg = log(bsum);
      prev_ls += lg;
      lsf = (float)((double)lsf + lg);
      cfac = 1.0;
    }
    cfac *= cf_j;
    if (pl.lane == 0 && wv == 0 && j < jls) lsout[j] = lsf;
  }
  retire_touch(touched);
}

template <int S, int TAIL, int MODE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) repaint_fwd_kernel(const RepaintParams p) {
  __shared__ float stage[WAVES][16 * 64];
  __shared__ WaveLinkStorage link;
  WaveLink<WAVES> lk;
  lk.s = &link;
  lk.w = WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  repaint_forward<S, TAIL, MODE, WAVES>(p, p.order[blockIdx.x], stage[lk.w], lk);
}
// One wave per SIMD (the LDS strips decide that): beta in registers, the block's checkpoint row in LDS.  The strips
// are dynamic LDS so that the compiler budgets registers for two waves per SIMD (256 VGPRs, no AGPR copies).
template <int S>
constexpr int strip_doubles() {
  return HeldRow<S>::LREG * 64 > 16 * 64 / 2 ? HeldRow<S>::LREG * 64 : 16 * 64 / 2;
}
template <int S, int TAIL, int MODE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) repaint_bwd_kernel(const RepaintParams p) {
  extern __shared__ double strips[];  // [WAVES][strip]: stage (16 * 64 floats) first, then the held checkpoint row
  __shared__ WaveLinkStorage link;
  WaveLink<WAVES> lk;
  lk.s = &link;
  lk.w = WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  double *strip = strips + lk.w * strip_doubles<S>();
  repaint_backward<S, TAIL, MODE, WAVES>(p, p.order[blockIdx.x], (float *)strip, strip, lk);
}

// ... and without the strip (HeldRow NOSTRIP): the part launches of a bounded window, two waves a SIMD
template <int S, int TAIL, int MODE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) repaint_bwd_nostrip_kernel(const RepaintParams p) {
  __shared__ float stage[WAVES][16 * 64];
  __shared__ WaveLinkStorage link;
  WaveLink<WAVES> lk;
  lk.s = &link;
  lk.w = WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  repaint_backward<S, TAIL, MODE, WAVES, true>(p, p.order[blockIdx.x], stage[lk.w], nullptr, lk);
}
template <int S, int TAIL, int WAVES>
static hipError_t launch_repaint_t(const RepaintParams &p, hipStream_t stream) {
  constexpr size_t strips = (size_t)WAVES * strip_doubles<S>() * sizeof(double);
  // (two waves of S = 80: 72 KB of dynamic LDS, more than a launch may ask for without saying so)
  // (the attribute belongs to the device: once per device and instantiation)
  if (strips > 48 * 1024) {
    static std::atomic<unsigned long long> done[2];  // devices 0..127
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev > 127 || !(done[(dev >> 6) & 1].load(std::memory_order_acquire) & bit)) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&repaint_bwd_kernel<S, TAIL, RL_MODE, WAVES>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)strips);
      if (e != hipSuccess) return e;
      if (dev <= 127) done[(dev >> 6) & 1].fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL((repaint_fwd_kernel<S, TAIL, RL_MODE, WAVES>), dim3(p.nloc), dim3(64 * WAVES), 0, stream, p);
  // the part launches of a bounded window without the strip, whole windows with it
  if (p.partial && p.nostrip)
    hipLaunchKernelGGL((repaint_bwd_nostrip_kernel<S, TAIL, RL_MODE, WAVES>), dim3(p.nloc), dim3(64 * WAVES), 0, stream, p);
  else
    hipLaunchKernelGGL((repaint_bwd_kernel<S, TAIL, RL_MODE, WAVES>), dim3(p.nloc), dim3(64 * WAVES), strips, stream, p);
  return hipGetLastError();
}

template <>
hipError_t launch_repaint_mode<RL_MODE>(const RepaintParams &p, int S, int waves, hipStream_t stream) {
  if (waves == 1) {
    switch (S) {
#define RL_CASE(s, t) \
  case s:             \
    return launch_repaint_t<s, t, 1>(p, stream);
      RL_FOR_EACH_S(RL_CASE)
#undef RL_CASE
    }
  } else if (waves == 2) {
#ifndef RL_ONLY_S
    switch (S) {
#define RL_CASE(s, t) \
  case s:             \
    return launch_repaint_t<s, t, 2>(p, stream);
      RL_FOR_EACH_S_2WAVES(RL_CASE)
#undef RL_CASE
    }
#endif
  }
  return hipErrorInvalidValue;
}

}  // namespace rl
