// exact_sum.h -- the reference's SERIAL double-precision sum, reproduced bit
// for bit by a wavefront in parallel.
//
// Problem.  The reference adds the N donor terms left to right
// (fast_painting.cpp:300-303, 495-503); the result is not the correctly
// rounded sum but carries that particular order's roundings, and the tree
// builder downstream breaks exact float ties, so the order must be kept
// (SURVEY.md 7 H1).  A literal serial sum costs one dependent add per donor
// (sum_exact: 64*S wave instructions per step).
//
// Idea.  Lane l owns a contiguous run of terms.  If lane l knew its exact
// entry value s_l it could add its run serially on its own, and all lanes
// could do so at once.  It does not, but:
//   (1) an ordinary parallel scan gives P_l with |s_l - P_l| <= 2600 ulp
//       (every rounding of either order is <= 1/2 ulp of a partial sum,
//       <= 10240 + 86 additions);
//   (2) within one binade, fl(a + x) - a depends on `a` only through the
//       parity of a/ulp (round-half-even), and across ONE binade crossing only
//       through a mod 4 ulp.  So lane l runs its serial sum from four starts
//       r_0..r_3 with r_h = h (mod 4 ulp), and the true run is the run of the
//       matching r_h translated by the constant s_l - r_h (a multiple of 4 ulp)
//       -- provided the true run and that r_h-run change binade at the same
//       terms;
//   (3) r_0 and r_3 are placed 16384 ulp below / above P_l, so s_l lies between
//       them; fl-addition is monotone in its start value, hence if the r_0 and
//       r_3 runs show the same exponent after every term (their high words are
//       xor-ed and or-accumulated term by term) so does every run in between,
//       the true one included.
// What remains sequential is a walk over the few lanes whose run is not a pure
// translation (a tie or a binade crossing happened inside): integer offsets
// delta_l = (s_l - P_l)/ulp are pushed through 4-entry tables.  Lanes whose
// run jumps two or more binades (a donor much larger than the prefix) redo
// their run from the exact entry value; anything irregular (P_l within 16384
// ulp of a power of two, ...) falls back to sum_exact for that step.
// The result is ALWAYS the serial sum, bit for bit.
#pragma once
#include "paint_device.h"

namespace rl {

// ---- DPP helpers (gfx9 row/bcast controls) -------------------------------
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
RL_DEV double dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, BANK_MASK, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, BANK_MASK, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
RL_DEV int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, true);
}
enum { DPP_ROW_SHR = 0x110, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143 };

// inclusive scans over the 64 lanes (zeros shifted in)
RL_DEV double wave_scan_f64(double v) {
  v += dpp_f64<DPP_ROW_SHR + 1>(v);
  v += dpp_f64<DPP_ROW_SHR + 2>(v);
  v += dpp_f64<DPP_ROW_SHR + 4>(v);
  v += dpp_f64<DPP_ROW_SHR + 8>(v);
  v += dpp_f64<DPP_ROW_BCAST15, 0xa>(v);
  v += dpp_f64<DPP_ROW_BCAST31, 0xc>(v);
  return v;
}
RL_DEV int wave_scan_i32(int v) {
  v += dpp_i32<DPP_ROW_SHR + 1>(v);
  v += dpp_i32<DPP_ROW_SHR + 2>(v);
  v += dpp_i32<DPP_ROW_SHR + 4>(v);
  v += dpp_i32<DPP_ROW_SHR + 8>(v);
  v += dpp_i32<DPP_ROW_BCAST15, 0xa>(v);
  v += dpp_i32<DPP_ROW_BCAST31, 0xc>(v);
  return v;
}

RL_DEV int hi32(double v) { return (int)(__double_as_longlong(v) >> 32); }
RL_DEV int expo_field(double v) { return (hi32(v) >> 20) & 0x7ff; }  // v >= 0
RL_DEV double rd_lane_f64(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// 2^(k) for a biased exponent field f = k + 1023
RL_DEV double pow2_field(int f) { return __longlong_as_double((long long)f << 52); }

// Term generators.  get(i, th, nth) returns the i-th term of the lane; the two
// weights are passed in so that each pass of sum_exact_fast can use its own
// opaque copies: otherwise the compiler shares the S selected weights (or the
// S products) between the passes and keeps them alive in registers.
template <int S>
struct RegTerm {  // forward: the terms are the alpha registers themselves
  static constexpr bool REG = true;
  const double (&a)[S];
  double th = 0.0, nth = 0.0;
  unsigned long long *stats = nullptr;  // experiment builds (-DRL_STATS) only
  RL_DEV double get(int i, double, double) const { return a[i]; }
};
template <int S>
struct WeightedTerm {  // backward: e(i) * beta[i], e = theta on a mismatch else 1 - theta
  static constexpr bool REG = false;
  const LaneBits<S> &m;
  const double (&b)[S];
  double th, nth;
  unsigned long long *stats = nullptr;
  RL_DEV double get(int i, double t, double n) const { return (m.get(i) ? t : n) * b[i]; }
};

// The literal serial sum as the rare-path fallback of sum_exact_fast: same
// result as sum_exact, but the terms are recomputed in every round (the
// opaque multiply by 1.0 stops the compiler from hoisting S doubles of terms
// out of the round loop, which would cost the hot path its registers).
template <int S, typename T>
RL_DEV double sum_exact_fallback(const T &term) {
  double s = 0.0;
  for (int l = 0; l < 64; l++) {
    double th = term.th, nth = term.nth;
    double tmp = s;
#pragma unroll
    for (int i = 0; i < S; i++) {
      if constexpr (!T::REG) asm volatile("" : "+v"(th), "+v"(nth), "+v"(tmp));
      tmp += term.get(i, th, nth);
    }
    s = wave_bcast(tmp, l);
  }
  return s;
}

#ifdef RL_STATS
#define RL_STAT(i, v) do { if (term.stats && (threadIdx.x & 63) == 0) atomicAdd(&term.stats[i], (unsigned long long)(v)); } while (0)
#else
#define RL_STAT(i, v) do { } while (0)
#endif

template <int S, typename T>
RL_DEV double sum_exact_fast(const T &term) {
  constexpr bool REG_TERM = T::REG;
  const int lane = threadIdx.x & 63;
  constexpr int G4 = 16384;  // half-width of the bracket [r_0, r_3] in ulps

  // ---- A. local serial sums and the approximate prefix
  double L = 0.0;
  {
    double th = term.th, nth = term.nth;
#pragma unroll
    for (int i = 0; i < S; i++) {
      // the fake dependency makes term i+1 wait for sum i: otherwise all S
      // terms are computed up front and stay alive
      if constexpr (!REG_TERM) asm volatile("" : "+v"(th), "+v"(nth), "+v"(L));
      L += term.get(i, th, nth);
    }
  }
  const double Q = wave_scan_f64(L);          // ~ sum over lanes <= l
  double P = dpp_f64<DPP_WAVE_SHR1>(Q);       // ~ entry value of this lane (lane 0: +0.0)
  const long long pb = __double_as_longlong(P);
  const bool zero_entry = pb == 0;            // nothing but zeros before this lane: entry exactly 0

  // ---- B. four runs from r_h = h (mod 4 ulp), r_0 / r_3 bracketing the true entry
  const long long base = pb & ~3ll;
  const int p0 = (int)(pb & 3);
  double c0 = __longlong_as_double(base - G4);
  double c1 = __longlong_as_double(base + 1);
  double c2 = __longlong_as_double(base + 2);
  double c3 = __longlong_as_double(base + 3 + G4);
  if (zero_entry) { c0 = 0.0; c1 = 0.0; c2 = 0.0; c3 = 0.0; }
  const double r0 = c0, r1 = c1, r2 = c2, r3 = c3;
  const int e_in = expo_field(r0);
  const bool entry_ok = zero_entry || (e_in == expo_field(r3) && e_in > 64);
  int exdiff = 0;  // OR over the terms of (high word of c0) xor (high word of c3)
  double thB = term.th, nthB = term.nth;
#pragma unroll
  for (int i = 0; i < S; i++) {
    const double x = term.get(i, thB, nthB);
    c0 += x;
    c1 += x;
    c2 += x;
    c3 += x;
    // exponent fields of the two bracketing runs must agree after every term;
    // one 3-input bit op per term (bits 20..30 of the accumulated xor)
    exdiff |= hi32(c0) ^ hi32(c3);
    // tie the check to its partial sums: otherwise the scheduler first runs
    // the chains to the end and keeps all partial sums alive
    asm volatile("" : "+v"(exdiff), "+v"(c0), "+v"(c3), "+v"(thB), "+v"(nthB));
  }
  const int e_out = expo_field(c0);
  // exit unit = entry unit of the next lane = ulp of the binade of Q
  const int e_next = expo_field(Q);
  const double inv_u_out = pow2_field(1075 + 1023 - e_next);  // 1 / 2^(e_next - 1075)

  // ---- classification
  const double D0 = c0 - r0, D1 = c1 - r1, D2 = c2 - r2, D3 = c3 - r3;
  const bool same_seq = (exdiff >> 20) == 0;
  const int sh = e_out - e_in;
  const bool pure_cand = (D0 == D1) && (D1 == D2) && (D2 == D3) && sh == 0;
  bool invalid = !entry_ok || !same_seq;
  // the exit must sit in the binade the next lane (or the caller) measures in
  if (!zero_entry && e_out != e_next) invalid = true;
  if (zero_entry && expo_field(c0) != e_next) invalid = true;
  const bool special = !zero_entry && !invalid && sh >= 2;
  const bool pure = zero_entry || (!invalid && !special && pure_cand);
  const bool table = !invalid && !special && !pure;

  // offsets are in units of the exit ulp; for a pure lane delta_out = delta_in + cinc
  int cinc = 0;
  if (pure && !zero_entry) cinc = (int)((D0 - (Q - P)) * inv_u_out);
  // exit offsets of the four runs, |A_h| < 2^15, packed two per word
  int w01 = 0, w23 = 0;
  if (table) {
    const int A0 = (int)((c0 - Q) * inv_u_out), A1 = (int)((c1 - Q) * inv_u_out);
    const int A2 = (int)((c2 - Q) * inv_u_out), A3 = (int)((c3 - Q) * inv_u_out);
    w01 = (A0 & 0xffff) | (A1 << 16);
    w23 = (A2 & 0xffff) | (A3 << 16);
  }
  const int meta = p0 | (sh << 2) | (special ? 16 : 0);

  // the caller's unit: the total is returned as Q_63 + delta * ulp(Q_63); Q_63
  // must not be within the bracket width of a power of two
  const double Qt = rd_lane_f64(Q, 63);
  {
    const long long qb = __double_as_longlong(Qt) & ~3ll;
    const int ea = expo_field(__longlong_as_double(qb - G4)), eb = expo_field(__longlong_as_double(qb + 3 + G4));
    if (ea != eb || ea <= 64) invalid = true;
  }
  RL_STAT(0, 1);
  if (__ballot(invalid) != 0ull) {
    RL_STAT(1, 1);
#ifdef RL_X_NOFALLBACK
    return 0.0;
#else
    return sum_exact_fallback<S>(term);  // irregular step: literal serial sum
#endif
  }

  // ---- C. walk the non-pure lanes
  const int cpre = wave_scan_i32(cinc);
  unsigned long long todo = __ballot(!pure);
  RL_STAT(2, __builtin_popcountll(todo));
  RL_STAT(3, __builtin_popcountll(__ballot(special)));
  int delta = 0;  // offset at the exit of lane `prev`
  // lane 0 enters at exactly 0: its local sum is exact and Q_0 == L_0
  int cpre_prev = __builtin_amdgcn_readlane(cpre, 0);
  while (todo) {
    const int q = __builtin_ctzll(todo);
    todo &= todo - 1;
    // pure lanes strictly between prev and q (cinc of a non-pure lane is 0)
    const int cq = __builtin_amdgcn_readlane(cpre, q);
    delta += cq - cpre_prev;
    cpre_prev = cq;
    const int m = __builtin_amdgcn_readlane(meta, q);
#ifdef RL_X_NOSPECIAL
    if (false) {
#else
    if (m & 16) {
#endif
      // multi-binade run: redo it from the exact entry value
      const double Pq = rd_lane_f64(P, q);
      const double uq = pow2_field(expo_field(Pq) - 52);
      double t = Pq + (double)delta * uq;  // exact
      double thC = term.th, nthC = term.nth;
#pragma unroll
      for (int i = 0; i < S; i++) {
        if constexpr (!REG_TERM) asm volatile("" : "+v"(thC), "+v"(nthC), "+v"(t));
        t += term.get(i, thC, nthC);
      }
      const double v = rd_lane_f64(t, q);
      delta = (int)((v - rd_lane_f64(Q, q)) * rd_lane_f64(inv_u_out, q));
    } else {
      const int qp0 = m & 3, qsh = (m >> 2) & 3;
      const int h = (qp0 + delta) & 3;
      const int w = __builtin_amdgcn_readlane((h & 2) ? w23 : w01, q);
      const int A = (int)(short)((h & 1) ? (w >> 16) : w);
      const int B = h == 0 ? (-qp0 - G4) : (h == 3 ? (3 - qp0 + G4) : (h - qp0));
      delta = A + ((delta - B) >> qsh);
    }
  }
  delta += __builtin_amdgcn_readlane(cpre, 63) - cpre_prev;
  return Qt + (double)delta * pow2_field(expo_field(Qt) - 52);
}

template <int MODE, int S, typename T>
RL_DEV double wave_sum(const T &term) {
  if constexpr (MODE == 1)
    return sum_exact_fast<S>(term);
  else if constexpr (MODE == 2)
    return sum_exact<S>([&](int i) { return term.get(i, term.th, term.nth); });
  else
    return sum_lanes<S>([&](int i) { return term.get(i, term.th, term.nth); });
}

}  // namespace rl
