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
//   (1) an ordinary parallel scan gives P_l with |s_l - P_l| <= 10250 ulp
//       (each of the <= 10240 additions of either order errs by <= 1/2 ulp
//       of a partial sum; typically a few tens of ulp);
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
// In integer offsets delta_l = (s_l - P_l)/ulp a lane without a rounding tie
// acts as delta -> ((delta + K) >> s) + C (s = binade crossings inside, 0 or 1);
// such maps compose into maps of the same form, so a segmented wavefront scan
// composes them all at once.  What remains sequential is a walk over the few
// lanes (measured ~3.5 per sum at N = 5000) where a tie made the four runs
// disagree (4-entry table) or where the run jumps two or more binades (a term
// much larger than the prefix, ~0.9 per sum: that lane redoes its run from its
// exact entry value).  Anything else irregular (P_l within 16384 ulp of a power
// of two, ...) falls back to the literal serial sum for that step (measured:
// < 1e-4 of the steps).
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

// inclusive scan over the 64 lanes (zeros shifted in)
RL_DEV double wave_scan_f64(double v) {
  v += dpp_f64<DPP_ROW_SHR + 1>(v);
  v += dpp_f64<DPP_ROW_SHR + 2>(v);
  v += dpp_f64<DPP_ROW_SHR + 4>(v);
  v += dpp_f64<DPP_ROW_SHR + 8>(v);
  v += dpp_f64<DPP_ROW_BCAST15, 0xa>(v);
  v += dpp_f64<DPP_ROW_BCAST31, 0xc>(v);
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

// Term generators.  for_each(th, nth, f) calls f(i, term_i) for the lane's S
// terms in order; the two weights are passed in so that each pass of
// sum_exact_fast can use its own opaque copies: otherwise the compiler shares
// the S selected weights (or the S products) between the passes and keeps them
// alive in registers.
template <int S>
struct RegTerm {  // forward: the terms are the alpha registers themselves
  static constexpr bool REG = true;
  const double (&a)[S];
  double th = 0.0, nth = 0.0;
  unsigned long long *stats = nullptr;  // experiment builds (-DRL_STATS) only
  RL_DEV double get(int i, double, double) const { return a[i]; }
  template <typename F>
  RL_DEV void for_each(double &, double &, F &&f) const {
#pragma unroll
    for (int i = 0; i < S; i++) f(i, a[i]);
  }
};
template <int S>
struct MaskTerm {  // backward, lane-mask panel: e(i) * beta[i] with the site's row words as EXEC masks
  static constexpr bool REG = false;
  MaskRow row;  // mismatch row of the site (S words)
  const double (&b)[S];
  double th, nth;
  unsigned long long *stats = nullptr;
  template <typename F>
  RL_DEV void for_each(double &t, double &n, F &&f) const {
    for_each_chunk<S, 4>(row, [&](int j0, const u64x4 &m) {
      double x[4];
      weighted4(x, b[j0], b[j0 + 1], b[j0 + 2], b[j0 + 3], m, t, n);
#pragma unroll
      for (int jj = 0; jj < 4; jj++) f(j0 + jj, x[jj]);
    });
  }
};

// ---- targets spread over several waves ------------------------------------
// For N > 5120 the stepping-stone kernel gives a target to a workgroup of
// WAVES = 2 waves (virtual lanes 0..127, wave w holds lanes 64w..64w+63) so
// that each lane keeps S <= 80 registers.  A sum then runs over the waves in
// order; they meet in LDS (WaveLink) at workgroup barriers.  With WAVES = 1
// every exchange below compiles away.
struct WaveLinkStorage {
  double tot[2][2];  // [phase][wave]: approximate total of the wave's lane sums
  double result;     // the finished sum / the hand-over value of the serial fallback
  int bad[2];        // wave w needs the fallback
  int delta;         // exact offset at the exit of wave 0 (units: ulp of its exit binade)
};
template <int WAVES>
struct WaveLink {
  WaveLinkStorage *s = nullptr;
  int w = 0;          // this wave
  unsigned phase = 0;  // alternates per sum: a wave may run one sum ahead of the other
  RL_DEV void barrier() const {
    if constexpr (WAVES > 1) __syncthreads();
  }
};

// The literal serial sum as the rare-path fallback of sum_exact_fast: same
// result as sum_exact, but the terms are recomputed in every round (the empty
// asm stops the compiler from hoisting S doubles of terms out of the round
// loop, which would cost the hot path its registers).
template <int S, typename T>
RL_DEV double sum_exact_fallback(const T &term, double s = 0.0) {
  for (int l = 0; l < 64; l++) {
    double th = term.th, nth = term.nth;
    double tmp = s;
    term.for_each(th, nth, [&](int, double x) {
      tmp += x;
      if constexpr (!T::REG) asm volatile("" : "+v"(th), "+v"(nth), "+v"(tmp));
    });
    s = wave_bcast(tmp, l);
  }
  return s;
}
// the same over the waves of a workgroup, in order; every wave returns the total
template <int S, int WAVES, typename T>
RL_DEV double sum_exact_fallback_linked(const T &term, const WaveLink<WAVES> &lk) {
  if constexpr (WAVES == 1) {
    return sum_exact_fallback<S>(term);
  } else {
    double r = 0.0;
    if (lk.w == 0) {
      r = sum_exact_fallback<S>(term);
      if ((threadIdx.x & 63) == 0) lk.s->result = r;
    }
    lk.barrier();
    if (lk.w == 1) {
      r = sum_exact_fallback<S>(term, lk.s->result);
    }
    lk.barrier();  // wave 0 has handed over; wave 1 may now overwrite
    if (lk.w == 1 && (threadIdx.x & 63) == 0) lk.s->result = r;
    lk.barrier();
    r = lk.s->result;
    lk.barrier();  // everyone has read it before the next sum writes
    return r;
  }
}

#ifdef RL_STATS
#define RL_CLK() __builtin_readcyclecounter()
#define RL_STAT(i, v) do { if (term.stats && (threadIdx.x & 63) == 0) atomicAdd(&term.stats[i], (unsigned long long)(v)); } while (0)
#else
#define RL_CLK() 0ull
#define RL_STAT(i, v) do { } while (0)
#endif

// L: this lane's serial sum of its terms from +0.0 (the caller accumulates it
// inside its update loop, where the dependent adds hide behind other work).
template <int S, int WAVES, typename T>
RL_DEV double sum_exact_fast(const T &term, double L, WaveLink<WAVES> &lk) {
  constexpr bool REG_TERM = T::REG;
  constexpr int G4 = 16384;  // half-width of the bracket [r_0, r_3] in ulps

  const unsigned long long tk0 = RL_CLK();
  // ---- A. approximate prefix from the lanes' local serial sums
  double Q = wave_scan_f64(L);                // ~ sum over lanes <= l (of this wave)
  double P = dpp_f64<DPP_WAVE_SHR1>(Q);       // ~ entry value of this lane (lane 0: +0.0)
  if constexpr (WAVES > 1) {
    // wave 1 continues where wave 0 ends: shift its prefixes by wave 0's (approximate) total
    const int lane = threadIdx.x & 63;
    const unsigned ph = lk.phase & 1u;
    if (lane == 63) lk.s->tot[ph][lk.w] = Q;
    lk.barrier();
    if (lk.w == 1) {
      const double before = lk.s->tot[ph][0];
      Q = before + Q;
      P = dpp_f64<DPP_WAVE_SHR1>(Q);
      if (lane == 0) P = before;
    }
  }
  const long long pb = __double_as_longlong(P);
  // Lanes whose entry value is known exactly need no bracket: nothing but zeros before the lane (entry
  // +0.0; lane 0 always), and lane 1 of the first wave, whose entry is lane 0's local sum -- a serial sum
  // from +0.0, i.e. the true prefix.  Their four runs start AT the entry, so the run is the true one whatever
  // it crosses (lane 1 is where two-binade jumps are most frequent), and the lane's map is the constant
  // exit offset.
  const bool zero_entry = pb == 0 || ((threadIdx.x & 63) == 1 && lk.w == 0);

  const unsigned long long tk1 = RL_CLK();
  // ---- B. four runs from r_h = h (mod 4 ulp), r_0 / r_3 bracketing the true entry
  const long long base = pb & ~3ll;
  const int p0 = (int)(pb & 3);
  double c0 = __longlong_as_double(base - G4);
  double c1 = __longlong_as_double(base + 1);
  double c2 = __longlong_as_double(base + 2);
  double c3 = __longlong_as_double(base + 3 + G4);
  if (zero_entry) { c0 = P; c1 = P; c2 = P; c3 = P; }
  const double r0 = c0, r3 = c3;
  const int e_in = expo_field(r0);
  const bool entry_ok = zero_entry || (e_in == expo_field(r3) && e_in > 64);
  int exdiff = 0;  // OR over the terms of (high word of c0) xor (high word of c3)
  double thB = term.th, nthB = term.nth;
  term.for_each(thB, nthB, [&](int, double x) {
    c0 += x;
    c1 += x;
    c2 += x;
    c3 += x;
    // exponent fields of the two bracketing runs must agree after every term:
    // exdiff |= hi(c0) ^ hi(c3), one v_bitop3_b32 per term (bits 20..30 count)
    exdiff = __builtin_amdgcn_bitop3_b32(exdiff, hi32(c0), hi32(c3), 0xF6);
    // Tie the check to its partial sums: otherwise the scheduler first runs the chains to the end and keeps all
    // partial sums alive.  A scheduling barrier, not an empty asm statement with the sums as operands: the hazard
    // recognizer treats an inline asm that names registers a double-precision instruction has just written as a
    // reader of them and puts a wait state in front -- one s_nop per term, 80 of the ~1000 issue slots of a forward
    // step (round 4, seen in the ISA).
    // (The backward pass, whose terms are recomputed here four at a time, keeps the asm statement: with the barrier
    //  alone its chain block spills 47 register pairs.)
    if constexpr (REG_TERM)
      __builtin_amdgcn_sched_barrier(0);
    else
      asm volatile("" : "+v"(exdiff), "+v"(c1), "+v"(c2), "+v"(thB), "+v"(nthB));  // (c0, c3: through exdiff)
  });
  const unsigned long long tk2 = RL_CLK();
  const int e_out = expo_field(c0);
  // exit unit = entry unit of the next lane = ulp of the binade of Q
  const int e_next = expo_field(Q);
  // An exit offset (c - Q) / ulp is asked for only where c lies in Q's binade (every other lane is invalid or walked
  // from its exact entry): there it is the difference of the two bit patterns -- one integer subtraction of the low
  // words instead of a subtraction, a multiplication and a quarter-rate conversion in double precision, six times
  // per sum (round 4).
  const unsigned q_lo = (unsigned)__double_as_longlong(Q);
  auto exit_offset = [&](double c) -> int { return (int)((unsigned)__double_as_longlong(c) - q_lo); };

  // ---- classification
  const bool same_seq = (exdiff >> 20) == 0;
  const int sh = e_out - e_in;
  bool invalid = !entry_ok || !same_seq;
  // the exit must sit in the binade the next lane (or the caller) measures in
  if (!zero_entry && e_out != e_next) invalid = true;
  if (zero_entry && expo_field(c0) != e_next) invalid = true;
  // a run that jumps two or more binades (a term much larger than the prefix)
  // does not fit the mod-4 argument: the walk redoes it from its exact entry
  const bool jump = !zero_entry && !invalid && sh >= 2;
  // ... unless the two bracketing runs END on the same value: addition is
  // monotone in its start, so every run started inside the bracket ends there
  // too, the true one included -- the exit is known outright and nothing to
  // the left of this lane matters any more (the usual case: the big term
  // swamps the 2^15 ulp of the bracket)
  const bool constant = jump && __double_as_longlong(c0) == __double_as_longlong(c3);
  const bool special = jump && !constant;

  // offsets are in units of the exit ulp.  Exit offsets of the four runs
  // (|A_h| < 2^15) and entry offsets B_h = (r_h - P)/ulp_in of their starts:
  int A0 = 0, A1 = 0, A2 = 0, A3 = 0;
  if (constant) A0 = exit_offset(c0);
  // exact entry: the exit is c0 itself; offset 0 wherever the approximate prefix Q is that same sum (lane 0,
  // all-zero prefixes), a few ulp for lane 1
  if (zero_entry && __double_as_longlong(c0) != __double_as_longlong(Q)) A0 = exit_offset(c0);
  if (!invalid && !jump && !zero_entry) {
    A0 = exit_offset(c0);
    A1 = exit_offset(c1);
    A2 = exit_offset(c2);
    A3 = exit_offset(c3);
  }
  const int B0 = -p0 - G4, B1 = 1 - p0, B2 = 2 - p0, B3 = 3 - p0 + G4;
  // Is the lane's map delta -> A_h + ((delta - B_h) >> sh), h = (p0 + delta) & 3,
  // of the tie-free form ((delta + K) >> sh) + C ?   (K in {0,1} when sh = 1)
  int mK = 0, mC = zero_entry ? A0 : 0;  // exact entry: delta is 0 on entry (nothing but exact lanes before), A0 on exit
  bool affine = zero_entry;
  if (!invalid && !jump && !zero_entry) {
    if (sh == 0) {
      const int d = A0 - B0;
      affine = (A1 - B1 == d) && (A2 - B2 == d) && (A3 - B3 == d);
      mC = d;
    } else {
      const int d0 = A0 - (B0 >> 1), d1 = A0 - ((B0 + 1) >> 1);
      const bool k0 = (A1 - (B1 >> 1) == d0) && (A2 - (B2 >> 1) == d0) && (A3 - (B3 >> 1) == d0);
      const bool k1 = (A1 - ((B1 + 1) >> 1) == d1) && (A2 - ((B2 + 1) >> 1) == d1) && (A3 - ((B3 + 1) >> 1) == d1);
      affine = k0 || k1;
      mK = k0 ? 0 : 1;
      mC = k0 ? d0 : d1;
    }
  }
  const int mS = (affine && !zero_entry) ? sh : 0;
  if (!affine) { mK = 0; mC = 0; }
  const int w01 = (A0 & 0xffff) | (A1 << 16), w23 = (A2 & 0xffff) | (A3 << 16);
  const int meta = p0 | (sh << 2) | (special ? 16 : 0);

  // the caller's unit: the total is returned as Q_63 + delta * ulp(Q_63); Q_63
  // must not be within the bracket width of a power of two
  const double Qt = rd_lane_f64(Q, 63);
  {
    const long long qb = __double_as_longlong(Qt) & ~3ll;
    const int ea = expo_field(__longlong_as_double(qb - G4)), eb = expo_field(__longlong_as_double(qb + 3 + G4));
    if (ea != eb || ea <= 64) invalid = true;
  }
  // ---- C. compose the tie-free lanes with a segmented scan (segments end at
  // the lanes that need the walk), then walk those lanes
  int sK = mK, sC = mC, sS = mS | (affine ? 0 : 64);  // bit 6: segment head
  auto combine = [&](int pK, int pC, int pS) {
    // (pK,pC,pS) = composite of the lanes further left; no-op if a head lies in between
    if (!(sS & 64)) {
      const int ps = pS & 63;
      sK = pK + ((pC + sK) << ps);
      sS = (ps + (sS & 63)) | (pS & 64);
      // sC unchanged
    }
  };
#define RL_SCAN_STEP(CTRL, RM)                                              \
  {                                                                         \
    const int pK = __builtin_amdgcn_update_dpp(0, sK, CTRL, RM, 0xf, true); \
    const int pC = __builtin_amdgcn_update_dpp(0, sC, CTRL, RM, 0xf, true); \
    const int pS = __builtin_amdgcn_update_dpp(0, sS, CTRL, RM, 0xf, true); \
    combine(pK, pC, pS);                                                    \
  }
  RL_SCAN_STEP(DPP_ROW_SHR + 1, 0xf)
  RL_SCAN_STEP(DPP_ROW_SHR + 2, 0xf)
  RL_SCAN_STEP(DPP_ROW_SHR + 4, 0xf)
  RL_SCAN_STEP(DPP_ROW_SHR + 8, 0xf)
  RL_SCAN_STEP(DPP_ROW_BCAST15, 0xa)
  RL_SCAN_STEP(DPP_ROW_BCAST31, 0xc)
#undef RL_SCAN_STEP
  // lanes not selected by a step read (0,0,0) = the identity map: harmless
  RL_STAT(0, 1);
  bool fallback = __ballot(invalid || (sS & 63) > 20) != 0ull;
  if constexpr (WAVES > 1) {  // the waves agree on the path
    if ((threadIdx.x & 63) == 0) lk.s->bad[lk.w] = fallback;
    lk.barrier();
    fallback = lk.s->bad[0] | lk.s->bad[1];
  }
  if (fallback) {
    RL_STAT(1, 1);
    return sum_exact_fallback_linked<S, WAVES>(term, lk);  // irregular step: literal serial sum
  }
  const unsigned long long tk3 = RL_CLK();
  unsigned long long todo = __ballot(!affine);
  RL_STAT(2, __builtin_popcountll(todo));
  {
    const unsigned long long sp = __ballot(special);
    (void)sp;
    RL_STAT(3, __builtin_popcountll(sp));
  }
  int delta = 0;  // lane 0 enters at exactly 0 (its local sum is exact, Q_0 == L_0)
  if constexpr (WAVES > 1) {
    // wave 1 walks after wave 0 and enters at wave 0's exact exit offset (same unit: the entry
    // value of its lane 0 IS wave 0's Q_63)
    if (lk.w == 1) {
      lk.barrier();
      delta = lk.s->delta;
    }
  }
  {  // start behind the last lane whose exit is known outright
    const unsigned long long cm = __ballot(constant);
    if (cm) {
      const int qc = 63 - __builtin_clzll(cm);
      delta = (int)(short)__builtin_amdgcn_readlane(w01, qc);  // its A0
      todo &= ~((2ull << qc) - 1ull);
    }
  }
  while (todo) {
    const int q = __builtin_ctzll(todo);
    todo &= todo - 1;
    {  // tie-free lanes since the previous walked lane (composite sits in lane q-1)
      const int K = __builtin_amdgcn_readlane(sK, q - 1), C = __builtin_amdgcn_readlane(sC, q - 1);
      const int sft = __builtin_amdgcn_readlane(sS, q - 1) & 63;
      delta = ((delta + K) >> sft) + C;
    }
    const int m = __builtin_amdgcn_readlane(meta, q);
    if (m & 16) {
      // multi-binade run: redo it from the exact entry value
      const double Pq = rd_lane_f64(P, q);
      const double uq = pow2_field(expo_field(Pq) - 52);
      double t = Pq + (double)delta * uq;  // exact
      double thC = term.th, nthC = term.nth;
      term.for_each(thC, nthC, [&](int, double x) {
        t += x;
        if constexpr (!REG_TERM) asm volatile("" : "+v"(thC), "+v"(nthC));  // (not t: a wait state per term, see the chain pass)
      });
      const double v = rd_lane_f64(t, q);
      // (the true exit lies between the bracketing runs, which end in Q's binade -- else the lane were invalid)
      delta = (int)((unsigned)__double_as_longlong(v) - (unsigned)__builtin_amdgcn_readlane((int)q_lo, q));
    } else {
      const int qp0 = m & 3, qsh = (m >> 2) & 3;
      const int h = (qp0 + delta) & 3;
      const int w = __builtin_amdgcn_readlane((h & 2) ? w23 : w01, q);
      const int A = (int)(short)((h & 1) ? (w >> 16) : w);
      const int B = h == 0 ? (-qp0 - G4) : (h == 3 ? (3 - qp0 + G4) : (h - qp0));
      delta = A + ((delta - B) >> qsh);
    }
  }
  {  // trailing tie-free lanes (lane 63 holds their composite; identity if it was walked)
    const int K = __builtin_amdgcn_readlane(sK, 63), C = __builtin_amdgcn_readlane(sC, 63);
    const int sft = __builtin_amdgcn_readlane(sS, 63) & 63;
    delta = ((delta + K) >> sft) + C;
  }
  {
    const unsigned long long tk4 = RL_CLK();
    (void)tk0; (void)tk1; (void)tk2; (void)tk3; (void)tk4;
    RL_STAT(4, tk1 - tk0);  // scan
    RL_STAT(5, tk2 - tk1);  // four runs
    RL_STAT(6, tk3 - tk2);  // classification + map scan
    RL_STAT(7, tk4 - tk3);  // walk
  }
  if constexpr (WAVES > 1) {
    const int lane = threadIdx.x & 63;
    if (lk.w == 0) {
      if (lane == 0) lk.s->delta = delta;
      lk.barrier();  // wave 1 starts its walk
      lk.barrier();  // ... and has finished it
    } else {
      if (lane == 0) lk.s->result = Qt + (double)delta * pow2_field(expo_field(Qt) - 52);
      lk.barrier();
    }
    const double r = lk.s->result;
    lk.barrier();  // read by everyone before the next sum's hand-over may overwrite it
    return r;
  }
  return Qt + (double)delta * pow2_field(expo_field(Qt) - 52);
}

// L = the lane's local serial sum of its terms (see sum_exact_fast)
template <int MODE, int S, int WAVES, typename T>
RL_DEV double wave_sum(const T &term, double L, WaveLink<WAVES> &lk) {
  lk.phase++;
  if constexpr (MODE == 1) {
    return sum_exact_fast<S, WAVES>(term, L, lk);
  } else if constexpr (MODE == 2) {
    return sum_exact_fallback_linked<S, WAVES>(term, lk);
  } else {
    const double t = wave_sum_butterfly(L);
    if constexpr (WAVES == 1) {
      return t;
    } else {  // the top level of the balanced tree over 128 lane sums
      const unsigned ph = lk.phase & 1u;
      if ((threadIdx.x & 63) == 0) lk.s->tot[ph][lk.w] = t;
      lk.barrier();
      return lk.s->tot[ph][0] + lk.s->tot[ph][1];
    }
  }
}
// one wave per target (K2, the test hook)
template <int MODE, int S, typename T>
RL_DEV double wave_sum(const T &term, double L) {
  WaveLink<1> lk;
  return wave_sum<MODE, S, 1>(term, L, lk);
}

template <int S, typename T>
RL_DEV double local_sum(const T &term) {
  double L = 0.0, th = term.th, nth = term.nth;
  term.for_each(th, nth, [&](int, double x) { L += x; });
  return L;
}

}  // namespace rl
