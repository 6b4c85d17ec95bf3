// paint_device.h -- device-side building blocks of the Li-Stephens kernels.
//
// One 64-lane wavefront paints one target haplotype k.  The P = N-1 donors
// live in registers, S doubles per lane (rl::Layout); one forward or backward
// step is
//   elementwise update (lane-local, fast_painting.cpp:288-295 / 481-488)
//   + one normalising sum over all donors (fast_painting.cpp:300-303 / 495-503)
// and the only data read per step is the site's N-bit panel row.
//
// Compiled with -ffp-contract=off: every operator is one IEEE operation, in
// the operand order of the reference.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace rl {

#define RL_DEV __device__ __forceinline__

RL_DEV double wave_bcast(double v, int src_lane) { return __shfl(v, src_lane, 64); }

// DPP move of a double (two 32-bit DPP movs); lanes a control does not select read +0.0
template <int CTRL, int ROW_MASK = 0xf>
RL_DEV double dpp_mov_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes as a balanced binary tree in lane order -- the value an
// xor-butterfly with masks 1,2,4,8,16,32 gives (IEEE addition is commutative),
// which is what the oracle's RO_SUM_LANES reproduces -- but on the DPP
// crossbar instead of six LDS round trips: quad swaps, half-row and row
// mirrors, then the two row broadcasts; the total lands in lane 63.
RL_DEV double wave_sum_butterfly(double v) {
  v += dpp_mov_f64<0xB1>(v);         // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E>(v);         // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141>(v);        // row_half_mirror
  v += dpp_mov_f64<0x140>(v);        // row_mirror
  v += dpp_mov_f64<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
  v += dpp_mov_f64<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// fast_log.hpp:6-21
RL_DEV float fast_log_dev(float val) {
  int x = __float_as_int(val);
  const int log_2 = ((x >> 23) & 255) - 128;
  x &= ~(255 << 23);
  x += 127 << 23;
  val = __int_as_float(x);
  val = ((-1.0f / 3) * val + 2) * val - 2.0f / 3;
  return (val + log_2) * 0.69314718f;
}

template <int S>
struct LaneCtx {
  static constexpr int NW = (S + 31) / 32;  // words of lane bits
  int lane;
  int start;  // first physical donor index of this lane
  int len;    // valid registers
  int k;      // target
  int w0;     // first panel word this lane reads
  int sh;     // bit shift of `start` inside w0
  uint32_t lowmask[NW];  // bit i set: register i maps to donor start+i (< k)

  RL_DEV void init(const Layout &lay, int k_) {
    lane = threadIdx.x & 63;
    k = k_;
    start = lane * lay.q + (lane < lay.rem ? lane : lay.rem);
    len = lay.q + (lane < lay.rem ? 1 : 0);
    w0 = start >> 5;
    sh = start & 31;
    int t = k - start;  // registers [0,t) are donors below k
    t = t < 0 ? 0 : (t > S ? S : t);
#pragma unroll
    for (int w = 0; w < NW; w++) {
      int r = t - 32 * w;
      lowmask[w] = r >= 32 ? 0xffffffffu : (r <= 0 ? 0u : ((1u << r) - 1u));
    }
  }
  // donor index of register i
  RL_DEV int donor(int i) const {
    int p = start + i;
    return p + (p >= k ? 1 : 0);
  }
};

// Raw panel words of one row for this lane (prefetchable).
template <int S>
struct RawBits {
  static constexpr int NR = (S + 31) / 32 + 1;
  uint32_t r[NR];
  RL_DEV void load(const uint32_t *__restrict__ row, int w0) {
#pragma unroll
    for (int t = 0; t < NR; t++) r[t] = row[w0 + t];
  }
};

// Donor bits of the lane in register order, donor k deleted.
template <int S>
struct LaneBits {
  static constexpr int NW = (S + 31) / 32;
  uint32_t w[NW];
  RL_DEV void from_raw(const RawBits<S> &raw, const LaneCtx<S> &lc) {
#pragma unroll
    for (int t = 0; t < NW; t++) {
      uint64_t lo = ((uint64_t)raw.r[t + 1] << 32) | raw.r[t];
      uint32_t a = (uint32_t)(lo >> lc.sh);        // donors start+32t+j
      uint32_t b = (uint32_t)(lo >> (lc.sh + 1));  // donors start+32t+j+1
      w[t] = (a & lc.lowmask[t]) | (b & ~lc.lowmask[t]);
    }
  }
  // mismatch mask "target derived, donor ancestral" (fast_painting.cpp:290)
  RL_DEV void to_mismatch(bool seqk) {
#pragma unroll
    for (int t = 0; t < NW; t++) w[t] = seqk ? ~w[t] : 0u;
  }
  RL_DEV bool get(int i) const { return (w[i >> 5] >> (i & 31)) & 1u; }
};

// Lane-masked double ops: `if (lane in mask) t = t (op) k`, executed by
// narrowing EXEC to the mask for one instruction instead of computing both
// variants and selecting (2 x v_cndmask_b32 per double).  mask is a
// wave-uniform 64-bit lane mask (one v_cmp away from the lanes' bits).
RL_DEV void masked_mul(double &t, unsigned long long mask, double k) {
  asm volatile("s_mov_b64 exec, %1\n\tv_mul_f64 %0, %0, %2\n\ts_mov_b64 exec, -1" : "+v"(t) : "s"(mask), "v"(k));
}
RL_DEV void masked_add(double &t, unsigned long long mask, double k) {
  asm volatile("s_mov_b64 exec, %1\n\tv_add_f64 %0, %0, %2\n\ts_mov_b64 exec, -1" : "+v"(t) : "s"(mask), "v"(k));
}

// ---- normalising sums ---------------------------------------------------
// EXACT: the donors are added left to right in physical order (= donor order
// with the zero of donor k skipped, which is a no-op), lane 0's registers
// first.  Lane l continues from the running sum handed over by lane l-1.
template <int S, typename F>
RL_DEV double sum_exact(F term) {
  double s = 0.0;
  for (int l = 0; l < 64; l++) {
    double tmp = s;
#pragma unroll
    for (int i = 0; i < S; i++) tmp += term(i);
    s = wave_bcast(tmp, l);
  }
  return s;
}

template <int S, typename F>
RL_DEV double sum_lanes(F term) {
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < S; i++) s += term(i);
  return wave_sum_butterfly(s);
}

}  // namespace rl
