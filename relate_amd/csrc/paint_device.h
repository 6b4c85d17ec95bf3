// paint_device.h -- device-side building blocks of the Li-Stephens kernels.
//
// One 64-lane wavefront paints one target haplotype k.  The N donors (the
// target itself keeps a slot that is pinned to +0.0) live in registers, S
// doubles per lane (rl::Layout); one forward or backward step is
//   elementwise update (lane-local, fast_painting.cpp:288-295 / 481-488)
//   + one normalising sum over all donors (fast_painting.cpp:300-303 / 495-503)
// and the only data read per step is the site's row of lane masks.
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

// a / b, correctly rounded, for a divisor whose correctly rounded reciprocal y = RN(1 / b) is at hand (theta and
// 1 - theta: fast_painting.cpp:474-475 divide by them in every backward step).  The compiler's IEEE division is
// ~16 dependent instructions (v_div_scale x 2, v_rcp, a Newton chain, v_div_fmas, v_div_fixup); with y known:
//   q0 = RN(a y)                      within 2 ulp of a/b   (y and the product each err by <= 2^-53 relative)
//   q1 = RN(q0 + RN(a - b q0) y)      faithful              (the residual by FMA)
//   q2 = RN(q1 + (a - b q1) y)        = RN(a / b)           Markstein's theorem: q faithful, the residual exact (it is,
//                                                            for a faithful q), y = RN(1/b)  =>  RN(q + r y) = RN(a/b)
// a is a positive normal number far from the ends of the range here (1e-10 <= sum <= 1e10 times an interval
// coefficient), so neither the products nor the residuals under- or overflow.  (2.2e9 random a per divisor against
// the hardware division on the host: no difference, already after q1.)
RL_DEV double div_by_const(double a, double b, double y) {
  const double q0 = a * y;
  const double q1 = __builtin_fma(__builtin_fma(-q0, b, a), y, q0);
  return __builtin_fma(__builtin_fma(-q1, b, a), y, q1);
}

// ---- lane-mask panel -------------------------------------------------------
// The Li-Stephens kernels (K1 stepping stones, K2 RePaint) read the panel in "lane-mask" form: for site s and
// register j one 64-bit word whose bit l says that the donor lane l holds in
// register j is ANCESTRAL at s (layout over all N donors, the target included;
// slots past a lane's run read 0).  At a site where the target is derived this
// is exactly the mismatch mask "target derived, donor ancestral"
// (fast_painting.cpp:290); where it is ancestral (only the first / last site
// can be) the kernel reads the all-zero row L instead.  The words are
// wave-uniform: they are fetched with scalar loads straight into SGPRs and
// moved into EXEC -- no per-register v_cmp, no per-lane bit extraction.
//
// Each wave issues at most one instruction per issue slot, scalar or vector, so
// the EXEC writes are batched: a chunk of registers does all its unmasked
// vector work under one "exec = -1".
typedef unsigned long long u64;
typedef u64 u64x4 __attribute__((ext_vector_type(4)));
typedef u64 u64x8 __attribute__((ext_vector_type(8)));
typedef u64 u64x16 __attribute__((ext_vector_type(16)));
// constant address space: uniform loads become s_load_dwordx8 / x16
typedef const __attribute__((address_space(4))) u64 *MaskRow;
template <int CH> struct MaskChunk;
template <> struct MaskChunk<4> { typedef u64x4 type; };
template <> struct MaskChunk<8> { typedef u64x8 type; };
template <> struct MaskChunk<16> { typedef u64x16 type; };
template <int CH>
RL_DEV typename MaskChunk<CH>::type load_masks(MaskRow row, int c) {
  typedef const __attribute__((address_space(4))) typename MaskChunk<CH>::type *P;
  return ((P)row)[c];
}

// Walk the S masks of a row in chunks of CH: wait for chunk c, issue the load
// of chunk c+1, then run f(j0, chunk) on chunk c.  (s_load results return out
// of order, so the only wait is lgkmcnt(0): the next load must be issued AFTER
// the wait for the current chunk, which the empty asm enforces.)
template <int S, int CH, typename F>
RL_DEV void for_each_chunk(MaskRow row, F &&f) {
  static_assert(S % CH == 0, "S must be a multiple of the chunk");
  auto cur = load_masks<CH>(row, 0);
#pragma unroll
  for (int c = 0; c < S / CH; c++) {
    auto nxt = cur;
    if (c + 1 < S / CH) {
      asm volatile("" : "+s"(row) : "s"(cur[0]));
      nxt = load_masks<CH>(row, c + 1);
    }
    f(c * CH, cur);
    cur = nxt;
  }
}
// The same with chunk 0 already requested by the caller (`first`).
template <int S, int CH, typename C, typename F>
RL_DEV void for_each_chunk_from(MaskRow row, C first, F &&f) {
  static_assert(S % CH == 0, "S must be a multiple of the chunk");
  C cur = first;
#pragma unroll
  for (int c = 0; c < S / CH; c++) {
    C nxt = cur;
    if (c + 1 < S / CH) {
      asm volatile("" : "+s"(row) : "s"(cur[0]));
      nxt = load_masks<CH>(row, c + 1);
    }
    f(c * CH, cur);
    cur = nxt;
  }
}
// Two rows at once (the backward update needs the masks of the later site and
// of this one), chunk 0 of each already requested by the caller, and with the
// chunks of `vrow` -- per-register validity masks, the lanes whose run reaches
// that register -- alongside for the chunks that touch the last TAIL
// registers: f(j0, chunk a, chunk b, validity chunk).
template <int S, int CH, int TAIL, typename C, typename F>
RL_DEV void for_each_chunk2_tail(MaskRow rowa, MaskRow rowb, MaskRow vrow, C firsta, C firstb, F &&f) {
  static_assert(S % CH == 0, "S must be a multiple of the chunk");
  C ca = firsta, cb = firstb, va = firsta;
  if (CH > S - TAIL) va = load_masks<CH>(vrow, 0);
#pragma unroll
  for (int c = 0; c < S / CH; c++) {
    C na = ca, nb = cb, nva = va;
    if (c + 1 < S / CH) {
      asm volatile("" : "+s"(rowa), "+s"(rowb), "+s"(vrow) : "s"(ca[0]), "s"(cb[0]));
      na = load_masks<CH>(rowa, c + 1);
      nb = load_masks<CH>(rowb, c + 1);
      if ((c + 2) * CH > S - TAIL) nva = load_masks<CH>(vrow, c + 1);
    }
    f(c * CH, ca, cb, va);
    ca = na;
    cb = nb;
    va = nva;
  }
}

// One target's view of the layout over all N donors.  The donors are cut
// into 64*WAVES balanced runs ("virtual lanes"); wave w of the target's
// workgroup holds virtual lanes 64w .. 64w+63 (WAVES = 1 everywhere except in
// the stepping-stone kernel for N > 5120).
template <int S>
struct PaintLane {
  int lane, start, len, k;
  int q;         // registers 0..q-1 are valid in every lane
  u64 rem_mask;  // lanes of this wave in which register q is valid
  int jk;        // donor k itself sits in register jk ...
  u64 kbit;      // ... of this lane of this wave (0: another wave holds it): pinned to +0.0
  RL_DEV void init(const Layout &lay, int k_, int wave = 0) {
    lane = threadIdx.x & 63;
    const int vl = wave * 64 + lane;  // virtual lane
    k = k_;
    q = lay.q;
    start = vl * lay.q + (vl < lay.rem ? vl : lay.rem);
    len = lay.q + (vl < lay.rem ? 1 : 0);
    const int rem_here = lay.rem - wave * 64;  // virtual lanes of this wave that hold one more
    rem_mask = rem_here >= 64 ? ~0ull : (rem_here > 0 ? (~0ull >> (64 - rem_here)) : 0ull);
    const int big = lay.rem * (lay.q + 1);
    const int lk = k < big ? k / (lay.q + 1) : lay.rem + (k - big) / (lay.q > 0 ? lay.q : 1);
    jk = k - (lk * lay.q + (lk < lay.rem ? lk : lay.rem));
    kbit = (lk >> 6) == wave ? 1ull << (lk & 63) : 0ull;
  }
  RL_DEV u64 valid(int j) const { return j < q ? ~0ull : (j == q ? rem_mask : 0ull); }
};

// ---- helpers shared by K1 and K2 -------------------------------------------
// Kernel arguments that only rare code needs (stone / bookkeeping writes) are
// read from the kernarg segment at the point of use, through a pointer the
// compiler cannot see through: kept in SGPRs across the site loop they would
// crowd out the mask chunks (and spill into VGPR lanes at every step).  P is the
// kernel's FIRST argument.
template <typename P>
RL_DEV const __attribute__((address_space(4))) P *cold_params() {
  typedef const __attribute__((address_space(4))) P *Q;
  Q q = (Q)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return q;
}
// a loop constant that the exec-masked asm takes in a VGPR: keep it there
RL_DEV double in_vgpr(double v) {
  asm volatile("" : "+v"(v));
  return v;
}
// site word of the plan (bit 31: the target is derived there) -> the row of
// mismatch masks the target sees: the site's row, or the all-zero row L
// (the panel holds WAVES consecutive rows of S words per site, one per wave of the target's workgroup)
RL_DEV MaskRow site_row(const u64 *masks, int S, int L, int sv, int waves = 1, int wave = 0) {
  sv = __builtin_amdgcn_readfirstlane(sv);
  const int r = sv < 0 ? (sv & 0x7fffffff) : L;
  return (MaskRow)(masks + ((size_t)r * waves + wave) * S);
}
// Pull the row of a site that the NEXT step will read into L2 (one dword per
// 16 bytes; the value is kept alive until then, which also parks the wait
// for it a whole step later).  Scalar loads that miss L2 cost ~800 cycles per
// chunk of masks; from L2 the other wave on the SIMD covers them.
RL_DEV uint32_t touch_row(const u64 *masks, int S, int sv, int lane, int waves = 1, int wave = 0) {
  sv = __builtin_amdgcn_readfirstlane(sv);
  const uint32_t *row = (const uint32_t *)(masks + ((size_t)(sv & 0x7fffffff) * waves + wave) * S);
  const int o = lane * 4, last = S * 2 - 1;
  return row[o < last ? o : last];
}
RL_DEV void retire_touch(uint32_t t) { asm volatile("" : : "v"(t)); }

// ---- exec-masked vector ops ----------------------------------------------
RL_DEV void masked_mov(double &t, u64 mask, double v) {
  asm volatile("s_mov_b64 exec, %1\n\tv_mov_b64 %0, %2\n\ts_mov_b64 exec, -1" : "+v"(t) : "s"(mask), "v"(v));
}
// v[i] *= k in the lanes of m[OFF + i]
template <int OFF = 0, typename M>
RL_DEV void masked_mul8(double *v, const M &m, double k) {
  asm volatile(
      "s_mov_b64 exec, %8\n\tv_mul_f64 %0, %0, %16\n\t"
      "s_mov_b64 exec, %9\n\tv_mul_f64 %1, %1, %16\n\t"
      "s_mov_b64 exec, %10\n\tv_mul_f64 %2, %2, %16\n\t"
      "s_mov_b64 exec, %11\n\tv_mul_f64 %3, %3, %16\n\t"
      "s_mov_b64 exec, %12\n\tv_mul_f64 %4, %4, %16\n\t"
      "s_mov_b64 exec, %13\n\tv_mul_f64 %5, %5, %16\n\t"
      "s_mov_b64 exec, %14\n\tv_mul_f64 %6, %6, %16\n\t"
      "s_mov_b64 exec, %15\n\tv_mul_f64 %7, %7, %16\n\t"
      "s_mov_b64 exec, -1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
      : "s"(m[OFF]), "s"(m[OFF + 1]), "s"(m[OFF + 2]), "s"(m[OFF + 3]), "s"(m[OFF + 4]), "s"(m[OFF + 5]),
        "s"(m[OFF + 6]), "s"(m[OFF + 7]), "v"(k));
}
// v += k in the lanes whose run reaches register j (j < len; j a constant
// after unrolling): the validity test of a TAIL register, one v_cmp from the
// lane's run length
RL_DEV void tail_add(double &v, int len, int j, double k) {
  asm volatile("v_cmp_lt_i32 vcc, %2, %1\n\ts_mov_b64 exec, vcc\n\tv_add_f64 %0, %0, %3\n\ts_mov_b64 exec, -1"
               : "+v"(v)
               : "v"(len), "i"(j), "v"(k)
               : "vcc");
}
// x[i] = (lane in m[i] ? th : nth) * b[i]   (fast_painting.cpp:495-503)
template <typename M>
RL_DEV void weighted4(double (&x)[4], double b0, double b1, double b2, double b3, const M &m, double th, double nth) {
  asm volatile(
      "v_mul_f64 %0, %4, %13\n\tv_mul_f64 %1, %5, %13\n\tv_mul_f64 %2, %6, %13\n\tv_mul_f64 %3, %7, %13\n\t"
      "s_mov_b64 exec, %8\n\tv_mul_f64 %0, %4, %12\n\t"
      "s_mov_b64 exec, %9\n\tv_mul_f64 %1, %5, %12\n\t"
      "s_mov_b64 exec, %10\n\tv_mul_f64 %2, %6, %12\n\t"
      "s_mov_b64 exec, %11\n\tv_mul_f64 %3, %7, %12\n\t"
      "s_mov_b64 exec, -1"
      : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
      : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]), "v"(th), "v"(nth));
}
// backward update of four registers (fast_painting.cpp:483-484):
//   v = ((v + mis*bt) + b1) * (mis ? K1 : 1.0), mis = lanes of mn[i],
// and their weighted terms x = (lanes of mh[i] ? th : nth) * v
template <typename M>
RL_DEV void backward4(double (&v)[4], double (&x)[4], const M &mn, const M &mh, double bt, double b1, double K1,
                      double th, double nth) {
  asm volatile(
      "s_mov_b64 exec, %8\n\tv_add_f64 %0, %0, %16\n\t"
      "s_mov_b64 exec, %9\n\tv_add_f64 %1, %1, %16\n\t"
      "s_mov_b64 exec, %10\n\tv_add_f64 %2, %2, %16\n\t"
      "s_mov_b64 exec, %11\n\tv_add_f64 %3, %3, %16\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_add_f64 %0, %0, %17\n\tv_add_f64 %1, %1, %17\n\tv_add_f64 %2, %2, %17\n\tv_add_f64 %3, %3, %17\n\t"
      "s_mov_b64 exec, %8\n\tv_mul_f64 %0, %0, %18\n\t"
      "s_mov_b64 exec, %9\n\tv_mul_f64 %1, %1, %18\n\t"
      "s_mov_b64 exec, %10\n\tv_mul_f64 %2, %2, %18\n\t"
      "s_mov_b64 exec, %11\n\tv_mul_f64 %3, %3, %18\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_mul_f64 %4, %0, %20\n\tv_mul_f64 %5, %1, %20\n\tv_mul_f64 %6, %2, %20\n\tv_mul_f64 %7, %3, %20\n\t"
      "s_mov_b64 exec, %12\n\tv_mul_f64 %4, %0, %19\n\t"
      "s_mov_b64 exec, %13\n\tv_mul_f64 %5, %1, %19\n\t"
      "s_mov_b64 exec, %14\n\tv_mul_f64 %6, %2, %19\n\t"
      "s_mov_b64 exec, %15\n\tv_mul_f64 %7, %3, %19\n\t"
      "s_mov_b64 exec, -1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
      : "s"(mn[0]), "s"(mn[1]), "s"(mn[2]), "s"(mn[3]), "s"(mh[0]), "s"(mh[1]), "s"(mh[2]), "s"(mh[3]), "v"(bt),
        "v"(b1), "v"(K1), "v"(th), "v"(nth));
}
// the same with the "+ b1" confined to the lanes of va[i] (registers of the TAIL)
template <typename M>
RL_DEV void backward4_tail(double (&v)[4], double (&x)[4], const M &mn, const M &mh, const M &va, double bt,
                           double b1, double K1, double th, double nth) {
  asm volatile(
      "s_mov_b64 exec, %8\n\tv_add_f64 %0, %0, %16\n\t"
      "s_mov_b64 exec, %9\n\tv_add_f64 %1, %1, %16\n\t"
      "s_mov_b64 exec, %10\n\tv_add_f64 %2, %2, %16\n\t"
      "s_mov_b64 exec, %11\n\tv_add_f64 %3, %3, %16\n\t"
      "s_mov_b64 exec, %21\n\tv_add_f64 %0, %0, %17\n\t"
      "s_mov_b64 exec, %22\n\tv_add_f64 %1, %1, %17\n\t"
      "s_mov_b64 exec, %23\n\tv_add_f64 %2, %2, %17\n\t"
      "s_mov_b64 exec, %24\n\tv_add_f64 %3, %3, %17\n\t"
      "s_mov_b64 exec, %8\n\tv_mul_f64 %0, %0, %18\n\t"
      "s_mov_b64 exec, %9\n\tv_mul_f64 %1, %1, %18\n\t"
      "s_mov_b64 exec, %10\n\tv_mul_f64 %2, %2, %18\n\t"
      "s_mov_b64 exec, %11\n\tv_mul_f64 %3, %3, %18\n\t"
      "s_mov_b64 exec, -1\n\t"
      "v_mul_f64 %4, %0, %20\n\tv_mul_f64 %5, %1, %20\n\tv_mul_f64 %6, %2, %20\n\tv_mul_f64 %7, %3, %20\n\t"
      "s_mov_b64 exec, %12\n\tv_mul_f64 %4, %0, %19\n\t"
      "s_mov_b64 exec, %13\n\tv_mul_f64 %5, %1, %19\n\t"
      "s_mov_b64 exec, %14\n\tv_mul_f64 %6, %2, %19\n\t"
      "s_mov_b64 exec, %15\n\tv_mul_f64 %7, %3, %19\n\t"
      "s_mov_b64 exec, -1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
      : "s"(mn[0]), "s"(mn[1]), "s"(mn[2]), "s"(mn[3]), "s"(mh[0]), "s"(mh[1]), "s"(mh[2]), "s"(mh[3]), "v"(bt),
        "v"(b1), "v"(K1), "v"(th), "v"(nth), "s"(va[0]), "s"(va[1]), "s"(va[2]), "s"(va[3]));
}

// a[j] = v in the lanes of `bit`, j wave-uniform but not a compile-time
// constant: a branch tree over static cases, each one exec-masked v_mov_b64.
// Written the obvious way ("+v"(a[J]) in every case) the cases define new
// values of a[J] that merge after the switch, and the register allocator
// settles the merges with up to S register copies per call.  So the cases
// take a[J] as an INPUT and overwrite its register behind the compiler's
// back; the empty asm statements before and after (every a[i] in and out,
// same register) pin the array to registers across the switch and tell the
// compiler that any element may have changed.
RL_DEV void poke_slot(const double &t, u64 mask, double v) {
  asm volatile("s_mov_b64 exec, %1\n\tv_mov_b64 %0, %2\n\ts_mov_b64 exec, -1" : : "v"(t), "s"(mask), "v"(v));
}
template <int S>
RL_DEV void pin_registers(double (&a)[S]) {
  static_assert(S % 8 == 0, "S must be a multiple of 8");
#pragma unroll
  for (int i = 0; i < S; i += 8)
    asm volatile("" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]), "+v"(a[i + 4]), "+v"(a[i + 5]),
                 "+v"(a[i + 6]), "+v"(a[i + 7]));
}
#define RL_SLOT(J) \
  case J:          \
    if constexpr ((J) < S) poke_slot(a[(J) < S ? (J) : 0], bit, v); \
    break;
#define RL_SLOT8(B) RL_SLOT(B) RL_SLOT(B + 1) RL_SLOT(B + 2) RL_SLOT(B + 3) RL_SLOT(B + 4) RL_SLOT(B + 5) RL_SLOT(B + 6) RL_SLOT(B + 7)
template <int S>
RL_DEV void set_slot(double (&a)[S], int j, u64 bit, double v) {
  static_assert(S <= 80, "extend the case list");
  pin_registers<S>(a);
  switch (j) {
    RL_SLOT8(0) RL_SLOT8(8) RL_SLOT8(16) RL_SLOT8(24) RL_SLOT8(32) RL_SLOT8(40) RL_SLOT8(48) RL_SLOT8(56)
    RL_SLOT8(64) RL_SLOT8(72)
    default: break;
  }
  pin_registers<S>(a);
}
#undef RL_SLOT8
#undef RL_SLOT

}  // namespace rl
