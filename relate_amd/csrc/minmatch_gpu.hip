// minmatch_gpu.hip -- MinMatch::QuickBuild on the GPU: one workgroup builds one tree, matrices in HBM.
//
// Reference: src/tree_builder.cpp (MinMatch, sample_ages empty): :59-146 / :1647-1735 (Initialize),
// :296-598 / :1844-2070 (Coalesce), :1061-1303 / :2358-2644 (QuickBuild).  The host builder
// (minmatch.cpp) states the algorithm and why a merge splits into a parallel part and an ordered part with
// identical results; this file is the same split on one workgroup:
//   * the N-1 merges of a tree are sequential and each walks down two columns of a 100 MB matrix -- a new
//     line per cluster: on a host dozens of open sections are bound by DRAM, in HBM one
//     workgroup per tree leaves the other 255 CUs to the trees of the other sections;
//   * the parts that are order-free (distance updates, row-minimum rescans, candidate tests, reductions) run
//     on all 1024 threads; the random draws -- one per feasible pair, in the reference's order -- and the
//     candidate bookkeeping run on thread 0 over lists the parallel parts leave in order;
//   * std::mt19937 (seed 1 per build) and libstdc++'s generate_canonical<double, 53> are restated below.
// The symmetric fallback (no mutually closest pair left, :255-293 / :968-1058) is here too: "the first cluster in
// scan order that reaches the minimum" is a lexicographic (value, position) reduction.  The state both builders
// carry from tree to tree (min_values_CF and the stale candidate indices, minmatch.h) is copied in before and
// out after every build, so host and device builders can alternate on a section.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "common.h"
#include "minmatch.h"

namespace rl {

namespace {

constexpr int MM_BLOCK = 1024;
constexpr int MM_WAVES = MM_BLOCK / 64;
constexpr int MM_GATHER = 32;  // updated clusters whose half of the candidate test is kept as mask bits

struct MMParams {
  int N;
  float threshold, threshold_CF;
  float *D;   // [N*N] destroyed
  float *CF;  // [N*N] destroyed, or nullptr
  float *SYM;  // [N*N] room for the symmetric matrix
  float *min_values_sym, *mcs_dist;
  int *mcs_lin1, *mcs_lin2;
  float *min_values, *min_values_CF;
  float *mc_dist, *mc_dist2;
  int *mc_lin1, *mc_lin2;
  int *cluster_index, *cluster_index2, *convert_index;
  float *cluster_size;
  unsigned char *kflag;
  unsigned *kmask;
  int *visit_list, *cand_j, *upd_pos;  // [N] each
  int *feas, *feas_off;                // [N*?] feasible partners of the rebuilt clusters, [N+1] offsets
  int *rowlist;                        // [MM_WAVES][N] pair-scan survivors per wave
  int *parent, *child_left, *child_right;
  int *status;
  long long feas_cap;
  long long *timers;  // optional: 100 MHz ticks per phase (RELATE_AMD_TIMING)
};

struct Rng {  // std::mt19937
  uint32_t mt[624];
  int idx;
};

__host__ __device__ inline void rng_seed(Rng &r, uint32_t seed) {
  r.mt[0] = seed;
  for (int i = 1; i < 624; i++) r.mt[i] = 1812433253u * (r.mt[i - 1] ^ (r.mt[i - 1] >> 30)) + (uint32_t)i;
  r.idx = 624;
}
__host__ __device__ inline uint32_t rng_next(Rng &r) {
  if (r.idx >= 624) {
    for (int i = 0; i < 624; i++) {
      const uint32_t y = (r.mt[i] & 0x80000000u) | (r.mt[(i + 1) % 624] & 0x7fffffffu);
      r.mt[i] = r.mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    r.idx = 0;
  }
  uint32_t y = r.mt[r.idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
// std::uniform_real_distribution<double>(0,1)(rng) of libstdc++: generate_canonical<double, 53> = two draws,
// sum = g1 + g2 * 2^32 in double, / 2^64, a result of 1 replaced by nextafter(1, 0)
__host__ __device__ inline double rng_unif(Rng &r) {
  const double g1 = (double)rng_next(r);
  const double g2 = (double)rng_next(r);
  const double sum = g1 + g2 * 4294967296.0;
  double ret = sum / 18446744073709551616.0;
  if (ret >= 1.0) ret = 0.99999999999999988897769753748434595763683319091796875;
  return ret;
}

struct Best {
  float dist, dist2;
  int lin1, lin2;
};

struct Shared {
  Rng rng;
  Best best, best_sym;
  int use_sym;
  int n, ipos;
  int wave_i[MM_WAVES];
  float wave_f[MM_WAVES];
  int wave_i2[MM_WAVES];
  int wave_i3[MM_WAVES];
  float wave_f2[MM_WAVES];
  int count;
  int rowcount[MM_WAVES];
  float sym_dist;
};

__device__ inline float wave_min_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline float block_min_f(float v, float *buf) {
  v = wave_min_f(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = buf[0];
#pragma unroll
  for (int w = 1; w < MM_WAVES; w++) r = fminf(r, buf[w]);
  return r;
}
// exclusive prefix of v over the threads in order; *total = sum
__device__ inline int block_scan(int v, int *total, int *buf) {
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if ((int)(threadIdx.x & 63) >= o) x += y;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 63) buf[threadIdx.x >> 6] = x;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < MM_WAVES; w++) {
    if (w < (int)(threadIdx.x >> 6)) base += buf[w];
    tot += buf[w];
  }
  *total = tot;
  return base + x - v;
}

// three independent minima with one exchange
__device__ inline void block_min3(float &f, int &a, int &b, float *bf, int *ba, int *bb) {
  f = wave_min_f(f);
  a = wave_min_i(a);
  b = wave_min_i(b);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    bf[threadIdx.x >> 6] = f;
    ba[threadIdx.x >> 6] = a;
    bb[threadIdx.x >> 6] = b;
  }
  __syncthreads();
  f = bf[0];
  a = ba[0];
  b = bb[0];
#pragma unroll
  for (int w = 1; w < MM_WAVES; w++) {
    f = fminf(f, bf[w]);
    a = min(a, ba[w]);
    b = min(b, bb[w]);
  }
}
// lexicographic minimum of (d1, d2, pos)
__device__ inline bool lex_less(float a1, float a2, int ap, float b1, float b2, int bp) {
  return a1 < b1 || (a1 == b1 && (a2 < b2 || (a2 == b2 && ap < bp)));
}
__device__ inline void block_lex_min(float &d1, float &d2, int &pos, float *b1, float *b2, int *bp) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float o1 = __shfl_xor(d1, o, 64), o2 = __shfl_xor(d2, o, 64);
    const int op = __shfl_xor(pos, o, 64);
    if (lex_less(o1, o2, op, d1, d2, pos)) {
      d1 = o1;
      d2 = o2;
      pos = op;
    }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    b1[threadIdx.x >> 6] = d1;
    b2[threadIdx.x >> 6] = d2;
    bp[threadIdx.x >> 6] = pos;
  }
  __syncthreads();
  d1 = b1[0];
  d2 = b2[0];
  pos = bp[0];
#pragma unroll
  for (int w = 1; w < MM_WAVES; w++)
    if (lex_less(b1[w], b2[w], bp[w], d1, d2, pos)) {
      d1 = b1[w];
      d2 = b2[w];
      pos = bp[w];
    }
}
// two exclusive prefixes with one exchange
__device__ inline void block_scan2(int v1, int v2, int &e1, int &e2, int &t1, int &t2, int *buf1, int *buf2) {
  int x1 = v1, x2 = v2;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y1 = __shfl_up(x1, o, 64), y2 = __shfl_up(x2, o, 64);
    if ((int)(threadIdx.x & 63) >= o) {
      x1 += y1;
      x2 += y2;
    }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 63) {
    buf1[threadIdx.x >> 6] = x1;
    buf2[threadIdx.x >> 6] = x2;
  }
  __syncthreads();
  int base1 = 0, base2 = 0;
  t1 = 0;
  t2 = 0;
#pragma unroll
  for (int w = 0; w < MM_WAVES; w++) {
    if (w < (int)(threadIdx.x >> 6)) {
      base1 += buf1[w];
      base2 += buf2[w];
    }
    t1 += buf1[w];
    t2 += buf2[w];
  }
  e1 = base1 + x1 - v1;
  e2 = base2 + x2 - v2;
}

#define DD(a, b) p.D[(size_t)(a) * N + (b)]
#define CC(a, b) p.CF[(size_t)(a) * N + (b)]
#define SS(a, b) p.SYM[(size_t)(a) * N + (b)]

// One feasible pair (thread 0): symmetric distance, one draw, both clusters' best candidate
// (tree_builder.cpp:1699-1716); minmatch.cpp: consider().
__device__ inline void consider(const MMParams &p, Shared &sh, int x, int y) {
  const int N = p.N;
  float sym;
  if (p.CF) {
    const bool both = (CC(x, y) <= p.min_values_CF[x]) && (CC(y, x) <= p.min_values_CF[y]);
    sym = both ? 0.0f : DD(y, x) + DD(x, y);
  } else {
    sym = DD(y, x) + DD(x, y);
  }
  sh.sym_dist = sym;
  const float rnd = (float)rng_unif(sh.rng);
  const float ad = p.mc_dist[x], ad2 = p.mc_dist2[x];
  if (ad > sym || (ad == sym && ad2 > rnd)) {
    p.mc_lin1[x] = x;
    p.mc_lin2[x] = y;
    p.mc_dist[x] = sym;
    p.mc_dist2[x] = rnd;
  }
  const float bd = p.mc_dist[y], bd2 = p.mc_dist2[y];
  if (bd > sym || (bd == sym && bd2 > rnd)) {
    p.mc_lin1[y] = x;
    p.mc_lin2[y] = y;
    p.mc_dist[y] = sym;
    p.mc_dist2[y] = rnd;
  }
}

// one workgroup per tree: workgroup b builds the tree of params[b]
__global__ void __launch_bounds__(MM_BLOCK, 1) minmatch_kernel(const MMParams *__restrict__ params) {
  const MMParams p = params[blockIdx.x];
  __shared__ Shared sh;
  const int N = p.N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float INF = INFINITY;
  const float threshold = p.threshold, threshold_CF = p.threshold_CF;
  int *ci = p.cluster_index, *ci_next = p.cluster_index2;  // the live clusters in order; the list is rewritten per merge
  long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tmark = wall_clock64();
  const long long tstart = tmark, cstart = clock64();
#define LAP(x)                           \
  {                                      \
    const long long tn = wall_clock64(); \
    tacc[x] += tn - tmark;               \
    tmark = tn;                          \
  }

  // ---- QuickBuild set-up (:1061-1100)
  for (int c = tid; c < N; c += MM_BLOCK) {
    ci[c] = c;
    p.convert_index[c] = c;
    p.cluster_size[c] = 1.0f;
    p.min_values[c] = INF;
    p.mc_dist[c] = INF;
    p.mc_dist2[c] = INF;
  }
  for (int c = tid; c < 2 * N - 1; c += MM_BLOCK) p.parent[c] = -1;
  if (tid == 0) {
    rng_seed(sh.rng, 1u);
    sh.best.dist = INF;
    sh.best.dist2 = INF;
    sh.best.lin1 = -1;
    sh.best.lin2 = -1;
    sh.best_sym = sh.best;
    sh.use_sym = 0;
    sh.n = N;
  }
  __syncthreads();

  // ---- Initialize (:59-146 / :1647-1735): row minima (+ threshold), one wave per row
  for (int a = wave; a < N; a += MM_WAVES) {
    const float *row = p.D + (size_t)a * N;
    float mv = INF;
#pragma unroll 8
    for (int l = lane; l < N; l += 64)  // (unrolled: eight loads in flight per lane, a row is 20 KB)
      if (l != a) mv = fminf(mv, row[l]);
    mv = wave_min_f(mv);
    if (p.CF) {
      const float *crow = p.CF + (size_t)a * N;
      float mc_ = INF;
#pragma unroll 8
      for (int l = lane; l < N; l += 64)
        if (l != a) mc_ = fminf(mc_, crow[l]);
      mc_ = wave_min_f(mc_);
      if (lane == 0) {
        const float old = p.min_values_CF[a];  // carried over from the previous build (:2399-2400)
        p.min_values_CF[a] = (old > mc_ ? mc_ : old) + threshold_CF;
      }
    }
    if (lane == 0) p.min_values[a] = mv + threshold;
  }
  __syncthreads();
  LAP(0);
  // mutually close pairs in (a, b) order: the waves test 16 rows at a time, thread 0 draws in order
  for (int base = 0; base < N; base += MM_WAVES) {
    const int a = base + wave;
    int cnt = 0;
    if (a < N) {
      const float mva = p.min_values[a];
      const float *row = p.D + (size_t)a * N;
      int *out = p.rowlist + (size_t)wave * N;
      for (int b0 = a + 1; b0 < N; b0 += 64) {
        const int b = b0 + lane;
        bool hit = false;
        if (b < N && mva >= row[b]) hit = p.min_values[b] >= DD(b, a);
        const unsigned long long m = __ballot(hit);
        if (hit) out[cnt + __popcll(m & ((1ull << lane) - 1ull))] = b;
        cnt += __popcll(m);
      }
    }
    if (lane == 0) sh.rowcount[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
      for (int w = 0; w < MM_WAVES && base + w < N; w++) {
        const int aa = base + w;
        const int *lst = p.rowlist + (size_t)w * N;
        for (int e = 0; e < sh.rowcount[w]; e++) {
          const int b = lst[e];
          consider(p, sh, aa, b);
          const float md = p.mc_dist[b], md2 = p.mc_dist2[b];
          if (sh.best.dist > md || (sh.best.dist == md && sh.best.dist2 > md2)) {
            sh.best.lin1 = aa;
            sh.best.lin2 = b;
            sh.best.dist = sh.sym_dist;
            sh.best.dist2 = md2;
          }
        }
      }
    }
    __syncthreads();
  }

  LAP(1);
  // ---- the merges
  for (int num_nodes = N; num_nodes < 2 * N - 1; num_nodes++) {
    const int n = sh.n;
    if (sh.best.dist == INF && !sh.use_sym) {
      // no mutually closest pair: from here on the symmetric matrix picks the pair when there is none
      // (initialize_sym, tree_builder.cpp:255-293): s(a,l) = d(a,l) + d(l,a) over the live clusters, row minima
      // with the first cluster that reaches them, the smallest of those (first row, first cluster)
      if (!p.SYM) {  // (the caller gave no room for it)
        if (tid == 0) *p.status = 1;
        return;
      }
      float bs = INF;
      int bs_pos = n;
      for (int ia = wave; ia < n; ia += MM_WAVES) {
        const int a = ci[ia];
        float mv = INF;
        int mp = n;
        for (int il = lane; il < n; il += 64) {
          const int l = ci[il];
          if (l == a) continue;
          const float v = DD(a, l) + DD(l, a);
          SS(a, l) = v;
          if (v < mv) {  // (ascending per lane: the first one stays)
            mv = v;
            mp = il;
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ov = __shfl_xor(mv, o, 64);
          const int op = __shfl_xor(mp, o, 64);
          if (ov < mv || (ov == mv && op < mp)) {
            mv = ov;
            mp = op;
          }
        }
        if (lane == 0) {
          p.min_values_sym[a] = mv;
          p.mcs_dist[a] = mv;
          if (mv < INF) {
            p.mcs_lin1[a] = a;
            p.mcs_lin2[a] = ci[mp];
          }
          if (mv < bs) {  // (rows ascending per wave)
            bs = mv;
            bs_pos = ia;
          }
        }
      }
      float bs2 = 0.0f;
      block_lex_min(bs, bs2, bs_pos, sh.wave_f, sh.wave_f2, sh.wave_i);
      if (tid == 0) {
        sh.use_sym = 1;
        sh.best_sym.dist = bs;
        if (bs_pos < n && bs < INF) {
          sh.best_sym.lin1 = p.mcs_lin1[ci[bs_pos]];
          sh.best_sym.lin2 = p.mcs_lin2[ci[bs_pos]];
        }
      }
      __syncthreads();
    }
    const bool by_sym = sh.best.dist == INF;
    const int i = by_sym ? sh.best_sym.lin1 : sh.best.lin1, j = by_sym ? sh.best_sym.lin2 : sh.best.lin2;
    const float csi = p.cluster_size[i], csj = p.cluster_size[j];
    const float added = csi + csj;
    if (tid == 0) {
      const int conv_i = p.convert_index[i], conv_j = p.convert_index[j];
      p.parent[conv_i] = num_nodes;
      p.parent[conv_j] = num_nodes;
      p.child_left[num_nodes - N] = conv_i;
      p.child_right[num_nodes - N] = conv_j;
      sh.count = 0;
    }
    __syncthreads();

    // -- phase 1: distance updates of both matrices, which rows rescan their minimum, which clusters rebuild
    float mv_cf = INF;
    // (four clusters per thread at a time, every load of the four issued before the first store: the loads walk
    //  down matrix columns, a new line each, and would otherwise wait for each other behind the stores)
    for (int base = tid; base < n; base += 4 * MM_BLOCK) {
      int kk[4];
      float c4[4][4], d4[4][4], mvk4[4];
      int l14[4], l24[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int ik = base + q * MM_BLOCK;
        kk[q] = ik < n ? ci[ik] : -1;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int k = kk[q];
        if (k < 0 || k == j || k == i) continue;
        if (p.CF) {
          c4[q][0] = CC(k, j);
          c4[q][1] = CC(k, i);
          c4[q][2] = CC(i, k);
          c4[q][3] = CC(j, k);
        }
        d4[q][0] = DD(k, j);
        d4[q][1] = DD(k, i);
        d4[q][2] = DD(i, k);
        d4[q][3] = DD(j, k);
        mvk4[q] = p.min_values[k];
        l14[q] = p.mc_lin1[k];
        l24[q] = p.mc_lin2[k];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int ik = base + q * MM_BLOCK;
        const int k = kk[q];
        if (k < 0) continue;
        p.kmask[ik] = 0;
        if (k == j || k == i) {
          p.kflag[ik] = 0;
          if (k == i) sh.ipos = ik;
          continue;
        }
        if (p.CF) {
          const float ckj = c4[q][0], cki = c4[q][1], cik = c4[q][2], cjk = c4[q][3];
          float njk = cjk;
          if (cik != cjk) {
            njk = (csi * cik + csj * cjk) / added;
            CC(j, k) = njk;
          }
          if (cki != ckj) CC(k, j) = (csi * cki + csj * ckj) / added;
          if (mv_cf > njk) mv_cf = njk;
        }
        const float dkj = d4[q][0], dki = d4[q][1], dik = d4[q][2], djk = d4[q][3];
        if (dik != djk) DD(j, k) = (csi * dik + csj * djk) / added;
        if (dki != dkj) DD(k, j) = (csi * dki + csj * dkj) / added;
        bool rescan = false;
        if (dkj != dki) {
          const float mvk = mvk4[q];
          rescan = (double)fabsf(mvk - threshold - dkj) < 1e-4 || (double)fabsf(mvk - threshold - dki) < 1e-4;
        }
        const int l1 = l14[q], l2 = l24[q];
        const bool touches = l1 == j || l2 == j || l1 == i || l2 == i;
        p.kflag[ik] = (unsigned char)(((rescan || touches) ? 2 : 0) | (rescan ? 4 : 0));
        if (rescan) p.visit_list[atomicAdd(&sh.count, 1)] = ik;  // (scratch use: rows to rescan, any order)
      }
    }
    if (p.CF) {
      const float m = block_min_f(mv_cf, sh.wave_f);
      if (tid == 0) p.min_values_CF[j] = m + threshold_CF;
    }
    __syncthreads();
    LAP(2);
    // row-minimum rescans (:1875-1890).  The reference scans row k in order and stops when the running minimum
    // equals the old one: the result is the old minimum if it occurs before any smaller entry, else the row's.
    const int nres = sh.count;
    for (int r = 0; r < nres; r++) {
      const int k = ci[p.visit_list[r]];
      const float old = p.min_values[k] - threshold;
      const float *row = p.D + (size_t)k * N;
      float fm = INF;
      int pos_old = n, pos_less = n;
#pragma unroll 4
      for (int il = tid; il < n; il += MM_BLOCK) {
        const int l = ci[il];
        if (l != i && l != k) {
          const float v = row[l];
          fm = fminf(fm, v);
          if (v == old) pos_old = min(pos_old, il);
          if (v < old) pos_less = min(pos_less, il);
        }
      }
      block_min3(fm, pos_old, pos_less, sh.wave_f, sh.wave_i, sh.wave_i2);
      if (tid == 0) p.min_values[k] = ((pos_old < n && pos_old < pos_less) ? old : fm) + threshold;
    }
    __syncthreads();
    LAP(3);

    // -- phase 1b: the rebuilt clusters in order; per later cluster, which of the first 32 it is a candidate of
    const int per = (n + MM_BLOCK - 1) / MM_BLOCK;
    const int lo = min(n, tid * per), hi = min(n, lo + per);
    int total;
    {
      int c = 0;
      for (int ik = lo; ik < hi; ik++) c += (p.kflag[ik] & 2) ? 1 : 0;
      int at = block_scan(c, &total, sh.wave_i);
      for (int ik = lo; ik < hi; ik++)
        if (p.kflag[ik] & 2) p.upd_pos[at++] = ik;
    }
    const int nupd = total;
    const int nu = min(nupd, MM_GATHER);
    __syncthreads();
    LAP(4);
    const int overflow_from = nupd > nu ? p.upd_pos[nu - 1] + 1 : n;
    const float *rowj = p.D + (size_t)j * N;
    float mvj = INF;
#pragma unroll 2
    for (int ik = tid; ik < n; ik += MM_BLOCK) {
      const int k = ci[ik];
      unsigned m = 0;
      for (int u = 0; u < nu; u++) {
        const int up = p.upd_pos[u];
        if (ik > up) {
          const int l = ci[up];
          if (DD(l, k) <= p.min_values[l]) m |= 1u << u;
        }
      }
      p.kmask[ik] = m;
      if (k != j && k != i) mvj = fminf(mvj, rowj[k]);
    }
    mvj = block_min_f(mvj, sh.wave_f);
    const float min_value_j = mvj + threshold;
    LAP(5);
    // best candidate among the clusters the ordered part does not visit: smallest (dist, dist2), earliest
    // cluster among exact ties
    float bd = INF, bd2 = INF;
    int bpos = n;
#pragma unroll 4
    for (int ik = tid; ik < n; ik += MM_BLOCK) {
      const int k = ci[ik];
      if (k == j || k == i) continue;
      const bool visited = (p.kflag[ik] & 2) || p.kmask[ik] != 0 || ik >= overflow_from;
      if (!visited) {
        const float d1 = p.mc_dist[k], d2 = p.mc_dist2[k];
        if (bd > d1 || (bd == d1 && bd2 > d2)) {  // (ascending positions per thread: the first one wins)
          bd = d1;
          bd2 = d2;
          bpos = ik;
        }
      }
    }
    block_lex_min(bd, bd2, bpos, sh.wave_f, sh.wave_f2, sh.wave_i);  // (dist, dist2, position) over the workgroup
    LAP(6);
    // lists in cluster order: the clusters the ordered part visits, the candidates of the merged cluster
    int nvisit, ncand;
    {
      unsigned vbits = 0, cbits = 0;  // per <= 32 positions per thread
      for (int ik = lo; ik < hi; ik++) {
        const int k = ci[ik];
        if (k == j || k == i) continue;
        if ((p.kflag[ik] & 2) || p.kmask[ik] != 0 || ik >= overflow_from) vbits |= 1u << (ik - lo);
        if (rowj[k] <= min_value_j && DD(k, j) <= p.min_values[k]) cbits |= 1u << (ik - lo);
      }
      int at, atc;
      block_scan2(__popc(vbits), __popc(cbits), at, atc, nvisit, ncand, sh.wave_i, sh.wave_i2);
      for (int ik = lo; ik < hi; ik++) {
        if (vbits & (1u << (ik - lo))) p.visit_list[at++] = ik;
        if (cbits & (1u << (ik - lo))) p.cand_j[atc++] = ci[ik];
      }
    }
    LAP(7);
    // feasible partners of every rebuilt cluster k: the clusters l before it with d(k,l) <= min_k and
    // d(l,k) <= min_l, in order (:1893-1911)
    int feas_total = 0;
    bool feas_overflow = false;
    for (int u = 0; u < nupd; u++) {
      const int up = p.upd_pos[u];
      const int k = ci[up];
      const float mvk = p.min_values[k];
      const float *rowk = p.D + (size_t)k * N;
      const int perk = (up + MM_BLOCK - 1) / MM_BLOCK;
      const int l0 = min(up, tid * perk), l1 = min(up, l0 + perk);
      unsigned bits = 0;  // perk <= 32 for N <= 32768
      for (int il = l0; il < l1; il++) {
        const int l = ci[il];
        if (rowk[l] <= mvk && l != j && l != i && DD(l, k) <= p.min_values[l]) bits |= 1u << (il - l0);
      }
      int tot;
      int at = feas_total + block_scan(__popc(bits), &tot, sh.wave_i3);
      if (feas_total + tot > p.feas_cap) {
        feas_overflow = true;
        break;
      }
      for (int il = l0; il < l1; il++)
        if (bits & (1u << (il - l0))) p.feas[at++] = ci[il];
      if (tid == 0) p.feas_off[u] = feas_total;
      feas_total += tot;
    }
    if (feas_overflow) {
      if (tid == 0) *p.status = 2;
      return;
    }
    if (tid == 0) p.feas_off[nupd] = feas_total;
    __syncthreads();
    LAP(8);

    // -- phase 2 (thread 0, in cluster order: it draws the random numbers)
    if (tid == 0) {
      // (the unvisited best is a copy taken before the visits: they may still change that cluster's candidate)
      const int bl1 = bpos < n ? p.mc_lin1[ci[bpos]] : -1;
      const int bl2 = bpos < n ? p.mc_lin2[ci[bpos]] : -1;
      float sd = INF, sd2 = INF;
      int sl1 = -1, sl2 = -1, spos = n;
      int ucs = 0;
      for (int v = 0; v < nvisit; v++) {
        const int ik = p.visit_list[v];
        const int k = ci[ik];
        const float mvk = p.min_values[k];
        if (p.kflag[ik] & 2) {
          const int u = ucs++;
          p.mc_dist[k] = INF;
          p.mc_dist2[k] = INF;
          for (int e = p.feas_off[u]; e < p.feas_off[u + 1]; e++) consider(p, sh, k, p.feas[e]);
        } else {
          const unsigned m = p.kmask[ik];
          for (int u = 0; u < ucs; u++) {
            const int l = ci[p.upd_pos[u]];
            if (u < nu) {
              if (((m >> u) & 1u) && DD(k, l) <= mvk) consider(p, sh, k, l);
            } else if (DD(k, l) <= mvk) {
              if (DD(l, k) <= p.min_values[l]) consider(p, sh, k, l);
            }
          }
        }
        const float d1 = p.mc_dist[k], d2 = p.mc_dist2[k];
        if (sd > d1 || (sd == d1 && sd2 > d2)) {
          sd = d1;
          sd2 = d2;
          sl1 = p.mc_lin1[k];
          sl2 = p.mc_lin2[k];
          spos = ik;
        }
      }
      // the loop's running "best": smallest (dist, dist2), the earliest cluster among exact ties
      Best b;
      b.dist = INF;
      b.dist2 = INF;
      b.lin1 = sh.best.lin1;
      b.lin2 = sh.best.lin2;
      int pos = n;
      if (bpos < n) {
        b.dist = bd;
        b.dist2 = bd2;
        b.lin1 = bl1;
        b.lin2 = bl2;
        pos = bpos;
      }
      if (spos < n) {
        if (b.dist > sd || (b.dist == sd && (b.dist2 > sd2 || (b.dist2 == sd2 && spos < pos)))) {
          b.dist = sd;
          b.dist2 = sd2;
          b.lin1 = sl1;
          b.lin2 = sl2;
          pos = spos;
        }
      }
      p.min_values[j] = min_value_j;
      p.mc_dist[j] = INF;
      p.mc_dist2[j] = INF;
      for (int e = 0; e < ncand; e++) consider(p, sh, p.cand_j[e], j);
      const float jd = p.mc_dist[j], jd2 = p.mc_dist2[j];
      if (b.dist > jd || (b.dist == jd && b.dist2 > jd2)) {
        b.dist = jd;
        b.dist2 = jd2;
        b.lin1 = p.mc_lin1[j];
        b.lin2 = p.mc_lin2[j];
      }
      sh.best = b;
      p.cluster_size[j] = csi + csj;
      p.convert_index[j] = num_nodes;
    }
    // -- the same merge in the symmetric matrix once it is in use (coalesce_sym, tree_builder.cpp:968-1058)
    if (sh.use_sym) {
      if (tid == 0) sh.count = 0;  // (thread 0 is past its ordered part; the others wait at the barrier below)
      __syncthreads();
      for (int ik = tid; ik < n; ik += MM_BLOCK) {
        const int k = ci[ik];
        if (k == j || k == i) continue;
        const float dkj = SS(k, j), dki = SS(k, i), dik = SS(i, k), djk = SS(j, k);
        const float mvk = p.min_values_sym[k];
        if (dik != djk) SS(j, k) = (csi * dik + csj * djk) / added;
        if (dki != dkj) SS(k, j) = (csi * dki + csj * dkj) / added;
        if (dkj != dki) {
          if ((double)fabsf(mvk - dkj) < 1e-6 || (double)fabsf(mvk - dki) < 1e-6)
            p.upd_pos[atomicAdd(&sh.count, 1)] = ik;  // (scratch use: rows to rescan)
        } else {
          if (p.mcs_lin1[k] == i) p.mcs_lin1[k] = j;
          if (p.mcs_lin2[k] == i) p.mcs_lin2[k] = j;
        }
      }
      __syncthreads();
      const int nres_s = sh.count;
      for (int r = 0; r < nres_s; r++) {
        const int k = ci[p.upd_pos[r]];
        const float old = p.min_values_sym[k];
        const float *row = p.SYM + (size_t)k * N;
        float fm = INF, fm2 = 0.0f;
        int fpos = n, pos_old = n, pos_less = n;
        for (int il = tid; il < n; il += MM_BLOCK) {
          const int l = ci[il];
          if (l != i && l != k) {
            const float v = row[l];
            if (v < fm) {
              fm = v;
              fpos = il;
            }
            if (v == old) pos_old = min(pos_old, il);
            if (v < old) pos_less = min(pos_less, il);
          }
        }
        float dummy = 0.0f;
        block_min3(dummy, pos_old, pos_less, sh.wave_f, sh.wave_i, sh.wave_i2);
        block_lex_min(fm, fm2, fpos, sh.wave_f, sh.wave_f2, sh.wave_i3);
        if (tid == 0) {
          const bool stops = pos_old < n && pos_old < pos_less;
          const float v = stops ? old : fm;
          const int at = stops ? pos_old : fpos;
          p.min_values_sym[k] = v;
          p.mcs_dist[k] = v;
          if (v < INF) {
            p.mcs_lin1[k] = k;
            p.mcs_lin2[k] = ci[at];
          }
        }
      }
      __syncthreads();
      float b1 = INF, b2 = 0.0f, mj = INF, mj2 = 0.0f;
      int bp = n, mjp = n;
      const float *srowj = p.SYM + (size_t)j * N;
      for (int ik = tid; ik < n; ik += MM_BLOCK) {
        const int k = ci[ik];
        if (k == j || k == i) continue;
        const float dk = p.mcs_dist[k];
        if (dk < b1) {
          b1 = dk;
          bp = ik;
        }
        const float sj = srowj[k];
        if (sj < mj) {
          mj = sj;
          mjp = ik;
        }
      }
      block_lex_min(b1, b2, bp, sh.wave_f, sh.wave_f2, sh.wave_i);
      block_lex_min(mj, mj2, mjp, sh.wave_f, sh.wave_f2, sh.wave_i3);
      if (tid == 0) {
        Best b = sh.best_sym;
        b.dist = INF;
        if (bp < n && b1 < INF) {
          const int k = ci[bp];
          b.dist = b1;
          b.lin1 = p.mcs_lin1[k];
          b.lin2 = p.mcs_lin2[k];
        }
        p.min_values_sym[j] = mj;
        p.mcs_dist[j] = mj;  // (INF when nothing is left)
        if (mjp < n && mj < INF) {
          p.mcs_lin1[j] = ci[mjp];
          p.mcs_lin2[j] = j;
        }
        if (b.dist > mj) {
          b.dist = mj;
          b.lin1 = p.mcs_lin1[j];
          b.lin2 = p.mcs_lin2[j];
        }
        sh.best_sym = b;
      }
    }
    LAP(9);
    // -- the merged-away cluster leaves the list (copied to the other buffer, one position up behind it)
    {
      const int ipos = sh.ipos;
      for (int ik = tid; ik < n - 1; ik += MM_BLOCK) ci_next[ik] = ci[ik >= ipos ? ik + 1 : ik];
      int *t = ci;
      ci = ci_next;
      ci_next = t;
    }
    if (tid == 0) sh.n = n - 1;
    __syncthreads();
    LAP(10);
  }
  if (tid == 0) {
    *p.status = 0;
    if (p.timers) {
      tacc[11] = (clock64() - cstart) * 100 / (wall_clock64() - tstart + 1);  // shader clock, MHz
      for (int x = 0; x < 12; x++) p.timers[x] = tacc[x];
    }
  }
}

// Carrier penalty of AncesTreeBuilder::BuildTopology (anc_builder.cpp:563-581) on the device: every entry of a
// carrier's row gets + val, and - val again where the column is a carrier too (the same two operations in
// the same order per entry as the host loop).
__global__ void penalty_kernel(float *__restrict__ D, int N, const unsigned char *__restrict__ member, float val) {
  const int c = blockIdx.x;
  if (!member[c]) return;
  float *row = D + (size_t)c * N;
  for (int col = threadIdx.x; col < N; col += blockDim.x) {
    float x = row[col] + val;
    if (member[col]) x -= val;
    row[col] = x;
  }
}

// Clade prior from the previous tree (treeseq.cpp: clade_prior; anc_builder.cpp:583-606): row a holds, for
// every other leaf b, acc[depth(parent(a)) - depth(mrca(a,b))]; the leaves under the sibling of every ancestor
// step are a contiguous range of the depth-first leaf order.
__global__ void prior_kernel(float *__restrict__ CF, int N, const int *__restrict__ parent,
                             const int *__restrict__ child_left, const int *__restrict__ child_right,
                             const int *__restrict__ depth, const int *__restrict__ lo,
                             const int *__restrict__ size, const int *__restrict__ order,
                             const float *__restrict__ acc) {
  const int a = blockIdx.x;
  float *row = CF + (size_t)a * N;
  if (threadIdx.x == 0) row[a] = 0.0f;
  const int da = depth[parent[a]];
  int child = a;
  for (int v = parent[a]; v >= 0; child = v, v = parent[v]) {
    const int other = child_left[v] == child ? child_right[v] : child_left[v];
    const float x = acc[da - depth[v]];
    const int b = lo[other], e = b + size[other];
    for (int q = b + threadIdx.x; q < e; q += blockDim.x) row[order[q]] = x;
  }
}

}  // namespace

// The restated generator against the library's, on the host: how many of n draws differ (0 expected).
int rng_restatement_mismatches(unsigned seed, int n) {
  Rng r;
  rng_seed(r, seed);
  std::mt19937 ref(seed);
  std::uniform_real_distribution<double> unif(0.0, 1.0);
  int bad = 0;
  for (int i = 0; i < n; i++) bad += rng_unif(r) != unif(ref);
  return bad;
}

// ---- host side
// Trees of different sections are built at the same time, but a process has a handful of hardware queues (4 by
// default): 40 builders with a stream and a one-workgroup launch each run 4 at a time.  So builders hand their
// request to a dispatcher of the device, which puts all that are waiting into one launch -- a workgroup each --
// and keeps up to three launches in flight.
class BuildDispatcher {
 public:
  struct Request {
    MMParams p;
    bool done = false;
    int rc = 0;
  };
  static BuildDispatcher &of(int device) {
    static std::mutex gm;
    static std::vector<BuildDispatcher *> all;
    std::lock_guard<std::mutex> lk(gm);
    if ((int)all.size() <= device) all.resize(device + 1, nullptr);
    if (!all[device]) all[device] = new BuildDispatcher(device);  // lives as long as the process
    return *all[device];
  }
  void enroll(int delta) {
    std::lock_guard<std::mutex> lk(m_);
    builders_ += delta;
  }
  int run(Request &r) {
    std::unique_lock<std::mutex> lk(m_);
    pending_.push_back(&r);
    cv_work_.notify_one();
    cv_done_.wait(lk, [&] { return r.done; });
    return r.rc;
  }

 private:
  explicit BuildDispatcher(int device) : device_(device) {
    for (int t = 0; t < 3; t++) workers_.emplace_back([this] { worker(); });
    for (auto &w : workers_) w.detach();
  }
  void worker() {
    (void)hipSetDevice(device_);
    hipStream_t stream = nullptr;
    (void)hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    DevBuf d_params;
    std::vector<Request *> batch;
    std::vector<MMParams> params;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [&] { return !pending_.empty(); });
        // The builders that are not being served right now are about to ask too (they come in bursts, after a
        // launch completes and their hosts have prepared the next matrices): wait for them a little, a launch
        // takes as long as its slowest tree however many it carries.
        const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(40);
        while ((int)pending_.size() < builders_ - inflight_ && std::chrono::steady_clock::now() < until) {
          lk.unlock();
          std::this_thread::sleep_for(std::chrono::milliseconds(1));
          lk.lock();
        }
        batch.swap(pending_);
        inflight_ += (int)batch.size();
      }
      if (batch.empty()) continue;
      params.clear();
      for (Request *r : batch) params.push_back(r->p);
      int rc = d_params.alloc(params.size() * sizeof(MMParams));
      if (!rc && hipMemcpyAsync(d_params.p, params.data(), params.size() * sizeof(MMParams), hipMemcpyHostToDevice,
                                stream) != hipSuccess)
        rc = RL_EHIP;
      if (!rc) {
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(minmatch_kernel, dim3((unsigned)batch.size()), dim3(MM_BLOCK), 0, stream,
                           d_params.as<MMParams>());
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) rc = RL_EHIP;
        if (getenv("RELATE_AMD_TIMING"))
          fprintf(stderr, "[tree builder launch] %zu trees, %.1f ms\n", batch.size(),
                  1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      }
      {
        std::lock_guard<std::mutex> lk(m_);
        for (Request *r : batch) {
          r->rc = rc;
          r->done = true;
        }
        inflight_ -= (int)batch.size();
      }
      cv_done_.notify_all();
      batch.clear();
    }
  }
  int device_;
  int builders_ = 0, inflight_ = 0;  // builders alive on this device; requests in a launch
  std::mutex m_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<Request *> pending_;
  std::vector<std::thread> workers_;
};

struct DeviceMinMatch::Impl {
  int N = 0, device = 0;
  hipStream_t stream = nullptr;
  DevBuf d_D, d_CF, d_SYM, d_f, d_i, d_feas, d_rowlist, d_status, d_flags, d_member, d_tab, d_acc;
  long long feas_cap = 0;
};

DeviceMinMatch::DeviceMinMatch(int N, int device) : impl(new Impl()) {
  impl->N = N;
  impl->device = device;
  BuildDispatcher::of(device).enroll(1);
}
DeviceMinMatch::~DeviceMinMatch() {
  BuildDispatcher::of(impl->device).enroll(-1);
  if (impl->stream) (void)hipStreamDestroy(impl->stream);
  delete impl;
}

float *DeviceMinMatch::device_matrix() {
  Impl &m = *impl;
  if (hipSetDevice(m.device) != hipSuccess) return nullptr;
  if (m.d_D.alloc((size_t)m.N * m.N * 4)) return nullptr;
  return m.d_D.as<float>();
}

int DeviceMinMatch::apply_penalty(const char *member, float val) {
  Impl &m = *impl;
  const int N = m.N;
  RL_HIP(hipSetDevice(m.device));
  if (!m.stream) RL_HIP(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
  if (m.d_member.alloc((size_t)N)) return -1;
  RL_HIP(hipMemcpyAsync(m.d_member.p, member, (size_t)N, hipMemcpyHostToDevice, m.stream));
  hipLaunchKernelGGL(penalty_kernel, dim3(N), dim3(256), 0, m.stream, m.d_D.as<float>(), N,
                     m.d_member.as<unsigned char>(), val);
  RL_HIP(hipGetLastError());
  return 0;
}

int DeviceMinMatch::apply_prior(const HostTree &t, float val) {
  Impl &m = *impl;
  const int N = m.N, T = 2 * N - 1;
  RL_HIP(hipSetDevice(m.device));
  if (!m.stream) RL_HIP(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
  // the same tables as the host's clade_prior (treeseq.cpp)
  std::vector<int> tab((size_t)6 * T + N);
  int *parent = tab.data(), *cl = parent + T, *cr = cl + T, *depth = cr + T, *lo = depth + T, *size = lo + T,
      *order = size + T;
  for (int v = 0; v < T; v++) {
    parent[v] = t.parent[v];
    cl[v] = t.child_left[v];
    cr[v] = t.child_right[v];
    depth[v] = 0;
    lo[v] = 0;
    size[v] = 1;
  }
  for (int v = T - 1; v >= N; v--) depth[v] = (parent[v] >= 0 ? depth[parent[v]] : 0) + 1;
  for (int v = N; v < T; v++) size[v] = size[cl[v]] + size[cr[v]];
  for (int v = T - 1; v >= N; v--) {
    lo[cl[v]] = lo[v];
    lo[cr[v]] = lo[v] + size[cl[v]];
  }
  for (int i = 0; i < N; i++) order[lo[i]] = i;
  std::vector<float> acc((size_t)N + 1, 0.0f);
  for (int c = 1; c <= N; c++) acc[c] = acc[c - 1] + val;
  int rc = m.d_tab.alloc(tab.size() * 4);
  rc = rc ? rc : m.d_acc.alloc(acc.size() * 4);
  rc = rc ? rc : m.d_CF.alloc((size_t)N * N * 4);
  if (rc) return -1;
  RL_HIP(hipMemcpyAsync(m.d_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, m.stream));
  RL_HIP(hipMemcpyAsync(m.d_acc.p, acc.data(), acc.size() * 4, hipMemcpyHostToDevice, m.stream));
  const int *q = m.d_tab.as<int>();
  hipLaunchKernelGGL(prior_kernel, dim3(N), dim3(256), 0, m.stream, m.d_CF.as<float>(), N, q, q + T, q + 2 * (size_t)T,
                     q + 3 * (size_t)T, q + 4 * (size_t)T, q + 5 * (size_t)T, q + 6 * (size_t)T, m.d_acc.as<float>());
  RL_HIP(hipGetLastError());
  RL_HIP(hipStreamSynchronize(m.stream));  // (tab and acc are locals)
  return 0;
}

int DeviceMinMatch::build_resident(MinMatch &tb, bool with_prior, HostTree &tree) {
  return build_impl(tb, nullptr, nullptr, true, with_prior, tree);
}

int DeviceMinMatch::build(MinMatch &tb, const float *d, const float *prior, HostTree &tree) {
  return build_impl(tb, d, prior, false, prior != nullptr, tree);
}

int DeviceMinMatch::build_impl(MinMatch &tb, const float *d, const float *prior_host, bool resident, bool with_prior,
                               HostTree &tree) {
  Impl &m = *impl;
  const int N = m.N;
  if (N < 2 || N > 32768) return 1;
  RL_HIP(hipSetDevice(m.device));
  if (!m.stream) RL_HIP(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
  const bool prior = with_prior;
  const size_t NN = (size_t)N * N;
  m.feas_cap = (long long)8 * N;
  int rc = m.d_D.alloc(NN * 4);
  rc = rc ? rc : (prior ? m.d_CF.alloc(NN * 4) : 0);
  rc = rc ? rc : m.d_SYM.alloc(NN * 4);
  rc = rc ? rc : m.d_f.alloc((size_t)7 * N * 4);  // min_values, min_values_CF, mc_dist, mc_dist2, cluster_size, + 2 sym
  rc = rc ? rc : m.d_i.alloc(((size_t)9 * N + (size_t)7 * N + 8) * 4);  // ints, see below
  rc = rc ? rc : m.d_feas.alloc((size_t)m.feas_cap * 4);
  rc = rc ? rc : m.d_rowlist.alloc((size_t)MM_WAVES * N * 4);
  rc = rc ? rc : m.d_status.alloc(16 + 12 * 8);
  rc = rc ? rc : m.d_flags.alloc((size_t)N);
  if (rc) return -1;
  MMParams p;
  p.N = N;
  p.threshold = tb.threshold;
  p.threshold_CF = tb.threshold_CF;
  p.D = m.d_D.as<float>();
  p.CF = prior ? m.d_CF.as<float>() : nullptr;
  float *f = m.d_f.as<float>();
  p.min_values = f;
  p.min_values_CF = f + N;
  p.mc_dist = f + 2 * (size_t)N;
  p.mc_dist2 = f + 3 * (size_t)N;
  p.cluster_size = f + 4 * (size_t)N;
  p.min_values_sym = f + 5 * (size_t)N;
  p.mcs_dist = f + 6 * (size_t)N;
  p.SYM = m.d_SYM.as<float>();
  int *q = m.d_i.as<int>();
  p.mc_lin1 = q;
  p.mc_lin2 = q + N;
  p.cluster_index = q + 2 * (size_t)N;
  p.convert_index = q + 3 * (size_t)N;
  p.kmask = reinterpret_cast<unsigned *>(q + 4 * (size_t)N);
  p.visit_list = q + 5 * (size_t)N;
  p.cand_j = q + 6 * (size_t)N;
  p.upd_pos = q + 7 * (size_t)N;
  p.feas_off = q + 8 * (size_t)N;  // [N+1]
  p.parent = q + 9 * (size_t)N + 4;         // [2N-1]
  p.child_left = q + 11 * (size_t)N + 4;    // [N-1]
  p.child_right = q + 12 * (size_t)N + 4;   // [N-1]
  p.cluster_index2 = q + 13 * (size_t)N + 4;
  p.mcs_lin1 = q + 14 * (size_t)N + 4;
  p.mcs_lin2 = q + 15 * (size_t)N + 4;
  p.kflag = m.d_flags.as<unsigned char>();
  p.feas = m.d_feas.as<int>();
  p.feas_cap = m.feas_cap;
  p.rowlist = m.d_rowlist.as<int>();
  p.status = m.d_status.as<int>();
  const bool timing = getenv("RELATE_AMD_TIMING") != nullptr;
  p.timers = timing ? reinterpret_cast<long long *>(m.d_status.as<char>() + 16) : nullptr;

  // state the builders carry from tree to tree: in
  std::vector<int> lin((size_t)2 * N);
  for (int c = 0; c < N; c++) {
    lin[c] = tb.mc[c].lin1;
    lin[(size_t)N + c] = tb.mc[c].lin2;
  }
  RL_HIP(hipMemcpyAsync(p.mc_lin1, lin.data(), (size_t)2 * N * 4, hipMemcpyHostToDevice, m.stream));
  RL_HIP(hipMemcpyAsync(p.min_values_CF, tb.min_values_CF.data(), (size_t)N * 4, hipMemcpyHostToDevice, m.stream));
  if (!resident) {
    RL_HIP(hipMemcpyAsync(p.D, d, NN * 4, hipMemcpyHostToDevice, m.stream));
    if (prior) RL_HIP(hipMemcpyAsync(p.CF, prior_host, NN * 4, hipMemcpyHostToDevice, m.stream));
  }
  const int minus1 = -1;
  RL_HIP(hipMemcpyAsync(p.status, &minus1, 4, hipMemcpyHostToDevice, m.stream));
  RL_HIP(hipStreamSynchronize(m.stream));  // inputs in place
  {
    BuildDispatcher::Request req;
    req.p = p;
    if (BuildDispatcher::of(m.device).run(req)) {
      set_error("tree builder launch failed");
      return -1;
    }
  }
  int status = -1;
  RL_HIP(hipMemcpyAsync(&status, p.status, 4, hipMemcpyDeviceToHost, m.stream));
  RL_HIP(hipStreamSynchronize(m.stream));
  if (status != 0) return status > 0 ? status : -1;
  if (timing) {
    long long tk[12];
    RL_HIP(hipMemcpy(tk, p.timers, sizeof(tk), hipMemcpyDeviceToHost));
    fprintf(stderr, "[gpu tree builder] N=%d, us:", N);
    static const char *names[12] = {"row minima", "pair scan", "updates", "rescans", "rebuilt list", "masks+min_j",
                                    "unvisited best", "lists", "partners", "ordered", "erase", ""};
    for (int x = 0; x < 11; x++) fprintf(stderr, " %s %.0f", names[x], tk[x] / 100.0);
    fprintf(stderr, " shader_MHz %lld\n", tk[11]);
  }
  // out: the tree and the carried state
  std::vector<int> tr((size_t)4 * N);
  RL_HIP(hipMemcpyAsync(tr.data(), p.parent, ((size_t)4 * N - 1) * 4, hipMemcpyDeviceToHost, m.stream));
  RL_HIP(hipMemcpyAsync(lin.data(), p.mc_lin1, (size_t)2 * N * 4, hipMemcpyDeviceToHost, m.stream));
  RL_HIP(hipMemcpyAsync(tb.min_values_CF.data(), p.min_values_CF, (size_t)N * 4, hipMemcpyDeviceToHost, m.stream));
  RL_HIP(hipStreamSynchronize(m.stream));
  tree.reset(N);
  for (int c = 0; c < 2 * N - 1; c++) tree.parent[c] = tr[c];
  for (int c = 0; c < N - 1; c++) {
    tree.child_left[N + c] = tr[(size_t)2 * N + c];
    tree.child_right[N + c] = tr[(size_t)3 * N + c];
  }
  for (int c = 0; c < N; c++) {
    tb.mc[c].lin1 = lin[c];
    tb.mc[c].lin2 = lin[(size_t)N + c];
  }
  return 0;
}

}  // namespace rl

// ---- C ABI: a tree builder that keeps MinMatch's state from tree to tree
struct rl_builder {
  int N;
  rl::MinMatch tb;
  rl::DeviceMinMatch *dev;
  int last_on_gpu;
  rl_builder(int n, double theta, int device)
      : N(n), tb(n, theta), dev(device >= 0 ? new rl::DeviceMinMatch(n, device) : nullptr), last_on_gpu(0) {}
  ~rl_builder() { delete dev; }
};

extern "C" {

rl_builder *rl_builder_create(int N, double theta, int device) {
  if (N < 2 || !(theta > 0.0 && theta < 1.0)) {
    rl::set_error("rl_builder_create: bad arguments");
    return nullptr;
  }
  return new rl_builder(N, theta, device);
}

void rl_builder_destroy(rl_builder *b) { delete b; }

int rl_builder_build(rl_builder *b, float *d, const float *d_prior, int *parent, int *child_left, int *child_right) {
  if (!b || !d || !parent) {
    rl::set_error("rl_builder_build: bad arguments");
    return RL_EINVAL;
  }
  rl::HostTree t;
  int st = 1;
  if (b->dev) {
    st = b->dev->build(b->tb, d, d_prior, t);
    if (st < 0) return RL_EHIP;
  }
  b->last_on_gpu = st == 0;
  if (st != 0) b->tb.quick_build(d, d_prior, t);
  const int N = b->N;
  for (int i = 0; i < 2 * N - 1; i++) parent[i] = t.parent[i];
  for (int i = N; i < 2 * N - 1; i++) {
    if (child_left) child_left[i - N] = t.child_left[i];
    if (child_right) child_right[i - N] = t.child_right[i];
  }
  return RL_OK;
}

int rl_builder_last_on_gpu(const rl_builder *b) { return b ? b->last_on_gpu : RL_EINVAL; }

int rl_debug_rng_mismatches(unsigned seed, int n) { return rl::rng_restatement_mismatches(seed, n); }

}  // extern "C"
