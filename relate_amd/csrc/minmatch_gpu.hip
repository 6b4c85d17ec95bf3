// minmatch_gpu.hip -- MinMatch::QuickBuild on the GPU: one workgroup builds one tree, matrices in HBM.
//
// Reference: src/tree_builder.cpp (MinMatch, sample_ages empty): :59-146 / :1647-1735 (Initialize),
// :296-598 / :1844-2070 (Coalesce), :1061-1303 / :2358-2644 (QuickBuild).  The host builder
// (minmatch.cpp) states the algorithm and why a merge splits into a parallel part and an ordered part with
// identical results; this file is the same split on one workgroup:
//   * the N-1 merges of a tree are sequential and each walks down two columns of a 100 MB matrix -- a new
//     line per cluster: on a host dozens of open sections are bound by DRAM (DESIGN_NOTES.md 5), in HBM one
//     workgroup per tree leaves the other 255 CUs to the trees of the other sections;
//   * the parts that are order-free (distance updates, row-minimum rescans, candidate tests, reductions) run
//     on all 512 threads; the random draws -- one per feasible pair, in the reference's order -- and the
//     candidate bookkeeping run on wave 0 over lists the parallel parts leave in order;
//   * std::mt19937 (seed 1 per build) and libstdc++'s generate_canonical<double, 53> are restated below.
// The symmetric fallback (no mutually closest pair left, :255-293 / :968-1058) is here too: "the first cluster in
// scan order that reaches the minimum" is a lexicographic (value, position) reduction.  The state both builders
// carry from tree to tree (min_values_CF and the stale candidate indices, minmatch.h) is copied in before and
// out after every build, so host and device builders can alternate on a section.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <type_traits>
#include <vector>

#include "common.h"
#include "minmatch.h"

namespace rl {

namespace {

constexpr int MM_BLOCK = 512;   // (8 waves; a thread keeps ~10 clusters of a merge in registers.  256 threads: 166 ms per N = 5000 tree, 1024: 156, 512: 110)
constexpr int MM_WAVES = MM_BLOCK / 64;
constexpr int MM_HITS = 64;       // mutually close pairs per row the weave keeps; more: the host builds the tree
constexpr int MM_UPD_MAX = 512;   // rebuilt clusters of one merge; more: the host builds the tree (the AGES build keeps a longer list)
constexpr int MM_UPD_LDS = 256;   // ... of which this many in LDS, the rest in global memory
constexpr int MM_PAIRS_LDS = 320; // feasible pairs of one merge kept in LDS (8 per merge on average); more go through global scratch
constexpr int MM_BUCKET_MIN = 2048;  // (the AGES build) more pairs than this are put in order bucket by bucket
// M is stored in column panels of 64: element (a, b) at ((b / 64) * N + a) * 64 + b % 64.  A row is N / 64 runs of
// 1 KB (a wavefront's 64 consecutive clusters: one run), 64 * N elements apart; a COLUMN -- what a merge scatters its
// one store per cluster down -- strides by 1 KB inside one panel of N KB, a few 2 MB pages, instead of by a whole
// row (80 KB at N = 5000: every store on another page, the address translation of 2500 pages per merge).
// (Rows in blocks of 128, so that a row spans fewer pages: no gain, DESIGN_NOTES.md 5.)
constexpr int MM_PANEL = 64;  // (narrower panels -- a column's stores 128 / 256 / 512 B apart instead of 1 KB -- are no faster:
                              //  118.1 / 115.5 / 114.9 ms per tree against 114.7, profiles/r06_panel_width.json)
__host__ __device__ inline unsigned mm_index(unsigned a, unsigned b, unsigned N) {
  return ((b / MM_PANEL) * N + a) * MM_PANEL + (b % MM_PANEL);
}
__host__ __device__ inline size_t mm_elements(size_t N) { return ((N + MM_PANEL - 1) / MM_PANEL) * N * MM_PANEL; }
constexpr int MM_MAXN = 10240;    // one thread holds up to 20 clusters of a merge in registers
constexpr int MM_Q_LDS = 5120 / MM_BLOCK, MM_Q_GLOB = MM_MAXN / MM_BLOCK;  // register slots per thread: N <= 5120 / N <= 10240
constexpr int MM_Q_SMALL = 2048 / MM_BLOCK;                                // ... and N <= 2048
// Where the per-cluster state of a build lives:
//   L_HOT    (the plain build, every N): what a merge reads or writes for EVERY cluster or on the ordered part's path --
//            the candidate (dist, dist2, lin1, lin2) and the rebuilt mark, 13 bytes per cluster -- in LDS; the rest
//            (row minima of both matrices, sizes, the live list) in global memory, read with the rows of a merge or
//            kept in registers.  65 KB at N = 5000: TWO workgroups per CU, 130 KB at N = 10,000;
//   L_LDS    (the AGES build, N <= 4100) everything in LDS, 33 bytes per cluster;
//   L_GLOBAL (the AGES build above that) everything in the global arrays of MMParams.
//   L_WARM   L_HOT with the row minima of both matrices in LDS too (21 bytes per cluster): one workgroup per CU.
enum { L_LDS = 0, L_GLOBAL = 1, L_HOT = 2, L_WARM = 3 };

// A tree's parameters.  The matrices of a build, woven: M[a][b] = (d(a,b), d(b,a), cf(a,b), cf(b,a)).  A merge needs,
// per cluster k, the entries (i,k), (k,i), (j,k), (k,j) of both matrices: two 16-byte loads along the rows of i and j
// instead of four loads along rows and four down columns (a scattered 4-byte access costs one CU ~3.5 cycles:
// measured 3.7 us per column of 2500 clusters, six of them per merge in the plain layout); what stays scattered is
// ONE 16-byte store per cluster, M[k][j], which waits for nobody.  Packed from the distance matrix (K3 + carrier
// penalty) and the clade prior by weave_kernel; row minima of both by the penalty / prior passes (rowmin_penalty_kernel,
// prior_kernel).
//
// One field list, two structs of the same layout: MMParams (plain pointers: what the host fills in) and MMParamsDev
// (GLOBAL pointers: what the build kernel works with).  A worker reads its parameters from a queue, not from
// kernel arguments, and a pointer of unknown address space compiles to flat_load / flat_store, which count against
// the LDS counter too and turn every wait for an LDS read into a wait for the loads in flight -- the phases of a
// merge overlap exactly these two.
#define MM_PARAM_FIELDS(PTR)                                                                                         \
  int N;                                                                                                             \
  int layout; /* L_LDS / L_GLOBAL / L_HOT: where the per-cluster state lives for the build */                       \
  float threshold, threshold_CF;                                                                                     \
  PTR(MM_F4) M; /* [N*N] destroyed */                                                                                \
  int has_prior; /* the cf halves of M are in use */                                                                 \
  PTR(const float) rowmin_D; /* [N] minimum of each row off the diagonal */                                          \
  PTR(const float) rowmin_CF;                                                                                        \
  /* the mutually close pairs (a, b > a) of the untouched matrix, row by row in order, from weave_kernel   */       \
  PTR(const int) hit_cnt;    /* [N] (more than MM_HITS: not all kept) */                                             \
  PTR(const unsigned) hit_b; /* [N][MM_HITS] */                                                                      \
  PTR(const float) hit_sym;  /* [N][MM_HITS] symmetric distance of the pair (0: the prior makes it a certain pair) */ \
  /* room for the symmetric matrix of the fallback (rare): a pool of the device, `sym_slots` matrices of N*N floats \
     at SYM, taken with a compare-and-swap on sym_locks[slot] when a tree first needs one and given back when it    \
     is out; none free (or no pool): the tree is the host's */                                                       \
  PTR(float) SYM;                                                                                                    \
  PTR(int) sym_locks;                                                                                                \
  int sym_slots;                                                                                                     \
  PTR(float) min_values_sym;                                                                                         \
  PTR(float) mcs_dist;                                                                                               \
  PTR(int) mcs_lin1;                                                                                                 \
  PTR(int) mcs_lin2;                                                                                                 \
  PTR(float) min_values; /* min_values_CF, mc_lin1, mc_lin2: carried from tree to tree (in and out) */               \
  PTR(float) min_values_CF;                                                                                          \
  /* (pinned host memory; L_LDS and L_HOT) where the worker reads the carried state and leaves it again: candidate  \
     indices [2N], min_values_CF [N] -- L_GLOBAL: the host copies to and from the arrays above */                    \
  PTR(int) io_lin;                                                                                                   \
  PTR(float) io_mvcf;                                                                                                \
  PTR(float) mc_dist;                                                                                                \
  PTR(float) mc_dist2;                                                                                               \
  PTR(int) mc_lin1;                                                                                                  \
  PTR(int) mc_lin2;                                                                                                  \
  PTR(int) cluster_index;                                                                                            \
  PTR(int) cluster_size;                                                                                             \
  PTR(unsigned char) kflag;                                                                                          \
  PTR(int) upd_pos;     /* [N] scratch of the symmetric path */                                                      \
  PTR(unsigned) pair_g; /* [6 * pair_cap] feasible pairs of a merge beyond MM_PAIRS_LDS: key, x<<16|y, sym */        \
  PTR(int) rowlist;     /* [MM_WAVES][N] pair-scan survivors per wave */                                             \
  PTR(int) merge_i;     /* [N-1] the merges as (cluster i, cluster j) in order: the host names the nodes from them */ \
  PTR(int) merge_j;                                                                                                  \
  PTR(int) status;                                                                                                   \
  /* (pinned host memory) the status again, stored when everything else of the tree is out and flushed: the         \
     builder's host thread watches it and takes its tree while the other workers still build theirs */              \
  PTR(volatile int) host_done;                                                                                       \
  long long pair_cap;                                                                                                \
  PTR(long long) timers; /* optional: 100 MHz ticks per phase (RELATE_AMD_TIMING) */                                 \
  PTR(unsigned) trace;   /* optional (RELATE_AMD_TIMING=2): [0] merge, [1] phase the workgroup has reached */     \
  /* --sample_ages (the AGES build): the candidates' draw as a double and their third key as the INDEX of the age   \
     among the sorted distinct sample ages (MM_AGE_EMPTY: no candidate); per cluster the index of its oldest sample \
     (age_lvl0: as the samples are, age_lvl: the build's own copy); the distinct ages, how many samples have each */ \
  PTR(double) mc_dist2d;                                                                                             \
  PTR(int) mc_lvl;                                                                                                   \
  PTR(int) age_lvl;                                                                                                  \
  PTR(const int) age_lvl0;                                                                                           \
  PTR(const double) unique_ages;                                                                                     \
  PTR(const int) ages_count;                                                                                         \
  int n_levels;                                                                                                      \
  int Ne; /* pipeline/BuildTopology.cpp:36 */                                                                        \
  /* the rebuilt clusters of a merge beyond MM_UPD_LDS (with sample ages a renamed candidate follows its lineage up  \
     the tree, :2342, and a merge of that lineage sends every cluster holding one through the rebuilding branch) */  \
  PTR(unsigned) upd_g; /* [N] */                                                                                     \
  PTR(float) updv_g;   /* [N] */                                                                                     \
  /* ... and, for a merge with thousands of feasible pairs, the pairs per later cluster: counts / fill marks and    \
     offsets */                                                                                                      \
  PTR(int) bucket;     /* [N + 1] */                                                                                 \
  PTR(int) bucket_off; /* [N + 1] */
#define MM_HOST_PTR(T) T *
#define MM_GLOBAL_PTR(T) __attribute__((address_space(1))) T *
// (a class type -- HIP's float4 -- has no member functions outside the generic address space: the device struct
//  takes the elements of M as native vectors)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MM_F4 float4
struct MMParams {
  MM_PARAM_FIELDS(MM_HOST_PTR)
};
#undef MM_F4
#define MM_F4 f32x4
struct MMParamsDev {
  MM_PARAM_FIELDS(MM_GLOBAL_PTR)
};
#undef MM_F4
static_assert(sizeof(MMParams) == sizeof(MMParamsDev), "one layout");

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
// The state renewal of rng_next by one wavefront: word i takes the old words i, i+1 and word i+397 -- old for
// i < 227, already renewed (word i-227) behind that; 64 consecutive words per round, every lane reads before any
// lane writes, and a round never needs a word of its own round (227 > 64).  The last word takes the new word 0.
// (The lanes exchange words through LDS between rounds: to the compiler one thread's next-round loads are
// provably other words than its store and may move above it -- the fences pin the rounds.)
__device__ __noinline__ void rng_renew_wave(Rng &r, int lane) {
  for (int base = 0; base < 623; base += 64) {
    const int i = base + lane;
    uint32_t v = 0;
    if (i < 623) {
      const uint32_t y = (r.mt[i] & 0x80000000u) | (r.mt[i + 1] & 0x7fffffffu);
      v = r.mt[i < 227 ? i + 397 : i - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (i < 623) r.mt[i] = v;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    const uint32_t y = (r.mt[623] & 0x80000000u) | (r.mt[0] & 0x7fffffffu);
    r.mt[623] = r.mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
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

constexpr int MM_PARAM_WORDS = 104;  // sizeof(MMParams) / 4 rounded up (static_assert below)
constexpr int MM_ROWS_MAX = 2;       // rows of rebuilt clusters a pass of a merge scans at most (build_tree: ROWS; 3: 119 ms per N = 5000 tree, 2: 110, 4: 134)
struct Shared {
  unsigned praw[MM_PARAM_WORDS];  // the tree's parameters as the worker read them from the queue
  unsigned ticket;
  int sym_slot;
  long long tacc[16], tmark;
  int cnt_rescan_only;  // (statistics, RELATE_AMD_TIMING) rebuilt clusters of the merge whose candidate does not touch i or j
  Rng rng;
  Best best, best_sym;
  int use_sym;
  int n, ipos;
  int count;
  int nupd, npairs;
  int wave_i[MM_WAVES];
  float wave_f[MM_WAVES];
  int wave_i2[MM_WAVES];
  int wave_i3[MM_WAVES];
  float wave_f2[MM_WAVES];
  float lex_d[MM_WAVES], lex_d2[MM_WAVES];
  int lex_p[MM_WAVES], lex_k[MM_WAVES];
  double lex_d2d[MM_WAVES];  // (the AGES build: the draw is a double)
  int a_lw;                  // ... and the last age level its clock has reached, for all waves
  int rowcount[MM_WAVES];
  float sym_dist;
  float red_f[MM_ROWS_MAX][MM_WAVES];
  int red_a[MM_ROWS_MAX][MM_WAVES], red_b[MM_ROWS_MAX][MM_WAVES];
  // the live list lives in the threads' registers (position t + 512 q in slot q of thread t): when a cluster leaves,
  // the positions behind it move up by one -- a lane takes its neighbour's, the last lane of a wave the first of the
  // next wave's from here
  short edge[MM_WAVES][MM_Q_GLOB];
  // rebuilt clusters of the merge, any order: position | cluster << 14 | rescan << 28; their new d(k, j), which is not
  // in memory until the end of the merge; their row minimum (+ threshold), as it was and then as the rescan leaves it
  unsigned upd[MM_UPD_LDS];
  float updv[MM_UPD_LDS];
  float updmv[MM_UPD_LDS];
  // feasible pairs of the merge, [0]: as found, [1]: in the reference's order
  unsigned pk[2][MM_PAIRS_LDS];   // key: position of the later cluster << 16 | position of the earlier one
  unsigned pxy[2][MM_PAIRS_LDS];  // later cluster << 16 | earlier cluster
  float psym[2][MM_PAIRS_LDS];    // symmetric distance of the pair (0 if the prior makes it a certain pair)
};

// The per-cluster state of a build and where each field lives (L_LDS / L_GLOBAL / L_HOT above).  HOT fields: read or
// written for every cluster of every merge from LDS or on the ordered part's path; COLD fields: read with the rows of
// a merge (row minima), by a few threads (sizes, min_values_CF of a pair) or from registers (the live list: `ci` is its
// mirror in memory, kept for the symmetric fallback and the AGES walk).
template <int LAY>
struct State {
  static constexpr bool hot_lds = LAY != L_GLOBAL, cold_lds = LAY == L_LDS, warm_lds = LAY == L_LDS || LAY == L_WARM;
  typedef typename std::conditional<hot_lds, short, int>::type hidx_t;
  typedef typename std::conditional<cold_lds, short, int>::type cidx_t;
  template <typename T>
  using hot = typename std::conditional<hot_lds, T *, MM_GLOBAL_PTR(T)>::type;  // (LDS: inferred)
  template <typename T>
  using warm = typename std::conditional<warm_lds, T *, MM_GLOBAL_PTR(T)>::type;
  template <typename T>
  using cold = typename std::conditional<cold_lds, T *, MM_GLOBAL_PTR(T)>::type;
  hot<float> mcd, mcd2;      // candidate (dist, dist2)
  hot<hidx_t> lin1, lin2;    // candidate pair (stale indices are part of the carried state)
  hot<unsigned char> flag;   // rebuilt in this merge
  warm<float> mv, mvcf;      // min_values, min_values_CF
  cold<cidx_t> ci;           // the live clusters in order (mirror of the threads' registers)
  cold<cidx_t> csz;          // cluster sizes (floats in the reference: exact integers)
  // the AGES build: the candidate's draw as a double and its age level, the cluster's own age level
  hot<double> d2d;
  hot<hidx_t> lv;
  cold<cidx_t> alv;
};

// Wave reductions on the DPP crossbar (quad swaps, half-row and row mirrors, the two row broadcasts: the total
// lands in lane 63 and is read back to all lanes) -- a __shfl_xor butterfly is six dependent LDS round trips per value,
// and a merge has a dozen reductions on its critical path.  Lanes a step does not write keep their own value.
template <int CTRL, int RM>
__device__ inline float dpp_keep_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, RM, 0xf, false));
}
template <int CTRL, int RM>
__device__ inline int dpp_keep_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, RM, 0xf, false);
}
#define MM_DPP_STEPS(X) X(0xB1, 0xf) X(0x4E, 0xf) X(0x141, 0xf) X(0x140, 0xf) X(0x142, 0xa) X(0x143, 0xc)
__device__ inline float wave_min_f(float v) {
#define MM_STEP(C, R) v = fminf(v, dpp_keep_f<C, R>(v));
  MM_DPP_STEPS(MM_STEP)
#undef MM_STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ inline int wave_min_i(int v) {
#define MM_STEP(C, R) v = min(v, dpp_keep_i<C, R>(v));
  MM_DPP_STEPS(MM_STEP)
#undef MM_STEP
  return __builtin_amdgcn_readlane(v, 63);
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
// lexicographic minimum of (d1, d2, pos)
__device__ inline bool lex_less(float a1, float a2, int ap, float b1, float b2, int bp) {
  return a1 < b1 || (a1 == b1 && (a2 < b2 || (a2 == b2 && ap < bp)));
}
__device__ inline void wave_lex_min(float &d1, float &d2, int &pos) {
#define MM_STEP(C, R)                                                          \
  {                                                                            \
    const float o1 = dpp_keep_f<C, R>(d1), o2 = dpp_keep_f<C, R>(d2);          \
    const int op = dpp_keep_i<C, R>(pos);                                      \
    if (lex_less(o1, o2, op, d1, d2, pos)) {                                   \
      d1 = o1;                                                                 \
      d2 = o2;                                                                 \
      pos = op;                                                                \
    }                                                                          \
  }
  MM_DPP_STEPS(MM_STEP)
#undef MM_STEP
  d1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d1), 63));
  d2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), 63));
  pos = __builtin_amdgcn_readlane(pos, 63);
}
__device__ inline void block_lex_min(float &d1, float &d2, int &pos, float *b1, float *b2, int *bp) {
  wave_lex_min(d1, d2, pos);
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

// ---- the AGES build (`--sample_ages`; tree_builder.cpp:3-22, :149-252, :601-965 and their twins with a prior)
// A candidate (dist, dist2, dist3, replace): dist3 -- the older of the pair's two sample ages -- is one of the distinct
// sample ages and is kept as its index in their sorted table (comparisons of ages = comparisons of indices);
// `replace` is not stored: after the refresh every merge begins with (:618) it equals "dist3 is beyond the clock" for
// every candidate there is, and that is how it is set when one is made (:216, :234).  lw = the largest index whose
// age the clock has reached.
constexpr int MM_AGE_EMPTY = 0x7fff;  // (more than any level: N <= 10240; fits the shorts of the state in LDS)
struct AgeCand {
  float d;
  double d2;
  int lv, l1, l2;
};
// operator> of tree_builder.cpp:7-22 (a = the holder): with a.replace and a.dist3 >= b.dist3 the first block returns
// for a.dist3 > b.dist3 and otherwise tests what the second block tests again
__device__ inline bool ages_gt(const AgeCand &a, float bd, double bd2, int blv, int lw) {
  const bool a_replace = a.lv != MM_AGE_EMPTY && a.lv > lw;
  return (a_replace && a.lv > blv) || a.d > bd || (a.d == bd && a.d2 > bd2);
}
// "may b take a's place": the test in front of every assignment to a candidate slot (:209-218) and to the running
// best (:230-237)
__device__ inline bool ages_takes(const AgeCand &a, float bd, double bd2, int blv, int lw) {
  return (a.d == INFINITY || blv <= lw) && ages_gt(a, bd, bd2, blv, lw);
}
// lexicographic minimum of (d1, d2, pos) over the wave, in every lane
__device__ inline void wave_lex_min_d(float &d1, double &d2, int &pos) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float o1 = __shfl_xor(d1, o, 64);
    const double o2 = __shfl_xor(d2, o, 64);
    const int op = __shfl_xor(pos, o, 64);
    if (o1 < d1 || (o1 == d1 && (o2 < d2 || (o2 == d2 && op < pos)))) {
      d1 = o1;
      d2 = o2;
      pos = op;
    }
  }
}
// one expected coalescence of k lineages (:1155, :1226): 2 / (k (k - 1)) * Ne, every operation rounded by itself
__device__ inline double ages_step(int k, int Ne) {
  return __dmul_rn(__ddiv_rn(2.0, __dmul_rn((double)k, (double)k - 1.0)), (double)Ne);
}

// (32-bit element offsets from a scalar base -- N <= 10240: one address register per load instead of two)
#define MM(a, b) p.M[mm_index((unsigned)(a), (unsigned)(b), (unsigned)N)]
#define MM2(a, b) (((MM_GLOBAL_PTR(const f32x2))(p.M + mm_index((unsigned)(a), (unsigned)(b), (unsigned)N)))[0])  // (d(a,b), d(b,a))
// (one half of a pair only: d(a,b) -- the first filter of a test runs on it, the few survivors fetch d(b,a))
#define MM1(a, b) (((MM_GLOBAL_PTR(const float))(p.M + mm_index((unsigned)(a), (unsigned)(b), (unsigned)N)))[0])
#define MMY(a, b) (((MM_GLOBAL_PTR(const float))(p.M + mm_index((unsigned)(a), (unsigned)(b), (unsigned)N)))[1])
#define SS(a, b) symm[(unsigned)(a) * (unsigned)N + (unsigned)(b)]

// One workgroup per tree: workgroup b builds the tree of params[b].
//
// A merge (i into j) on the workgroup, thread t holding the clusters at positions t, t + 512, ... of the live list IN
// ITS REGISTERS for the whole build (a_k; the positions behind a cluster that leaves move up by one lane):
//   A. every thread: the four entries (k,j), (k,i), (i,k), (j,k) of both matrices for its clusters k (all loads of a
//      pass issued at once, with the clusters' row minima, which stay in registers to the end of the merge), the
//      size-weighted updates written back; which clusters rebuild their
//      candidates (row minimum on a changed entry, or candidate touching i or j: appended to a list in LDS, their
//      candidates reset); partial minima of the merged cluster's row (both matrices) and the best candidate among
//      the clusters that keep theirs, (dist, dist2, position)-lexicographic -- one exchange for all of it;
//   B. the rows of the rebuilt clusters, ROWS at a time, d(k,l) per thread and position (one half of a pair: 4 of an
//      element's 16 bytes): first the row-minimum rescans (the reference's scan with early exit = "the old minimum if
//      it occurs before any smaller entry, else the row's minimum": three reductions, finished by one wave per row),
//      then, on the values still in registers, the row half of the candidate test of every pair the rebuilt cluster
//      is part of; the few survivors fetch the other half, d(l,k), themselves and append the feasible pairs, keyed by
//      (position of the later cluster, position of the earlier one), to a list in LDS;
//   C. the merged cluster's own pairs from its row as A left it (d(j,k) read back, in flight since the start of B; the
//      survivors fetch d(k,j)), keyed behind all others;
//   D. the pairs sorted by key (rank = number of smaller keys) -- the order in which the reference meets them --
//      and ONE lane draws the random numbers and updates the candidates in that order, then settles the running
//      best: per cluster the candidate it holds at its turn of the reference's loop, smallest (dist, dist2),
//      earliest position among exact ties;
//   E. the merged-away cluster leaves the list.
// Three dependent memory round trips and eight workgroup barriers per merge; everything else is LDS.
//
// AGES (`--sample_ages`): the same merge with the candidates' third key and the coalescence clock.  What is a
// reduction above -- the running best over the clusters in order -- is order-dependent there (a candidate beyond the
// clock takes an EMPTY place only, one within the clock displaces it whatever its distance), so wave 0 walks the
// live list once per merge, the pairs applied at their clusters' turns: exact, and slower (data sets with ancient
// samples are small).  Its per-cluster state lives in one place: LDS up to N = 4100 (L_LDS), global memory above.
template <int LAY, int MAXQ, bool AGES, int ROWS>
__device__ __forceinline__ void build_tree(const MMParamsDev &p, Shared &sh, unsigned char *dyn) {
  typedef typename State<LAY>::hidx_t hidx_t;
  typedef typename State<LAY>::cidx_t cidx_t;
  static_assert(ROWS <= MM_ROWS_MAX && ROWS <= MM_WAVES, "a wave finishes a row's rescan");
  static_assert((LAY != L_HOT && LAY != L_WARM) || !AGES, "the AGES build keeps all of its state in one place");
  const int N = p.N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float INF = INFINITY;
  const float threshold = p.threshold, threshold_CF = p.threshold_CF;
  State<LAY> st;
  if constexpr (LAY == L_LDS && AGES) {  // 33 bytes per cluster (the float draw of the plain build has no place here)
    st.d2d = reinterpret_cast<double *>(dyn);
    float *f = reinterpret_cast<float *>(st.d2d + N);
    st.mv = f;
    st.mvcf = f + N;
    st.mcd = f + 2 * (size_t)N;
    st.mcd2 = nullptr;
    short *s = reinterpret_cast<short *>(f + 3 * (size_t)N);
    st.lin1 = s;
    st.lin2 = s + N;
    st.ci = s + 2 * (size_t)N;
    st.csz = s + 3 * (size_t)N;
    st.lv = s + 4 * (size_t)N;
    st.alv = s + 5 * (size_t)N;
    st.flag = reinterpret_cast<unsigned char *>(s + 6 * (size_t)N);
  } else if constexpr (LAY == L_LDS) {
    float *f = reinterpret_cast<float *>(dyn);
    st.mv = f;
    st.mvcf = f + N;
    st.mcd = f + 2 * (size_t)N;
    st.mcd2 = f + 3 * (size_t)N;
    short *s = reinterpret_cast<short *>(f + 4 * (size_t)N);
    st.lin1 = s;
    st.lin2 = s + N;
    st.ci = s + 2 * (size_t)N;
    st.csz = s + 3 * (size_t)N;
    st.flag = reinterpret_cast<unsigned char *>(s + 4 * (size_t)N);
  } else if constexpr (LAY == L_HOT || LAY == L_WARM) {  // 13 / 21 bytes per cluster in LDS
    float *f = reinterpret_cast<float *>(dyn);
    st.mcd = f;
    st.mcd2 = f + N;
    if constexpr (LAY == L_WARM) {
      st.mv = f + 2 * (size_t)N;
      st.mvcf = f + 3 * (size_t)N;
      f += 2 * (size_t)N;
    } else {
      st.mv = p.min_values;
      st.mvcf = p.min_values_CF;
    }
    short *s = reinterpret_cast<short *>(f + 2 * (size_t)N);
    st.lin1 = s;
    st.lin2 = s + N;
    st.flag = reinterpret_cast<unsigned char *>(s + 2 * (size_t)N);
    st.ci = p.cluster_index;
    st.csz = p.cluster_size;
  } else {
    st.d2d = p.mc_dist2d;
    st.lv = p.mc_lvl;
    st.alv = p.age_lvl;
    st.mv = p.min_values;
    st.mvcf = p.min_values_CF;
    st.mcd = p.mc_dist;
    st.mcd2 = p.mc_dist2;
    st.lin1 = p.mc_lin1;
    st.lin2 = p.mc_lin2;
    st.ci = p.cluster_index;
    st.csz = p.cluster_size;
    st.flag = p.kflag;
  }
  // (timing: thread 0 only, accumulators in LDS -- twelve 64-bit counters per thread would cost the kernel its registers)
  if (tid == 0) {
    for (int x = 0; x < 16; x++) sh.tacc[x] = 0;
    sh.cnt_rescan_only = 0;
    sh.tmark = wall_clock64();
  }
  const long long tstart = wall_clock64(), cstart = clock64();
#define LAP(x)                           \
  if (p.timers && tid == 0) {            \
    const long long tn = wall_clock64(); \
    sh.tacc[x] += tn - sh.tmark;         \
    sh.tmark = tn;                       \
  }                                      \
  if (p.trace && tid == 0) {             \
    p.trace[0] = (unsigned)sh.n;         \
    p.trace[1] = (unsigned)(x);          \
  }

  // The tree leaves the workgroup -- built (0) or handed to the host (1: no room for the symmetric matrix, 2: more
  // tied candidates than the lists hold): thread 0, at a point all threads have reached.  The symmetric matrix goes
  // back to the pool; status, the tree and the carried state reach memory BEFORE the host hears of them in pinned
  // memory -- the builder's host thread takes its tree while the other workgroups still build theirs, no
  // end-of-kernel write-back in between.
  auto leave = [&](int code) {
    if (tid == 0) {
      *p.status = code;
      __threadfence_system();
      if (sh.sym_slot >= 0) __hip_atomic_store(&p.sym_locks[sh.sym_slot], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      *p.host_done = code;
      __threadfence_system();
    }
  };
  MM_GLOBAL_PTR(float) symm = nullptr;  // the symmetric matrix of the fallback, once one is taken from the pool
  if (tid == 0) sh.sym_slot = -1;

  // ---- QuickBuild set-up (:1061-1100)
  // (L_HOT: the list's mirror in memory is written when the symmetric fallback first needs it)
  constexpr bool CI_ALWAYS = LAY != L_HOT && LAY != L_WARM;
  for (int c = tid; c < N; c += MM_BLOCK) {
    if constexpr (CI_ALWAYS) st.ci[c] = (cidx_t)c;
    st.csz[c] = (cidx_t)1;
    st.mcd[c] = INF;
    st.flag[c] = 0;
    if constexpr (AGES) {
      st.d2d[c] = (double)INF;
      st.lv[c] = (hidx_t)MM_AGE_EMPTY;
      st.alv[c] = (cidx_t)p.age_lvl0[c];
    } else {
      st.mcd2[c] = INF;
    }
    if constexpr (LAY != L_GLOBAL) {  // the state carried from tree to tree comes in (pinned host memory)
      st.lin1[c] = (hidx_t)p.io_lin[c];
      st.lin2[c] = (hidx_t)p.io_lin[(size_t)N + c];
      // (L_HOT: min_values_CF as carried over is in the device array already -- the weave reads it there)
      if constexpr (LAY != L_HOT) st.mvcf[c] = p.io_mvcf[c];
    }
  }
  // the live list: position t + MM_BLOCK q in slot q of thread t, for the whole build
  int a_k[MAXQ];
#pragma unroll
  for (int q = 0; q < MAXQ; q++) a_k[q] = q * MM_BLOCK + tid < N ? q * MM_BLOCK + tid : -1;
  if (tid == 0) {
    rng_seed(sh.rng, 1u);
    sh.best.dist = INF;
    sh.best.dist2 = INF;
    sh.best.lin1 = -1;
    sh.best.lin2 = -1;
    sh.best_sym = sh.best;
    sh.use_sym = 0;
    sh.n = N;
    sh.nupd = 0;
    sh.npairs = 0;
  }
  __syncthreads();

  // ---- Initialize (:59-146 / :1647-1735): row minima (+ threshold); the minima themselves come from the penalty / prior passes
  for (int a = tid; a < N; a += MM_BLOCK) {
    st.mv[a] = p.rowmin_D[a] + threshold;
    if (p.has_prior) {
      const float old = st.mvcf[a], mc_ = p.rowmin_CF[a];  // old: carried over from the previous build (:2399-2400)
      st.mvcf[a] = (old > mc_ ? mc_ : old) + threshold_CF;
    }
  }
  __syncthreads();
  LAP(0);
  // One wave runs every ordered part in step -- the values are the same in all 64 lanes (LDS reads of one address),
  // lane 0 does the writes; the lanes matter when the generator's state is renewed (624 words, 64 at a time).
  // Per pair the LDS reads -- two words of the generator, both clusters' candidates, the next pair's record -- are
  // independent and go out together: one LDS latency per pair.
  // The draws come 64 at a time: lane l tempers words ridx + 2l, ridx + 2l + 1 of the state and forms the
  // (l + 1)-th uniform float from now -- std::uniform_real_distribution<double>(0,1) of libstdc++ (rng_unif),
  // narrowed to float (tree_builder.hpp:60) --; a pair then takes its number with one readlane instead of two LDS
  // reads and thirty dependent instructions on the ordered path.  (624 is even: a draw never straddles a renewal.)
  // (the AGES build keeps the draw as the double it is: Candidate::dist2, tree_builder.hpp:26, assigned from the
  //  distribution directly, :205)
  typedef typename std::conditional<AGES, double, float>::type rnd_t;
  int ridx = 624;       // (wave 0: position in the generator's state)
  rnd_t rnd_lane = 0;   // lane l: draw number l of the batch
  int rnd_have = 0, rnd_at = 0;
  auto next_rnd = [&]() -> rnd_t {
    if (rnd_at >= rnd_have) {
      if (ridx >= 624) {
        rng_renew_wave(sh.rng, lane);
        ridx = 0;
      }
      rnd_have = min(64, (624 - ridx) / 2);
      const int i0 = lane < rnd_have ? ridx + 2 * lane : 0;
      uint32_t y1 = sh.rng.mt[i0], y2 = sh.rng.mt[i0 + 1];
      y1 ^= y1 >> 11;
      y2 ^= y2 >> 11;
      y1 ^= (y1 << 7) & 0x9d2c5680u;
      y2 ^= (y2 << 7) & 0x9d2c5680u;
      y1 ^= (y1 << 15) & 0xefc60000u;
      y2 ^= (y2 << 15) & 0xefc60000u;
      y1 ^= y1 >> 18;
      y2 ^= y2 >> 18;
      const double sum = (double)y1 + (double)y2 * 4294967296.0;
      double ret = sum / 18446744073709551616.0;
      if (ret >= 1.0) ret = 0.99999999999999988897769753748434595763683319091796875;
      rnd_lane = (rnd_t)ret;
      ridx += 2 * rnd_have;
      rnd_at = 0;
    }
    const int at = __builtin_amdgcn_readfirstlane(rnd_at);
    rnd_at++;
    if constexpr (AGES) {
      const long long bits = __double_as_longlong(rnd_lane);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), at);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), at);
      return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    } else {
      return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rnd_lane), at));
    }
  };
  // ---- AGES: the clock (wave 0 keeps it; tree_builder.cpp:1123-1155 / :2407-2440) and the candidate slots
  int a_level = 0, a_lins = 0, a_lw = -1;  // sampling level reached, lineages alive there, last level within the clock
  double a_age = 0.0;
  auto clock_reaches = [&]() {  // a_lw follows a_age
    while (a_lw + 1 < p.n_levels && p.unique_ages[a_lw + 1] <= a_age) a_lw++;
  };
  if constexpr (AGES) {
    a_lins = p.ages_count[0];
    // the clock starts one expected coalescence in without a prior (:1155), at the youngest samples with one (:2440)
    a_age = p.has_prior ? p.unique_ages[0] : __dadd_rn(p.unique_ages[0], ages_step(a_lins, p.Ne));
    clock_reaches();
    if (tid == 0) sh.a_lw = a_lw;
  }
  auto slot_of = [&](int c) -> AgeCand {
    return AgeCand{st.mcd[c], st.d2d[c], (int)st.lv[c], (int)st.lin1[c], (int)st.lin2[c]};
  };
  // MinMatch's `cand` offered to both of its clusters (:205-218 and every block like it); xs / ys: what x and y hold
  // afterwards
  auto apply_ages = [&](int x, int y, float sym, AgeCand &xs, AgeCand &ys) {
    const double rnd = (double)next_rnd();
    xs = slot_of(x);
    ys = slot_of(y);
    const int clv = max((int)st.alv[x], (int)st.alv[y]);
    const AgeCand c{sym, rnd, clv, x, y};
    const bool tx = ages_takes(xs, sym, rnd, clv, a_lw), ty = ages_takes(ys, sym, rnd, clv, a_lw);
    if (tx) xs = c;
    if (ty) ys = c;
    if (lane == 0) {
      if (tx) {
        st.lin1[x] = (hidx_t)x;
        st.lin2[x] = (hidx_t)y;
        st.mcd[x] = sym;
        st.d2d[x] = rnd;
        st.lv[x] = (hidx_t)clv;
      }
      if (ty) {
        st.lin1[y] = (hidx_t)x;
        st.lin2[y] = (hidx_t)y;
        st.mcd[y] = sym;
        st.d2d[y] = rnd;
        st.lv[y] = (hidx_t)clv;
      }
    }
  };
  // the running best of Initialize / Coalesce (:230-237): wave 0's registers
  AgeCand abest{INF, (double)INF, MM_AGE_EMPTY, -1, -1};
  auto best_takes = [&](const AgeCand &mcand) {
    if (ages_takes(abest, mcand.d, mcand.d2, mcand.lv, a_lw)) abest = mcand;
  };
  // one feasible pair in the reference's order: one draw, both clusters' best candidate (:1704-1716); -> the draw
  auto apply = [&](int x, int y, float sym) -> float {
    const float rnd = next_rnd();
    const float ad = st.mcd[x], ad2 = st.mcd2[x], bdd = st.mcd[y], bdd2 = st.mcd2[y];
    if (lane == 0) {
      if (ad > sym || (ad == sym && ad2 > rnd)) {
        st.lin1[x] = (hidx_t)x;
        st.lin2[x] = (hidx_t)y;
        st.mcd[x] = sym;
        st.mcd2[x] = rnd;
      }
      if (bdd > sym || (bdd == sym && bdd2 > rnd)) {
        st.lin1[y] = (hidx_t)x;
        st.lin2[y] = (hidx_t)y;
        st.mcd[y] = sym;
        st.mcd2[y] = rnd;
      }
    }
    return rnd;
  };
  // mutually close pairs in (a, b) order (:1690-1722): weave_kernel has found them; MM_BLOCK rows at a time they
  // are staged in LDS in order and wave 0 draws
  auto draw_staged = [&](int count) {  // wave 0: the staged pairs sh.pxy[0] / sh.psym[0] in order (:1704-1722)
    if constexpr (AGES) {  // (:205-237)
      for (int e = 0; e < count; e++) {
        const unsigned xy = sh.pxy[0][e];
        AgeCand xs, ys;
        apply_ages((int)(xy >> 16), (int)(xy & 0xffffu), sh.psym[0][e], xs, ys);
        best_takes(ys);
      }
      if (lane == 0) {
        sh.best.dist = abest.d;
        sh.best.lin1 = abest.l1;
        sh.best.lin2 = abest.l2;
      }
      return;
    }
    Best bb = sh.best;
    for (int e = 0; e < count; e++) {
      const unsigned xy = sh.pxy[0][e];
      const float sym = sh.psym[0][e];
      const int aa = (int)(xy >> 16), b = (int)(xy & 0xffffu);
      // (b's candidate after this pair: what apply() is about to leave there)
      const float od = st.mcd[b], od2 = st.mcd2[b];
      const float rnd = apply(aa, b, sym);
      const bool took = od > sym || (od == sym && od2 > rnd);
      const float md = took ? sym : od, md2 = took ? rnd : od2;
      if (bb.dist > md || (bb.dist == md && bb.dist2 > md2)) {
        bb.lin1 = aa;
        bb.lin2 = b;
        bb.dist = sym;
        bb.dist2 = md2;
      }
    }
    if (lane == 0) sh.best = bb;
  };
  for (int a0 = 0; a0 < N; a0 += MM_BLOCK) {
    const int a = a0 + tid;
    const int c = a < N ? p.hit_cnt[a] : 0;
    int total;
    const int off = block_scan(c, &total, sh.wave_i);
    if (__syncthreads_or(c > MM_HITS)) {
      // a row of this stretch has more partners than the pair scan keeps (flat matrices): its rows are scanned here,
      // one by one, MM_BLOCK columns at a time
      for (int ar = a0; ar < min(N, a0 + MM_BLOCK); ar++) {
        const float mva = st.mv[ar];
        for (int b0 = ar + 1; b0 < N; b0 += MM_BLOCK) {
          const int b = b0 + tid;
          f32x4 e = {0.f, 0.f, 0.f, 0.f};
          bool hit = false;
          if (b < N) {
            e = MM(ar, b);
            hit = mva >= e.x && st.mv[b] >= e.y;
          }
          int cnt;
          const int at = block_scan(hit ? 1 : 0, &cnt, sh.wave_i);
          if (hit) {
            float sym = e.y + e.x;
            if constexpr (AGES) {  // (:1792-1797: the pairs the prior agrees with are kept, the others voided)
              if (p.has_prior && !(e.z <= st.mvcf[ar] && e.w <= st.mvcf[b])) sym = INF;
            } else {
              if (p.has_prior && e.z <= st.mvcf[ar] && e.w <= st.mvcf[b]) sym = 0.0f;
            }
            sh.pxy[0][at] = ((unsigned)ar << 16) | (unsigned)b;
            sh.psym[0][at] = sym;
          }
          __syncthreads();
          if (wave == 0) draw_staged(cnt);
          __syncthreads();
        }
      }
      continue;
    }
    // whole rows, as many as the staging area holds (all of them, usually), pass by pass
    for (int base = 0; base < total;) {
      if (tid == 0) sh.count = total;
      __syncthreads();
      const bool mine = c > 0 && off >= base && off + c <= base + MM_PAIRS_LDS;
      if (mine) {
        for (int e = 0; e < c; e++) {
          sh.pxy[0][off - base + e] = ((unsigned)a << 16) | p.hit_b[(size_t)a * MM_HITS + e];
          sh.psym[0][off - base + e] = p.hit_sym[(size_t)a * MM_HITS + e];
        }
      } else if (c > 0 && off >= base) {
        atomicMin(&sh.count, off);  // the first row that has to wait
      }
      __syncthreads();
      const int next = sh.count;
      if (wave == 0) draw_staged(next - base);
      __syncthreads();
      base = next;
    }
  }
  LAP(1);

  // ---- the merges
  // A tree the lists cannot hold is handed to the host (bail >= 0: the code).  ONE way out of the function, through
  // wave-uniform branches the compiler can see are uniform (readfirstlane): the workgroup goes on to its next tree
  // behind this call, and an exit the compiler takes for divergent -- it hangs on values read from LDS -- is laid
  // out as a loop in which a wave passes the caller's barriers once per group of lanes.
  int bail = -1;
  for (int num_nodes = N; num_nodes < 2 * N - 1; num_nodes++) {
    const int n = sh.n;
    if (sh.best.dist == INF && !sh.use_sym) {
      // no mutually closest pair: from here on the symmetric matrix picks the pair when there is none
      // (initialize_sym, tree_builder.cpp:255-293): s(a,l) = d(a,l) + d(l,a) over the live clusters, row minima
      // with the first cluster that reaches them, the smallest of those (first row, first cluster)
      if (tid == 0) {  // a matrix of the device's pool
        int got = -1;
        for (int sl = 0; sl < p.sym_slots && got < 0; sl++) {
          int expected = 0;
          if (__hip_atomic_compare_exchange_strong(&p.sym_locks[sl], &expected, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT))
            got = sl;
        }
        sh.sym_slot = got;
      }
      __syncthreads();
      const int slot = __builtin_amdgcn_readfirstlane(sh.sym_slot);
      if (slot < 0) {  // (none free, or no pool)
        bail = 1;
        break;
      }
      symm = p.SYM + (size_t)slot * N * N;
      if constexpr (!CI_ALWAYS) {  // the live list as it stands in the registers: from here on it is kept in memory too
#pragma unroll
        for (int q = 0; q < MAXQ; q++)
          if (a_k[q] >= 0) st.ci[q * MM_BLOCK + tid] = (cidx_t)a_k[q];
        __syncthreads();
      }
      float bs = INF;
      int bs_pos = n;
      for (int ia = wave; ia < n; ia += MM_WAVES) {
        const int a = st.ci[ia];
        float mv = INF;
        int mp = n;
        for (int il = lane; il < n; il += 64) {
          const int l = st.ci[il];
          if (l == a) continue;
          const f32x2 e = MM2(a, l);
          const float v = e.x + e.y;
          SS(a, l) = v;
          if (v < mv) {  // (ascending per lane: the first one stays)
            mv = v;
            mp = il;
          }
        }
        {
          float zero = 0.0f;
          wave_lex_min(mv, zero, mp);  // (value, position)
        }
        if (lane == 0) {
          p.min_values_sym[a] = mv;
          p.mcs_dist[a] = mv;
          if (mv < INF) {
            p.mcs_lin1[a] = a;
            p.mcs_lin2[a] = st.ci[mp];
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
          sh.best_sym.lin1 = p.mcs_lin1[st.ci[bs_pos]];
          sh.best_sym.lin2 = p.mcs_lin2[st.ci[bs_pos]];
        }
      }
      __syncthreads();
    }
    const bool by_sym = sh.best.dist == INF;
    const int i = by_sym ? sh.best_sym.lin1 : sh.best.lin1, j = by_sym ? sh.best_sym.lin2 : sh.best.lin2;
    const float csi = (float)st.csz[i], csj = (float)st.csz[j];
    const float added = csi + csj;
    if (tid == 0) {  // (:2437-2460 happens on the host, from this log: no loads on the merge's path)
      p.merge_i[num_nodes - N] = i;
      p.merge_j[num_nodes - N] = j;
    }

    // -- A: this thread's clusters
    // (a / added for the size-weighted means: `added` is one integer <= 10240 for the whole merge, so the quotient
    //  comes from a double product with its reciprocal -- the exact quotient of a 24-bit by a 14-bit number is at
    //  least 2^-39 (relative) off every rounding boundary of float unless it is a float itself, the product is
    //  within 2^-51: the same float as the division's.  Quotients in the subnormal range take the division.)
    const double rc_added = 1.0 / (double)added;
    auto over_added = [&](float a) -> float {
      const float q = (float)((double)a * rc_added);
      return fabsf(q) >= 1e-30f || a == 0.0f ? q : a / added;
    };
    // (the live list shrinks from N to 2: the register slots past it are skipped by wave-uniform branches, not by
    //  predication -- half of all slots over a build)
    const int nq = (n + MM_BLOCK - 1) / MM_BLOCK;
    float mvreg[MAXQ];  // the row minima (+ threshold) of this thread's clusters as the merge finds them
    float mv_cf = INF, mvj = INF, bd = INF, bd2 = INF;
    double bd2d = (double)INF;  // (AGES)
    const int lw_all = AGES ? sh.a_lw : 0;
    int bpos = n, bk = -1;
#ifndef MM_QC20
#define MM_QC20 4
#endif
    // clusters per pass (ten at once cost more in spilled registers than the second round trip)
#ifndef MM_QC10
#define MM_QC10 5
#endif
    constexpr int QC = MAXQ == MM_Q_GLOB ? MM_QC20 : MAXQ % 5 == 0 ? (ROWS == 1 ? 5 : MM_QC10) : 4;
#pragma unroll
    for (int q0 = 0; q0 < MAXQ; q0 += QC) {
      if (q0 * MM_BLOCK >= n) break;
      // every load of the pass from memory first: the two rows and the clusters' row minima; what lives in LDS is read
      // cluster by cluster under their latency (state in global memory: asked for with the rows)
      f32x4 ei[QC], ej[QC];
#pragma unroll
      for (int qq = 0; qq < QC; qq++) {
        const int k = a_k[q0 + qq];
        if (k < 0 || k == j || k == i) continue;
        ei[qq] = MM(i, k);
        ej[qq] = MM(j, k);
      }
      // the candidate's fields from LDS: all of the pass at once while the registers allow (256 per lane: 2 ms of a
      // tree under load), at the cluster's turn in the 128-register kernels
      constexpr bool JIT = State<LAY>::hot_lds && (ROWS == 1 || QC > 5);
      float s_d1[JIT ? 1 : QC], s_d2[JIT ? 1 : QC];
      int s_l1[JIT ? 1 : QC], s_l2[JIT ? 1 : QC];
      double s_d2d[AGES && !JIT ? QC : 1];
      int s_lv[AGES && !JIT ? QC : 1];
#pragma unroll
      for (int qq = 0; qq < QC; qq++) {
        const unsigned k = a_k[q0 + qq] >= 0 ? (unsigned)a_k[q0 + qq] : 0u;
        mvreg[q0 + qq] = st.mv[k];
        if constexpr (!JIT) {
          s_l1[qq] = st.lin1[k];
          s_l2[qq] = st.lin2[k];
          s_d1[qq] = st.mcd[k];
          if constexpr (!AGES) s_d2[qq] = st.mcd2[k];
          if constexpr (AGES) {
            s_d2d[qq] = st.d2d[k];
            s_lv[qq] = st.lv[k];
          }
        }
      }
#pragma unroll
      for (int qq = 0; qq < QC; qq++) {
        const int q = q0 + qq;
        const int ik = q * MM_BLOCK + tid;
        const int k = a_k[q];
        if (k < 0) continue;
        if (k == j || k == i) {
          if (k == i) sh.ipos = ik;
          continue;
        }
        float ncjk = 0.0f, nckj = 0.0f;
        if (p.has_prior) {
          const float ckj = ej[qq].w, cki = ei[qq].w, cik = ei[qq].z, cjk = ej[qq].z;
          ncjk = cjk;
          nckj = ckj;
          if (cik != cjk) ncjk = over_added(csi * cik + csj * cjk);
          if (cki != ckj) nckj = over_added(csi * cki + csj * ckj);
          if (mv_cf > ncjk) mv_cf = ncjk;
        }
        const float dkj = ej[qq].y, dki = ei[qq].y, dik = ei[qq].x, djk = ej[qq].x;
        float njk = djk, nkj = dkj;
        if (dik != djk) njk = over_added(csi * dik + csj * djk);
        if (dki != dkj) nkj = over_added(csi * dki + csj * dkj);
        // (written whether changed or not: the same bits where the reference leaves the entry alone)
        MM(j, k) = f32x4{njk, nkj, ncjk, nckj};
        MM(k, j) = f32x4{nkj, njk, nckj, ncjk};
        if (njk < mvj) mvj = njk;
        bool rescan = false;
        const float mvk = mvreg[q];
        if (dkj != dki) {
          rescan = (double)fabsf(mvk - threshold - dkj) < 1e-4 || (double)fabsf(mvk - threshold - dki) < 1e-4;
        }
        const int sq = JIT ? 0 : qq;
        if constexpr (JIT) {
          s_l1[0] = st.lin1[k];
          s_l2[0] = st.lin2[k];
          s_d1[0] = st.mcd[k];
          if constexpr (!AGES) s_d2[0] = st.mcd2[k];
          if constexpr (AGES) {
            s_d2d[0] = st.d2d[k];
            s_lv[0] = st.lv[k];
          }
        }
        const int l1 = s_l1[sq], l2 = s_l2[sq];
        const bool touches = l1 == j || l2 == j || l1 == i || l2 == i;
        if (p.timers && rescan && !touches) atomicAdd(&sh.cnt_rescan_only, 1);
        // (AGES with a prior: a candidate that touches i or j alone does not send k through the rebuilding branch
        //  when none of its four distances moved -- :2131 against :653 --, the candidate is renamed instead, :2342)
        bool rebuilds = rescan || touches;
        if constexpr (AGES) rebuilds = rescan || (touches && (!p.has_prior || dkj != dki || djk != dik));
        if (rebuilds) {  // k rebuilds its candidates (:1893-1911)
          st.flag[k] = 1;
          st.mcd[k] = INF;
          if constexpr (AGES) {
            st.d2d[k] = (double)INF;
            st.lv[k] = (hidx_t)MM_AGE_EMPTY;
          } else {
            st.mcd2[k] = INF;
          }
          const int slot = atomicAdd(&sh.nupd, 1);
          const unsigned rec = (unsigned)ik | ((unsigned)k << 14) | (rescan ? 1u << 28 : 0u);
          if (slot < MM_UPD_LDS) {
            sh.upd[slot] = rec;
            sh.updv[slot] = nkj;
            sh.updmv[slot] = mvk;
          } else if (AGES || slot < MM_UPD_MAX) {  // (the tail of a long list: the row minimum stays in memory)
            p.upd_g[slot - MM_UPD_LDS] = rec;
            p.updv_g[slot - MM_UPD_LDS] = nkj;
          }
        } else if constexpr (AGES) {
          if (l1 == i) st.lin1[k] = (hidx_t)j;
          if (l2 == i) st.lin2[k] = (hidx_t)j;
          // k keeps its candidate.  The running best (:230-237) takes the first candidate WITHIN the clock with a
          // finite distance from whatever it holds, from then on nothing but a smaller one of that kind, and a slot
          // leaves that kind only for a smaller one of it: if one of the clusters that keep theirs holds such a
          // candidate, the best of the merge is the smallest (dist, dist2) among them all -- a reduction.  If none
          // does, wave 0 walks the list (below).
          const float d1 = s_d1[sq];
          const double d2 = s_d2d[sq];
          if (s_lv[sq] <= lw_all && d1 < INF && (bd > d1 || (bd == d1 && bd2d > d2))) {
            bd = d1;
            bd2d = d2;
            bpos = ik;
            bk = k;
          }
        } else {  // k keeps its candidate: what the reference's running best sees at k's turn
          const float d1 = s_d1[sq], d2 = s_d2[sq];
          if (bd > d1 || (bd == d1 && bd2 > d2)) {  // (ascending positions per thread: the first one wins)
            bd = d1;
            bd2 = d2;
            bpos = ik;
            bk = k;
          }
        }
      }
    }
    mvj = wave_min_f(mvj);
    mv_cf = wave_min_f(mv_cf);
    const int bpos_mine = bpos;
    if constexpr (AGES) wave_lex_min_d(bd, bd2d, bpos);
    else wave_lex_min(bd, bd2, bpos);
    if (bpos < n && bpos_mine == bpos) sh.lex_k[wave] = bk;  // (one lane: positions are unique)
    if (lane == 0) {
      sh.wave_f[wave] = mvj;
      sh.wave_f2[wave] = mv_cf;
      sh.lex_d[wave] = bd;
      sh.lex_d2[wave] = bd2;
      if constexpr (AGES) sh.lex_d2d[wave] = bd2d;
      sh.lex_p[wave] = bpos;
    }
    __syncthreads();
    LAP(2);
    float min_value_j = sh.wave_f[0], mvcf_j = sh.wave_f2[0];
#pragma unroll
    for (int w = 1; w < MM_WAVES; w++) {
      min_value_j = fminf(min_value_j, sh.wave_f[w]);
      mvcf_j = fminf(mvcf_j, sh.wave_f2[w]);
    }
    min_value_j += threshold;
    mvcf_j += threshold_CF;
    const int nupd = __builtin_amdgcn_readfirstlane(sh.nupd);
    if (!AGES && nupd > MM_UPD_MAX) {  // (degenerate matrices: this tree is the host's)
      bail = 2;
      break;
    }
    // (a list longer than MM_UPD_LDS has its tail in global memory; there the row minimum is read where it lives)
    auto upd_at = [&](int u) -> unsigned { return u < MM_UPD_LDS ? sh.upd[u] : p.upd_g[u - MM_UPD_LDS]; };
    auto updv_at = [&](int u) -> float { return u < MM_UPD_LDS ? sh.updv[u] : p.updv_g[u - MM_UPD_LDS]; };
    auto updmv_at = [&](int u, int k) -> float { return u < MM_UPD_LDS ? sh.updmv[u] : st.mv[k]; };
    auto rec_pos = [](unsigned e) -> int { return (int)(e & 0x3fffu); };
    auto rec_k = [](unsigned e) -> int { return (int)((e >> 14) & 0x3fffu); };
    auto rec_rescan = [](unsigned e) -> bool { return (e >> 28) != 0; };
    if (p.timers && tid == 0) {
      sh.tacc[12] += nupd * 100;
      sh.tacc[13] += sh.cnt_rescan_only * 100;
      sh.tacc[14] += nupd == 0 ? 100 : 0;
      sh.tacc[15] += sh.cnt_rescan_only > 0 ? 100 : 0;
      sh.cnt_rescan_only = 0;
    }

    // -- B: rows of the rebuilt clusters: (d(k,l), d(l,k)) along row k of M, ROWS rows per pass.  The row-minimum
    // rescans (:1875-1890) come first -- the reference's scan with early exit = "the old minimum if it occurs before
    // any smaller entry, else the row's minimum", three reductions finished by one wave per row --, then the
    // candidate tests of every pair a rebuilt cluster is part of (:1893-1911 for the clusters before it, :1913-2018
    // for the ones behind it), on the same registers when the merge has no more than ROWS rebuilt clusters.
    float cj[MAXQ];  // row j of M as this thread wrote it in A, d(j,k): asked for now, used in C
#pragma unroll
    for (int q = 0; q < MAXQ; q++) {
      if (q >= nq) break;
      const int k = a_k[q];
      cj[q] = (k >= 0 && k != i && k != j) ? MM1(j, k) : INF;
    }
    float v[ROWS][MAXQ];  // d(k, l) along the rows of the rebuilt clusters k
    auto load_rows = [&](const int (&ks)[ROWS]) {
#pragma unroll
      for (int r = 0; r < ROWS; r++) {
        if (ks[r] < 0) continue;
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
          if (q >= nq) break;
          v[r][q] = a_k[q] >= 0 ? MM1(ks[r], a_k[q]) : INF;
        }
      }
    };
    // rescans of the rows ks[r] >= 0 (entries us[r] of the list) held in v; patch[r]: the row's new entry at column j
    // (not in memory yet).  The new minimum goes to memory and, for the tests, to the list.
    auto rescan_rows = [&](const int (&ks)[ROWS], const int (&us)[ROWS], const float (&patch)[ROWS]) {
#pragma unroll
      for (int r = 0; r < ROWS; r++) {
        const int k = ks[r];
        if (k < 0) continue;
        const float old = updmv_at(us[r], k) - threshold;
        float fm = INF;
        int pos_old = n, pos_less = n;
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
          if (q >= nq) break;
          const int l = a_k[q];
          if (l < 0 || l == i || l == k) continue;
          const float x = l == j ? patch[r] : v[r][q];
          fm = fminf(fm, x);
          if (x == old) pos_old = min(pos_old, q * MM_BLOCK + tid);
          if (x < old) pos_less = min(pos_less, q * MM_BLOCK + tid);
        }
        fm = wave_min_f(fm);
        pos_old = wave_min_i(pos_old);
        pos_less = wave_min_i(pos_less);
        if (lane == 0) {
          sh.red_f[r][wave] = fm;
          sh.red_a[r][wave] = pos_old;
          sh.red_b[r][wave] = pos_less;
        }
      }
      __syncthreads();
      int myk = -1, myu = 0;  // one wave finishes one row
#pragma unroll
      for (int r = 0; r < ROWS; r++)
        if (wave == r) {
          myk = ks[r];
          myu = us[r];
        }
      if (myk >= 0) {
        float fm = lane < MM_WAVES ? sh.red_f[wave][lane] : INF;
        int pos_old = lane < MM_WAVES ? sh.red_a[wave][lane] : n;
        int pos_less = lane < MM_WAVES ? sh.red_b[wave][lane] : n;
        fm = wave_min_f(fm);
        pos_old = wave_min_i(pos_old);
        pos_less = wave_min_i(pos_less);
        const float old = updmv_at(myu, myk) - threshold;
        if (lane == 0) {
          const float nv = ((pos_old < n && pos_old < pos_less) ? old : fm) + threshold;
          st.mv[myk] = nv;
          if (myu < MM_UPD_LDS) sh.updmv[myu] = nv;
        }
      }
      __syncthreads();
    };
    // (a feasible pair: its symmetric distance is worked out when the pairs are put in order -- one more element of
    //  M per pair, fetched for all pairs at once)
    auto append_pair = [&](unsigned key, int x, int y) {
      const int slot = atomicAdd(&sh.npairs, 1);
      if (slot < MM_PAIRS_LDS) {
        sh.pk[0][slot] = key;
        sh.pxy[0][slot] = ((unsigned)x << 16) | (unsigned)y;
      } else if (slot - MM_PAIRS_LDS < p.pair_cap) {
        auto g = p.pair_g + (size_t)(slot - MM_PAIRS_LDS) * 3;
        g[0] = key;
        g[1] = ((unsigned)x << 16) | (unsigned)y;
      }
    };
    // (slot q of this thread's registers for a q known at run time only -- the few survivors of a test: a chain of
    //  selects, not an indexed array, which would live in scratch memory)
    auto slot_k = [&](int q) -> int {
      int k = -1;
#pragma unroll
      for (int x = 0; x < MAXQ; x++) k = x == q ? a_k[x] : k;
      return k;
    };
    // this thread's cluster k in slot q: its row minimum as A found it, or -- rebuilt in this merge -- as the rescans left it
    auto mv_now = [&](int q, int k) -> float {
      if (st.flag[k]) return st.mv[k];
      float x = INF;
#pragma unroll
      for (int y = 0; y < MAXQ; y++) x = y == q ? mvreg[y] : x;
      return x;
    };
    // candidate tests of the rows u0 .. u0+ROWS-1 of the list, held in v / w
    auto test_rows = [&](int u0) {
#pragma unroll
      for (int r = 0; r < ROWS; r++) {
        if (u0 + r >= nupd) continue;
        const unsigned rec = upd_at(u0 + r);
        const int up = rec_pos(rec), ku = rec_k(rec);
        const float mvk = updmv_at(u0 + r, ku);
        unsigned surv = 0;  // bit q: (row r, this thread's cluster q) passes the row's half of the test
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
          if (q >= nq) break;
          const int l = a_k[q];
          const bool ok = l >= 0 && l != i && l != j && l != ku && v[r][q] <= mvk;
          surv |= ok ? 1u << q : 0u;
        }
        while (surv) {  // (few)
          const int q = __ffs((int)surv) - 1;
          surv &= surv - 1;
          const int il = q * MM_BLOCK + tid, l = slot_k(q);
          // a later cluster meets the rebuilt ones before it; a rebuilt one meets them from its own row
          if (il > up && st.flag[l]) continue;
          if (!(MMY(ku, l) <= mv_now(q, l))) continue;  // d(l, ku): the other half of the pair, from memory
          if (il < up)  // the rebuilt cluster meets the clusters before it
            append_pair(((unsigned)up << 16) | (unsigned)il, ku, l);
          else
            append_pair(((unsigned)il << 16) | (unsigned)up, l, ku);
        }
      }
    };
    if (nupd <= ROWS) {  // (the usual case) one pass
      int ks[ROWS], kres[ROWS], us[ROWS];
      float patch[ROWS];
      bool anyres = false;
#pragma unroll
      for (int r = 0; r < ROWS; r++) {
        const unsigned e = r < nupd ? sh.upd[r] : 0u;
        ks[r] = r < nupd ? rec_k(e) : -1;
        kres[r] = (r < nupd && rec_rescan(e)) ? ks[r] : -1;
        us[r] = r;
        patch[r] = r < nupd ? sh.updv[r] : INF;
        anyres |= kres[r] >= 0;
      }
      load_rows(ks);
      if (anyres) rescan_rows(kres, us, patch);
      LAP(3);
      test_rows(0);
    } else {
      for (int u = 0; u < nupd;) {  // the flagged rows, ROWS at a time
        int ks[ROWS], us[ROWS];
        float patch[ROWS];
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          ks[r] = -1;
          us[r] = 0;
          patch[r] = INF;
        }
        for (; u < nupd && cnt < ROWS; u++) {
          const unsigned e = upd_at(u);
          if (!rec_rescan(e)) continue;
          const int k = rec_k(e);
          const float pv = updv_at(u);
#pragma unroll
          for (int r = 0; r < ROWS; r++)
            if (r == cnt) {
              ks[r] = k;
              us[r] = u;
              patch[r] = pv;
            }
          cnt++;
        }
        if (cnt == 0) break;
        load_rows(ks);
        rescan_rows(ks, us, patch);
      }
      LAP(3);
      for (int u0 = 0; u0 < nupd; u0 += ROWS) {
        int ks[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) ks[r] = u0 + r < nupd ? rec_k(upd_at(u0 + r)) : -1;
        load_rows(ks);
        test_rows(u0);
      }
    }
    // -- C: candidates with the merged cluster j, behind all others (:2033-2064)
    {
      unsigned cand = 0;
#pragma unroll
      for (int q = 0; q < MAXQ; q++) {
        if (q >= nq) break;
        const int k = a_k[q];
        const bool ok = k >= 0 && k != i && k != j && cj[q] <= min_value_j;
        cand |= ok ? 1u << q : 0u;
      }
      while (cand) {  // (few)
        const int q = __ffs((int)cand) - 1;
        cand &= cand - 1;
        const int k = slot_k(q);
        if (MMY(j, k) <= mv_now(q, k)) append_pair(0x80000000u | (unsigned)(q * MM_BLOCK + tid), k, j);
      }
    }
    __syncthreads();
    LAP(4);
    // -- D: the pairs in the reference's order
    const int m = __builtin_amdgcn_readfirstlane(sh.npairs);
    if (m - MM_PAIRS_LDS > p.pair_cap) {
      bail = 2;
      break;
    }
    auto pair_at = [&](int side, int e, unsigned &key, unsigned &xy, float &sym) {
      if (e < MM_PAIRS_LDS) {
        key = sh.pk[side][e];
        xy = sh.pxy[side][e];
        sym = sh.psym[side][e];
      } else {
        auto g = p.pair_g + ((size_t)side * p.pair_cap + (size_t)(e - MM_PAIRS_LDS)) * 3;
        key = g[0];
        xy = g[1];
        sym = __uint_as_float(g[2]);
      }
    };
    // Every pair's symmetric distance (tree_builder.cpp:1699-1702): d(y,x) + d(x,y), or 0 when the pair is also
    // mutually closest under the prior -- one element M[x][y] per pair, all pairs at once -- and its place in the
    // reference's order (rank = number of smaller keys).
    int side = m > 0 ? 1 : 0;
    bool bucketed = false;
    if constexpr (AGES) bucketed = m > MM_BUCKET_MIN;
    if (bucketed) {
      // A merge that sends most clusters through the rebuilding branch has tens of thousands of feasible pairs and the
      // rank above is quadratic: the pairs are counted per later cluster (bucket n: the merged cluster's), moved to
      // their bucket, and ranked inside it -- sides 0 -> 1 -> 0.  (Counts are read back past the L1: atomics do not
      // update it.)
      auto pair_put = [&](int sd, int e, unsigned key, unsigned xy, float sym) {
        if (e < MM_PAIRS_LDS) {
          sh.pk[sd][e] = key;
          sh.pxy[sd][e] = xy;
          sh.psym[sd][e] = sym;
        } else {
          auto g = p.pair_g + ((size_t)sd * p.pair_cap + (size_t)(e - MM_PAIRS_LDS)) * 3;
          g[0] = key;
          g[1] = xy;
          g[2] = __float_as_uint(sym);
        }
      };
      auto bucket_of = [&](unsigned key) -> int { return (key & 0x80000000u) ? n : (int)(key >> 16); };
      auto count_of = [&](int b) -> int { return __hip_atomic_load(&p.bucket[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      for (int x = tid; x <= n; x += MM_BLOCK) __hip_atomic_store(&p.bucket[x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      for (int e = tid; e < m; e += MM_BLOCK) {
        unsigned key, xy;
        float sym;
        pair_at(0, e, key, xy, sym);
        __hip_atomic_fetch_add(&p.bucket[bucket_of(key)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      int run = 0;
      for (int x0 = 0; x0 <= n; x0 += MM_BLOCK) {
        const int x = x0 + tid;
        const int c = x <= n ? count_of(x) : 0;
        int tot;
        const int off = block_scan(c, &tot, sh.wave_i);
        if (x <= n) p.bucket_off[x] = run + off;
        run += tot;
        __syncthreads();
      }
      for (int x = tid; x <= n; x += MM_BLOCK) __hip_atomic_store(&p.bucket[x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      for (int e = tid; e < m; e += MM_BLOCK) {
        unsigned key, xy;
        float sym;
        pair_at(0, e, key, xy, sym);
        const int b = bucket_of(key);
        const int at = p.bucket_off[b] + __hip_atomic_fetch_add(&p.bucket[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pair_put(1, at, key, xy, 0.0f);
      }
      __syncthreads();
      for (int e = tid; e < m; e += MM_BLOCK) {
        unsigned key, xy;
        float sym;
        pair_at(1, e, key, xy, sym);
        const int b = bucket_of(key);
        const int lo = p.bucket_off[b], hi = lo + count_of(b);
        int rank = lo;
        for (int g = lo; g < hi; g++) {
          unsigned kg, xg;
          float sg;
          pair_at(1, g, kg, xg, sg);
          rank += kg < key ? 1 : 0;
        }
        const int x = (int)(xy >> 16), y = (int)(xy & 0xffffu);
        const f32x4 f = MM(x, y);
        sym = f.y + f.x;
        if (p.has_prior && f.z <= st.mvcf[x] && f.w <= (y == j ? mvcf_j : st.mvcf[y])) sym = 0.0f;
        // (side 0 is read by nobody any more: every thread is past its last pair_at(0, ...) -- the barrier above)
        pair_put(0, rank, key, xy, sym);
      }
      __syncthreads();
      side = 0;
    } else if (m > 0) {
      for (int e = tid; e < m; e += MM_BLOCK) {
        unsigned key, xy;
        float sym;
        pair_at(0, e, key, xy, sym);
        const int x = (int)(xy >> 16), y = (int)(xy & 0xffffu);
        const f32x4 f = MM(x, y);  // (d(x,y), d(y,x), cf(x,y), cf(y,x))
        int rank = 0;
        const int ml = min(m, MM_PAIRS_LDS);
        for (int g = 0; g < ml; g++) rank += sh.pk[0][g] < key ? 1 : 0;
        for (int g = MM_PAIRS_LDS; g < m; g++) rank += p.pair_g[(size_t)(g - MM_PAIRS_LDS) * 3] < key ? 1 : 0;
        sym = f.y + f.x;
        if (p.has_prior && f.z <= st.mvcf[x] && f.w <= (y == j ? mvcf_j : st.mvcf[y])) sym = 0.0f;
        if (rank < MM_PAIRS_LDS) {
          sh.pk[1][rank] = key;
          sh.pxy[1][rank] = xy;
          sh.psym[1][rank] = sym;
        } else {
          auto g = p.pair_g + ((size_t)p.pair_cap + (size_t)(rank - MM_PAIRS_LDS)) * 3;
          g[0] = key;
          g[1] = xy;
          g[2] = __float_as_uint(sym);
        }
      }
      __syncthreads();
    }
    if (p.timers && tid == 0) sh.tacc[9] += m;  // (pairs drawn)
    LAP(5);
    // -- E (first half): the merged-away cluster leaves the list, the rebuilt marks are taken back -- under the
    // ordered part, which names clusters, not positions, from here on (the symmetric path, rare, keeps the list
    // until it is through)
    const bool sym_now = AGES || sh.use_sym != 0;  // (the AGES walk names positions: the list stays until it is through)
    // The positions behind i move up by one: a lane takes its neighbour's cluster, the last lane of a wave the first
    // lane's of the next wave (sh.edge, written before the barrier between the two halves).
    auto erase_read = [&]() {
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
          if (q >= nq) break;
          sh.edge[wave][q] = (short)a_k[q];
        }
      }
      for (int u = tid; u < nupd; u += MM_BLOCK) st.flag[rec_k(upd_at(u))] = 0;
    };
    auto erase_write = [&]() {
      const int ipos = sh.ipos;
      const bool mirror = CI_ALWAYS || sh.use_sym != 0;
#pragma unroll
      for (int q = 0; q < MAXQ; q++) {
        if (q >= nq) break;
        const int ik = q * MM_BLOCK + tid;
        // (lane l takes lane l + 1's: one DPP move across the wave, wave_shl:1; lane 63 -- it keeps -1 -- from the edge)
        int nx = __builtin_amdgcn_update_dpp(-1, a_k[q], 0x130, 0xf, 0xf, false);
        if (lane == 63) {
          if (wave + 1 < MM_WAVES) nx = sh.edge[wave + 1][q];
          else if (q + 1 < MAXQ) nx = sh.edge[0][q + 1];
          else nx = -1;
        }
        if (ik >= ipos) {
          a_k[q] = ik + 1 < n ? nx : -1;
          if (mirror && ik + 1 < n) st.ci[ik] = (cidx_t)nx;
        }
      }
      if (tid == 0) sh.n = n - 1;
    };
    // The best among the clusters that keep their candidate (a copy taken before the draws: they may still
    // change such a cluster's candidate, behind its turn)
    float ud = INF, ud2 = INF;
    double ud2d = (double)INF;  // (AGES)
    int upos = n, bl1 = -1, bl2 = -1;
    if (wave == 0) {
      ud = lane < MM_WAVES ? sh.lex_d[lane] : INF;
      ud2 = lane < MM_WAVES ? sh.lex_d2[lane] : INF;
      upos = lane < MM_WAVES ? sh.lex_p[lane] : n;
      const int upos_mine = upos;
      if constexpr (AGES) {
        ud2d = lane < MM_WAVES ? sh.lex_d2d[lane] : (double)INF;
        wave_lex_min_d(ud, ud2d, upos);
      } else {
        wave_lex_min(ud, ud2, upos);
      }
      if (upos < n) {  // (the cluster at that position: its wave left it in lex_k)
        const unsigned long long from = __ballot(lane < MM_WAVES && upos_mine == upos);
        const int kb = sh.lex_k[__builtin_ctzll(from)];
        bl1 = st.lin1[kb];
        bl2 = st.lin2[kb];
      }
    }
    if (!sym_now) {
      erase_read();
      __syncthreads();
      erase_write();
    }
    if (p.timers && tid == 0) sh.tacc[10] += wall_clock64() - sh.tmark;  // (of "ordered": before the first draw)
    if constexpr (AGES) {
      if (wave == 0) {
        // Coalesce's loop over the clusters in order (:613-881 / :2085-2341) as far as candidates go: a cluster that
        // is the LATER one of feasible pairs gets them at its turn (the reset of a rebuilt one happened in A: nothing
        // reaches a slot before its cluster's turn -- pairs are applied at the turn of their later cluster), then the
        // running best looks at what the cluster holds.  64 positions are fetched at a time; the empty ones are
        // skipped, and what was fetched for a cluster is what it holds at its turn unless it has pairs of its own.
        abest = AgeCand{INF, (double)INF, MM_AGE_EMPTY, sh.best.lin1, sh.best.lin2};  // (:611)
        int e = 0;
        unsigned key = 0x80000000u, xy = 0;
        float sym = 0.0f;
        if (m > 0) pair_at(side, 0, key, xy, sym);
        // One of the clusters that keep their candidate holds one within the clock (A): the best of the merge is the
        // smallest such candidate -- theirs, or what a cluster with pairs of its own holds behind them, or the merged
        // cluster's.  Otherwise: the walk.
        const bool reduced = upos < n;
        if (reduced) {
          AgeCand sb{INF, (double)INF, MM_AGE_EMPTY, -1, -1};
          int spos = n;
          while (e < m && !(key & 0x80000000u)) {
            const int cur = (int)(key >> 16);
            AgeCand t{INF, (double)INF, MM_AGE_EMPTY, -1, -1}, ys;
            while (e < m && !(key & 0x80000000u) && (int)(key >> 16) == cur) {
              const int x = (int)(xy >> 16), y = (int)(xy & 0xffffu);
              const float symc = sym;
              e++;
              if (e < m) pair_at(side, e, key, xy, sym);
              apply_ages(x, y, symc, t, ys);
            }
            if (t.lv <= a_lw && t.d < INF && (sb.d > t.d || (sb.d == t.d && sb.d2 > t.d2))) {  // (ascending positions)
              sb = t;
              spos = cur;
            }
          }
          abest = AgeCand{ud, ud2d, 0, bl1, bl2};
          if (spos < n && (sb.d < ud || (sb.d == ud && (sb.d2 < ud2d || (sb.d2 == ud2d && spos < upos))))) abest = sb;
        }
        for (int base = 0; base < n && !reduced; base += 64) {
          const int pos_l = base + lane;
          const int k_l = pos_l < n ? (int)st.ci[pos_l] : -1;
          const bool valid = k_l >= 0 && k_l != i && k_l != j;
          AgeCand r{INF, (double)INF, MM_AGE_EMPTY, -1, -1};
          if (valid) r = slot_of(k_l);
          unsigned long long mask = __ballot(valid && r.lv != MM_AGE_EMPTY);
          for (;;) {
            const unsigned key_u = (unsigned)__builtin_amdgcn_readfirstlane((int)key);
            const int pnext = (e < m && !(key_u & 0x80000000u)) ? (int)(key_u >> 16) : 0x7fffffff;
            const int pa = mask ? base + (int)__builtin_ctzll(mask) : 0x7fffffff;
            const int pb = pnext < base + 64 ? pnext : 0x7fffffff;
            const int ppos = min(pa, pb);
            if (ppos == 0x7fffffff) break;
            const int ln = __builtin_amdgcn_readfirstlane(ppos - base);
            mask &= ~(1ull << ln);
            AgeCand t;
            if (ppos == pb) {  // its own pairs first
              AgeCand ys;
              t = AgeCand{INF, (double)INF, MM_AGE_EMPTY, -1, -1};
              while (e < m && !(key & 0x80000000u) && (int)(key >> 16) == ppos) {
                const int x = (int)(xy >> 16), y = (int)(xy & 0xffffu);
                const float symc = sym;
                e++;
                if (e < m) pair_at(side, e, key, xy, sym);
                apply_ages(x, y, symc, t, ys);
              }
            } else {
              const long long bits = __double_as_longlong(r.d2);
              const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffll), ln);
              const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), ln);
              t.d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.d), ln));
              t.d2 = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
              t.lv = __builtin_amdgcn_readlane(r.lv, ln);
              t.l1 = __builtin_amdgcn_readlane(r.l1, ln);
              t.l2 = __builtin_amdgcn_readlane(r.l2, ln);
            }
            best_takes(t);
          }
        }
        // the merged cluster: new minima, an empty slot, its pairs (:883-963 / :2343-2353), the last look of the best
        const int szi = st.csz[i], szj = st.csz[j];
        const int li = st.alv[i], lj = st.alv[j];
        if (lane == 0) {
          st.mv[j] = min_value_j;
          if (p.has_prior) st.mvcf[j] = mvcf_j;
          st.mcd[j] = INF;
          st.d2d[j] = (double)INF;
          st.lv[j] = (hidx_t)MM_AGE_EMPTY;
        }
        AgeCand js{INF, (double)INF, MM_AGE_EMPTY, -1, -1};
        while (e < m) {
          const int x = (int)(xy >> 16);
          const float symc = sym;
          e++;
          if (e < m) pair_at(side, e, key, xy, sym);
          AgeCand xs;
          apply_ages(x, j, symc, xs, js);
        }
        if (reduced) {
          if (js.lv <= a_lw && js.d < INF && (abest.d > js.d || (abest.d == js.d && abest.d2 > js.d2))) abest = js;
        } else {
          best_takes(js);
        }
        // ages[j], the clock (:1219-1226 / :2513-2524: with a prior it steps before the lineage count drops)
        const int newl = max(li, lj);
        const double before = a_age;
        if (p.has_prior) a_age = __dadd_rn(a_age, ages_step(a_lins, p.Ne));
        a_lins--;
        while (a_level < newl) {
          a_level++;
          a_lins += p.ages_count[a_level];
        }
        if (!p.has_prior) a_age = __dadd_rn(a_age, ages_step(a_lins, p.Ne));
        if (!(a_age >= before)) a_lw = -1;  // (0 or 1 lineages: the step is not a number the clock can add)
        clock_reaches();
        if (lane == 0) {
          st.alv[j] = (cidx_t)newl;
          sh.a_lw = a_lw;
          sh.best.dist = abest.d;
          sh.best.lin1 = abest.l1;
          sh.best.lin2 = abest.l2;
          st.csz[j] = (cidx_t)(szi + szj);
          sh.nupd = 0;
          sh.npairs = 0;
        }
      }
    } else if (wave == 0) {
      // the clusters whose candidates change, at their turn: after their own pairs
      float sd = INF, sd2 = INF;
      int sl1 = -1, sl2 = -1, spos = n;
      int cur = -1, curk = -1;
      auto settle = [&]() {
        const float d1 = st.mcd[curk], d2 = st.mcd2[curk];
        const int c1 = st.lin1[curk], c2 = st.lin2[curk];
        if (sd > d1 || (sd == d1 && sd2 > d2)) {
          sd = d1;
          sd2 = d2;
          sl1 = c1;
          sl2 = c2;
          spos = cur;
        }
      };
      int e = 0;
      unsigned key = 0, xy = 0;
      float sym = 0.0f;
      if (m > 0) pair_at(side, 0, key, xy, sym);
      for (; e < m; e++) {
        if (key & 0x80000000u) break;
        const int pos = (int)(key >> 16), x = (int)(xy >> 16), y = (int)(xy & 0xffffu);
        const float symc = sym;
        if (e + 1 < m) pair_at(side, e + 1, key, xy, sym);  // the next record, requested along with this pair's reads
        if (pos != cur) {
          if (cur >= 0) settle();
          cur = pos;
          curk = x;
        }
        apply(x, y, symc);
      }
      if (cur >= 0) settle();
      // the loop's running "best": smallest (dist, dist2), the earliest cluster among exact ties
      Best b;
      b.dist = INF;
      b.dist2 = INF;
      b.lin1 = sh.best.lin1;
      b.lin2 = sh.best.lin2;
      int pos = n;
      if (upos < n) {
        b.dist = ud;
        b.dist2 = ud2;
        b.lin1 = bl1;
        b.lin2 = bl2;
        pos = upos;
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
      if (lane == 0) {
        st.mv[j] = min_value_j;
        if (p.has_prior) st.mvcf[j] = mvcf_j;
        st.mcd[j] = INF;
        st.mcd2[j] = INF;
      }
      for (; e < m; e++) {
        const int x = (int)(xy >> 16);
        const float symc = sym;
        if (e + 1 < m) pair_at(side, e + 1, key, xy, sym);
        apply(x, j, symc);
      }
      const float jd = st.mcd[j], jd2 = st.mcd2[j];
      if (b.dist > jd || (b.dist == jd && b.dist2 > jd2)) {
        b.dist = jd;
        b.dist2 = jd2;
        b.lin1 = st.lin1[j];
        b.lin2 = st.lin2[j];
      }
      if (lane == 0) {
        sh.best = b;
        st.csz[j] = (cidx_t)(int)added;  // (exact integers in floats)
        sh.nupd = 0;
        sh.npairs = 0;
      }
    }
    LAP(6);
    // -- the same merge in the symmetric matrix once it is in use (coalesce_sym, tree_builder.cpp:968-1058)
    if (sh.use_sym) {
      if (tid == 0) sh.count = 0;  // (thread 0 is past its ordered part; the others wait at the barrier below)
      __syncthreads();
      for (int ik = tid; ik < n; ik += MM_BLOCK) {
        const int k = st.ci[ik];
        if (k == j || k == i) continue;
        // (the symmetric matrix is symmetric bit for bit -- a float sum commutes, and the two updates below
        //  apply one formula to equal operands --: the column entries are read from the rows of i and j)
        const float dik = SS(i, k), djk = SS(j, k), dkj = djk, dki = dik;
        const float mvk = p.min_values_sym[k];
        if (dik != djk) SS(j, k) = (csi * dik + csj * djk) / added;
        if (dki != dkj) SS(k, j) = (csi * dki + csj * dkj) / added;
        if (dkj != dki) {
          if ((double)fabsf(mvk - dkj) < 1e-6 || (double)fabsf(mvk - dki) < 1e-6)
            p.upd_pos[atomicAdd(&sh.count, 1)] = ik;  // rows to rescan
        } else {
          if (p.mcs_lin1[k] == i) p.mcs_lin1[k] = j;
          if (p.mcs_lin2[k] == i) p.mcs_lin2[k] = j;
        }
      }
      __syncthreads();
      const int nres_s = sh.count;
      for (int r = 0; r < nres_s; r++) {
        const int k = st.ci[p.upd_pos[r]];
        const float old = p.min_values_sym[k];
        auto row = symm + (size_t)k * N;
        float fm = INF, fm2 = 0.0f;
        int fpos = n, pos_old = n, pos_less = n;
        for (int il = tid; il < n; il += MM_BLOCK) {
          const int l = st.ci[il];
          if (l != i && l != k) {
            const float x = row[l];
            if (x < fm) {
              fm = x;
              fpos = il;
            }
            if (x == old) pos_old = min(pos_old, il);
            if (x < old) pos_less = min(pos_less, il);
          }
        }
        float dummy = 0.0f;
        block_min3(dummy, pos_old, pos_less, sh.wave_f, sh.wave_i, sh.wave_i2);
        block_lex_min(fm, fm2, fpos, sh.wave_f, sh.wave_f2, sh.wave_i3);
        if (tid == 0) {
          const bool stops = pos_old < n && pos_old < pos_less;
          const float x = stops ? old : fm;
          const int at = stops ? pos_old : fpos;
          p.min_values_sym[k] = x;
          p.mcs_dist[k] = x;
          if (x < INF) {
            p.mcs_lin1[k] = k;
            p.mcs_lin2[k] = st.ci[at];
          }
        }
      }
      __syncthreads();
      float b1 = INF, b2 = 0.0f, mj = INF, mj2 = 0.0f;
      int bp = n, mjp = n;
      auto srowj = symm + (size_t)j * N;
      for (int ik = tid; ik < n; ik += MM_BLOCK) {
        const int k = st.ci[ik];
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
          const int k = st.ci[bp];
          b.dist = b1;
          b.lin1 = p.mcs_lin1[k];
          b.lin2 = p.mcs_lin2[k];
        }
        p.min_values_sym[j] = mj;
        p.mcs_dist[j] = mj;  // (INF when nothing is left)
        if (mjp < n && mj < INF) {
          p.mcs_lin1[j] = st.ci[mjp];
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
    if (sym_now) {
      __syncthreads();
      erase_read();
      __syncthreads();
      erase_write();
    }
    LAP(7);
    __syncthreads();
    LAP(8);
  }
  if (bail < 0) {
    if constexpr (LAY != L_GLOBAL) {  // the carried state goes out (pinned host memory)
      for (int c = tid; c < N; c += MM_BLOCK) {
        p.io_lin[c] = (int)st.lin1[c];
        p.io_lin[(size_t)N + c] = (int)st.lin2[c];
        p.io_mvcf[c] = st.mvcf[c];
      }
    }
    if (tid == 0 && p.timers) {
      sh.tacc[11] = (clock64() - cstart) * 100 / (wall_clock64() - tstart + 1);  // shader clock, MHz
      for (int x = 0; x < 16; x++) p.timers[x] = sh.tacc[x];
    }
    bail = 0;
  }
  __threadfence_system();  // every wave's part of the tree and of the carried state
  __syncthreads();
  leave(bail);
#undef LAP
}

// A tree's parameters as a worker reads them from the queue: one 32-bit word per thread, uncached; from LDS every
// word goes through readfirstlane, so the builder keeps them in scalar registers as it did when they were kernel
// arguments.
__device__ inline void params_from_words(MMParamsDev &p, const unsigned *w) {
  unsigned *out = reinterpret_cast<unsigned *>(&p);
#pragma unroll
  for (int x = 0; x < (int)(sizeof(MMParams) / 4); x++) out[x] = __builtin_amdgcn_readfirstlane(w[x]);
}

// The builders' requests of one device and tree size, in pinned host memory: the host writes a request's words, then
// `tail`; workers claim tickets from `head` (device memory) and read the words of their ticket.
constexpr int MM_STAGING = 32;      // staging pairs (row-major distance matrix + clade prior of a tree, K3 -> weave) per device
constexpr int MM_QUEUE_CAP = 1024;  // requests in flight <= builders alive (a section has one tree in flight)
constexpr int MM_LAUNCHES = 6;      // worker launches alive at once (a stream, i.e. a hardware queue, each)
constexpr int MM_XCDS = 8;          // a launch's workgroups are dealt to the XCDs in turn
constexpr int MM_WHOLE_ROUNDS = 64; // from this many workers on, launches and the goal are whole rounds of the XCDs
struct WorkQueue {
  unsigned tail;
  unsigned pad[15];
  unsigned gone[16];  // per launch: workers that have left (the host counts a launch's LIVE workers, not its size)
  unsigned words[MM_QUEUE_CAP][MM_PARAM_WORDS];
};
// device memory: the ticket counter, and per launch how many of its workers are building and when one last was
struct WorkerState {
  unsigned head;
  unsigned pad[15];
  struct {
    int busy;           // workers of the launch that are building
    unsigned activity;  // counts its claims and finished trees
    long long pad2[7];
  } launch[MM_LAUNCHES];
};
static_assert(sizeof(MMParams) % 4 == 0 && sizeof(MMParams) / 4 <= MM_PARAM_WORDS, "MM_PARAM_WORDS");

// One workgroup = one WORKER: it takes the next tree of the queue, builds it, says so in the request's own word of
// pinned memory and takes the next -- a tree starts the moment a worker is free instead of waiting for a launch to
// gather, and no launch lasts as long as its slowest tree.  The workers of a launch leave TOGETHER, when none of
// them has had a tree for `idle_ticks` (100 MHz): a launch holds its stream -- one of a few hardware queues -- until
// its last workgroup is gone, so workers that trickled away one by one would leave streams occupied by a few
// stragglers and no way to bring the others back.  Nothing a worker waits for can fail to arrive: the only loop
// without a tree in it is the idle one, bounded by the clock.
// OCC workgroups per CU: 2 * OCC waves per SIMD, 256 / OCC registers per lane (the plain build at N <= 5120: two).
template <int LAY, int MAXQ, bool AGES, int OCC>
__global__ void __launch_bounds__(MM_BLOCK) __attribute__((amdgpu_waves_per_eu(2 * OCC, 2 * OCC)))
    minmatch_worker(WorkQueue *q, WorkerState *ws, int launch, long long idle_ticks, int trace) {
  __shared__ Shared sh;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  const int tid = threadIdx.x;
  auto &mine = ws->launch[launch];
  // (progress marks for RELATE_AMD_TIMING=2, in the queue's spare words: workers started / tickets claimed / trees
  //  left / workers gone)
  auto mark = [&](int which) {
    if (trace && tid == 0) __hip_atomic_fetch_add(&q->pad[which], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  mark(0);
  // (one divergent block per turn -- thread 0 settles the accounts of the tree just built and claims the next --,
  //  then barriers and wave-uniform branches only: what follows an `if (tid == 0)` behind the build is laid out by
  //  the compiler as a loop over groups of lanes, and a barrier inside it is passed twice by thread 0's wave)
  bool had_tree = false;
  for (;;) {
    if (tid == 0) {
      if (had_tree) {
        if (trace) __hip_atomic_fetch_add(&q->pad[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_add(&mine.activity, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&mine.busy, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      unsigned ticket = ~0u;
      // (idle time is measured on this workgroup's own clock, from the last change of the launch's activity count it
      //  has seen: the counters of different XCDs are not one clock)
      unsigned seen = __hip_atomic_load(&mine.activity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long long since = wall_clock64();
      for (;;) {
        const unsigned h = __hip_atomic_load(&ws->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t = __hip_atomic_load(&q->tail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(t - h) > 0) {
          unsigned expected = h;
          if (__hip_atomic_compare_exchange_strong(&ws->head, &expected, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT)) {
            ticket = h;
            __hip_atomic_fetch_add(&mine.busy, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&mine.activity, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (trace) __hip_atomic_fetch_add(&q->pad[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
          continue;
        }
        const unsigned act = __hip_atomic_load(&mine.activity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long now = wall_clock64();
        if (act != seen) {
          seen = act;
          since = now;
        } else if (now - since > idle_ticks &&
                   __hip_atomic_load(&mine.busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          break;
        }
        __builtin_amdgcn_s_sleep(127);  // (~3 us: the queue is read across PCIe)
      }
      if (trace && ticket == ~0u) __hip_atomic_fetch_add(&q->pad[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (ticket == ~0u) __hip_atomic_fetch_add(&q->gone[launch], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      sh.ticket = ticket;
    }
    __syncthreads();
    const unsigned ticket = __builtin_amdgcn_readfirstlane(sh.ticket);
    if (ticket == ~0u) break;
    if (tid < MM_PARAM_WORDS)
      sh.praw[tid] = __hip_atomic_load(&q->words[ticket % MM_QUEUE_CAP][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // what the tree's inputs were written with -- kernels of the builder's stream, finished before the request was
    // published -- may sit in other XCDs' L2 or stale in this one's: acquire at system scope, every wave
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    MMParamsDev p;
    params_from_words(p, sh.praw);
    // rows of rebuilt clusters per pass of a merge: two while the registers allow (2: 110 ms per N = 5000 tree, 3: 119,
    // 4: 134 with 256 registers per lane)
    // (three / four rows, possible since a row costs 10 registers instead of 20: 114.2 / 114.4 ms per tree against 115.0
    //  -- the tests gain what the ordered part loses to spills: profiles/r06_rows_per_pass.json)
    build_tree<LAY, MAXQ, AGES, (OCC > 1 || MAXQ > MM_Q_LDS) ? 1 : 2>(p, sh, dyn);
    __syncthreads();  // (sh is the next tree's)
    had_tree = true;
  }
}

// ---- the per-tree kernels around the build: every tree pays them on the CUs the workers leave, next to RePaint, so
// they are counted in bytes.  Round 5: four passes over the matrices instead of six --
//   rowmin_penalty_kernel  the carrier penalty of a tree's distance matrix (in place) AND its row minima, one pass;
//   prior_kernel           the clade prior row by row AND its row minima (the row is in LDS anyway);
//   weave_kernel           M[a][b] = (d(a,b), d(b,a), cf(a,b), cf(b,a)) AND the pair scan of the initialisation (the
//                          row minima are complete by then): every element of D / CF read once, M written once and
//                          never read back.
// Before: penalty (in place), prior, pack (read 2 x 2 matrices, write M), rowmin (read both matrices again), pairscan
// (read half of M): 0.42 ms of kernels and ~1.5 GB per tree at N = 5000; now ~1.1 GB.

// Carrier penalty of AncesTreeBuilder::BuildTopology (anc_builder.cpp:563-581): every entry of a carrier's row gets
// + val, and - val again where the column is a carrier too (the same two operations in the same order per entry as
// the host loop); member == nullptr: no penalty (the first tree of a section, --no_consistency).  And, of the row as
// it then is, out[a] = min over l != a: the row minima of tree_builder.cpp:1659-1666.  A workgroup per row.
__global__ void __launch_bounds__(256) rowmin_penalty_kernel(float *__restrict__ D, int N,
                                                            const unsigned char *__restrict__ member, float val,
                                                            float *__restrict__ out) {
  __shared__ float part[4];
  const int a = blockIdx.x;
  float *row = D + (size_t)a * N;
  const bool carrier = member && member[a];
  float mv = INFINITY;
  for (int col = threadIdx.x; col < N; col += 256) {
    float x = row[col];
    if (carrier) {
      x = x + val;
      if (member[col]) x -= val;
      row[col] = x;
    }
    if (col != a) mv = fminf(mv, x);
  }
  mv = wave_min_f(mv);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mv;
  __syncthreads();
  if (threadIdx.x == 0) out[a] = fminf(fminf(part[0], part[1]), fminf(part[2], part[3]));
}

// The woven matrix M[a][b] = (d(a,b), d(b,a), cf(a,b), cf(b,a)) from the row-major matrices (CF may be null) and, in
// the same pass, the pair scan of MinMatch's initialisation (tree_builder.cpp:1690-1722): row a's partners b > a
// with d(a,b) <= min_a and d(b,a) <= min_b (minima + threshold), in order, with the symmetric distance the
// reference's loop would compute for the pair (0 when the pair is also mutually closest under the prior,
// :1699-1702).  mvcf_old: min_values_CF as carried over from the previous tree (:2399-2400).
// One workgroup per strip of 32 rows, walking the column panels of 64 from its own on, in order (so a row's hits
// come out in order): wave w holds rows a0 + 8 w + r, lane l column 64 pb + l.  d(a,b) and cf(a,b) arrive as the lanes need
// them (a wave reads 256 contiguous bytes of a row), d(b,a) and cf(b,a) -- 64 rows b, 32 columns -- through LDS;
// a wave's store is 1 KB of M's panel.  The next panel's loads are in flight while this one is woven.
constexpr int WV_ROWS = 32;
// (Every strip walking ALL panels and weaving its own rows only -- each element read twice, no 512-byte runs -- was
//  within the noise of this in the C3 stage: removed, DESIGN_NOTES.md 11.)
template <bool PRIOR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) weave_kernel(const float *__restrict__ D, const float *__restrict__ CF,
                                                    float4 *__restrict__ M, int N, const float *__restrict__ rowmin_D,
                                                    const float *__restrict__ rowmin_CF,
                                                    const float *__restrict__ mvcf_old, int ages, float threshold,
                                                    float threshold_CF, int *__restrict__ hit_cnt,
                                                    unsigned *__restrict__ hit_b, float *__restrict__ hit_sym) {
  __shared__ float tD[64][WV_ROWS + 1], tC[PRIOR ? 64 : 1][WV_ROWS + 1];  // [b][a]: d(b,a), cf(b,a)
  __shared__ float tX[WV_ROWS][64 + 1], tZ[PRIOR ? WV_ROWS : 1][64 + 1];   // [a][b]: d(a,b), cf(a,b)
  // (the wave's index as a scalar: its rows' addresses are scalar bases, the lane's column a 32-bit offset)
  const int a0 = blockIdx.x * WV_ROWS, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int trow = threadIdx.x >> 2, tcol = (threadIdx.x & 3) * 8;  // the thread's 8 floats of the transposed tile
  typedef const __attribute__((address_space(1))) float *GF;
  typedef const __attribute__((address_space(1))) f32x4 *GF4;
  const GF Dg = (GF)D, CFg = (GF)CF;
  const bool vec = (N & 3) == 0;                                     // (rows 16-byte aligned: float4 loads)
  float mva[8], mvcfa[8];
  int cnt[8];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int a = a0 + wave * 8 + r;
    cnt[r] = 0;
    mva[r] = a < N ? rowmin_D[a] + threshold : 0.0f;
    mvcfa[r] = 0.0f;
    if (PRIOR && a < N) {
      const float old = mvcf_old[a], mc_ = rowmin_CF[a];
      mvcfa[r] = (old > mc_ ? mc_ : old) + threshold_CF;
    }
  }
  struct Panel {
    float x[8], z[8];    // d(a,b), cf(a,b) of the wave's rows at the lane's column
    float td[8], tc[8];  // the thread's piece of the transposed tile: d(b', a0 + tcol ..), cf likewise
  };
  auto load = [&](int pb, Panel &q) {
    const unsigned b = (unsigned)(pb * 64 + lane);
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int a = a0 + wave * 8 + r;  // (wave-uniform)
      const bool in = a < N && (int)b < N;
      const GF drow = Dg + (size_t)a * N, crow = CFg + (size_t)a * N;
      q.x[r] = in ? drow[b] : 0.0f;
      q.z[r] = (PRIOR && in) ? crow[b] : 0.0f;
    }
    const int bb = pb * 64 + trow;
    const unsigned toff = (unsigned)bb * (unsigned)N + (unsigned)(a0 + tcol);  // (N <= 10240: N * N < 2^32)
    if (vec && bb < N && a0 + tcol + 8 <= N) {
      const GF4 s4 = (GF4)(Dg + toff);
      const f32x4 u = s4[0], v = s4[1];
      q.td[0] = u.x, q.td[1] = u.y, q.td[2] = u.z, q.td[3] = u.w, q.td[4] = v.x, q.td[5] = v.y, q.td[6] = v.z, q.td[7] = v.w;
      if (PRIOR) {
        const GF4 c4 = (GF4)(CFg + toff);
        const f32x4 cu = c4[0], cv = c4[1];
        q.tc[0] = cu.x, q.tc[1] = cu.y, q.tc[2] = cu.z, q.tc[3] = cu.w, q.tc[4] = cv.x, q.tc[5] = cv.y, q.tc[6] = cv.z, q.tc[7] = cv.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const bool in = bb < N && a0 + tcol + k < N;
        q.td[k] = in ? Dg[toff + k] : 0.0f;
        q.tc[k] = (PRIOR && in) ? CFg[toff + k] : 0.0f;
      }
    }
  };
  // Every element of D and CF is read ONCE: the strip walks the panels from its own on (pa), and a tile right of
  // the diagonal panel is woven both ways -- M[a][b] for the strip's rows, a wave's store 1 KB of panel pb, and
  // M[b][a] = (d(b,a), d(a,b), cf(b,a), cf(a,b)) for the panel's 64 rows b at the strip's 32 columns, 512 B runs of
  // panel pa -- so nobody reads the tiles left of its diagonal.
  const int pa = a0 / 64;
  auto weave = [&](int pb, const Panel &q) {
    const bool both = pb > pa;  // (wave-uniform)
#pragma unroll
    for (int k = 0; k < 8; k++) {
      tD[trow][tcol + k] = q.td[k];
      if (PRIOR) tC[trow][tcol + k] = q.tc[k];
    }
    if (both) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        tX[wave * 8 + r][lane] = q.x[r];
        if (PRIOR) tZ[wave * 8 + r][lane] = q.z[r];
      }
    }
    __syncthreads();
    const int b = pb * 64 + lane;
    const float mvb = b < N ? rowmin_D[b] + threshold : 0.0f;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int a = a0 + wave * 8 + r;
      const float y = tD[lane][wave * 8 + r], w = PRIOR ? tC[lane][wave * 8 + r] : 0.0f;
      const bool in = a < N && b < N;
      if (in) M[mm_index(a, b, N)] = make_float4(q.x[r], y, q.z[r], w);
      const bool hit = in && b > a && mva[r] >= q.x[r] && mvb >= y;
      const unsigned long long m = __ballot(hit);
      if (m) {  // (wave-uniform, rare)
        const int at = cnt[r] + __popcll(m & ((1ull << lane) - 1ull));
        if (hit && at < MM_HITS) {
          float sym = y + q.x[r];
          if (PRIOR) {
            const float old = mvcf_old[b], mc_ = rowmin_CF[b];
            const float mvcf_b = (old > mc_ ? mc_ : old) + threshold_CF;
            // (with sample ages the initialisation keeps the pairs the prior agrees with and voids the others, :1792-1797)
            const bool agrees = q.z[r] <= mvcfa[r] && w <= mvcf_b;
            if (ages ? !agrees : agrees) sym = ages ? INFINITY : 0.0f;
          }
          hit_b[(size_t)a * MM_HITS + at] = (unsigned)b;
          hit_sym[(size_t)a * MM_HITS + at] = sym;
        }
        cnt[r] += __popcll(m);
      }
    }
    if (both) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int bl = wave * 16 + 2 * k + (lane >> 5), al = lane & 31;
        const int bq = pb * 64 + bl, aq = a0 + al;
        if (aq < N && bq < N)
          M[mm_index(bq, aq, N)] = make_float4(tD[bl][al], tX[al][bl], PRIOR ? tC[bl][al] : 0.0f, PRIOR ? tZ[al][bl] : 0.0f);
      }
    }
    __syncthreads();
  };
  const int P = (N + 63) / 64;
  Panel p0, p1;
  load(pa, p0);
  for (int pb = pa; pb < P; pb += 2) {
    if (pb + 1 < P) load(pb + 1, p1);
    weave(pb, p0);
    if (pb + 1 < P) {
      if (pb + 2 < P) load(pb + 2, p0);
      weave(pb + 1, p1);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int a = a0 + wave * 8 + r;
      if (a < N) hit_cnt[a] = cnt[r];
    }
  }
}

// Clade prior from the previous tree (treeseq.cpp: clade_prior; anc_builder.cpp:583-606): row a holds, for
// every other leaf b, acc[depth(parent(a)) - depth(mrca(a,b))]; the leaves under the sibling of every ancestor
// step are a contiguous range of the depth-first leaf order.
__global__ void __launch_bounds__(256) prior_kernel(float *__restrict__ CF, int N, const int *__restrict__ parent,
                             const int *__restrict__ child_left, const int *__restrict__ child_right,
                             const int *__restrict__ depth, const int *__restrict__ lo,
                             const int *__restrict__ size, const int *__restrict__ order,
                             const float *__restrict__ acc, float *__restrict__ rowmin) {
  // The row is put together in LDS -- the ranges are contiguous in depth-first order, the entries they name are
  // scattered over the row (order[q]): as 4-byte stores to HBM the kernel wrote its 100 MB at 0.4 TB/s, 0.24 ms per
  // tree at N = 5000 and the largest of the per-tree kernels that share the chip with RePaint -- and leaves the block
  // in one coalesced pass.  (Every leaf but a itself lies under exactly one sibling on the way up: all of the row is
  // written.)
  extern __shared__ float prior_row[];  // N floats
  const int a = blockIdx.x;
  if (threadIdx.x == 0) prior_row[a] = 0.0f;
  const int da = depth[parent[a]];
  int child = a;
  for (int v = parent[a]; v >= 0; child = v, v = parent[v]) {
    const int other = child_left[v] == child ? child_right[v] : child_left[v];
    const float x = acc[da - depth[v]];
    const int b = lo[other], e = b + size[other];
    for (int q = b + threadIdx.x; q < e; q += blockDim.x) prior_row[order[q]] = x;
  }
  __syncthreads();
  float *__restrict__ row = CF + (size_t)a * N;
  float mv = INFINITY;  // the row's minimum off the diagonal (tree_builder.cpp:1659-1666), while the row is here
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const float x = prior_row[j];
    row[j] = x;
    if (j != a) mv = fminf(mv, x);
  }
  __shared__ float part[4];
  mv = wave_min_f(mv);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mv;
  __syncthreads();
  if (threadIdx.x == 0) rowmin[a] = fminf(fminf(part[0], part[1]), fminf(part[2], part[3]));
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
// The worker kernel for trees of N leaves: where the state lives, register slots per thread, workgroups per CU.
//   plain build: L_HOT for every N -- 13 bytes of LDS per cluster next to ~14 KB of lists: two workgroups per CU up to
//                N = 5120 (4 slots per thread up to N = 2048, 10 above), one with 20 slots up to N = 10,240;
//   AGES build:  everything in LDS (33 bytes per cluster) up to N = 4100, in global memory above.
struct WorkerKind {
  int layout, maxq, occ;
  bool ages;
  const void *fn;
  size_t dyn;  // dynamic LDS per workgroup
};
static size_t lds_state_bytes(int layout, int N) {
  const size_t per = layout == L_LDS ? 33 : layout == L_HOT ? 13 : layout == L_WARM ? 21 : 0;
  return (per * (size_t)N + 255) & ~(size_t)255;
}
static size_t static_lds(const void *fn) {
  hipFuncAttributes a;
  return hipFuncGetAttributes(&a, fn) == hipSuccess ? a.sharedSizeBytes : (size_t)1 << 30;
}
static WorkerKind worker_kind(int N, bool ages) {
  const size_t lds_cu = (size_t)160 * 1024;
  WorkerKind k{};
  k.ages = ages;
  if (ages) {
    const void *f_lds = reinterpret_cast<const void *>(&minmatch_worker<L_LDS, MM_Q_LDS, true, 1>);
    if (N <= MM_Q_LDS * MM_BLOCK && static_lds(f_lds) + lds_state_bytes(L_LDS, N) <= lds_cu)
      k = WorkerKind{L_LDS, MM_Q_LDS, 1, true, f_lds, lds_state_bytes(L_LDS, N)};
    else if (N <= MM_Q_LDS * MM_BLOCK)
      k = WorkerKind{L_GLOBAL, MM_Q_LDS, 1, true, reinterpret_cast<const void *>(&minmatch_worker<L_GLOBAL, MM_Q_LDS, true, 1>), 0};
    else
      k = WorkerKind{L_GLOBAL, MM_Q_GLOB, 1, true, reinterpret_cast<const void *>(&minmatch_worker<L_GLOBAL, MM_Q_GLOB, true, 1>), 0};
    return k;
  }
  const size_t dyn = lds_state_bytes(L_HOT, N);
  // Two workgroups per CU (128 registers per lane) hold half the CUs for the same workers -- the default where HBM
  // holds hundreds of sections' trees at once (small N; no measurable difference on a C4 chunk either way); at
  // N = 5000 a stage has ~134 sections open and a tree's latency decides: one per CU at 256 registers per lane, 1.33 x
  // faster per tree (profiles/r06_builder_many.jsonl, r06_c4_chunk_occ.json).  RELATE_AMD_BUILD_OCC=1 / 2 decides
  // otherwise.
  static const int occ_env = getenv("RELATE_AMD_BUILD_OCC") ? atoi(getenv("RELATE_AMD_BUILD_OCC")) : 0;
  const int occ_max = occ_env > 0 ? occ_env : (N <= MM_Q_SMALL * MM_BLOCK ? 2 : 1);
  if (N <= MM_Q_SMALL * MM_BLOCK && occ_max >= 2) {
    const void *f = reinterpret_cast<const void *>(&minmatch_worker<L_HOT, MM_Q_SMALL, false, 2>);
    if (2 * (static_lds(f) + dyn) <= lds_cu) return WorkerKind{L_HOT, MM_Q_SMALL, 2, false, f, dyn};
  }
  if (N <= MM_Q_LDS * MM_BLOCK && occ_max >= 2) {
    const void *f = reinterpret_cast<const void *>(&minmatch_worker<L_HOT, MM_Q_LDS, false, 2>);
    if (2 * (static_lds(f) + dyn) <= lds_cu) return WorkerKind{L_HOT, MM_Q_LDS, 2, false, f, dyn};
  }
  if (N <= MM_Q_LDS * MM_BLOCK) {  // one per CU: the row minima in LDS too
    const void *f = reinterpret_cast<const void *>(&minmatch_worker<L_WARM, MM_Q_LDS, false, 1>);
    return WorkerKind{L_WARM, MM_Q_LDS, 1, false, f, lds_state_bytes(L_WARM, N)};  // (107 + 15 KB at N = 5120)
  }
  return WorkerKind{L_HOT, MM_Q_GLOB, 1, false, reinterpret_cast<const void *>(&minmatch_worker<L_HOT, MM_Q_GLOB, false, 1>), dyn};
}
// workgroups of a tree's worker kernel a CU holds
int device_builder_workers_per_cu(int N, bool ages) { return worker_kind(N, ages).occ; }

static int env_int(const char *name, int fallback, int lo, int hi) {
  const char *e = getenv(name);
  return e ? std::max(lo, std::min(hi, atoi(e))) : fallback;
}

// What the builders of one device share.
//
// * Staging: the row-major distance matrix (K3 writes it, the carrier penalty edits it) and clade prior of a tree
//   are needed from K3 until the weave (weave_kernel) -- ~30 ms of a tree's ~0.2 s with a hundred sections sharing
//   the hardware queues: 32 pairs (12 were a queue of their own at 134 sections: 11 s of every section's 60).  A builder takes a pair from
//   a small pool for that time instead of owning 8 N^2 bytes for life (200 MB of the 725 MB a section used to pin
//   at N = 5000: why 91 sections were all that fitted).
// * The symmetric matrix of the fallback (tree_builder.cpp:255-293): a pool of `sym_slots` matrices the build
//   kernels take and give back themselves (MMParams::sym_locks).
class DeviceShare {
 public:
  struct Staging {
    DevBuf D, CF;
  };
  static DeviceShare &of(int device) {
    static std::mutex gm;
    static std::vector<DeviceShare *> all;
    std::lock_guard<std::mutex> lk(gm);
    if ((int)all.size() <= device) all.resize(device + 1, nullptr);
    if (!all[device]) all[device] = new DeviceShare();  // lives as long as the process
    return *all[device];
  }
  // a staging pair with room for N x N floats each (blocks while all are out); nullptr: no memory
  Staging *take(int N) {
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
      if (!free_.empty()) {
        Staging *s = free_.back();
        free_.pop_back();
        lk.unlock();
        if (s->D.alloc((size_t)N * N * 4) || s->CF.alloc((size_t)N * N * 4)) {
          give(s);
          return nullptr;
        }
        return s;
      }
      if (made_ < cap_) {
        made_++;
        lk.unlock();
        Staging *s = new Staging();
        if (s->D.alloc((size_t)N * N * 4) || s->CF.alloc((size_t)N * N * 4)) {
          delete s;
          lk.lock();
          made_--;
          if (made_ == 0) return nullptr;  // not even one: out of memory
          cap_ = made_;                    // what there is has to do
          continue;
        }
        return s;
      }
      cv_.wait(lk);
    }
  }
  void give(Staging *s) {
    {
      std::lock_guard<std::mutex> lk(m_);
      free_.push_back(s);
    }
    cv_.notify_one();
  }
  // a pair that is free or can still be made; nullptr rather than waiting
  Staging *take_if_any(int N) {
    {
      std::lock_guard<std::mutex> lk(m_);
      if (free_.empty() && made_ >= cap_) return nullptr;
    }
    return take(N);
  }
  // the pool of symmetric matrices for trees of N leaves (grown on demand, never shrunk); *slots = 0: none
  void sym_pool(int N, float **base, int **locks, int *slots) {
    std::lock_guard<std::mutex> lk(m_);
    if (sym_n_ < N) {
      int want = env_int("RELATE_AMD_TEST_SYM_SLOTS", 8, 0, 64);
      DevBuf nb, nl;
      while (want > 0 && nb.alloc((size_t)want * N * N * 4)) want /= 2;
      if (want > 0 && !nl.alloc(64 * 4) && hipMemset(nl.p, 0, 64 * 4) == hipSuccess) {
        // (builds in flight hold pointers into the old pool: it is kept, not freed)
        retired_.emplace_back(new DevBuf());
        retired_.back()->swap(sym_);
        retired_.emplace_back(new DevBuf());
        retired_.back()->swap(sym_locks_);
        sym_.swap(nb);
        sym_locks_.swap(nl);
        sym_n_ = N;
        sym_slots_ = want;
      }
    }
    *base = sym_n_ >= N ? sym_.as<float>() : nullptr;
    *locks = sym_locks_.as<int>();
    *slots = sym_n_ >= N ? sym_slots_ : 0;
  }
  // every staging pair and the symmetric pool now (the stage admits its sections against what is free after this)
  int prefill(int N) {
    std::vector<Staging *> got;
    for (int x = 0; x < cap_; x++) {
      Staging *s = take_if_any(N);
      if (!s) break;
      got.push_back(s);
    }
    const bool any = !got.empty();
    for (Staging *s : got) give(s);
    float *b;
    int *l, n;
    sym_pool(N, &b, &l, &n);
    return any ? 0 : -1;
  }
  // bytes of HBM the shared pools take for trees of N leaves (the stage's admission counts them once)
  static double bytes(int N) {
    return (double)MM_STAGING * 8.0 * N * N +
           (double)env_int("RELATE_AMD_TEST_SYM_SLOTS", 8, 0, 64) * 4.0 * N * N;
  }

 private:
  DeviceShare() : cap_(MM_STAGING) {}
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<Staging *> free_;
  int made_ = 0, cap_;
  DevBuf sym_, sym_locks_;
  std::vector<std::unique_ptr<DevBuf>> retired_;
  int sym_n_ = 0, sym_slots_ = 0;
};
double device_builder_shared_bytes(int N) { return DeviceShare::bytes(N); }
int device_builder_reserve_shared(int device, int N) {
  if (hipSetDevice(device) != hipSuccess) return -1;
  return DeviceShare::of(device).prefill(N);
}

// The builders of one device and one tree size hand their trees to WORKERS -- resident workgroups that pull
// requests from a queue in pinned host memory (minmatch_worker).  Why not a launch per tree, or per batch of
// trees: a process has a dozen hardware queues and kernels of one queue run one after the other, so 150 sections
// with a stream each build a dozen trees at a time; batched launches (round 2) waited 10 ms to gather, lasted as
// long as their slowest tree and kept at most 4 x ~15 trees in flight on 256 CUs.
// The launcher thread adds workers when trees wait: up to MM_LAUNCHES launches alive (a lowest-priority stream
// each; more than ~20 hardware queues in all and the device time-slices them), each at least half as large as
// what is alive -- 16, 16, 16, 24, 36, 54 ... -- up to the workers the job can use (the CUs less an eighth, or the
// builders the stage announced, expect()).
class BuildQueue {
 public:
  // (ages: the builders with sample ages have workers of their own -- another kernel)
  static BuildQueue *of(int device, int N, bool ages = false) {
    // (the queues live as long as the process and are never destroyed: their launcher threads wait on them)
    static std::mutex gm;
    static std::vector<BuildQueue *> *all = new std::vector<BuildQueue *>();
    std::lock_guard<std::mutex> lk(gm);
    for (BuildQueue *q : *all)
      if (q->device_ == device && q->N_ == N && q->ages_ == ages) return q;
    BuildQueue *q = new BuildQueue(device, N, ages);
    if (!q->ok_) return nullptr;
    all->push_back(q);
    return q;
  }
  // the most workers the caller wants alive (the stage: its section threads, fewer when RePaint needs the chip); 0: no
  // word
  void expect(int builders) {
    std::lock_guard<std::mutex> lk(m_);
    expected_ = builders;
  }
  // ... of SEVERAL callers that share the device (target ranges as threads of one process, shard.cpp): each adds what
  // it asks for and takes it off again
  void expect_add(int delta) {
    std::lock_guard<std::mutex> lk(m_);
    expected_ = std::max(0, expected_ + delta);
  }
  // Publishes the tree and returns when ITS worker says it is out (p.host_done, -1 until then): 0, or RL_EHIP.
  int run(const MMParams &p) {
    {
      std::lock_guard<std::mutex> lk(m_);
      const unsigned t = published_++;
      MMParams pp = p;
      pp.trace = timing_level() >= 2 ? &q_->pad[8] : nullptr;
      memcpy(q_->words[t % MM_QUEUE_CAP], &pp, sizeof(MMParams));
      __atomic_store_n(&q_->tail, published_, __ATOMIC_RELEASE);
      outstanding_++;
    }
    cv_.notify_one();
    int rc = 0;
    const auto t0 = std::chrono::steady_clock::now();
    static const bool trace = (timing_level() >= 2);
    auto last_trace = t0;
    for (unsigned spins = 0;; spins++) {
      if (__atomic_load_n(p.host_done, __ATOMIC_ACQUIRE) != -1) break;
      if (trace && std::chrono::steady_clock::now() - last_trace > std::chrono::seconds(1)) {
        last_trace = std::chrono::steady_clock::now();
        fprintf(stderr, "[mm trace] N=%d waiting %.0f s: published %u, workers started %u, tickets claimed %u, trees left %u, "
                "workers gone %u, outstanding %d; last mark: %u clusters left, phase %u\n", N_,
                std::chrono::duration<double>(last_trace - t0).count(),
                __atomic_load_n(&q_->tail, __ATOMIC_ACQUIRE), __atomic_load_n(&q_->pad[0], __ATOMIC_ACQUIRE),
                __atomic_load_n(&q_->pad[1], __ATOMIC_ACQUIRE), __atomic_load_n(&q_->pad[2], __ATOMIC_ACQUIRE),
                __atomic_load_n(&q_->pad[3], __ATOMIC_ACQUIRE), outstanding_, __atomic_load_n(&q_->pad[8], __ATOMIC_ACQUIRE),
                __atomic_load_n(&q_->pad[9], __ATOMIC_ACQUIRE));
        fflush(stderr);
      }
      if (failed_.load()) {
        rc = RL_EHIP;
        break;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(spins < 50 ? 100 : 250));
      if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(600)) {
        // (the request may still be claimed or finished by a worker: the queue is done for -- every builder's wait
        //  ends with an error -- and the caller keeps this tree's buffers out of circulation, ~DeviceMinMatch)
        set_error("tree builder: the tree was not built within 10 minutes");
        failed_.store(true);
        rc = RL_EHIP;
        break;
      }
    }
    std::lock_guard<std::mutex> lk(m_);
    outstanding_--;
    return rc;
  }

 private:
  BuildQueue(int device, int N, bool ages) : device_(device), N_(N), ages_(ages), kind_(worker_kind(N, ages)) {
    if (hipSetDevice(device) != hipSuccess) return;
    if (hipHostMalloc(reinterpret_cast<void **>(&q_), sizeof(WorkQueue), hipHostMallocCoherent) != hipSuccess) return;
    memset(q_, 0, sizeof(WorkQueue));
    if (d_state_.alloc(sizeof(WorkerState)) || hipMemset(d_state_.p, 0, sizeof(WorkerState)) != hipSuccess) return;
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    // `occ` workers share a CU; an eighth of the chip stays free for the stage's short kernels (RePaint, distance
    // matrices, weave) unless the stage says otherwise (expect())
    cap_ = env_int("RELATE_AMD_BUILD_WORKERS", kind_.occ * (cus - cus / 8), 1, 1024);
    cap_from_env_ = getenv("RELATE_AMD_BUILD_WORKERS") != nullptr;
    ok_ = true;
    std::thread([this] { launcher(); }).detach();
  }
  void launcher() {
    (void)hipSetDevice(device_);
    // The workers run for as long as there are trees; the short kernels of the stage must not queue up behind a
    // launch in a hardware queue they happen to share: lowest-priority streams, which the runtime maps to hardware
    // queues of their own.
    hipStream_t streams[MM_LAUNCHES];
    int size[MM_LAUNCHES] = {};
    for (auto &st : streams)
      if (make_stream(&st, true) != hipSuccess) {
        failed_.store(true);
        return;
      }
    const size_t dyn = kind_.dyn;
    // (the per-cluster state of a build in LDS: more than the default 64 KB of dynamic LDS)
    if (dyn > 0 && hipFuncSetAttribute(kind_.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
      failed_.store(true);
      return;
    }
    const bool verbose = getenv("RELATE_AMD_TIMING") != nullptr;
    for (;;) {
      int demand = 0, goal = 0;
      {
        std::unique_lock<std::mutex> lk(m_);
        static const bool trace = (timing_level() >= 2);
        while (trace && !cv_.wait_for(lk, std::chrono::milliseconds(500), [&] { return outstanding_ > 0; })) {
          fprintf(stderr, "[mm trace] N=%d idle: published %u, workers started %u, tickets claimed %u, trees left %u, workers "
                  "gone %u; streams busy:", N_, __atomic_load_n(&q_->tail, __ATOMIC_ACQUIRE),
                  __atomic_load_n(&q_->pad[0], __ATOMIC_ACQUIRE), __atomic_load_n(&q_->pad[1], __ATOMIC_ACQUIRE),
                  __atomic_load_n(&q_->pad[2], __ATOMIC_ACQUIRE), __atomic_load_n(&q_->pad[3], __ATOMIC_ACQUIRE));
          for (int l = 0; l < MM_LAUNCHES; l++) fprintf(stderr, " %d", hipStreamQuery(streams[l]) == hipSuccess ? 0 : size[l]);
          fprintf(stderr, "\n");
          fflush(stderr);
        }
        // While a launch of workers is alive the loop keeps polling (its workers leave 50 ms after the last tree):
        // the library's memory cache may trim -- hipFree waits for the device -- only when none is (g_worker_launches).
        int launches_alive = 0;
        for (int l = 0; l < MM_LAUNCHES; l++) launches_alive += size[l] > 0;
        if (launches_alive > 0) {
          if (!cv_.wait_for(lk, std::chrono::milliseconds(20), [&] { return outstanding_ > 0; })) {
            lk.unlock();
            for (int l = 0; l < MM_LAUNCHES; l++)
              if (size[l] > 0 && hipStreamQuery(streams[l]) == hipSuccess) {
                size[l] = 0;
                g_worker_launches.fetch_sub(1);
              }
            continue;
          }
        } else {
          cv_.wait(lk, [&] { return outstanding_ > 0; });
        }
        demand = outstanding_;
        // (the stage's word, expect(), is a limit -- it knows what RePaint needs of the chip; RELATE_AMD_BUILD_WORKERS
        //  overrides it)
        goal = (expected_ > 0 && !cap_from_env_) ? std::min(cap_, expected_) : cap_;
        // whole rounds of the XCDs (below): the goal too -- where the workers are many enough for their spread over the
        // XCDs to matter (a stage of 43 sections lost 3 of its workers to the rounding: config #5 on one GPU)
        if (goal >= MM_WHOLE_ROUNDS) goal -= goal % MM_XCDS;
      }
      int alive = 0, free_stream = -1;
      for (int l = 0; l < MM_LAUNCHES; l++) {
        if (size[l] > 0 && hipStreamQuery(streams[l]) == hipSuccess) {  // its last worker has left
          size[l] = 0;
          g_worker_launches.fetch_sub(1);
        }
        // (workers decide to leave one by one -- idle past the limit and nobody of the launch building --, so a late
        //  claim can keep one of them at work while its peers are gone: a launch counts for the workers it still has)
        alive += size[l] - std::min(size[l], (int)__atomic_load_n(&q_->gone[l], __ATOMIC_ACQUIRE));
        if (size[l] == 0 && free_stream < 0) free_stream = l;
      }
      // Workers follow the trees that are waiting or being built, not the sections that exist: a section spends half
      // of its time outside the build (distance matrix, RePaint, mapping), and a worker without a tree still holds
      // its CU -- with one per section RePaint had 122 CUs of 256 and was the stage's bottleneck (82 % busy, 0.7 s of
      // queue per launch).  A launch adds what is missing plus a margin, at least half of what is alive (the
      // launches are few: MM_LAUNCHES streams), never past the goal.
      if (demand + 4 > alive && alive < goal && free_stream >= 0) {
        int n = std::max({demand + 8 - alive, alive / 2, 16});
        // Workgroup b of a launch goes to XCD b % 8, every launch starting at XCD 0 again: launches of any size pile
        // their remainders on the low XCDs (six launches: up to six workers more there than on XCD 7), and a RePaint
        // launch -- its workgroups dealt to the XCDs in the same round-robin -- lasts as long as the XCD with the
        // fewest CUs left.  Whole rounds only.
        if (goal >= MM_WHOLE_ROUNDS) {
          n = (n + MM_XCDS - 1) / MM_XCDS * MM_XCDS;
          // (`alive` counts live workers since round 4, so what is missing need not be a whole round: wait until it is)
          n = std::min(n, (goal - alive) / MM_XCDS * MM_XCDS);
          if (n <= 0) {
            std::this_thread::sleep_for(std::chrono::microseconds(300));
            continue;
          }
        } else {
          n = std::max(1, std::min(n, goal - alive));
        }
        const int l = free_stream;
        __atomic_store_n(&q_->gone[l], 0u, __ATOMIC_RELEASE);  // (the stream's previous launch is through)
        const long long idle = (long long)idle_ms_ * 100000LL;
        const int trace_flag = timing_level() >= 2 ? 1 : 0;
        {
          WorkQueue *a_q = q_;
          WorkerState *a_ws = d_state_.as<WorkerState>();
          int a_l = l, a_trace = trace_flag;
          long long a_idle = idle;
          void *args[] = {&a_q, &a_ws, &a_l, &a_idle, &a_trace};
          (void)hipLaunchKernel(kind_.fn, dim3((unsigned)n), dim3(MM_BLOCK), args, dyn, streams[l]);
        }
        if (hipGetLastError() != hipSuccess) {
          // (this thread ends: the launches it still counted alive are nobody's to take off the count any more --
          //  the memory cache would wait for them on every failed allocation, and never trim)
          for (int m = 0; m < MM_LAUNCHES; m++)
            if (size[m] > 0) {
              size[m] = 0;
              g_worker_launches.fetch_sub(1);
            }
          failed_.store(true);
          return;
        }
        size[l] = n;
        g_worker_launches.fetch_add(1);
        if (verbose) {
          fprintf(stderr, "[tree builder workers] +%d on stream %d: %d alive, %d trees waiting or being built, goal %d\n", n,
                  l, alive + n, demand, goal);
          fflush(stderr);
        }
      }
      std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
  }
  int device_, N_;
  bool ages_, ok_ = false;
  WorkerKind kind_;
  WorkQueue *q_ = nullptr;
  DevBuf d_state_;
  int cap_ = 224, idle_ms_ = 50;
  bool cap_from_env_ = false;
  std::mutex m_;
  std::condition_variable cv_;
  unsigned published_ = 0;
  int outstanding_ = 0, expected_ = 0;
  std::atomic<bool> failed_{false};
};
int device_builder_expect(int device, int N, int builders, bool ages) {
  if (hipSetDevice(device) != hipSuccess) return -1;
  BuildQueue *q = BuildQueue::of(device, N, ages);
  if (!q) return -1;
  q->expect(builders);
  return 0;
}
int device_builder_expect_add(int device, int N, int delta, bool ages) {
  if (hipSetDevice(device) != hipSuccess) return -1;
  BuildQueue *q = BuildQueue::of(device, N, ages);
  if (!q) return -1;
  q->expect_add(delta);
  return 0;
}

// the word of coherent pinned memory a worker tells its builder "this tree is out" in: taken from / given back to a
// list (give != nullptr), never freed
static int *done_word(int *give) {
  static std::mutex m;
  static std::vector<int *> spare;
  std::lock_guard<std::mutex> lk(m);
  if (give) {
    spare.push_back(give);
    return nullptr;
  }
  if (!spare.empty()) {
    int *w = spare.back();
    spare.pop_back();
    return w;
  }
  int *w = nullptr;
  return hipHostMalloc(reinterpret_cast<void **>(&w), 64, hipHostMallocCoherent) == hipSuccess ? w : nullptr;
}

struct DeviceMinMatch::Impl {
  int N = 0, device = 0;
  hipStream_t stream = nullptr;
  DeviceShare::Staging *staging = nullptr;  // held from device_matrix() / the upload until the matrices are woven
  DevBuf d_M, d_hits, d_f, d_i, d_feas, d_rowlist, d_status, d_flags, d_member, d_tab, d_acc;
  DevBuf d_ages;  // --sample_ages: the third key and the draw of the candidates, the clusters' age levels, the table
  bool ages_up = false;  // ... whose constant part is in place
  long long builds = 0, n_built = 0;
  double t_prep = 0, t_wait = 0, t_out = 0;  // RELATE_AMD_TIMING: uploads + weave, submit -> tree done, copy-out (s)
  long long n_timed = 0;
  int *h_done = nullptr;  // pinned: the worker's "this tree is out"
  // pinned: what goes in and comes out per tree (candidate indices 2N, min_values_CF N, the merges 2N) -- from
  // pageable memory every one of these small copies is staged and synchronised by the runtime, milliseconds each
  // with a hundred sections copying at once
  int *h_io = nullptr;
  size_t h_io_bytes = 0, h_tab_bytes = 0;
  int *h_tab = nullptr;   // pinned: the previous tree's tables for prior_kernel (6T + N ints), then N carrier flags
  float acc_val = 0.0f;   // d_acc holds acc[c] = val added c times for this val
  bool acc_valid = false;
  bool abandoned = false;  // a tree was given up while its ticket may be live: nothing of this builder is recycled
  // the row minima of the staged matrices are in d_f already (apply_penalty / apply_prior took them in their pass)
  bool rowmin_d_ready = false, rowmin_cf_ready = false;
  void drop_staging() {
    if (staging) DeviceShare::of(device).give(staging);
    staging = nullptr;
    rowmin_d_ready = rowmin_cf_ready = false;
  }
};

DeviceMinMatch::DeviceMinMatch(int N, int device) : impl(new Impl()) {
  impl->N = N;
  impl->device = device;
}
DeviceMinMatch::~DeviceMinMatch() {
  impl->drop_staging();
  if (getenv("RELATE_AMD_TIMING") && impl->n_timed)
    fprintf(stderr, "[gpu tree builder] %lld trees, host ms per tree: uploads + weave %.2f, submit -> done %.2f, "
                    "copy-out %.2f\n", impl->n_timed, 1e3 * impl->t_prep / impl->n_timed,
            1e3 * impl->t_wait / impl->n_timed, 1e3 * impl->t_out / impl->n_timed);
  if (impl->abandoned) {
    // (leaked on purpose: handed to the caches these blocks would be given out again at once, and a worker that
    //  claims or finishes the abandoned ticket later would write into somebody else's tree)
    for (DevBuf *b : {&impl->d_M, &impl->d_hits, &impl->d_f, &impl->d_i, &impl->d_feas, &impl->d_rowlist, &impl->d_status,
                      &impl->d_flags, &impl->d_member, &impl->d_tab, &impl->d_acc, &impl->d_ages}) {
      b->p = nullptr;
      b->bytes = 0;
    }
    impl->staging = nullptr;
    delete impl;
    return;
  }
  if (impl->stream) (void)hipStreamDestroy(impl->stream);
  // (nothing goes back to the driver here: hipFree / hipHostFree would wait for the workers of the other builders)
  if (impl->h_done) done_word(impl->h_done);
  if (impl->h_io) pinned_cache_release(impl->h_io, impl->h_io_bytes);
  if (impl->h_tab) pinned_cache_release(impl->h_tab, impl->h_tab_bytes);
  delete impl;
}

void DeviceMinMatch::forget_ages() { impl->ages_up = false; }

float *DeviceMinMatch::device_matrix() {
  Impl &m = *impl;
  if (hipSetDevice(m.device) != hipSuccess) return nullptr;
  if (!m.staging) m.staging = DeviceShare::of(m.device).take(m.N);
  return m.staging ? m.staging->D.as<float>() : nullptr;
}

int DeviceMinMatch::stage_from_device(const float *dD, const float *dCF) {
  Impl &m = *impl;
  RL_HIP(hipSetDevice(m.device));
  if (!m.stream) RL_HIP(make_stream(&m.stream, false, true));
  if (!m.staging) m.staging = DeviceShare::of(m.device).take(m.N);
  if (!m.staging) return -1;
  const size_t bytes = (size_t)m.N * m.N * 4;
  RL_HIP(hipMemcpyAsync(m.staging->D.p, dD, bytes, hipMemcpyDeviceToDevice, m.stream));
  if (dCF) RL_HIP(hipMemcpyAsync(m.staging->CF.p, dCF, bytes, hipMemcpyDeviceToDevice, m.stream));
  m.rowmin_d_ready = m.rowmin_cf_ready = false;
  return 0;
}

float *DeviceMinMatch::rowmin_device() {
  Impl &m = *impl;
  if (hipSetDevice(m.device) != hipSuccess || m.d_f.alloc((size_t)8 * m.N * 4)) return nullptr;
  return m.d_f.as<float>() + 6 * (size_t)m.N;
}
void DeviceMinMatch::rowmin_is_ready() { impl->rowmin_d_ready = true; }

int DeviceMinMatch::apply_penalty(const char *member, float val) {
  Impl &m = *impl;
  const int N = m.N;
  if (!m.staging) return -1;
  RL_HIP(hipSetDevice(m.device));
  if (!m.stream) RL_HIP(make_stream(&m.stream, false, true));
  if (m.d_member.alloc((size_t)N)) return -1;
  const size_t tab_ints = (size_t)6 * (2 * N - 1) + N;
  if (!m.h_tab && !(m.h_tab = static_cast<int *>(pinned_cache_alloc(tab_ints * 4 + (size_t)N, &m.h_tab_bytes)))) {
    set_error("tree builder: no pinned host memory for the prior's tables");
    return -1;
  }
  char *flags = reinterpret_cast<char *>(m.h_tab + tab_ints);
  memcpy(flags, member, (size_t)N);
  if (m.d_f.alloc((size_t)8 * N * 4)) return -1;  // (reserve()'s: the row minima live at 6N and 7N)
  RL_HIP(hipMemcpyAsync(m.d_member.p, flags, (size_t)N, hipMemcpyHostToDevice, m.stream));
  // the penalty and, of the rows as they then are, the row minima of the build: one pass over the matrix
  hipLaunchKernelGGL(rowmin_penalty_kernel, dim3(N), dim3(256), 0, m.stream, m.staging->D.as<float>(), N,
                     m.d_member.as<unsigned char>(), val, m.d_f.as<float>() + 6 * (size_t)N);
  RL_HIP(hipGetLastError());
  m.rowmin_d_ready = true;
  return 0;
}

int DeviceMinMatch::apply_prior(const HostTree &t, float val) {
  Impl &m = *impl;
  const int N = m.N, T = 2 * N - 1;
  if (!m.staging) return -1;
  RL_HIP(hipSetDevice(m.device));
  if (!m.stream) RL_HIP(make_stream(&m.stream, false, true));
  // the same tables as the host's clade_prior (treeseq.cpp), in pinned memory: nothing here waits for the copy
  // (the next tree of this builder -- the next writer of the block -- comes after this one is built)
  const size_t tab_ints = (size_t)6 * T + N;
  if (!m.h_tab && !(m.h_tab = static_cast<int *>(pinned_cache_alloc(tab_ints * 4 + (size_t)N, &m.h_tab_bytes)))) {
    set_error("tree builder: no pinned host memory for the prior's tables");
    return -1;
  }
  int *parent = m.h_tab, *cl = parent + T, *cr = cl + T, *depth = cr + T, *lo = depth + T, *size = lo + T,
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
  int rc = m.d_tab.alloc(tab_ints * 4);
  rc = rc ? rc : m.d_acc.alloc(((size_t)N + 1) * 4);
  rc = rc ? rc : m.d_f.alloc((size_t)8 * N * 4);  // (reserve()'s: the row minima live at 6N and 7N)
  if (rc) return -1;
  if (!m.acc_valid || m.acc_val != val) {  // (one val per section: once)
    std::vector<float> acc((size_t)N + 1, 0.0f);
    for (int c = 1; c <= N; c++) acc[c] = acc[c - 1] + val;
    RL_HIP(hipMemcpy(m.d_acc.p, acc.data(), acc.size() * 4, hipMemcpyHostToDevice));
    m.acc_val = val;
    m.acc_valid = true;
  }
  RL_HIP(hipMemcpyAsync(m.d_tab.p, m.h_tab, tab_ints * 4, hipMemcpyHostToDevice, m.stream));
  const int *q = m.d_tab.as<int>();
  hipLaunchKernelGGL(prior_kernel, dim3(N), dim3(256), (size_t)N * sizeof(float), m.stream, m.staging->CF.as<float>(), N, q, q + T,
                     q + 2 * (size_t)T, q + 3 * (size_t)T, q + 4 * (size_t)T, q + 5 * (size_t)T, q + 6 * (size_t)T,
                     m.d_acc.as<float>(), m.d_f.as<float>() + 7 * (size_t)N);
  RL_HIP(hipGetLastError());
  m.rowmin_cf_ready = true;
  return 0;
}

int DeviceMinMatch::reserve(bool ages) {
  Impl &m = *impl;
  const int N = m.N;
  if (hipSetDevice(m.device) != hipSuccess) return -1;
  const long long pair_cap = (long long)(ages ? 32 : 8) * N;  // (build_impl)
  int rc = m.d_M.alloc(mm_elements(N) * 16);
  rc = rc ? rc : m.d_hits.alloc(((size_t)N + (size_t)2 * N * MM_HITS) * 4);
  rc = rc ? rc : m.d_f.alloc((size_t)8 * N * 4);  // min_values, min_values_CF, mc_dist, mc_dist2, 2 of the symmetric path, 2 row minima
  rc = rc ? rc : m.d_i.alloc(((size_t)14 * N + 8) * 4);  // ints, see build_impl
  rc = rc ? rc : m.d_feas.alloc((size_t)pair_cap * 6 * 4);
  rc = rc ? rc : m.d_rowlist.alloc((size_t)MM_WAVES * N * 4);
  rc = rc ? rc : m.d_status.alloc(16 + 16 * 8);
  rc = rc ? rc : m.d_flags.alloc((size_t)N);
  return rc ? -1 : 0;
}

int DeviceMinMatch::build_resident(MinMatch &tb, bool with_prior, HostTree &tree) {
  const int rc = build_impl(tb, nullptr, nullptr, nullptr, true, with_prior, tree);
  impl->drop_staging();
  return rc;
}

int DeviceMinMatch::build(MinMatch &tb, const float *d, const float *prior, HostTree &tree) {
  const int rc = build_impl(tb, nullptr, d, prior, false, prior != nullptr, tree);
  impl->drop_staging();
  return rc;
}

int DeviceMinMatch::build_resident(MinMatchAges &tb, const std::vector<double> &sample_ages, bool with_prior, HostTree &tree) {
  const int rc = build_impl(tb, &sample_ages, nullptr, nullptr, true, with_prior, tree);
  impl->drop_staging();
  return rc;
}

int DeviceMinMatch::build(MinMatchAges &tb, const std::vector<double> &sample_ages, const float *d, const float *prior,
                          HostTree &tree) {
  const int rc = build_impl(tb, &sample_ages, d, prior, false, prior != nullptr, tree);
  impl->drop_staging();
  return rc;
}

template <class TB>
int DeviceMinMatch::build_impl(TB &tb, const std::vector<double> *sample_ages, const float *d, const float *prior_host,
                               bool resident, bool with_prior, HostTree &tree) {
  Impl &m = *impl;
  const int N = m.N;
  constexpr bool ages = std::is_same<TB, MinMatchAges>::value;
  if (ages && (!sample_ages || (int)sample_ages->size() != N)) {
    set_error("tree builder: one sample age per haplotype");
    return -1;
  }
  if (N < 2 || N > MM_MAXN) return 1;  // (the painting kernels stop at N = 10240 too)
  if (resident) {  // tests: every k-th resident build is handed to the host, as a tree with too many tied candidates is
    static const int every = getenv("RELATE_AMD_TEST_HANDOVER_EVERY") ? atoi(getenv("RELATE_AMD_TEST_HANDOVER_EVERY")) : 0;
    if (every > 0 && ++m.builds % every == 0) return 2;
  }
  RL_HIP(hipSetDevice(m.device));
  if (!m.stream) RL_HIP(make_stream(&m.stream, false, true));
  BuildQueue *queue = BuildQueue::of(m.device, N, ages);
  if (!queue) {
    set_error("tree builder: no queue on device %d (pinned host memory)", m.device);
    return -1;
  }
  if (!m.staging) {
    if (resident) {
      set_error("tree builder: build_resident without device_matrix()");
      return -1;
    }
    m.staging = DeviceShare::of(m.device).take(N);
    if (!m.staging) return -1;
  }
  const bool prior = with_prior;
  const size_t NN = (size_t)N * N;
  const auto tb0 = std::chrono::steady_clock::now();
  // (with sample ages a merge of the lineage the last tree's candidates were renamed to sends most clusters through
  //  the rebuilding branch: every feasible pair of the matrix again)
  const long long pair_cap = (long long)(ages ? 32 : 8) * N;
  if (reserve(ages)) return -1;
  MMParams p;
  memset(&p, 0, sizeof(p));
  p.N = N;
  const WorkerKind kind = worker_kind(N, ages);
  p.layout = kind.layout;
  p.threshold = tb.threshold;
  p.threshold_CF = tb.threshold_CF;
  p.M = m.d_M.as<float4>();
  p.has_prior = prior ? 1 : 0;
  float *f = m.d_f.as<float>();
  p.rowmin_D = f + 6 * (size_t)N;
  p.rowmin_CF = f + 7 * (size_t)N;
  int *hits = m.d_hits.as<int>();
  p.hit_cnt = hits;
  p.hit_b = reinterpret_cast<unsigned *>(hits + N);
  p.hit_sym = reinterpret_cast<float *>(hits + N + (size_t)N * MM_HITS);
  p.min_values = f;
  p.min_values_CF = f + N;
  p.mc_dist = f + 2 * (size_t)N;
  p.mc_dist2 = f + 3 * (size_t)N;
  p.min_values_sym = f + 4 * (size_t)N;
  p.mcs_dist = f + 5 * (size_t)N;
  DeviceShare::of(m.device).sym_pool(N, &p.SYM, &p.sym_locks, &p.sym_slots);
  int *q = m.d_i.as<int>();
  p.mc_lin1 = q;
  p.mc_lin2 = q + N;
  p.cluster_index = q + 2 * (size_t)N;
  p.cluster_size = q + 3 * (size_t)N;
  p.upd_pos = q + 5 * (size_t)N;
  p.mcs_lin1 = q + 6 * (size_t)N;
  p.mcs_lin2 = q + 7 * (size_t)N;
  p.merge_i = q + 8 * (size_t)N;           // [N-1]
  p.merge_j = q + 9 * (size_t)N;           // [N-1]
  p.upd_g = reinterpret_cast<unsigned *>(q + 12 * (size_t)N);  // (the AGES build: a list of its own, below)
  p.updv_g = reinterpret_cast<float *>(q + 13 * (size_t)N);
  p.kflag = m.d_flags.as<unsigned char>();
  p.pair_g = m.d_feas.as<unsigned>();
  p.pair_cap = pair_cap;
  p.rowlist = m.d_rowlist.as<int>();
  p.status = m.d_status.as<int>();
  if (!m.h_done && !(m.h_done = done_word(nullptr))) {
    set_error("tree builder: no pinned host memory for the completion word");
    return -1;
  }
  p.host_done = m.h_done;
  if constexpr (ages) {
    // the age levels: a candidate's third key is the index of an age in the sorted table of the distinct sample ages
    // (tree_builder.cpp:1125-1152)
    tb.prepare_levels(*sample_ages);
    const std::vector<double> *unique = &tb.unique_ages;
    const std::vector<int> *count = &tb.ages_count;
    p.Ne = tb.Ne;
    const int nlev = (int)unique->size();
    if (m.d_ages.alloc((size_t)48 * N + 64)) return -1;
    double *ad = m.d_ages.as<double>();
    int *ai = reinterpret_cast<int *>(ad + 2 * (size_t)N);
    p.mc_dist2d = ad;
    p.unique_ages = ad + N;
    p.mc_lvl = ai;
    p.age_lvl = ai + N;
    p.age_lvl0 = ai + 2 * (size_t)N;
    p.ages_count = ai + 3 * (size_t)N;
    p.upd_g = reinterpret_cast<unsigned *>(ai + 4 * (size_t)N);
    p.updv_g = reinterpret_cast<float *>(ai + 5 * (size_t)N);
    p.bucket = ai + 6 * (size_t)N;
    p.bucket_off = ai + 7 * (size_t)N + 8;
    p.n_levels = nlev;
    if (!m.ages_up) {  // (one set of ages per builder)
      std::vector<int> lvl0((size_t)N);
      for (int c = 0; c < N; c++)
        lvl0[c] = (int)(std::lower_bound(unique->begin(), unique->end(), (*sample_ages)[c]) - unique->begin());
      RL_HIP(hipMemcpy(ad + N, unique->data(), (size_t)nlev * 8, hipMemcpyHostToDevice));
      RL_HIP(hipMemcpy(ai + 2 * (size_t)N, lvl0.data(), (size_t)N * 4, hipMemcpyHostToDevice));
      RL_HIP(hipMemcpy(ai + 3 * (size_t)N, count->data(), (size_t)nlev * 4, hipMemcpyHostToDevice));
      m.ages_up = true;
    }
  }
  const bool timing = getenv("RELATE_AMD_TIMING") != nullptr;
  p.timers = nullptr;  // (set below: a corner of the pinned block, read when the tree is out -- no copy)

  // state the builders carry from tree to tree: in
  if (!m.h_io && !(m.h_io = static_cast<int *>(pinned_cache_alloc((size_t)5 * N * 4 + 16 * 8 + 8, &m.h_io_bytes)))) {
    set_error("tree builder: no pinned host memory for the per-tree copies");
    return -1;
  }
  int *lin = m.h_io, *tr = m.h_io + 3 * (size_t)N;
  long long *h_timers = reinterpret_cast<long long *>((reinterpret_cast<uintptr_t>(m.h_io + 5 * (size_t)N) + 7) & ~(uintptr_t)7);
  if (timing) p.timers = h_timers;
  float *mvcf = reinterpret_cast<float *>(m.h_io + 2 * (size_t)N);
  for (int c = 0; c < N; c++) {
    lin[c] = tb.mc[c].lin1;
    lin[(size_t)N + c] = tb.mc[c].lin2;
    mvcf[c] = tb.min_values_CF[c];
  }
  // (min_values_CF as carried over is also what weave_kernel tests the prior with: that copy goes to the device)
  const float *mvcf_dev = p.min_values_CF;
  RL_HIP(hipMemcpyAsync(p.min_values_CF, mvcf, (size_t)N * 4, hipMemcpyHostToDevice, m.stream));
  if (p.layout != L_GLOBAL) {
    // The worker reads the carried state from the pinned block and leaves it there again together with the merges:
    // 100 KB across PCIe per tree instead of three copies and a stream synchronisation behind the build (16 ms per
    // tree with a hundred sections sharing the hardware queues).  The system-scope fences around a tree in the worker
    // order them with the request and with "done".
    p.io_lin = lin;
    p.io_mvcf = mvcf;
    p.merge_i = tr;
    p.merge_j = tr + N;
  } else {  // (the state lives in these arrays throughout the build: device memory)
    RL_HIP(hipMemcpyAsync(p.mc_lin1, lin, (size_t)2 * N * 4, hipMemcpyHostToDevice, m.stream));
  }
  float *dD = m.staging->D.as<float>(), *dCF = prior ? m.staging->CF.as<float>() : nullptr;
  if (!resident) {
    RL_HIP(hipMemcpyAsync(dD, d, NN * 4, hipMemcpyHostToDevice, m.stream));
    if (prior) RL_HIP(hipMemcpyAsync(dCF, prior_host, NN * 4, hipMemcpyHostToDevice, m.stream));
  }
  {
    // the row minima (where the penalty / prior passes have not left them already), then the woven matrix and the
    // pair scan of the initialisation in one pass, on the whole chip; the build itself is one workgroup
    if (!m.rowmin_d_ready)
      hipLaunchKernelGGL(rowmin_penalty_kernel, dim3(N), dim3(256), 0, m.stream, dD, N, (const unsigned char *)nullptr, 0.0f,
                         f + 6 * (size_t)N);
    if (prior && !m.rowmin_cf_ready)
      hipLaunchKernelGGL(rowmin_penalty_kernel, dim3(N), dim3(256), 0, m.stream, dCF, N, (const unsigned char *)nullptr, 0.0f,
                         f + 7 * (size_t)N);
    const dim3 grid((N + WV_ROWS - 1) / WV_ROWS);
    if (prior)
      hipLaunchKernelGGL(weave_kernel<true>, grid, dim3(256), 0, m.stream, dD, dCF, p.M, N, p.rowmin_D, p.rowmin_CF, mvcf_dev,
                         ages ? 1 : 0, p.threshold, p.threshold_CF, hits, reinterpret_cast<unsigned *>(hits + N),
                         reinterpret_cast<float *>(hits + N + (size_t)N * MM_HITS));
    else
      hipLaunchKernelGGL(weave_kernel<false>, grid, dim3(256), 0, m.stream, dD, dCF, p.M, N, p.rowmin_D, p.rowmin_CF, mvcf_dev,
                         ages ? 1 : 0, p.threshold, p.threshold_CF, hits, reinterpret_cast<unsigned *>(hits + N),
                         reinterpret_cast<float *>(hits + N + (size_t)N * MM_HITS));
    RL_HIP(hipGetLastError());
  }
  static const bool trace = (timing_level() >= 2);
  if (trace) fprintf(stderr, "[mm trace] N=%d inputs submitted\n", N), fflush(stderr);
  RL_HIP(hipStreamSynchronize(m.stream));  // inputs in place
  if (trace) fprintf(stderr, "[mm trace] N=%d inputs in place\n", N), fflush(stderr);
  if (const char *dump = getenv("RELATE_AMD_TEST_MM_DUMP")) {
    // (tools/bench_builder_variants.py) the matrices of builds first..last of this builder, as the kernel gets them:
    // "<dir>:<first>:<last>"; the process ends behind the last one
    char dir[512];
    int first = 0, last = 0;
    if (sscanf(dump, "%511[^:]:%d:%d", dir, &first, &last) == 3 && m.n_built >= first && m.n_built <= last) {
      std::vector<float> host(NN);
      for (int which = 0; which < (prior ? 2 : 1); which++) {
        RL_HIP(hipMemcpy(host.data(), which ? (const void *)dCF : (const void *)dD, NN * 4, hipMemcpyDeviceToHost));
        const std::string fn = std::string(dir) + (which ? "/cf_" : "/d_") + std::to_string(m.n_built) + ".bin";
        FILE *fp = fopen(fn.c_str(), "wb");
        if (fp) {
          fwrite(host.data(), 4, NN, fp);
          fclose(fp);
        }
      }
      if (m.n_built == last) _exit(0);
    }
  }
  m.n_built++;
  m.drop_staging();                        // (woven: the row-major matrices go back to the pool)
  const auto tb1 = std::chrono::steady_clock::now();
  __atomic_store_n(m.h_done, -1, __ATOMIC_RELEASE);
  if (queue->run(p)) {
    set_error("tree builder: the workers of device %d failed", m.device);
    m.abandoned = true;  // a worker may still write this tree's state, merges and completion word
    return -1;
  }
  const auto tb2 = std::chrono::steady_clock::now();
  // (the worker stores the code where it stores "done", behind a system-scope fence: no copy of p.status, which
  //  could still sit in the worker's L2 for a copy engine)
  const int status = __atomic_load_n(m.h_done, __ATOMIC_ACQUIRE);
  if (trace) fprintf(stderr, "[mm trace] N=%d tree out, status %d\n", N, status), fflush(stderr);
  if (status != 0) return status > 0 ? status : -1;
  if (timing) {
    const long long *tk = h_timers;
    fprintf(stderr, "[gpu tree builder] N=%d, us:", N);
    static const char *names[16] = {"row minima", "pair scan", "updates", "rescans", "pair tests", "pair order",
                                    "ordered", "symmetric", "erase", "pairs_x100", "before_draws", "", "rebuilt_x100",
                                    "rescan_only_x100", "merges_none_rebuilt_x100", "merges_with_rescan_only_x100"};
    for (int x = 0; x < 11; x++) fprintf(stderr, " %s %.0f", names[x], tk[x] / 100.0);
    for (int x = 12; x < 16; x++) fprintf(stderr, " %s %.0f", names[x], tk[x] / 100.0);
    fprintf(stderr, " shader_MHz %lld\n", tk[11]);
  }
  // out: the tree and the carried state
  // (merge_i [N), merge_j [N) lie back to back)
  if (p.layout == L_GLOBAL) {
    RL_HIP(hipMemcpyAsync(tr, p.merge_i, ((size_t)2 * N - 1) * 4, hipMemcpyDeviceToHost, m.stream));
    RL_HIP(hipMemcpyAsync(lin, p.mc_lin1, (size_t)2 * N * 4, hipMemcpyDeviceToHost, m.stream));
    RL_HIP(hipMemcpyAsync(mvcf, p.min_values_CF, (size_t)N * 4, hipMemcpyDeviceToHost, m.stream));
    RL_HIP(hipStreamSynchronize(m.stream));
  }
  for (int c = 0; c < N; c++) tb.min_values_CF[c] = mvcf[c];
  tree.reset(N);
  {  // the nodes of the tree from the merges (tree_builder.cpp:2437-2460, 2631-2636): cluster j lives on as the new node
    std::vector<int> node(N);
    for (int c = 0; c < N; c++) node[c] = c;
    for (int mth = 0; mth < N - 1; mth++) {
      const int ci = tr[mth], cj = tr[(size_t)N + mth], nn = N + mth;
      if (ci < 0 || ci >= N || cj < 0 || cj >= N) {
        set_error("tree builder on the device: merge %d names clusters %d and %d", mth, ci, cj);
        return -1;
      }
      tree.parent[node[ci]] = nn;
      tree.parent[node[cj]] = nn;
      tree.child_left[nn] = node[ci];
      tree.child_right[nn] = node[cj];
      node[cj] = nn;
    }
  }
  for (int c = 0; c < N; c++) {
    tb.mc[c].lin1 = lin[c];
    tb.mc[c].lin2 = lin[(size_t)N + c];
  }
  {
    const auto tb3 = std::chrono::steady_clock::now();
    m.t_prep += std::chrono::duration<double>(tb1 - tb0).count();
    m.t_wait += std::chrono::duration<double>(tb2 - tb1).count();
    m.t_out += std::chrono::duration<double>(tb3 - tb2).count();
    m.n_timed++;
  }
  return 0;
}

}  // namespace rl

// ---- C ABI: a tree builder that keeps MinMatch's state from tree to tree
struct rl_builder {
  int N;
  double theta;
  rl::MinMatch tb;
  rl::DeviceMinMatch *dev;
  rl::MinMatchAges *ages_tb = nullptr;  // rl_builder_set_sample_ages: the builder with the third key and the clock
  std::vector<double> ages;
  int last_on_gpu;
  rl_builder(int n, double th, int device)
      : N(n), theta(th), tb(n, th), dev(device >= 0 ? new rl::DeviceMinMatch(n, device) : nullptr), last_on_gpu(0) {}
  ~rl_builder() {
    delete dev;
    delete ages_tb;
  }
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
    st = b->ages_tb ? b->dev->build(*b->ages_tb, b->ages, d, d_prior, t) : b->dev->build(b->tb, d, d_prior, t);
    if (st < 0) return RL_EHIP;
  }
  b->last_on_gpu = st == 0;
  if (st != 0) {
    if (b->ages_tb) b->ages_tb->quick_build(d, d_prior, b->ages, t);
    else b->tb.quick_build(d, d_prior, t);
  }
  const int N = b->N;
  for (int i = 0; i < 2 * N - 1; i++) parent[i] = t.parent[i];
  for (int i = N; i < 2 * N - 1; i++) {
    if (child_left) child_left[i - N] = t.child_left[i];
    if (child_right) child_right[i - N] = t.child_right[i];
  }
  return RL_OK;
}

int rl_builder_set_sample_ages(rl_builder *b, const double *ages, int n) {
  if (!b || n != b->N || !ages) {
    rl::set_error("rl_builder_set_sample_ages: one age per haplotype");
    return RL_EINVAL;
  }
  b->ages.assign(ages, ages + n);
  delete b->ages_tb;
  b->ages_tb = new rl::MinMatchAges(b->N, b->theta);
  if (b->dev) b->dev->forget_ages();
  return RL_OK;
}

int rl_builder_last_on_gpu(const rl_builder *b) { return b ? b->last_on_gpu : RL_EINVAL; }

int rl_debug_rng_mismatches(unsigned seed, int n) { return rl::rng_restatement_mismatches(seed, n); }

// (measurement hook, tools/bench_builder_many.py) `builders` device builders, a host thread each, build the SAME tree
// `reps` times side by side -- the matrices stay on the device: a copy per build into the builder's staging pair, as
// K3 and the prior kernel would leave them -- with `workers` resident workgroups at most (0: the queue's own limit).
// seconds: wall-clock from the first submit to the last tree; mismatches: builds whose parent array differs from
// builder 0's of the same repetition (0 expected: every builder carries the same state through the same trees).
int rl_debug_builder_throughput(int N, double theta, int device, int builders, int reps, int workers, const float *d,
                                const float *prior, double *seconds, int *mismatches, int *first_parents) {
  using namespace rl;
  if (N < 2 || builders < 1 || reps < 1 || !d || !seconds || !mismatches) return RL_EINVAL;
  RL_HIP(hipSetDevice(device));
  const size_t NN = (size_t)N * N;
  DevBuf masterD, masterCF;
  if (masterD.alloc(NN * 4) || (prior && masterCF.alloc(NN * 4))) return RL_ENOMEM;
  RL_HIP(hipMemcpy(masterD.p, d, NN * 4, hipMemcpyHostToDevice));
  if (prior) RL_HIP(hipMemcpy(masterCF.p, prior, NN * 4, hipMemcpyHostToDevice));
  std::vector<std::unique_ptr<MinMatch>> tbs;
  std::vector<std::unique_ptr<DeviceMinMatch>> devs;
  for (int b = 0; b < builders; b++) {
    tbs.emplace_back(new MinMatch(N, theta));
    devs.emplace_back(new DeviceMinMatch(N, device));
    if (devs.back()->reserve(false)) return RL_ENOMEM;
  }
  if (device_builder_expect(device, N, workers, false)) return RL_EHIP;
  std::vector<std::vector<int>> parents((size_t)builders * reps);
  std::atomic<int> ready(0), failed(0);
  std::atomic<bool> go(false);
  std::vector<std::thread> th;
  for (int b = 0; b < builders; b++)
    th.emplace_back([&, b] {
      (void)hipSetDevice(device);
      ready.fetch_add(1);
      while (!go.load()) std::this_thread::sleep_for(std::chrono::microseconds(50));
      for (int r = 0; r < reps && !failed.load(); r++) {
        HostTree t;
        if (devs[b]->stage_from_device(masterD.as<float>(), prior ? masterCF.as<float>() : nullptr) ||
            devs[b]->build_resident(*tbs[b], prior != nullptr, t) != 0) {
          failed.store(1);
          break;
        }
        parents[(size_t)b * reps + r].assign(t.parent.begin(), t.parent.begin() + (2 * N - 1));
      }
    });
  while (ready.load() < builders) std::this_thread::sleep_for(std::chrono::microseconds(100));
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true);
  for (auto &x : th) x.join();
  *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  (void)device_builder_expect(device, N, 0, false);
  if (failed.load()) return RL_EHIP;
  int bad = 0;
  for (int b = 1; b < builders; b++)
    for (int r = 0; r < reps; r++) bad += parents[(size_t)b * reps + r] != parents[r];
  *mismatches = bad;
  if (first_parents)
    for (int r = 0; r < reps; r++) memcpy(first_parents + (size_t)r * (2 * N - 1), parents[r].data(), ((size_t)2 * N - 1) * 4);
  return RL_OK;
}

}  // extern "C"
