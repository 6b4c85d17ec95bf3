// minmatch.h -- host tree builder (MinMatch agglomerative clustering).
//
// Restates MinMatch::QuickBuild of the reference (src/tree_builder.cpp) for
// empty sample_ages and no template tree -- the configuration
// AncesTreeBuilder::BuildTopology uses (src/anc_builder.cpp:447, :608).
// The object is deliberately STATEFUL across calls exactly where the
// reference's MinMatch is (candidate pair indices, min_values_CF): stale state
// steers which branches redraw random numbers, and the draw order is part of
// the result.
#pragma once
#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <utility>
#include <limits>
#include <random>
#include <vector>

#include "sym_pairs.h"

namespace rl {

// std::vector allocator for the N x N matrices of the tree builder: 2 MiB-aligned
// and advised to use transparent huge pages.  MinMatch::coalesce walks down
// matrix columns (one element per row): with 4 KiB pages every step is a TLB
// miss on top of the cache miss.
template <typename T>
struct HugePageAllocator {
  typedef T value_type;
  HugePageAllocator() = default;
  template <typename U>
  HugePageAllocator(const HugePageAllocator<U> &) {}
  T *allocate(size_t n);
  void deallocate(T *p, size_t) { free(p); }
  template <typename U>
  bool operator==(const HugePageAllocator<U> &) const { return true; }
  template <typename U>
  bool operator!=(const HugePageAllocator<U> &) const { return false; }
};
typedef std::vector<float, HugePageAllocator<float>> MatrixBuf;

struct HostTree {
  int N = 0;
  std::vector<int> parent, child_left, child_right;  // 2N-1 entries, -1 = none
  std::vector<float> num_events;
  std::vector<int> snp_begin, snp_end;
  int pos = 0;
  void reset(int n) {
    N = n;
    const int T = 2 * n - 1;
    parent.assign(T, -1);
    child_left.assign(T, -1);
    child_right.assign(T, -1);
    num_events.assign(T, 0.0f);
    snp_begin.assign(T, 0);
    snp_end.assign(T, 0);
  }
};

template <typename T>
T *HugePageAllocator<T>::allocate(size_t n) {
  const size_t bytes = n * sizeof(T), huge = (size_t)2 << 20;
  void *p = nullptr;
  if (bytes >= 2 * huge) {
    const size_t rounded = (bytes + huge - 1) & ~(huge - 1);
    p = aligned_alloc(huge, rounded);
    if (p) (void)madvise(p, rounded, MADV_HUGEPAGE);
  } else {
    p = malloc(bytes ? bytes : 1);
  }
  if (!p) throw std::bad_alloc();
  return static_cast<T *>(p);
}

// A few helper threads of one tree builder.  The merges of a tree are
// sequential, but inside a merge the distance updates of the two matrices are
// independent per cluster and are memory-latency-bound walks down two matrix
// columns: run(job) has every thread (the caller included) execute job(t, T).
// Helpers spin briefly between jobs (a merge lasts tens of microseconds) and
// fall asleep when none comes (distance matrix fetch, prior, mapping).
class BuildThreads {
 public:
  explicit BuildThreads(int T);
  ~BuildThreads();
  int size() const { return T_; }
  void run(const std::function<void(int, int)> &job);

 private:
  void worker(int t);
  int T_;
  std::vector<std::thread> th_;
  const std::function<void(int, int)> *job_ = nullptr;
  std::atomic<uint64_t> gen_{0};
  std::atomic<int> done_{0}, sleepers_{0};
  std::atomic<bool> stop_{false};
  std::mutex m_;
  std::condition_variable cv_;
};
// threads per tree builder: RELATE_AMD_BUILD_THREADS, else what the stage driver set (default 1)
int build_threads();
void set_build_threads(int T);

class MinMatch {
  friend class DeviceMinMatch;

 public:
  MinMatch(int N, double theta);
  // d: N*N floats, destroyed.  prior: N*N floats or nullptr.
  void quick_build(float *d, const float *prior, HostTree &tree);
  // wall-clock of the parts, accumulated over the builds (seconds): row minima + pair scan, the parallel and the
  // ordered half of the merges
  double t_init = 0, t_phase1 = 0, t_phase1a = 0, t_phase2 = 0;
  long long n_updated = 0, n_merges = 0;  // clusters that rebuilt their candidates / merges

 private:
  struct Cand {
    int lin1 = -1, lin2 = -1;
    double dist = std::numeric_limits<float>::infinity();
    double dist2 = std::numeric_limits<float>::infinity();
  };
  int N;
  float threshold, threshold_CF;
  std::mt19937 rng;
  std::vector<int> convert_index, cluster_index, updated_cluster;
  std::vector<float> cluster_size;
  std::vector<Cand> mc;
  Cand best;
  std::vector<float> min_values, min_values_CF;
  MatrixBuf d_CF;
  SymPairs sym;  // merges without a mutually closest pair (sym_pairs.h)
  float sym_dist = 0.f, dist_random = 0.f;

  float *D = nullptr;         // current asymmetric matrix
  const float *CF = nullptr;  // current prior matrix (d_CF) or nullptr
  std::uniform_real_distribution<double> unif{0.0, 1.0};

  inline float &d(int a, int b) { return D[(size_t)a * N + b]; }
  void consider(int x, int y);  // feasible pair: one draw, update mc[x], mc[y] with (lin1=x, lin2=y)
  void initialize();
  void coalesce(int i, int j);

  // phase 1 of a merge (parallel) leaves per cluster: 1 = distances or candidate changed, 2 = candidates rebuilt
  std::vector<unsigned char> kflag;
  std::vector<uint32_t> kmask;  // bit u: d(l_u, k) <= min_values[l_u] for the u-th updated cluster of this merge
  std::vector<int> upos;        // positions (in cluster_index) of the first updated clusters of this merge
  std::vector<std::vector<std::pair<int, int>>> pairs;  // initialize(): half-tested pairs per thread, in order
  std::vector<float> part_cf, part_mvj;  // per-thread partial results of a merge
  std::vector<Cand> part_best;
  std::vector<size_t> part_pos;
  std::vector<std::vector<int>> visit_list;  // per thread: positions phase 2 visits
  std::vector<std::vector<int>> cand_j;      // per thread: clusters close to the merged one (superset)
  BuildThreads pool;
  size_t min_parallel = 512;
};

// MinMatch::QuickBuild with sample ages (`--sample_ages`; minmatch_ages.cpp): the candidates carry the older of the two
// sample ages as a third key, gated by an expected-coalescence clock.  Sequential, stateful across builds where the
// reference is (min_values_CF, the unique-age table, the `cand` scratch candidate).
class MinMatchAges {
  friend class DeviceMinMatch;

 public:
  MinMatchAges(int N, double theta);
  // the sorted distinct sample ages and how many samples have each (tree_builder.cpp:1125-1152): once per builder
  void prepare_levels(const std::vector<double> &sample_ages);
  // d: N*N floats, destroyed.  prior: N*N floats or nullptr.  sample_ages: N values.
  void quick_build(float *d, const float *prior, const std::vector<double> &sample_ages, HostTree &tree);

 private:
  struct Cand {
    int lin1 = -1, lin2 = -1;
    double dist = std::numeric_limits<float>::infinity();
    double dist2 = std::numeric_limits<float>::infinity();
    double dist3 = std::numeric_limits<float>::infinity();
    bool replace = false;
  };
  static bool gt(const Cand &a, const Cand &b);
  void offer(int slot, int lin1, int lin2);
  void consider(int x, int y, float sym, const std::vector<double> &ages);
  void take_best(const Cand &m);
  float sym_of(int x, int y) const;
  void initialize(const std::vector<double> &ages);
  void coalesce(int i, int j, const std::vector<double> &ages);
  inline float &d(int a, int b) { return D[(size_t)a * N + b]; }

  int N, Ne;
  float threshold, threshold_CF;
  std::mt19937 rng;
  std::uniform_real_distribution<double> unif{0.0, 1.0};
  std::vector<int> cluster_index, convert_index, updated_cluster;
  std::vector<float> cluster_size;
  std::vector<Cand> mc;
  Cand best, cand;
  double age = 0.0;
  std::vector<double> unique_ages;
  std::vector<int> ages_count;
  std::vector<float> min_values, min_values_CF;
  std::vector<float> d_CF;
  SymPairs sym;  // merges without a mutually closest pair (sym_pairs.h)
  float *D = nullptr;
  const float *CF = nullptr;
};

// The same builder on the GPU (minmatch_gpu.hip): one workgroup per tree, matrices in HBM.  build() takes the
// state a MinMatch carries from tree to tree from `tb` and puts it back, so the two can alternate: it returns
// 0 when the tree is built, > 0 when this tree needs the host (symmetric fallback; tb untouched), < 0 on error.
class DeviceMinMatch {
 public:
  DeviceMinMatch(int N, int device);
  ~DeviceMinMatch();
  DeviceMinMatch(const DeviceMinMatch &) = delete;
  DeviceMinMatch &operator=(const DeviceMinMatch &) = delete;
  int build(MinMatch &tb, const float *d, const float *prior, HostTree &tree);
  // with sample ages (`--sample_ages`): the builder whose state this one takes and puts back is a MinMatchAges
  int build(MinMatchAges &tb, const std::vector<double> &sample_ages, const float *d, const float *prior, HostTree &tree);
  int build_resident(MinMatchAges &tb, const std::vector<double> &sample_ages, bool with_prior, HostTree &tree);
  // (measurement hook) the matrices of the next build_resident() copied from device memory into the staging pair
  int stage_from_device(const float *dD, const float *dCF);
  void forget_ages();  // the next build with sample ages comes with other ages: their table goes to the device again
  // the buffers a build keeps for itself (the woven matrix, 16 N^2 B, and small ones), allocated now: 0 or < 0
  int reserve(bool ages = false);
  // The same without the matrices crossing PCIe: the caller has the distance matrix written into
  // device_matrix() (K3, rl_window_matrix_rows_device), the carrier penalty and the clade prior of the previous
  // tree are applied on the device (anc_builder.cpp:563-606), build_resident builds from what is there.
  float *device_matrix();
  // where the row minima of the staged distance matrix go when the caller's matrix kernel takes them itself
  // (rl_window_matrix_rows_device_ex), and the caller's word that they are there (with the penalty applied)
  float *rowmin_device();
  void rowmin_is_ready();
  int apply_penalty(const char *member, float val);
  int apply_prior(const HostTree &previous, float val);
  int build_resident(MinMatch &tb, bool with_prior, HostTree &tree);

 private:
  template <class TB>
  int build_impl(TB &tb, const std::vector<double> *sample_ages, const float *d, const float *prior, bool resident,
                 bool with_prior, HostTree &tree);
  struct Impl;
  Impl *impl;
};

// HBM of the pools the device builders of one GPU share (row-major staging matrices, symmetric matrices of the
// fallback), and the pools allocated now rather than at the first tree (the stage's admission counts on them)
double device_builder_shared_bytes(int N);
int device_builder_reserve_shared(int device, int N);
// how many builders of trees of N leaves will ask the device's workers at the same time (0: unknown)
// (ages: the builders with sample ages -- their workers are another kernel)
int device_builder_expect(int device, int N, int builders, bool ages = false);
int device_builder_expect_add(int device, int N, int delta, bool ages = false);
// workgroups of the worker kernel for trees of N leaves a CU holds (1 or 2)
int device_builder_workers_per_cu(int N, bool ages = false);
// the stage's rule for how many resident workers to ask for (treeseq.cpp)
int stage_worker_goal(int cus, int open_sections, bool bounded_windows, int per_cu);

}  // namespace rl
