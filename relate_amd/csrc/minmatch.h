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

#include <cstdint>
#include <cstdlib>
#include <new>
#include <limits>
#include <random>
#include <vector>

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

class MinMatch {
 public:
  MinMatch(int N, double theta);
  // d: N*N floats, destroyed.  prior: N*N floats or nullptr.
  void quick_build(float *d, const float *prior, HostTree &tree);

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
  std::vector<Cand> mc, mc_sym;
  Cand best, best_sym;
  std::vector<float> min_values, min_values_sym, min_values_CF;
  MatrixBuf sym_d, d_CF;
  float sym_dist = 0.f, dist_random = 0.f;

  float *D = nullptr;         // current asymmetric matrix
  const float *CF = nullptr;  // current prior matrix (d_CF) or nullptr
  std::uniform_real_distribution<double> unif{0.0, 1.0};

  inline float &d(int a, int b) { return D[(size_t)a * N + b]; }
  void consider(int x, int y);  // feasible pair: one draw, update mc[x], mc[y] with (lin1=x, lin2=y)
  void initialize();
  void initialize_sym();
  void coalesce(int i, int j);
  void coalesce_sym(int i, int j);
  void coalesce_cf(int i, int j);
};

}  // namespace rl
