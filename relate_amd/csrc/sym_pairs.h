// sym_pairs.h -- MinMatch's way out when no pair of clusters is mutually closest: from that merge on the pair with
// the smallest SYMMETRIC distance s(a,b) = d(a,b) + d(b,a) is merged whenever the asymmetric search comes up empty.
//
// Behaviour: MinMatch::InitializeSym / CoalesceSym (tree_builder.cpp:255-293, :968-1058).  What has to be reproduced
// is which pair wins, ties included:
//   * a cluster's partner is the FIRST live cluster (in the order of the live list) that reaches its row minimum;
//   * the pair to merge is that of the FIRST cluster whose row minimum is the smallest (the merged cluster j is
//     looked at after all the others);
//   * after a merge a row is searched again only if its entries towards i and j differed AND its minimum sat
//     (within 1e-6) on one of them; the search stops early on an entry equal to the old minimum.  A row whose two
//     entries differed but whose minimum sat elsewhere keeps its partner as it is -- even if that partner is i,
//     which no longer exists; a row whose two entries agreed has i renamed to j;
//   * the row of j is rebuilt from scratch, its pair recorded as (partner, j).
// The state is laid out as three passes over the live list per merge (means of row / column j; rows to search again;
// the winner) instead of one interleaved loop: a row k reads nothing of the merge but its own entry (k, j), so the
// passes see the values the single loop sees.  One instance serves both host builders (minmatch.cpp,
// minmatch_ages.cpp); the matrix is kept from build to build, nothing else is.
#pragma once
#include <cmath>
#include <cstddef>
#include <limits>
#include <utility>
#include <vector>

namespace rl {

class SymPairs {
 public:
  struct Pair {
    int first = -1, second = -1;
    float dist = std::numeric_limits<float>::infinity();
  };

  explicit SymPairs(int N) : N_(N), rowmin_((size_t)N), pair_((size_t)N), differed_((size_t)N) {}

  // a new build: the symmetric search has not started
  void reset() {
    started_ = false;
    best_ = Pair();
  }
  bool started() const { return started_; }
  const Pair &closest() const { return best_; }

  // d(a, b): the asymmetric distance as it stands; live: the clusters alive, in the builder's order
  template <class Dist>
  void start(const std::vector<int> &live, Dist &&d) {
    if (s_.size() != (size_t)N_ * N_) s_.assign((size_t)N_ * N_, 0.0f);
    const size_t n = live.size();
    for (size_t x = 0; x < n; x++)
      for (size_t y = x + 1; y < n; y++) {
        const int a = live[x], b = live[y];
        const float v = d(a, b) + d(b, a);
        at(a, b) = v;
        at(b, a) = v;
      }
    best_ = Pair();
    for (int a : live) {
      search_row(a, live, a, a, false, 0.0f);
      take(pair_[a]);
    }
    started_ = true;
  }

  // clusters i and j (sizes size_i, size_j) become j; `live` still lists both
  void merge(int i, int j, float size_i, float size_j, const std::vector<int> &live) {
    const float total = size_i + size_j;
    // 1. row and column j: size-weighted means where i's and j's entries differ (the others stay bit for bit)
    for (int k : live) {
      if (k == i || k == j) continue;
      float &jk = at(j, k), &kj = at(k, j);
      const float ik = at(i, k), ki = at(k, i), kj_before = kj;
      if (ik != jk) jk = (size_i * ik + size_j * jk) / total;
      if (ki != kj_before) kj = (size_i * ki + size_j * kj_before) / total;
      // 0: the entries agreed; 1: they differed; 2: ... and the row's minimum sat on one of them
      differed_[k] = ki == kj_before ? 0
                     : (std::fabs(rowmin_[k] - kj_before) < 1e-6 || std::fabs(rowmin_[k] - ki) < 1e-6) ? 2
                                                                                                        : 1;
    }
    // 2. the other rows: searched again, left alone, or renamed
    for (int k : live) {
      if (k == i || k == j) continue;
      if (differed_[k] == 2) {
        search_row(k, live, i, k, true, rowmin_[k]);
      } else if (differed_[k] == 0) {
        if (pair_[k].first == i) pair_[k].first = j;
        if (pair_[k].second == i) pair_[k].second = j;
      }
    }
    // 3. row j anew -- recorded as (partner, j) --, then the winner: the others in order, j last
    search_row(j, live, i, j, false, 0.0f);
    std::swap(pair_[j].first, pair_[j].second);
    best_ = Pair();
    for (int k : live)
      if (k != i && k != j) take(pair_[k]);
    take(pair_[j]);
  }

 private:
  static constexpr float kInf = std::numeric_limits<float>::infinity();
  float &at(int a, int b) { return s_[(size_t)a * N_ + b]; }
  void take(const Pair &p) {
    if (best_.dist > p.dist) best_ = p;
  }
  // row a over the live clusters other than skip1 / skip2: the first one to reach the minimum; early: stop on a new
  // minimum equal to stop_at.  Nothing found: the pair keeps its clusters, at infinity.
  void search_row(int a, const std::vector<int> &live, int skip1, int skip2, bool early, float stop_at) {
    float m = kInf;
    pair_[a].dist = kInf;
    const float *row = &s_[(size_t)a * N_];
    for (int l : live) {
      if (l == skip1 || l == skip2) continue;
      if (m > row[l]) {
        m = row[l];
        pair_[a].first = a;
        pair_[a].second = l;
        pair_[a].dist = m;
        if (early && m == stop_at) break;
      }
    }
    rowmin_[a] = m;
  }

  int N_;
  bool started_ = false;
  std::vector<float> s_;       // [N][N], symmetric
  std::vector<float> rowmin_;  // per cluster: the minimum of its row as last searched
  std::vector<Pair> pair_;     // per cluster: (itself, partner) -- (partner, j) for a merged cluster
  std::vector<unsigned char> differed_;
  Pair best_;
};

}  // namespace rl
