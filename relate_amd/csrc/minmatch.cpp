// minmatch.cpp -- host tree builder, see minmatch.h.
//
// Reference: src/tree_builder.cpp.  "MinMatch": clusters i,j may merge iff each
// is within `threshold` of the other's row minimum; among feasible pairs the
// smallest d[i][j]+d[j][i] wins, ties broken by a random draw made once per
// feasible pair examined (the draw ORDER is part of the result); merged
// distances are size-weighted means; if no pair is feasible the symmetric
// matrix d[i][j]+d[j][i] picks the pair (:255-293, :968-1058).
#include "minmatch.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

#include "common.h"

namespace rl {

static constexpr size_t PREFETCH_AHEAD = 12;  // clusters; see MinMatch::coalesce (6..48 measure the same)
static inline void pf(const void *p) { __builtin_prefetch(p, 0, 3); }
static constexpr int MAX_GATHER = 32;          // columns of updated clusters gathered per merge

static const float INF = std::numeric_limits<float>::infinity();
static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- helper threads --------------------------------------------------------
static std::atomic<int> g_build_threads{1};
void set_build_threads(int T) { g_build_threads = T < 1 ? 1 : (T > 64 ? 64 : T); }
// fewer live clusters than this: a merge is too short to be worth a hand-off (RELATE_AMD_TEST_BUILD_MIN)
static int build_min_clusters() {
  const char *e = getenv("RELATE_AMD_TEST_BUILD_MIN");
  const int v = e ? atoi(e) : 512;
  return v < 2 ? 2 : v;
}
int build_threads() {
  if (const char *e = getenv("RELATE_AMD_BUILD_THREADS")) {
    const int v = atoi(e);
    return v < 1 ? 1 : (v > 64 ? 64 : v);
  }
  return g_build_threads.load();
}

BuildThreads::BuildThreads(int T) : T_(T < 1 ? 1 : T) {
  for (int t = 1; t < T_; t++) th_.emplace_back([this, t] { worker(t); });
}
BuildThreads::~BuildThreads() {
  stop_ = true;
  {
    std::lock_guard<std::mutex> lk(m_);
    gen_.fetch_add(1);
  }
  cv_.notify_all();
  for (auto &t : th_) t.join();
}
void BuildThreads::worker(int t) {
  uint64_t seen = 0;
  for (;;) {
    int spins = 0;
    while (gen_.load(std::memory_order_acquire) == seen) {  // spin, then sleep
      if (++spins < 20000) {
        __builtin_ia32_pause();
      } else {
        std::unique_lock<std::mutex> lk(m_);
        sleepers_.fetch_add(1);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        sleepers_.fetch_sub(1);
      }
    }
    seen = gen_.load(std::memory_order_acquire);
    if (stop_) return;
    (*job_)(t, T_);
    done_.fetch_add(1, std::memory_order_release);
  }
}
void BuildThreads::run(const std::function<void(int, int)> &job) {
  if (T_ == 1) {
    job(0, 1);
    return;
  }
  job_ = &job;
  done_.store(0, std::memory_order_relaxed);
  if (sleepers_.load() > 0) {
    std::lock_guard<std::mutex> lk(m_);
    gen_.fetch_add(1, std::memory_order_release);
    cv_.notify_all();
  } else {
    gen_.fetch_add(1, std::memory_order_release);
    if (sleepers_.load() > 0) {  // one went to sleep in between
      std::lock_guard<std::mutex> lk(m_);
      cv_.notify_all();
    }
  }
  job(0, T_);
  while (done_.load(std::memory_order_acquire) != T_ - 1) __builtin_ia32_pause();
}

MinMatch::MinMatch(int N_, double theta) : N(N_), sym(N_), pool(N_ >= 2 * build_min_clusters() ? build_threads() : 1) {
  kflag.resize(N);
  kmask.resize(N);
  upos.resize(MAX_GATHER);
  part_cf.resize(64);
  part_mvj.resize(64);
  part_best.resize(64);
  part_pos.resize(64);
  visit_list.resize(64);
  cand_j.resize(64);
  min_parallel = (size_t)build_min_clusters();
  // tree_builder.cpp:43-44 (double log narrowed to float members)
  threshold = -0.2 * std::log(theta / (1.0 - theta));
  threshold_CF = -0.001 * std::log(theta / (1.0 - theta));
  convert_index.resize(N);
  cluster_size.resize(N);
  min_values.resize(N);
  min_values_CF.resize(N);  // zero-initialised and never refilled (:2399-2400)
  mc.resize(N);
  updated_cluster.resize(N);
}

// One feasible pair: symmetric distance (0 if the pair is also mutually
// closest under the prior, :1699-1702), one random draw (:1704), update of
// both clusters' best candidate (:1705-1716).
void MinMatch::consider(int x, int y) {
  if (CF) {
    const bool both = (CF[(size_t)x * N + y] <= min_values_CF[x]) && (CF[(size_t)y * N + x] <= min_values_CF[y]);
    sym_dist = both ? 0.0f : d(y, x) + d(x, y);
  } else {
    sym_dist = d(y, x) + d(x, y);
  }
  dist_random = unif(rng);  // double narrowed to the float member (tree_builder.hpp:60)
  Cand &a = mc[x];
  if (a.dist > sym_dist || (a.dist == sym_dist && a.dist2 > dist_random)) {
    a.lin1 = x; a.lin2 = y; a.dist = sym_dist; a.dist2 = dist_random;
  }
  Cand &b = mc[y];
  if (b.dist > sym_dist || (b.dist == sym_dist && b.dist2 > dist_random)) {
    b.lin1 = x; b.lin2 = y; b.dist = sym_dist; b.dist2 = dist_random;
  }
}

// tree_builder.cpp:59-146 (no prior) / :1647-1735 (prior)
void MinMatch::initialize() {
  const size_t n = cluster_index.size();
  const bool par = pool.size() > 1 && n >= min_parallel;
  // row minima (+ threshold): independent per cluster
  auto rows = [&](int t, int T) {
    const size_t lo = n * (size_t)t / T, hi = n * (size_t)(t + 1) / T;
    for (size_t ia = lo; ia < hi; ia++) {
      const int a = cluster_index[ia];
      mc[a].dist = INF;
      mc[a].dist2 = INF;
      float mv = min_values[a];
      for (int l : cluster_index)
        if (mv > d(a, l) && l != a) mv = d(a, l);
      mv += threshold;
      min_values[a] = mv;
      if (CF) {
        float mc_ = min_values_CF[a];  // carried over from the previous build (:2399-2400)
        for (int l : cluster_index)
          if (mc_ > CF[(size_t)a * N + l] && l != a) mc_ = CF[(size_t)a * N + l];
        mc_ += threshold_CF;
        min_values_CF[a] = mc_;
      }
    }
  };
  if (par)
    pool.run(rows);
  else
    rows(0, 1);
  // mutually close pairs in (a, b) order.  The first half of the test (a row scan) runs in parallel and
  // leaves the few survivors per thread in order; the second half and the random draws stay sequential.
  const int T = par ? pool.size() : 1;
  if ((int)pairs.size() < T) pairs.resize(T);
  auto scan = [&](int t, int TT) {
    // contiguous ranges of a with equal numbers of pairs: a < n(1 - sqrt(1 - t/T))
    auto bound = [&](int x) { return x >= TT ? n : (size_t)((double)n * (1.0 - std::sqrt(1.0 - (double)x / TT))); };
    std::vector<std::pair<int, int>> &out = pairs[t];
    out.clear();
    for (size_t ia = bound(t); ia < bound(t + 1); ia++) {
      const int a = cluster_index[ia];
      const float mva = min_values[a];
      const float *row = D + (size_t)a * N;
      for (size_t ib = ia + 1; ib < n; ib++) {
        const int b = cluster_index[ib];
        if (mva >= row[b]) out.emplace_back(a, b);
      }
    }
  };
  if (par)
    pool.run(scan);
  else
    scan(0, 1);
  for (int t = 0; t < T; t++)
    for (const auto &pr : pairs[t]) {
      const int a = pr.first, b = pr.second;
      if (min_values[b] >= d(b, a)) {
        consider(a, b);
        if (best.dist > mc[b].dist || (best.dist == mc[b].dist && best.dist2 > mc[b].dist2)) {
          best.lin1 = a; best.lin2 = b; best.dist = sym_dist; best.dist2 = mc[b].dist2;
        }
      }
    }
}

// tree_builder.cpp:296-598 (no prior) / :1844-2070 (prior)
void MinMatch::coalesce(int i, int j) {
  const float added = cluster_size[i] + cluster_size[j];
  const float csi = cluster_size[i], csj = cluster_size[j];
  const size_t n = cluster_index.size();
  const double tp0 = now_s();
  float *cf = CF ? d_CF.data() : nullptr;
  const bool par = pool.size() > 1 && n >= min_parallel;
  const int T = par ? pool.size() : 1;

  // The reference loop over the clusters k interleaves several things; iteration k reads and writes only
  // row k, column k and the four entries (i|j, k), (k, i|j) of the matrix, and changes candidate state
  // (mc[], random draws) of k and of clusters BEFORE k only.  So it splits, with identical results, into
  //
  // Phase 1 (parallel over k): the size-weighted update of d(j,k), d(k,j) -- and of the prior's matrix
  //   (coalesce_cf, :2583-2607) --; the re-scan of row k's minimum when the old minimum sat on a changed
  //   entry (:1875-1890); whether k's candidates have to be rebuilt.  The column reads walk the matrices
  //   with a stride of one row (a new cache line each): requested a few clusters ahead.
  auto phase1 = [&](int t, int TT) {
    const size_t lo = n * (size_t)t / TT, hi = n * (size_t)(t + 1) / TT;
    float mv_cf = INF;
    for (size_t ik = lo; ik < hi; ik++) {
      const int k = cluster_index[ik];
      if (ik + PREFETCH_AHEAD < hi) {
        const size_t ro = (size_t)cluster_index[ik + PREFETCH_AHEAD] * N;
        pf(D + ro + j);
        pf(D + ro + i);
        if (cf) {
          pf(cf + ro + j);
          pf(cf + ro + i);
        }
      }
      kmask[ik] = 0;
      if (k == j || k == i) {
        kflag[ik] = 0;
        continue;
      }
      if (cf) {
        const float ckj = cf[(size_t)k * N + j], cki = cf[(size_t)k * N + i];
        const float cik = cf[(size_t)i * N + k], cjk = cf[(size_t)j * N + k];
        if (cik != cjk) cf[(size_t)j * N + k] = (csi * cik + csj * cjk) / added;
        if (cki != ckj) cf[(size_t)k * N + j] = (csi * cki + csj * ckj) / added;
        if (mv_cf > cf[(size_t)j * N + k]) mv_cf = cf[(size_t)j * N + k];
      }
      const float dkj = d(k, j), dki = d(k, i), dik = d(i, k), djk = d(j, k);
      if (dik != djk) d(j, k) = (csi * dik + csj * djk) / added;
      if (dki != dkj) d(k, j) = (csi * dki + csj * dkj) / added;
      bool min_value_changed = false;
      if (dkj != dki) {
        float min_value_k = min_values[k];
        if (std::fabs(min_value_k - threshold - dkj) < 1e-4 || std::fabs(min_value_k - threshold - dki) < 1e-4) {
          // row minimum may have moved: rescan, stop early if the old minimum is still there
          const float min_value_old = min_value_k - threshold;
          min_value_k = INF;
          min_value_changed = true;
          for (int l : cluster_index) {
            if (l != i && l != k) {
              if (min_value_k > d(k, l)) {
                min_value_k = d(k, l);
                if (min_value_k == min_value_old) break;
              }
            }
          }
          min_value_k += threshold;
          min_values[k] = min_value_k;
        }
      }
      const bool touches = mc[k].lin1 == j || mc[k].lin2 == j || mc[k].lin1 == i || mc[k].lin2 == i;
      // 2: k's candidates are rebuilt (it joins the "updated" list).  (A cluster with unchanged distances
      // whose candidate does not touch i or j needs nothing: the reference's renaming of i to j there is a no-op.)
      kflag[ik] = (unsigned char)((min_value_changed || touches) ? 2 : 0);
    }
    part_cf[t] = mv_cf;
  };
  if (par)
    pool.run(phase1);
  else
    phase1(0, 1);
  t_phase1a += now_s() - tp0;
  if (cf) {  // the prior's row minimum of the merged cluster: a plain minimum, order-free
    float mv = INF;
    for (int t = 0; t < T; t++)
      if (mv > part_cf[t]) mv = part_cf[t];
    min_values_CF[j] = mv + threshold_CF;
  }

  // Phase 1b (parallel over k): a pair (k, l) of a later cluster k and an updated cluster l is a candidate when
  //   d(k,l) <= min_values[k] and d(l,k) <= min_values[l].  The reference tests k's half first, one more walk
  //   down column l per updated cluster; both halves are pure, so l's half -- a scan along row l -- goes first
  //   here, into a bit mask per k, and phase 2 looks up k's half for the few survivors only.  The same
  //   sweep reduces what the reference loop accumulates over all k: the row minimum of the merged cluster and
  //   the best candidate among the clusters phase 2 will not visit (first one wins among exact ties, as
  //   in the sequential loop: partial results are combined in cluster order).
  int nu = 0, nupd = 0;
  for (size_t ik = 0; ik < n; ik++)
    if (kflag[ik] & 2) {
      if (nu < MAX_GATHER) upos[nu++] = (int)ik;
      nupd++;
    }
  const bool overflow = nupd > nu;  // more updated clusters than mask bits: phase 2 visits every later cluster
  const size_t overflow_from = overflow ? (size_t)upos[nu - 1] + 1 : n;
  auto sweep = [&](int t, int TT) {
    const size_t lo = n * (size_t)t / TT, hi = n * (size_t)(t + 1) / TT;
    Cand b;
    size_t bpos = n;
    float mvj = INF;
    for (int u = 0; u < nu; u++) {  // row of updated cluster u, clusters after it only
      const int l = cluster_index[upos[u]];
      const float *rowl = D + (size_t)l * N;
      const float mvl = min_values[l];
      const size_t from = std::max(lo, (size_t)upos[u] + 1);
      for (size_t ik = from; ik < hi; ik++)
        if (rowl[cluster_index[ik]] <= mvl) kmask[ik] |= 1u << u;
    }
    const float *rowj = D + (size_t)j * N;
    std::vector<int> &vis = visit_list[t];
    vis.clear();
    for (size_t ik = lo; ik < hi; ik++) {
      const int k = cluster_index[ik];
      if (k == j || k == i) continue;
      if (rowj[k] < mvj) mvj = rowj[k];
      const bool visited = (kflag[ik] & 2) || kmask[ik] != 0 || ik >= overflow_from;
      if (visited) {
        vis.push_back((int)ik);
      } else if (b.dist > mc[k].dist || (b.dist == mc[k].dist && b.dist2 > mc[k].dist2)) {
        b = mc[k];
        bpos = ik;
      }
    }
    part_best[t] = b;
    part_pos[t] = bpos;
    part_mvj[t] = mvj;
    // survivors of the first half of the merged cluster's candidate test, d(j,k) <= min_j + threshold: the
    // slice's own minimum is an upper bound of min_j, so this is a superset; filtered once min_j is known
    std::vector<int> &cj = cand_j[t];
    cj.clear();
    const float bound = mvj + threshold;
    for (size_t ik = lo; ik < hi; ik++) {
      const int k = cluster_index[ik];
      if (rowj[k] <= bound) cj.push_back(k);
    }
  };
  if (par)
    pool.run(sweep);
  else
    sweep(0, 1);
  const double tp1 = now_s();
  t_phase1 += tp1 - tp0;
  n_updated += nupd;
  n_merges++;

  // Phase 2 (in cluster order: it draws the random numbers): the clusters whose candidates change
  Cand sbest;  // best among the visited clusters, and where
  size_t spos = n;
  int ucs = 0;
  auto visit = [&](size_t ik) {
    const int k = cluster_index[ik];
    const float min_value_k = min_values[k];
    if (kflag[ik] & 2) {
      updated_cluster[ucs++] = k;
      mc[k].dist = INF;
      mc[k].dist2 = INF;
      for (size_t il = 0; il < ik; il++) {  // clusters before k in iteration order
        const int l = cluster_index[il];
        if (d(k, l) <= min_value_k) {
          const float min_value_l = min_values[l];
          if (l != j && l != i) {
            if (d(l, k) <= min_value_l) consider(k, l);
          }
        }
      }
    } else {
      for (int u = 0; u < ucs; u++) {
        const int l = updated_cluster[u];
        if (u < nu) {  // l's half of the test is in the mask
          if (((kmask[ik] >> u) & 1u) && d(k, l) <= min_value_k) consider(k, l);
        } else if (d(k, l) <= min_value_k) {
          if (d(l, k) <= min_values[l]) consider(k, l);
        }
      }
    }
    if (sbest.dist > mc[k].dist || (sbest.dist == mc[k].dist && sbest.dist2 > mc[k].dist2)) {
      sbest = mc[k];
      spos = ik;
    }
  };
  for (int t = 0; t < T; t++)  // slices are contiguous and in order
    for (int ik : visit_list[t]) visit((size_t)ik);
  // the loop's running "best": smallest (dist, dist2), the earliest cluster among exact ties
  best.dist = INF;
  best.dist2 = INF;
  size_t bpos = n;
  auto take = [&](const Cand &c, size_t pos) {
    if (pos >= n) return;
    if (best.dist > c.dist || (best.dist == c.dist && (best.dist2 > c.dist2 || (best.dist2 == c.dist2 && pos < bpos)))) {
      best = c;
      bpos = pos;
    }
  };
  for (int t = 0; t < T; t++) take(part_best[t], part_pos[t]);
  take(sbest, spos);
  float min_value_j = INF;
  for (int t = 0; t < T; t++)
    if (part_mvj[t] < min_value_j) min_value_j = part_mvj[t];
  min_value_j += threshold;
  min_values[j] = min_value_j;

  // candidates with the merged cluster j (first half of the test done in the sweep)
  mc[j].dist = INF;
  mc[j].dist2 = INF;
  for (int t = 0; t < T; t++)
    for (int k : cand_j[t])
      if (d(j, k) <= min_value_j) {
        if (d(k, j) <= min_values[k]) {
          if (k != i && k != j) consider(k, j);
        }
      }
  if (best.dist > mc[j].dist || (best.dist == mc[j].dist && best.dist2 > mc[j].dist2)) best = mc[j];
  t_phase2 += now_s() - tp1;
}

// merge i into j in the prior matrix and refresh j's row minimum (:2571-2596)
// tree_builder.cpp:1061-1303 (no prior), :2358-2644 (prior); sample_ages empty
void MinMatch::quick_build(float *dmat, const float *prior, HostTree &tree) {
  rng.seed(1);
  unif.reset();
  D = dmat;
  if (prior) {
    d_CF.assign(prior, prior + (size_t)N * N);
    CF = d_CF.data();
  } else {
    CF = nullptr;
  }
  tree.reset(N);
  cluster_index.resize(N);
  for (int c = 0; c < N; c++) {
    cluster_index[c] = c;
    convert_index[c] = c;
    cluster_size[c] = 1.0f;
  }
  std::fill(min_values.begin(), min_values.end(), INF);
  best.dist = INF;
  best.dist2 = INF;
  sym.reset();

  {
    const double t0 = now_s();
    initialize();
    t_init += now_s() - t0;
  }

  for (int num_nodes = N; num_nodes < 2 * N - 1; num_nodes++) {
    int i, j;
    if (best.dist == INF) {  // no mutually closest pair: the smallest symmetric distance, from here on (sym_pairs.h)
      if (!sym.started()) sym.start(cluster_index, [&](int a, int b) { return d(a, b); });
      i = sym.closest().first;
      j = sym.closest().second;
    } else {
      i = best.lin1;
      j = best.lin2;
    }
    const int conv_i = convert_index[i], conv_j = convert_index[j];
    tree.parent[conv_i] = num_nodes;
    tree.parent[conv_j] = num_nodes;
    tree.num_events[conv_i] = 0.0f;
    tree.num_events[conv_j] = 0.0f;
    tree.child_left[num_nodes] = conv_i;
    tree.child_right[num_nodes] = conv_j;
    coalesce(i, j);
    if (sym.started()) sym.merge(i, j, cluster_size[i], cluster_size[j], cluster_index);
    cluster_size[j] = cluster_size[i] + cluster_size[j];
    convert_index[j] = num_nodes;
    cluster_index.erase(std::find(cluster_index.begin(), cluster_index.end(), i));
  }
  D = nullptr;
  CF = nullptr;
}

}  // namespace rl

extern "C" int rl_quickbuild(int N, double theta, float *d, const float *d_prior, int *parent, int *child_left,
                             int *child_right) {
  if (N < 2 || !d || !parent || !(theta > 0.0 && theta < 1.0)) {
    rl::set_error("rl_quickbuild: bad arguments");
    return RL_EINVAL;
  }
  rl::MinMatch mm(N, theta);
  rl::HostTree t;
  mm.quick_build(d, d_prior, t);
  for (int i = 0; i < 2 * N - 1; i++) parent[i] = t.parent[i];
  for (int i = N; i < 2 * N - 1; i++) {
    if (child_left) child_left[i - N] = t.child_left[i];
    if (child_right) child_right[i - N] = t.child_right[i];
  }
  return RL_OK;
}
