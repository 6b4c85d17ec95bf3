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
#include <cmath>
#include <cstring>

#include "common.h"

namespace rl {

static constexpr size_t PREFETCH_AHEAD = 12;  // clusters; see MinMatch::coalesce

static const float INF = std::numeric_limits<float>::infinity();

MinMatch::MinMatch(int N_, double theta) : N(N_) {
  // tree_builder.cpp:43-44 (double log narrowed to float members)
  threshold = -0.2 * std::log(theta / (1.0 - theta));
  threshold_CF = -0.001 * std::log(theta / (1.0 - theta));
  convert_index.resize(N);
  cluster_size.resize(N);
  min_values.resize(N);
  min_values_sym.resize(N);
  min_values_CF.resize(N);  // zero-initialised and never refilled (:2399-2400)
  mc.resize(N);
  mc_sym.resize(N);
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
  for (int a : cluster_index) {
    mc[a].dist = INF;
    mc[a].dist2 = INF;
    float mv = min_values[a];
    for (int l : cluster_index)
      if (mv > d(a, l) && l != a) mv = d(a, l);
    mv += threshold;
    min_values[a] = mv;
  }
  if (CF) {
    for (int a : cluster_index) {
      float mv = min_values_CF[a];  // carried over from the previous build (:2399-2400)
      for (int l : cluster_index)
        if (mv > CF[(size_t)a * N + l] && l != a) mv = CF[(size_t)a * N + l];
      mv += threshold_CF;
      min_values_CF[a] = mv;
    }
  }
  const size_t n = cluster_index.size();
  for (size_t ia = 0; ia < n; ia++) {
    const int a = cluster_index[ia];
    for (size_t ib = ia + 1; ib < n; ib++) {
      const int b = cluster_index[ib];
      if (min_values[a] >= d(a, b)) {
        if (min_values[b] >= d(b, a)) {
          consider(a, b);
          if (best.dist > mc[b].dist || (best.dist == mc[b].dist && best.dist2 > mc[b].dist2)) {
            best.lin1 = a; best.lin2 = b; best.dist = sym_dist; best.dist2 = mc[b].dist2;
          }
        }
      }
    }
  }
}

// tree_builder.cpp:255-293
void MinMatch::initialize_sym() {
  sym_d.resize((size_t)N * N);
  const size_t n = cluster_index.size();
  for (size_t ia = 0; ia < n; ia++)
    for (size_t ib = ia + 1; ib < n; ib++) {
      const int a = cluster_index[ia], b = cluster_index[ib];
      sym_d[(size_t)a * N + b] = d(a, b) + d(b, a);
      sym_d[(size_t)b * N + a] = sym_d[(size_t)a * N + b];
    }
  for (int a : cluster_index) {
    float &mv = min_values_sym[a];
    mc_sym[a].dist = INF;
    for (int l : cluster_index) {
      const float v = sym_d[(size_t)a * N + l];
      if (mv > v && l != a) {
        mv = v;
        if (mc_sym[a].dist > mv) { mc_sym[a].lin1 = a; mc_sym[a].lin2 = l; mc_sym[a].dist = mv; }
        if (best_sym.dist > mc_sym[a].dist) { best_sym.lin1 = a; best_sym.lin2 = l; best_sym.dist = mv; }
      }
    }
  }
}

// tree_builder.cpp:296-598 (no prior) / :1844-2070 (prior)
void MinMatch::coalesce(int i, int j) {
  const float added = cluster_size[i] + cluster_size[j];
  float min_value_k, min_value_j = INF;
  int ucs = 0;
  best.dist = INF;
  best.dist2 = INF;
  const size_t n = cluster_index.size();
  for (size_t ik = 0; ik < n; ik++) {
    const int k = cluster_index[ik];
    // the two column reads below walk the matrix with a stride of one row (a new cache line and,
    // without huge pages, a new page each): request them a few clusters ahead
    if (ik + PREFETCH_AHEAD < n) {
      const float *nxt = D + (size_t)cluster_index[ik + PREFETCH_AHEAD] * N;
      __builtin_prefetch(nxt + j, 1);
      __builtin_prefetch(nxt + i, 0);
    }
    if (k == j || k == i) continue;
    const float dkj = d(k, j), dki = d(k, i), dik = d(i, k), djk = d(j, k);
    min_value_k = min_values[k];
    if (dik != djk) d(j, k) = (cluster_size[i] * dik + cluster_size[j] * djk) / added;
    if (dki != dkj) d(k, j) = (cluster_size[i] * dki + cluster_size[j] * dkj) / added;

    bool min_value_changed = false;
    if (dkj != dki) {
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
    if (dkj != dki || djk != dik || touches) {
      if (min_value_changed || touches) {
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
          if (d(k, l) <= min_value_k) {
            if (d(l, k) <= min_values[l]) consider(k, l);
          }
        }
      }
    } else {
      if (mc[k].lin1 == i) mc[k].lin1 = j;
      if (mc[k].lin2 == i) mc[k].lin2 = j;
      for (int u = 0; u < ucs; u++) {
        const int l = updated_cluster[u];
        if (d(k, l) <= min_value_k) {
          if (d(l, k) <= min_values[l]) consider(k, l);
        }
      }
    }
    if (best.dist > mc[k].dist || (best.dist == mc[k].dist && best.dist2 > mc[k].dist2)) best = mc[k];
    if (d(j, k) < min_value_j) min_value_j = d(j, k);
  }
  min_value_j += threshold;
  min_values[j] = min_value_j;

  // candidates with the merged cluster j
  mc[j].dist = INF;
  mc[j].dist2 = INF;
  for (int k : cluster_index) {
    if (d(j, k) <= min_value_j) {
      if (d(k, j) <= min_values[k]) {
        if (k != i && k != j) consider(k, j);
      }
    }
  }
  if (best.dist > mc[j].dist || (best.dist == mc[j].dist && best.dist2 > mc[j].dist2)) best = mc[j];
}

// tree_builder.cpp:968-1058
void MinMatch::coalesce_sym(int i, int j) {
  const float added = cluster_size[i] + cluster_size[j];
  float min_value_k, min_value_j = INF;
  auto s = [&](int a, int b) -> float & { return sym_d[(size_t)a * N + b]; };
  best_sym.dist = INF;
  mc_sym[j].dist = INF;
  const size_t n = cluster_index.size();
  for (size_t ik = 0; ik < n; ik++) {
    const int k = cluster_index[ik];
    if (ik + PREFETCH_AHEAD < n) {
      const float *nxt = sym_d.data() + (size_t)cluster_index[ik + PREFETCH_AHEAD] * N;
      __builtin_prefetch(nxt + j, 1);
      __builtin_prefetch(nxt + i, 0);
    }
    if (k == j || k == i) continue;
    const float dkj = s(k, j), dki = s(k, i), dik = s(i, k), djk = s(j, k);
    min_value_k = min_values_sym[k];
    if (dik != djk) s(j, k) = (cluster_size[i] * dik + cluster_size[j] * djk) / added;
    if (dki != dkj) s(k, j) = (cluster_size[i] * dki + cluster_size[j] * dkj) / added;
    if (dkj != dki) {
      if (std::fabs(min_value_k - dkj) < 1e-6 || std::fabs(min_value_k - dki) < 1e-6) {
        const float min_value_old = min_value_k;
        min_value_k = INF;
        mc_sym[k].dist = INF;
        for (int l : cluster_index) {
          if (l != i && l != k) {
            if (min_value_k > s(k, l)) {
              min_value_k = s(k, l);
              if (mc_sym[k].dist > min_value_k) { mc_sym[k].lin1 = k; mc_sym[k].lin2 = l; mc_sym[k].dist = min_value_k; }
              if (min_value_k == min_value_old) break;
            }
          }
        }
        min_values_sym[k] = min_value_k;
      }
    } else {
      if (mc_sym[k].lin1 == i) mc_sym[k].lin1 = j;
      if (mc_sym[k].lin2 == i) mc_sym[k].lin2 = j;
    }
    if (best_sym.dist > mc_sym[k].dist) best_sym = mc_sym[k];
    if (s(j, k) < min_value_j) {
      min_value_j = s(j, k);
      if (mc_sym[j].dist > s(j, k)) { mc_sym[j].lin1 = k; mc_sym[j].lin2 = j; mc_sym[j].dist = s(j, k); }
    }
  }
  min_values_sym[j] = min_value_j;
  if (best_sym.dist > mc_sym[j].dist) best_sym = mc_sym[j];
}

// merge i into j in the prior matrix and refresh j's row minimum (:2571-2596)
void MinMatch::coalesce_cf(int i, int j) {
  float *cf = d_CF.data();
  min_values_CF[j] = INF;
  const float added = cluster_size[i] + cluster_size[j];
  const size_t n = cluster_index.size();
  for (size_t ik = 0; ik < n; ik++) {
    const int k = cluster_index[ik];
    if (ik + PREFETCH_AHEAD < n) {
      const float *nxt = cf + (size_t)cluster_index[ik + PREFETCH_AHEAD] * N;
      __builtin_prefetch(nxt + j, 1);
      __builtin_prefetch(nxt + i, 0);
    }
    if (k == j || k == i) continue;
    const float dkj = cf[(size_t)k * N + j], dki = cf[(size_t)k * N + i];
    const float dik = cf[(size_t)i * N + k], djk = cf[(size_t)j * N + k];
    if (dik != djk) cf[(size_t)j * N + k] = (cluster_size[i] * dik + cluster_size[j] * djk) / added;
    if (dki != dkj) cf[(size_t)k * N + j] = (cluster_size[i] * dki + cluster_size[j] * dkj) / added;
    if (min_values_CF[j] > cf[(size_t)j * N + k]) min_values_CF[j] = cf[(size_t)j * N + k];
  }
  min_values_CF[j] += threshold_CF;
}

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
  std::fill(min_values_sym.begin(), min_values_sym.end(), INF);
  best.dist = INF;
  best.dist2 = INF;
  best_sym.dist = INF;

  initialize();

  bool use_sym = false;
  for (int num_nodes = N; num_nodes < 2 * N - 1; num_nodes++) {
    int i, j;
    if (best.dist == INF) {  // no mutually-closest pair: symmetric fallback
      if (!use_sym) {
        initialize_sym();
        use_sym = true;
      }
      i = best_sym.lin1;
      j = best_sym.lin2;
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
    if (CF) coalesce_cf(i, j);
    coalesce(i, j);
    if (use_sym) coalesce_sym(i, j);
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
