// minmatch_ages.cpp -- MinMatch::QuickBuild with sample ages (`--sample_ages`: ancient samples), on the host.
//
// Reference: src/tree_builder.cpp -- the third candidate key and its comparison (:3-22), Initialize with ages
// (:149-252, with a prior :1738-1841), Coalesce with ages (:601-965, with a prior :2073-2355), the
// expected-coalescence clock of QuickBuild (:1123-1233, with a prior :2407-2531).  A pair of clusters carries
// dist3 = the older of its two sample ages; a candidate older than the clock `age` (which advances by
// 2 / (k (k-1)) * Ne per merge, k = lineages alive at the current sampling level) may only fill an empty slot and is
// marked `replace`: any candidate within the clock displaces it, whatever its distance.
//
// Sequential, as the reference: data sets with ancient samples are small, and every decision below -- which slot a
// pair may take, what `replace` a copy carries, when the clock steps -- is order-dependent.  tmpl_tree is never
// passed by BuildTopology (anc_builder.cpp:447, :608, :612), so the template branches do not exist here.
#include <algorithm>
#include <cmath>
#include <limits>

#include "minmatch.h"

namespace rl {

namespace {
const float INF = std::numeric_limits<float>::infinity();
}

// tree_builder.cpp:7-22
bool MinMatchAges::gt(const Cand &a, const Cand &b) {
  if (a.replace && a.dist3 >= b.dist3) {
    if (a.dist3 > b.dist3) return true;
    if (a.dist > b.dist || (a.dist == b.dist && a.dist2 > b.dist2)) return true;
  }
  return a.dist > b.dist || (a.dist == b.dist && a.dist2 > b.dist2);
}

MinMatchAges::MinMatchAges(int N_, double theta) : N(N_), sym(N_) {
  Ne = (int)std::max(17.5f * N, 30000.0f);  // pipeline/BuildTopology.cpp:36 (int Data::Ne)
  threshold = -0.2 * std::log(theta / (1.0 - theta));  // tree_builder.cpp:43-44
  threshold_CF = -0.001 * std::log(theta / (1.0 - theta));
  convert_index.resize(N);
  cluster_size.resize(N);
  min_values.resize(N);
  min_values_CF.resize(N);  // zero-initialised and never refilled (:2399-2400)
  mc.resize(N);
  updated_cluster.resize(N);
}

// one slot of a feasible pair: the candidate `cand` may take it if the slot is empty or the pair is within the clock,
// and the slot's candidate "is greater" (:209-218 and every block like it)
void MinMatchAges::offer(int slot, int lin1, int lin2) {
  Cand &m = mc[slot];
  if ((m.dist == INF || cand.dist3 <= age) && gt(m, cand)) {
    cand.replace = cand.dist3 > age;
    m = cand;
    m.lin1 = lin1;
    m.lin2 = lin2;
  }
}

// a feasible pair (x visits y): key 3, one draw -- kept as a double here (Candidate::dist2, tree_builder.hpp:26) --,
// both slots
void MinMatchAges::consider(int x, int y, float sym, const std::vector<double> &ages) {
  cand.dist = sym;
  cand.dist3 = std::max(ages[x], ages[y]);
  cand.dist2 = unif(rng);
  offer(x, x, y);
  offer(y, x, y);
}

void MinMatchAges::take_best(const Cand &m) {  // :230-237, :874-881, :956-963
  if ((best.dist == INF || m.dist3 <= age) && gt(best, m)) {
    best = m;
    best.replace = best.dist3 > age;
  }
}

// 0 if the pair is also mutually closest under the prior, else d(x,y) + d(y,x) (:2152-2155 and the blocks like it)
float MinMatchAges::sym_of(int x, int y) const {
  if (CF) {
    float s = 1 - (CF[(size_t)y * N + x] <= min_values_CF[y]) * (CF[(size_t)x * N + y] <= min_values_CF[x]);
    if (s > 0) s = D[(size_t)y * N + x] + D[(size_t)x * N + y];
    return s;
  }
  return D[(size_t)y * N + x] + D[(size_t)x * N + y];
}

// tree_builder.cpp:149-252 / :1738-1841
void MinMatchAges::initialize(const std::vector<double> &ages) {
  for (int a : cluster_index) {
    mc[a].dist = mc[a].dist2 = mc[a].dist3 = INF;
    mc[a].replace = false;
    float mv = min_values[a];
    for (int l : cluster_index)
      if (mv > d(a, l) && l != a) mv = d(a, l);
    min_values[a] = mv + threshold;
  }
  if (CF)
    for (int a : cluster_index) {
      float mv = min_values_CF[a];  // carried over from the previous build
      for (int l : cluster_index)
        if (mv > CF[(size_t)a * N + l] && l != a) mv = CF[(size_t)a * N + l];
      min_values_CF[a] = mv + threshold_CF;
    }
  const size_t n = cluster_index.size();
  for (size_t ia = 0; ia < n; ia++) {
    const int a = cluster_index[ia];
    for (size_t ib = ia + 1; ib < n; ib++) {
      const int b = cluster_index[ib];
      if (min_values[a] >= d(a, b) && min_values[b] >= d(b, a)) {
        float sym;
        if (CF) {
          // (the initialisation with a prior keeps the pairs the prior agrees with and voids the others, :1792-1797 --
          //  the opposite of what the merges do with the same test)
          sym = 1 - (CF[(size_t)a * N + b] <= min_values_CF[a]) * (CF[(size_t)b * N + a] <= min_values_CF[b]);
          sym = sym == 0 ? d(a, b) + d(b, a) : INF;
        } else {
          sym = d(a, b) + d(b, a);
        }
        consider(a, b, sym, ages);
        take_best(mc[b]);
      }
    }
  }
}

// tree_builder.cpp:601-965 / :2073-2355
void MinMatchAges::coalesce(int i, int j, const std::vector<double> &ages) {
  const float added = cluster_size[i] + cluster_size[j];
  float min_value_j = INF;
  int updated = 0;
  best.dist = best.dist2 = best.dist3 = INF;
  best.replace = false;
  const size_t n = cluster_index.size();
  for (size_t ik = 0; ik < n; ik++) {
    const int k = cluster_index[ik];
    if (k == j || k == i) continue;
    const float dkj = d(k, j), dki = d(k, i), dik = d(i, k), djk = d(j, k);
    float min_value_k = min_values[k];
    if (mc[k].dist3 <= age) mc[k].replace = false;
    if (dik != djk) d(j, k) = (cluster_size[i] * dik + cluster_size[j] * djk) / added;
    if (dki != dkj) d(k, j) = (cluster_size[i] * dki + cluster_size[j] * dkj) / added;
    bool min_value_changed = false;
    if (dkj != dki) {
      if (std::fabs(min_value_k - threshold - dkj) < 1e-4 || std::fabs(min_value_k - threshold - dki) < 1e-4) {
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
    // (without a prior a candidate touching i or j alone sends k through the first branch, :653; with one it does
    //  not, :2131)
    const bool moved = dkj != dki || djk != dik || (!CF && touches);
    auto meet_updated = [&]() {  // k against the clusters rebuilt so far in this merge
      for (int u = 0; u < updated; u++) {
        const int l = updated_cluster[u];
        if (d(k, l) <= min_value_k)
          if (d(l, k) <= min_values[l]) consider(k, l, sym_of(k, l), ages);
      }
    };
    if (moved) {
      if (min_value_changed || touches) {
        updated_cluster[updated++] = k;
        mc[k].dist = mc[k].dist2 = mc[k].dist3 = INF;
        mc[k].replace = false;
        for (size_t il = 0; il < ik; il++) {
          const int l = cluster_index[il];
          if (d(k, l) <= min_value_k) {
            const float min_value_l = min_values[l];
            if (l != j && l != i)
              if (d(l, k) <= min_value_l) consider(k, l, sym_of(k, l), ages);
          }
        }
      } else {
        meet_updated();
      }
    } else {
      if (mc[k].lin1 == i) mc[k].lin1 = j;
      if (mc[k].lin2 == i) mc[k].lin2 = j;
      meet_updated();
    }
    take_best(mc[k]);
    if (d(j, k) < min_value_j) min_value_j = d(j, k);
  }
  min_value_j += threshold;
  min_values[j] = min_value_j;
  mc[j].dist = mc[j].dist2 = mc[j].dist3 = INF;
  mc[j].replace = false;
  for (int k : cluster_index)
    if (d(j, k) <= min_value_j)
      if (d(k, j) <= min_values[k])
        if (k != i && k != j) consider(k, j, sym_of(k, j), ages);
  take_best(mc[j]);
}

// :1125-1152, once per builder
void MinMatchAges::prepare_levels(const std::vector<double> &ages) {
  if (!unique_ages.empty()) return;
  std::vector<double> sorted = ages;
  std::sort(sorted.begin(), sorted.end());
  double a = sorted[0];
  unique_ages.assign(sorted.size(), 0.0);
  ages_count.assign(sorted.size(), 0);
  int u = 0;
  unique_ages[0] = a;
  for (double x : sorted) {
    if (x == a) {
      ages_count[u]++;
    } else {
      a = x;
      u++;
      unique_ages[u] = a;
      ages_count[u]++;
    }
  }
  unique_ages.resize(u + 1);
  ages_count.resize(u + 1);
}

// tree_builder.cpp:1061-1233 (no prior), :2358-2531 (prior); sample_ages.size() == N
void MinMatchAges::quick_build(float *dmat, const float *prior, const std::vector<double> &sample_ages_in, HostTree &tree) {
  rng.seed(1);
  unif.reset();
  D = dmat;
  std::vector<double> ages = sample_ages_in;
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
  best.dist = best.dist2 = best.dist3 = INF;
  if (!prior) best.replace = false;  // (:1117 -- the build with a prior leaves the flag as the last build left it)
  sym.reset();

  prepare_levels(ages);
  int level = 0;
  int num_lins = ages_count[level];
  // the clock starts one expected coalescence in without a prior (:1155), at the youngest samples with one (:2440)
  age = prior ? unique_ages[level] : unique_ages[level] + 2.0 / ((double)num_lins * (num_lins - 1.0)) * Ne;

  initialize(ages);

  for (int num_nodes = N; num_nodes < 2 * N - 1; num_nodes++) {
    int i, j;
    if (best.dist == INF) {  // (sym_pairs.h)
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

    if (CF) {  // the same merge in the prior's matrix, and cluster j's minimum there (:2482-2511)
      min_values_CF[j] = INF;
      const float added = cluster_size[i] + cluster_size[j];
      float *cf = d_CF.data();
      for (int k : cluster_index) {
        if (k == j || k == i) continue;
        const float ckj = cf[(size_t)k * N + j], cki = cf[(size_t)k * N + i];
        const float cik = cf[(size_t)i * N + k], cjk = cf[(size_t)j * N + k];
        if (cik != cjk) cf[(size_t)j * N + k] = (cluster_size[i] * cik + cluster_size[j] * cjk) / added;
        if (cki != ckj) cf[(size_t)k * N + j] = (cluster_size[i] * cki + cluster_size[j] * ckj) / added;
        if (min_values_CF[j] > cf[(size_t)j * N + k]) min_values_CF[j] = cf[(size_t)j * N + k];
      }
      min_values_CF[j] += threshold_CF;
    }
    coalesce(i, j, ages);
    if (sym.started()) sym.merge(i, j, cluster_size[i], cluster_size[j], cluster_index);

    ages[j] = std::max(ages[i], ages[j]);
    if (prior) age += 2.0 / ((double)num_lins * (num_lins - 1.0)) * Ne;  // (:2516: before the lineage count drops)
    num_lins--;
    while (unique_ages[level] < ages[j]) {
      level++;
      num_lins += ages_count[level];
    }
    if (!prior) age += 2.0 / ((double)num_lins * (num_lins - 1.0)) * Ne;  // (:1226: after)
    cluster_size[j] = cluster_size[i] + cluster_size[j];
    convert_index[j] = num_nodes;
    cluster_index.erase(std::find(cluster_index.begin(), cluster_index.end(), i));
  }
  D = nullptr;
  CF = nullptr;
}

}  // namespace rl
