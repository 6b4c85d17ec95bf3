// equivalent.cpp -- the stage after BuildTopology: find equivalent branches in
// neighbouring trees and carry branch attributes across them (host code).
//
// Restates pipeline/FindEquivalentBranches.cpp:13-167 with
//   AncesTreeBuilder::PreCalcPotentialBranches  (src/anc_builder.cpp:1432-1452)
//   AncesTreeBuilder::BranchAssociation         (src/anc_builder.cpp:1454-1613)
//   AncesTreeBuilder::AssociateTrees            (src/anc_builder.cpp:658-800)
//   Correlation::Pearson                        (src/anc.cpp:823-859)
//   Tree::FindAllLeaves / ReadTreeBin           (src/anc.cpp:450-520, 83-125)
// File in -> file out: reads <out>/chunk_<c>/<out>_<w>.anc of every window,
// rewrites them with num_events / SNP_begin / SNP_end propagated along
// equivalent branches.  The reference parks the per-window equivalence tables
// in equivalent_branches_<w>.bin and deletes them at the end; here they stay in
// memory.  std::sort is called on the same data in the same way, so unstable
// tie orders agree with the reference built against the same libstdc++.
#include <tgmath.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

using namespace rl;

namespace {

struct AncTree {
  int pos = 0;
  std::vector<int> parent, snp_begin, snp_end;
  std::vector<double> branch_length;
  std::vector<float> num_events;
  std::vector<int> child_left, child_right;  // as Tree::ReadTreeBin assigns them: first / second child in node order
};

struct AncFile {
  bool has_ages = false;
  unsigned N = 0;
  std::vector<double> ages;
  std::vector<AncTree> trees;
};

// AncesTree::ReadBin (src/anc.cpp:941-968) + Tree::ReadTreeBin (:83-125)
int read_anc(const std::string &fn, AncFile &a) {
  FILE *fp = fopen(fn.c_str(), "rb");
  if (!fp) {
    set_error("cannot open %s", fn.c_str());
    return RL_EIO;
  }
  unsigned T = 0;
  bool ok = fread(&a.has_ages, sizeof(bool), 1, fp) == 1 && fread(&a.N, 4, 1, fp) == 1;
  if (ok && a.has_ages) {
    a.ages.resize(a.N);
    ok = fread(a.ages.data(), 8, a.N, fp) == a.N;
  }
  ok = ok && fread(&T, 4, 1, fp) == 1;
  const int nodes = 2 * (int)a.N - 1;
  a.trees.assign(ok ? T : 0, AncTree());
  for (unsigned t = 0; ok && t < T; t++) {
    AncTree &tr = a.trees[t];
    tr.parent.resize(nodes);
    tr.snp_begin.resize(nodes);
    tr.snp_end.resize(nodes);
    tr.branch_length.resize(nodes);
    tr.num_events.resize(nodes);
    tr.child_left.assign(nodes, -1);
    tr.child_right.assign(nodes, -1);
    ok = fread(&tr.pos, 4, 1, fp) == 1;
    for (int i = 0; ok && i < nodes; i++) {
      ok = fread(&tr.parent[i], 4, 1, fp) == 1 && fread(&tr.branch_length[i], 8, 1, fp) == 1 &&
           fread(&tr.num_events[i], 4, 1, fp) == 1 && fread(&tr.snp_begin[i], 4, 1, fp) == 1 &&
           fread(&tr.snp_end[i], 4, 1, fp) == 1;
      const int p = tr.parent[i];
      if (ok && p != -1) {
        if (p < 0 || p >= nodes) {
          ok = false;
        } else if (tr.child_left[p] == -1) {
          tr.child_left[p] = i;
        } else {
          tr.child_right[p] = i;
        }
      }
    }
  }
  fclose(fp);
  if (!ok) {
    set_error("%s: truncated or malformed .anc file", fn.c_str());
    return RL_EIO;
  }
  return RL_OK;
}

// AncesTree::DumpBin (src/anc.cpp:1104-1167)
int write_anc(const std::string &fn, const AncFile &a) {
  FILE *fp = fopen(fn.c_str(), "wb");
  if (!fp) {
    set_error("cannot open %s for writing", fn.c_str());
    return RL_EIO;
  }
  const unsigned T = (unsigned)a.trees.size();
  fwrite(&a.has_ages, sizeof(bool), 1, fp);
  fwrite(&a.N, 4, 1, fp);
  if (a.has_ages) fwrite(a.ages.data(), 8, a.N, fp);
  fwrite(&T, 4, 1, fp);
  const int nodes = 2 * (int)a.N - 1;
  for (const AncTree &tr : a.trees) {
    fwrite(&tr.pos, 4, 1, fp);
    for (int i = 0; i < nodes; i++) {
      fwrite(&tr.parent[i], 4, 1, fp);
      fwrite(&tr.branch_length[i], 8, 1, fp);
      fwrite(&tr.num_events[i], 4, 1, fp);
      fwrite(&tr.snp_begin[i], 4, 1, fp);
      fwrite(&tr.snp_end[i], 4, 1, fp);
    }
  }
  fclose(fp);
  return RL_OK;
}

struct Leaves {
  std::vector<int> member;  // sorted leaf labels below the node
  int num_leaves = 0;
};

// Tree::FindLeaves (src/anc.cpp:470-520): post-order merge of the children's sorted lists
void find_leaves(const AncTree &t, int node, std::vector<Leaves> &lv) {
  if (t.child_left[node] != -1) {
    const int c1 = t.child_left[node], c2 = t.child_right[node];
    find_leaves(t, c1, lv);
    find_leaves(t, c2, lv);
    Leaves &out = lv[node];
    out.member.resize(lv[c1].member.size() + lv[c2].member.size());
    std::merge(lv[c1].member.begin(), lv[c1].member.end(), lv[c2].member.begin(), lv[c2].member.end(),
               out.member.begin());
    out.num_leaves = lv[c1].num_leaves + lv[c2].num_leaves;
  } else {
    lv[node].member.assign(1, node);
    lv[node].num_leaves = 1;
  }
}
// Tree::FindAllLeaves (:450-467)
void find_all_leaves(const AncTree &t, std::vector<Leaves> &lv) {
  const int nodes = (int)t.parent.size(), N = (nodes + 1) / 2;
  lv.assign(nodes, Leaves());
  int root = nodes - 1;
  if (t.parent[root] != -1)
    for (int i = N; i < nodes; i++)
      if (t.parent[i] == -1) {
        root = i;
        break;
      }
  find_leaves(t, root, lv);
}

// Correlation::Pearson (src/anc.cpp:823-859): float arithmetic as written there
struct Correlation {
  int N;
  float N_float;
  explicit Correlation(int n) : N(n), N_float((float)n) {}
  float pearson(const Leaves &set1, const Leaves &set2) const {
    if (set1.num_leaves == N || set2.num_leaves == N) {
      if (set1.num_leaves == set2.num_leaves) return 1;
      return 0;
    }
    float prod = 0.0;
    auto i1 = set1.member.begin(), i2 = set2.member.begin();
    const auto e1 = set1.member.end(), e2 = set2.member.end();
    while (i1 != e1 && i2 != e2) {
      if (*i1 == *i2) {
        prod += 1.0;
        i1++;
        i2++;
      } else if (*i1 < *i2) {
        i1++;
      } else {
        i2++;
      }
    }
    if (prod == set1.num_leaves && prod == set2.num_leaves) return 1.0;
    float r = prod - set1.num_leaves * (((float)set2.num_leaves) / N_float);
    if (r <= 0.0) return 0.0;
    r /= sqrt(((((float)set1.num_leaves) / N_float) * (N_float - set1.num_leaves)) *
              ((((float)set2.num_leaves) / N_float) * (N_float - set2.num_leaves)));
    return r;
  }
};

struct EquivalentNode {
  int node1, node2;
  float corr;
  bool operator>(const EquivalentNode &n) const { return corr > n.corr; }
};

struct BranchMatcher {
  int N, N_total;
  float threshold_brancheq = 0.95;
  std::vector<std::vector<int>> potential_branches;

  explicit BranchMatcher(int n) : N(n), N_total(2 * n - 1) {
    // PreCalcPotentialBranches (src/anc_builder.cpp:1432-1452)
    potential_branches.resize(N);
    float threshold_inv = 1 / (threshold_brancheq * threshold_brancheq);
    float N_float = N;
    for (int i = 1; i <= N; i++) {
      potential_branches[i - 1].push_back(i);
      for (int j = i + 1; j <= N; j++) {
        if (threshold_inv >= j / (N_float - j) * ((N_float - i) / i)) {
          potential_branches[i - 1].push_back(j);
          potential_branches[j - 1].push_back(i);
        }
      }
    }
  }

  // BranchAssociation (src/anc_builder.cpp:1454-1613): eq[i] = branch of ref_tree equivalent to branch i of tree
  void associate(const AncTree &ref_tree, const AncTree &tree, std::vector<int> &eq) const {
    eq.assign(N_total, -1);
    std::vector<int> eq_ref(N_total, -1);
    Correlation cor(N);
    std::vector<Leaves> tr_leaves, rtr_leaves;
    find_all_leaves(tree, tr_leaves);
    find_all_leaves(ref_tree, rtr_leaves);

    std::vector<int> sorted_branches(N_total);
    std::size_t n(0);
    std::generate(std::begin(sorted_branches), std::end(sorted_branches), [&] { return n++; });
    std::sort(std::begin(sorted_branches), std::end(sorted_branches),
              [&](int i1, int i2) { return rtr_leaves[i1].num_leaves < rtr_leaves[i2].num_leaves; });
    std::vector<int> index_sorted_branches(N, 0);
    for (auto it = rtr_leaves.begin(); it != std::prev(rtr_leaves.end(), 1); it++) index_sorted_branches[it->num_leaves]++;
    int cum = 0;
    for (auto &v : index_sorted_branches) {
      v += cum;
      cum = v;
    }

    std::vector<int> unpaired;
    for (int i = 0; i < N; i++) {  // leaves (:1502-1550)
      if (eq[i] != -1) continue;
      const int parent = tree.parent[i], ref_parent = ref_tree.parent[i];
      const int sibling = tree.child_left[parent] == i ? tree.child_right[parent] : tree.child_left[parent];
      if (sibling < N) {
        if (sibling == ref_tree.child_right[ref_parent] || sibling == ref_tree.child_left[ref_parent]) {
          eq[i] = i;
          eq_ref[i] = i;
          eq[sibling] = sibling;
          eq_ref[sibling] = sibling;
        }
      } else {
        if (cor.pearson(tr_leaves[parent], rtr_leaves[ref_parent]) >= threshold_brancheq) {
          eq[i] = i;
          eq_ref[i] = i;
        }
      }
    }
    for (int i = N; i < N_total - 1; i++) {  // internal branches (:1553-1583)
      if (cor.pearson(tr_leaves[i], rtr_leaves[i]) >= 0.9999 &&
          cor.pearson(tr_leaves[tree.parent[i]], rtr_leaves[ref_tree.parent[i]]) >= 0.9999) {
        eq[i] = i;
        eq_ref[i] = i;
      }
      if (eq[i] == -1) {
        const int nl = tr_leaves[i].num_leaves;
        for (auto it = std::next(sorted_branches.begin(), index_sorted_branches[nl - 1]);
             it != std::next(sorted_branches.begin(), index_sorted_branches[nl]); it++) {
          if (cor.pearson(tr_leaves[i], rtr_leaves[*it]) >= 0.9999 &&
              cor.pearson(tr_leaves[tree.parent[i]], rtr_leaves[ref_tree.parent[*it]]) >= 0.9999) {
            eq[i] = *it;
            eq_ref[*it] = i;
            break;
          }
        }
      }
      if (eq[i] == -1) unpaired.push_back(i);
    }
    std::vector<EquivalentNode> possible_pairs;  // approximate matches (:1586-1601)
    for (int u : unpaired) {
      const int nl = tr_leaves[u].num_leaves - 1;
      for (int k : potential_branches[nl]) {
        for (auto it = std::next(sorted_branches.begin(), index_sorted_branches[k - 1]);
             it != std::next(sorted_branches.begin(), index_sorted_branches[k]); it++) {
          if (eq_ref[*it] == -1) {
            float score = cor.pearson(tr_leaves[u], rtr_leaves[*it]);
            if (score >= threshold_brancheq &&
                cor.pearson(tr_leaves[tree.parent[u]], rtr_leaves[ref_tree.parent[*it]]) >= threshold_brancheq) {
              possible_pairs.push_back(EquivalentNode{u, *it, score});
            }
          }
        }
      }
    }
    std::sort(std::begin(possible_pairs), std::end(possible_pairs), std::greater<EquivalentNode>());
    for (const EquivalentNode &e : possible_pairs) {
      if (eq[e.node1] == -1 && eq_ref[e.node2] == -1) {
        eq[e.node1] = e.node2;
        eq_ref[e.node2] = e.node1;
      }
    }
  }
};

}  // namespace

extern "C" int rl_stage_find_equivalent_branches(const char *out_dir, int chunk_index) {
  if (!out_dir) return RL_EINVAL;
  const std::string out(out_dir);
  int N = 0, L = 0, W = 0;
  {
    const std::string pf = out + "/parameters_c" + std::to_string(chunk_index) + ".bin";
    FILE *fp = fopen(pf.c_str(), "rb");
    if (!fp) {
      set_error("cannot open %s", pf.c_str());
      return RL_EIO;
    }
    const bool ok = fread(&N, 4, 1, fp) == 1 && fread(&L, 4, 1, fp) == 1 && fread(&W, 4, 1, fp) == 1;
    fclose(fp);
    if (!ok || N < 2 || W < 2) {
      set_error("%s: malformed", pf.c_str());
      return RL_EIO;
    }
    W--;  // the file stores the number of window boundaries
  }
  // the reference names the files <out>/chunk_<c>/<basename(out)>_<w>.anc (out is a bare name in cwd there)
  const std::string base = out.substr(out.find_last_of('/') == std::string::npos ? 0 : out.find_last_of('/') + 1);
  const std::string dir = out + "/chunk_" + std::to_string(chunk_index) + "/";
  std::vector<AncFile> ancs(W);
  for (int w = 0; w < W; w++) {
    int rc = read_anc(dir + base + "_" + std::to_string(w) + ".anc", ancs[w]);
    if (rc) return rc;
    if ((int)ancs[w].N != N) {
      set_error("%s_%d.anc holds %u haplotypes, the chunk %d", base.c_str(), w, ancs[w].N, N);
      return RL_EIO;
    }
  }
  // the trees of the chunk as one sequence
  std::vector<AncTree *> seq;
  for (auto &a : ancs)
    for (auto &t : a.trees) seq.push_back(&t);
  const size_t M = seq.size();
  const BranchMatcher bm(N);
  std::vector<std::vector<int>> eq(M);  // eq[m]: branches of tree m -> branches of tree m-1
  {
    const int T = std::min<int>(host_threads(), (int)std::max<size_t>(1, M));
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        for (size_t m = 1 + t; m < M; m += T) bm.associate(*seq[m - 1], *seq[m], eq[m]);
      });
    for (auto &x : th) x.join();
  }
  // AssociateTrees (src/anc_builder.cpp:658-800): forward ...
  const int nodes = 2 * N - 1;
  for (size_t m = 1; m < M; m++) {
    AncTree &cur = *seq[m];
    const AncTree &prev = *seq[m - 1];
    for (int i = 0; i < nodes; i++) {
      const int e = eq[m][i];
      if (e != -1) {
        cur.num_events[i] += prev.num_events[e];
        cur.snp_begin[i] = prev.snp_begin[e];
      }
    }
  }
  // ... and back
  for (size_t m = M - 1; m >= 1; m--) {
    const AncTree &next = *seq[m];
    AncTree &cur = *seq[m - 1];
    for (int i = 0; i < nodes; i++) {
      const int e = eq[m][i];
      if (e != -1) {
        cur.num_events[e] = next.num_events[i];
        cur.snp_end[e] = next.snp_end[i];
      }
    }
  }
  for (int w = 0; w < W; w++) {
    int rc = write_anc(dir + base + "_" + std::to_string(w) + ".anc", ancs[w]);
    if (rc) return rc;
  }
  return RL_OK;
}
