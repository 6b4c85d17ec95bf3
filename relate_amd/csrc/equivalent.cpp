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
// memory, and the pairs of neighbouring trees are matched on all host threads.
// Leaf sets: the reference intersects two sorted label lists per correlation;
// here the leaves of one tree of a pair are numbered depth-first, a clade of it
// is an interval, and an intersection is two binary searches (below) -- the
// stage takes half the reference's time on one thread
// (tools/check_feb_against_reference.py: N = 1000, 128 windows, 4.4 s -> 2.1 s,
// byte-identical .anc files).  The two std::sort calls of the matching are made
// on the same data with the same comparisons as the reference's, so that their
// (unstable) orders of equal keys agree when built against the same libstdc++.
#include <tgmath.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "minmatch.h"

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
  // a tree = `int pos` + nodes records of 24 bytes (parent, branch_length, num_events, SNP_begin, SNP_end, written
  // field by field: no padding): one read per tree -- field by field the C3 chunk's 22.9 GB were 4.8e9 fread calls
  std::vector<unsigned char> rec((size_t)nodes * 24);
  for (unsigned t = 0; ok && t < T; t++) {
    AncTree &tr = a.trees[t];
    tr.parent.resize(nodes);
    tr.snp_begin.resize(nodes);
    tr.snp_end.resize(nodes);
    tr.branch_length.resize(nodes);
    tr.num_events.resize(nodes);
    tr.child_left.assign(nodes, -1);
    tr.child_right.assign(nodes, -1);
    ok = fread(&tr.pos, 4, 1, fp) == 1 && fread(rec.data(), 24, (size_t)nodes, fp) == (size_t)nodes;
    for (int i = 0; ok && i < nodes; i++) {
      const unsigned char *q = rec.data() + (size_t)i * 24;
      memcpy(&tr.parent[i], q, 4);
      memcpy(&tr.branch_length[i], q + 4, 8);
      memcpy(&tr.num_events[i], q + 12, 4);
      memcpy(&tr.snp_begin[i], q + 16, 4);
      memcpy(&tr.snp_end[i], q + 20, 4);
      const int p = tr.parent[i];
      if (p != -1) {
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
  std::vector<unsigned char> rec((size_t)nodes * 24);
  for (const AncTree &tr : a.trees) {
    fwrite(&tr.pos, 4, 1, fp);
    for (int i = 0; i < nodes; i++) {
      unsigned char *q = rec.data() + (size_t)i * 24;
      memcpy(q, &tr.parent[i], 4);
      const double bl = tr.branch_length.empty() ? 0.0 : tr.branch_length[i];  // (empty: all zero, BuildTopology's trees)
      memcpy(q + 4, &bl, 8);
      memcpy(q + 12, &tr.num_events[i], 4);
      memcpy(q + 16, &tr.snp_begin[i], 4);
      memcpy(q + 20, &tr.snp_end[i], 4);
    }
    fwrite(rec.data(), 24, (size_t)nodes, fp);
  }
  const bool bad = ferror(fp) != 0;
  if (fclose(fp) != 0 || bad) {  // (a full disc must not pass for a tree file)
    set_error("writing %s failed", fn.c_str());
    return RL_EIO;
  }
  return RL_OK;
}

// ---- leaf sets of two neighbouring trees
// Everything BranchAssociation (src/anc_builder.cpp:1454-1613) asks of the trees is |A n B| for a node A of one tree
// and a node B of the other (Correlation::Pearson, src/anc.cpp:823-859; the reference intersects two sorted label
// lists per question, Tree::FindAllLeaves :450-520).  Here the leaves of the REFERENCE tree are numbered in the order
// a depth-first walk meets them, so the leaves below any of its nodes are one interval of positions; the other tree
// keeps, per node, the positions of its leaves sorted -- and an intersection is two binary searches.

// nodes of a tree, every node after its children
// (every container of a pair's association is filled into storage the calling thread keeps from pair to pair: with
//  fresh vectors -- several of them past malloc's mmap threshold -- 256 threads spent their time in the kernel's
//  address-space lock: 95,000 pairs of N = 5000 trees took 196 s of wall-clock for 3157 s of CPU)
void children_first(const AncTree &t, std::vector<int> &order, std::vector<int> &todo) {
  const int nodes = (int)t.parent.size(), N = (nodes + 1) / 2;
  int root = nodes - 1;
  if (t.parent[root] != -1)  // (as Tree::FindAllLeaves looks for it, :452-461)
    for (int i = N; i < nodes; i++)
      if (t.parent[i] == -1) {
        root = i;
        break;
      }
  order.clear();
  order.reserve(nodes);
  todo.assign(1, root);
  while (!todo.empty()) {  // parents first, left subtree last to leave the stack ...
    const int v = todo.back();
    todo.pop_back();
    order.push_back(v);
    if (t.child_left[v] != -1) {
      todo.push_back(t.child_left[v]);
      todo.push_back(t.child_right[v]);
    }
  }
  std::reverse(order.begin(), order.end());  // ... reversed: children first
}

struct IntervalTree {  // the reference tree of a pair
  std::vector<int> lo, size;   // per node: its leaves are the positions [lo, lo + size)
  std::vector<int> position;   // per leaf
  std::vector<int> leaf_at;    // per position
  std::vector<int> order, todo;
  void build(const AncTree &t) {
    const int nodes = (int)t.parent.size();
    lo.assign(nodes, 0);
    size.assign(nodes, 0);
    position.assign((nodes + 1) / 2, 0);
    leaf_at.assign((nodes + 1) / 2, 0);
    children_first(t, order, todo);
    for (int v : order) size[v] = t.child_left[v] == -1 ? 1 : size[t.child_left[v]] + size[t.child_right[v]];
    for (auto it = order.rbegin(); it != order.rend(); ++it) {  // parents first: hand the interval down
      const int v = *it;
      if (t.child_left[v] == -1) {
        position[v] = lo[v];
        leaf_at[lo[v]] = v;
      } else {
        lo[t.child_left[v]] = lo[v];
        lo[t.child_right[v]] = lo[v] + size[t.child_left[v]];
      }
    }
  }
};

struct PositionSets {  // the other tree of the pair: per node the sorted positions (in the reference tree) of its leaves
  std::vector<size_t> off;
  std::vector<int> pos;
  std::vector<int> order, todo, n_below;
  void build(const AncTree &t, const IntervalTree &ref) {
    const int nodes = (int)t.parent.size();
    children_first(t, order, todo);
    n_below.assign(nodes, 0);
    for (int v : order) n_below[v] = t.child_left[v] == -1 ? 1 : n_below[t.child_left[v]] + n_below[t.child_right[v]];
    off.assign((size_t)nodes + 1, 0);
    for (int v = 0; v < nodes; v++) off[v + 1] = off[v] + (size_t)n_below[v];
    pos.resize(off[nodes]);
    for (int v : order) {
      int *out = pos.data() + off[v];
      if (t.child_left[v] == -1) {
        *out = ref.position[v];
      } else {
        const int a = t.child_left[v], b = t.child_right[v];
        std::merge(pos.data() + off[a], pos.data() + off[a + 1], pos.data() + off[b], pos.data() + off[b + 1], out);
      }
    }
  }
  int size(int v) const { return (int)(off[v + 1] - off[v]); }
  int shared(int v, int lo, int n) const {  // leaves of v with a position in [lo, lo + n)
    const int *b = pos.data() + off[v], *e = pos.data() + off[v + 1];
    return (int)(std::lower_bound(b, e, lo + n) - std::lower_bound(b, e, lo));
  }
};

// Correlation::Pearson (src/anc.cpp:823-859) of two leaf sets given their sizes and what they share: float arithmetic
// in the reference's order of operations (the thresholds 0.9999 / 0.95 are compared against these floats)
float leaf_set_correlation(int n1, int n2, int both, int N) {
  if (n1 == N || n2 == N) return n1 == n2 ? 1.0f : 0.0f;
  const float Nf = (float)N, prod = (float)both;
  if (prod == n1 && prod == n2) return 1.0f;
  float r = prod - n1 * (((float)n2) / Nf);
  if (r <= 0.0) return 0.0f;
  r /= sqrt(((((float)n1) / Nf) * (Nf - n1)) * ((((float)n2) / Nf) * (Nf - n2)));
  return r;
}

struct ScoredPair {
  int node, ref_node;
  float corr;
  bool operator>(const ScoredPair &o) const { return corr > o.corr; }
};

struct BranchMatcher {
  int N, nodes;
  float close_enough = 0.95;  // threshold_brancheq
  // (PreCalcPotentialBranches, src/anc_builder.cpp:1432-1452, lists for every clade size i the sizes j that can
  //  correlate >= close_enough with it, j ascending: `near` below is its membership test)

  // can clades of i and j leaves correlate >= close_enough (PreCalcPotentialBranches' test, :1440-1448)
  bool near(int i, int j) const {
    if (i == j) return true;
    if (i > j) std::swap(i, j);
    const float bound = 1 / (close_enough * close_enough), Nf = N;
    return bound >= j / (Nf - j) * ((Nf - i) / i);
  }
  // (the table itself -- row i - 1: the sizes near i, ascending -- is what the reference loops over; here only its
  //  membership test is used, below)
  explicit BranchMatcher(int n) : N(n), nodes(2 * n - 1) {}

  // BranchAssociation (src/anc_builder.cpp:1454-1613): match[i] = the branch of ref_tree equivalent to branch i of
  // tree, or -1.  Three rounds: leaves; internal branches with a perfect counterpart (the same label first, then the
  // reference's branches of the same clade size); the rest by descending correlation among counterparts of a
  // compatible size, greedily.
  void associate(const AncTree &ref_tree, const AncTree &tree, std::vector<int> &match) const {
    struct Workspace {
      IntervalTree ref;
      PositionSets sets;
      std::vector<int> taken, by_size, open, rank, chain;
      std::vector<ScoredPair> candidates;
    };
    static thread_local Workspace ws;
    match.assign(nodes, -1);
    std::vector<int> &taken = ws.taken;  // per reference branch: who has it
    taken.assign(nodes, -1);
    IntervalTree &ref = ws.ref;
    ref.build(ref_tree);
    PositionSets &sets = ws.sets;
    sets.build(tree, ref);
    auto corr = [&](int v, int rv) {
      return leaf_set_correlation(sets.size(v), ref.size[rv], sets.shared(v, ref.lo[rv], ref.size[rv]), N);
    };
    auto pair_up = [&](int v, int rv) {
      match[v] = rv;
      taken[rv] = v;
    };

    // the reference's branches by clade size.  (std::sort, unstable, on the identity permutation with this
    // comparison: the order INSIDE a size class is whatever that call leaves, and the greedy round below depends on
    // it through its own unstable sort -- the same two calls on the same data as anc_builder.cpp:1477-1480, :1603.)
    std::vector<int> &by_size = ws.by_size;
    by_size.resize(nodes);
    for (int v = 0; v < nodes; v++) by_size[v] = v;
    std::sort(by_size.begin(), by_size.end(), [&](int a, int b) { return ref.size[a] < ref.size[b]; });
    for (int leaf = 0; leaf < N; leaf++) {  // :1502-1550
      if (match[leaf] != -1) continue;
      const int up = tree.parent[leaf], ref_up = ref_tree.parent[leaf];
      const int sibling = tree.child_left[up] == leaf ? tree.child_right[up] : tree.child_left[up];
      if (sibling < N) {  // a cherry: equivalent if it is a cherry of the reference tree too
        if (sibling == ref_tree.child_right[ref_up] || sibling == ref_tree.child_left[ref_up]) {
          pair_up(leaf, leaf);
          pair_up(sibling, sibling);
        }
      } else if (corr(up, ref_up) >= close_enough) {
        pair_up(leaf, leaf);
      }
    }
    // Where the reference walks whole size classes of the other tree's branches (:1553-1601), only ONE chain of them can
    // pass its tests: a Pearson correlation of two leaf sets above 1/sqrt(2) needs more than half of each set in the
    // other (with a = n1/N, b = n2/N, c = shared/N: c <= a/2 gives (c - ab) / sqrt(a(1-a)b(1-b)) <= sqrt((1-a)/(2-a))
    // < 0.708), the reference tree's clades are intervals of its depth-first leaf order, and an interval that holds
    // more than half of a sorted set holds its median -- so every branch that can score >= 0.95 (let alone 0.9999)
    // against clade v is an ancestor of the reference leaf at the median position of v's leaves.  The chain is
    // walked instead, and put in the order the reference's loops would have met its members in (ascending size,
    // inside a size class the order std::sort left): the same candidates in the same order, a few dozen
    // correlations per open branch instead of hundreds (C3: 3100 s of CPU for the chunk's 95,000 pairs of trees).
    std::vector<int> &rank = ws.rank;  // position of a reference branch in by_size
    rank.resize(nodes);
    for (int x = 0; x < nodes; x++) rank[by_size[x]] = x;
    auto median_leaf = [&](int v) { return ref.leaf_at[sets.pos[sets.off[v] + (size_t)(sets.size(v) / 2)]]; };
    std::vector<int> &open = ws.open;
    open.clear();
    for (int v = N; v < nodes - 1; v++) {  // :1553-1583
      auto perfect = [&](int rv) { return corr(v, rv) >= 0.9999 && corr(tree.parent[v], ref_tree.parent[rv]) >= 0.9999; };
      if (perfect(v)) pair_up(v, v);
      if (match[v] == -1) {
        // (the first branch of v's size class that is perfect: at most one member of the class lies on the chain)
        const int n1 = sets.size(v);
        for (int b = median_leaf(v); b != -1 && ref.size[b] <= n1; b = ref_tree.parent[b])
          if (ref.size[b] == n1 && b != nodes - 1 && n1 < N && perfect(b)) {
            pair_up(v, b);
            break;
          }
      }
      if (match[v] == -1) open.push_back(v);
    }
    std::vector<ScoredPair> &candidates = ws.candidates;  // :1586-1601
    candidates.clear();
    std::vector<int> &chain = ws.chain;
    for (int v : open) {
      const int n1 = sets.size(v);
      chain.clear();
      for (int b = median_leaf(v); b != -1; b = ref_tree.parent[b]) {
        const int s2 = ref.size[b];
        if (s2 >= N) break;
        if (b == nodes - 1) continue;  // (skipped, not the end of the chain: a root need not be labelled nodes - 1)
        if (near(n1, s2)) chain.push_back(b);
      }
      std::sort(chain.begin(), chain.end(), [&](int a, int b) { return rank[a] < rank[b]; });
      for (int b : chain) {
        if (taken[b] != -1) continue;
        const float score = corr(v, b);
        if (score >= close_enough && corr(tree.parent[v], ref_tree.parent[b]) >= close_enough)
          candidates.push_back(ScoredPair{v, b, score});
      }
    }
    std::sort(candidates.begin(), candidates.end(), std::greater<ScoredPair>());
    for (const ScoredPair &c : candidates)
      if (match[c.node] == -1 && taken[c.ref_node] == -1) pair_up(c.node, c.ref_node);
  }
};

// AssociateTrees (src/anc_builder.cpp:658-800): num_events / SNP_begin carried forward along equivalent branches,
// num_events / SNP_end back; eq[m]: branches of tree m -> branches of tree m-1 (eq[0] unused)
void propagate(const std::vector<AncTree *> &seq, const std::vector<std::vector<int>> &eq, int N) {
  const size_t M = seq.size();
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
}

}  // namespace

// ---- the same stage fused behind BuildTopology (rl_stage_opts.find_equivalent_branches): the sections' trees never
// leave memory between the two stages.  A section that is built hands its trees over (feb_job_add_section); a few
// pool threads associate the neighbouring trees INSIDE finished sections while the stage's other sections are still
// building (the host's cores are idle then: a section thread mostly waits for its tree on the device); when the last
// section is in, feb_job_finish associates the pairs ACROSS section boundaries, carries events and SNP ranges
// forward and back over the whole chunk (the only sequential part: ~1e9 simple operations at C3) and writes every
// .anc file ONCE, as FindEquivalentBranches would leave it -- instead of BuildTopology writing 22.9 GB, this stage
// reading them back, and writing them again.
namespace rl {

struct FebJob {
  int N = 0, W = 0;
  std::vector<AncFile> ancs;
  std::vector<std::vector<std::vector<int>>> eq;  // eq[w][t]: tree t of section w -> the tree before it (t = 0: the
                                                  // last tree of section w - 1)
  std::vector<char> have;
  BranchMatcher bm;
  std::mutex m;
  std::condition_variable cv;
  std::vector<std::pair<int, int>> tasks;  // (section, tree >= 1), taken from the back
  int running = 0;
  bool closing = false;
  std::vector<std::thread> pool;
  double cpu_s = 0.0;
  explicit FebJob(int n, int w, int threads) : N(n), W(w), ancs(w), eq(w), have(w, 0), bm(n) {
    for (int t = 0; t < threads; t++) pool.emplace_back([this]() { work(); });
  }
  void work() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [&] { return closing || !tasks.empty(); });
      if (tasks.empty()) {
        if (closing) return;
        continue;
      }
      const std::pair<int, int> t = tasks.back();
      tasks.pop_back();
      running++;
      lk.unlock();
      bm.associate(ancs[t.first].trees[t.second - 1], ancs[t.first].trees[t.second], eq[t.first][t.second]);
      lk.lock();
      running--;
      if (tasks.empty() && running == 0) cv.notify_all();
    }
  }
  ~FebJob() {
    {
      std::lock_guard<std::mutex> lk(m);
      closing = true;
      tasks.clear();
    }
    cv.notify_all();
    for (auto &t : pool) t.join();
  }
};

FebJob *feb_job_create(int N, int W, int threads) { return new FebJob(N, W, std::max(1, threads)); }
void feb_job_destroy(FebJob *j) { delete j; }

int feb_job_add_section(FebJob *j, int w, const std::vector<HostTree> &trees) {
  if (!j || w < 0 || w >= j->W || trees.empty() || j->have[w]) {
    set_error("find equivalent branches (fused): section %d out of range, empty or given twice", w);
    return RL_EINVAL;
  }
  const int nodes = 2 * j->N - 1;
  AncFile &a = j->ancs[w];
  a.has_ages = false;
  a.N = (unsigned)j->N;
  a.trees.assign(trees.size(), AncTree());
  for (size_t t = 0; t < trees.size(); t++) {  // what read_anc would make of the file rl_treeseq_write writes
    const HostTree &h = trees[t];
    AncTree &tr = a.trees[t];
    tr.pos = h.pos;
    tr.parent = h.parent;
    tr.snp_begin = h.snp_begin;
    tr.snp_end = h.snp_end;
    tr.num_events = h.num_events;
    tr.child_left.assign(nodes, -1);
    tr.child_right.assign(nodes, -1);
    for (int i = 0; i < nodes; i++) {  // children as Tree::ReadTreeBin assigns them: in node order
      const int p = tr.parent[i];
      if (p == -1) continue;
      if (p < 0 || p >= nodes) {
        set_error("find equivalent branches (fused): section %d, tree %zu: parent of node %d out of range", w, t, i);
        return RL_EINVAL;
      }
      if (tr.child_left[p] == -1) tr.child_left[p] = i;
      else tr.child_right[p] = i;
    }
  }
  j->eq[w].assign(trees.size(), std::vector<int>());
  {
    std::lock_guard<std::mutex> lk(j->m);
    j->have[w] = 1;
    for (int t = (int)trees.size() - 1; t >= 1; t--) j->tasks.emplace_back(w, t);
  }
  j->cv.notify_all();
  return RL_OK;
}

int feb_job_finish(FebJob *j, const std::string &dir_base) {
  const bool timing = getenv("RELATE_AMD_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  auto lap = [&](const char *what) {
    const double t = now();
    if (timing) fprintf(stderr, "[find equivalent branches, fused] %-45s %.3f s\n", what, t - t_mark);
    t_mark = t;
  };
  for (int w = 0; w < j->W; w++)
    if (!j->have[w]) {
      set_error("find equivalent branches (fused): section %d was not built", w);
      return RL_ESTATE;
    }
  {  // what the pool has not got to yet: with every host thread now
    std::vector<std::pair<int, int>> rest;
    {
      std::lock_guard<std::mutex> lk(j->m);
      rest.swap(j->tasks);
    }
    for (int w = 1; w < j->W; w++) rest.emplace_back(w, 0);  // the pairs across section boundaries
    std::atomic<size_t> next(0);
    const int T = std::max(1, std::min<int>(host_threads(), (int)rest.size()));
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&]() {
        for (size_t k = next.fetch_add(1); k < rest.size(); k = next.fetch_add(1)) {
          const int w = rest[k].first, tt = rest[k].second;
          const AncTree &prev = tt > 0 ? j->ancs[w].trees[tt - 1] : j->ancs[w - 1].trees.back();
          j->bm.associate(prev, j->ancs[w].trees[tt], j->eq[w][tt]);
        }
      });
    for (auto &x : th) x.join();
    std::unique_lock<std::mutex> lk(j->m);  // (pairs a pool thread had taken before the swap)
    j->cv.wait(lk, [&] { return j->running == 0; });
  }
  lap("associations left after the last section");
  std::vector<AncTree *> seq;
  std::vector<std::vector<int>> eq;
  for (int w = 0; w < j->W; w++)
    for (size_t t = 0; t < j->ancs[w].trees.size(); t++) {
      seq.push_back(&j->ancs[w].trees[t]);
      eq.emplace_back();
      eq.back().swap(j->eq[w][t]);
    }
  propagate(seq, eq, j->N);
  lap("events and SNP ranges carried forward and back");
  const int W = j->W, T = std::max(1, std::min({host_threads(), W, 32}));
  std::vector<int> rcs(T, RL_OK);
  std::vector<std::string> msgs(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++)
    th.emplace_back([&, t]() {
      for (int w = t; w < W && rcs[t] == RL_OK; w += T)
        if ((rcs[t] = write_anc(dir_base + "_" + std::to_string(w) + ".anc", j->ancs[w])) != RL_OK) msgs[t] = rl_last_error();
    });
  for (auto &x : th) x.join();
  for (int t = 0; t < T; t++)
    if (rcs[t] != RL_OK) {
      set_error("%s", msgs[t].c_str());
      return rcs[t];
    }
  lap(".anc files written (once)");
  return RL_OK;
}

}  // namespace rl

// Test hook: the fused job fed from FILES -- every section's .anc read back into the stage's own tree type and handed
// to the job in the given order (as sections finish in any order), then the job's tail.  Same bytes as the stage below
// (tests/test_equivalent_cpu.py: the pool, the pairs across section boundaries, the order of arrival; no GPU).
extern "C" int rl_debug_feb_fused_from_files(const char *out_dir, int chunk_index, const int *order, int n_order,
                                             int pool_threads) {
  if (!out_dir || !order) return RL_EINVAL;
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
    if (!ok || N < 2 || W < 2) return RL_EIO;
    W--;
  }
  if (n_order != W) {
    set_error("rl_debug_feb_fused_from_files: %d sections, an order of %d", W, n_order);
    return RL_EINVAL;
  }
  const std::string base = out.substr(out.find_last_of('/') == std::string::npos ? 0 : out.find_last_of('/') + 1);
  const std::string dir = out + "/chunk_" + std::to_string(chunk_index) + "/";
  FebJob *job = feb_job_create(N, W, pool_threads);
  int rc = RL_OK;
  for (int k = 0; k < W && !rc; k++) {
    const int w = order[k];
    AncFile a;
    if ((rc = read_anc(dir + base + "_" + std::to_string(w) + ".anc", a))) break;
    std::vector<HostTree> trees(a.trees.size());
    for (size_t t = 0; t < a.trees.size(); t++) {
      trees[t].reset(N);
      trees[t].pos = a.trees[t].pos;
      trees[t].parent = a.trees[t].parent;
      trees[t].num_events = a.trees[t].num_events;
      trees[t].snp_begin = a.trees[t].snp_begin;
      trees[t].snp_end = a.trees[t].snp_end;
    }
    rc = feb_job_add_section(job, w, trees);
  }
  if (!rc) rc = feb_job_finish(job, dir + base);
  feb_job_destroy(job);
  return rc;
}

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
  const bool timing = getenv("RELATE_AMD_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  auto lap = [&](const char *what) {
    const double t = now();
    if (timing) fprintf(stderr, "[find equivalent branches] %-52s %.3f s\n", what, t - t_mark);
    t_mark = t;
  };
  std::vector<AncFile> ancs(W);
  // the windows' files are independent: read (and, below, written) on a few threads each taking files round-robin
  auto over_files = [&](const std::function<int(int)> &one) -> int {
    const int T = std::max(1, std::min({host_threads(), W, 32}));
    std::vector<int> rcs(T, RL_OK);
    std::vector<std::string> msgs(T);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        for (int w = t; w < W && rcs[t] == RL_OK; w += T)
          if ((rcs[t] = one(w)) != RL_OK) msgs[t] = rl_last_error();  // (the error text is per thread)
      });
    for (auto &x : th) x.join();
    for (int t = 0; t < T; t++)
      if (rcs[t] != RL_OK) {
        set_error("%s", msgs[t].c_str());
        return rcs[t];
      }
    return RL_OK;
  };
  int frc = over_files([&](int w) -> int {
    int rc = read_anc(dir + base + "_" + std::to_string(w) + ".anc", ancs[w]);
    if (rc) return rc;
    if ((int)ancs[w].N != N) {
      set_error("%s_%d.anc holds %u haplotypes, the chunk %d", base.c_str(), w, ancs[w].N, N);
      return RL_EIO;
    }
    return RL_OK;
  });
  if (frc) return frc;
  lap("read the sections' .anc files");
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
  lap("branches of neighbouring trees associated");
  propagate(seq, eq, N);
  lap("events and SNP ranges carried forward and back");
  frc = over_files([&](int w) -> int { return write_anc(dir + base + "_" + std::to_string(w) + ".anc", ancs[w]); });
  if (frc) return frc;
  lap("files rewritten");
  // chunk_<c>.bits (this library's MakeChunks under RELATE_AMD_CHUNK_BITS=1) has been read by the last stage that
  // wants it: the reference's later stages do not know the file and end on an rmdir of the directory
  // (Finalize.cpp:290, Clean.cpp:120), which a leftover would fail
  (void)remove((out + "/chunk_" + std::to_string(chunk_index) + ".bits").c_str());
  return RL_OK;
}
