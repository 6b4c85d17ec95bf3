// treeseq.cpp -- the tree-sequence loop of one section and the BuildTopology stage.
//
// Host restatement of AncesTreeBuilder::BuildTopology (src/anc_builder.cpp:398-656),
// MapMutation / ForceMapMutation / PropagateMutation* (:1064-1413), the .anc
// writer AncesTree::DumpBin (src/anc.cpp:1104-1167) and the .mut writer
// Mutations::DumpShortFormat (src/mutations.cpp:548-581), for the
// configuration the stage driver uses (pipeline/BuildTopology.cpp:14-167):
// ancestral_state = true, sample_ages empty.
//
// Distance matrices come from a provider (callbacks): the stage wires the GPU
// rl_window in; nothing here computes painting on the CPU.
#include <sched.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

#include "common.h"
#include "minmatch.h"

namespace rl {

struct SnpInfo {
  int tree = 0;
  std::vector<int> branch;
  bool flipped = false;
};

}  // namespace rl

struct rl_treeseq {
  int N = 0, L = 0;
  double theta = 0.001;
  std::vector<uint32_t> bits_own;  // panel (a copy: rl_treeseq_create), or
  const uint32_t *bits = nullptr;  // ... the caller's, which outlives the object (the stage: its context's panel)
  int row_words = 0;
  std::vector<double> rpos;
  std::vector<int> bp, state;
  int thr = 0;
  std::vector<rl::HostTree> trees;
  std::vector<rl::SnpInfo> info;  // indexed by snp - start
  int start = 0, end = 0;
  std::vector<char> member;  // carriers of the current SNP
  int num_carriers = 0;
  int build_device = -1;  // >= 0: trees are built on that GPU (minmatch_gpu.hip), the host builder as fallback
  rl_matrix_dev_fn matrix_dev = nullptr;  // with it the distance matrices never leave the device
  rl_matrix_dev_ex_fn matrix_dev_ex = nullptr;  // ... and arrive with the carrier penalty applied and their row minima
  long long gpu_trees = 0, host_trees = 0;
  std::vector<double> sample_ages;  // N values (--sample_ages) or empty
  // the device builder's buffers (16 N^2 B of woven matrix) outlive a section: the next one of this object reuses them
  rl::DeviceMinMatch *dev_builder = nullptr;
  ~rl_treeseq() { delete dev_builder; }

  bool derived(int snp, int n) const { return (bits[(size_t)snp * row_words + (n >> 5)] >> (n & 31)) & 1u; }
};

namespace rl {

double device_builder_shared_bytes(int N);  // (minmatch_gpu.hip) HBM of the pools the device builders share

// ---- Where a SNP sits on a tree (anc_builder.cpp:1064-1413: MapMutation, ForceMapMutation and their two
// recursions PropagateMutationGlobal / PropagateMutationLocal).
//
// The reference walks the tree recursively and carries eight counters per call.  Everything it decides at a node
// follows from two numbers -- the carriers and the leaves below it -- and from what its two children decided, so
// here the tree is swept once, bottom-up, over flat arrays:
//   * MinMatch labels a merged node N + (merge index): children always carry smaller labels than their parent, so
//     the label order 0 .. 2N-2 IS a bottom-up order (no stack, no recursion 5000 deep);
//   * for the orders in which the reference *emits* branches (ForceMapMutation's lists) the post-order of the
//     recursion (left subtree, right subtree, node) is made explicit once per call.
// Float semantics kept: every ratio is a float division compared against the double literals 0.3 / 0.7 / 0.03.

// A branch and how many leaves it misplaces; `leaves == INT_MAX`: no admissible branch in the subtree.
struct Fit {
  int leaves = INT_MAX;
  int branch = -1;
};

// the two ratio tests every clade has to pass in either orientation (:1265-1279, :1295-1310)
static inline bool small_share(int part, float whole) { return (float)part / whole < 0.3; }
static inline bool pure(int agreeing, int all) { return all <= 0 || (float)agreeing / (float)all > 0.7; }

// MapMutation's search (:1237-1341): over all branches, the one whose clade is closest to the carrier set -- as it
// stands (`direct`: the leaves below carry the derived allele) and with the alleles flipped (`flipped`).  A clade
// replaces the best of its subtrees only when strictly better than both; of two subtrees the left one keeps ties.
static void best_branches(const rl_treeseq &ts, const HostTree &t, Fit &direct_root, Fit &flipped_root) {
  const int N = ts.N, T = 2 * N - 1;
  const float carriers = (float)ts.num_carriers, others = (float)N - carriers;
  static thread_local std::vector<int> below_c, below_n;  // carriers / non-carriers below each node
  static thread_local std::vector<Fit> direct, flipped;
  below_c.resize(T);
  below_n.resize(T);
  direct.resize(T);
  flipped.resize(T);
  for (int v = 0; v < N; v++) {  // leaves (:1313-1339): one ratio test per orientation, two for the "wrong" allele
    const int c = ts.member[v] == 1 ? 1 : 0, in_c = c, in_n = 1 - c;
    const int out_c = (int)(carriers - (float)in_c), out_n = (int)(others - (float)in_n);
    below_c[v] = in_c;
    below_n[v] = in_n;
    Fit d, f;
    if (c) {
      if (small_share(out_c, carriers)) d = Fit{out_c, v};
      if (small_share(in_c, carriers) && small_share(out_n, others)) f = Fit{out_n + in_c, v};
    } else {
      if (small_share(out_c, carriers) && small_share(in_n, others)) d = Fit{out_c + in_n, v};
      if (small_share(out_n, others)) f = Fit{out_n, v};
    }
    direct[v] = d;
    flipped[v] = f;
  }
  for (int v = N; v < T; v++) {
    const int l = t.child_left[v], r = t.child_right[v];
    const int in_c = below_c[l] + below_c[r], in_n = below_n[l] + below_n[r];
    const int out_c = (int)(carriers - (float)in_c), out_n = (int)(others - (float)in_n);
    below_c[v] = in_c;
    below_n[v] = in_n;
    {  // as it stands: carriers outside and non-carriers inside are the misplaced leaves
      const int wrong = out_c + in_n;
      const bool ok = small_share(out_c, carriers) && small_share(in_n, others) && pure(in_c, in_c + in_n) &&
                      pure(out_n, out_c + out_n);
      if (ok && direct[l].leaves > wrong && direct[r].leaves > wrong)
        direct[v] = Fit{wrong, v};
      else
        direct[v] = direct[l].leaves > direct[r].leaves ? direct[r] : direct[l];
    }
    {  // flipped: carriers inside and non-carriers outside
      const int wrong = in_c + out_n;
      const bool ok = small_share(in_c, carriers) && small_share(out_n, others) && pure(out_c, out_c + out_n) &&
                      pure(in_n, in_c + in_n);
      if (ok && flipped[l].leaves > wrong && flipped[r].leaves > wrong)
        flipped[v] = Fit{wrong, v};
      else
        flipped[v] = flipped[l].leaves > flipped[r].leaves ? flipped[r] : flipped[l];
    }
  }
  direct_root = direct[T - 1];
  flipped_root = flipped[T - 1];
}

// MapMutation (:1064-1139, the version without random flipping): 1 = the SNP maps, 2 = maps with the alleles
// flipped, 3 = no branch within the tolerance (si untouched).  misfit: the misplaced leaves of the best branch.
static int map_mutation(rl_treeseq &ts, HostTree &t, SnpInfo &si, float &misfit, bool count_event) {
  const int N = ts.N, root = 2 * N - 2;
  if (ts.num_carriers == 0 || ts.num_carriers == N) {  // nothing to place; a fixed site sits on the root branch
    misfit = 0.0f;
    si.flipped = false;
    si.branch.clear();
    if (ts.num_carriers == N) {
      si.branch.push_back(root);
      t.num_events[root] += 1.0f;  // (counted whatever the SNP's state, :1075)
    }
    return 1;
  }
  Fit direct, flipped;
  best_branches(ts, t, direct, flipped);
  const bool as_is = direct.leaves <= flipped.leaves;  // (ties go to the unflipped reading)
  const Fit &best = as_is ? direct : flipped;
  misfit = (float)best.leaves;
  if (best.leaves > ts.thr) return 3;
  si.branch.assign(1, best.branch);
  si.flipped = !as_is;
  if (count_event) t.num_events[best.branch] += 1.0f;
  return as_is ? 1 : 2;
}

// ForceMapMutation (:1143-1204) with PropagateMutationLocal (:1344-1413): a SNP that fits no single branch goes
// onto the maximal clades that are (almost, < 3 % of their leaves) pure in carriers -- or, flipped, in non-carriers
// -- whichever reading needs fewer branches.  A clade stays "open" while it is pure and both children are open; its
// branch is the node itself if both children hold leaves of the kind, else the one child's branch.  When a clade
// closes, its children's open branches are emitted (left, then right) -- in the order the recursion reaches the
// closing nodes, i.e. post-order, which the sweep below follows through an explicit list.
static void force_map_mutation(rl_treeseq &ts, const HostTree &t, SnpInfo &si) {
  const int N = ts.N, T = 2 * N - 1;
  if (ts.num_carriers == 0 || ts.num_carriers == N) return;
  static thread_local std::vector<int> post, stack, kind_count[2], open_branch[2];
  post.clear();
  stack.assign(1, T - 1);
  while (!stack.empty()) {  // (root, right, left) reversed = (left, right, root)
    const int v = stack.back();
    stack.pop_back();
    post.push_back(v);
    if (t.child_left[v] >= 0) {
      stack.push_back(t.child_left[v]);
      stack.push_back(t.child_right[v]);
    }
  }
  std::reverse(post.begin(), post.end());
  std::vector<int> emitted[2];  // [0] carriers' clades, [1] non-carriers' clades (the flipped reading)
  for (int s = 0; s < 2; s++) {
    kind_count[s].resize(T);
    open_branch[s].resize(T);
  }
  for (int v : post) {
    const int l = t.child_left[v], r = t.child_right[v];
    if (l < 0) {
      const int s = ts.member[v] == 1 ? 0 : 1;
      kind_count[s][v] = 1;
      kind_count[1 - s][v] = 0;
      open_branch[s][v] = v;
      open_branch[1 - s][v] = -1;
      continue;
    }
    const int n0 = kind_count[0][l] + kind_count[0][r], n1 = kind_count[1][l] + kind_count[1][r];
    kind_count[0][v] = n0;
    kind_count[1][v] = n1;
    const float leaves = (float)(n0 + n1);
    for (int s = 0; s < 2; s++) {
      const int foreign = s == 0 ? n1 : n0;
      const int bl = open_branch[s][l], br = open_branch[s][r];
      if ((float)foreign / leaves < 0.03 && bl != -1 && br != -1) {
        const bool in_l = kind_count[s][l] > 0, in_r = kind_count[s][r] > 0;
        open_branch[s][v] = in_l && in_r ? v : (in_l ? bl : br);
      } else {
        if (bl != -1) emitted[s].push_back(bl);
        if (br != -1) emitted[s].push_back(br);
        open_branch[s][v] = -1;
      }
    }
  }
  // fewer branches win, the unflipped reading on ties -- unless it found none (:1180-1203)
  const bool as_is = emitted[1].empty() || (!emitted[0].empty() && emitted[0].size() <= emitted[1].size());
  if (!as_is) si.flipped = true;
  si.branch = emitted[as_is ? 0 : 1];
}

// Prior `dist` from the previous tree's clades (anc_builder.cpp:583-606):
// dist[i][j] = val added once per clade that contains i but not j.  Computed
// here per pair from leaf depths and the depth of the pair's lowest common
// ancestor; the repeated float additions of `val` are replayed through a table
// (acc[c] = val added c times, left to right), so every entry is bit-identical
// to the reference's accumulation.
// f(row) for row = 0..N-1 on the tree builder's share of host threads (minmatch.h build_threads)
template <typename F>
static void parallel_rows(int N, F f) {
  const int T = std::min(build_threads(), std::max(1, N / 256));
  if (T <= 1) {
    for (int r = 0; r < N; r++) f(r);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++)
    th.emplace_back([=]() {
      for (int r = t; r < N; r += T) f(r);
    });
  for (auto &x : th) x.join();
}

// dist[a][b] = val added (#clades of the previous tree that contain a but not b) times, accumulated in
// float as the reference does (:588-606): the clades containing a but not b are a's internal ancestors
// strictly below lca(a, b).  Row a is filled ancestor by ancestor: for ancestor v, the leaves below v's other
// child have their lca with a at v; leaves below a node are a contiguous range of a depth-first leaf order.
static void clade_prior(const HostTree &t, float val, MatrixBuf &dist) {
  const int N = t.N, T = 2 * N - 1;
  dist.resize((size_t)N * N);
  std::vector<int> depth(T, 0);  // internal nodes on the path node..root, inclusive of node if internal
  for (int v = T - 1; v >= N; v--) depth[v] = (t.parent[v] >= 0 ? depth[t.parent[v]] : 0) + 1;
  std::vector<float> acc((size_t)N + 1, 0.0f);
  for (int c = 1; c <= N; c++) acc[c] = acc[c - 1] + val;
  // depth-first leaf order; [lo, hi) of every node (labels increase towards the root: T-1 is the root)
  std::vector<int> size(T, 1), lo(T, 0), order(N);
  for (int v = N; v < T; v++) size[v] = size[t.child_left[v]] + size[t.child_right[v]];
  for (int v = T - 1; v >= N; v--) {
    lo[t.child_left[v]] = lo[v];
    lo[t.child_right[v]] = lo[v] + size[t.child_left[v]];
  }
  for (int i = 0; i < N; i++) order[lo[i]] = i;
  parallel_rows(N, [&](int a) {
    float *row = &dist[(size_t)a * N];
    row[a] = 0.0f;
    const int da = depth[t.parent[a]];
    int child = a;
    for (int v = t.parent[a]; v >= 0; child = v, v = t.parent[v]) {
      const int other = t.child_left[v] == child ? t.child_right[v] : t.child_left[v];
      const float x = acc[da - depth[v]];
      for (int p = lo[other]; p < lo[other] + size[other]; p++) row[order[p]] = x;
    }
  });
}

}  // namespace rl

using namespace rl;

extern "C" {

rl_treeseq *rl_treeseq_create(int N, int L, const uint32_t *bits, int row_words, const double *rpos,
                              const int *bp_pos, const int *state, double theta) {
  if (N < 2 || L < 2 || !bits || row_words < (N + 31) / 32 || !rpos || !(theta > 0.0 && theta < 1.0)) {
    set_error("rl_treeseq_create: bad arguments");
    return nullptr;
  }
  rl_treeseq *ts = new rl_treeseq();
  ts->N = N;
  ts->L = L;
  ts->theta = theta;
  ts->row_words = row_words;
  ts->bits_own.assign(bits, bits + (size_t)L * row_words);
  ts->bits = ts->bits_own.data();
  ts->rpos.assign(rpos, rpos + L + 1);
  if (bp_pos) ts->bp.assign(bp_pos, bp_pos + L);
  if (state)
    ts->state.assign(state, state + L);
  else
    ts->state.assign(L, 1);
  ts->thr = (int)(0.03 * N);  // anc_builder.cpp:383
  ts->member.assign(N, 0);
  return ts;
}

void rl_treeseq_destroy(rl_treeseq *ts) { delete ts; }

}  // extern "C"

// as rl_treeseq_create, the panel borrowed instead of copied (also used by shard.cpp)
namespace rl {
rl_treeseq *treeseq_borrowing(int N, int L, const uint32_t *bits, int row_words, const double *rpos,
                                     const int *bp_pos, const int *state, double theta) {
  rl_treeseq *ts = new rl_treeseq();
  ts->N = N;
  ts->L = L;
  ts->theta = theta;
  ts->row_words = row_words;
  ts->bits = bits;
  ts->rpos.assign(rpos, rpos + L + 1);
  ts->bp.assign(bp_pos, bp_pos + L);
  ts->state.assign(state, state + L);
  ts->thr = (int)(0.03 * N);  // anc_builder.cpp:383
  ts->member.assign(N, 0);
  return ts;
}
}  // namespace rl

// How many resident tree-builder workers a stage asks for.  The reference's unit of parallelism is a section job
// (scripts/RelateParallel/RelateParallel.sh:231-257: one process per section range); here a section has at most ONE tree
// in flight, so:
//   * never more workers than open sections (what HBM admits: ~134 at N = 5000, ~70 at N = 10,000, 256 at N = 2000);
//   * whole windows resident (RePaint runs once per window, then the chip is the trees'): a worker per section, up to
//     7/8 of the worker slots -- an eighth of the CUs stays with the sections' own short kernels;
//   * bounded windows (RePaint runs all through the stage, on the CUs the workers leave, and every section waits in
//     its queue): 29/64 of the CUs.  Measured at C3 over three rounds (profiles/r03..r05_c3_*): 3/8 -> 13/32 -> 29/64
//     as the per-tree kernels got lighter; past ~31/64 the stage is bistable (RePaint's lane saturates and the
//     sections fall into a convoy: 143 s and 185 s from the same 124 workers), so the rule stays a whole XCD round
//     (8 workers) below that edge.  profiles/r06_worker_rule.json: the rule against 0.75x / 1.25x / 1.5x of it at
//     N = 2000, 5000 and 10,000;
//   * `per_cu` worker slots per CU (two for the small-N kernel, minmatch_gpu.hip worker_kind): the CU share above is a
//     share of CUs, the slots on them follow.
// (A tree's time in the stage does not depend on the worker count -- 142-150 ms from 116 to 148 workers,
//  profiles/r05_c3_runs.json --; past the share above it is RePaint, on the CUs left, that every section waits for.)
namespace rl {
int stage_worker_goal(int cus, int open_sections, bool bounded_windows, int per_cu) {
  const int share = bounded_windows ? 29 * cus / 64 : cus - cus / 8;
  int goal = std::max(1, std::min(open_sections, share * std::max(1, per_cu)));
  // (whole rounds of the 8 XCDs once the workers are many -- a launch's workgroups go to the XCDs in turn, and RePaint
  //  lasts as long as the XCD with the fewest CUs left: C3's 116 are 112 alive)
  if (goal >= 64) goal -= goal % 8;
  return goal;
}
}  // namespace rl

extern "C" {

int rl_treeseq_build(rl_treeseq *ts, int start, int end, rl_matrix_fn matrix, rl_advance_fn advance, void *user,
                     int flags, int fb) {
  if (!ts || !matrix || start < 0 || end >= ts->L || start > end) {
    set_error("rl_treeseq_build: bad arguments");
    return RL_EINVAL;
  }
  if (fb > 0 && ts->bp.empty()) {
    set_error("rl_treeseq_build: --fb needs bp positions");
    return RL_EINVAL;
  }
  const int N = ts->N;
  const bool consistency = !(flags & 1);
  ts->start = start;
  ts->end = end;
  ts->trees.clear();
  ts->info.assign((size_t)(end - start + 1), SnpInfo());
  // carriers: the reference fills them for snp < section_endpos only, so the
  // last SNP of a window is processed with none (anc_builder.cpp:407-414)
  auto set_carriers = [&](int snp) {
    ts->num_carriers = 0;
    std::fill(ts->member.begin(), ts->member.end(), 0);
    if (snp < end)
      for (int i = 0; i < N; i++)
        if (ts->derived(snp, i)) {
          ts->member[i] = 1;
          ts->num_carriers++;
        }
  };
  MinMatch tb(N, ts->theta);
  // with sample ages (ancient samples) the candidates carry a third key and a clock: its own builder (and its own
  // workers on the device, minmatch_gpu.hip AGES)
  std::unique_ptr<MinMatchAges> tb_ages;
  if ((int)ts->sample_ages.size() == N) tb_ages.reset(new MinMatchAges(N, ts->theta));
  DeviceMinMatch *dev = nullptr;
  if (ts->build_device >= 0 && N <= 10240) {  // (its registers per thread)
    if (!ts->dev_builder) ts->dev_builder = new DeviceMinMatch(N, ts->build_device);
    dev = ts->dev_builder;
  }
  int build_rc = 0;
  auto host_build = [&](float *dm, const float *prior, HostTree &t) {
    ts->host_trees++;
    if (tb_ages) tb_ages->quick_build(dm, prior, ts->sample_ages, t);
    else tb.quick_build(dm, prior, t);
  };
  auto build_tree = [&](float *dm, const float *prior, HostTree &t) {
    if (dev) {
      const int st = tb_ages ? dev->build(*tb_ages, ts->sample_ages, dm, prior, t) : dev->build(tb, dm, prior, t);
      if (st == 0) {
        ts->gpu_trees++;
        return;
      }
      if (st < 0) build_rc = RL_EHIP;
    }
    host_build(dm, prior, t);
  };
  auto build_resident = [&](bool with_prior, HostTree &t) {
    return tb_ages ? dev->build_resident(*tb_ages, ts->sample_ages, with_prior, t) : dev->build_resident(tb, with_prior, t);
  };
  MatrixBuf d((size_t)N * N), dist;
  float min_value = 0.f, min_value_alt = 0.f;
  int rc;

  const bool resident = dev && ts->matrix_dev;
  ts->trees.emplace_back();
  if (resident) {
    float *dd = dev->device_matrix();
    if (!dd) return RL_ENOMEM;
    float *rmin = ts->matrix_dev_ex ? dev->rowmin_device() : nullptr;
    if (rmin) {  // (no penalty for a section's first tree: the row minima alone)
      if ((rc = ts->matrix_dev_ex(user, start, dd, nullptr, 0.0f, rmin))) return rc;
      dev->rowmin_is_ready();
    } else if ((rc = ts->matrix_dev(user, start, dd))) {
      return rc;
    }
    const int st = build_resident(false, ts->trees.back());
    if (st < 0) return RL_EHIP;
    if (st > 0) {  // this tree is the host's (tb untouched): the matrix again, to the host
      if ((rc = matrix(user, start, d.data()))) return rc;
      host_build(d.data(), nullptr, ts->trees.back());
    } else {
      ts->gpu_trees++;
    }
  } else {
    if ((rc = matrix(user, start, d.data()))) return rc;
    build_tree(d.data(), nullptr, ts->trees.back());  // :447, no prior for the first tree
  }
  ts->trees.back().pos = start;
  std::fill(ts->trees.back().snp_begin.begin(), ts->trees.back().snp_begin.end(), start);
  set_carriers(start);
  ts->info[0].tree = 0;
  int is_mapping = map_mutation(*ts, ts->trees.back(), ts->info[0], min_value, ts->state[start] != 0);
  if (is_mapping > 2) force_map_mutation(*ts, ts->trees.back(), ts->info[0]);

  int num_tree = 1;
  const float val = -std::log(ts->theta / (1.0 - ts->theta));  // :555
  // RELATE_AMD_TIMING=1: where a section's wall-clock goes (stderr)
  const bool timing = getenv("RELATE_AMD_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_matrix = 0, t_prior = 0, t_build = 0, t_map = 0, t_mark = 0;
  int builds = 0;
  auto lap = [&](double &acc) {
    const double t = now();
    acc += t - t_mark;
    t_mark = t;
  };
  for (int snp = start + 1; snp <= end; snp++) {
    SnpInfo &si = ts->info[(size_t)(snp - start)];
    set_carriers(snp);
    if (advance && snp < end && (rc = advance(user, snp))) return rc;  // :487-495 (carriers only)
    si.tree = num_tree - 1;
    const bool use = ts->state[snp] != 0;
    t_mark = now();
    is_mapping = map_mutation(*ts, ts->trees.back(), si, min_value, use);
    lap(t_map);
    bool force_new = false;
    if (snp < end && fb > 0)
      if (((int)(ts->bp[snp + 1] / fb)) - ((int)(ts->bp[snp] / fb)) >= 1) force_new = true;

    if (is_mapping > 1 || force_new) {
      int prev_branch = -1;
      if (is_mapping == 2 || (is_mapping == 1 && force_new)) prev_branch = si.branch[0];
      ts->trees.emplace_back();
      HostTree &nt = ts->trees.back();
      HostTree &pt = ts->trees[ts->trees.size() - 2];
      t_mark = now();
      builds++;
      if (resident) {  // matrix, carrier penalty, clade prior and the build itself on the device
        float *dd = dev->device_matrix();
        if (!dd) return RL_ENOMEM;
        float *rmin = ts->matrix_dev_ex ? dev->rowmin_device() : nullptr;
        if (rmin) {  // the penalty and the row minima in the matrix kernel's own last pass over each row
          if ((rc = ts->matrix_dev_ex(user, snp, dd, consistency ? ts->member.data() : nullptr, val, rmin))) return rc;
          dev->rowmin_is_ready();
        } else if ((rc = ts->matrix_dev(user, snp, dd))) {
          return rc;
        }
        lap(t_matrix);
        int st = 0;
        if (consistency) {
          if (!rmin) st = dev->apply_penalty(ts->member.data(), val);
          st = st ? st : dev->apply_prior(pt, val);
          lap(t_prior);
        }
        st = st ? st : build_resident(consistency, nt);
        if (st < 0) {
          set_error("tree builder on the device failed at SNP %d", snp);
          return RL_EHIP;
        }
        if (st > 0) {
          // The device hands this tree to the host (more tied candidates than its lists hold, or no room for
          // the symmetric matrix): the device matrices are half merged by now, so the distance matrix comes
          // again, to the host, with the penalty and the prior of the host path.  tb is untouched.
          if ((rc = matrix(user, snp, d.data()))) return rc;
          if (consistency) {
            parallel_rows(N, [&](int c) {
              if (ts->member[c]) {
                float *row = &d[(size_t)c * N];
                for (int col = 0; col < N; col++) row[col] += val;
                for (int c2 = 0; c2 < N; c2++)
                  if (ts->member[c2]) row[c2] -= val;
              }
            });
            clade_prior(pt, val, dist);
          }
          host_build(d.data(), consistency ? dist.data() : nullptr, nt);
        } else {
          ts->gpu_trees++;
        }
      } else if ((rc = matrix(user, snp, d.data()))) {
        return rc;
      } else if (consistency) {
        lap(t_matrix);
        // carrier penalty (:563-581): d[c][*] += val, then d[c][c'] -= val
        parallel_rows(N, [&](int c) {
          if (ts->member[c]) {
            float *row = &d[(size_t)c * N];
            for (int col = 0; col < N; col++) row[col] += val;
            for (int c2 = 0; c2 < N; c2++)
              if (ts->member[c2]) row[c2] -= val;
          }
        });
        clade_prior(pt, val, dist);
        lap(t_prior);
        build_tree(d.data(), dist.data(), nt);
      } else {
        build_tree(d.data(), nullptr, nt);
      }
      if (build_rc) return build_rc;
      lap(t_build);
      nt.pos = snp;
      const int is_mapping_alt = map_mutation(*ts, nt, si, min_value_alt, use);
      lap(t_map);
      if (is_mapping_alt > 1 && min_value_alt >= min_value && !force_new) {
        // new tree is not better: keep the old one (:621-630)
        if (is_mapping == 2) si.branch[0] = prev_branch;
        ts->trees.pop_back();
        if (is_mapping > 2) force_map_mutation(*ts, ts->trees.back(), si);
      } else {
        if (is_mapping == 2 || (is_mapping == 1 && force_new)) {
          if (use) pt.num_events[prev_branch] -= 1.0f;
        }
        if (is_mapping_alt > 2) force_map_mutation(*ts, nt, si);
        si.tree = num_tree;
        std::fill(pt.snp_end.begin(), pt.snp_end.end(), snp);
        std::fill(nt.snp_begin.begin(), nt.snp_begin.end(), snp);
        num_tree++;
      }
    }
  }
  std::fill(ts->trees.back().snp_end.begin(), ts->trees.back().snp_end.end(), end);
  if (timing) {
    // (the host builder's own phases only when it built trees: for the device's workers the per-phase times are the
    //  "[gpu tree builder]" lines, one per tree, and its host side the line the builder prints when it is destroyed)
    char host_phases[256] = "";
    if (ts->host_trees > 0)
      snprintf(host_phases, sizeof host_phases,
               "; host builder: row minima + pair scan %.2f, merges: parallel part %.2f [updates %.2f] + ordered part "
               "%.2f, %.2f rebuilt clusters per merge", tb.t_init, tb.t_phase1, tb.t_phase1a, tb.t_phase2,
               (double)tb.n_updated / std::max<long long>(1, tb.n_merges));
    fprintf(stderr,
            "[tree sequence] SNPs %d..%d: %d trees kept of %d built; distance matrices %.2f s, penalty + clade prior "
            "%.2f s, tree builds %.2f s (%lld trees on the GPU, %lld on the host%s), mutation mapping %.2f s\n",
            start, end, num_tree, builds + 1, t_matrix, t_prior, t_build, ts->gpu_trees, ts->host_trees, host_phases, t_map);
  }
  return RL_OK;
}

int rl_treeseq_set_sample_ages(rl_treeseq *ts, const double *ages, int n) {
  if (!ts || (n != 0 && (n != ts->N || !ages))) {
    rl::set_error("rl_treeseq_set_sample_ages: one age per haplotype (%d), or none", ts ? ts->N : 0);
    return RL_EINVAL;
  }
  ts->sample_ages.assign(ages, ages + n);
  if (ts->dev_builder) ts->dev_builder->forget_ages();
  return RL_OK;
}

int rl_treeseq_set_build_device(rl_treeseq *ts, int device) {
  if (!ts) return RL_EINVAL;
  if (ts->dev_builder && device != ts->build_device) {
    delete ts->dev_builder;
    ts->dev_builder = nullptr;
  }
  ts->build_device = device;
  return RL_OK;
}

int rl_treeseq_set_device_matrix_ex(rl_treeseq *ts, rl_matrix_dev_ex_fn matrix_dev_ex) {
  if (!ts) return RL_EINVAL;
  ts->matrix_dev_ex = matrix_dev_ex;
  return RL_OK;
}

int rl_treeseq_set_device_matrix(rl_treeseq *ts, rl_matrix_dev_fn matrix_dev) {
  if (!ts) return RL_EINVAL;
  ts->matrix_dev = matrix_dev;
  return RL_OK;
}

int rl_treeseq_num_trees(const rl_treeseq *ts) { return ts ? (int)ts->trees.size() : RL_EINVAL; }

int rl_treeseq_get_tree(const rl_treeseq *ts, int t, int *pos, int *parent) {
  if (!ts || t < 0 || t >= (int)ts->trees.size()) return RL_EINVAL;
  if (pos) *pos = ts->trees[t].pos;
  if (parent) memcpy(parent, ts->trees[t].parent.data(), sizeof(int) * (2 * ts->N - 1));
  return RL_OK;
}

// AncesTree::DumpBin (anc.cpp:1104-1167) and Mutations::DumpShortFormat (mutations.cpp:548-581)
int rl_treeseq_write(const rl_treeseq *ts, const char *anc_path, const char *mut_path) {
  if (!ts || ts->trees.empty()) {
    set_error("rl_treeseq_write: nothing built");
    return RL_ESTATE;
  }
  const int N = ts->N, T = 2 * N - 1;
  if (anc_path) {
    FILE *fp = fopen(anc_path, "wb");
    if (!fp) {
      set_error("cannot open %s for writing", anc_path);
      return RL_EIO;
    }
    const unsigned char has_ages = 0;
    const unsigned int uN = N, nt = (unsigned int)ts->trees.size();
    fwrite(&has_ages, 1, 1, fp);
    fwrite(&uN, 4, 1, fp);
    fwrite(&nt, 4, 1, fp);
    std::vector<unsigned char> buf((size_t)T * 24);
    for (const HostTree &t : ts->trees) {
      fwrite(&t.pos, 4, 1, fp);
      const double bl = 0.0;
      for (int i = 0; i < T; i++) {
        unsigned char *p = &buf[(size_t)i * 24];
        memcpy(p, &t.parent[i], 4);
        memcpy(p + 4, &bl, 8);
        memcpy(p + 12, &t.num_events[i], 4);
        memcpy(p + 16, &t.snp_begin[i], 4);
        memcpy(p + 20, &t.snp_end[i], 4);
      }
      fwrite(buf.data(), 1, buf.size(), fp);
    }
    const bool bad = ferror(fp) != 0;
    if (fclose(fp) != 0 || bad) {  // (a full disc must not pass for a tree sequence)
      set_error("writing %s failed", anc_path);
      return RL_EIO;
    }
  }
  if (mut_path) {
    FILE *fp = fopen(mut_path, "w");
    if (!fp) {
      set_error("cannot open %s for writing", mut_path);
      return RL_EIO;
    }
    fputs("tree_index;branch_index;is_mapping;is_flipped;age_of_mutation\n", fp);
    for (const SnpInfo &si : ts->info) {
      fprintf(fp, "%d;", si.tree);
      for (size_t b = 0; b < si.branch.size(); b++) fprintf(fp, b ? " %d" : "%d", si.branch[b]);
      fputs(si.branch.size() > 1 ? ";1;" : ";0;", fp);
      fprintf(fp, "%d;0;0;\n", si.flipped ? 1 : 0);
    }
    const bool bad = ferror(fp) != 0;
    if (fclose(fp) != 0 || bad) {
      set_error("writing %s failed", mut_path);
      return RL_EIO;
    }
  }
  return RL_OK;
}

// ---- the stage: pipeline/BuildTopology.cpp:14-167 -------------------------
// Sections are independent (the reference's scripts run them as separate
// processes, scripts/RelateParallel/RelateParallel.sh:231-257): here a few host
// threads each run one section's tree-sequence loop (MinMatch etc. on the
// host) and share the GPU -- window opening / RePaint and every distance
// matrix go through one mutex, the GPU work per call being milliseconds.
namespace {
std::mutex g_gpu_mutex;
std::string g_stage_sample_ages;  // rl_stage_set_sample_ages: the --sample_ages file of the following stage calls
}
// (a window's matrices run on the window's own stream; what the windows of a context share -- RePaint's strips --
//  is locked inside, window.cpp)
static int win_matrix(void *user, int snp, float *d) { return rl_window_matrix((rl_window *)user, snp, d, nullptr); }
static int win_matrix_dev(void *user, int snp, void *d_dev) {
  return rl_window_matrix_rows_device((rl_window *)user, snp, d_dev, nullptr);
}
static int win_matrix_dev_ex(void *user, int snp, void *d_dev, const char *carriers, float val, void *d_rowmin) {
  return rl_window_matrix_rows_device_ex((rl_window *)user, snp, d_dev, carriers, val, d_rowmin, nullptr);
}
static int win_advance(void *user, int snp) { return rl_window_advance((rl_window *)user, snp); }

// Host cores that share a last-level cache, as cpu sets this process may run on.  One section (its builder thread
// and the helpers that split a merge, minmatch.h) is kept inside one such group: the helpers hand each other a few
// cache lines per merge, tens of thousands of times a second, and the matrices are first touched -- so placed on
// the NUMA node -- by the thread that builds with them.  RELATE_AMD_PIN=0 leaves placement to the scheduler.
// A stage's knobs: the caller's rl_stage_opts (per call), overridden by the RELATE_AMD_* environment variables --
// those are for experiments (tools/, profiles/): what a consumer of the ABI sets goes through the struct.
static long long knob(const char *env, long long from_opts, bool opts_set, long long fallback) {
  if (const char *e = getenv(env)) return atoll(e);
  return opts_set ? from_opts : fallback;
}

static std::vector<cpu_set_t> cache_groups(int pin_opt) {
  std::vector<cpu_set_t> groups;
  if (knob("RELATE_AMD_PIN", pin_opt, pin_opt >= 0, 1) == 0) return groups;
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return groups;
  std::vector<std::string> keys;
  for (int cpu = 0; cpu < CPU_SETSIZE; cpu++) {
    if (!CPU_ISSET(cpu, &allowed)) continue;
    char path[128];
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    FILE *fp = fopen(path, "r");
    if (!fp) return std::vector<cpu_set_t>();
    char buf[256] = {0};
    const bool ok = fgets(buf, sizeof(buf), fp) != nullptr;
    fclose(fp);
    if (!ok) return std::vector<cpu_set_t>();
    size_t g = 0;
    while (g < keys.size() && keys[g] != buf) g++;
    if (g == keys.size()) {
      keys.push_back(buf);
      cpu_set_t empty;
      CPU_ZERO(&empty);
      groups.push_back(empty);
    }
    CPU_SET(cpu, &groups[g]);
  }
  if (groups.size() < 2) groups.clear();  // one cache for everything: nothing to choose
  return groups;
}

// The sections [first_section, last_section] of the chunk loaded in ctx: windows from the paint files of the Paint
// stage (from_files) or from the stepping stones the context has just painted, which never leave HBM (the file's
// float / run-length quantisation applied on the device, window.cpp).  Consumes ctx.
// --sample_ages file (pipeline/BuildTopology.cpp:93-108): the first N numbers of a plain or gzip text file.  A file
// that cannot be opened leaves N zeros (and a warning), one with fewer than N numbers no ages at all -- as there.
static std::vector<double> read_sample_ages(const char *fn, int N) {
  std::vector<double> ages(N, 0.0);
  gzFile gz = gzopen(fn, "rb");  // (plain text passes through zlib unchanged; no shell, whatever the path holds)
  if (!gz) {
    std::cerr << "Warning: unable to open sample ages file" << std::endl;
    return ages;
  }
  std::string text;
  char buf[1 << 16];
  int got;
  while ((got = gzread(gz, buf, sizeof(buf))) > 0) text.append(buf, (size_t)got);
  const bool broken = got < 0;
  gzclose(gz);
  int i = 0;
  const char *p = text.c_str();
  while (i < N) {
    char *e = nullptr;
    const double v = strtod(p, &e);
    if (e == p) break;
    ages[i++] = v;
    p = e;
  }
  if (i < N) {  // (the reference's loop leaves the vector short and the builder then ignores it: no ages)
    std::cerr << "Warning: " << fn << (broken ? " cannot be decompressed: " : " holds ") << i << " of " << N
              << " sample ages; building without sample ages" << std::endl;
    ages.clear();
  }
  return ages;
}

static int build_sections(rl_ctx *ctx, const char *out_dir, int chunk_index, int first_section, int last_section,
                          const rl_stage_opts &o, bool from_files) {
  int rc = RL_OK;
  const int flags = o.flags, fb = o.fb, sum_mode = o.sum_mode, device = o.device;
  const char *sample_ages_file = (o.sample_ages_path && *o.sample_ages_path) ? o.sample_ages_path : nullptr;
  const auto entry_t0 = std::chrono::steady_clock::now();
  const int W = ctx->W, L = ctx->L;
  if (first_section >= W) {  // BuildTopology.cpp:45
    rl_destroy(ctx);
    return 1;
  }
  last_section = std::min(W - 1, last_section);
  const std::string od(out_dir), c = std::to_string(chunk_index);
  // output name = basename of -o (Relate requires -o to be a bare name, Relate.cpp:50-58)
  std::string base = od;
  while (!base.empty() && base.back() == '/') base.pop_back();
  const size_t sl = base.find_last_of('/');
  if (sl != std::string::npos) base = base.substr(sl + 1);
  std::vector<double> sample_ages;
  if (sample_ages_file) sample_ages = read_sample_ages(sample_ages_file, ctx->N);
  std::vector<int> bp(L, 0), state(L, 1);
  // chunk_<c>.bp / .state (data.cpp:485-516, :307-345): --fb and the mapping of transitions read them; a missing or
  // short file would give wrong trees without a word
  auto read_ints = [&](const std::string &fn, std::vector<int> &v) -> int {
    FILE *fp = fopen(fn.c_str(), "rb");
    if (!fp) {
      set_error("cannot open %s", fn.c_str());
      return RL_EIO;
    }
    int n = 0;
    const bool ok = fread(&n, 4, 1, fp) == 1 && n == L && fread(v.data(), 4, (size_t)L, fp) == (size_t)L;
    fclose(fp);
    if (!ok) {
      set_error("%s does not hold %d values", fn.c_str(), L);
      return RL_EFORMAT;
    }
    return RL_OK;
  };
  if ((rc = read_ints(od + "/chunk_" + c + ".bp", bp)) || (rc = read_ints(od + "/chunk_" + c + ".state", state))) {
    rl_destroy(ctx);
    return rc;
  }
  // rl_stage_opts.find_equivalent_branches: checked before any device or builder set-up
  const bool want_feb = knob("RELATE_AMD_FUSED_FEB", o.find_equivalent_branches, true, 0) != 0;
  if (want_feb && (first_section != 0 || last_section != W - 1 || W < 2)) {
    set_error("find_equivalent_branches needs a call that covers all %d sections of the chunk (and at least two)", W);
    rl_destroy(ctx);
    return RL_EINVAL;
  }
  std::cerr << "---------------------------------------------------------" << std::endl;
  std::cerr << "Estimating topologies of AncesTrees in sections " << first_section << "-" << last_section << "..."
            << std::endl;
  // How many sections are open at once is bounded by host threads and by HBM: a window's posterior rows (sum_n D_n
  // rows of S*64*waves floats, 20 GB at N = 5000) stay resident while its trees are built.  When the sections the
  // host could work on do not fit, every window keeps a part of its rows and repaints as its builder moves on
  // (rl_window_open_bounded; RELATE_AMD_WINDOW_ROWS sets the rows per window by hand).  Every open is admitted
  // against the HBM that is free at that moment (window_bytes below); the estimate here sizes the thread pools.
  if ((!ctx->plan.valid && build_plan(ctx)) || upload_plan(ctx)) {  // (before the section threads open their windows)
    rl_destroy(ctx);
    return RL_EINVAL;
  }
  const double row_bytes = 4.0 * ctx->S * 64 * ctx->waves;
  long long cap_rows = 0;  // posterior rows a window keeps resident; 0: all
  double max_rows = 0;
  int maxD = 1;
  std::vector<double> rows_of(W, 0.0);
  for (int w = first_section; w <= last_section; w++) {
    for (int n = ctx->k0; n < ctx->k0 + ctx->nloc; n++) {
      const int D = ctx->plan.ie[(size_t)n * W + w] - ctx->plan.ia[(size_t)n * W + w] + 1;
      rows_of[w] += D;
      maxD = std::max(maxD, D);
    }
    max_rows = std::max(max_rows, rows_of[w]);
  }
  // The trees themselves are built on the GPU too (minmatch_gpu.hip) when the call covers several sections: a tree
  // takes one workgroup ~115 ms at N = 5000 against ~85 ms on 8 host threads, but the workgroups of different
  // sections run side by side (40 sections: 98 s against 331 s).  RELATE_AMD_GPU_BUILD=0 / 1 decides otherwise.
  const bool gpu_build = knob("RELATE_AMD_GPU_BUILD", o.gpu_build, o.gpu_build >= 0, last_section > first_section) != 0;
  // Next to its rows a window holds cursors and small per-target arrays, a stage from paint files also the decoded
  // stones (2 N^2 floats; the fused stage re-paints from the context's own); a tree builder on the device keeps the
  // woven float4 matrix of the build (16 N^2 B, minmatch_gpu.hip) -- the row-major matrices of a tree and the
  // symmetric matrix of the fallback come from pools of the device, counted once.  RePaint's strips are one buffer
  // of the context (reserved below).
  const double NN = (double)ctx->N * (ctx->N + 64.0);
  const double builder_bytes = gpu_build ? 17.0 * NN : 0.0;  // (allocated once per section thread, kept)
  const double fixed_bytes = 4.0 * max_rows + 48e6 + (from_files || ctx->h_alpha ? 8.0 * ctx->N * ctx->nloc : 0.0) +
                             (gpu_build && ctx->nloc == ctx->N ? 0.0 : 4.0 * ctx->N * ctx->nloc);
  if (gpu_build && device_builder_reserve_shared(device, ctx->N)) {  // the pools now: the admission sees what is left
    rl_destroy(ctx);
    return RL_ENOMEM;
  }
  // (a bounded window also keeps one state of RePaint's backward pass per target, doubles: window.cpp)
  const double bstate_bytes = 2.0 * (2.0 * row_bytes * ctx->nloc + 24.0 * ctx->nloc);  // (one of each pass)
  auto window_bytes = [&](int w) {
    const double kept = cap_rows > 0 ? std::min(rows_of[w], (double)cap_rows) : rows_of[w];
    return kept * row_bytes + fixed_bytes + (cap_rows > 0 && cap_rows < rows_of[w] ? bstate_bytes : 0.0);
  };
  // (section threads of device builds mostly wait for their tree: as many as there are CUs to build on)
  int nthreads = gpu_build ? 256 : std::max(1, std::min(host_threads() / 2, 64));  // (host_threads: this rank's share)
  nthreads = std::max(1, (int)knob("RELATE_AMD_SECTION_THREADS", o.section_threads, o.section_threads > 0, nthreads));
  nthreads = std::min(nthreads, last_section - first_section + 1);
  int concurrent = nthreads;
  {
    const size_t strips = repaint_scratch_bytes((int64_t)max_rows, ctx->nloc, ctx->S, ctx->waves);
    if (ctx->d_k2_scratch.alloc(strips)) {
      rl_destroy(ctx);
      return RL_ENOMEM;
    }
    size_t free_b = 0, total_b = 0;
    const bool known = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    const double room = known ? 0.9 * (double)free_b : 0.0;
    if (getenv("RELATE_AMD_WINDOW_ROWS") || o.window_rows != 0) {  // (rows per window by hand; < 0 in the struct: all of them)
      cap_rows = std::max(0LL, knob("RELATE_AMD_WINDOW_ROWS", o.window_rows, true, 0));
    } else if (known) {
      // The trees of different sections are what fills the chip (a workgroup per tree), so as many sections as HBM
      // holds should be open -- but a window that keeps 1/P of its rows runs RePaint P times, about half a window's
      // pass each time, on the CUs the builders leave free: not below a P-th of the largest window,
      // P <= RELATE_AMD_WINDOW_PARTS (32).  And the sections go in WAVES of whatever is open at once: 267 sections
      // 112 at a time are three waves, the last one a third full; 134 at a time are two full ones.  So: the fewest
      // waves the memory allows at P_max, the sections spread evenly over them, and the smallest P that opens
      // that many (C3: 2 waves of 134, P = 20: 234 s -> see DESIGN_NOTES.md 6).
      const int parts_max = std::max(1, (int)knob("RELATE_AMD_WINDOW_PARTS", o.window_parts, o.window_parts > 0, 32));
      auto fits = [&](int parts) {
        return (int)(room / (max_rows / parts * row_bytes + fixed_bytes + builder_bytes + (parts > 1 ? bstate_bytes : 0.0)));
      };
      const int nsec = last_section - first_section + 1;
      const int most = std::max(1, std::min(nthreads, fits(parts_max)));
      const int waves = (nsec + most - 1) / most;
      nthreads = std::min(nthreads, (nsec + waves - 1) / waves);
      int parts = 1;
      while (parts < parts_max && fits(parts) < nthreads) parts++;
      if (parts > 1) cap_rows = (long long)std::max({max_rows / parts, 3.0 * ctx->nloc + 64.0});
    }
    // A SECOND lane of RePaint launches for bounded windows (common.h; RELATE_AMD_REPAINT_LANES / rl_stage_opts decide
    // otherwise): a launch is a forward and a backward kernel of one workgroup per target, each as long as its longest
    // target, on the CUs the tree builder's workers leave -- two windows' launches side by side fill each other's
    // tails, and no section waits for its turn behind a hundred others.  Round 3 sized its strips like the first
    // lane's, for a window's first whole pass (7 GB at C3: 45 sections' worth of rows); a partial launch addresses its
    // strips compactly (window.cpp place_rows) and the second lane takes partial launches only: kept rows / 6 + 2 per
    // target rows of doubles + the side records, 0.7 GB.  NOT the default all the same: with all 134 sections open the
    // two lanes' launches share the CUs the workers leave and each takes as much longer as it overlaps (C3, 104
    // workers: 162.6 s with two lanes, 161.9 s with one; 116 workers 166.5 s; 128 workers 219 s,
    // profiles/r04_c3_workers.json) -- RePaint is bound by the CUs it gets, not by its queue.
    if (last_section > first_section && cap_rows > 0 &&
        knob("RELATE_AMD_REPAINT_LANES", o.repaint_lanes, o.repaint_lanes > 0, 1) == 2 && !ctx->two_lanes) {
      auto &ln = ctx->lane2;
      const size_t row_doubles = (size_t)ctx->S * 64 * ctx->waves;
      const size_t small = (size_t)(((double)cap_rows / REPAINT_CHECKPOINT + 2.0 * ctx->nloc + 64.0) * (double)row_doubles +
                                    max_rows * REPAINT_SIDE) * sizeof(double);
      if (ln.scratch.alloc(small) == 0 && make_stream(&ln.s, false) == hipSuccess &&
          hipEventCreate(&ln.e0) == hipSuccess && hipEventCreate(&ln.e1) == hipSuccess) {
        ctx->two_lanes = true;
      } else {
        (void)hipGetLastError();
        ln.scratch.release();
      }
    }
    if (known) {
      const double per_window = window_bytes((first_section + last_section) / 2) + builder_bytes;
      concurrent = std::max(1, std::min(nthreads, (int)(room / std::max(per_window, 1.0))));
      nthreads = std::min(nthreads, concurrent);
    }
  }
  int asked_workers = 0;
  if (gpu_build) {
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 8) cus = 256;
    int workers = stage_worker_goal(cus, nthreads, cap_rows > 0, device_builder_workers_per_cu(ctx->N, !sample_ages.empty()));
    if (o.workers > 0) workers = o.workers;  // (RELATE_AMD_BUILD_WORKERS overrides either, minmatch_gpu.hip)
    (void)device_builder_expect(device, ctx->N, workers, !sample_ages.empty());
    asked_workers = workers;
  }
  // Host threads left over by the sections help inside each tree build (minmatch.h BuildThreads).  Helpers are the
  // less efficient use of a core (a merge is split 8 ways for a 2.5x shorter build) and a helper that loses its
  // core stalls every merge, so they get a quarter of the physical cores at most: measured on 2 x 64 cores with 8
  // sections open, 4 helpers each finished in 109 s, 8 in 177 s, none in 165 s.
  set_build_threads(std::min(8, std::max(1, host_threads() / (8 * std::max(1, concurrent)))));
  // (test hook, read once per stage: tests/test_stage_gpu.py)
  std::atomic<int> fail_opens{getenv("RELATE_AMD_TEST_FAIL_OPENS") ? atoi(getenv("RELATE_AMD_TEST_FAIL_OPENS")) : 0};
  std::atomic<int> open_sections(0);
  int most_open = 0;            // (under g_gpu_mutex)
  const auto stage_t0 = std::chrono::steady_clock::now();
  if (getenv("RELATE_AMD_TIMING"))
    fprintf(stderr, "[stage] set-up before the section threads (plan, reservations) %.3f s\n",
            std::chrono::duration<double>(stage_t0 - entry_t0).count());
  double reserved_bytes = 0.0;  // (under g_gpu_mutex) HBM promised to windows that are being opened
  // The sections are dealt to the threads longest first: with about two sections per thread (C3: 267 on 134) and
  // 228 - 496 trees per section, dealing them in index order left the threads with 500 - 950 trees each, and the
  // stage ended on a long tail of a few open sections with most of the tree builder's workers idle (the same deal
  // replayed with the measured tree counts: 214 s in index order, 176 s longest first, 174 s perfectly even).  How
  // many trees a section needs is not known before its SNPs are mapped; its SNP count is (correlation 0.90).
  // The files of a section do not depend on when it is built.
  std::vector<int> deal(last_section - first_section + 1);
  for (size_t x = 0; x < deal.size(); x++) deal[x] = first_section + (int)x;
  if (nthreads > 1)
    std::stable_sort(deal.begin(), deal.end(), [&](int a, int b) {
      auto snps = [&](int w) { return (w < W - 1 ? ctx->wb[w + 1] : L) - ctx->wb[w]; };
      return snps(a) > snps(b);
    });
  std::atomic<int> next(0);
  std::atomic<int> first_error(0);
  std::mutex err_mutex;
  std::string first_message;  // (rl_last_error is per thread: the failing worker's text is carried to the caller)
  auto fail = [&](int code) {
    std::lock_guard<std::mutex> lk(err_mutex);
    int expected = 0;
    if (first_error.compare_exchange_strong(expected, code)) first_message = rl_last_error();
  };
  const std::vector<cpu_set_t> groups = cache_groups(o.pin_threads);
  std::atomic<int> next_slot(0);
  // rl_stage_opts.find_equivalent_branches: the stage downstream on the trees while they are in memory (equivalent.cpp)
  FebJob *feb = nullptr;
  if (want_feb) feb = feb_job_create(ctx->N, W, std::max(4, std::min(32, host_threads() / 8)));
  auto worker = [&]() {
    cpu_set_t before;
    CPU_ZERO(&before);
    const bool pinned = !groups.empty() && sched_getaffinity(0, sizeof(before), &before) == 0;
    if (pinned) {  // slots alternate between the two halves of the list (= the sockets, as the cpus are numbered)
      // (several ranks on the host, one per GPU: each keeps to its own share of the groups)
      const int LW = std::min(local_world_size(), (int)groups.size()), lr = local_rank() % LW;
      const int g0 = (int)groups.size() * lr / LW, G = (int)groups.size() * (lr + 1) / LW - g0;
      const int slot = next_slot.fetch_add(1), half = (G + 1) / 2;
      const int g = g0 + ((slot % 2) * half + (slot / 2) % half) % G;
      sched_setaffinity(0, sizeof(cpu_set_t), &groups[g]);
    }
    // (the panel is the context's, immutable for the stage: a copy per section thread would be 0.3 GB each at C3)
    rl_treeseq *ts = treeseq_borrowing(ctx->N, L, ctx->bits.data(), ctx->row_words, ctx->rpos.data(), bp.data(),
                                       state.data(), ctx->theta);
    if (!ts) {
      fail(RL_EINVAL);
      if (pinned) sched_setaffinity(0, sizeof(before), &before);
      return;
    }
    if (!sample_ages.empty()) rl_treeseq_set_sample_ages(ts, sample_ages.data(), (int)sample_ages.size());
    if (gpu_build) {
      rl_treeseq_set_build_device(ts, device);
      if (ctx->nloc == ctx->N) {
        rl_treeseq_set_device_matrix(ts, win_matrix_dev);
        rl_treeseq_set_device_matrix_ex(ts, win_matrix_dev_ex);
      }
    }
    for (;;) {
      const int turn = next.fetch_add(1);
      if (turn >= (int)deal.size() || first_error.load()) break;
      const int section = deal[turn];
      const int start = ctx->wb[section];
      int end = (section < W - 1) ? ctx->wb[section + 1] - 1 : L - 1;
      if (end >= L) end = L - 1;
      const std::string pf = od + "/chunk_" + c + "/paint/relate_" + std::to_string(section) + ".bin";
      rl_window *win = nullptr;
      // (this thread's builder keeps its buffers from section to section: only the first one asks for them)
      const bool builder_new = gpu_build && !ts->dev_builder;
      const double need = window_bytes(section) + (builder_new ? builder_bytes : 0.0);
      // cache_alloc hands out whole cached blocks: of what the cache holds only blocks that can take one of the
      // window's large buffers (a kept state of a bounded window, or else its rows) count as room
      const double kept_rows = cap_rows > 0 ? std::min(rows_of[section], (double)cap_rows) : rows_of[section];
      const size_t min_block = (size_t)(0.85 * (cap_rows > 0 && cap_rows < rows_of[section]
                                                    ? std::min(kept_rows * row_bytes, bstate_bytes / 4.0)
                                                    : kept_rows * row_bytes));
      int open_rc = RL_EIO, oom_alone = 0, oom_total = 0;
      for (;;) {  // admission: wait until the window fits next to the ones that are open or being opened
        bool admitted = false;
        {
          std::lock_guard<std::mutex> lk(g_gpu_mutex);
          size_t free_b = 0, total_b = 0;
          const bool known = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
          // (closed windows give their blocks to the library's cache, not to the driver: what the cache holds in
          //  blocks of a useful size is free for the next window too)
          if (!known || (double)free_b + (double)device_cache_held(device, min_block) - reserved_bytes >= need ||
              (open_sections.load() == 0 && reserved_bytes == 0.0)) {
            reserved_bytes += need;
            admitted = true;
          }
        }
        if (admitted) {  // (reading and decoding the paint file, the uploads and RePaint: outside the lock)
          const unsigned oom_before = tl_alloc_failures;
          // (test hook, RELATE_AMD_TEST_FAIL_OPENS=k: k > 0 -- the next k admitted opens that happen while another
          //  section is open fail as if the device were out of memory; k < 0 -- the next |k| opens whatever else is
          //  open.  The retry below is otherwise reached only when the allocator really runs dry)
          bool injected = false;
          if (fail_opens.load() > 0 && open_sections.load() > 0) injected = fail_opens.fetch_sub(1) > 0;
          else if (fail_opens.load() < 0) injected = fail_opens.fetch_add(1) < 0;
          if (injected) {
            tl_alloc_failures++;
            set_error("hipMalloc failed (injected by RELATE_AMD_TEST_FAIL_OPENS)");
            win = nullptr;
          } else
          win = rl_window_open_bounded(ctx, section, from_files ? pf.c_str() : nullptr, start, sum_mode, cap_rows,
                                       nullptr);
          if (win && builder_new) {  // (while the reservation stands)
            ts->dev_builder = new DeviceMinMatch(ctx->N, device);
            if (ts->dev_builder->reserve(!sample_ages.empty())) {
              rl_window_close(win);
              win = nullptr;
              delete ts->dev_builder;
              ts->dev_builder = nullptr;
              tl_alloc_failures++;
            }
          }
          bool others;
          {
            std::lock_guard<std::mutex> lk(g_gpu_mutex);
            reserved_bytes -= need;
            if (win) most_open = std::max(most_open, ++open_sections);
            others = open_sections.load() > 0 || reserved_bytes > 0.0;
          }
          if (win) break;
          // Out of memory although admitted (the estimate counts bytes, the allocator needs blocks): not the
          // stage's failure while other sections hold memory they will give back -- wait for one to close and ask
          // again.  Alone on the device it IS the failure.
          // (bounded: two threads that keep failing while each sees the other's reservation would otherwise retry for
          //  ever -- after 20 admitted-but-failed opens of this section it IS the stage's failure)
          if (tl_alloc_failures != oom_before && others && !first_error.load() && ++oom_total <= 20) {
            open_rc = RL_ENOMEM;
            const int open_then = open_sections.load();
            for (int i = 0; i < 1200 && open_sections.load() >= open_then && open_sections.load() > 0 &&
                            !first_error.load(); i++)
              std::this_thread::sleep_for(std::chrono::milliseconds(50));
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
            continue;
          }
          // (alone by now -- the sections that held memory a moment ago may just have closed, and what they gave back
          //  sits in the cache: twice more before it is the stage's failure)
          if (tl_alloc_failures != oom_before && !first_error.load() && ++oom_alone <= 2) {
            open_rc = RL_ENOMEM;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
            continue;
          }
          if (tl_alloc_failures != oom_before) open_rc = RL_ENOMEM;
          break;
        }
        if (first_error.load()) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
      }
      int r = win ? RL_OK : (first_error.load() ? first_error.load() : open_rc);
      const double t_open = std::chrono::duration<double>(std::chrono::steady_clock::now() - stage_t0).count();
      if (!r) r = rl_treeseq_build(ts, start, end, win_matrix, win_advance, win, flags, fb);
      if (getenv("RELATE_AMD_TIMING"))  // (when the sections start and end: the stage's ramp and tail)
        fprintf(stderr, "[section %d] turn %d, %d SNPs: window open %.1f s after the stage began, trees built at %.1f s\n",
                section, turn, end - start + 1, t_open,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - stage_t0).count());
      if (win) {
        std::lock_guard<std::mutex> lk(g_gpu_mutex);
        rl_window_close(win);
        open_sections--;
      }
      if (!r) {
        const std::string b = od + "/chunk_" + c + "/" + base + "_" + std::to_string(section);
        // (fused FindEquivalentBranches: the .anc is written at the end, once, with what that stage carries across)
        r = rl_treeseq_write(ts, feb ? nullptr : (b + ".anc").c_str(), (b + ".mut").c_str());
        if (!r && feb) r = feb_job_add_section(feb, section, ts->trees);
      }
      if (r) {
        fail(r);
        break;
      }
      std::cerr << "[" << section << "/" << last_section << "]\r";
      std::cerr.flush();
    }
    rl_treeseq_destroy(ts);
    if (pinned) sched_setaffinity(0, sizeof(before), &before);
  };
  if (nthreads <= 1) {
    worker();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(worker);
    for (auto &x : th) x.join();
  }
  if (gpu_build) (void)device_builder_expect(device, ctx->N, 0, !sample_ages.empty());
  rc = first_error.load();
  if (rc) set_error("%s", first_message.c_str());
  if (feb) {
    if (!rc) rc = feb_job_finish(feb, od + "/chunk_" + c + "/" + base);
    feb_job_destroy(feb);
    // (as rl_stage_find_equivalent_branches: the last stage that reads chunk_<c>.bits removes it)
    if (!rc) (void)remove((od + "/chunk_" + c + ".bits").c_str());
  }
  if (getenv("RELATE_AMD_TIMING"))
    fprintf(stderr, "[stage] sections %d..%d on %d threads, up to %d open at once, %lld of at most %.0f posterior rows "
            "resident per window, %lld RePaint launches (%.1f s on the device), %s tree builder (%d workers asked for), %.1f s\n",
            first_section, last_section, nthreads, most_open, cap_rows > 0 ? cap_rows : (long long)max_rows, max_rows,
            ctx->repaint_launches.load(), 1e-6 * (double)ctx->repaint_us.load(), gpu_build ? "GPU" : "host", asked_workers,
            std::chrono::duration<double>(std::chrono::steady_clock::now() - stage_t0).count());
  const auto destroy_t0 = std::chrono::steady_clock::now();
  rl_destroy(ctx);
  if (getenv("RELATE_AMD_TIMING"))
    fprintf(stderr, "[stage] context released in %.3f s\n",
            std::chrono::duration<double>(std::chrono::steady_clock::now() - destroy_t0).count());
  if (!rc) {
    rusage usage;
    getrusage(RUSAGE_SELF, &usage);
    std::cerr << "CPU Time spent: " << usage.ru_utime.tv_sec << "." << std::setfill('0') << std::setw(6)
              << usage.ru_utime.tv_usec << "s; Max Memory usage: " << usage.ru_maxrss / 1000.0 << "Mb."
              << std::endl;
    std::cerr << "---------------------------------------------------------" << std::endl << std::endl;
  }
  return rc;
}

int rl_debug_stage_worker_goal(int cus, int open_sections, int bounded_windows, int workers_per_cu) {
  return rl::stage_worker_goal(cus, open_sections, bounded_windows != 0, workers_per_cu);
}

void rl_stage_opts_init(rl_stage_opts *o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->size = sizeof(*o);
  o->sum_mode = RL_SUM_EXACT;
  o->theta = 0.001;
  o->rho = 1.0;
  o->gpu_build = -1;
  o->pin_threads = -1;
}

// the caller's struct, whatever its age: fields past its `size` keep their defaults
// (a struct that never went through rl_stage_opts_init -- size 0, or a size no version of the header ever had -- is
//  refused: silently running with the defaults instead of the caller's options is the worse answer)
int rl_internal_resolve_opts(const rl_stage_opts *in, rl_stage_opts *out) {
  rl_stage_opts_init(out);
  if (!in) return RL_OK;
  if (in->size < 2 * sizeof(size_t) || in->size > 4096 || in->size % sizeof(int) != 0) {
    set_error("rl_stage_opts.size = %zu: initialise the struct with rl_stage_opts_init()", in->size);
    return RL_EINVAL;
  }
  memcpy(out, in, std::min(in->size, sizeof(*out)));
  out->size = sizeof(*out);
  return RL_OK;
}

int rl_stage_build_topology_ex(const char *out_dir, int chunk_index, int first_section, int last_section,
                               const rl_stage_opts *opts) {
  if (!out_dir) return RL_EINVAL;
  rl_stage_opts o;
  if (int orc = rl_internal_resolve_opts(opts, &o)) return orc;
  rl_ctx *ctx = rl_create(o.device);
  if (!ctx) return RL_ENODEVICE;
  int rc = rl_load_chunk(ctx, out_dir, chunk_index);
  if (!rc && o.use_painting) rc = rl_set_painting(ctx, o.theta, o.rho);
  if (rc) {
    rl_destroy(ctx);
    return rc;
  }
  return build_sections(ctx, out_dir, chunk_index, first_section, last_section, o, true);
}

// (the positional form; --sample_ages through the deprecated process-wide rl_stage_set_sample_ages)
static rl_stage_opts positional_opts(int use_painting, double theta, double rho, int flags, int fb, int sum_mode,
                                     int device) {
  rl_stage_opts o;
  rl_stage_opts_init(&o);
  o.use_painting = use_painting;
  o.theta = theta;
  o.rho = rho;
  o.flags = flags;
  o.fb = fb;
  o.sum_mode = sum_mode;
  o.device = device;
  o.sample_ages_path = g_stage_sample_ages.empty() ? nullptr : g_stage_sample_ages.c_str();
  return o;
}

int rl_stage_build_topology(const char *out_dir, int chunk_index, int first_section, int last_section,
                            int use_painting, double theta, double rho, int flags, int fb, int sum_mode,
                            int device) {
  const rl_stage_opts o = positional_opts(use_painting, theta, rho, flags, fb, sum_mode, device);
  return rl_stage_build_topology_ex(out_dir, chunk_index, first_section, last_section, &o);
}

int rl_stage_set_sample_ages(const char *file) {
  g_stage_sample_ages = file ? file : "";
  return RL_OK;
}

int rl_stage_paint_build_topology_ex(const char *out_dir, int chunk_index, int first_section, int last_section,
                                     const rl_stage_opts *opts) {
  if (!out_dir) return RL_EINVAL;
  rl_stage_opts o;
  if (int orc = rl_internal_resolve_opts(opts, &o)) return orc;
  const int device = o.device, sum_mode = o.sum_mode, use_painting = o.use_painting;
  const double theta = o.theta, rho = o.rho;
  // RELATE_AMD_TIMING=1: wall-clock of what precedes the sections on stderr
  const bool timing = getenv("RELATE_AMD_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  auto lap = [&](const char *what) {
    const double t1 = now();
    if (timing) fprintf(stderr, "[fused stage] %-36s %8.3f s\n", what, t1 - t0);
    t0 = t1;
  };
  rl_ctx *ctx = rl_create(device);
  if (!ctx) return RL_ENODEVICE;
  lap("device context");
  int rc = rl_load_chunk(ctx, out_dir, chunk_index);
  lap("read chunk files");
  if (!rc && use_painting) rc = rl_set_painting(ctx, theta, rho);
  if (!rc) {
    std::cerr << "---------------------------------------------------------" << std::endl;
    std::cerr << "Painting sequences (stepping stones stay on the device)..." << std::endl;
    rc = rl_paint(ctx, sum_mode, nullptr);
  }
  lap("plan + uploads + masks + paint kernels");
  // RELATE_AMD_PARK_STONES=1: the stones to pinned host memory, their HBM to the sections' windows -- for chunks that
  // would not fit otherwise; at C3 (53 GB of stones) it opens 112 sections instead of 91 and the chunk takes 330 s
  // instead of 291 s: past ~90 trees in flight the build kernels slow each other down
  if (!rc && knob("RELATE_AMD_PARK_STONES", o.park_stones, true, 0) != 0) rc = rl_park_stones(ctx);
  if (!rc) ctx->stones_disposable = true;  // (nobody writes paint files from this context: the windows may edit them)
  if (!rc) {  // (the Paint stage makes this directory for its files; the trees go there)
    const std::string cdir = std::string(out_dir) + "/chunk_" + std::to_string(chunk_index);
    if (mkdir(cdir.c_str(), 0777) != 0 && errno != EEXIST) {
      set_error("cannot create %s", cdir.c_str());
      rc = RL_EIO;
    }
  }
  if (rc) {
    rl_destroy(ctx);
    return rc;
  }
  return build_sections(ctx, out_dir, chunk_index, first_section, last_section, o, false);
}

int rl_stage_paint_build_topology(const char *out_dir, int chunk_index, int first_section, int last_section,
                                  int use_painting, double theta, double rho, int flags, int fb, int sum_mode,
                                  int device) {
  const rl_stage_opts o = positional_opts(use_painting, theta, rho, flags, fb, sum_mode, device);
  return rl_stage_paint_build_topology_ex(out_dir, chunk_index, first_section, last_section, &o);
}

}  // extern "C"
