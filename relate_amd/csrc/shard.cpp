// shard.cpp -- one chunk sharded by TARGET haplotype over several GPUs (BASELINE.json config #5).
//
// The reference shards a chunk by section only (scripts/RelateParallel/RelateParallel.sh:231-257): every
// BuildTopology process reads the paint file of its window, i.e. the stepping stones of ALL N targets
// (src/anc_builder.cpp:49-106), re-paints all of them and fills every row of every distance matrix
// (src/anc_builder.cpp:109-207).  At N = 10,000 x L = 200k the stones alone are 8 N^2 W = 288 GB, so no single
// GPU can even hold the chunk.  What the reference's units need from each other, restated for G GPUs:
//
//   * Paint, RePaintSection and row n of GetMatrix depend on target n alone (fast_painting.cpp:18-618, :621-1092,
//     anc_builder.cpp:116-194: the row minimum is taken over the row) -> a rank (one process per GPU) holds the
//     bit-packed panel (replicated, N L / 8 bytes) and the stones, posterior rows and cursors of ITS contiguous
//     range of targets only: an rl_shard;
//   * a section's tree-sequence loop (anc_builder.cpp:398-656) is sequential and needs whole matrices: it runs on
//     ONE rank, the section's owner (rl_shard_build_section), and every matrix it asks for is assembled from the
//     row blocks of all ranks (rl_shard_rows on each of them, exchanged by the caller: one RCCL all-gather of
//     N^2 floats per matrix, relate_amd/dist.py run_chunk_by_targets).
//
// A shard serves rows for ANY section, owned or not: per section it keeps one (bounded) window open and replays the
// cursor updates of anc_builder.cpp:487-495 from the last SNP it was asked about to the next -- they are a function
// of the panel, so the owner only has to name the SNP.
#include <sys/stat.h>

#include <cerrno>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace rl {
// (treeseq.cpp) a tree-sequence object that borrows the panel instead of copying it
rl_treeseq *treeseq_borrowing(int N, int L, const uint32_t *bits, int row_words, const double *rpos, const int *bp_pos,
                              const int *state, double theta);
int device_builder_reserve_shared(int device, int N);
int device_builder_expect_add(int device, int N, int delta, bool ages = false);
}  // namespace rl

using namespace rl;

struct rl_shard {
  rl_ctx *ctx = nullptr;
  std::string out_dir, base;
  int chunk = 0, sum_mode = RL_SUM_EXACT, device = 0;
  bool from_files = false;
  long long window_rows = 0;  // posterior rows a window keeps resident (0: all)
  std::vector<int> bp, state;
  struct Section {
    rl_window *win = nullptr;
    int last = 0;  // the cursors stand behind this SNP
    std::mutex m;
  };
  std::mutex m;  // guards `open`
  std::map<int, std::unique_ptr<Section>> open;
  int expected = 0;
};

static int read_ints(const std::string &fn, int L, std::vector<int> &v) {
  FILE *fp = fopen(fn.c_str(), "rb");
  if (!fp) {
    set_error("cannot open %s", fn.c_str());
    return RL_EIO;
  }
  int n = 0;
  v.assign((size_t)L, 0);
  const bool ok = fread(&n, 4, 1, fp) == 1 && n == L && fread(v.data(), 4, (size_t)L, fp) == (size_t)L;
  fclose(fp);
  if (!ok) {
    set_error("%s does not hold %d values", fn.c_str(), L);
    return RL_EFORMAT;
  }
  return RL_OK;
}

extern "C" {

rl_shard *rl_shard_open(const char *out_dir, int chunk_index, int k_begin, int k_end, int use_painting, double theta,
                        double rho, int sum_mode, int device, int from_paint_files) {
  if (!out_dir) {
    set_error("rl_shard_open: no directory");
    return nullptr;
  }
  rl_ctx *ctx = rl_create(device);
  if (!ctx) return nullptr;
  int rc = rl_load_chunk(ctx, out_dir, chunk_index);
  if (!rc && use_painting) rc = rl_set_painting(ctx, theta, rho);
  if (!rc) rc = rl_set_target_range(ctx, k_begin, k_end);
  std::unique_ptr<rl_shard> s(new rl_shard());
  s->out_dir = out_dir;
  s->chunk = chunk_index;
  s->sum_mode = sum_mode;
  s->device = device;
  s->from_files = from_paint_files != 0;
  const std::string c = std::to_string(chunk_index);
  if (!rc) rc = read_ints(s->out_dir + "/chunk_" + c + ".bp", ctx->L, s->bp);
  if (!rc) rc = read_ints(s->out_dir + "/chunk_" + c + ".state", ctx->L, s->state);
  if (!rc && !s->from_files) {
    // the shard's own stepping stones: painted here, kept in HBM (W * (k_end - k_begin) * N floats per direction),
    // quantised where they lie when a window re-paints from them (window.cpp)
    rc = rl_paint(ctx, sum_mode, nullptr);
    if (!rc) ctx->stones_disposable = true;
  }
  if (!rc) {  // (the Paint stage makes this directory for its files; the trees go there)
    const std::string cdir = s->out_dir + "/chunk_" + c;
    if (mkdir(cdir.c_str(), 0777) != 0 && errno != EEXIST) {
      set_error("cannot create %s", cdir.c_str());
      rc = RL_EIO;
    }
  }
  if (rc) {
    rl_destroy(ctx);
    return nullptr;
  }
  // output name = basename of -o (Relate.cpp:50-58)
  std::string base = s->out_dir;
  while (!base.empty() && base.back() == '/') base.pop_back();
  const size_t sl = base.find_last_of('/');
  s->base = sl == std::string::npos ? base : base.substr(sl + 1);
  s->ctx = ctx;
  return s.release();
}

void rl_shard_close(rl_shard *s) {
  if (!s) return;
  for (auto &kv : s->open)
    if (kv.second->win) rl_window_close(kv.second->win);
  s->open.clear();
  if (s->expected > 0) (void)device_builder_expect_add(s->device, s->ctx->N, -s->expected);
  rl_destroy(s->ctx);
  delete s;
}

int rl_shard_dims(const rl_shard *s, int *N, int *L, int *W, int *k_begin, int *k_end) {
  if (!s) return RL_EINVAL;
  if (N) *N = s->ctx->N;
  if (L) *L = s->ctx->L;
  if (W) *W = s->ctx->W;
  if (k_begin) *k_begin = s->ctx->k0;
  if (k_end) *k_end = s->ctx->k0 + s->ctx->nloc;
  return RL_OK;
}

int rl_shard_section_bounds(const rl_shard *s, int section, int *start, int *end) {
  if (!s || section < 0 || section >= s->ctx->W) {
    set_error("rl_shard_section_bounds: no section %d", section);
    return RL_EINVAL;
  }
  const rl_ctx *ctx = s->ctx;
  if (start) *start = ctx->wb[section];
  if (end) *end = std::min(ctx->L - 1, section < ctx->W - 1 ? ctx->wb[section + 1] - 1 : ctx->L - 1);
  return RL_OK;
}

int rl_shard_set_window_rows(rl_shard *s, long long rows) {
  if (!s || rows < 0) return RL_EINVAL;
  s->window_rows = rows;
  return RL_OK;
}

int rl_shard_rows(rl_shard *s, int section, int snp, void *d_rows) {
  if (!s || !d_rows) {
    set_error("rl_shard_rows: bad arguments");
    return RL_EINVAL;
  }
  int start = 0, end = 0;
  int rc = rl_shard_section_bounds(s, section, &start, &end);
  if (rc) return rc;
  if (snp < start || snp > end) {
    set_error("rl_shard_rows: SNP %d outside section %d (%d..%d)", snp, section, start, end);
    return RL_EINVAL;
  }
  rl_shard::Section *sec = nullptr;
  {
    std::lock_guard<std::mutex> lk(s->m);
    auto &slot = s->open[section];
    if (!slot) slot.reset(new rl_shard::Section());
    sec = slot.get();
  }
  std::lock_guard<std::mutex> lk(sec->m);
  if (!sec->win) {  // DistanceMeasure::GetTopologyWithRepaint for this shard's targets (anc_builder.cpp:49-106)
    const std::string pf = s->out_dir + "/chunk_" + std::to_string(s->chunk) + "/paint/relate_" + std::to_string(section) + ".bin";
    sec->win = rl_window_open_bounded(s->ctx, section, s->from_files ? pf.c_str() : nullptr, start, s->sum_mode,
                                      s->window_rows, nullptr);
    if (!sec->win) return RL_EIO;
    sec->last = start;
  }
  if (snp < sec->last) {  // (the tree builder only moves forward, anc_builder.cpp:487-495)
    set_error("rl_shard_rows: section %d has moved on to SNP %d, SNP %d is behind it", section, sec->last, snp);
    return RL_ESTATE;
  }
  for (int x = sec->last + 1; x <= snp; x++)
    if (x < end && (rc = rl_window_advance(sec->win, x))) return rc;  // (as rl_treeseq_build calls it: not for the last SNP)
  sec->last = snp;
  return rl_window_matrix_rows_device(sec->win, snp, d_rows, nullptr);
}

int rl_shard_release_section(rl_shard *s, int section) {
  if (!s) return RL_EINVAL;
  std::unique_ptr<rl_shard::Section> sec;
  {
    std::lock_guard<std::mutex> lk(s->m);
    auto it = s->open.find(section);
    if (it == s->open.end()) return RL_OK;
    sec = std::move(it->second);
    s->open.erase(it);
  }
  std::lock_guard<std::mutex> lk(sec->m);
  if (sec->win) rl_window_close(sec->win);
  sec->win = nullptr;
  return RL_OK;
}

int rl_shard_expect_builders(rl_shard *s, int builders) {
  if (!s || builders < 0) return RL_EINVAL;
  if (builders > 0 && device_builder_reserve_shared(s->device, s->ctx->N)) return RL_ENOMEM;
  // (added to what the other shards of this process on this device ask for: two target ranges as threads of one
  //  process each asking for 4 workers used to leave 4 for their 8 trees -- the later word won)
  const int delta = builders - s->expected;
  s->expected = builders;
  return device_builder_expect_add(s->device, s->ctx->N, delta) ? RL_EHIP : RL_OK;
}

int rl_shard_build_section(rl_shard *s, int section, int flags, int fb, int build_device, rl_matrix_fn matrix,
                           rl_matrix_dev_fn matrix_dev, void *user, int *num_trees) {
  if (!s || !matrix) {
    set_error("rl_shard_build_section: bad arguments");
    return RL_EINVAL;
  }
  int start = 0, end = 0;
  int rc = rl_shard_section_bounds(s, section, &start, &end);
  if (rc) return rc;
  const rl_ctx *ctx = s->ctx;
  rl_treeseq *ts = treeseq_borrowing(ctx->N, ctx->L, ctx->bits.data(), ctx->row_words, ctx->rpos.data(), s->bp.data(),
                                     s->state.data(), ctx->theta);
  if (!ts) return RL_EINVAL;
  if (build_device >= 0) {
    rl_treeseq_set_build_device(ts, build_device);
    if (matrix_dev) rl_treeseq_set_device_matrix(ts, matrix_dev);
  }
  // (no `advance` callback: every shard replays the cursors itself, rl_shard_rows)
  rc = rl_treeseq_build(ts, start, end, matrix, nullptr, user, flags, fb);
  if (!rc) {
    const std::string b = s->out_dir + "/chunk_" + std::to_string(s->chunk) + "/" + s->base + "_" + std::to_string(section);
    rc = rl_treeseq_write(ts, (b + ".anc").c_str(), (b + ".mut").c_str());
  }
  if (!rc && num_trees) *num_trees = rl_treeseq_num_trees(ts);
  rl_treeseq_destroy(ts);
  return rc;
}

int rl_device_copy(void *dst, const void *src, size_t bytes, int device) {
  if (!dst || !src) return RL_EINVAL;
  RL_HIP(hipSetDevice(device));
  // (a device-to-device hipMemcpy may return before the copy has run, and the builder's kernels are on non-blocking
  //  streams that do not wait for the null stream: the copy is complete when this returns)
  RL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, nullptr));
  RL_HIP(hipStreamSynchronize(nullptr));
  return RL_OK;
}

}  // extern "C"
