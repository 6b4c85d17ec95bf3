// window.cpp -- rl_window: DistanceMeasure for one window, topology in HBM.
//
// Host counterpart of DistanceMeasure (anc_builder.hpp:50-109):
//   rl_window_open    = GetTopologyWithRepaint   (anc_builder.cpp:49-106)
//   rl_window_advance = the cursor update of AncesTreeBuilder::BuildTopology
//                       (anc_builder.cpp:487-495)
//   rl_window_matrix  = GetMatrix                (anc_builder.cpp:109-207)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>

#include "common.h"

using namespace rl;

struct rl_window {
  rl_ctx *ctx = nullptr;
  int w = 0;
  int start = 0, end = 0;  // section_startpos / section_endpos as stored in the paint file
  int k0 = 0, nloc = 0;          // targets of the context when the window was opened; arrays below are [nloc]
  std::vector<int64_t> top_off;  // [nloc+1]
  std::vector<int64_t> ck_off;   // [nloc+1] checkpoint rows of the targets' forward passes (repaint_kernels.hip)
  std::vector<float> logscales;  // host copy, [sum D]
  std::vector<int32_t> v_snp_prev;
  std::vector<double> v_rpos_prev, v_rpos_next;
  DevBuf d_top, d_ls, d_top_off, d_matrix;
  DevBuf d_vsp, d_direct, d_wl, d_wr, d_epn, d_enp;
  // What RePaint needs to run again (a bounded window keeps part of its posterior rows and recomputes as the
  // tree builder moves on): the decoded stones, the plan slices, the last-interval coefficients.
  DevBuf d_ab, d_be, d_la, d_lb, d_ib, d_ie, d_cfl, d_nxl, d_order, d_ck_off;
  // the window's stones: its own decoded copy (d_ab ..), or the context's slice quantised in place (fused stage)
  const float *ab = nullptr, *be = nullptr, *la = nullptr, *lb = nullptr;
  // a bounded window's ONE kept state of the backward pass per target (repaint_kernels.hip): the row it stands
  // before (-1: none), and what the next launch does with it
  DevBuf d_bstate, d_bscal, d_fstate, d_fscal;
  std::vector<int32_t> f_row, fstart_row, fsave_row;  // the forward pass's kept state likewise
  // the per-launch arrays (slab_off, slab_base: int64; row_lo, row_hi, start_row, save_row: int32; [nloc] each) go to
  // the device in ONE copy from a pinned block, on RePaint's stream and in its turn: six blocking copies from
  // pageable memory took 50-180 ms per launch with a hundred sections copying
  DevBuf d_place;
  int64_t ck_rows_launch = 0;  // checkpoint rows the strips of the next launch hold (compact for a partial launch)
  unsigned char *h_place = nullptr;
  size_t h_place_bytes = 0;
  std::vector<int32_t> b_row, start_row, save_row;  // [nloc]
  int anchor_snp = -1;  // the SNP the kept backward states stand above (the same for all targets), -1: none
  // RELATE_AMD_TIMING: where the window's wall-clock goes (seconds): choosing rows + uploads, waiting for the context's
  // RePaint turn, the launch until it is through, the matrices' own part (arguments, kernel, wait)
  double t_place = 0, t_turn = 0, t_launch = 0, t_matrix = 0;
  long long n_matrices = 0;
  std::vector<int64_t> slab_off, slab_base;  // [nloc]
  std::vector<int32_t> row_lo, row_hi;       // [nloc] resident posterior rows [lo, hi) of each target
  int64_t cap_rows = 0;                      // rows d_top holds; >= all rows: the whole window is resident
  int maxD = 0, sum_mode = 0, repaints = 0;
  bool have_logscales = false;
  // a window's distance matrices run on its own stream: the sections of a stage ask for theirs at the same time
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e2 = nullptr;
  bool matrix_ready = false;  // stream, events, argument block and cursor caches of rl_window_matrix all exist
  unsigned char *h_stage = nullptr;  // pinned: the per-target arguments of one matrix (MatrixArg[nloc]), read by the
  size_t h_stage_bytes = 0;          // kernel where they lie (d_args: the block's device address)
  void *d_args = nullptr;
  DevBuf d_stage;                    // ... and where a small kernel puts them for the matrix kernel
  std::vector<int32_t> e_cursor;     // [nloc] cursor position e_pn / e_np were computed for
  std::vector<float> e_pn, e_np;
  ~rl_window() {
    if (getenv("RELATE_AMD_TIMING") && n_matrices > 0)
      fprintf(stderr, "[window %d] %lld matrices, %d RePaint launches; s: rows + uploads %.2f, waiting for RePaint's turn %.2f, "
              "RePaint launches %.2f, matrices %.2f\n", w, n_matrices, repaints, t_place, t_turn, t_launch, t_matrix);
    if (stream) (void)hipStreamDestroy(stream);
    if (e0) (void)hipEventDestroy(e0);
    if (e2) (void)hipEventDestroy(e2);
    if (h_stage) pinned_cache_release(h_stage, h_stage_bytes);
    if (h_place) pinned_cache_release(h_place, h_place_bytes);
  }
};

static inline bool derived(const rl_ctx *ctx, int snp, int n) {
  return (ctx->bits[(size_t)snp * ctx->row_words + (n >> 5)] >> (n & 31)) & 1u;
}

// RePaintSection for all targets of the window, keeping the posterior rows [row_lo, row_hi) (already uploaded).
static int repaint_rows(rl_window *win, float *kernel_ms) {
  rl_ctx *ctx = win->ctx;
  const int N = ctx->N, S = ctx->S, waves = ctx->waves, nloc = win->nloc;
  // the checkpoint rows and side records of the forward kernel are scratch of the launch: one buffer per context,
  // launches are serialised
  const int64_t ck_doubles = win->ck_rows_launch * (int64_t)S * 64 * waves;
  const size_t scratch_bytes = (size_t)(ck_doubles + win->top_off[nloc] * REPAINT_SIDE) * sizeof(double);
  // (the second lane's strips are sized for partial launches: a window's first, whole pass takes the first lane)
  const bool first_lane_only = !win->have_logscales || scratch_bytes > ctx->lane2.scratch.bytes;
  const auto t_ask = std::chrono::steady_clock::now();
  // (whichever lane is free; with both taken, the windows queue up behind the two in turn)
  static std::atomic<unsigned> turn{0};
  std::unique_lock<std::mutex> one_at_a_time(ctx->repaint_mutex, std::defer_lock);
  bool second = false;
  if (!ctx->two_lanes || first_lane_only) {
    one_at_a_time.lock();
  } else if (!one_at_a_time.try_lock()) {
    std::unique_lock<std::mutex> other(ctx->lane2.m, std::try_to_lock);
    if (!other.owns_lock()) {
      if (turn.fetch_add(1) & 1u)
        other.lock();
      else
        one_at_a_time.lock();
    }
    if (other.owns_lock()) {
      one_at_a_time = std::move(other);
      second = true;
    }
  }
  hipStream_t stream = second ? ctx->lane2.s : ctx->s0;
  hipEvent_t e0 = second ? ctx->lane2.e0 : ctx->ev0, e1 = second ? ctx->lane2.e1 : ctx->ev2;
  rl::DevBuf &scratch = second ? ctx->lane2.scratch : ctx->d_k2_scratch;
  const auto t_got = std::chrono::steady_clock::now();
  win->t_turn += std::chrono::duration<double>(t_got - t_ask).count();
  int rc = scratch.alloc(scratch_bytes);
  if (rc) return rc;
  RepaintParams p;
  p.lay = ctx->lay;
  p.c = ctx->consts;
  p.L = ctx->L;
  p.k0 = win->k0;
  p.nloc = nloc;
  p.masks = ctx->d_masks.as<unsigned long long>();
  p.plan_off = ctx->d_off.as<int64_t>();
  p.sites = ctx->d_sites.as<int32_t>();
  p.cf = ctx->d_cf.as<double>();
  p.nxt = ctx->d_nxt.as<double>();
  p.ib = win->d_ib.as<int32_t>();
  p.ie = win->d_ie.as<int32_t>();
  p.cf_last = win->d_cfl.as<double>();
  p.nxt_last = win->d_nxl.as<double>();
  p.alpha_begin = win->ab;
  p.beta_end = win->be;
  p.ls_alpha = win->la;
  p.ls_beta = win->lb;
  p.top_off = win->d_top_off.as<int64_t>();
  {
    const unsigned char *dp = win->d_place.as<unsigned char>();
    p.slab_off = reinterpret_cast<const int64_t *>(dp);
    p.row_lo = reinterpret_cast<const int32_t *>(dp + (size_t)nloc * 16);
    p.row_hi = reinterpret_cast<const int32_t *>(dp + (size_t)nloc * 20);
    p.start_row = reinterpret_cast<const int32_t *>(dp + (size_t)nloc * 24);
    p.save_row = reinterpret_cast<const int32_t *>(dp + (size_t)nloc * 28);
    p.fstart_row = reinterpret_cast<const int32_t *>(dp + (size_t)nloc * 32);
    p.fsave_row = reinterpret_cast<const int32_t *>(dp + (size_t)nloc * 36);
  }
  p.topology = win->d_top.as<float>();
  p.logscales = win->d_ls.as<float>();
  p.scratch = scratch.as<double>();
  p.ck_off = reinterpret_cast<const int64_t *>(win->d_place.as<unsigned char>() + (size_t)nloc * 40);
  p.side = p.scratch + ck_doubles;
  p.order = win->d_order.as<int32_t>();
  p.sum_mode = win->sum_mode;
  p.partial = win->have_logscales ? 1 : 0;
  // The part launches of a bounded window run their backward kernel WITHOUT the LDS strip (repaint_kernels.hip:
  // HeldRow NOSTRIP): a part is ~6 rows per target behind ~40 beta-only steps of descent, the checkpoint rows are read
  // where they lie (through the L2), and two waves share a SIMD.  C3, 116 workers, same boxes, alternating: 147.6 /
  // 147.1 / 144.6 s against 161.6 / 148.0 / 150.4 s with the strip; RePaint 57-68 s on the device instead of 77-120,
  // a section waits 1-2 s per window for its turn (profiles/r05_c3_runs.json).  A WHOLE window keeps the strip: every
  // one of a block's six rows would re-read the checkpoint row, 15.7 ms against 14.9 ms (profiles/r06_k2_whole_window_strip.json).
  p.nostrip = 1;
  p.bstate = win->d_bstate.as<double>();
  p.bscal = win->d_bscal.as<double>();
  p.fstate = win->d_fstate.as<double>();
  p.fscal = win->d_fscal.as<double>();
  (void)N;
  bool ok = hipMemcpyAsync(win->d_place.p, win->h_place, (size_t)nloc * 48, hipMemcpyHostToDevice, stream) == hipSuccess;
  ok = ok && hipEventRecord(e0, stream) == hipSuccess;
  hipError_t le = ok ? launch_repaint(p, S, waves, stream) : hipErrorUnknown;
  ok = ok && le == hipSuccess;
  ok = ok && hipEventRecord(e1, stream) == hipSuccess;
  hipError_t se = hipEventSynchronize(e1);
  ok = ok && se == hipSuccess;
  if (!ok) {
    set_error("repaint launch failed: %s / %s", hipGetErrorString(le), hipGetErrorString(se));
    return RL_EHIP;
  }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (kernel_ms) *kernel_ms = ms;
  ctx->repaint_us += (long long)(1e3 * ms);
  ctx->repaint_launches++;
  one_at_a_time.unlock();
  win->t_launch += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_got).count();
  win->repaints++;
  for (size_t t = 0; t < win->save_row.size(); t++) {
    if (win->save_row[t] >= 0) win->b_row[t] = win->save_row[t];
    if (win->fsave_row[t] >= 0) win->f_row[t] = win->fsave_row[t];
  }
  if (!win->have_logscales) {  // (every launch writes all of them, with the same values)
    const int64_t rows = win->top_off[nloc];
    win->logscales.resize((size_t)rows);
    if (hipMemcpy(win->logscales.data(), win->d_ls.p, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost) !=
        hipSuccess) {
      set_error("copy of logscales failed");
      return RL_EHIP;
    }
    win->have_logscales = true;
  }
  return RL_OK;
}

// Choose the resident rows from the cursors at `snp` onwards and recompute them.  A tree at SNP s reads rows
// v_snp_prev and v_snp_prev + 1 of every target, and v_snp_prev grows by one at each of the target's derived
// sites: the rows from the cursor to two past the derived sites in [snp, last] cover all trees up to `last`,
// which is taken as far as the capacity allows.
static int place_rows(rl_window *win, int snp, float *kernel_ms) {
  rl_ctx *ctx = win->ctx;
  const auto t_in = std::chrono::steady_clock::now();
  const int nloc = win->nloc, k0 = win->k0, L = ctx->L;
  const int64_t all_rows = win->top_off[nloc];
  int part_last = L - 1;            // the last SNP whose trees the part serves
  std::vector<int32_t> part_extra;  // derived sites of each target in [snp, part_last]
  if (win->cap_rows >= all_rows) {
    for (int t = 0; t < nloc; t++) {
      win->row_lo[t] = 0;
      win->row_hi[t] = (int32_t)(win->top_off[t + 1] - win->top_off[t]);
    }
  } else {
    std::vector<int32_t> extra(nloc, 0);
    part_last = std::max(snp, 0);
    int64_t used = 2 * (int64_t)nloc;
    const int last_wanted = std::min(L - 1, win->end + 1);
    // The sections of a stage open together and would come back for their next part together, every one of them
    // waiting for a hundred RePaint launches ahead of it: the FIRST part of window w is cut short by a fraction that
    // differs from window to window (golden-ratio sequence), which spreads the later requests over the period.
    int64_t cap_now = win->cap_rows;
    if (!win->have_logscales) {
      const double frac = std::fmod(0.6180339887498949 * (win->w + 1), 1.0);
      cap_now = std::max<int64_t>(3 * (int64_t)nloc + 64, (int64_t)((0.15 + 0.85 * frac) * (double)win->cap_rows));
    }
    for (int s0 = std::max(snp, 0); s0 <= last_wanted; s0++) {
      int64_t pop = 0;
      for (int t = 0; t < nloc; t++) pop += derived(ctx, s0, k0 + t);
      if (used + pop > cap_now && s0 > snp) break;
      used += pop;
      part_last = s0;
      for (int t = 0; t < nloc; t++) extra[t] += derived(ctx, s0, k0 + t);
    }
    part_extra = extra;
    for (int t = 0; t < nloc; t++) {
      const int D = (int)(win->top_off[t + 1] - win->top_off[t]);
      win->row_lo[t] = std::min(std::max(win->v_snp_prev[t], 0), D);
      win->row_hi[t] = (int32_t)std::min<int64_t>(D, (int64_t)win->row_lo[t] + extra[t] + 2);
    }
  }
  int64_t at = 0;
  for (int t = 0; t < nloc; t++) {
    win->slab_off[t] = at;
    win->slab_base[t] = at - win->row_lo[t];
    at += win->row_hi[t] - win->row_lo[t];
  }
  if (at > std::max(win->cap_rows, (int64_t)1) && win->cap_rows < all_rows) {
    set_error("window %d: %lld posterior rows do not fit the %lld allowed", win->w, (long long)at,
              (long long)win->cap_rows);
    return RL_ENOMEM;
  }
  int rc = win->d_place.alloc((size_t)nloc * 48);
  if (!rc && !win->h_place && !(win->h_place = static_cast<unsigned char *>(pinned_cache_alloc((size_t)nloc * 48, &win->h_place_bytes)))) {
    set_error("window %d: no pinned host memory for the launch arguments", win->w);
    tl_alloc_failures++;
    rc = RL_ENOMEM;
  }
  if (rc) return rc;
  const bool keep_state = win->cap_rows < all_rows;
  win->start_row.assign(nloc, -1);
  win->save_row.assign(nloc, -1);
  win->fstart_row.assign(nloc, -1);
  win->fsave_row.assign(nloc, -1);
  if (win->f_row.empty()) win->f_row.assign(nloc, -1);
  if (keep_state) {
    // The backward pass of a launch runs from the window's last row down to the part's first: half a window on
    // average, however small the part.  One kept state per target cuts that: a launch that has to come down from
    // the last row leaves the states above an ANCHOR SNP halfway between the end of its part and the end of the
    // window, and the launches of the parts that end at or before the anchor start from them.  The anchor is a SNP,
    // not a row, so that ALL targets switch in the same launches (a target's state stands before the row its part
    // would end at if the part ended at the anchor): a launch is as long as its longest descent, and with a midpoint
    // per target a few targets came down from the last row in most launches.  24 parts: 132 part-lengths of
    // descent instead of 300, and four or five long launches per window instead of a dozen.
    const size_t row_doubles = (size_t)ctx->S * 64 * ctx->waves;
    rc = win->d_bstate.alloc((size_t)nloc * row_doubles * sizeof(double));
    rc = rc ? rc : win->d_bscal.alloc((size_t)nloc * 2 * sizeof(double));
    if (rc) return rc;
    if (win->b_row.empty()) win->b_row.assign(nloc, -1);
    const int last_snp = std::min(L - 1, win->end + 1);
    if (win->have_logscales && win->anchor_snp >= part_last) {
      for (int t = 0; t < nloc; t++) {
        const int D = (int)(win->top_off[t + 1] - win->top_off[t]);
        if (win->b_row[t] >= 0 && win->b_row[t] >= win->row_hi[t] - 1 && win->b_row[t] <= D - 2)
          win->start_row[t] = win->b_row[t];  // (else: this target from the stone)
      }
    } else if (last_snp - part_last >= 24) {
      // (Tried in round 4: the anchor 1.2 sqrt(parts left) parts above this one instead of halfway -- by a count of
      //  part-lengths, tools/exp/anchor_rule.py, 177 instead of 252 for 37 parts -- made the stage SLOWER: 15 ms per
      //  launch instead of 10, 172.7 s instead of 159.4 s at 108 workers.  The count is not the cost.)
      const int anchor = part_last + (last_snp - part_last) / 2;
      std::vector<int32_t> more(nloc, 0);
      for (int s0 = part_last + 1; s0 <= anchor; s0++)
        for (int t = 0; t < nloc; t++) more[t] += derived(ctx, s0, k0 + t);
      for (int t = 0; t < nloc; t++) {
        const int D = (int)(win->top_off[t + 1] - win->top_off[t]);
        // the row a part from here to the anchor would end at, and one more (a later part counts its first SNP twice)
        const int r = win->row_lo[t] + part_extra[t] + more[t] + 3;
        if (r <= D - 2 && r > win->row_hi[t]) win->save_row[t] = r;
      }
      win->anchor_snp = anchor;
      for (int t = 0; t < nloc; t++)
        if (win->save_row[t] < 0) win->b_row[t] = -1;  // (no state for this target under the new anchor)
    }
    // The forward pass likewise ran from the window's first row up to the part's last, every time: the state behind
    // the checkpoint row in whose block the NEXT part begins (its first row is this part's last but one) is kept, and
    // the next launch starts from it -- a part's own rows instead of half a window (half of RePaint's work with the
    // backward state in place).
    constexpr int CK = REPAINT_CHECKPOINT;
    rc = win->d_fstate.alloc((size_t)nloc * row_doubles * sizeof(double));
    rc = rc ? rc : win->d_fscal.alloc((size_t)nloc * 4 * sizeof(double));
    if (rc) return rc;
    for (int t = 0; t < nloc; t++) {
      const int lo_blk = win->row_lo[t] - win->row_lo[t] % CK;
      if (win->have_logscales && win->f_row[t] > 0 && win->f_row[t] <= lo_blk) win->fstart_row[t] = win->f_row[t];
      const int next_lo = std::max(0, win->row_hi[t] - 2), f = next_lo - next_lo % CK;
      if (f > std::max(0, win->fstart_row[t])) win->fsave_row[t] = f;
    }
  }
  {  // (the block is rewritten only after the launch that read it is through: repaint_rows waits for its launch)
    unsigned char *hp = win->h_place;
    memcpy(hp, win->slab_off.data(), (size_t)nloc * 8);
    memcpy(hp + (size_t)nloc * 8, win->slab_base.data(), (size_t)nloc * 8);
    memcpy(hp + (size_t)nloc * 16, win->row_lo.data(), (size_t)nloc * 4);
    memcpy(hp + (size_t)nloc * 20, win->row_hi.data(), (size_t)nloc * 4);
    memcpy(hp + (size_t)nloc * 24, win->start_row.data(), (size_t)nloc * 4);
    memcpy(hp + (size_t)nloc * 28, win->save_row.data(), (size_t)nloc * 4);
    memcpy(hp + (size_t)nloc * 32, win->fstart_row.data(), (size_t)nloc * 4);
    memcpy(hp + (size_t)nloc * 36, win->fsave_row.data(), (size_t)nloc * 4);
    // Where target t's checkpoint rows lie in the launch's strips.  The first launch of a window makes ALL rows'
    // logscales and checkpoints: the window-wide offsets (7 GB of strips at C3).  A later, partial launch writes and
    // reads only the blocks its kept rows [row_lo, row_hi) are rebuilt from: those are packed target after target --
    // ck_off'[t] + block = (blocks before t) - block(row_lo[t]) + block --, (kept rows / 6 + 2 per target) rows of
    // doubles, 0.6 GB at C3: what lets a SECOND lane of RePaint launches cost no sections (round 4).
    int64_t *ck = reinterpret_cast<int64_t *>(hp + (size_t)nloc * 40);
    constexpr int CKR = REPAINT_CHECKPOINT;
    if (!win->have_logscales) {
      for (int t = 0; t < nloc; t++) ck[t] = win->ck_off[t];
      win->ck_rows_launch = win->ck_off[nloc];
    } else {
      int64_t before = 0;
      for (int t = 0; t < nloc; t++) {
        const int b0 = win->row_lo[t] / CKR, b1 = std::max(win->row_hi[t] - 1, win->row_lo[t]) / CKR;
        ck[t] = before - b0;
        before += b1 - b0 + 1;
      }
      win->ck_rows_launch = before;
    }
  }
  win->t_place += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count();
  return rc ? rc : repaint_rows(win, kernel_ms);
}


static int read_file(const char *fn, std::vector<unsigned char> &buf) {
  FILE *fp = fopen(fn, "rb");
  if (!fp) {
    set_error("cannot open %s", fn);
    return RL_EIO;
  }
  fseek(fp, 0, SEEK_END);
  long len = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  buf.resize((size_t)len);
  size_t got = len ? fread(buf.data(), 1, (size_t)len, fp) : 0;
  fclose(fp);
  if (got != (size_t)len) {
    set_error("short read on %s", fn);
    return RL_EIO;
  }
  return RL_OK;
}

extern "C" {

rl_window *rl_window_open(rl_ctx *ctx, int w, const char *paint_file, int first_snp, int sum_mode,
                          float *kernel_ms) {
  return rl_window_open_bounded(ctx, w, paint_file, first_snp, sum_mode, 0, kernel_ms);
}

rl_window *rl_window_open_bounded(rl_ctx *ctx, int w, const char *paint_file, int first_snp, int sum_mode,
                                  long long max_rows, float *kernel_ms) {
  if (!ctx || !ctx->have_chunk || w < 0 || w >= ctx->W) {
    set_error("rl_window_open: bad arguments");
    return nullptr;
  }
  if (sum_mode != RL_SUM_EXACT && sum_mode != RL_SUM_LANES && sum_mode != RL_SUM_EXACT_SERIAL && sum_mode != RL_SUM_LANES32) {
    set_error("rl_window_open: bad sum_mode");
    return nullptr;
  }
  if (hipSetDevice(ctx->device) != hipSuccess) {
    set_error("hipSetDevice failed");
    return nullptr;
  }
  if (upload_plan(ctx)) return nullptr;
  const int N = ctx->N, W = ctx->W, S = ctx->S, waves = ctx->waves;
  const int k0 = ctx->k0, nloc = ctx->nloc;  // this context's targets; t = n - k0 indexes the per-target arrays
  const Plan &pl = ctx->plan;

  // ---- stepping stones of this window, after the file's float/RLE quantisation
  std::vector<float> ab((size_t)nloc * N), be((size_t)nloc * N), la(nloc), lb(nloc);
  std::vector<int> bb(nloc), bend(nloc);
  int start = ctx->wb[w], end = ctx->wb[w + 1] - 1;
  if (paint_file) {
    std::vector<unsigned char> buf;
    if (read_file(paint_file, buf)) return nullptr;
    size_t pos = 0;
    std::vector<float> skip(N);
    for (int n = 0; n < k0 + nloc; n++) {  // anc_builder.cpp:61-73; records of other contexts' targets are skipped
      if (pos + 8 > buf.size()) {
        set_error("%s: truncated at target %d", paint_file, n);
        return nullptr;
      }
      memcpy(&start, buf.data() + pos, 4);
      memcpy(&end, buf.data() + pos + 4, 4);
      pos += 8;
      const bool mine = n >= k0;
      const int t = mine ? n - k0 : 0;
      int bs = 0;
      float ls = 0.f;
      size_t u = decode_stone(buf.data() + pos, buf.size() - pos, N, mine ? &ab[(size_t)t * N] : skip.data(),
                              mine ? &bb[t] : &bs, mine ? &la[t] : &ls);
      if (!u) {
        set_error("%s: malformed alpha record of target %d", paint_file, n);
        return nullptr;
      }
      pos += u;
      u = decode_stone(buf.data() + pos, buf.size() - pos, N, mine ? &be[(size_t)t * N] : skip.data(),
                       mine ? &bend[t] : &bs, mine ? &lb[t] : &ls);
      if (!u) {
        set_error("%s: malformed beta record of target %d", paint_file, n);
        return nullptr;
      }
      pos += u;
    }
  } else {
    if (!ctx->painted) {
      set_error("rl_window_open: no paint file given and rl_paint has not run");
      return nullptr;
    }
    // the stones stay in HBM: copied and run through the file's quantisation on the device below
    for (int t = 0; t < nloc; t++) {
      bb[t] = pl.bb[(size_t)(k0 + t) * W + w];
      bend[t] = pl.be[(size_t)(k0 + t) * W + w];
    }
  }

  // ---- per-target slices of the visited-site plan
  std::vector<int32_t> ib(nloc), ie(nloc);
  std::vector<double> cf_last(nloc), nxt_last(nloc);
  std::vector<double> r(ctx->r);
  if (ctx->rho != 1.0)
    for (auto &x : r) x *= ctx->rho;
  rl_window *win = new rl_window();
  win->ctx = ctx;
  win->w = w;
  win->start = start;
  win->end = end;
  win->k0 = k0;
  win->nloc = nloc;
  win->top_off.assign((size_t)nloc + 1, 0);
  win->ck_off.assign((size_t)nloc + 1, 0);
  int maxD = 0;
  for (int t = 0; t < nloc; t++) {
    const int n = k0 + t;
    ib[t] = pl.ia[(size_t)n * W + w];
    ie[t] = pl.ie[(size_t)n * W + w];
    if (pl.bb[(size_t)n * W + w] != bb[t] || pl.be[(size_t)n * W + w] != bend[t]) {
      set_error("paint file boundary SNPs of target %d (%d,%d) disagree with the chunk (%d,%d)", n, bb[t],
                bend[t], pl.bb[(size_t)n * W + w], pl.be[(size_t)n * W + w]);
      delete win;
      return nullptr;
    }
    const int D = ie[t] - ib[t] + 1;
    maxD = std::max(maxD, D);
    win->top_off[t + 1] = win->top_off[t] + D;
    win->ck_off[t + 1] = win->ck_off[t] + (D + REPAINT_CHECKPOINT - 1) / REPAINT_CHECKPOINT;
    // last interval of RePaintSection: r[last_snp] only (fast_painting.cpp:702-716)
    interval_coeffs(ctx->consts, N, r[bend[t]], &cf_last[t], &nxt_last[t]);
  }
  const int64_t rows = win->top_off[nloc];
  std::vector<int32_t> order(nloc);  // targets (global index), longest slice first
  std::iota(order.begin(), order.end(), k0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    return (ie[a - k0] - ib[a - k0]) > (ie[b - k0] - ib[b - k0]);
  });

  win->maxD = maxD;
  win->sum_mode = sum_mode;
  win->cap_rows = (max_rows > 0 && max_rows < rows) ? std::max<int64_t>(max_rows, 3 * (int64_t)nloc + 64) : rows;
  win->slab_off.assign(nloc, 0);
  win->slab_base.assign(nloc, 0);
  win->row_lo.assign(nloc, 0);
  win->row_hi.assign(nloc, 0);
  int rc = 0;
  if (paint_file) {
    rc = rc ? rc : win->d_ab.upload(ab);
    rc = rc ? rc : win->d_be.upload(be);
    rc = rc ? rc : win->d_la.upload(la);
    rc = rc ? rc : win->d_lb.upload(lb);
  } else if (ctx->stones_disposable && !ctx->h_alpha) {
    // decode(encode(x)): the paint file's round trip is part of the numerics -- floats, and runs of
    // nearly equal values replaced by their first (collapsed_matrix.hpp:228-296) -- applied to the context's slice
    // where it lies, once
    const size_t sn = (size_t)nloc * N;
    float *ca = ctx->d_alpha.as<float>() + (size_t)w * sn, *cb = ctx->d_beta.as<float>() + (size_t)w * sn;
    std::lock_guard<std::mutex> s0_is_mine(ctx->repaint_mutex);
    if (ctx->stone_quantised.size() != (size_t)W) ctx->stone_quantised.assign((size_t)W, 0);
    if (!ctx->stone_quantised[w]) {
      if (launch_quantise(ca, nloc, N, ctx->s0) != hipSuccess || launch_quantise(cb, nloc, N, ctx->s0) != hipSuccess ||
          hipStreamSynchronize(ctx->s0) != hipSuccess) {
        set_error("stone quantisation on the device failed");
        rc = RL_EHIP;
      }
      ctx->stone_quantised[w] = 1;
    }
    win->ab = ca;
    win->be = cb;
    win->la = ctx->d_lsa.as<float>() + (size_t)w * nloc;
    win->lb = ctx->d_lsb.as<float>() + (size_t)w * nloc;
  } else {
    // the same on a copy: the context's stones stay as painted (rl_write_paint_files may still want them)
    const size_t sn = (size_t)nloc * N;
    rc = rc ? rc : win->d_ab.alloc(sn * 4);
    rc = rc ? rc : win->d_be.alloc(sn * 4);
    rc = rc ? rc : win->d_la.alloc((size_t)nloc * 4);
    rc = rc ? rc : win->d_lb.alloc((size_t)nloc * 4);
    std::lock_guard<std::mutex> s0_is_mine(ctx->repaint_mutex);
    auto d2d = [&](void *dst, const float *src, size_t n) {
      return hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, ctx->s0) == hipSuccess ? 0 : RL_EHIP;
    };
    auto h2d = [&](void *dst, const float *src, size_t n) {
      return hipMemcpyAsync(dst, src, n * 4, hipMemcpyHostToDevice, ctx->s0) == hipSuccess ? 0 : RL_EHIP;
    };
    if (ctx->h_alpha) {  // (parked on the host by the fused stage)
      rc = rc ? rc : h2d(win->d_ab.p, ctx->h_alpha + (size_t)w * sn, sn);
      rc = rc ? rc : h2d(win->d_be.p, ctx->h_beta + (size_t)w * sn, sn);
    } else {
      rc = rc ? rc : d2d(win->d_ab.p, ctx->d_alpha.as<float>() + (size_t)w * sn, sn);
      rc = rc ? rc : d2d(win->d_be.p, ctx->d_beta.as<float>() + (size_t)w * sn, sn);
    }
    rc = rc ? rc : d2d(win->d_la.p, ctx->d_lsa.as<float>() + (size_t)w * nloc, nloc);
    rc = rc ? rc : d2d(win->d_lb.p, ctx->d_lsb.as<float>() + (size_t)w * nloc, nloc);
    if (!rc && (launch_quantise(win->d_ab.as<float>(), nloc, N, ctx->s0) != hipSuccess ||
                launch_quantise(win->d_be.as<float>(), nloc, N, ctx->s0) != hipSuccess ||
                hipStreamSynchronize(ctx->s0) != hipSuccess)) {
      set_error("stone quantisation on the device failed");
      rc = RL_EHIP;
    }
  }
  if (!win->ab) {
    win->ab = win->d_ab.as<float>();
    win->be = win->d_be.as<float>();
    win->la = win->d_la.as<float>();
    win->lb = win->d_lb.as<float>();
  }
  rc = rc ? rc : win->d_ib.upload(ib);
  rc = rc ? rc : win->d_ie.upload(ie);
  rc = rc ? rc : win->d_cfl.upload(cf_last);
  rc = rc ? rc : win->d_nxl.upload(nxt_last);
  rc = rc ? rc : win->d_order.upload(order);
  rc = rc ? rc : win->d_top_off.upload(win->top_off);
  rc = rc ? rc : win->d_ck_off.upload(win->ck_off);
  rc = rc ? rc : win->d_top.alloc((size_t)std::min(rows, win->cap_rows) * S * 64 * waves * sizeof(float));
  rc = rc ? rc : win->d_ls.alloc((size_t)rows * sizeof(float));
  if (rc) {  // (d_matrix: when a matrix is first asked for into host memory)
    delete win;
    return nullptr;
  }

  // ---- cursors (anc_builder.cpp:81-101)
  const int snp = first_snp < 0 ? ctx->wb[w] : first_snp;
  win->v_snp_prev.assign(nloc, 0);
  win->v_rpos_prev.assign(nloc, 0.0);
  win->v_rpos_next.assign(nloc, 0.0);
  if (snp > 0) {
    for (int s = snp; s >= win->start; s--)
      for (int t = 0; t < nloc; t++)
        if (derived(ctx, s, k0 + t)) win->v_snp_prev[t]++;
  }
  for (int t = 0; t < nloc; t++) {
    int s = snp;
    while (!derived(ctx, s, k0 + t) && s > 0) s--;
    win->v_rpos_prev[t] = ctx->rpos[s];
    win->v_rpos_next[t] = win->v_rpos_prev[t];
  }
  if (place_rows(win, snp, kernel_ms)) {  // RePaintSection (anc_builder.cpp:75-78)
    delete win;
    return nullptr;
  }
  return win;
}

void rl_window_close(rl_window *win) { delete win; }

int rl_window_repaints(const rl_window *win) { return win ? win->repaints : RL_EINVAL; }

int rl_window_bounds(const rl_window *win, int *start, int *end) {
  if (!win) return RL_EINVAL;
  if (start) *start = win->start;
  if (end) *end = win->end;
  return RL_OK;
}

int rl_window_rows(const rl_window *win, int n) {
  if (!win || n < win->k0 || n >= win->k0 + win->nloc) return RL_EINVAL;
  const int t = n - win->k0;
  return (int)(win->top_off[t + 1] - win->top_off[t]);
}

int rl_window_get_topology(rl_window *win, int n, float *top, float *logscales) {
  if (!win || n < win->k0 || n >= win->k0 + win->nloc) {
    set_error("rl_window_get_topology: bad arguments (target outside the window's range)");
    return RL_EINVAL;
  }
  const rl_ctx *ctx = win->ctx;
  RL_HIP(hipSetDevice(ctx->device));
  const int N = ctx->N, S = ctx->S, waves = ctx->waves;
  const int t = n - win->k0;
  const int D = (int)(win->top_off[t + 1] - win->top_off[t]);
  const Layout &lay = ctx->lay;
  if (top && (win->row_lo[t] != 0 || win->row_hi[t] != D)) {
    set_error("rl_window_get_topology: the window is bounded, rows %d..%d of %d of target %d are resident",
              win->row_lo[t], win->row_hi[t] - 1, D, n);
    return RL_ESTATE;
  }
  if (logscales) memcpy(logscales, &win->logscales[win->top_off[t]], (size_t)D * sizeof(float));
  if (top) {
    const size_t stride = (size_t)S * 64 * waves;  // one posterior row: [wave][register][lane]
    std::vector<float> phys((size_t)D * stride);
    RL_HIP(hipMemcpy(phys.data(), win->d_top.as<float>() + win->slab_off[t] * (int64_t)stride,
                     phys.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int d = 0; d < D; d++) {
      const float *row = &phys[(size_t)d * stride];
      float *o = top + (size_t)d * N;
      int p = 0;  // donor order = lane runs in order (the target's own entry is 0: alpha[n] = 0, fast_painting.cpp:781)
      for (int vl = 0; vl < 64 * waves; vl++) {
        const int len = lay.q + (vl < lay.rem ? 1 : 0);
        for (int i = 0; i < len; i++, p++) o[p] = row[((size_t)(vl >> 6) * S + i) * 64 + (vl & 63)];
      }
    }
  }
  return RL_OK;
}

int rl_window_advance(rl_window *win, int snp) {
  if (!win || snp < 0 || snp >= win->ctx->L) {
    set_error("rl_window_advance: bad arguments");
    return RL_EINVAL;
  }
  const rl_ctx *ctx = win->ctx;
  for (int t = 0; t < win->nloc; t++)
    if (derived(ctx, snp, win->k0 + t)) {  // anc_builder.cpp:489-494
      win->v_snp_prev[t]++;
      win->v_rpos_prev[t] = ctx->rpos[snp];
    }
  return RL_OK;
}

// rows of the window's targets: [nloc][N] (the whole matrix for a context with all targets)
static int window_matrix(rl_window *win, int snp, float *d_host, void *d_dev, float *kernel_ms,
                         const char *member = nullptr, float val = 0.0f, float *rowmin_dev = nullptr) {
  if (!win || snp < 0 || snp >= win->ctx->L) {
    set_error("rl_window_matrix: bad arguments");
    return RL_EINVAL;
  }
  rl_ctx *ctx = win->ctx;
  RL_HIP(hipSetDevice(ctx->device));
  const int N = ctx->N, L = ctx->L, k0 = win->k0, nloc = win->nloc;
  // the per-target arguments are written where the kernel reads them: a pinned block of the window
  if (!win->matrix_ready) {  // (first use; a failure half way leaves nothing a later call would trust)
    if (!win->stream && make_stream(&win->stream, false, true) != hipSuccess) win->stream = nullptr;
    if (win->stream && !win->e0 && hipEventCreate(&win->e0) != hipSuccess) win->e0 = nullptr;
    if (win->stream && !win->e2 && hipEventCreate(&win->e2) != hipSuccess) win->e2 = nullptr;
    if (win->stream && !win->h_stage) {
      // (the records of one matrix and, behind them, the N carrier flags of rl_window_matrix_rows_device_ex)
      win->h_stage = static_cast<unsigned char *>(pinned_cache_alloc((size_t)nloc * sizeof(MatrixArg) + (size_t)N + 64, &win->h_stage_bytes));
      if (!win->h_stage) tl_alloc_failures++;
    }
    if (!win->stream || !win->e0 || !win->e2 || !win->h_stage ||
        hipHostGetDevicePointer(&win->d_args, win->h_stage, 0) != hipSuccess) {
      win->d_args = nullptr;
      set_error("rl_window_matrix: stream / argument block creation failed");
      return RL_EHIP;
    }
    win->e_cursor.assign((size_t)nloc, -2);
    win->e_pn.assign((size_t)nloc, 1.0f);
    win->e_np.assign((size_t)nloc, 1.0f);
    win->matrix_ready = true;
  }
  MatrixArg *args = reinterpret_cast<MatrixArg *>(win->h_stage);
  bool covered = true;  // a bounded window: are the rows this tree reads resident?
  for (int t = 0; t < nloc; t++) {
    const int n = k0 + t;
    const int p = win->v_snp_prev[t];
    const int D = (int)(win->top_off[t + 1] - win->top_off[t]);
    const bool direct = derived(ctx, snp, n) || snp == 0 || snp == L - 1;
    if (p < 0 || p >= D || (!direct && p + 1 >= D)) {
      set_error("rl_window_matrix: cursor of target %d (%d) outside its %d posterior rows at SNP %d", n, p,
                D, snp);
      return RL_ESTATE;
    }
    MatrixArg &a = args[t];
    a.v_snp_prev = p;
    a.direct = direct ? 1 : 0;
    a.wl = a.wr = 0.5;
    a.e_pn = a.e_np = 1.0f;
    covered = covered && p >= win->row_lo[t] && p + (direct ? 0 : 1) < win->row_hi[t];
    if (direct) continue;
    if (win->v_rpos_next[t] <= win->v_rpos_prev[t]) {  // anc_builder.cpp:134-141
      for (int l = snp; l < L; l++)
        if (derived(ctx, l, n) || l == L - 1) {
          win->v_rpos_next[t] = ctx->rpos[l];
          break;
        }
    }
    const double rp = win->v_rpos_prev[t], rn = win->v_rpos_next[t];
    if (rp != rn) {  // :146-153
      const double denom = rn - rp;
      a.wl = (rn - ctx->rpos[snp]) / denom;
      a.wr = (ctx->rpos[snp] - rp) / denom;
    }
    if (win->e_cursor[t] != p) {  // (two expf per target and cursor position, not per matrix)
      const float lsp = win->logscales[win->top_off[t] + p], lsn = win->logscales[win->top_off[t] + p + 1];
      win->e_pn[t] = expf(lsp - lsn);  // float expf of a float difference (:167-168), glibc
      win->e_np[t] = expf(lsn - lsp);
      win->e_cursor[t] = p;
    }
    a.e_pn = win->e_pn[t];
    a.e_np = win->e_np[t];
  }
  if (!covered) {  // move on to the part of the window that starts here
    const int prc = place_rows(win, snp, nullptr);
    if (prc) return prc;
  }
  const auto t_mx = std::chrono::steady_clock::now();
  int rc = RL_OK;
  MatrixParams p;
  p.N = N;
  p.k0 = k0;
  p.nloc = nloc;
  p.topology = win->d_top.as<float>();
  p.logscales = win->d_ls.as<float>();
  p.top_off = win->d_top_off.as<int64_t>();
  p.slab_base = reinterpret_cast<const int64_t *>(win->d_place.as<unsigned char>() + (size_t)nloc * 8);
  if ((rc = win->d_stage.alloc((size_t)nloc * sizeof(MatrixArg) + (size_t)N + 64))) return rc;
  p.args = win->d_stage.as<MatrixArg>();
  p.host_args = static_cast<const MatrixArg *>(win->d_args);
  p.member = nullptr;
  p.val = val;
  p.rowmin = rowmin_dev;
  if (member) {  // the carrier flags travel with the records (one staging kernel, no copy of their own)
    memcpy(win->h_stage + (size_t)nloc * sizeof(MatrixArg), member, (size_t)N);
    p.member = win->d_stage.as<unsigned char>() + (size_t)nloc * sizeof(MatrixArg);
  }
  if (!d_dev && (rc = win->d_matrix.alloc((size_t)nloc * N * sizeof(float)))) return rc;
  p.matrix = d_dev ? static_cast<float *>(d_dev) : win->d_matrix.as<float>();
  RL_HIP(hipEventRecord(win->e0, win->stream));
  RL_HIP(launch_matrix(p, ctx->lay, ctx->S, ctx->waves, win->stream));
  RL_HIP(hipEventRecord(win->e2, win->stream));
  if (d_host)
    RL_HIP(hipMemcpyAsync(d_host, p.matrix, (size_t)nloc * N * sizeof(float), hipMemcpyDeviceToHost, win->stream));
  RL_HIP(hipStreamSynchronize(win->stream));
  if (kernel_ms) RL_HIP(hipEventElapsedTime(kernel_ms, win->e0, win->e2));
  win->t_matrix += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_mx).count();
  win->n_matrices++;
  return RL_OK;
}

int rl_window_matrix(rl_window *win, int snp, float *d_host, float *kernel_ms) {
  return window_matrix(win, snp, d_host, nullptr, kernel_ms);
}

int rl_window_matrix_rows_device(rl_window *win, int snp, void *d_rows, float *kernel_ms) {
  if (!d_rows) {
    set_error("rl_window_matrix_rows_device: null device pointer");
    return RL_EINVAL;
  }
  return window_matrix(win, snp, nullptr, d_rows, kernel_ms);
}

int rl_window_matrix_rows_device_ex(rl_window *win, int snp, void *d_rows, const char *carriers, float val,
                                    void *d_rowmin, float *kernel_ms) {
  if (!d_rows) {
    set_error("rl_window_matrix_rows_device_ex: null device pointer");
    return RL_EINVAL;
  }
  return window_matrix(win, snp, nullptr, d_rows, kernel_ms, carriers, val, static_cast<float *>(d_rowmin));
}

}  // extern "C"
