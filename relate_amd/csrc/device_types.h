// device_types.h -- parameter blocks shared by host code and HIP kernels.
#pragma once
#include <stdint.h>

namespace rl {

// Lane layout of the donors (DESIGN.md "register layout").
// The wave of target k holds all P = N donors in donor order (the slot of k
// itself is pinned to +0.0); lane l owns the contiguous run
//   [start_l, start_l + len_l),  start_l = l*q + min(l, rem),
//   len_l = q + (l < rem),       q = P / 64, rem = P % 64
// in registers 0..len_l-1; registers >= len_l hold +0.0 and stay +0.0.
struct Layout {
  int N, P, q, rem;
};

struct PaintConsts {
  double theta, ntheta;
  double K1;          // theta_ratio + 1.0           (fast_painting.hpp:36)
  double init0;       // prior_ntheta                (fast_painting.hpp:35)
  double init1;       // prior_theta + prior_ntheta  (fast_painting.cpp:219)
  double log_Nm1;     // log(N-1)                    (fast_painting.cpp:399)
  double log_ntheta;  // log(1-theta)
  double lower, upper;  // rescaling thresholds 1e-10 / 1e10
  double inv_theta, inv_ntheta;  // RN(1 / theta), RN(1 / ntheta): div_by_const (paint_device.h)
};

struct PaintParams {
  Layout lay;
  PaintConsts c;
  int L, W;
  int k0, nloc;             // this context's targets are k0 .. k0+nloc-1 (rl_set_target_range); stones hold those rows
  int S;                    // words per row of the lane-mask panel (= register tile)
  const unsigned long long *masks;  // [L+2][S] lane-mask panel (paint_device.h), built by panel_kernels.hip
  const int64_t *plan_off;  // [N+1] offsets of target k's visited sites
  const int32_t *sites;     // visited site | (seq_k derived ? 1<<31 : 0)
  const double *cf;         // r_prob_i / ((1 - r_prob_i) * (N-1))
  const double *nxt;        // nor_x_theta_i
  const int32_t *stone_ia;  // [N][W] visited index of boundarySNP_begin[w]
  const int32_t *stone_ie;  // [N][W] visited index of boundarySNP_end[w]
  const double *binit;      // [N] beta_sum at the last SNP (serial, host)
  const int32_t *order;     // [nloc] launch order -> target (longest first)
  float *alpha, *beta;      // [W][nloc][N] stepping stones, donor order
  float *ls_alpha, *ls_beta;  // [W][nloc]
  int sum_mode;             // RL_SUM_EXACT / RL_SUM_LANES / RL_SUM_EXACT_SERIAL
  int merge_order;          // one launch for both directions: 0 = backward blocks first, then forward; 1 = interleaved
  unsigned long long *stats;  // 16 event counters (experiment builds with -DRL_STATS), else null
};

// RePaintSection over one window, all targets.
// The forward kernel keeps every REPAINT_CHECKPOINT-th alpha row of every target in HBM; the backward kernel
// rebuilds the rows in between from the nearest one (repaint_kernels.hip).  Per step it also keeps 3 doubles (side
// record).
#ifndef RL_REPAINT_CHECKPOINT
#define RL_REPAINT_CHECKPOINT 6
#endif
constexpr int REPAINT_CHECKPOINT = RL_REPAINT_CHECKPOINT;  // (2 .. 8; -DRL_REPAINT_CHECKPOINT=k for experiments)
// bytes of the forward kernel's output for a window of `rows` target-site rows over nloc targets (an upper bound:
// every target rounds its checkpoint rows up)
inline size_t repaint_scratch_bytes(int64_t rows, int nloc, int S, int waves) {
  return (size_t)((rows / REPAINT_CHECKPOINT + nloc) * (int64_t)S * 64 * waves + rows * 3) * sizeof(double);
}
constexpr int REPAINT_SIDE = 3;  // doubles per step: the step's additive constant, its rescaling divisor (0: none), logscale
struct RepaintParams {
  Layout lay;
  PaintConsts c;
  int L;
  int k0, nloc;  // targets k0 .. k0+nloc-1; the per-target arrays below are indexed by t = n - k0
  const unsigned long long *masks;  // [L+2][S] lane-mask panel, as in PaintParams
  const int64_t *plan_off;
  const int32_t *sites;
  const double *cf;
  const double *nxt;
  // per target: slice [ib, ie] of its visited list covered by this window and
  // the coefficients of the window's last interval (r[last_snp] only,
  // fast_painting.cpp:702-716)
  const int32_t *ib, *ie;     // [nloc]
  const double *cf_last;      // [nloc]
  const double *nxt_last;     // [nloc]
  const float *alpha_begin;   // [nloc][N] decoded stones, donor order
  const float *beta_end;      // [nloc][N]
  const float *ls_alpha;      // [nloc]
  const float *ls_beta;       // [nloc]
  const int64_t *top_off;     // [nloc+1] row offsets into logscales (every posterior row of the window)
  // Posterior rows [row_lo, row_hi) of target t are kept, row j at topology row slab_off[t] + j - row_lo[t]: the
  // whole window (row_lo = 0, row_hi = D, slab_off = top_off) or the part of it the tree builder is working in.
  const int64_t *slab_off;    // [nloc]
  const int32_t *row_lo, *row_hi;  // [nloc]
  float *topology;            // [kept rows][S*64] register-major: row[i*64 + lane] = donor start_lane + i
  float *logscales;           // [sum D]
  double *scratch;            // checkpoint alpha rows [ck_off[nloc]][waves][S*64]: target t's rows 0, CK, 2CK ... of
                              // its D_t forward rows start at row ck_off[t]
  const int64_t *ck_off;      // [nloc+1]
  double *side;               // side records [top_off[nloc]][REPAINT_SIDE], target t's at top_off[t]
  const int32_t *order;       // [nloc] targets (global index), longest first
  int sum_mode;
  int partial;                // the logscales of every row are in place (an earlier launch of this window wrote them):
  int nostrip;                // a part launch's backward kernel without the LDS strip, two waves a SIMD (repaint_kernels.hip)
                              // the forward pass may stop below row_hi, the backward pass at row_lo
  // A bounded window keeps ONE state of the backward pass per target -- beta (doubles, register-major like a
  // checkpoint row), the step's factor and the running logscale as they stand before row r is done -- so that a
  // later launch, whose rows lie below r, starts there instead of at the window's last row.  save_row[t] >= 0: this
  // launch leaves the state at that row; start_row[t] >= 0: it starts from the state kept there (window.cpp).
  double *bstate;             // [nloc][waves][S*64], or null
  double *bscal;              // [nloc][2]
  const int32_t *start_row, *save_row;  // [nloc]
  // ... and ONE of the forward pass: alpha, the step's factor and the logscales as they stand behind row r, a
  // multiple of the checkpoint interval -- where the rows of the next part begin, so that its launch starts there
  // instead of at the window's first row.  fsave_row / fstart_row as above.
  double *fstate;             // [nloc][waves][S*64], or null
  double *fscal;              // [nloc][4]: cfac, prev_ls, lsf
  const int32_t *fstart_row, *fsave_row;  // [nloc]
};

// what one row of a distance matrix is told by the host (anc_builder.cpp:130-168), 32 bytes per target
struct MatrixArg {
  double wl, wr;        // interpolation weights
  float e_pn, e_np;     // expf(ls_prev - ls_next), expf(ls_next - ls_prev): glibc's, for the reference's bits
  int32_t v_snp_prev;   // the target's cursor
  int32_t direct;       // 1: no interpolation (the target is derived at the SNP)
};
struct MatrixParams {
  int N;
  int k0, nloc;  // rows of targets k0 .. k0+nloc-1; per-row arrays and `matrix` are indexed by t = n - k0
  const float *topology;
  const float *logscales;
  const int64_t *top_off;   // [nloc+1] row offsets into logscales
  const int64_t *slab_base;   // [nloc] posterior row j of target t is topology row slab_base[t] + j
  const MatrixArg *args;      // [nloc] in HBM ...
  const MatrixArg *host_args; // ... copied there from this pinned host block by a small kernel ahead of the matrix
                              // kernel (matrix_kernels.hip), or null: a copy engine between the host's loop and a
                              // 0.08 ms kernel cost more than the kernel (0.5 ms per matrix through the ABI)
  float *matrix;              // [nloc][N]
  // optional (rl_window_matrix_rows_device_ex): the carrier penalty of the tree-sequence loop applied to the rows as
  // they leave (anc_builder.cpp:563-581: + val along a carrier's row, - val again at carrier columns) and the rows'
  // minima off the diagonal as they then are (tree_builder.cpp:1659-1666) -- what a pass of its own over the matrix did
  const unsigned char *member;  // [N] carrier flags behind `args` in HBM (staged with them), or null: no penalty
  float val;
  float *rowmin;                // [nloc], or null
};

}  // namespace rl
