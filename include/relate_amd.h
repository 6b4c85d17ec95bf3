/*
 * relate_amd.h -- C ABI of the MI355X-native Relate Paint -> BuildTopology path.
 *
 * Plain C types only (no torch, no C++ types).  Every entry point names the
 * reference interface it replaces; paths are relative to
 * /root/reference/include (MyersGroup/relate @ 2025-04-10).
 *
 * Conventions
 *   - all functions return 0 on success and a negative RL_E* code on failure;
 *     rl_last_error() returns a thread-local description of the last failure
 *     (the reference asserts / exit(1)s instead: pipeline/Paint.cpp:24,78).
 *   - caller owns every host buffer passed in or out; device memory is owned
 *     by the rl_ctx / rl_window objects.
 *   - one rl_ctx per host thread and GPU.  No CPU fallback exists: functions
 *     that need the GPU fail with RL_ENODEVICE when none is visible.
 */
#ifndef RELATE_AMD_H
#define RELATE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  RL_OK = 0,
  RL_EINVAL = -1,    /* bad argument                                  */
  RL_ENODEVICE = -2, /* no usable HIP device                          */
  RL_EHIP = -3,      /* HIP runtime error (see rl_last_error)         */
  RL_EIO = -4,       /* file could not be opened / read / written     */
  RL_EFORMAT = -5,   /* malformed chunk / paint file                  */
  RL_ESTATE = -6,    /* call sequence error (e.g. paint before load)  */
  RL_ENOMEM = -7
};

/* Summation order of the per-site normalising constants.
 *   RL_SUM_EXACT : the reference's serial left-to-right order over donors
 *                  n = 0..N-1 (src/fast_painting.cpp:300-303, 495-503),
 *                  reproduced bit for bit by a parallel algorithm
 *                  (relate_amd/csrc/exact_sum.h).  Paint files / .anc / .mut
 *                  are byte-identical to the reference's.
 *   RL_SUM_EXACT_SERIAL : the same result computed literally (one dependent
 *                  add per donor); slower, kept as the in-kernel fallback of
 *                  RL_SUM_EXACT and for cross-checking it.
 *   RL_SUM_LANES : per-lane partial sums + wavefront xor-butterfly.  Same
 *                  arithmetic, different association: distances agree to
 *                  ~1e-7 relative before min-subtraction, trees may differ
 *                  where MinMatch breaks float ties.
 *   RL_SUM_LANES32 : the FAST mode.  As RL_SUM_LANES, with the per-donor state of
 *                  the stepping-stone pass (rl_paint) in packed FP32 instead of
 *                  double (the stones are floats in the reference too,
 *                  fast_painting.cpp:241-245); sums, factors, logscales and all of
 *                  RePaintSection stay double.  Distances within 1e-5 *
 *                  max(|d|, |logscale|) of the reference's (asserted in the
 *                  tests); NOT byte-identical files.                           */
enum { RL_SUM_EXACT = 0, RL_SUM_LANES = 1, RL_SUM_EXACT_SERIAL = 2, RL_SUM_LANES32 = 3 };

typedef struct rl_ctx rl_ctx;
typedef struct rl_window rl_window;

const char *rl_last_error(void);
const char *rl_version(void);
/* number of visible HIP devices (0 when none; never initialises a context) */
int rl_device_count(void);

/* ---------------------------------------------------------------- context */
rl_ctx *rl_create(int device);
/* A process that runs one stage and ends (the CLI, relate_amd/csrc/main.cpp) says so: the device blocks its contexts
 * release then stay in the library's cache until the process is gone instead of going back to the driver one hipFree
 * at a time (C3: ~3000 blocks, 4 s).  No reference counterpart (the reference frees with the process too). */
void rl_keep_cache_until_exit(int on);
void rl_destroy(rl_ctx *ctx);

/* Replaces `Data::Data(chunk files)` (src/data.cpp:86-97) + the parameter
 * read of pipeline/Paint.cpp:23-31: loads <dir>/parameters_c<c>.bin and
 * <dir>/chunk_<c>.{hap,r,rpos,bp,dist,state}. */
int rl_load_chunk(rl_ctx *ctx, const char *dir, int chunk_index);

/* Same, from memory (what a cgo/JNI-style binding of the reference's `Data`
 * struct, src/data.hpp:44-103, would pass).  seq: L*N chars '0'/'1',
 * SNP-major; r: L; rpos: L+1; wb: W+1 window boundaries (wb[W]==L). */
int rl_set_chunk(rl_ctx *ctx, int N, int L, const uint8_t *seq,
                 const double *r, const double *rpos, const int *wb, int W);
/* Same, panel already bit-packed (bit n of row s = derived), row_words
 * uint32 per row. */
int rl_set_chunk_bits(rl_ctx *ctx, int N, int L, const uint32_t *bits,
                      int row_words, const double *r, const double *rpos,
                      const int *wb, int W);

/* `--painting theta,rho` (pipeline/Paint.cpp:38-61): data.theta = theta,
 * data.r[l] *= rho.  Must precede rl_paint / rl_window_open. */
int rl_set_painting(rl_ctx *ctx, double theta, double rho);

/* Target-haplotype sharding of ONE chunk (BASELINE.json config #5):
 * this context paints, re-paints and measures targets k_begin .. k_end-1 only
 * (default: all).  The panel is replicated; stepping stones, posteriors and
 * distance-matrix rows are held for those targets only (1/G of the memory on G
 * GPUs).  Every per-target output below then has k_end-k_begin rows in target
 * order; the caller all-gathers the distance rows (rl_window_matrix_rows_device
 * + RCCL, relate_amd/dist.py) for the tree builder.  No reference counterpart:
 * the reference shards a chunk by section only (RelateParallel.sh:231-257).
 * Call after rl_set_chunk / rl_load_chunk and before rl_prepare / rl_paint. */
int rl_set_target_range(rl_ctx *ctx, int k_begin, int k_end);
int rl_target_range(const rl_ctx *ctx, int *k_begin, int *k_end);

int rl_chunk_dims(const rl_ctx *ctx, int *N, int *L, int *W);
/* sum_k D_k: visited (target, site) pairs of the chunk; 2*N*this is the
 * number of directional haplotype-pair.SNP updates of one Paint. */
long long rl_total_sites(rl_ctx *ctx);

/* ------------------------------------------------------------------ Paint */
/* Builds the per-target visited-site plan (host; src/fast_painting.cpp:41-157),
 * uploads panel + plan and allocates the stepping-stone buffers, so that a
 * following rl_paint finds everything resident in HBM.  Optional: rl_paint
 * does the same on first use. */
int rl_prepare(rl_ctx *ctx);

/* Replaces the hot loop `for hap: FastPainting::PaintSteppingStones`
 * (pipeline/Paint.cpp:81-87, src/fast_painting.cpp:18-618) for all N
 * targets: forward/backward Li-Stephens over the bit-packed panel, stepping
 * stones (alpha, beta, logscales at window boundaries) left in HBM.
 * kernel_ms (optional) receives the GPU time of the kernels (HIP events). */
int rl_paint(rl_ctx *ctx, int sum_mode, float *kernel_ms);
/* rl_paint normally paints both directions in ONE launch (2 workgroups per
 * target: the passes are independent in this stage, fast_painting.cpp:207-378
 * vs :380-586).  split != 0: one launch per direction (backward, then forward),
 * which is what rl_paint_times reports on. */
int rl_set_paint_split(rl_ctx *ctx, int split);
/* HIP-event durations of the last rl_paint's two launches (forward kernel,
 * backward kernel), in milliseconds; both 0 unless rl_set_paint_split(ctx, 1). */
int rl_paint_times(const rl_ctx *ctx, float *fwd_ms, float *bwd_ms);
/* Register tile of the loaded chunk: S doubles per lane, `waves` wavefronts
 * per target (relate_amd/csrc/launch.h) -- which kernel instantiation runs. */
int rl_register_tile(const rl_ctx *ctx, int *S, int *waves);

/* Test hook: `batch` arrays of n doubles each are summed, one wavefront per
 * array, with the summation machinery of the painting kernels (sum_mode as
 * for rl_paint); out[b] receives the sum of array b.  RL_SUM_EXACT and
 * RL_SUM_EXACT_SERIAL must return the left-to-right IEEE sum bit for bit. */
int rl_debug_wave_sum(const double *x, int n, int batch, int sum_mode, double *out);

/* Experiment builds only (kernels compiled with -DRL_STATS and
 * RELATE_AMD_STATS set): 16 event counters of the last rl_paint. */
int rl_debug_stats(rl_ctx *ctx, unsigned long long *out16);
/* ... and all 32 (16..24: cycles of the forward / backward step by segment, paint_kernels.hip) */
int rl_debug_stats32(rl_ctx *ctx, unsigned long long *out32);

/* Copy stepping stones of window w to the host: alpha, beta: N*N floats
 * ([target][donor]); ls_alpha, ls_beta: N floats; bsnp_begin/end: N ints.
 * Any pointer may be NULL. */
int rl_get_stones(rl_ctx *ctx, int w, float *alpha, float *beta,
                  float *ls_alpha, float *ls_beta, int *bsnp_begin,
                  int *bsnp_end);

/* Replaces the dump at src/fast_painting.cpp:589-601 +
 * CollapsedMatrix::DumpToFile (src/collapsed_matrix.hpp:228-265): writes
 * <paint_dir>/relate_<w>.bin for every window, byte-compatible with the
 * reference (lossy RLE included). */
int rl_write_paint_files(rl_ctx *ctx, const char *paint_dir);
/* One window's file only (what a section-parallel consumer of the reference,
 * scripts/RelateParallel/RelateParallel.sh:231-257, reads), at `path`. */
int rl_write_paint_file(rl_ctx *ctx, int w, const char *path);
/* One record of window w's paint file: target k's `int start, int end`, then
 * the alpha and the beta stone as CollapsedMatrix::DumpToFile writes them
 * (src/fast_painting.cpp:589-601, src/collapsed_matrix.hpp:228-265) -- the
 * bytes FastPainting::PaintSteppingStones(data, wb, pfiles, k) appends to
 * pfiles[w].  Two rows of N floats leave the device.  *len receives the
 * record's length; with out == NULL only the bound a buffer needs. */
int rl_paint_record(rl_ctx *ctx, int w, int k, unsigned char *out, size_t cap, size_t *len);

/* The whole `Relate --mode Paint` stage (pipeline/Paint.cpp:17-108): load,
 * optional --painting, mkdir chunk_<c>/paint, paint, write files. */
int rl_stage_paint(const char *out_dir, int chunk_index, int use_painting,
                   double theta, double rho, int sum_mode, int device);

/* ------------------------------------------------------- RePaint + matrix */
/* Replaces DistanceMeasure::GetTopologyWithRepaint
 * (src/anc_builder.cpp:49-106): decodes window w's stepping stones from
 * paint_file (or, if NULL, from the stones resident after rl_paint, passed
 * through the same float/RLE quantisation the file applies) and runs
 * FastPainting::RePaintSection (src/fast_painting.cpp:621-1092) for all N
 * targets.  The posterior rows (`topology`) and logscales stay in HBM.
 * first_snp initialises the cursors v_snp_prev / v_rpos_prev / v_rpos_next
 * (anc_builder.cpp:81-101). */
rl_window *rl_window_open(rl_ctx *ctx, int w, const char *paint_file,
                          int first_snp, int sum_mode, float *kernel_ms);
/* The same with at most max_rows posterior rows resident (0 = all of them).
 * The reference's DistanceMeasure keeps the whole window (`topology`,
 * src/anc_builder.hpp:58, filled at anc_builder.cpp:75-78); a tree at `snp`
 * reads two rows per target and the tree builder only moves forward
 * (anc_builder.cpp:487-495), so a bounded window keeps the rows from its
 * cursors onwards and runs RePaintSection again (same operations, same bits)
 * when rl_window_matrix is asked for a SNP past them.  N = 5000 keeps 20 GB
 * per window otherwise, which bounds how many sections one GPU serves.
 * rl_window_repaints: how many times RePaintSection has run for the window. */
rl_window *rl_window_open_bounded(rl_ctx *ctx, int w, const char *paint_file,
                                  int first_snp, int sum_mode,
                                  long long max_rows, float *kernel_ms);
int rl_window_repaints(const rl_window *win);
void rl_window_close(rl_window *win);
int rl_window_bounds(const rl_window *win, int *start, int *end);
/* rows D_n of target n's topology and copies of it (tests / debugging):
 * top: D_n*N floats in donor order, logscales: D_n floats. */
int rl_window_rows(const rl_window *win, int n);
int rl_window_get_topology(rl_window *win, int n, float *top,
                           float *logscales);

/* AncesTreeBuilder::BuildTopology's cursor update for the carriers of `snp`
 * (src/anc_builder.cpp:487-495). */
int rl_window_advance(rl_window *win, int snp);
/* Replaces DistanceMeasure::GetMatrix(snp) (src/anc_builder.cpp:109-207):
 * N*N float distance matrix at `snp` (the rows of the context's targets, all
 * by default), row-min subtracted, into d_host
 * (may be NULL to leave the result on the device only).
 * kernel_ms optional. */
int rl_window_matrix(rl_window *win, int snp, float *d_host, float *kernel_ms);
/* The same, written to a caller-owned DEVICE buffer of (targets of the
 * context) * N floats -- the send buffer of the all-gather when the chunk is
 * sharded by target (rl_set_target_range). */
int rl_window_matrix_rows_device(rl_window *win, int snp, void *d_rows, float *kernel_ms);
/* The same with what AncesTreeBuilder::BuildTopology does to the matrix next folded into the kernel's last pass over
 * each row: carriers (N flags, host memory; NULL: none) -- the carrier penalty of src/anc_builder.cpp:563-581,
 * d[c][j] += val along a carrier's row, -= val again at carrier columns --, and d_rowmin (device, one float per row
 * of the context's targets; NULL: not wanted) -- the row minima off the diagonal MinMatch::Initialize starts from
 * (src/tree_builder.cpp:1659-1666), of the rows as they then are. */
int rl_window_matrix_rows_device_ex(rl_window *win, int snp, void *d_rows, const char *carriers, float val,
                                    void *d_rowmin, float *kernel_ms);

/* --------------------------------------------------------------- host side */
/* Replaces MinMatch::QuickBuild (src/tree_builder.cpp:1061-1303 without
 * prior, :2358-2644 with prior) for sample_ages == empty: builds the tree
 * from the asymmetric N*N matrix d (destroyed).  parent[2N-1] receives the
 * parent of every node (-1 for the root); child_left/child_right (optional)
 * the children of the N-1 internal nodes N..2N-2. */
int rl_quickbuild(int N, double theta, float *d, const float *d_prior,
                  int *parent, int *child_left, int *child_right);
/* The same as an object, the way AncesTreeBuilder::BuildTopology uses one
 * MinMatch for all trees of a section (src/anc_builder.cpp:436, :447, :608):
 * the builder keeps what MinMatch keeps from one QuickBuild to the next
 * (min_values_CF, the stale candidate indices -- they steer the random
 * draws).  device >= 0: trees are built on that GPU, one workgroup per tree
 * with the matrices in HBM (src/tree_builder.cpp restated in
 * relate_amd/csrc/minmatch_gpu.hip); a tree that needs the symmetric
 * fallback (tree_builder.cpp:255-293, :968-1058) is built by the host code
 * with the same state.  device < 0: the host builder.  d is destroyed.
 * rl_builder_last_on_gpu: 1 if the last tree was built on the GPU. */
typedef struct rl_builder rl_builder;
rl_builder *rl_builder_create(int N, double theta, int device);
int rl_builder_build(rl_builder *b, float *d, const float *d_prior,
                     int *parent, int *child_left, int *child_right);
/* MinMatch::QuickBuild's sample_ages (N doubles): from here on the builder's trees are built with the sample-age
 * key and clock of src/tree_builder.cpp:1123-1233 / :2407-2531 -- on the builder's device (device >= 0), else on the
 * host. */
int rl_builder_set_sample_ages(rl_builder *b, const double *ages, int n);
int rl_builder_last_on_gpu(const rl_builder *b);
void rl_builder_destroy(rl_builder *b);
/* Test hook (host only): the device builder's restatement of std::mt19937 +
 * libstdc++'s uniform_real_distribution<double> against the library itself,
 * n draws from `seed`; returns how many differ. */
int rl_debug_rng_mismatches(unsigned seed, int n);
/* Measurement hook (tools/bench_builder_many.py): `builders` device builders, one host thread each, build the same tree
 * (d, prior: N*N floats, prior may be NULL) `reps` times side by side, the matrices staying on the device, with at most
 * `workers` resident workgroups (0: the queue's own limit).  seconds: wall-clock of all builds; mismatches: builds
 * whose parent array differs from builder 0's of the same repetition; first_parents (may be NULL): builder 0's
 * [reps][2N-1].  MinMatch::QuickBuild, src/tree_builder.cpp:2358-2644. */
int rl_debug_builder_throughput(int N, double theta, int device, int builders, int reps, int workers, const float *d,
                                const float *prior, double *seconds, int *mismatches, int *first_parents);
/* Test hook (host only): the stage's rule for how many resident tree-builder workgroups it asks for -- never more
 * than open sections (a section has one tree in flight: the unit of parallelism of
 * scripts/RelateParallel/RelateParallel.sh:231-257); 7/8 of the worker slots with whole windows resident, 29/64 of the
 * CUs (times the kernel's workgroups per CU) when the windows are bounded and RePaint runs all through the stage. */
int rl_debug_stage_worker_goal(int cus, int open_sections, int bounded_windows, int workers_per_cu);

/* Tree-sequence loop of one section, AncesTreeBuilder::BuildTopology
 * (src/anc_builder.cpp:398-656): first tree from the distance matrix at
 * `start`, then for every SNP map its carriers onto the current tree
 * (MapMutation, :1064-1139) and rebuild (carrier penalty + previous-tree clade
 * prior, :555-606; MinMatch) when it does not map.  The distance matrices come
 * from a provider: `matrix(user, snp, d)` must fill the N*N matrix of
 * DistanceMeasure::GetMatrix(snp) and `advance(user, snp)` (may be NULL) is
 * called for every SNP start < snp < end before its matrix can be requested
 * (the cursor update of :487-495).  rl_stage_build_topology plugs rl_window
 * in; any other DistanceMeasure implementation can be plugged the same way.
 * bits/row_words: bit-packed panel as for rl_set_chunk_bits; bp_pos (L, only
 * for --fb) and state (L, chunk_<c>.state) may be NULL.
 * flags bit0 = --no_consistency; fb = --fb (0 = off). */
typedef struct rl_treeseq rl_treeseq;
typedef int (*rl_matrix_fn)(void *user, int snp, float *d);
typedef int (*rl_advance_fn)(void *user, int snp);
rl_treeseq *rl_treeseq_create(int N, int L, const uint32_t *bits, int row_words,
                              const double *rpos, const int *bp_pos,
                              const int *state, double theta);
void rl_treeseq_destroy(rl_treeseq *ts);
/* sample ages (N doubles; n = 0: none) for the trees of rl_treeseq_build: MinMatch::QuickBuild's sample_ages argument
 * (src/anc_builder.cpp:373-390). */
int rl_treeseq_set_sample_ages(rl_treeseq *ts, const double *ages, int n);
/* device >= 0: rl_treeseq_build builds its trees on that GPU (see rl_builder_create),
 * < 0 (default): on the host. */
int rl_treeseq_set_build_device(rl_treeseq *ts, int device);
/* With a build device: where the distance matrix of `snp` comes from when it
 * is to stay on the device (N*N floats at d_dev, e.g. through
 * rl_window_matrix_rows_device); carrier penalty, clade prior and the build
 * then run there too and no matrix crosses PCIe. */
typedef int (*rl_matrix_dev_fn)(void *user, int snp, void *d_dev);
int rl_treeseq_set_device_matrix(rl_treeseq *ts, rl_matrix_dev_fn matrix_dev);
/* ... and a provider that also applies the carrier penalty and leaves the row minima (rl_window_matrix_rows_device_ex):
 * with it the device builder skips its own pass over the matrix.  carriers: N flags or NULL. */
typedef int (*rl_matrix_dev_ex_fn)(void *user, int snp, void *d_matrix, const char *carriers, float val,
                                   void *d_rowmin);
int rl_treeseq_set_device_matrix_ex(rl_treeseq *ts, rl_matrix_dev_ex_fn matrix_dev_ex);
int rl_treeseq_build(rl_treeseq *ts, int start, int end, rl_matrix_fn matrix,
                     rl_advance_fn advance, void *user, int flags, int fb);
int rl_treeseq_num_trees(const rl_treeseq *ts);
/* position (first SNP) and parent array (2N-1, root = -1) of tree t */
int rl_treeseq_get_tree(const rl_treeseq *ts, int t, int *pos, int *parent);
/* AncesTree::DumpBin (src/anc.cpp:1104-1167) and Mutations::DumpShortFormat
 * (src/mutations.cpp:548-581); either path may be NULL. */
int rl_treeseq_write(const rl_treeseq *ts, const char *anc_path,
                     const char *mut_path);

/* The whole `Relate --mode BuildTopology` stage
 * (pipeline/BuildTopology.cpp:14-167) for sections first..last: writes
 * <out>/chunk_<c>/<out>_<section>.anc and .mut.
 * flags: bit0 = --no_consistency, fb = --fb value (0 = off). */
int rl_stage_build_topology(const char *out_dir, int chunk_index,
                            int first_section, int last_section,
                            int use_painting, double theta, double rho,
                            int flags, int fb, int sum_mode, int device);

/* Everything a stage call can be told, per call (the positional entry points above and below take the reference's
 * command-line options only: pipeline/BuildTopology.cpp:20-40).  rl_stage_opts_init fills in the defaults and the
 * struct's size; a caller built against an older header passes a shorter struct and gets the defaults for the rest;
 * a struct whose size field was never set (0) is refused with RL_EINVAL.
 * 0 / -1 = "decide from the chunk and the device" wherever noted.  The RELATE_AMD_* environment variables of the
 * same names override the struct -- they exist for experiments (tools/, profiles/), not as the interface. */
typedef struct rl_stage_opts {
  size_t size;              /* sizeof(rl_stage_opts), set by rl_stage_opts_init                                   */
  int sum_mode;             /* RL_SUM_*                                                                            */
  int device;               /* HIP device                                                                          */
  int use_painting;         /* --painting given: theta, rho (pipeline/Paint.cpp:38-61)                             */
  double theta, rho;
  int flags;                /* bit 0: --no_consistency                                                             */
  int fb;                   /* --fb (0: off)                                                                       */
  const char *sample_ages_path; /* --sample_ages (NULL: none); replaces rl_stage_set_sample_ages                   */
  int gpu_build;            /* trees on the device (1), on the host (0); -1: device when several sections are asked for */
  long long window_rows;    /* posterior rows a window keeps resident; 0: from the HBM that is free; < 0: all      */
  int window_parts;         /* a window keeps at least 1/window_parts of its rows (0: 32)                          */
  int section_threads;      /* sections open at once at most (0: from HBM and the device)                          */
  int workers;              /* tree-builder workgroups on the device (0: the rule of treeseq.cpp stage_worker_goal -- one per open section, 29/64 of the CUs when the windows are bounded, times the kernel's workgroups per CU) */
  int repaint_lanes;        /* RePaint launches side by side: 1 or 2 (0: 1)                                        */
  int park_stones;          /* fused stage: stepping stones to pinned host memory after Paint (rl_park_stones)     */
  int pin_threads;          /* section threads pinned to L3 groups: 1 / 0 (-1: 1)                                  */
  int find_equivalent_branches; /* 1: the next stage, FindEquivalentBranches (pipeline/FindEquivalentBranches.cpp:13-167), fused
                               behind BuildTopology -- the sections' trees are associated in memory while other sections
                               still build, every .anc is written ONCE, as that stage would leave it (same bytes as
                               BuildTopology followed by rl_stage_find_equivalent_branches).  Only for a call that covers
                               all sections of the chunk; host memory: ~0.3 MB per tree at N = 5000                  */
} rl_stage_opts;
void rl_stage_opts_init(rl_stage_opts *opts);
int rl_stage_paint_ex(const char *out_dir, int chunk_index, const rl_stage_opts *opts);
int rl_stage_build_topology_ex(const char *out_dir, int chunk_index, int first_section,
                               int last_section, const rl_stage_opts *opts);
int rl_stage_paint_build_topology_ex(const char *out_dir, int chunk_index, int first_section,
                                     int last_section, const rl_stage_opts *opts);

/* DEPRECATED (process-wide state): use rl_stage_opts.sample_ages_path.
 * `--sample_ages <file>` of BuildTopology (pipeline/BuildTopology.cpp:93-108; ancient samples): the file (plain or
 * gzip text, one age per haplotype) the FOLLOWING rl_stage_build_topology / rl_stage_paint_build_topology calls of
 * this process read; NULL or "" for none.  With ages the trees are built by the host's sequential restatement of
 * MinMatch's third candidate key and coalescence clock (src/tree_builder.cpp:7-22, 149-252, 601-965, 1123-1233,
 * 1738-1841, 2073-2355, 2407-2531). */
int rl_stage_set_sample_ages(const char *file);

/* Paint and BuildTopology of one chunk in one process, as `Relate --mode All` runs
 * them back to back per chunk (pipeline/Relate.cpp:257-283): the stepping stones
 * never leave HBM -- no chunk_<c>/paint/relate_<w>.bin (2.4 GB per 20 windows at
 * N = 5000) is written or read; the float / run-length quantisation the file
 * would apply (src/collapsed_matrix.hpp:228-296) is applied on the device.  Same
 * .anc / .mut as rl_stage_paint followed by rl_stage_build_topology. */
/* (between rl_paint and the windows of a context, for chunks whose stones would not leave room for the sections'
 * windows) moves the painted stepping stones to pinned host memory and frees their HBM; a window opened without a
 * paint file afterwards takes its slice back.  What the fused stage does under RELATE_AMD_PARK_STONES=1. */
int rl_park_stones(rl_ctx *ctx);

int rl_stage_paint_build_topology(const char *out_dir, int chunk_index,
                            int first_section, int last_section,
                            int use_painting, double theta, double rho,
                            int flags, int fb, int sum_mode, int device);

/* ------------------------------------- one chunk sharded by target haplotype */
/* BASELINE.json config #5 (N = 10,000 x L = 200k: 8 N^2 W = 288 GB of stepping stones, more than one GPU holds).
 * The reference's unit is the section (scripts/RelateParallel/RelateParallel.sh:231-257): each BuildTopology process
 * decodes the stones of ALL targets of its window, re-paints them and fills whole matrices (src/anc_builder.cpp:49-106,
 * :109-207).  Here a rank (one process per GPU) is an rl_shard: the chunk loaded with rl_set_target_range(k_begin,
 * k_end), its own targets painted (from_paint_files = 0: stones stay in HBM, W*(k_end-k_begin)*N floats per
 * direction) or their records read from the Paint stage's files (from_paint_files = 1).
 *   rl_shard_rows: rows k_begin..k_end-1 of DistanceMeasure::GetMatrix(snp) of `section` into a device buffer of
 *     (k_end-k_begin)*N floats.  The first call for a section opens its window (GetTopologyWithRepaint for the
 *     shard's targets); every call replays the cursor updates of anc_builder.cpp:487-495 from the SNP of the previous
 *     call up to snp (they are a function of the panel), so SNPs must not decrease within a section.
 *   rl_shard_release_section: the section's trees are done, its window goes.
 *   rl_shard_build_section: the tree-sequence loop of ONE section (AncesTreeBuilder::BuildTopology,
 *     anc_builder.cpp:398-656) on this rank, its OWNER: every matrix comes from `matrix` (N*N floats to the host) or
 *     -- build_device >= 0 and matrix_dev given -- from `matrix_dev` (N*N floats to a device buffer of that GPU;
 *     penalty, prior and the build run there), which the caller implements by asking every shard for its rows and
 *     exchanging them (relate_amd/dist.py run_chunk_by_targets: an RCCL all-gather per matrix).  Writes
 *     <out>/chunk_<c>/<out>_<section>.anc/.mut like rl_stage_build_topology.
 *   rl_shard_expect_builders: how many sections this rank will own at once with build_device >= 0 (sizes the device
 *     tree builder's worker pool); 0 when done.
 *   rl_shard_set_window_rows: posterior rows a window keeps resident (rl_window_open_bounded's max_rows; 0 = all).
 *   rl_device_copy: device-to-device copy on `device` (received row blocks into the matrix_dev buffer). */
typedef struct rl_shard rl_shard;
rl_shard *rl_shard_open(const char *out_dir, int chunk_index, int k_begin, int k_end,
                        int use_painting, double theta, double rho, int sum_mode,
                        int device, int from_paint_files);
void rl_shard_close(rl_shard *s);
int rl_shard_dims(const rl_shard *s, int *N, int *L, int *W, int *k_begin, int *k_end);
int rl_shard_section_bounds(const rl_shard *s, int section, int *start, int *end);
int rl_shard_set_window_rows(rl_shard *s, long long rows);
int rl_shard_rows(rl_shard *s, int section, int snp, void *d_rows);
int rl_shard_release_section(rl_shard *s, int section);
int rl_shard_expect_builders(rl_shard *s, int builders);
int rl_shard_build_section(rl_shard *s, int section, int flags, int fb, int build_device,
                           rl_matrix_fn matrix, rl_matrix_dev_fn matrix_dev, void *user,
                           int *num_trees);
int rl_device_copy(void *dst, const void *src, size_t bytes, int device);

/* ------------------------------------------------------------ MakeChunks */
/* Replaces Data::MakeChunks (src/data.cpp:117-518): parses .haps / .sample
 * (plain or gzip) and the genetic map, decides chunks and windows from the
 * memory allowance (GB, reference default 5) and writes parameters.bin,
 * parameters_c<i>.bin, chunk_<i>.{hap,state,bp,dist,rpos,r} and props.bin into
 * the existing directory out_dir, byte-identical to the reference.  dist_fn may
 * be NULL ("unspecified").  use_transitions = 0 is --transversion. */
int rl_make_chunks(const char *haps_fn, const char *sample_fn, const char *map_fn,
                   const char *dist_fn, const char *out_dir, int use_transitions,
                   float memory_gb);
/* The `Relate --mode MakeChunks` stage (pipeline/MakeChunks.cpp:13-114):
 * refuses an existing out_dir, creates it, then rl_make_chunks. */
int rl_stage_make_chunks(const char *haps_fn, const char *sample_fn,
                         const char *map_fn, const char *dist_fn,
                         const char *out_dir, int transversion, float memory_gb);

/* ------------------------------------------------- FindEquivalentBranches */
/* The stage after BuildTopology (pipeline/FindEquivalentBranches.cpp:13-167;
 * AncesTreeBuilder::BranchAssociation / AssociateTrees, src/anc_builder.cpp:
 * 1454-1613, 658-800): reads <out_dir>/chunk_<c>/<name>_<w>.anc of every window,
 * finds equivalent branches in neighbouring trees and rewrites the files with
 * num_events / SNP_begin / SNP_end carried along them.  Host code. */
int rl_stage_find_equivalent_branches(const char *out_dir, int chunk_index);
/* Test hook (no GPU): the association fused behind BuildTopology (rl_stage_opts.find_equivalent_branches) fed from the
 * chunk's .anc FILES, sections handed over in `order` (all of them, any order), pool_threads associating. */
int rl_debug_feb_fused_from_files(const char *out_dir, int chunk_index, const int *order, int n_order,
                                  int pool_threads);

/* ------------------------------------------------------------------ tools */
/* Synthetic block-coalescent panel (stand-in for MakeChunks input,
 * SURVEY.md 8d).  seq_chars (L*N) and/or bits (L*row_words) may be NULL. */
int rl_synth_panel(int N, int L, uint64_t seed, int block, int jitter,
                   uint8_t *seq_chars, uint32_t *bits, int row_words, int *bp,
                   double *r, double *rpos);
/* the reference's window rule (src/data.cpp:213-229); returns W (<0 on error) */
int rl_synth_windows(int N, int L, const uint8_t *seq_chars, double budget,
                     int *wb, int max_windows);
int rl_synth_windows_bits(int N, int L, const uint32_t *bits, int row_words,
                          double budget, int *wb, int max_windows);
/* chunk files as Data::MakeChunks writes them (src/data.cpp:261-298,485-516) */
int rl_write_chunk_files(const char *dir, int chunk, int N, int L,
                         const uint8_t *seq_chars, const int *bp,
                         const double *r, const double *rpos, const int *wb,
                         int W);

#ifdef __cplusplus
}
#endif
#endif
