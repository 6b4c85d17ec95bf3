/*
 * relate_oracle.h -- CPU oracle for the Relate Paint -> BuildTopology hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (MyersGroup/relate @ 2025-04-10) used as the checker for the HIP
 * path and as the `cpu_baseline` leg of bench.py.  Nothing under
 * relate_amd/ may include, link or call it.
 *
 * Parity status: PINNED.  The restatement reproduces, byte for byte, the
 * paint files written by the reference binary compiled from /root/reference
 * (oracle/Makefile -> oracle/_ref/Relate) and the RePaintSection / GetMatrix
 * dumps of oracle/ref_harness.cpp on the fixtures under tests/golden/
 * (see tests/test_oracle_golden.py and tools/make_golden.py).
 *
 * Every function cites the reference lines it follows
 * (paths relative to /root/reference/include/src/).
 */
#ifndef RELATE_ORACLE_H
#define RELATE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One chunk of haplotype data, laid out exactly as the reference's `Data`
 * struct holds it (data.hpp:44-103): L x N chars '0'/'1', SNP-major. */
typedef struct {
  int N, L;
  const char *seq;    /* L*N chars */
  const double *r;    /* L   : recombination distance snp -> snp+1 (scaled) */
  const double *rpos; /* L+1 : cumulative recombination position            */
  double theta;       /* mutation probability for painting (data.cpp:95)    */
} ro_data;

/* How the per-site normalising sums are accumulated.
 *   RO_SUM_SERIAL : left-to-right over n = 0..N-1, as the reference does
 *                   (fast_painting.cpp:300-303, 495-503).
 *   RO_SUM_LANES  : the order of the HIP RL_SUM_LANES kernels: the N-1 donors
 *                   n != k are cut into 64 contiguous, balanced runs, each run
 *                   summed left to right, then an xor-butterfly over the 64
 *                   partial sums (masks 1,2,4,8,16,32).  Used to check those
 *                   kernels bit for bit; it is NOT the reference order. */
enum { RO_SUM_SERIAL = 0, RO_SUM_LANES = 1, RO_SUM_PROBE = 2 };
/* RO_SUM_PROBE (experiments, tools/exact_sum_stats.py): the serial order, and
 * every normalising sum's term vector is shown to ro_sum_probe first
 * (dir 0: forward, alpha; 1: backward, e(n)*beta).  Not thread-safe. */
typedef void (*ro_sum_probe_fn)(const double *terms, int N, int dir);
void ro_set_sum_probe(ro_sum_probe_fn fn);

typedef struct {
  int mode;   /* RO_SUM_* */
  int seg;    /* unused */
  int nwaves; /* unused */
} ro_sum_order;

/* fast_log.hpp:6-21 */
float ro_fast_log(float val);

/* Visited-site plan of target k (fast_painting.cpp:41-157).
 * Outputs (caller allocates L entries each, L+1 for r_prob):
 *   site[D], r_prob[D+1], nor_x_theta[D], bsnp_begin[W], bsnp_end[W].
 * Returns D (number of visited sites). */
int ro_plan_target(const ro_data *d, const int *wb, int W, int k, int *site,
                   double *r_prob, double *nor_x_theta, int *bsnp_begin,
                   int *bsnp_end);

/* FastPainting::PaintSteppingStones (fast_painting.cpp:18-618) for target k,
 * returning the stepping stones instead of writing them.
 *   alpha, beta : W*N floats;  ls_alpha, ls_beta : W floats
 *   bsnp_begin, bsnp_end : W ints
 * Returns the number of visited sites D_k, <0 on error. */
int ro_paint_stepping_stones(const ro_data *d, const int *wb, int W, int k,
                             const ro_sum_order *order, int *bsnp_begin,
                             int *bsnp_end, float *alpha, float *beta,
                             float *ls_alpha, float *ls_beta);

/* CollapsedMatrix<float>::DumpToFile(fp,i,boundarySNP,logscales)
 * (collapsed_matrix.hpp:228-265): encode one stepping stone.  `out` must
 * hold ro_stone_max_bytes(N).  Returns bytes written. */
size_t ro_stone_max_bytes(int N);
size_t ro_encode_stone(const float *v, int N, int bsnp, float logscale,
                       unsigned char *out);
/* CollapsedMatrix<float>::ReadFromFile(fp,boundarySNP,logscale)
 * (collapsed_matrix.hpp:268-296).  Returns bytes consumed, 0 on error. */
size_t ro_decode_stone(const unsigned char *in, size_t avail, int N, float *v,
                       int *bsnp, float *logscale);

/* The Paint stage (pipeline/Paint.cpp:66-93): paints every target and writes
 * <dir>/relate_<w>.bin for all windows.  nthreads>1 splits targets across
 * pthreads (output identical).  If targets_limit>0 only targets
 * [0,targets_limit) are painted (bench sampling).  Returns 0 on success.
 * total_sites (optional) receives sum_k D_k over the painted targets. */
int ro_paint_chunk(const ro_data *d, const int *wb, int W, const char *dir,
                   int nthreads, int targets_limit, const ro_sum_order *order,
                   long long *total_sites);

/* Timing-only variant for bench.py's cpu_baseline: paints targets
 * k0, k0+stride, ... (count targets) on nthreads threads, discards output.
 * Returns sum of D_k, <0 on error. */
long long ro_paint_sample(const ro_data *d, const int *wb, int W, int k0,
                          int stride, int count, int nthreads);

/* FastPainting::RePaintSection (fast_painting.cpp:621-1092).
 * topology: caller allocates (bsnp_end-bsnp_begin+2)*N floats, logscales
 * (bsnp_end-bsnp_begin+2) floats.  Returns D (rows written). */
int ro_repaint_section(const ro_data *d, const float *alpha_begin,
                       const float *beta_end, int bsnp_begin, int bsnp_end,
                       float ls_alpha, float ls_beta, int k,
                       const ro_sum_order *order, float *topology,
                       float *logscales);

/* One row of DistanceMeasure::GetMatrix (anc_builder.cpp:116-194).
 *   top_prev/top_next : rows v_snp_prev[n] and v_snp_prev[n]+1 of target n's
 *                       topology (top_next may be NULL when `direct`)
 *   direct            : seq[snp][n]=='1' || snp==0 || snp==L-1
 *   rpos_prev/next/snp: interpolation positions (ignored when direct)
 * Writes N floats to row (min-subtracted, diagonal 0). */
void ro_distance_row(int N, int n, int direct, const float *top_prev,
                     const float *top_next, float ls_prev, float ls_next,
                     double rpos_prev, double rpos_next, double rpos_snp,
                     float *row);

/* DistanceMeasure state for one window, CPU side: holds top[n], log[n] for
 * all targets (anc_builder.hpp:50-109). */
typedef struct ro_window ro_window;
/* GetTopologyWithRepaint (anc_builder.cpp:49-106): read <paint_prefix>_<w>.bin
 * and repaint every target.  snp = first SNP the matrix is asked for. */
ro_window *ro_window_open(const ro_data *d, const char *paint_file, int snp,
                          int nthreads);
void ro_window_free(ro_window *w);
/* BuildTopology's cursor update for carriers of `snp`
 * (anc_builder.cpp:487-495). */
void ro_window_advance(ro_window *w, int snp);
/* GetMatrix(snp) (anc_builder.cpp:109-207) into matrix (N*N floats). */
void ro_window_matrix(ro_window *w, int snp, float *matrix);
/* accessors used by the tests */
int ro_window_rows(const ro_window *w, int n);
const float *ro_window_top(const ro_window *w, int n);
const float *ro_window_log(const ro_window *w, int n);
int ro_window_start(const ro_window *w);
int ro_window_end(const ro_window *w);

#ifdef __cplusplus
}
#endif
#endif
