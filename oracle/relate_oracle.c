/*
 * relate_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see relate_oracle.h).
 *
 * Plain-C restatement of the reference's Paint / RePaint / GetMatrix
 * arithmetic.  Build with -ffp-contract=off: every C operator below must be
 * exactly one IEEE operation, as in the reference's baseline x86-64 build
 * (SURVEY.md App. A).  Citations are to /root/reference/include/src/.
 */
#define _GNU_SOURCE
#include "relate_oracle.h"

#include <assert.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* EXPERIMENT (tools/exp_fp32_state.py): -DRO_FP32_STATE builds a variant in which the per-donor
 * state of the Li-Stephens passes (alpha, beta) is rounded to float after every update -- what a kernel with packed
 * FP32 state would hold -- while sums, factors and logscales stay double.  The default build is the restatement. */
#ifdef RO_FP32_STATE
#define RO_STATE(v) ((double)(float)(v))
#else
#define RO_STATE(v) (v)
#endif


/* ------------------------------------------------------------------ */
/* fast_log.hpp:6-21 : exponent extraction + quadratic on the mantissa */
float ro_fast_log(float val) {
  int32_t x;
  memcpy(&x, &val, 4);
  const int log_2 = ((x >> 23) & 255) - 128;
  x &= ~(255 << 23);
  x += 127 << 23;
  memcpy(&val, &x, 4);
  val = ((-1.0f / 3) * val + 2) * val - 2.0f / 3;
  return (val + log_2) * 0.69314718f;
}

/* ------------------------------------------------------------------ */
/* constants of FastPainting::FastPainting (fast_painting.hpp:26-39)   */
typedef struct {
  double lower, upper;
  double Nminusone, prior_theta, prior_ntheta, theta_ratio, log_ntheta,
      log_small;
  double theta, ntheta;
} paint_consts;

static void consts_init(paint_consts *c, const ro_data *d) {
  c->lower = 1e-10;
  c->upper = 1.0 / c->lower;
  c->theta = d->theta;
  c->ntheta = 1.0 - d->theta;
  c->Nminusone = d->N - 1.0;
  c->prior_theta = c->theta / c->Nminusone - c->ntheta / c->Nminusone;
  c->prior_ntheta = c->ntheta / c->Nminusone;
  c->theta_ratio = c->theta / (1.0 - c->theta) - 1.0;
  c->log_ntheta = log(c->ntheta);
  c->log_small = log(0.01);
}

/* r_prob / nor_x_theta of one interval (fast_painting.cpp:72-80) */
static inline void interval_coeffs(const paint_consts *c, double rho,
                                   double *r_prob, double *nor_x_theta) {
  double nxt = -rho + c->log_ntheta;
  double rp = 1.0 - exp(-rho);
  if (rp > 0.99) {
    rp = 0.99;
    nxt = c->log_small + c->log_ntheta;
  }
  *r_prob = rp;
  *nor_x_theta = nxt;
}

/* transition factor r/((1-r)(N-1)) (fast_painting.cpp:260,351) */
static inline double trans_factor(const paint_consts *c, double rp) {
  return rp / ((1.0 - rp) * c->Nminusone);
}

/* ------------------------------------------------------------------ */
/* fast_painting.cpp:41-157 */
static int plan_target(const ro_data *d, const paint_consts *c, const int *wb,
                       int W, int k, int *site, double *r_prob,
                       double *nor_x_theta, int *bsnp_begin, int *bsnp_end) {
  const int N = d->N, L = d->L, last = L - 1;
  const char *seq = d->seq;
  int pb = 0, pe = 0; /* next boundary slots */
  int window_index = 1;
  int window_end = wb[1];
  assert(wb[W] == L);

  bsnp_begin[pb++] = 0;

  int D = 0;
  int snp = 1;
  double acc = d->r[0];
  site[0] = 0;
  for (;;) {
    /* skip sites where k is ancestral (:56-59, :93-96) */
    while (seq[(size_t)snp * N + k] != '1' && snp != last) {
      acc += d->r[snp];
      snp++;
    }
    /* window bookkeeping (:60-69, :98-107) */
    if (snp >= window_end && site[D] < window_end) {
      while (window_end <= snp) {
        bsnp_end[pe++] = snp;
        bsnp_begin[pb++] = site[D];
        window_index++;
        window_end = wb[window_index];
      }
    }
    interval_coeffs(c, acc, &r_prob[D], &nor_x_theta[D]);
    D++;
    site[D] = snp;
    acc = d->r[snp];
    snp++;
    if (snp >= L) break;
  }
  interval_coeffs(c, acc, &r_prob[D], &nor_x_theta[D]);
  D++;
  r_prob[D] = 1.0; /* "just a technicality" (:147) */
  bsnp_end[pe++] = last;
  assert(pb == W && pe == W);
  return D;
}

int ro_plan_target(const ro_data *d, const int *wb, int W, int k, int *site,
                   double *r_prob, double *nor_x_theta, int *bsnp_begin,
                   int *bsnp_end) {
  paint_consts c;
  consts_init(&c, d);
  return plan_target(d, &c, wb, W, k, site, r_prob, nor_x_theta, bsnp_begin,
                     bsnp_end);
}

/* ------------------------------------------------------------------ */
/* summation orders                                                   */
/* Order of the HIP RL_SUM_LANES kernels (relate_amd/csrc/paint_device.h):
 * all N donors (the target's own term is +0.0), in donor order, are cut into
 * nl contiguous runs (the first N%nl runs hold N/nl+1 donors, the rest N/nl);
 * each run is summed left to right from 0.0, then an xor-butterfly (masks
 * 1..nl/2) combines the partial sums.  nl = 64 lanes; for N > 5120 a target
 * gets two wavefronts (nl = 128; relate_amd/csrc/launch.h target_waves). */
static double sum_lanes(const double *t, int N, int k, int nl) {
  const int q = N / nl, rem = N % nl;
  double lane[128];
  int p = 0;
  (void)k;
  for (int l = 0; l < nl; l++) {
    const int len = q + (l < rem ? 1 : 0);
    double s = 0.0;
    for (int i = 0; i < len; i++, p++) s += t[p];
    lane[l] = s;
  }
  for (int m = 1; m < nl; m <<= 1) {
    double nxt[128];
    for (int l = 0; l < nl; l++) nxt[l] = lane[l] + lane[l ^ m];
    memcpy(lane, nxt, sizeof lane);
  }
  return lane[0];
}
static inline int paint_lanes(int N) { return N > 80 * 64 ? 128 : 64; }

static ro_sum_probe_fn g_probe = NULL;
void ro_set_sum_probe(ro_sum_probe_fn fn) { g_probe = fn; }

static inline double sum_alpha(const double *a, int N, int k, const ro_sum_order *o, int nl) {
  if (o != NULL && o->mode == RO_SUM_PROBE && g_probe) g_probe(a, N, 0);
  if (o == NULL || o->mode != RO_SUM_LANES) {
    double s = 0.0;
    for (int n = 0; n < N; n++) s += a[n]; /* :300-303 */
    return s;
  }
  return sum_lanes(a, N, k, nl);
}

/* sum_n e(n)*b[n], e = theta if (seq_k > row[n]) else ntheta  (:495-503) */
static inline double sum_beta(const double *b, const char *row, int k,
                              int N, const paint_consts *c,
                              const ro_sum_order *o, double *scratch, int nl) {
  const char seq_k = row[k];
  if (o != NULL && o->mode == RO_SUM_PROBE && g_probe) {
    for (int n = 0; n < N; n++)
      scratch[n] = (seq_k > row[n]) ? c->theta * b[n] : c->ntheta * b[n];
    g_probe(scratch, N, 1);
  }
  if (o == NULL || o->mode != RO_SUM_LANES) {
    double s = 0.0;
    for (int n = 0; n < N; n++) {
      if (seq_k > row[n])
        s += c->theta * b[n];
      else
        s += c->ntheta * b[n];
    }
    return s;
  }
  for (int n = 0; n < N; n++)
    scratch[n] = (seq_k > row[n]) ? c->theta * b[n] : c->ntheta * b[n];
  return sum_lanes(scratch, N, k, nl);
}

/* ------------------------------------------------------------------ */
/* PaintSteppingStones, fast_painting.cpp:18-618                       */
typedef struct {
  int *site;
  double *r_prob, *nor_x_theta;
  double *a, *b, *scratch;
  int cap_L, cap_N;
} paint_ws;

static int ws_init(paint_ws *ws, int N, int L) {
  ws->site = (int *)malloc(sizeof(int) * (size_t)(L + 1));
  ws->r_prob = (double *)malloc(sizeof(double) * (size_t)(L + 2));
  ws->nor_x_theta = (double *)malloc(sizeof(double) * (size_t)(L + 1));
  ws->a = (double *)malloc(sizeof(double) * (size_t)N);
  ws->b = (double *)malloc(sizeof(double) * (size_t)N);
  ws->scratch = (double *)malloc(sizeof(double) * (size_t)N);
  ws->cap_L = L;
  ws->cap_N = N;
  return (ws->site && ws->r_prob && ws->nor_x_theta && ws->a && ws->b &&
          ws->scratch)
             ? 0
             : -1;
}
static void ws_free(paint_ws *ws) {
  free(ws->site);
  free(ws->r_prob);
  free(ws->nor_x_theta);
  free(ws->a);
  free(ws->b);
  free(ws->scratch);
}

static int paint_target(const ro_data *d, const paint_consts *c, const int *wb,
                        int W, int k, const ro_sum_order *order, paint_ws *ws,
                        int *bsnp_begin, int *bsnp_end, float *alpha,
                        float *beta, float *ls_alpha, float *ls_beta) {
  const int N = d->N, L = d->L;
  const char *seq = d->seq;
  int *site = ws->site;
  double *r_prob = ws->r_prob, *nxt = ws->nor_x_theta;
  double *a = ws->a, *b = ws->b;

  const int D =
      plan_target(d, c, wb, W, k, site, r_prob, nxt, bsnp_begin, bsnp_end);

  /* ---------------- forward (:201-378) ---------------- */
  int wa = 0; /* next alpha stone */
  {
    const char *row = seq; /* SNP 0 */
    const char seq_k = row[k];
    for (int n = 0; n < N; n++) {
      double derived = (double)(seq_k > row[n]);
      a[n] = derived * c->prior_theta + c->prior_ntheta; /* :219 */
    }
    a[k] = 0.0;
  }
  double S = sum_alpha(a, N, k, order, paint_lanes(N));
  double ls = 0.0;
  while (wa < W && bsnp_begin[wa] == 0) { /* :233-253 */
    for (int n = 0; n < N; n++) alpha[(size_t)wa * N + n] = (float)a[n];
    ls_alpha[wa] = (float)ls;
    wa++;
  }
  double cfac = trans_factor(c, r_prob[0]) * S; /* :260 */
  for (int i = 1; i < D; i++) {
    const int snp = site[i];
    const char *row = seq + (size_t)snp * N;
    const char seq_k = row[k];
    ls += nxt[i - 1]; /* :281-282 */
    for (int n = 0; n < N; n++) { /* :288-295 */
      double v = a[n] + cfac;
      double derived = (double)(seq_k > row[n]);
      v *= derived * c->theta_ratio + 1.0;
      a[n] = RO_STATE(v);
    }
    a[k] = 0.0;
    S = sum_alpha(a, N, k, order, paint_lanes(N));
    cfac = S;
    if (cfac < c->lower || cfac > c->upper) { /* :334-347 */
      const double tmp = cfac;
      for (int n = 0; n < N; n++) a[n] = RO_STATE(a[n] / tmp);
      ls += log(tmp);
      cfac = 1.0;
    }
    cfac *= trans_factor(c, r_prob[i]); /* :349-352, r_prob[i] < 1 always */
    while (wa < W && bsnp_begin[wa] == snp) { /* :354-374 */
      for (int n = 0; n < N; n++) alpha[(size_t)wa * N + n] = (float)a[n];
      ls_alpha[wa] = (float)ls;
      wa++;
    }
  }
  assert(wa == W);

  /* ---------------- backward (:396-582) ---------------- */
  const double normalizing_constant =
      (double)log(c->Nminusone) - D * c->log_ntheta; /* :399 */
  ls = normalizing_constant;
  int we = W - 1; /* next beta stone, descending */
  double B = 0.0;
  {
    const char *row = seq + (size_t)(L - 1) * N;
    const char seq_k = row[k];
    for (int n = 0; n < N; n++) b[n] = 1.0;
    /* a sum of constants: always serial (the HIP path takes it from the host
     * plan, relate_amd/csrc/context.cpp build_plan step 3) */
    for (int n = 0; n < N; n++) { /* :421-430 */
      if (seq_k > row[n])
        B += c->theta;
      else
        B += c->ntheta;
    }
    B -= c->ntheta; /* :431 */
  }
  while (we >= 0 && bsnp_end[we] == L - 1) { /* :433-448 */
    for (int n = 0; n < N; n++) beta[(size_t)we * N + n] = (float)b[n];
    ls_beta[we] = (float)ls;
    we--;
  }
  cfac = trans_factor(c, r_prob[D - 1]) * B; /* :454-455 */
  for (int j = D - 2; j >= 0; j--) {
    const int snp = site[j], snp_next = site[j + 1];
    const char *row_next = seq + (size_t)snp_next * N;
    const char seqk_next = row_next[k];
    ls += nxt[j + 1]; /* :471-472 (interval AFTER the later site) */
    const double b1 = cfac / c->ntheta;       /* :474 */
    const double bt = cfac / c->theta - b1;   /* :475 */
    for (int n = 0; n < N; n++) {             /* :481-488 */
      double derived = (double)(seqk_next > row_next[n]);
      double v = b[n] + derived * bt + b1;
      v *= derived * c->theta_ratio + 1.0;
      b[n] = RO_STATE(v);
    }
    b[k] = 0.0;
    const char *row = seq + (size_t)snp * N;
    B = sum_beta(b, row, k, N, c, order, ws->scratch, paint_lanes(N));
    cfac = B;
    if (cfac < c->lower || cfac > c->upper) { /* :538-551 */
      const double tmp = cfac;
      for (int n = 0; n < N; n++) b[n] = RO_STATE(b[n] / tmp);
      ls += ro_fast_log((float)tmp); /* float fast_log here (:548) */
      cfac = 1.0;
    }
    cfac *= trans_factor(c, r_prob[j]); /* :553-556 */
    while (we >= 0 && bsnp_end[we] == snp) { /* :559-578 */
      for (int n = 0; n < N; n++) beta[(size_t)we * N + n] = (float)b[n];
      ls_beta[we] = (float)ls;
      we--;
    }
  }
  assert(we == -1);
  return D;
}

int ro_paint_stepping_stones(const ro_data *d, const int *wb, int W, int k,
                             const ro_sum_order *order, int *bsnp_begin,
                             int *bsnp_end, float *alpha, float *beta,
                             float *ls_alpha, float *ls_beta) {
  paint_consts c;
  paint_ws ws;
  consts_init(&c, d);
  if (ws_init(&ws, d->N, d->L)) return -1;
  int D = paint_target(d, &c, wb, W, k, order, &ws, bsnp_begin, bsnp_end, alpha,
                       beta, ls_alpha, ls_beta);
  ws_free(&ws);
  return D;
}

/* ------------------------------------------------------------------ */
/* stepping-stone records, collapsed_matrix.hpp:228-296                */
size_t ro_stone_max_bytes(int N) { return 8 + 8 + 4 + 4 + 4 + (size_t)N * 8; }

size_t ro_encode_stone(const float *v, int N, int bsnp, float logscale,
                       unsigned char *out) {
  float *uniq = (float *)malloc(sizeof(float) * (size_t)N);
  int *times = (int *)malloc(sizeof(int) * (size_t)N);
  for (int j = 0; j < N; j++) times[j] = 1;
  float current = v[0];
  int k = 0;
  uniq[0] = current;
  for (int j = 1; j < N; j++) {
    /* float difference, double product (:243) */
    float diff = fabsf(current - v[j]);
    float mn = (v[j] < current) ? v[j] : current; /* std::min(a,b): b<a?b:a */
    if ((double)diff < 1e-3 * (double)mn) {
      times[k]++;
    } else {
      current = v[j];
      k++;
      uniq[k] = current;
    }
  }
  k++;
  unsigned char *p = out;
  uint64_t isize = 1, isub = (uint64_t)N;
  memcpy(p, &isize, 8); p += 8;
  memcpy(p, &isub, 8); p += 8;
  memcpy(p, &bsnp, 4); p += 4;
  memcpy(p, &logscale, 4); p += 4;
  memcpy(p, &k, 4); p += 4;
  memcpy(p, uniq, (size_t)k * 4); p += (size_t)k * 4;
  memcpy(p, times, (size_t)k * 4); p += (size_t)k * 4;
  free(uniq);
  free(times);
  return (size_t)(p - out);
}

size_t ro_decode_stone(const unsigned char *in, size_t avail, int N, float *v,
                       int *bsnp, float *logscale) {
  if (avail < 28) return 0;
  uint64_t isize, isub;
  int k;
  memcpy(&isize, in, 8);
  memcpy(&isub, in + 8, 8);
  if (isize != 1 || isub != (uint64_t)N) return 0;
  memcpy(bsnp, in + 16, 4);
  memcpy(logscale, in + 20, 4);
  memcpy(&k, in + 24, 4);
  if (k < 0 || avail < 28 + (size_t)k * 8) return 0;
  const unsigned char *pu = in + 28, *pt = in + 28 + (size_t)k * 4;
  int i = 0;
  for (int j = 0; j < k; j++) {
    float u;
    int t;
    memcpy(&u, pu + (size_t)j * 4, 4);
    memcpy(&t, pt + (size_t)j * 4, 4);
    for (int q = 0; q < t; q++) {
      if (i >= N) return 0;
      v[i++] = u;
    }
  }
  if (i != N) return 0;
  return 28 + (size_t)k * 8;
}

/* ------------------------------------------------------------------ */
/* Paint stage, pipeline/Paint.cpp:66-93 + fast_painting.cpp:589-601    */
typedef struct {
  const ro_data *d;
  const int *wb;
  int W;
  int k0, k1, stride;
  const ro_sum_order *order;
  /* outputs: per target, per window encoded record (start,end,alpha,beta) */
  unsigned char **rec; /* [k*W + w] */
  size_t *rec_len;
  long long sites;
  int discard;
  int rc;
} paint_job;

static void *paint_worker(void *arg) {
  paint_job *j = (paint_job *)arg;
  const ro_data *d = j->d;
  const int N = d->N, W = j->W;
  paint_consts c;
  paint_ws ws;
  consts_init(&c, d);
  j->rc = 0;
  j->sites = 0;
  if (ws_init(&ws, N, d->L)) {
    j->rc = -1;
    return NULL;
  }
  int *bb = (int *)malloc(sizeof(int) * (size_t)W);
  int *be = (int *)malloc(sizeof(int) * (size_t)W);
  float *alpha = (float *)malloc(sizeof(float) * (size_t)W * N);
  float *beta = (float *)malloc(sizeof(float) * (size_t)W * N);
  float *la = (float *)malloc(sizeof(float) * (size_t)W);
  float *lb = (float *)malloc(sizeof(float) * (size_t)W);
  const size_t maxrec = 8 + 2 * ro_stone_max_bytes(N);
  for (int k = j->k0; k < j->k1; k += j->stride) {
    int D = paint_target(d, &c, j->wb, W, k, j->order, &ws, bb, be, alpha, beta,
                         la, lb);
    j->sites += D;
    if (j->discard) continue;
    for (int w = 0; w < W; w++) {
      unsigned char *buf = (unsigned char *)malloc(maxrec);
      unsigned char *p = buf;
      int start = j->wb[w], end = j->wb[w + 1] - 1; /* :591-594 */
      memcpy(p, &start, 4); p += 4;
      memcpy(p, &end, 4); p += 4;
      p += ro_encode_stone(alpha + (size_t)w * N, N, bb[w], la[w], p);
      p += ro_encode_stone(beta + (size_t)w * N, N, be[w], lb[w], p);
      size_t len = (size_t)(p - buf);
      j->rec[(size_t)k * W + w] = (unsigned char *)realloc(buf, len);
      j->rec_len[(size_t)k * W + w] = len;
    }
  }
  free(bb); free(be); free(alpha); free(beta); free(la); free(lb);
  ws_free(&ws);
  return NULL;
}

int ro_paint_chunk(const ro_data *d, const int *wb, int W, const char *dir,
                   int nthreads, int targets_limit, const ro_sum_order *order,
                   long long *total_sites) {
  const int N = d->N;
  const int NT = (targets_limit > 0 && targets_limit < N) ? targets_limit : N;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > NT) nthreads = NT;
  unsigned char **rec =
      (unsigned char **)calloc((size_t)NT * W, sizeof(unsigned char *));
  size_t *rec_len = (size_t *)calloc((size_t)NT * W, sizeof(size_t));
  paint_job *jobs = (paint_job *)calloc((size_t)nthreads, sizeof(paint_job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int t = 0; t < nthreads; t++) {
    jobs[t].d = d; jobs[t].wb = wb; jobs[t].W = W;
    jobs[t].k0 = t; jobs[t].k1 = NT; jobs[t].stride = nthreads;
    jobs[t].order = order; jobs[t].rec = rec; jobs[t].rec_len = rec_len;
    jobs[t].discard = 0;
    pthread_create(&th[t], NULL, paint_worker, &jobs[t]);
  }
  long long sites = 0;
  int rc = 0;
  for (int t = 0; t < nthreads; t++) {
    pthread_join(th[t], NULL);
    sites += jobs[t].sites;
    if (jobs[t].rc) rc = jobs[t].rc;
  }
  if (total_sites) *total_sites = sites;
  if (rc == 0 && dir != NULL) {
    for (int w = 0; w < W && rc == 0; w++) {
      char fn[2048];
      snprintf(fn, sizeof fn, "%s/relate_%i.bin", dir, w);
      FILE *fp = fopen(fn, "wb");
      if (!fp) { rc = -2; break; }
      for (int k = 0; k < NT; k++)
        fwrite(rec[(size_t)k * W + w], 1, rec_len[(size_t)k * W + w], fp);
      fclose(fp);
    }
  }
  for (size_t i = 0; i < (size_t)NT * W; i++) free(rec[i]);
  free(rec); free(rec_len); free(jobs); free(th);
  return rc;
}

long long ro_paint_sample(const ro_data *d, const int *wb, int W, int k0,
                          int stride, int count, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > count) nthreads = count;
  paint_job *jobs = (paint_job *)calloc((size_t)nthreads, sizeof(paint_job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  /* target list: k0 + i*stride, i<count; thread t takes i = t, t+T, ... */
  for (int t = 0; t < nthreads; t++) {
    jobs[t].d = d; jobs[t].wb = wb; jobs[t].W = W;
    jobs[t].k0 = k0 + t * stride;
    jobs[t].k1 = k0 + count * stride;
    if (jobs[t].k1 > d->N) jobs[t].k1 = d->N;
    jobs[t].stride = stride * nthreads;
    jobs[t].order = NULL; jobs[t].discard = 1;
    pthread_create(&th[t], NULL, paint_worker, &jobs[t]);
  }
  long long sites = 0;
  for (int t = 0; t < nthreads; t++) {
    pthread_join(th[t], NULL);
    if (jobs[t].rc) sites = -1;
    if (sites >= 0) sites += jobs[t].sites;
  }
  free(jobs); free(th);
  return sites;
}

/* ------------------------------------------------------------------ */
/* RePaintSection, fast_painting.cpp:621-1092                          */
int ro_repaint_section(const ro_data *d, const float *alpha_begin,
                       const float *beta_end, int first_snp, int last_snp,
                       float logscale_alpha, float logscale_beta, int k,
                       const ro_sum_order *order, float *topology,
                       float *logscales) {
  const int N = d->N;
  const char *seq = d->seq;
  paint_consts cc;
  consts_init(&cc, d);
  const paint_consts *c = &cc;
  const int max_snps = last_snp - first_snp + 2;
  int *site = (int *)malloc(sizeof(int) * (size_t)max_snps);
  double *r_prob = (double *)malloc(sizeof(double) * (size_t)(max_snps + 1));
  double *nxt = (double *)malloc(sizeof(double) * (size_t)max_snps);
  double *scratch = (double *)malloc(sizeof(double) * (size_t)N);

  /* visited sites inside [first_snp,last_snp] (:639-720) */
  int D = 0;
  {
    int snp = first_snp + 1;
    double acc = d->r[first_snp];
    site[0] = first_snp;
    for (;;) {
      while (seq[(size_t)snp * N + k] != '1' && snp != last_snp) {
        acc += d->r[snp];
        snp++;
      }
      interval_coeffs(c, acc, &r_prob[D], &nxt[D]);
      D++;
      site[D] = snp;
      acc = d->r[snp];
      snp++;
      if (snp > last_snp) break;
    }
    interval_coeffs(c, acc, &r_prob[D], &nxt[D]);
    D++;
    r_prob[D] = 1.0;
  }

  double *alpha = (double *)malloc(sizeof(double) * (size_t)D * N);
  double *b = (double *)malloc(sizeof(double) * (size_t)N);
  for (int i = 0; i < D; i++) logscales[i] = 0.0f;

  /* ---------------- forward (:769-885) ---------------- */
  logscales[0] = logscale_alpha;
  double *a = alpha;
  for (int n = 0; n < N; n++) a[n] = alpha_begin[n];
  a[k] = 0.0;
  double S = sum_alpha(a, N, k, order, paint_lanes(N));
  double cfac = trans_factor(c, r_prob[0]) * S;
  double prev_logscale = logscales[0];
  for (int i = 1; i < D; i++) {
    const char *row = seq + (size_t)site[i] * N;
    const char seq_k = row[k];
    prev_logscale += nxt[i - 1];
    logscales[i] = (float)prev_logscale; /* :806-807 */
    const double *ap = alpha + (size_t)(i - 1) * N;
    a = alpha + (size_t)i * N;
    for (int n = 0; n < N; n++) {
      double v = ap[n] + cfac;
      double derived = (double)(seq_k > row[n]);
      v *= derived * c->theta_ratio + 1.0;
      a[n] = RO_STATE(v);
    }
    a[k] = 0.0;
    S = sum_alpha(a, N, k, order, paint_lanes(N));
    cfac = S;
    if (cfac < c->lower || cfac > c->upper) { /* :865-877 */
      const double tmp = cfac;
      for (int n = 0; n < N; n++) a[n] = RO_STATE(a[n] / tmp);
      prev_logscale += log(tmp);
      logscales[i] = (float)(logscales[i] + log(tmp));
      cfac = 1.0;
    }
    cfac *= trans_factor(c, r_prob[i]);
  }

  /* ---------------- backward (:887-1073) ---------------- */
  logscales[D - 1] = logscales[D - 1] + logscale_beta; /* float += float :895 */
  for (int n = 0; n < N; n++) b[n] = beta_end[n];
  b[k] = 0.0;
  {
    const char *row = seq + (size_t)last_snp * N;
    double B0 = sum_beta(b, row, k, N, c, order, scratch, paint_lanes(N));
    a = alpha + (size_t)(D - 1) * N;
    float *t = topology + (size_t)(D - 1) * N;
    for (int n = 0; n < N; n++) t[n] = (float)(a[n] * b[n]); /* :930 */
    cfac = trans_factor(c, r_prob[D - 1]) * B0;
  }
  prev_logscale = logscale_beta; /* :951 */
  for (int j = D - 2; j >= 0; j--) {
    const char *row_next = seq + (size_t)site[j + 1] * N;
    const char seqk_next = row_next[k];
    prev_logscale += nxt[j + 1];
    logscales[j] = (float)(logscales[j] + prev_logscale); /* :962-963 */
    const double b1 = cfac / c->ntheta;
    const double bt = cfac / c->theta - b1;
    for (int n = 0; n < N; n++) {
      double derived = (double)(seqk_next > row_next[n]);
      double v = b[n] + derived * bt + b1;
      v *= derived * c->theta_ratio + 1.0;
      b[n] = RO_STATE(v);
    }
    b[k] = 0.0;
    const char *row = seq + (size_t)site[j] * N;
    double B = sum_beta(b, row, k, N, c, order, scratch, paint_lanes(N));
    cfac = B;
    a = alpha + (size_t)j * N;
    float *t = topology + (size_t)j * N;
    for (int n = 0; n < N; n++) t[n] = (float)(a[n] * b[n]); /* :1039 */
    if (cfac < c->lower || cfac > c->upper) { /* :1047-1061 */
      const double tmp = cfac;
      for (int n = 0; n < N; n++) b[n] = RO_STATE(b[n] / tmp);
      prev_logscale += log(tmp);
      logscales[j] = (float)(logscales[j] + log(tmp));
      cfac = 1.0;
    }
    cfac *= trans_factor(c, r_prob[j]);
  }

  free(site); free(r_prob); free(nxt); free(scratch); free(alpha); free(b);
  return D;
}

/* ------------------------------------------------------------------ */
/* GetMatrix rows, anc_builder.cpp:116-194                             */
void ro_distance_row(int N, int n, int direct, const float *top_prev,
                     const float *top_next, float ls_prev, float ls_next,
                     double rpos_prev, double rpos_next, double rpos_snp,
                     float *row) {
  const float scale = -1.0f;
  float min = INFINITY;
  if (direct) {
    for (int j = 0; j < N; j++) {
      row[j] = (ro_fast_log(top_prev[j]) + ls_prev) * scale; /* :128 */
      if (row[j] < min) min = row[j];
    }
  } else {
    double wl, wr;
    if (rpos_prev == rpos_next) { /* :146-153 */
      wl = 0.5;
      wr = 0.5;
    } else {
      double denom = rpos_next - rpos_prev;
      wl = (rpos_next - rpos_snp) / denom;
      wr = (rpos_snp - rpos_prev) / denom;
    }
    /* float expf of a float difference (:167-168) */
    const float e_pn = expf(ls_prev - ls_next);
    const float e_np = expf(ls_next - ls_prev);
    for (int j = 0; j < N; j++) {
      float x;
      if (ls_prev <= ls_next) { /* :172-178 */
        x = (float)(wl * top_prev[j] * e_pn + wr * top_next[j]);
        row[j] = (ro_fast_log(x) + ls_next) * scale;
      } else {
        x = (float)(wl * top_prev[j] + wr * top_next[j] * e_np);
        row[j] = (ro_fast_log(x) + ls_prev) * scale;
      }
      if (row[j] < min) min = row[j];
    }
  }
  row[n] = 0.0f;
  for (int j = 0; j < N; j++)
    if (j != n) row[j] -= min; /* :190-192 */
}

/* ------------------------------------------------------------------ */
/* DistanceMeasure window state, anc_builder.cpp:49-207                */
struct ro_window {
  const ro_data *d;
  int start, end;
  int *rows;      /* D_n */
  float **top;    /* D_n*N floats */
  float **log;    /* D_n floats */
  int *v_snp_prev;
  double *v_rpos_prev, *v_rpos_next;
};

typedef struct {
  ro_window *w;
  const unsigned char *buf;
  const size_t *off;
  size_t len;
  int n0, n1, stride;
  int rc;
} repaint_job;

static void *repaint_worker(void *arg) {
  repaint_job *j = (repaint_job *)arg;
  ro_window *w = j->w;
  const int N = w->d->N;
  float *ab = (float *)malloc(sizeof(float) * (size_t)N);
  float *be = (float *)malloc(sizeof(float) * (size_t)N);
  j->rc = 0;
  for (int n = j->n0; n < j->n1; n += j->stride) {
    const unsigned char *p = j->buf + j->off[n];
    size_t avail = j->len - j->off[n];
    int bb, bend;
    float la, lb;
    p += 8; avail -= 8;
    size_t u = ro_decode_stone(p, avail, N, ab, &bb, &la);
    if (!u) { j->rc = -1; break; }
    p += u; avail -= u;
    u = ro_decode_stone(p, avail, N, be, &bend, &lb);
    if (!u) { j->rc = -1; break; }
    int cap = bend - bb + 2;
    w->top[n] = (float *)malloc(sizeof(float) * (size_t)cap * N);
    w->log[n] = (float *)malloc(sizeof(float) * (size_t)cap);
    w->rows[n] = ro_repaint_section(w->d, ab, be, bb, bend, la, lb, n, NULL,
                                    w->top[n], w->log[n]);
  }
  free(ab); free(be);
  return NULL;
}

ro_window *ro_window_open(const ro_data *d, const char *paint_file, int snp,
                          int nthreads) {
  const int N = d->N, L = d->L;
  FILE *fp = fopen(paint_file, "rb");
  if (!fp) return NULL;
  fseek(fp, 0, SEEK_END);
  size_t len = (size_t)ftell(fp);
  fseek(fp, 0, SEEK_SET);
  unsigned char *buf = (unsigned char *)malloc(len);
  if (fread(buf, 1, len, fp) != len) { fclose(fp); free(buf); return NULL; }
  fclose(fp);

  ro_window *w = (ro_window *)calloc(1, sizeof(ro_window));
  w->d = d;
  w->rows = (int *)calloc((size_t)N, sizeof(int));
  w->top = (float **)calloc((size_t)N, sizeof(float *));
  w->log = (float **)calloc((size_t)N, sizeof(float *));
  w->v_snp_prev = (int *)calloc((size_t)N, sizeof(int));
  w->v_rpos_prev = (double *)calloc((size_t)N, sizeof(double));
  w->v_rpos_next = (double *)calloc((size_t)N, sizeof(double));

  /* index the N records (anc_builder.cpp:61-73) */
  size_t *off = (size_t *)malloc(sizeof(size_t) * (size_t)N);
  size_t pos = 0;
  int ok = 1;
  for (int n = 0; n < N && ok; n++) {
    off[n] = pos;
    if (pos + 8 > len) { ok = 0; break; }
    memcpy(&w->start, buf + pos, 4);
    memcpy(&w->end, buf + pos + 4, 4);
    pos += 8;
    for (int s = 0; s < 2; s++) {
      if (pos + 28 > len) { ok = 0; break; }
      int k;
      memcpy(&k, buf + pos + 24, 4);
      pos += 28 + (size_t)k * 8;
    }
  }
  if (!ok || pos > len) {
    free(off); free(buf); ro_window_free(w);
    return NULL;
  }
  if (nthreads < 1) nthreads = 1;
  if (nthreads > N) nthreads = N;
  repaint_job *jobs = (repaint_job *)calloc((size_t)nthreads, sizeof(repaint_job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int t = 0; t < nthreads; t++) {
    jobs[t].w = w; jobs[t].buf = buf; jobs[t].off = off; jobs[t].len = len;
    jobs[t].n0 = t; jobs[t].n1 = N; jobs[t].stride = nthreads;
    pthread_create(&th[t], NULL, repaint_worker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) {
    pthread_join(th[t], NULL);
    if (jobs[t].rc) ok = 0;
  }
  free(jobs); free(th); free(off); free(buf);
  if (!ok) { ro_window_free(w); return NULL; }

  /* cursors (anc_builder.cpp:81-101) */
  if (snp > 0) {
    for (int t = snp; t >= w->start; t--)
      for (int n = 0; n < N; n++)
        if (d->seq[(size_t)t * N + n] == '1') w->v_snp_prev[n]++;
  }
  for (int n = 0; n < N; n++) {
    int t = snp;
    while (d->seq[(size_t)t * N + n] != '1' && t > 0) t--;
    w->v_rpos_prev[n] = d->rpos[t];
    w->v_rpos_next[n] = w->v_rpos_prev[n];
  }
  (void)L;
  return w;
}

void ro_window_free(ro_window *w) {
  if (!w) return;
  if (w->top)
    for (int n = 0; n < w->d->N; n++) free(w->top[n]);
  if (w->log)
    for (int n = 0; n < w->d->N; n++) free(w->log[n]);
  free(w->top); free(w->log); free(w->rows);
  free(w->v_snp_prev); free(w->v_rpos_prev); free(w->v_rpos_next);
  free(w);
}

void ro_window_advance(ro_window *w, int snp) {
  const ro_data *d = w->d;
  const int N = d->N;
  for (int n = 0; n < N; n++) {
    if (d->seq[(size_t)snp * N + n] == '1') { /* anc_builder.cpp:489-494 */
      w->v_snp_prev[n]++;
      w->v_rpos_prev[n] = d->rpos[snp];
    }
  }
}

void ro_window_matrix(ro_window *w, int snp, float *matrix) {
  const ro_data *d = w->d;
  const int N = d->N, L = d->L;
  for (int n = 0; n < N; n++) {
    const int p = w->v_snp_prev[n];
    const int direct =
        (d->seq[(size_t)snp * N + n] == '1' || snp == 0 || snp == L - 1);
    if (direct) {
      ro_distance_row(N, n, 1, w->top[n] + (size_t)p * N, NULL, w->log[n][p],
                      0.0f, 0, 0, 0, matrix + (size_t)n * N);
    } else {
      if (w->v_rpos_next[n] <= w->v_rpos_prev[n]) { /* :134-141 */
        for (int l = snp; l < L; l++) {
          if (d->seq[(size_t)l * N + n] == '1' || l == L - 1) {
            w->v_rpos_next[n] = d->rpos[l];
            break;
          }
        }
      }
      ro_distance_row(N, n, 0, w->top[n] + (size_t)p * N,
                      w->top[n] + (size_t)(p + 1) * N, w->log[n][p],
                      w->log[n][p + 1], w->v_rpos_prev[n], w->v_rpos_next[n],
                      d->rpos[snp], matrix + (size_t)n * N);
    }
  }
}

int ro_window_rows(const ro_window *w, int n) { return w->rows[n]; }
const float *ro_window_top(const ro_window *w, int n) { return w->top[n]; }
const float *ro_window_log(const ro_window *w, int n) { return w->log[n]; }
int ro_window_start(const ro_window *w) { return w->start; }
int ro_window_end(const ro_window *w) { return w->end; }
