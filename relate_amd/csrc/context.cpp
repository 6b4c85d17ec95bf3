// context.cpp -- chunk state, visited-site plan and the Paint stage host code.
//
// Host counterpart of pipeline/Paint.cpp:17-108 and of the O(L) per-target
// precompute of fast_painting.cpp:41-157.  The precompute stays on the host
// on purpose: it needs glibc's exp() and costs O(N*L)
// additions against the kernels' O(N * sum_k D_k) updates.
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <mutex>
#include <string>
#include <chrono>
#include <cstring>
#include <numeric>
#include <thread>

#include <sched.h>

#include <map>

#include "common.h"

namespace rl {

static thread_local std::string g_err;
thread_local unsigned tl_alloc_failures = 0;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

// ranks of this job on this host (torch.distributed.run / mpirun export it): the host's cores are shared among them
int local_world_size() {
  for (const char *name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE"})
    if (const char *e = getenv(name))
      if (atoi(e) > 0) return atoi(e);
  return 1;
}
int local_rank() {
  for (const char *name : {"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK"})
    if (const char *e = getenv(name))
      if (atoi(e) >= 0) return atoi(e);
  return 0;
}

// host threads of this process: RELATE_AMD_THREADS, else this rank's share of the cores the process may run on
// ---- the memory cache (common.h)
namespace {
struct BlockCache {
  std::mutex m;
  std::multimap<std::pair<int, size_t>, void *> free_blocks;  // (device, bytes) -> block
  size_t held = 0;
};
BlockCache g_dev_cache, g_pin_cache;
}  // namespace
std::atomic<bool> g_keep_cache{false};
std::atomic<int> g_worker_launches{0};
namespace {

template <typename AllocFn>
void *cache_alloc(BlockCache &c, size_t bytes, size_t *got, AllocFn raw_alloc) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(c.m);
    auto it = c.free_blocks.lower_bound({dev, bytes});
    // (a block up to an eighth larger will do: the rows a window keeps differ a little from window to window)
    if (it != c.free_blocks.end() && it->first.first == dev && it->first.second <= bytes + bytes / 8 + 4096) {
      void *p = it->second;
      *got = it->first.second;
      c.held -= it->first.second;
      c.free_blocks.erase(it);
      return p;
    }
  }
  void *p = raw_alloc(bytes);
  // Out of memory: give the cached blocks back and try once more.  While the tree builder's workers are resident
  // hipFree would wait for them (for as long as any section has a tree in flight), so first ANY cached block that is
  // large enough will do -- the windows of a stage differ by up to 2 x in their rows, more than the eighth above --
  // and only then the trim, after the workers have left (they do, 50 ms after the last tree; 5 s at most here).
  if (!p) {
    {
      std::lock_guard<std::mutex> lk(c.m);
      auto it = c.free_blocks.lower_bound({dev, bytes});
      if (it != c.free_blocks.end() && it->first.first == dev) {
        void *q = it->second;
        *got = it->first.second;
        c.held -= it->first.second;
        c.free_blocks.erase(it);
        return q;
      }
    }
    // (workers leave 50 ms after the last tree: a few seconds cover a caller that follows a build; inside a busy
    //  stage they never leave, and the stage's admission loop is the one that waits -- treeseq.cpp)
    for (int waited = 0; g_worker_launches.load() > 0 && waited < 50; waited++)
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    if (g_worker_launches.load() == 0) {
      device_cache_trim();
      p = raw_alloc(bytes);
    }
  }
  *got = bytes;
  return p;
}
void cache_release(BlockCache &c, void *p, size_t bytes) {
  int dev = 0;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) == hipSuccess) dev = attr.device;
  std::lock_guard<std::mutex> lk(c.m);
  c.free_blocks.emplace(std::make_pair(dev, bytes), p);
  c.held += bytes;
}
}  // namespace

void *device_cache_alloc(size_t bytes, size_t *got) {
  return cache_alloc(g_dev_cache, bytes, got, [](size_t n) -> void * {
    void *p = nullptr;
    return hipMalloc(&p, n) == hipSuccess ? p : nullptr;
  });
}
void device_cache_release(void *p, size_t bytes) { cache_release(g_dev_cache, p, bytes); }
size_t device_cache_held(int device, size_t min_block) {
  std::lock_guard<std::mutex> lk(g_dev_cache.m);
  size_t sum = 0;
  for (auto it = g_dev_cache.free_blocks.lower_bound({device, min_block});
       it != g_dev_cache.free_blocks.end() && it->first.first == device; ++it)
    sum += it->first.second;
  return sum;
}
void *pinned_cache_alloc(size_t bytes, size_t *got) {
  return cache_alloc(g_pin_cache, bytes, got, [](size_t n) -> void * {
    void *p = nullptr;
    return hipHostMalloc(&p, n, 0) == hipSuccess ? p : nullptr;
  });
}
void pinned_cache_release(void *p, size_t bytes) { cache_release(g_pin_cache, p, bytes); }
void device_cache_trim() {
  std::vector<void *> dev_blocks, pin_blocks;
  {
    std::lock_guard<std::mutex> lk(g_dev_cache.m);
    for (auto &b : g_dev_cache.free_blocks) dev_blocks.push_back(b.second);
    g_dev_cache.free_blocks.clear();
    g_dev_cache.held = 0;
  }
  {
    std::lock_guard<std::mutex> lk(g_pin_cache.m);
    for (auto &b : g_pin_cache.free_blocks) pin_blocks.push_back(b.second);
    g_pin_cache.free_blocks.clear();
    g_pin_cache.held = 0;
  }
  for (void *p : dev_blocks) (void)hipFree(p);
  for (void *p : pin_blocks) (void)hipHostFree(p);
}

int host_threads() {
  const char *e = getenv("RELATE_AMD_THREADS");
  int n = 0;
  if (e) {
    n = atoi(e);
  } else {
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    n = sched_getaffinity(0, sizeof(allowed), &allowed) == 0 ? CPU_COUNT(&allowed)
                                                             : (int)std::thread::hardware_concurrency();
    n /= local_world_size();
  }
  if (n < 1) n = 1;
  if (n > 256) n = 256;
  return n;
}

template <typename F>
static void parallel_for(int n, F f) {
  int T = std::min(host_threads(), std::max(1, n));
  if (T == 1) {
    for (int i = 0; i < n; i++) f(i);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++)
    th.emplace_back([=]() {
      for (int i = t; i < n; i += T) f(i);
    });
  for (auto &x : th) x.join();
}

// FastPainting::FastPainting (fast_painting.hpp:26-39)
PaintConsts make_consts(int N, double theta) {
  PaintConsts c;
  const double ntheta = 1.0 - theta;
  const double Nm1 = N - 1.0;
  c.theta = theta;
  c.ntheta = ntheta;
  const double prior_theta = theta / Nm1 - ntheta / Nm1;
  const double prior_ntheta = ntheta / Nm1;
  const double theta_ratio = theta / (1.0 - theta) - 1.0;
  c.K1 = theta_ratio + 1.0;
  c.init0 = prior_ntheta;
  c.init1 = prior_theta + prior_ntheta;
  c.log_Nm1 = std::log(Nm1);
  c.log_ntheta = std::log(ntheta);
  c.lower = 1e-10;
  c.upper = 1.0 / c.lower;
  c.inv_theta = 1.0 / theta;  // (IEEE division: correctly rounded, which div_by_const relies on)
  c.inv_ntheta = 1.0 / ntheta;
  return c;
}

void interval_coeffs(const PaintConsts &c, int N, double rho, double *cf, double *nxt) {
  double nx = -rho + c.log_ntheta;
  double rp = 1.0 - std::exp(-rho);
  if (rp > 0.99) {
    rp = 0.99;
    nx = std::log(0.01) + c.log_ntheta;
  }
  *nxt = nx;
  *cf = rp / ((1.0 - rp) * (N - 1.0));
}

// ---------------------------------------------------------------------------
// Visited-site plan for every target.
int build_plan(rl_ctx *ctx) {
  const int N = ctx->N, L = ctx->L, W = ctx->W, rw = ctx->row_words;
  Plan &pl = ctx->plan;
  const uint32_t *bits = ctx->bits.data();
  ctx->consts = make_consts(N, ctx->theta);
  const PaintConsts c = ctx->consts;
  std::vector<double> r(ctx->r);
  if (ctx->rho != 1.0)
    for (auto &x : r) x *= ctx->rho;  // Paint.cpp:57-59

  // 1. counts of derived sites per target over SNPs 1..L-2
  std::vector<int64_t> cnt((size_t)N, 0);
  const int words = (N + 31) / 32;
  parallel_for(words, [&](int w) {
    int64_t local[32] = {0};
    for (int s = 1; s < L - 1; s++) {
      uint32_t x = bits[(size_t)s * rw + w];
      while (x) {
        int b = __builtin_ctz(x);
        local[b]++;
        x &= x - 1;
      }
    }
    for (int b = 0; b < 32 && w * 32 + b < N; b++) cnt[w * 32 + b] = local[b];
  });
  // only the targets of this context (rl_set_target_range) get a list
  const int k0 = ctx->k0, k1 = ctx->k0 + ctx->nloc;
  pl.off.assign((size_t)N + 1, 0);
  for (int k = 0; k < N; k++) pl.off[k + 1] = pl.off[k] + ((k >= k0 && k < k1) ? cnt[k] + 2 : 0);
  const int64_t total = pl.off[N];
  pl.sites.assign((size_t)total, 0);
  pl.cf.assign((size_t)total, 0.0);
  pl.nxt.assign((size_t)total, 0.0);
  pl.ia.assign((size_t)N * W, 0);
  pl.ie.assign((size_t)N * W, 0);
  pl.bb.assign((size_t)N * W, 0);
  pl.be.assign((size_t)N * W, 0);
  pl.binit.assign((size_t)N, 0.0);

  // 2. visited-site lists (site | flag "target derived here")
  parallel_for(words, [&](int w) {
    int64_t pos[32];
    uint32_t mine = 0;  // targets of this word that belong to this context
    for (int b = 0; b < 32 && w * 32 + b < N; b++) {
      int k = w * 32 + b;
      if (k < k0 || k >= k1) continue;
      mine |= 1u << b;
      pos[b] = pl.off[k];
      uint32_t x0 = bits[w];
      pl.sites[pos[b]++] = 0 | (((x0 >> b) & 1u) ? (int32_t)0x80000000 : 0);
    }
    if (!mine) return;
    for (int s = 1; s < L - 1; s++) {
      uint32_t x = bits[(size_t)s * rw + w] & mine;
      while (x) {
        int b = __builtin_ctz(x);
        pl.sites[pos[b]++] = s | (int32_t)0x80000000;
        x &= x - 1;
      }
    }
    uint32_t xl = bits[(size_t)(L - 1) * rw + w];
    for (int b = 0; b < 32 && w * 32 + b < N; b++)
      if ((mine >> b) & 1u) pl.sites[pos[b]++] = (L - 1) | (((xl >> b) & 1u) ? (int32_t)0x80000000 : 0);
  });

  // 3. interval coefficients, window boundaries, initial beta sum
  const std::vector<int> &wb = ctx->wb;
  parallel_for(N, [&](int k) {
    if (k < k0 || k >= k1) return;
    const int64_t o = pl.off[k];
    const int D = (int)(pl.off[k + 1] - o);
    int32_t *bb = &pl.bb[(size_t)k * W], *be = &pl.be[(size_t)k * W];
    int32_t *ia = &pl.ia[(size_t)k * W], *ie = &pl.ie[(size_t)k * W];
    int pb = 0, pe = 0, window_index = 1, window_end = wb[1];
    bb[pb] = 0;
    ia[pb++] = 0;
    for (int i = 0; i < D; i++) {
      const int s0 = pl.sites[o + i] & 0x7fffffff;
      double acc = r[s0];
      if (i + 1 < D) {
        const int s1 = pl.sites[o + i + 1] & 0x7fffffff;
        for (int s = s0 + 1; s < s1; s++) acc += r[s];  // :56-59, :93-96
        if (s1 >= window_end && s0 < window_end) {      // :60-69, :98-107
          while (window_end <= s1) {
            be[pe] = s1;
            ie[pe++] = i + 1;
            bb[pb] = s0;
            ia[pb++] = i;
            window_index++;
            window_end = wb[window_index];
          }
        }
      }
      interval_coeffs(c, N, acc, &pl.cf[o + i], &pl.nxt[o + i]);
    }
    be[pe] = L - 1;
    ie[pe++] = D - 1;
    // beta_sum at the last SNP: serial over n, including n == k (:421-431)
    const bool seqk = pl.sites[o + D - 1] < 0;
    const uint32_t *row = bits + (size_t)(L - 1) * rw;
    double B = 0.0;
    for (int n = 0; n < N; n++) {
      const bool dn = (row[n >> 5] >> (n & 31)) & 1u;
      if (seqk && !dn)
        B += c.theta;
      else
        B += c.ntheta;
    }
    B -= c.ntheta;
    pl.binit[k] = B;
  });

  // 4. launch order: longest target first
  pl.order.resize(ctx->nloc);
  std::iota(pl.order.begin(), pl.order.end(), k0);
  std::stable_sort(pl.order.begin(), pl.order.end(), [&](int a, int b) {
    return (pl.off[a + 1] - pl.off[a]) > (pl.off[b + 1] - pl.off[b]);
  });
  pl.valid = true;
  ctx->plan_on_device = false;
  return RL_OK;
}

int upload_plan(rl_ctx *ctx) {
  if (ctx->plan_on_device) return RL_OK;
  if (!ctx->plan.valid) {
    int rc = build_plan(ctx);
    if (rc) return rc;
  }
  Plan &pl = ctx->plan;
  int rc;
  if ((rc = ctx->d_bits.upload(ctx->bits))) return rc;
  // lane-mask form of the panel: `waves` rows of S words per site
  if ((rc = ctx->d_masks.alloc(((size_t)ctx->L + 2) * ctx->waves * ctx->S * sizeof(unsigned long long)))) return rc;
  RL_HIP(launch_lane_masks(ctx->d_bits.as<uint32_t>(), ctx->row_words, ctx->L, ctx->lay, ctx->S, ctx->waves,
                           ctx->d_masks.as<unsigned long long>(), nullptr));
  RL_HIP(hipDeviceSynchronize());
  if ((rc = ctx->d_off.upload(pl.off))) return rc;
  if ((rc = ctx->d_sites.upload(pl.sites))) return rc;
  if ((rc = ctx->d_cf.upload(pl.cf))) return rc;
  if ((rc = ctx->d_nxt.upload(pl.nxt))) return rc;
  if ((rc = ctx->d_ia.upload(pl.ia))) return rc;
  if ((rc = ctx->d_ie.upload(pl.ie))) return rc;
  if ((rc = ctx->d_binit.upload(pl.binit))) return rc;
  if ((rc = ctx->d_order.upload(pl.order))) return rc;
  ctx->plan_on_device = true;
  return RL_OK;
}

// ---------------------------------------------------------------------------
// paint-file codec
size_t encode_stone(const float *v, int N, int bsnp, float logscale, unsigned char *out) {
  // collapsed_matrix.hpp:228-265: runs merge while |first - v| < 1e-3*min(first, v)
  unsigned char *p = out;
  const uint64_t isize = 1, isub = (uint64_t)N;
  memcpy(p, &isize, 8); p += 8;
  memcpy(p, &isub, 8); p += 8;
  memcpy(p, &bsnp, 4); p += 4;
  memcpy(p, &logscale, 4); p += 4;
  unsigned char *pk = p;
  p += 4;
  float *uniq = reinterpret_cast<float *>(p);  // 4-byte aligned: 28-byte header
  std::vector<int> times;
  times.reserve(64);
  float current = v[0];
  int k = 0;
  memcpy(&uniq[0], &current, 4);
  times.push_back(1);
  for (int j = 1; j < N; j++) {
    const float diff = std::fabs(current - v[j]);
    const float mn = std::min(current, v[j]);
    if (diff < 1e-3 * mn) {
      times[k]++;
    } else {
      current = v[j];
      k++;
      memcpy(&uniq[k], &current, 4);
      times.push_back(1);
    }
  }
  k++;
  memcpy(pk, &k, 4);
  p += (size_t)k * 4;
  memcpy(p, times.data(), (size_t)k * 4);
  p += (size_t)k * 4;
  return (size_t)(p - out);
}

size_t decode_stone(const unsigned char *in, size_t avail, int N, float *v, int *bsnp, float *logscale) {
  // collapsed_matrix.hpp:268-296
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
  const unsigned char *pu = in + 28, *pt = pu + (size_t)k * 4;
  int i = 0;
  for (int j = 0; j < k; j++) {
    float u;
    int t;
    memcpy(&u, pu + (size_t)j * 4, 4);
    memcpy(&t, pt + (size_t)j * 4, 4);
    if (t < 0 || i + t > N) return 0;
    for (int q = 0; q < t; q++) v[i++] = u;
  }
  if (i != N) return 0;
  return 28 + (size_t)k * 8;
}

float fast_log_host(float val) {
  int32_t x;
  memcpy(&x, &val, 4);
  const int log_2 = ((x >> 23) & 255) - 128;
  x &= ~(255 << 23);
  x += 127 << 23;
  memcpy(&val, &x, 4);
  val = ((-1.0f / 3) * val + 2) * val - 2.0f / 3;
  return (val + log_2) * 0.69314718f;
}

static int read_all(const std::string &fn, std::vector<unsigned char> &buf) {
  FILE *fp = fopen(fn.c_str(), "rb");
  if (!fp) {
    set_error("cannot open %s", fn.c_str());
    return RL_EIO;
  }
  fseek(fp, 0, SEEK_END);
  long len = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  buf.resize((size_t)len);
  size_t got = len ? fread(buf.data(), 1, (size_t)len, fp) : 0;
  fclose(fp);
  if (got != (size_t)len) {
    set_error("short read on %s", fn.c_str());
    return RL_EIO;
  }
  return RL_OK;
}

template <typename T>
static int read_vec(const std::string &fn, std::vector<T> &v) {
  // Data::ReadVectorFromBin (data.hpp:90-99): u32 size, then size elements
  std::vector<unsigned char> buf;
  int rc = read_all(fn, buf);
  if (rc) return rc;
  if (buf.size() < 4) {
    set_error("%s: truncated", fn.c_str());
    return RL_EFORMAT;
  }
  uint32_t n;
  memcpy(&n, buf.data(), 4);
  if (buf.size() < 4 + (size_t)n * sizeof(T)) {
    set_error("%s: truncated (%u elements expected)", fn.c_str(), n);
    return RL_EFORMAT;
  }
  v.resize(n);
  memcpy(v.data(), buf.data() + 4, (size_t)n * sizeof(T));
  return RL_OK;
}

static int mkdir_p(const std::string &d) {
  // filesys::MakeDir (filesystem.cpp:4-23): mkdir 0700 if absent
  struct stat st;
  if (stat(d.c_str(), &st) == 0) return RL_OK;
  if (mkdir(d.c_str(), 0700) != 0 && stat(d.c_str(), &st) != 0) {
    set_error("cannot create directory %s", d.c_str());
    return RL_EIO;
  }
  return RL_OK;
}

}  // namespace rl

using namespace rl;

extern "C" {

const char *rl_last_error(void) { return g_err.c_str(); }
const char *rl_version(void) { return "relate_amd 0.1 (gfx950)"; }

int rl_device_count(void) {
  // The stage keeps a stream per open window and per tree builder; HIP maps them onto GPU_MAX_HW_QUEUES hardware
  // queues (4 by default: 91 s -> 96 s for 80 sections; past 16 the device time-slices them, DESIGN_NOTES.md 6).  Read by
  // the runtime when it initialises, so set before the first HIP call of the process; the user's value wins.
  setenv("GPU_MAX_HW_QUEUES", "12", 0);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void rl_keep_cache_until_exit(int on) { rl::g_keep_cache.store(on != 0); }

rl_ctx *rl_create(int device) {
  int n = rl_device_count();
  if (n <= 0 || device < 0 || device >= n) {
    set_error("no usable HIP device (visible devices: %d, requested %d)", n, device);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    set_error("hipSetDevice(%d) failed", device);
    return nullptr;
  }
  rl_ctx *ctx = new rl_ctx();
  ctx->device = device;
  if (make_stream(&ctx->s0, false) != hipSuccess ||
      make_stream(&ctx->s1, false) != hipSuccess ||
      hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
      hipEventCreate(&ctx->ev2) != hipSuccess) {
    set_error("stream/event creation failed");
    delete ctx;
    return nullptr;
  }
  return ctx;
}

void rl_destroy(rl_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->s0) (void)hipStreamDestroy(ctx->s0);
  if (ctx->s1) (void)hipStreamDestroy(ctx->s1);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
  if (ctx->lane2.s) (void)hipStreamDestroy(ctx->lane2.s);
  if (ctx->lane2.e0) (void)hipEventDestroy(ctx->lane2.e0);
  if (ctx->lane2.e1) (void)hipEventDestroy(ctx->lane2.e1);
  if (ctx->h_alpha) (void)hipHostFree(ctx->h_alpha);
  if (ctx->h_beta) (void)hipHostFree(ctx->h_beta);
  delete ctx;
  // (the context's buffers went to the cache: back to the driver -- unless the process says it is about to end,
  //  rl_keep_cache_until_exit, the command-line tool: the blocks stay reusable, and the exit releases them at once)
  if (!g_keep_cache.load())
    rl::device_cache_trim();
  else
    (void)hipDeviceSynchronize();  // (as the trim's hipFree does: the tree builder's idle workers have left when this returns)
}

// The stones of a painted chunk move to pinned host memory and their device buffers are released (C3: 2 x 26.7 GB,
// a quarter of the sections the stage can keep open).  Not an error if the host has no room: they stay where they are.
int rl_park_stones(rl_ctx *ctx) {
  if (!ctx || !ctx->painted || ctx->h_alpha || !ctx->d_alpha.p) return RL_OK;
  (void)hipSetDevice(ctx->device);
  const size_t bytes = (size_t)(ctx->wb.size() - 1) * ctx->nloc * ctx->N * sizeof(float);
  float *a = nullptr, *b = nullptr;
  if (hipHostMalloc(reinterpret_cast<void **>(&a), bytes, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void **>(&b), bytes, hipHostMallocDefault) != hipSuccess) {
    if (a) (void)hipHostFree(a);
    (void)hipGetLastError();
    return RL_OK;
  }
  if (hipMemcpy(a, ctx->d_alpha.p, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(b, ctx->d_beta.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipHostFree(a);
    (void)hipHostFree(b);
    set_error("copying the stepping stones to the host failed");
    return RL_EHIP;
  }
  ctx->h_alpha = a;
  ctx->h_beta = b;
  ctx->d_alpha.release();
  ctx->d_beta.release();
  rl::device_cache_trim();  // (released blocks wait in the cache: these are to be someone else's memory)
  return RL_OK;
}

// Room for the stepping stones of a Paint: in HBM -- unless the two buffers would take most of what is free (config
// #5 with all targets on one GPU: 2 x 144 GB): then they are painted straight into pinned host memory, the kernel's
// stores crossing PCIe (seconds, once per chunk), and every window of the stage copies its slice in when it opens,
// as it does for stones parked by rl_park_stones.  (The stores are the only traffic: nothing of the stones is read
// back by the painting.)
static int alloc_stones(rl_ctx *ctx) {
  const size_t bytes = (size_t)ctx->W * ctx->nloc * ctx->N * sizeof(float);
  if (ctx->h_alpha) {  // (stones of an earlier pass parked on the host)
    (void)hipHostFree(ctx->h_alpha);
    (void)hipHostFree(ctx->h_beta);
    ctx->h_alpha = ctx->h_beta = nullptr;
  }
  size_t free_b = 0, total_b = 0;
  const bool have = ctx->d_alpha.p && ctx->d_alpha.bytes >= bytes && ctx->d_beta.p && ctx->d_beta.bytes >= bytes;
  if (!have && hipMemGetInfo(&free_b, &total_b) == hipSuccess && 2.0 * (double)bytes > 0.6 * (double)free_b) {
    ctx->d_alpha.release();
    ctx->d_beta.release();
    float *a = nullptr, *b = nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&a), bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&b), bytes, hipHostMallocDefault) != hipSuccess) {
      if (a) (void)hipHostFree(a);
      (void)hipGetLastError();
      set_error("the stepping stones (2 x %.1f GB) fit neither the device nor pinned host memory", 1e-9 * (double)bytes);
      return RL_ENOMEM;
    }
    ctx->h_alpha = a;
    ctx->h_beta = b;
    return RL_OK;
  }
  int rc = ctx->d_alpha.alloc(bytes);
  return rc ? rc : ctx->d_beta.alloc(bytes);
}

static int set_common(rl_ctx *ctx, int N, int L, const double *r, const double *rpos, const int *wb, int W) {
  if (!ctx || N < 2 || L < 2 || W < 1 || !r || !rpos || !wb) {
    set_error("rl_set_chunk: bad arguments");
    return RL_EINVAL;
  }
  if (wb[0] != 0 || wb[W] != L) {
    set_error("window boundaries must start at 0 and end at L=%d", L);
    return RL_EINVAL;
  }
  for (int w = 0; w < W; w++)
    if (wb[w + 1] <= wb[w]) {
      set_error("window boundaries must be increasing");
      return RL_EINVAL;
    }
  // a workgroup of two waves per target once one wave would need more than 80 registers per lane
  const int waves = target_waves(N);
  Layout lay = make_layout(N, waves);
  int S = choose_S(lay);
  if (S == 0 || waves > 2) {
    set_error("N=%d exceeds the largest compiled register tile (N <= %d)", N, 2 * 80 * 64);
    return RL_EINVAL;
  }
  ctx->N = N; ctx->L = L; ctx->W = W; ctx->lay = lay; ctx->S = S; ctx->waves = waves;
  ctx->k0 = 0; ctx->nloc = N;  // all targets until rl_set_target_range
  ctx->r.assign(r, r + L);
  ctx->rpos.assign(rpos, rpos + L + 1);
  ctx->wb.assign(wb, wb + W + 1);
  ctx->row_words = ((N + 31) / 32 + 3 + 3) & ~3;  // >= 3 words of zero slack, 16-byte rows
  ctx->theta = 0.001;  // data.cpp:95
  ctx->rho = 1.0;
  ctx->plan.valid = false;
  ctx->plan_on_device = false;
  ctx->painted = false;
  ctx->have_chunk = true;
  return RL_OK;
}

int rl_set_chunk(rl_ctx *ctx, int N, int L, const uint8_t *seq, const double *r, const double *rpos,
                 const int *wb, int W) {
  if (!seq) {
    set_error("rl_set_chunk: seq is NULL");
    return RL_EINVAL;
  }
  int rc = set_common(ctx, N, L, r, rpos, wb, W);
  if (rc) return rc;
  const int rw = ctx->row_words;
  ctx->bits.assign((size_t)L * rw, 0u);
  uint32_t *bits = ctx->bits.data();
  parallel_for(L, [&](int s) {
    const uint8_t *row = seq + (size_t)s * N;
    uint32_t *o = bits + (size_t)s * rw;
    for (int n = 0; n < N; n++)
      if (row[n] == '1') o[n >> 5] |= 1u << (n & 31);
  });
  return RL_OK;
}

int rl_set_chunk_bits(rl_ctx *ctx, int N, int L, const uint32_t *bits, int row_words, const double *r,
                      const double *rpos, const int *wb, int W) {
  if (!bits || row_words < (N + 31) / 32) {
    set_error("rl_set_chunk_bits: bad panel");
    return RL_EINVAL;
  }
  int rc = set_common(ctx, N, L, r, rpos, wb, W);
  if (rc) return rc;
  const int rw = ctx->row_words, words = (N + 31) / 32;
  ctx->bits.assign((size_t)L * rw, 0u);
  uint32_t *o = ctx->bits.data();
  const uint32_t lastmask = (N & 31) ? ((1u << (N & 31)) - 1u) : 0xffffffffu;
  parallel_for(L, [&](int s) {
    memcpy(o + (size_t)s * rw, bits + (size_t)s * row_words, (size_t)words * 4);
    o[(size_t)s * rw + words - 1] &= lastmask;
  });
  return RL_OK;
}

int rl_load_chunk(rl_ctx *ctx, const char *dir, int chunk_index) {
  if (!ctx || !dir) {
    set_error("rl_load_chunk: bad arguments");
    return RL_EINVAL;
  }
  const std::string d(dir), c = std::to_string(chunk_index);
  // parameters_c<c>.bin: int N, L, W+1, wb[W+1]  (Paint.cpp:23-31)
  std::vector<unsigned char> pbuf;
  int rc = read_all(d + "/parameters_c" + c + ".bin", pbuf);
  if (rc) return rc;
  if (pbuf.size() < 12) {
    set_error("parameters file truncated");
    return RL_EFORMAT;
  }
  int N, L, nw;
  memcpy(&N, pbuf.data(), 4);
  memcpy(&L, pbuf.data() + 4, 4);
  memcpy(&nw, pbuf.data() + 8, 4);
  if (nw < 2 || pbuf.size() < 12 + (size_t)nw * 4) {
    set_error("parameters file malformed");
    return RL_EFORMAT;
  }
  std::vector<int> wb(nw);
  memcpy(wb.data(), pbuf.data() + 12, (size_t)nw * 4);
  std::vector<double> r, rpos;
  if ((rc = read_vec(d + "/chunk_" + c + ".r", r))) return rc;
  if ((rc = read_vec(d + "/chunk_" + c + ".rpos", rpos))) return rc;
  if ((int)r.size() != L || (int)rpos.size() != L + 1) {
    set_error(".r/.rpos sizes disagree with L=%d", L);
    return RL_EFORMAT;
  }
  // chunk_<c>.bits: the panel bit-packed, written by this library's MakeChunks (RELATE_AMD_CHUNK_BITS=1) next to the
  // reference's files (makechunks.cpp) -- an eighth of the .hap file, and already in the layout the device works on.
  // It names the .hap it belongs to by size and modification time: next to any other .hap it is stale and ignored.
  {
    FILE *fp = fopen((d + "/chunk_" + c + ".bits").c_str(), "rb");
    if (fp) {
      uint32_t head[4] = {0, 0, 0, 0};
      uint64_t of_hap[2] = {0, 0};
      std::vector<uint32_t> words;
      struct stat st;
      bool ok = fread(head, 4, 4, fp) == 4 && fread(of_hap, 8, 2, fp) == 2 && head[0] == 0x32424c52u &&
                (int)head[1] == N && (int)head[2] == L && head[3] >= (uint32_t)((N + 31) / 32) &&
                head[3] <= (uint32_t)((N + 31) / 32) + 64 && stat((d + "/chunk_" + c + ".hap").c_str(), &st) == 0 &&
                (uint64_t)st.st_size == of_hap[0] &&
                (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec == of_hap[1];
      if (ok) {
        words.resize((size_t)L * head[3]);
        ok = fread(words.data(), 4, words.size(), fp) == words.size();
      }
      fclose(fp);
      if (ok) return rl_set_chunk_bits(ctx, N, L, words.data(), (int)head[3], r.data(), rpos.data(), wb.data(), nw - 1);
      // (a stale or foreign file: the reference's own input decides)
    }
  }
  // chunk_<c>.hap: u64 L, u64 N, L*N chars (collapsed_matrix.hpp:204-225)
  std::vector<unsigned char> hap;
  rc = read_all(d + "/chunk_" + c + ".hap", hap);
  if (rc) return rc;
  uint64_t hL, hN;
  if (hap.size() < 16) {
    set_error(".hap truncated");
    return RL_EFORMAT;
  }
  memcpy(&hL, hap.data(), 8);
  memcpy(&hN, hap.data() + 8, 8);
  if ((int)hL != L || (int)hN != N || hap.size() < 16 + (size_t)L * N) {
    set_error(".hap dimensions (%llu x %llu) disagree with parameters (%d x %d)",
              (unsigned long long)hL, (unsigned long long)hN, L, N);
    return RL_EFORMAT;
  }
  return rl_set_chunk(ctx, N, L, hap.data() + 16, r.data(), rpos.data(), wb.data(), nw - 1);
}

int rl_set_painting(rl_ctx *ctx, double theta, double rho) {
  if (!ctx || !ctx->have_chunk) {
    set_error("rl_set_painting: no chunk loaded");
    return RL_ESTATE;
  }
  if (!(theta > 0.0 && theta < 1.0)) {
    set_error("theta must be in (0,1)");
    return RL_EINVAL;
  }
  ctx->theta = theta;
  ctx->rho = rho;
  ctx->plan.valid = false;
  ctx->plan_on_device = false;
  ctx->painted = false;
  return RL_OK;
}

int rl_set_target_range(rl_ctx *ctx, int k_begin, int k_end) {
  if (!ctx || !ctx->have_chunk) {
    set_error("rl_set_target_range: no chunk loaded");
    return RL_ESTATE;
  }
  if (k_begin < 0 || k_end > ctx->N || k_begin >= k_end) {
    set_error("rl_set_target_range: [%d, %d) is not a range of the %d haplotypes", k_begin, k_end, ctx->N);
    return RL_EINVAL;
  }
  ctx->k0 = k_begin;
  ctx->nloc = k_end - k_begin;
  ctx->plan.valid = false;
  ctx->plan_on_device = false;
  ctx->painted = false;
  return RL_OK;
}

int rl_target_range(const rl_ctx *ctx, int *k_begin, int *k_end) {
  if (!ctx || !ctx->have_chunk) {
    set_error("no chunk loaded");
    return RL_ESTATE;
  }
  if (k_begin) *k_begin = ctx->k0;
  if (k_end) *k_end = ctx->k0 + ctx->nloc;
  return RL_OK;
}

int rl_chunk_dims(const rl_ctx *ctx, int *N, int *L, int *W) {
  if (!ctx || !ctx->have_chunk) {
    set_error("no chunk loaded");
    return RL_ESTATE;
  }
  if (N) *N = ctx->N;
  if (L) *L = ctx->L;
  if (W) *W = ctx->W;
  return RL_OK;
}

long long rl_total_sites(rl_ctx *ctx) {
  if (!ctx || !ctx->have_chunk) return RL_ESTATE;
  if (!ctx->plan.valid && build_plan(ctx)) return RL_EINVAL;
  return ctx->plan.off[ctx->N];
}

int rl_prepare(rl_ctx *ctx) {
  if (!ctx || !ctx->have_chunk) {
    set_error("rl_prepare: no chunk loaded");
    return RL_ESTATE;
  }
  RL_HIP(hipSetDevice(ctx->device));
  int rc = upload_plan(ctx);
  if (rc) return rc;
  const size_t N = ctx->N, W = ctx->W, nloc = ctx->nloc;  // stones: this context's target rows only
  if ((rc = alloc_stones(ctx))) return rc;
  if ((rc = ctx->d_lsa.alloc(W * nloc * sizeof(float)))) return rc;
  if ((rc = ctx->d_lsb.alloc(W * nloc * sizeof(float)))) return rc;
  RL_HIP(hipDeviceSynchronize());
  return RL_OK;
}

int rl_paint(rl_ctx *ctx, int sum_mode, float *kernel_ms) {
  if (!ctx || !ctx->have_chunk) {
    set_error("rl_paint: no chunk loaded");
    return RL_ESTATE;
  }
  if (sum_mode != RL_SUM_EXACT && sum_mode != RL_SUM_LANES && sum_mode != RL_SUM_EXACT_SERIAL && sum_mode != RL_SUM_LANES32) {
    set_error("rl_paint: bad sum_mode");
    return RL_EINVAL;
  }
  RL_HIP(hipSetDevice(ctx->device));
  int rc = upload_plan(ctx);
  if (rc) return rc;
  const size_t N = ctx->N, W = ctx->W, nloc = ctx->nloc;  // stones: this context's target rows only
  if ((rc = alloc_stones(ctx))) return rc;
  if ((rc = ctx->d_lsa.alloc(W * nloc * sizeof(float)))) return rc;
  if ((rc = ctx->d_lsb.alloc(W * nloc * sizeof(float)))) return rc;

  PaintParams p;
  p.lay = ctx->lay;
  p.c = ctx->consts;
  p.L = ctx->L;
  p.W = ctx->W;
  p.k0 = ctx->k0;
  p.nloc = ctx->nloc;
  p.S = ctx->S;
  p.masks = ctx->d_masks.as<unsigned long long>();
  p.plan_off = ctx->d_off.as<int64_t>();
  p.sites = ctx->d_sites.as<int32_t>();
  p.cf = ctx->d_cf.as<double>();
  p.nxt = ctx->d_nxt.as<double>();
  p.stone_ia = ctx->d_ia.as<int32_t>();
  p.stone_ie = ctx->d_ie.as<int32_t>();
  p.binit = ctx->d_binit.as<double>();
  p.order = ctx->d_order.as<int32_t>();
  p.alpha = ctx->h_alpha ? ctx->h_alpha : ctx->d_alpha.as<float>();
  p.beta = ctx->h_alpha ? ctx->h_beta : ctx->d_beta.as<float>();
  p.ls_alpha = ctx->d_lsa.as<float>();
  p.ls_beta = ctx->d_lsb.as<float>();
  p.sum_mode = sum_mode;
  p.merge_order = 0;
  p.stats = nullptr;
  if (getenv("RELATE_AMD_TEST_STATS")) {  // experiment builds (-DRL_STATS): 16 counters, see tools/exp_stats.py
    if ((rc = ctx->d_stats.alloc(32 * sizeof(unsigned long long)))) return rc;
    RL_HIP(hipMemset(ctx->d_stats.p, 0, 32 * sizeof(unsigned long long)));
    p.stats = ctx->d_stats.as<unsigned long long>();
  }

  if (ctx->paint_split == 2) {
    // experiment: the two directions as two launches on two streams
    RL_HIP(hipEventRecord(ctx->ev0, ctx->s0));
    RL_HIP(hipStreamWaitEvent(ctx->s1, ctx->ev0, 0));
    RL_HIP(launch_paint(p, ctx->S, ctx->waves, 1, ctx->s0));
    RL_HIP(launch_paint(p, ctx->S, ctx->waves, 0, ctx->s1));
    RL_HIP(hipEventRecord(ctx->ev1, ctx->s1));
    RL_HIP(hipStreamWaitEvent(ctx->s0, ctx->ev1, 0));
    RL_HIP(hipEventRecord(ctx->ev2, ctx->s0));
    RL_HIP(hipEventSynchronize(ctx->ev2));
    RL_HIP(hipEventElapsedTime(&ctx->ms_paint, ctx->ev0, ctx->ev2));
    ctx->ms_bwd = ctx->ms_fwd = 0.f;
  } else if (ctx->paint_split) {
    // one direction per launch, backward then forward on one stream, each bracketed by HIP events
    RL_HIP(hipEventRecord(ctx->ev0, ctx->s0));
    RL_HIP(launch_paint(p, ctx->S, ctx->waves, 1, ctx->s0));
    RL_HIP(hipEventRecord(ctx->ev1, ctx->s0));
    RL_HIP(launch_paint(p, ctx->S, ctx->waves, 0, ctx->s0));
    RL_HIP(hipEventRecord(ctx->ev2, ctx->s0));
    RL_HIP(hipEventSynchronize(ctx->ev2));
    RL_HIP(hipEventElapsedTime(&ctx->ms_bwd, ctx->ev0, ctx->ev1));
    RL_HIP(hipEventElapsedTime(&ctx->ms_fwd, ctx->ev1, ctx->ev2));
    ctx->ms_paint = ctx->ms_bwd + ctx->ms_fwd;
  } else {
    // both directions in one launch of 2 * nloc workgroups (paint_kernels.hip)
    RL_HIP(hipEventRecord(ctx->ev0, ctx->s0));
    RL_HIP(launch_paint(p, ctx->S, ctx->waves, 2, ctx->s0));
    RL_HIP(hipEventRecord(ctx->ev2, ctx->s0));
    RL_HIP(hipEventSynchronize(ctx->ev2));
    RL_HIP(hipEventElapsedTime(&ctx->ms_paint, ctx->ev0, ctx->ev2));
    ctx->ms_bwd = ctx->ms_fwd = 0.f;
  }
  if (kernel_ms) *kernel_ms = ctx->ms_paint;
  ctx->painted = true;
  ctx->paint_mode = sum_mode;
  return RL_OK;
}

int rl_debug_stats(rl_ctx *ctx, unsigned long long *out16) {
  if (!ctx || !ctx->d_stats.p || !out16) return RL_ESTATE;
  RL_HIP(hipMemcpy(out16, ctx->d_stats.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return RL_OK;
}

int rl_debug_stats32(rl_ctx *ctx, unsigned long long *out32) {
  if (!ctx || !ctx->d_stats.p || !out32) return RL_ESTATE;
  RL_HIP(hipMemcpy(out32, ctx->d_stats.p, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return RL_OK;
}

int rl_set_paint_split(rl_ctx *ctx, int split) {
  if (!ctx) {
    set_error("rl_set_paint_split: no context");
    return RL_EINVAL;
  }
  ctx->paint_split = split;
  return RL_OK;
}

int rl_register_tile(const rl_ctx *ctx, int *S, int *waves) {
  if (!ctx || !ctx->have_chunk) {
    set_error("no chunk loaded");
    return RL_ESTATE;
  }
  if (S) *S = ctx->S;
  if (waves) *waves = ctx->waves;
  return RL_OK;
}

int rl_paint_times(const rl_ctx *ctx, float *fwd_ms, float *bwd_ms) {
  if (!ctx || !ctx->painted) {
    set_error("rl_paint_times: call rl_paint first");
    return RL_ESTATE;
  }
  if (fwd_ms) *fwd_ms = ctx->ms_fwd;
  if (bwd_ms) *bwd_ms = ctx->ms_bwd;
  return RL_OK;
}

int rl_get_stones(rl_ctx *ctx, int w, float *alpha, float *beta, float *ls_alpha, float *ls_beta,
                  int *bsnp_begin, int *bsnp_end) {
  if (!ctx || !ctx->painted) {
    set_error("rl_get_stones: call rl_paint first");
    return RL_ESTATE;
  }
  if (w < 0 || w >= ctx->W) {
    set_error("window %d out of range", w);
    return RL_EINVAL;
  }
  RL_HIP(hipSetDevice(ctx->device));
  const size_t N = ctx->N, W = ctx->W, nloc = ctx->nloc, k0 = ctx->k0;  // rows of targets k0 .. k0+nloc-1
  if (ctx->h_alpha) {  // (parked on the host: rl_park_stones)
    if (alpha) memcpy(alpha, ctx->h_alpha + w * nloc * N, nloc * N * 4);
    if (beta) memcpy(beta, ctx->h_beta + w * nloc * N, nloc * N * 4);
  } else {
    if (alpha)
      RL_HIP(hipMemcpy(alpha, ctx->d_alpha.as<float>() + w * nloc * N, nloc * N * 4, hipMemcpyDeviceToHost));
    if (beta)
      RL_HIP(hipMemcpy(beta, ctx->d_beta.as<float>() + w * nloc * N, nloc * N * 4, hipMemcpyDeviceToHost));
  }
  if (ls_alpha)
    RL_HIP(hipMemcpy(ls_alpha, ctx->d_lsa.as<float>() + w * nloc, nloc * 4, hipMemcpyDeviceToHost));
  if (ls_beta)
    RL_HIP(hipMemcpy(ls_beta, ctx->d_lsb.as<float>() + w * nloc, nloc * 4, hipMemcpyDeviceToHost));
  for (size_t t = 0; t < nloc; t++) {
    if (bsnp_begin) bsnp_begin[t] = ctx->plan.bb[(k0 + t) * W + w];
    if (bsnp_end) bsnp_end[t] = ctx->plan.be[(k0 + t) * W + w];
  }
  return RL_OK;
}

int rl_paint_record(rl_ctx *ctx, int w, int k, unsigned char *out, size_t cap, size_t *len) {
  if (!ctx || !ctx->painted || !len) {
    set_error("rl_paint_record: call rl_paint first");
    return RL_ESTATE;
  }
  const size_t N = ctx->N, W = ctx->W, nloc = ctx->nloc, k0 = ctx->k0;
  if (w < 0 || w >= ctx->W || k < (int)k0 || k >= (int)(k0 + nloc)) {
    set_error("rl_paint_record: window %d / target %d out of range", w, k);
    return RL_EINVAL;
  }
  const size_t maxrec = 8 + 2 * (28 + N * 8);
  *len = maxrec;
  if (!out || cap < maxrec) {  // (the bound, for the caller to size its buffer)
    if (!out) return RL_OK;
    set_error("rl_paint_record: buffer of %zu bytes, a record may take %zu", cap, maxrec);
    return RL_EINVAL;
  }
  RL_HIP(hipSetDevice(ctx->device));
  const size_t t = (size_t)k - k0;
  std::vector<float> a(N), b(N);
  float la, lb;
  if (ctx->h_alpha) {
    memcpy(a.data(), ctx->h_alpha + (w * nloc + t) * N, N * 4);
    memcpy(b.data(), ctx->h_beta + (w * nloc + t) * N, N * 4);
  } else {
    RL_HIP(hipMemcpy(a.data(), ctx->d_alpha.as<float>() + (w * nloc + t) * N, N * 4, hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(b.data(), ctx->d_beta.as<float>() + (w * nloc + t) * N, N * 4, hipMemcpyDeviceToHost));
  }
  RL_HIP(hipMemcpy(&la, ctx->d_lsa.as<float>() + w * nloc + t, 4, hipMemcpyDeviceToHost));
  RL_HIP(hipMemcpy(&lb, ctx->d_lsb.as<float>() + w * nloc + t, 4, hipMemcpyDeviceToHost));
  const int start = ctx->wb[w], end = ctx->wb[w + 1] - 1;  // fast_painting.cpp:591-594
  unsigned char *q = out;
  memcpy(q, &start, 4); q += 4;
  memcpy(q, &end, 4); q += 4;
  q += encode_stone(a.data(), (int)N, ctx->plan.bb[(size_t)k * W + w], la, q);
  q += encode_stone(b.data(), (int)N, ctx->plan.be[(size_t)k * W + w], lb, q);
  *len = (size_t)(q - out);
  return RL_OK;
}

static int write_paint_files(rl_ctx *ctx, const char *paint_dir, int only_window, const char *only_path);

int rl_write_paint_files(rl_ctx *ctx, const char *paint_dir) {
  if (!ctx || !ctx->painted || !paint_dir) {
    set_error("rl_write_paint_files: call rl_paint first");
    return RL_ESTATE;
  }
  return write_paint_files(ctx, paint_dir, -1, nullptr);
}

int rl_write_paint_file(rl_ctx *ctx, int w, const char *path) {
  if (!ctx || !ctx->painted || !path) {
    set_error("rl_write_paint_file: call rl_paint first");
    return RL_ESTATE;
  }
  if (w < 0 || w >= ctx->W) {
    set_error("window %d out of range", w);
    return RL_EINVAL;
  }
  return write_paint_files(ctx, nullptr, w, path);
}

static int write_paint_files(rl_ctx *ctx, const char *paint_dir, int only_window, const char *only_path) {
  if (ctx->nloc != ctx->N) {
    set_error("rl_write_paint_files: a paint file holds every target; this context paints targets %d..%d only",
              ctx->k0, ctx->k0 + ctx->nloc - 1);
    return RL_ESTATE;
  }
  const int N = ctx->N, W = ctx->W;
  const size_t maxrec = 8 + 2 * (28 + (size_t)N * 8);
  // Windows are independent files: a few writer threads each take windows round-robin (download of
  // the stones under a mutex, RLE of the N targets, one sequential write), so that encoding and the
  // file system work of different windows overlap.  Each holds ~ (2*4 + 8) * N^2 bytes of buffers.
  const size_t per_thread = (size_t)N * N * 8 + (size_t)N * maxrec;
  int nthreads = (int)std::max<size_t>(1, std::min<size_t>({(size_t)W, (size_t)4, ((size_t)6 << 30) / per_thread}));
  if (only_window >= 0) nthreads = 1;
  std::mutex gpu_mutex;
  std::atomic<int> failed{0};
  std::string first_error;
  auto worker = [&](int tid) {
    std::vector<float> a((size_t)N * N), b((size_t)N * N), la(N), lb(N);
    std::vector<int> bb(N), be(N);
    std::vector<unsigned char> recs((size_t)N * maxrec);
    std::vector<size_t> lens(N);
    for (int w = only_window >= 0 ? only_window : tid; w < (only_window >= 0 ? only_window + 1 : W) && !failed;
         w += nthreads) {
      int rc;
      {
        std::lock_guard<std::mutex> lk(gpu_mutex);
        rc = rl_get_stones(ctx, w, a.data(), b.data(), la.data(), lb.data(), bb.data(), be.data());
        if (rc && !failed.exchange(rc)) first_error = rl_last_error();
      }
      if (rc) return;
      const int start = ctx->wb[w], end = ctx->wb[w + 1] - 1;  // fast_painting.cpp:591-594
      auto encode = [&](int k) {
        unsigned char *p = recs.data() + (size_t)k * maxrec, *p0 = p;
        memcpy(p, &start, 4); p += 4;
        memcpy(p, &end, 4); p += 4;
        p += encode_stone(a.data() + (size_t)k * N, N, bb[k], la[k], p);
        p += encode_stone(b.data() + (size_t)k * N, N, be[k], lb[k], p);
        lens[k] = (size_t)(p - p0);
      };
      parallel_for(N, encode);
      const std::string fn = only_path ? std::string(only_path)
                                       : std::string(paint_dir) + "/relate_" + std::to_string(w) + ".bin";
      FILE *fp = fopen(fn.c_str(), "wb");
      if (!fp) {
        std::lock_guard<std::mutex> lk(gpu_mutex);
        if (!failed.exchange(RL_EIO)) first_error = "cannot open " + fn + " for writing";
        return;
      }
      for (int k = 0; k < N; k++) fwrite(recs.data() + (size_t)k * maxrec, 1, lens[k], fp);
      const bool bad = ferror(fp) != 0;
      if (fclose(fp) != 0 || bad) {  // (a full disc must not pass for a paint file)
        std::lock_guard<std::mutex> lk(gpu_mutex);
        if (!failed.exchange(RL_EIO)) first_error = "writing " + fn + " failed";
        return;
      }
    }
  };
  if (nthreads == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
    for (auto &t : th) t.join();
  }
  if (failed) {
    set_error("%s", first_error.c_str());
    return failed;
  }
  return RL_OK;
}

int rl_stage_paint(const char *out_dir, int chunk_index, int use_painting, double theta, double rho,
                   int sum_mode, int device) {
  // RELATE_AMD_TIMING=1: wall-clock of the stage's phases on stderr
  const bool timing = getenv("RELATE_AMD_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  auto lap = [&](const char *what) {
    const double t1 = now();
    if (timing) fprintf(stderr, "[paint stage] %-28s %8.3f s\n", what, t1 - t0);
    t0 = t1;
  };
  rl_ctx *ctx = rl_create(device);
  if (!ctx) return RL_ENODEVICE;
  lap("device context");
  int rc = rl_load_chunk(ctx, out_dir, chunk_index);
  lap("read chunk files");
  if (!rc && use_painting) rc = rl_set_painting(ctx, theta, rho);
  const std::string cdir = std::string(out_dir) + "/chunk_" + std::to_string(chunk_index);
  if (!rc) rc = mkdir_p(cdir);
  if (!rc) rc = mkdir_p(cdir + "/paint");
  if (!rc) rc = build_plan(ctx);
  lap("visited-site plan (host)");
  if (!rc) rc = rl_prepare(ctx);
  lap("uploads + lane-mask panel");
  float ms = 0.f;
  if (!rc) rc = rl_paint(ctx, sum_mode, &ms);
  lap("paint kernels");
  if (!rc) rc = rl_write_paint_files(ctx, (cdir + "/paint").c_str());
  lap("stones -> RLE -> paint files");
  rl_destroy(ctx);
  return rc;
}

int rl_stage_paint_ex(const char *out_dir, int chunk_index, const rl_stage_opts *opts) {
  rl_stage_opts o;
  if (int orc = rl_internal_resolve_opts(opts, &o)) return orc;
  return rl_stage_paint(out_dir, chunk_index, o.use_painting, o.theta, o.rho, o.sum_mode, o.device);
}

}  // extern "C"
