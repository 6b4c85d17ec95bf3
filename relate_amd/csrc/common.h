// common.h -- internal host-side declarations of librelate_amd.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "device_types.h"
#include "launch.h"
#include "relate_amd.h"

namespace rl {

void set_error(const char *fmt, ...);
// RELATE_AMD_TIMING: 0 / unset -- quiet; 1 -- where the wall-clock goes, on stderr; 2 -- also the tree builder's
// progress marks (which worker holds which tree, the merge and phase a waiting build has reached)
inline int timing_level() {
  static const int level = getenv("RELATE_AMD_TIMING") ? (atoi(getenv("RELATE_AMD_TIMING")) >= 2 ? 2 : 1) : 0;
  return level;
}

#define RL_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      rl::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RL_EHIP;                                                             \
    }                                                                             \
  } while (0)

// Streams of the stage, three priorities.  LOWEST: the tree builder's resident workers -- the runtime maps the
// priorities to hardware queues of their own, so nothing queues up behind a launch that lasts for seconds.  HIGHEST:
// the streams of the sections (a window's distance matrices; a builder's penalty, prior, weave and pair scan) -- a
// dozen kernels of a fraction of a millisecond per tree, on the section's critical path.  NORMAL: the context's
// stream (Paint; RePaint of the stage's windows, ~13 ms per launch on the CUs the workers leave free and busy 40 %
// of the time): with RePaint above them the sections' kernels waited for a whole launch to drain (2.4 ms each,
// 20-40 ms per tree), below them they take the CUs its workgroups free one by one.
// (Tried: CU masks that keep the other kernels off the builder's CUs, hipExtStreamCreateWithCUMask with 3 or 4 of
// every 8 CUs for the trees -- the 80-section stage did not finish in four times its usual time.)
inline hipError_t make_stream(hipStream_t *s, bool tree_builder, bool section = false) {
  int least = 0, greatest = 0;
  if ((tree_builder || section) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
      hipStreamCreateWithPriority(s, hipStreamNonBlocking, tree_builder ? least : greatest) == hipSuccess)
    return hipSuccess;
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

// Device and pinned-host memory of the library go through a cache: a released block is kept and handed to the next
// request of (about) its size instead of going back to the driver.  hipFree / hipHostFree wait until the device
// is idle -- with the tree builder's resident workers (minmatch_gpu.hip) that is "until no section has a tree in
// flight", i.e. a window closing in the middle of a stage would stall its thread until all the others stall too.
// Windows of a stage ask for the same sizes over and over, so the cache also saves ~25 driver calls per window.
// device_cache_trim() gives everything back (end of a stage; when an allocation fails).
void *device_cache_alloc(size_t bytes, size_t *got);  // nullptr: out of memory (after a trim)
void device_cache_release(void *p, size_t bytes);
void *pinned_cache_alloc(size_t bytes, size_t *got);
void pinned_cache_release(void *p, size_t bytes);
void device_cache_trim();
// launches of tree-builder workers that are alive (minmatch_gpu.hip): while any is, a failed allocation does not
// trim the cache (hipFree would wait for the workers) -- it takes a larger cached block, or waits for them to leave
extern std::atomic<int> g_worker_launches;
// bytes of `device`'s memory the cache holds for re-use in blocks of at least min_block bytes (cache_alloc hands out
// whole blocks only: smaller ones are no room for a request of min_block)
size_t device_cache_held(int device, size_t min_block);
// device / pinned allocations that failed on this thread (DevBuf::alloc, the window's pinned buffers): a caller that
// can wait for memory (the stage's admission loop) tells "out of memory" from any other failure by it
extern thread_local unsigned tl_alloc_failures;

// owning device buffer
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  void swap(DevBuf &o) {
    std::swap(p, o.p);
    std::swap(bytes, o.bytes);
  }
  void release() {
    if (p) device_cache_release(p, bytes);
    p = nullptr;
    bytes = 0;
  }
  int alloc(size_t n) {
    if (n <= bytes && p) return RL_OK;
    release();
    if (n == 0) n = 16;
    size_t got = 0;
    p = device_cache_alloc(n, &got);
    if (!p) {
      set_error("hipMalloc(%zu bytes) failed", n);
      tl_alloc_failures++;
      return RL_ENOMEM;
    }
    bytes = got;
    return RL_OK;
  }
  template <typename T>
  int upload(const std::vector<T> &v) {
    int rc = alloc(v.size() * sizeof(T));
    if (rc) return rc;
    if (!v.empty()) RL_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return RL_OK;
  }
  template <typename T>
  T *as() const { return static_cast<T *>(p); }
};

// Per-target visited-site plan (fast_painting.cpp:41-157), host side.
struct Plan {
  std::vector<int64_t> off;      // [N+1]
  std::vector<int32_t> sites;    // site | seq_k flag in bit 31
  std::vector<double> cf, nxt;   // per visited site
  std::vector<int32_t> ia, ie;   // [N][W] visited index of boundarySNP_begin/end
  std::vector<int32_t> bb, be;   // [N][W] boundarySNP_begin/end
  std::vector<double> binit;     // [N]
  std::vector<int32_t> order;    // targets, longest first
  bool valid = false;
};

PaintConsts make_consts(int N, double theta);
// r_prob / nor_x_theta of one interval -> (cf, nxt) (fast_painting.cpp:72-80,260)
void interval_coeffs(const PaintConsts &c, int N, double rho, double *cf, double *nxt);

}  // namespace rl

struct rl_ctx {
  int device = 0;
  hipStream_t s0 = nullptr, s1 = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  int N = 0, L = 0, W = 0;
  int k0 = 0, nloc = 0;  // targets of this context (rl_set_target_range), default all
  rl::Layout lay{};  // all N donors in 64*waves balanced runs
  int S = 0;         // register tile (doubles per lane) = words per row of the lane-mask panel
  int waves = 1;     // wavefronts per target: 2 for N > 5120 (launch.h target_waves)
  double theta = 0.001, rho = 1.0;
  int row_words = 0;
  std::vector<uint32_t> bits;  // host copy of the panel
  std::vector<double> r, rpos; // r is the unscaled map; rho applied in the plan
  std::vector<int> wb;
  rl::Plan plan;
  rl::PaintConsts consts{};
  rl::DevBuf d_bits, d_masks, d_off, d_sites, d_cf, d_nxt, d_ia, d_ie, d_binit, d_order;
  rl::DevBuf d_alpha, d_beta, d_lsa, d_lsb, d_stats;
  // the stepping stones parked in (pinned) host memory: the fused Paint + BuildTopology stage gives their HBM to the
  // windows of the sections (rl_park_stones; a window takes its slice back when it opens)
  float *h_alpha = nullptr, *h_beta = nullptr;
  rl::DevBuf d_k2_scratch;  // RePaint's checkpoint rows and side records of one launch, shared by the context's
                            // windows (window.cpp)
  // The fused Paint + BuildTopology stage owns the stones and nobody reads them after the windows: a window
  // quantises ITS slice where it lies (once: stone_quantised[w]) and re-paints from there, instead of keeping a
  // 2 N^2-float copy per open section.
  bool stones_disposable = false;
  std::vector<char> stone_quantised;  // [W], under repaint_mutex
  bool have_chunk = false, plan_on_device = false, painted = false;
  int paint_mode = -1;
  // A RePaint launch of one of the context's windows needs the forward strips (d_k2_scratch) and a stream: one at a
  // time on s0 (the windows' other work -- distance matrices -- runs on their own streams, side by side).
  std::mutex repaint_mutex;
  // A stage may open a SECOND lane (strips, stream, events of its own; RELATE_AMD_REPAINT_LANES=2, treeseq.cpp): a
  // launch is a forward and a backward kernel of one workgroup per target, each as long as its longest target, on
  // the CUs the tree builder's workers leave -- two windows' launches side by side fill each other's tails.
  struct RepaintLane {
    std::mutex m;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    rl::DevBuf scratch;
  } lane2;
  bool two_lanes = false;
  std::atomic<long long> repaint_launches{0};  // RePaint launches of the context's windows
  std::atomic<long long> repaint_us{0};        // ... and their time on the device (HIP events), microseconds
  float ms_fwd = 0.f, ms_bwd = 0.f, ms_paint = 0.f;
  int paint_split = 0;  // rl_set_paint_split: one launch per direction instead of one for both
};

namespace rl {
int build_plan(rl_ctx *ctx);
int upload_plan(rl_ctx *ctx);
int host_threads();
int local_world_size();  // ranks of the job on this host (LOCAL_WORLD_SIZE), 1 outside a launcher
int local_rank();
// paint-file codec (collapsed_matrix.hpp:228-296)
size_t encode_stone(const float *v, int N, int bsnp, float logscale, unsigned char *out);
size_t decode_stone(const unsigned char *in, size_t avail, int N, float *v, int *bsnp, float *logscale);
float fast_log_host(float v);
}  // namespace rl


// the caller's rl_stage_opts, whatever its age, onto the defaults (treeseq.cpp); RL_EINVAL for a struct that never
// went through rl_stage_opts_init
extern "C" int rl_internal_resolve_opts(const rl_stage_opts *in, rl_stage_opts *out);

// FindEquivalentBranches fused behind BuildTopology (equivalent.cpp): the sections' trees go from the stage to the
// association in memory, every .anc file is written once
namespace rl {
struct FebJob;
struct HostTree;
FebJob *feb_job_create(int N, int W, int threads);
int feb_job_add_section(FebJob *job, int w, const std::vector<HostTree> &trees);
int feb_job_finish(FebJob *job, const std::string &dir_base);  // writes <dir_base>_<w>.anc for every section
void feb_job_destroy(FebJob *job);
}  // namespace rl
