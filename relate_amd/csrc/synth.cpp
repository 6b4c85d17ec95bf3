// synth.cpp -- synthetic haplotype panels for tests and bench.py.
//
// Block-wise neutral coalescent (SURVEY.md 8d "Synthetic inputs"): every
// `block` SNPs a fresh Kingman tree on N leaves is drawn and each SNP of the
// block is placed on a branch with probability proportional to its length;
// the derived allele is carried by the branch's descendants.  This gives the
// 1/i site-frequency spectrum (mean derived frequency ~ 1/H_{N-1}, 0.11 at
// N=5000) that the painting's cost model depends on, because a target only
// visits sites where it is derived (fast_painting.cpp:93-96).
//
// Not part of the reference; it stands in for `Relate --mode MakeChunks`
// (data.cpp:117-518) as the producer of chunk files, whose formats it writes
// exactly.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "relate_amd.h"

namespace {

struct Rng {  // xoshiro256** seeded by splitmix64; platform independent
  uint64_t s[4];
  static uint64_t splitmix(uint64_t &x) {
    uint64_t z = (x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  explicit Rng(uint64_t seed) {
    for (auto &v : s) v = splitmix(seed);
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  double unif() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

}  // namespace

extern "C" {

int rl_synth_panel(int N, int L, uint64_t seed, int block, int jitter,
                   uint8_t *seq_chars, uint32_t *bits, int row_words, int *bp,
                   double *r, double *rpos) {
  if (N < 2 || L < 2 || block < 1) return -1;
  const int words = (N + 31) / 32;
  if (bits && row_words < words) return -1;
  const int nodes = 2 * N - 1;
  const int nblocks = (L + block - 1) / block;
  int T = (int)std::thread::hardware_concurrency();
  if (const char *e = getenv("RELATE_AMD_THREADS")) T = atoi(e);
  T = std::max(1, std::min(T, std::min(nblocks, 64)));
  auto worker = [&](int tid) {
  std::vector<uint32_t> desc((size_t)nodes * words);
  std::vector<double> height(nodes), cum(nodes);
  std::vector<int> active(N);
  for (int blk = tid; blk < nblocks; blk += T) {
    const int s0 = blk * block;
    Rng rng(seed * 0x9e3779b97f4a7c15ull + (uint64_t)blk + 1);  // one stream per block
    // Kingman tree
    std::fill(desc.begin(), desc.end(), 0u);
    for (int i = 0; i < N; i++) {
      desc[(size_t)i * words + (i >> 5)] = 1u << (i & 31);
      height[i] = 0.0;
      active[i] = i;
    }
    double t = 0.0, total = 0.0;
    int nact = N;
    for (int nn = N; nn < nodes; nn++) {
      double rate = 0.5 * nact * (nact - 1.0);
      t += -std::log(1.0 - rng.unif()) / rate;
      int a = (int)rng.below((uint32_t)nact);
      int b = (int)rng.below((uint32_t)(nact - 1));
      if (b >= a) b++;
      int ca = active[a], cb = active[b];
      height[nn] = t;
      cum[ca] = t - height[ca];  // branch length above child
      cum[cb] = t - height[cb];
      uint32_t *dn = &desc[(size_t)nn * words];
      const uint32_t *da = &desc[(size_t)ca * words], *db = &desc[(size_t)cb * words];
      for (int w = 0; w < words; w++) dn[w] = da[w] | db[w];
      // remove a and b, add nn
      if (a < b) std::swap(a, b);
      active[a] = active[nact - 1];
      active[b] = active[nact - 2];
      active[nact - 2] = nn;
      nact--;
    }
    for (int i = 0; i < nodes - 1; i++) {
      total += cum[i];
      cum[i] = total;
    }
    const int s1 = std::min(L, s0 + block);
    for (int s = s0; s < s1; s++) {
      double u = rng.unif() * total;
      int node = (int)(std::upper_bound(cum.begin(), cum.begin() + (nodes - 1), u) - cum.begin());
      if (node > nodes - 2) node = nodes - 2;
      const uint32_t *dn = &desc[(size_t)node * words];
      if (bits) {
        uint32_t *row = bits + (size_t)s * row_words;
        for (int w = 0; w < words; w++) row[w] = dn[w];
        for (int w = words; w < row_words; w++) row[w] = 0;
      }
      if (seq_chars) {
        uint8_t *row = seq_chars + (size_t)s * N;
        for (int n = 0; n < N; n++) row[n] = (uint8_t)('0' + ((dn[n >> 5] >> (n & 31)) & 1u));
      }
    }
  }
  };
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
    for (auto &x : th) x.join();
  }
  // positions: ~100 bp spacing, uniform 1 cM/Mb map (data.cpp:443-481)
  Rng rng(seed ^ 0x5851f42d4c957f2dull);
  int pos = 1000;
  std::vector<int> bpv((size_t)L + 1);
  for (int s = 0; s <= L; s++) {
    bpv[s] = pos;
    pos += jitter ? 1 + (int)rng.below(199) : 100;
  }
  for (int s = 0; s <= L; s++) rpos[s] = bpv[s] * 1e-6 * 1e-2;
  for (int s = 0; s < L; s++) {
    if (bp) bp[s] = bpv[s];
    double v = rpos[s + 1] - rpos[s];
    if (v < 1e-10) v = 1e-10;
    r[s] = v * 2500;
  }
  return 0;
}

// The reference's window rule (data.cpp:213-229): a window is closed at the
// SNP where the running sum of #carriers*(N+1) reaches `budget` floats and
// the window already holds more than 10 SNPs.  Returns W; wb gets W+1 entries.
int rl_synth_windows(int N, int L, const uint8_t *seq_chars, double budget,
                     int *wb, int max_windows) {
  int W = 1, in_window = 0;
  double mem = 0.0;
  wb[0] = 0;
  for (int s = 0; s < L; s++) {
    int nd = 0;
    const uint8_t *row = seq_chars + (size_t)s * N;
    for (int n = 0; n < N; n++) nd += (row[n] == '1');
    mem += (double)nd * (N + 1);
    if (mem >= budget && in_window > 10) {
      if (W >= max_windows) return -1;
      in_window = 0;
      mem = 0.0;
      wb[W++] = s;
    }
    in_window++;
  }
  wb[W] = L;
  return W;
}

int rl_synth_windows_bits(int N, int L, const uint32_t *bits, int row_words, double budget, int *wb,
                          int max_windows) {
  int W = 1, in_window = 0;
  double mem = 0.0;
  wb[0] = 0;
  const int words = (N + 31) / 32;
  for (int s = 0; s < L; s++) {
    int nd = 0;
    const uint32_t *row = bits + (size_t)s * row_words;
    for (int w = 0; w < words; w++) nd += __builtin_popcount(row[w]);
    mem += (double)nd * (N + 1);
    if (mem >= budget && in_window > 10) {
      if (W >= max_windows) return -1;
      in_window = 0;
      mem = 0.0;
      wb[W++] = s;
    }
    in_window++;
  }
  wb[W] = L;
  return W;
}

static int write_vec(const std::string &fn, const void *hdr, size_t hdr_len,
                     const void *data, size_t len) {
  FILE *fp = fopen(fn.c_str(), "wb");
  if (!fp) return -1;
  fwrite(hdr, 1, hdr_len, fp);
  if (len) fwrite(data, 1, len, fp);
  fclose(fp);
  return 0;
}

// chunk files in the formats of data.cpp:261-298, 485-516 / data.hpp:78-99
int rl_write_chunk_files(const char *dir, int chunk, int N, int L,
                         const uint8_t *seq_chars, const int *bp,
                         const double *r, const double *rpos, const int *wb,
                         int W) {
  std::string b = std::string(dir) + "/chunk_" + std::to_string(chunk);
  uint64_t h2[2] = {(uint64_t)L, (uint64_t)N};
  if (write_vec(b + ".hap", h2, 16, seq_chars, (size_t)L * N)) return -1;
  uint32_t uL = (uint32_t)L, uL1 = (uint32_t)L + 1;
  std::vector<int> dist(L), state(L, 1);
  for (int s = 0; s < L; s++) dist[s] = (s + 1 < L) ? bp[s + 1] - bp[s] : 1;
  if (write_vec(b + ".bp", &uL, 4, bp, (size_t)L * 4)) return -1;
  if (write_vec(b + ".dist", &uL, 4, dist.data(), (size_t)L * 4)) return -1;
  if (write_vec(b + ".r", &uL, 4, r, (size_t)L * 8)) return -1;
  if (write_vec(b + ".rpos", &uL1, 4, rpos, (size_t)(L + 1) * 8)) return -1;
  int iL = L;
  if (write_vec(b + ".state", &iL, 4, state.data(), (size_t)L * 4)) return -1;
  std::vector<int> p = {N, L, W + 1};
  p.insert(p.end(), wb, wb + W + 1);
  std::string pf = std::string(dir) + "/parameters_c" + std::to_string(chunk) + ".bin";
  if (write_vec(pf, p.data(), p.size() * 4, nullptr, 0)) return -1;
  return 0;
}

}  // extern "C"
