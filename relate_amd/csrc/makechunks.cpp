// makechunks.cpp -- `Relate --mode MakeChunks`: .haps/.sample/map -> chunk files.
//
// Host restatement of Data::MakeChunks (src/data.cpp:117-518) with its readers
// haps (src/data.hpp:108-169, data.cpp:543-573) and map (data.cpp:593-626), and
// of the stage driver pipeline/MakeChunks.cpp.  SURVEY.md 8f-3 ("next"): the
// producer of this path's inputs, so that Paint / BuildTopology no longer need
// the reference binary upstream.  Every file it writes (parameters.bin,
// parameters_c<i>.bin, chunk_<i>.{hap,state,bp,dist,rpos,r}, props.bin) is
// byte-identical to the reference's (tests/test_makechunks.py).  The haps text
// is read once, in blocks parsed on all host threads, into a bit-packed panel
// that stays in memory (read_haps below); the chunk files are written from it.
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace {

// gzip::open / close (src/data.cpp:6-77): gz files are read through `gunzip -c`
struct InFile {
  FILE *fp = nullptr;
  bool piped = false;
  bool open(const char *fn) {
    FILE *chk = fopen(fn, "rb");
    if (!chk) return false;
    unsigned char b[3] = {0, 0, 0};
    size_t got = fread(b, 1, 3, chk);
    fclose(chk);
    piped = got == 3 && b[0] == 0x1f && b[1] == 0x8b && b[2] == 0x08;
    if (piped) {
      std::string cmd = "gunzip -c '";  // (a quote in the path closes, escapes and reopens the quoting)
      for (const char *c = fn; *c; c++) cmd += *c == '\'' ? std::string("'\\''") : std::string(1, *c);
      cmd += "'";
      fp = popen(cmd.c_str(), "r");
    } else {
      fp = fopen(fn, "r");
    }
    return fp != nullptr;
  }
  void close() {
    if (!fp) return;
    if (piped)
      pclose(fp);
    else
      fclose(fp);
    fp = nullptr;
  }
};

int count_newlines(const char *fn, long *lines) {
  InFile f;
  if (!f.open(fn)) return -1;
  long n = 0;
  int c;
  while ((c = fgetc(f.fp)) != EOF)
    if (c == '\n') n++;
  f.close();
  *lines = n;
  return 0;
}

template <typename T>
void put_vec(const std::string &fn, unsigned int n, const T *data) {
  FILE *fp = fopen(fn.c_str(), "wb");
  if (!fp) return;
  fwrite(&n, sizeof(unsigned int), 1, fp);
  fwrite(data, sizeof(T), n, fp);
  fclose(fp);
}

// ---- the .haps text, read ONCE: blocks of whole lines parsed on all host threads into the bit-packed panel (an
// allele is one bit here against two characters in the file, so the whole panel stays in memory: N = 2000 x 5M SNPs
// are 20 GB of text and 1.25 GB of bits) plus what the chunk files repeat of every line.  The reference reads the
// file twice with fscanf, once for the sizes and once for the rows (haps::ReadSNP, data.cpp:543-573); a line is
// taken apart as its "%s %s %d %s %s" + fgets would: five fields, then the first N characters that are '0' or '1'.
struct HapsPanel {
  uint32_t row_words = 0;
  std::vector<uint32_t> bits;  // [L][row_words]: bit n of row s = haplotype n carries the second allele at SNP s
  std::vector<int> bp, derived;
  std::vector<std::string> rsid, ancestral, alternative;
};

struct LineFault {  // the first line of a block that does not parse
  long line = -1;
  int alleles = -1;  // -1: the five fields; else how many alleles the line holds
  std::string chr, rs;
  int bp = 0;
};

inline bool blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }

// one line [p, e) without its newline -> row `snp` of the panel; false: malformed (fault says how)
bool parse_haps_line(const char *p, const char *e, int N, HapsPanel &out, size_t snp, LineFault &fault) {
  auto field = [&](const char *&b, const char *&t) {
    while (p < e && blank(*p)) p++;
    b = p;
    while (p < e && !blank(*p)) p++;
    t = p;
    return t > b;
  };
  const char *chr_b, *chr_e, *rs_b, *rs_e, *anc_b, *anc_e, *alt_b, *alt_e;
  if (!field(chr_b, chr_e) || !field(rs_b, rs_e)) return false;
  while (p < e && blank(*p)) p++;
  bool negative = false;
  if (p < e && (*p == '-' || *p == '+')) negative = *p++ == '-';
  if (p >= e || *p < '0' || *p > '9') return false;
  long pos = 0;
  while (p < e && *p >= '0' && *p <= '9') pos = pos * 10 + (*p++ - '0');  // (%d: the digits, wherever they end)
  if (!field(anc_b, anc_e) || !field(alt_b, alt_e)) return false;
  uint32_t *row = &out.bits[snp * out.row_words];
  for (uint32_t w = 0; w < out.row_words; w++) row[w] = 0u;
  int filled = 0, ones = 0;
  for (; p < e && filled < N; p++) {
    const unsigned d = (unsigned)(*p - '0');
    if (d <= 1u) {
      row[filled >> 5] |= d << (filled & 31);
      ones += (int)d;
      filled++;
    }
  }
  out.bp[snp] = (int)(negative ? -pos : pos);
  if (filled != N) {
    fault.alleles = filled;
    fault.chr.assign(chr_b, chr_e);
    fault.rs.assign(rs_b, rs_e);
    fault.bp = out.bp[snp];
    return false;
  }
  out.derived[snp] = ones;
  out.rsid[snp].assign(rs_b, std::min<size_t>(rs_e - rs_b, 1023));  // (%1023s)
  out.ancestral[snp].assign(anc_b, std::min<size_t>(anc_e - anc_b, 1023));
  out.alternative[snp].assign(alt_b, std::min<size_t>(alt_e - alt_b, 1023));
  return true;
}

// 0, or an RL_E* code with the error text set.  Lines = newline characters, as the reference counts them: a last
// line without one is not a SNP.
int read_haps(const char *fn, int N, HapsPanel &out) {
  using rl::set_error;
  InFile f;
  if (!f.open(fn)) {
    set_error("Failed to open file %s", fn);
    return RL_EIO;
  }
  out.row_words = (uint32_t)((N + 31) / 32);
  const size_t block = (size_t)64 << 20;
  std::vector<char> buf(block + 1);
  std::vector<size_t> starts;
  size_t held = 0, L = 0;
  const int T = std::max(1, rl::host_threads());
  int rc = RL_OK;
  for (;;) {
    if (held == buf.size()) buf.resize(buf.size() * 2);  // (a single line longer than the block)
    const size_t got = fread(buf.data() + held, 1, buf.size() - held, f.fp);
    const size_t have = held + got;
    // whole lines of the block
    starts.clear();
    size_t at = 0;
    while (at < have) {
      const char *nl = (const char *)memchr(buf.data() + at, '\n', have - at);
      if (!nl) break;
      starts.push_back(at);
      at = (size_t)(nl - buf.data()) + 1;
    }
    starts.push_back(at);  // (end of the last whole line)
    const size_t n = starts.size() - 1;
    if (n) {
      out.bits.resize((L + n) * out.row_words);
      out.bp.resize(L + n);
      out.derived.resize(L + n);
      out.rsid.resize(L + n);
      out.ancestral.resize(L + n);
      out.alternative.resize(L + n);
      std::vector<LineFault> faults((size_t)T);
      auto work = [&](int t) {
        for (size_t i = n * t / T; i < n * (t + 1) / T; i++)
          if (!parse_haps_line(buf.data() + starts[i], buf.data() + starts[i + 1] - 1, N, out, L + i, faults[t])) {
            faults[t].line = (long)(L + i);
            return;
          }
      };
      std::vector<std::thread> th;
      for (int t = 1; t < T; t++) th.emplace_back(work, t);
      work(0);
      for (auto &x : th) x.join();
      for (const LineFault &ft : faults)  // (threads hold ascending ranges: the first fault is the lowest line)
        if (ft.line >= 0) {
          if (ft.alleles < 0)
            set_error("%s: malformed line %ld", fn, ft.line + 1);
          else
            set_error("%s: SNP %s %s %d has %d alleles, %d expected", fn, ft.chr.c_str(), ft.rs.c_str(), ft.bp, ft.alleles, N);
          rc = RL_EFORMAT;
          break;
        }
      if (rc) break;
      L += n;
    }
    held = have - at;
    memmove(buf.data(), buf.data() + at, held);
    if (got == 0) break;
  }
  f.close();
  return rc;
}

// a panel row as the characters of chunk_<i>.hap: eight alleles per table entry
void expand_row(const uint32_t *words, int N, char *chars) {
  static const struct Table {
    uint64_t v[256];
    Table() {
      for (int b = 0; b < 256; b++) {
        uint64_t x = 0;
        for (int k = 0; k < 8; k++) x |= (uint64_t)('0' + ((b >> k) & 1)) << (8 * k);
        v[b] = x;
      }
    }
  } table;
  int n = 0;
  for (; n + 8 <= N; n += 8) memcpy(chars + n, &table.v[(words[n >> 5] >> (n & 31)) & 0xffu], 8);
  for (; n < N; n++) chars[n] = (char)('0' + ((words[n >> 5] >> (n & 31)) & 1u));
}

}  // namespace

extern "C" int rl_make_chunks(const char *haps_fn, const char *sample_fn, const char *map_fn, const char *dist_fn,
                              const char *out_dir, int use_transitions, float memory_gb) {
  using rl::set_error;
  if (!haps_fn || !sample_fn || !map_fn || !out_dir) {
    set_error("rl_make_chunks: bad arguments");
    return RL_EINVAL;
  }
  const std::string file_out(out_dir);

  // ---- N from the sample file, L from the haps file (data.hpp:125-156)
  int N = 0;
  {
    InFile f;
    if (!f.open(sample_fn)) {
      set_error("Failed to open file %s", sample_fn);
      return RL_EIO;
    }
    char id1[1024], id2[1024], dummy[1024];
    if (fscanf(f.fp, "%1023s %1023s %1023s", id1, id2, dummy) != 3 ||
        fscanf(f.fp, "%1023s %1023s %1023s", id1, id2, dummy) != 3) {
      f.close();
      set_error("%s: two header lines expected", sample_fn);
      return RL_EFORMAT;
    }
    while (fscanf(f.fp, "%1023s %1023s %1023s", id1, id2, dummy) == 3) N += (strcmp(id1, id2) == 0) ? 2 : 1;
    f.close();
  }
  // RELATE_AMD_TIMING=1: wall-clock of the stage's phases on stderr
  const bool timing = getenv("RELATE_AMD_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double lap_t0 = now();
  auto lap = [&](const char *what) {
    const double t1 = now();
    if (timing) fprintf(stderr, "[makechunks] %-44s %8.3f s\n", what, t1 - lap_t0);
    lap_t0 = t1;
  };
  HapsPanel panel;
  if (int rc = read_haps(haps_fn, N, panel)) return rc;
  lap("haps text -> bit-packed panel");
  const int L = (int)panel.bp.size();
  if (N < 2 || L < 2) {
    set_error("MakeChunks: need at least 2 haplotypes and 2 SNPs (N=%d, L=%d)", N, L);
    return RL_EFORMAT;
  }
  std::vector<int> &bp_pos = panel.bp;
  bp_pos.resize((size_t)L + 1);
  const std::vector<std::string> &ancestral = panel.ancestral, &alternative = panel.alternative, &rsid = panel.rsid;

  // The budget of a window (data.cpp:129): `memory_gb` of floats minus the two N x N matrices and three vectors
  // the tree builder holds next to a window's posteriors.
  const double window_budget = (memory_gb) * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N);
  if (window_budget <= 0) {
    set_error("Error: Need larger memory allowance.");
    return RL_EINVAL;
  }
  const int max_windows = 500;  // per chunk, the carried ones included (data.cpp:131)
  const int overlap = 20000;    // SNPs a chunk repeats from its predecessor (data.cpp:133)
  int max_chunk_snps = std::min(L + 1, (int)(window_budget / N));
  if (memory_gb >= 100) max_chunk_snps = 2500000;

  // ---- what a SNP's row of posteriors will cost a window: (number of derived alleles) * (N + 1) floats
  // (data.cpp:216).  cost[s] is the running total before SNP s -- integers far below 2^53, so the doubles are exact
  // and a window boundary is a binary search instead of the reference's accumulate-and-compare over the SNPs.
  std::vector<double> cost((size_t)L + 1, 0.0);
  for (int s = 0; s < L; s++) cost[(size_t)s + 1] = cost[s] + (double)panel.derived[s] * (N + 1);
  bp_pos[L] = bp_pos[L - 1] + 1;

  // ---- the plan: chunks and their windows from the running cost alone
  struct ChunkPlan {
    int begin = 0, end = 0;      // the chunk's own SNPs [begin, end)
    int first_snp = 0;           // where its files start: begin, or begin - overlap behind the first chunk
    std::vector<int> cuts;       // window starts among its own SNPs, cuts[0] == begin
    std::vector<int> carried;    // window starts inside the repeated stretch, carried[0] == first_snp
  };
  std::vector<ChunkPlan> plan;
  double largest_window = 0.0;   // floats: what the biggest window of the data set costs (parameters.bin)
  int most_windows = 0;
  for (int begin = 0; begin < L;) {
    ChunkPlan c;
    c.begin = c.first_snp = begin;
    if (!plan.empty()) {
      const ChunkPlan &prev = plan.back();
      // the reference asserts here (data.cpp:170): a chunk must be longer than what it hands on
      if (begin - prev.first_snp < overlap || overlap > prev.end - prev.begin) {
        set_error("MakeChunks: chunk shorter than the %d-SNP overlap (raise --memory)", overlap);
        return RL_EINVAL;
      }
      c.first_snp = begin - overlap;
      c.carried.push_back(c.first_snp);
      for (int cut : prev.cuts)
        if (cut > c.first_snp) c.carried.push_back(cut);
      if ((int)c.carried.size() >= max_windows - 1) {
        set_error("MakeChunks: too many windows in the overlap (raise --memory)");
        return RL_EINVAL;
      }
    }
    const int limit = std::min(L, begin + max_chunk_snps);
    c.cuts.push_back(begin);
    c.end = limit;
    int open = begin;          // start of the window being filled
    double open_cost = cost[begin];  // running cost where the window's own account starts
    while (true) {
      // A window closes at the first SNP t, more than 10 SNPs in, whose row brings its account to the budget
      // (data.cpp:219-229); t opens the next window, whose account starts BEHIND t (t's own row is charged to the
      // window it closed).
      // (the window's account is cost - open_cost, exact integers; open_cost + budget would round a fractional
      //  budget -- --memory 0.3 -- and could close a window one SNP early)
      const size_t idx = std::partition_point(cost.begin(), cost.end(),
                                              [&](double c) { return c - open_cost < window_budget; }) - cost.begin();
      const long t = std::max<long>((long)idx - 1, (long)open + 11);
      if (idx > (size_t)L || t >= limit) break;
      largest_window = std::max(largest_window, cost[t + 1] - open_cost);
      c.cuts.push_back((int)t);
      open = (int)t;
      open_cost = cost[t + 1];
      if ((int)c.cuts.size() + (int)c.carried.size() == max_windows) {  // the chunk is full of windows:
        c.end = (int)t + 1;                                             // it ends behind the SNP that opened the last
        break;
      }
    }
    largest_window = std::max(largest_window, cost[c.end] - open_cost);  // the window left open at the chunk's end
    most_windows = std::max(most_windows, (int)c.cuts.size());
    const float snps_per_window = (c.end - c.begin) / (int)c.cuts.size();
    if (snps_per_window < 100)
      std::cerr << "Windows hold " << snps_per_window << " SNPs on average: raise --memory (default 5 GB) about "
                << 100 / snps_per_window << "-fold." << std::endl;
    begin = c.end;
    plan.push_back(std::move(c));
  }
  const int num_chunks = (int)plan.size();
  std::vector<int> section_boundary_start(num_chunks), section_boundary_end(num_chunks);
  for (int c = 0; c < num_chunks; c++) {
    section_boundary_start[c] = plan[c].first_snp;
    section_boundary_end[c] = plan[c].end;
  }

  // ---- parameters_c<i>.bin (data.cpp:254-298) and chunk_<i>.state (:307-345): from the plan and the alleles
  auto snp_state = [&](int snp) {  // 0: a transition left out with --transversion
    if (use_transitions) return 1;
    const std::string &a = ancestral[snp], &b = alternative[snp];
    const bool ts = (a == "C" && b == "T") || (a == "T" && b == "C") || (a == "G" && b == "A") || (a == "A" && b == "G");
    return ts ? 0 : 1;
  };
  for (int ci = 0; ci < num_chunks; ci++) {
    const ChunkPlan &c = plan[ci];
    const int Lc = c.end - c.first_snp;
    std::vector<int> rec = {N, Lc, (int)(c.carried.size() + c.cuts.size()) + 1};
    for (int cut : c.carried) rec.push_back(cut - c.first_snp);
    for (int cut : c.cuts) rec.push_back(cut - c.first_snp);
    rec.push_back(c.end - c.first_snp);
    std::vector<int> state((size_t)Lc + 1);
    state[0] = Lc;
    for (int t = 0; t < Lc; t++) state[(size_t)t + 1] = snp_state(c.first_snp + t);
    FILE *fp = fopen((file_out + "/parameters_c" + std::to_string(ci) + ".bin").c_str(), "w");
    FILE *fs = fopen((file_out + "/chunk_" + std::to_string(ci) + ".state").c_str(), "wb");
    if (!fp || !fs) {
      if (fp) fclose(fp);
      if (fs) fclose(fs);
      set_error("cannot write chunk files under %s", out_dir);
      return RL_EIO;
    }
    fwrite(rec.data(), 4, rec.size(), fp);
    fwrite(state.data(), 4, state.size(), fs);
    fclose(fp);
    fclose(fs);
  }

  // ---- chunk_<i>.hap (u64 L, u64 N, L rows of N chars; collapsed_matrix.hpp:204-225), the rows of the chunk -- its
  // own and the `overlap` it repeats from its predecessor -- expanded from the panel in slabs.
  // Next to it the same rows BIT-PACKED as they lie in memory, chunk_<i>.bits -- what the device path works on
  // (rl_set_chunk_bits: bit n of row s = haplotype n derived at SNP s): rl_load_chunk prefers it and never touches
  // the 8 x larger char file (2.5 GB at N = 5000 x L = 500k).  Header: "RLB2", N, L, row_words (u32 each), then the
  // size and the modification time (ns) of the chunk_<i>.hap it was written next to (u64 each: a .hap regenerated
  // later -- by the reference's MakeChunks, say -- makes the file stale and rl_load_chunk falls back to the .hap); then
  // L rows of row_words u32.  OPT-IN (RELATE_AMD_CHUNK_BITS=1): the file is foreign to the reference, whose later
  // stages delete the files they know and then rmdir the directory (Finalize.cpp:290, Clean.cpp:120; a leftover makes
  // that fail) -- this library's FindEquivalentBranches removes it, the reference's does not.  Without the option a
  // chunk_<i>.bits left by an earlier run is removed.
  lap("plan, parameters_c*.bin, chunk_*.state");
  // The chunks are written by a few threads of their own, side by side and next to props.bin below (3 KB per SNP):
  // the stage is bound by the file system from here on.
  const bool with_bits = getenv("RELATE_AMD_CHUNK_BITS") && atoi(getenv("RELATE_AMD_CHUNK_BITS")) != 0;
  auto write_chunk = [&](int ci) -> bool {
    const uint32_t rw = panel.row_words;
    const int first = plan[ci].first_snp, end = plan[ci].end;
    const std::string hap_name = file_out + "/chunk_" + std::to_string(ci) + ".hap";
    const std::string bits_name = file_out + "/chunk_" + std::to_string(ci) + ".bits";
    FILE *fh = fopen(hap_name.c_str(), "wb");
    if (!with_bits) (void)remove(bits_name.c_str());  // (a stale one would be read instead of this .hap)
    if (!fh) return false;
    const uint64_t dims[2] = {(uint64_t)(end - first), (uint64_t)N};
    fwrite(dims, 8, 2, fh);
    const int slab_rows = std::max(1, (int)(((size_t)8 << 20) / (size_t)N));
    std::vector<char> slab((size_t)slab_rows * N);
    for (int s0 = first; s0 < end; s0 += slab_rows) {
      const int n = std::min(slab_rows, end - s0);
      for (int i = 0; i < n; i++) expand_row(&panel.bits[(size_t)(s0 + i) * rw], N, &slab[(size_t)i * N]);
      fwrite(slab.data(), 1, (size_t)n * N, fh);
    }
    bool good = ferror(fh) == 0;
    good &= fclose(fh) == 0;
    if (good && with_bits) {
      struct stat st;
      FILE *fb = stat(hap_name.c_str(), &st) == 0 ? fopen(bits_name.c_str(), "wb") : nullptr;
      if (!fb) return false;
      const uint32_t head[4] = {0x32424c52u /* "RLB2" */, (uint32_t)N, (uint32_t)dims[0], rw};
      const uint64_t of_hap[2] = {(uint64_t)st.st_size,
                                  (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec};
      fwrite(head, 4, 4, fb);
      fwrite(of_hap, 8, 2, fb);
      fwrite(&panel.bits[(size_t)first * rw], 4, (size_t)(end - first) * rw, fb);
      good &= ferror(fb) == 0;
      good &= fclose(fb) == 0;
    }
    return good;
  };
  std::atomic<int> next_chunk(0), chunks_failed(0);
  std::vector<std::thread> chunk_writers;
  struct JoinAll {  // (whichever way the function is left)
    std::vector<std::thread> &v;
    ~JoinAll() {
      for (auto &t : v)
        if (t.joinable()) t.join();
    }
  } join_chunk_writers{chunk_writers};
  for (int t = 0; t < std::min({num_chunks, 8, std::max(1, rl::host_threads() / 2)}); t++)
    chunk_writers.emplace_back([&]() {
      for (int ci; (ci = next_chunk.fetch_add(1)) < num_chunks;)
        if (!write_chunk(ci)) chunks_failed++;
    });

  std::cerr << std::setprecision(2) << "Paint files will take at least "
            << 2.0 * (4.0 * N * N * (most_windows + 2.0)) / 1e9 << " GB of disc." << std::endl;

  {  // parameters.bin (data.cpp:361-375)
    FILE *fp = fopen((file_out + "/parameters.bin").c_str(), "w");
    if (!fp) {
      set_error("cannot write parameters.bin");
      return RL_EIO;
    }
    const double needed_gb = (largest_window + (2 * N * N + 3 * N)) * (4.0 / 1e9);
    fwrite(&N, 4, 1, fp);
    fwrite(&L, 4, 1, fp);
    fwrite(&num_chunks, 4, 1, fp);
    fwrite(&needed_gb, 8, 1, fp);
    fwrite(section_boundary_start.data(), 4, num_chunks, fp);
    fwrite(section_boundary_end.data(), 4, num_chunks, fp);
    fclose(fp);
  }

  // ---- dist (data.cpp:377-418)
  std::vector<int> dist(L);
  if (!dist_fn || std::string(dist_fn) == "unspecified") {
    for (int s = 0; s + 1 < L; s++) {
      dist[s] = bp_pos[s + 1] - bp_pos[s];
      if (dist[s] <= 0) {
        set_error("Failed at BP %d: SNPs are not sorted by bp or more than one SNP at same position.", bp_pos[s]);
        return RL_EFORMAT;
      }
    }
    dist[L - 1] = 1;
  } else {
    InFile f;
    if (!f.open(dist_fn)) {
      set_error("Failed to open file %s", dist_fn);
      return RL_EIO;
    }
    char buffer[40];
    if (fscanf(f.fp, "%39s %39s", buffer, buffer) != 2) buffer[0] = 0;
    int mbp, mdist, s = 0;
    while (fscanf(f.fp, "%d %d", &mbp, &mdist) == 2) {
      if (s >= L || bp_pos[s] != mbp) {
        f.close();
        set_error("%s disagrees with the haps file at line %d", dist_fn, s + 2);
        return RL_EFORMAT;
      }
      dist[s++] = mdist;
    }
    f.close();
  }

  {  // props.bin (data.cpp:420-439)
    FILE *fp = fopen((file_out + "/props.bin").c_str(), "wb");
    if (!fp) {
      set_error("cannot write props.bin");
      return RL_EIO;
    }
    // (3084 bytes per SNP, 15 GB at 5M SNPs: records are laid out in a slab and written with one call each)
    const size_t rec = 12 + 3 * 1024;
    const int per_slab = 4096;
    std::vector<char> slab(rec * per_slab);
    for (int s0 = 0; s0 < L; s0 += per_slab) {
      const int n = std::min(per_slab, L - s0);
      memset(slab.data(), 0, rec * n);
      for (int i = 0; i < n; i++) {
        char *q = &slab[rec * i];
        const int s = s0 + i;
        memcpy(q, &s, 4);
        memcpy(q + 4, &bp_pos[s], 4);
        memcpy(q + 8, &dist[s], 4);
        auto put = [&](char *dst, const std::string &v) { memcpy(dst, v.data(), std::min<size_t>(v.size(), 1023)); };
        put(q + 12, rsid[s]);
        put(q + 12 + 1024, ancestral[s]);
        put(q + 12 + 2048, alternative[s]);
      }
      fwrite(slab.data(), 1, rec * n, fp);
    }
    fclose(fp);
  }
  lap("dist, props.bin");
  for (auto &t : chunk_writers) t.join();
  lap("waiting for the chunk files' writers");
  if (chunks_failed.load()) {
    set_error("writing the chunk files under %s failed", out_dir);
    return RL_EIO;
  }

  // ---- genetic map (data.cpp:593-626) -> rpos, r (data.cpp:441-481)
  std::vector<int> mbp;
  std::vector<double> gen_pos;
  {
    long lines = 0;
    if (count_newlines(map_fn, &lines)) {
      set_error("Failed to open file %s", map_fn);
      return RL_EIO;
    }
    lines--;  // header
    InFile f;
    if (lines < 2 || !f.open(map_fn)) {
      set_error("%s: at least two map positions expected", map_fn);
      return RL_EFORMAT;
    }
    char buffer[1024];
    for (int h = 0; h < 3; h++)
      if (fscanf(f.fp, "%1023s", buffer) != 1) buffer[0] = 0;
    mbp.resize(lines);
    gen_pos.resize(lines);
    float dummy;
    double fbp;
    for (long s = 0; s < lines; s++) {
      if (fscanf(f.fp, "%lf %f %lf", &fbp, &dummy, &gen_pos[s]) != 3) {
        f.close();
        set_error("%s: malformed line %ld", map_fn, s + 2);
        return RL_EFORMAT;
      }
      mbp[s] = fbp;
    }
    f.close();
  }
  std::vector<double> r(L), rpos((size_t)L + 1);
  {
    size_t ir = 0, ib = 0, map_pos = 0;
    if (mbp[map_pos] > bp_pos[ib]) {
      rpos[ir++] = gen_pos[map_pos] * 1e-2;
      ib++;
    }
    for (; ir < rpos.size();) {
      while (mbp[map_pos + 1] <= bp_pos[ib] && map_pos < mbp.size() - 2) map_pos++;
      if (mbp[map_pos + 1] - mbp[map_pos] < 0) {
        set_error("genetic map is not sorted at bp %d", mbp[map_pos]);
        return RL_EFORMAT;
      }
      if (mbp[map_pos + 1] - mbp[map_pos] == 0 || mbp[map_pos] > bp_pos[ib]) {
        rpos[ir] = gen_pos[map_pos] * 1e-2;
      } else {
        rpos[ir] = ((bp_pos[ib] - mbp[map_pos]) / ((double)(mbp[map_pos + 1] - mbp[map_pos])) *
                        (gen_pos[map_pos + 1] - gen_pos[map_pos]) +
                    gen_pos[map_pos]) *
                   1e-2;
      }
      ir++;
      ib++;
    }
    const double lower_bound = 1e-10;
    for (int s = 0; s < L; s++) {
      r[s] = rpos[s + 1] - rpos[s];
      if (r[s] < lower_bound) r[s] = lower_bound;
      r[s] *= 2500;
    }
  }

  // ---- per-chunk position files (data.cpp:485-516)
  for (int chunk = 0; chunk < num_chunks; chunk++) {
    const std::string cbase = file_out + "/chunk_" + std::to_string(chunk);
    const int s0 = section_boundary_start[chunk];
    const unsigned int Lc = (unsigned int)(section_boundary_end[chunk] - s0);
    put_vec(cbase + ".bp", Lc, &bp_pos[s0]);
    put_vec(cbase + ".dist", Lc, &dist[s0]);
    put_vec(cbase + ".rpos", Lc + 1, &rpos[s0]);
    put_vec(cbase + ".r", Lc, &r[s0]);
  }
  return RL_OK;
}

// pipeline/MakeChunks.cpp:13-114: refuse an existing output directory, create it, chunk.
extern "C" int rl_stage_make_chunks(const char *haps_fn, const char *sample_fn, const char *map_fn,
                                    const char *dist_fn, const char *out_dir, int transversion, float memory_gb) {
  struct stat info;
  const std::string d = std::string(out_dir) + "/";
  if (stat(d.c_str(), &info) == 0) {
    rl::set_error("Error: Directory %s already exists. Relate will use this directory to store temporary files.",
                  out_dir);
    return RL_ESTATE;
  }
  if (mkdir(d.c_str(), 0700) != 0) {  // filesys::MakeDir (filesystem.cpp:4-23)
    rl::set_error("cannot create directory %s", out_dir);
    return RL_EIO;
  }
  return rl_make_chunks(haps_fn, sample_fn, map_fn, dist_fn, out_dir, transversion ? 0 : 1, memory_gb);
}
