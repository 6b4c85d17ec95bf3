// makechunks.cpp -- `Relate --mode MakeChunks`: .haps/.sample/map -> chunk files.
//
// Host restatement of Data::MakeChunks (src/data.cpp:117-518) with its readers
// haps (src/data.hpp:108-169, data.cpp:543-573) and map (data.cpp:593-626), and
// of the stage driver pipeline/MakeChunks.cpp.  SURVEY.md 8f-3 ("next"): the
// producer of this path's inputs, so that Paint / BuildTopology no longer need
// the reference binary upstream.  Every file it writes (parameters.bin,
// parameters_c<i>.bin, chunk_<i>.{hap,state,bp,dist,rpos,r}, props.bin) is
// byte-identical to the reference's (tests/test_makechunks.py).
#include <sys/stat.h>

#include <algorithm>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <string>
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
      std::string cmd = std::string("gunzip -c '") + fn + "'";
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

void put_field(FILE *fp, const std::string &s) {  // 1024 zero-padded bytes (data.cpp:424-436)
  char dummy[1024];
  memset(dummy, 0, sizeof dummy);
  memcpy(dummy, s.c_str(), std::min(s.size(), sizeof(dummy) - 1));
  fwrite(dummy, 1, 1024, fp);
}

template <typename T>
void put_vec(const std::string &fn, unsigned int n, const T *data) {
  FILE *fp = fopen(fn.c_str(), "wb");
  if (!fp) return;
  fwrite(&n, sizeof(unsigned int), 1, fp);
  fwrite(data, sizeof(T), n, fp);
  fclose(fp);
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
  long Llong = 0;
  if (count_newlines(haps_fn, &Llong)) {
    set_error("Failed to open file %s", haps_fn);
    return RL_EIO;
  }
  const int L = (int)Llong;
  if (N < 2 || L < 2) {
    set_error("MakeChunks: need at least 2 haplotypes and 2 SNPs (N=%d, L=%d)", N, L);
    return RL_EFORMAT;
  }

  std::vector<int> bp_pos((size_t)L + 1);
  std::vector<std::string> ancestral(L), alternative(L), rsid(L);

  // ---- chunk / window sizing (data.cpp:129-139)
  const double min_memory_size = (memory_gb) * 1e9 / 4.0 - (2.0 * N * N + 3.0 * N);
  double actual_min_memory_size = 0.0;
  if (min_memory_size <= 0) {
    set_error("Error: Need larger memory allowance.");
    return RL_EINVAL;
  }
  const int windows_per_section = 500;
  int max_windows_per_section = 0;
  const int overlap = 20000;
  int max_chunk_size = std::min(L + 1, (int)(min_memory_size / N));
  if (memory_gb >= 100) max_chunk_size = 2500000;

  std::vector<std::vector<char>> p_seq(max_chunk_size, std::vector<char>(N)), p_overlap(overlap);
  std::vector<int> window_boundaries(windows_per_section + 1), window_boundaries_overlap(windows_per_section + 1);
  std::vector<int> section_boundary_start(1, 0), section_boundary_end;
  int state_val = 1;
  int min_snps_in_window = max_chunk_size;
  int num_windows = 0, num_windows_overlap = 0, overlap_in_section = 0, chunk_size = 0, chunk_index = 0;
  double window_memory_size = 0.0;

  InFile haps;
  if (!haps.open(haps_fn)) {
    set_error("Failed to open file %s", haps_fn);
    return RL_EIO;
  }
  std::vector<char> line((size_t)2 * N + 10);
  char chr[1024], rs[1024], anc[1024], alt[1024];

  auto transition_state = [&](int snp_tmp) {
    if (use_transitions) return state_val;  // stays 1 (data.cpp:307-309)
    const std::string &a = ancestral[snp_tmp], &b = alternative[snp_tmp];
    const bool ts = (a == "C" && b == "T") || (a == "T" && b == "C") || (a == "G" && b == "A") || (a == "A" && b == "G");
    state_val = ts ? 0 : 1;
    return state_val;
  };

  int snp = 0;
  while (snp < L) {
    const std::string cbase = file_out + "/chunk_" + std::to_string(chunk_index);
    FILE *fp_hap = fopen((cbase + ".hap").c_str(), "wb");
    FILE *fp_state = fopen((cbase + ".state").c_str(), "wb");
    if (!fp_hap || !fp_state) {
      set_error("cannot write chunk files under %s", out_dir);
      haps.close();
      return RL_EIO;
    }
    if (snp > 0) {  // data.cpp:166-195: carry the last `overlap` SNPs into the next chunk
      if (snp - section_boundary_start.back() < overlap || overlap > chunk_size) {
        set_error("MakeChunks: chunk shorter than the %d-SNP overlap (raise --memory)", overlap);
        haps.close();
        return RL_EINVAL;
      }
      overlap_in_section = overlap;
      const int snp_section_begin = snp - overlap_in_section;
      section_boundary_start.push_back(snp_section_begin);
      for (int i = 0; i < overlap_in_section; i++) p_overlap[i] = p_seq[chunk_size - overlap_in_section + i];
      window_boundaries_overlap[0] = snp_section_begin;
      num_windows_overlap = 1;
      for (int w = 0; w < num_windows; w++)
        if (window_boundaries[w] > snp_section_begin) window_boundaries_overlap[num_windows_overlap++] = window_boundaries[w];
      if (num_windows_overlap >= windows_per_section - 1) {
        set_error("MakeChunks: too many windows in the overlap (raise --memory)");
        haps.close();
        return RL_EINVAL;
      }
    }

    const int snp_begin = snp;
    window_memory_size = 0.0;
    chunk_size = 0;
    window_boundaries[0] = snp_begin;
    num_windows = 1;
    int snps_in_window = 0;
    while (num_windows + num_windows_overlap < windows_per_section && chunk_size < max_chunk_size && snp < L) {
      // haps::ReadSNP (data.cpp:543-573)
      std::vector<char> &row = p_seq[chunk_size];
      if (fscanf(haps.fp, "%1023s %1023s %d %1023s %1023s", chr, rs, &bp_pos[snp], anc, alt) != 5 ||
          !fgets(line.data(), 2 * N + 10, haps.fp)) {
        set_error("%s: malformed line %d", haps_fn, snp + 1);
        haps.close();
        return RL_EFORMAT;
      }
      int filled = 0;
      for (int i = 0; line[i] != '\0' && filled < N; i++)
        if (line[i] == '0' || line[i] == '1') row[filled++] = line[i];
      if (filled != N) {
        set_error("%s: SNP %s %s %d has %d alleles, %d expected", haps_fn, chr, rs, bp_pos[snp], filled, N);
        haps.close();
        return RL_EFORMAT;
      }
      ancestral[snp] = anc;
      alternative[snp] = alt;
      rsid[snp] = rs;

      int num_derived = 0;
      for (char ch : row) num_derived += (ch == '1');
      window_memory_size += num_derived * (N + 1);
      if (window_memory_size >= min_memory_size && snps_in_window > 10) {  // data.cpp:219-229
        if (actual_min_memory_size < window_memory_size) actual_min_memory_size = window_memory_size;
        if (min_snps_in_window > snps_in_window) min_snps_in_window = snps_in_window;
        snps_in_window = 0;
        window_memory_size = 0.0;
        window_boundaries[num_windows] = snp;
        num_windows++;
      }
      snp++;
      snps_in_window++;
      chunk_size++;
    }
    if (actual_min_memory_size < window_memory_size) actual_min_memory_size = window_memory_size;
    if (min_snps_in_window > snps_in_window) min_snps_in_window = snps_in_window;
    const float mean_snps_in_window = chunk_size / num_windows;
    window_boundaries[num_windows] = snp;
    if (num_windows > max_windows_per_section) max_windows_per_section = num_windows;
    if (mean_snps_in_window < 100) {
      std::cerr << "Memory allowance should be set " << 100 / mean_snps_in_window << " times larger than" << std::endl;
      std::cerr << "the current setting using --memory (Default 5GB)." << std::endl;
    }
    section_boundary_end.push_back(snp);

    int snp_tmp = section_boundary_start.back();
    const uint64_t uN = (uint64_t)N;
    const std::string pfn = file_out + "/parameters_c" + std::to_string(chunk_index) + ".bin";
    FILE *fp = fopen(pfn.c_str(), "w");
    if (!fp) {
      set_error("cannot write %s", pfn.c_str());
      haps.close();
      return RL_EIO;
    }
    if (snp_begin == 0) {  // data.cpp:254-270
      const uint64_t uL = (uint64_t)chunk_size;
      fwrite(&uL, 8, 1, fp_hap);
      fwrite(&uN, 8, 1, fp_hap);
      const int nw = num_windows + 1;
      fwrite(&N, 4, 1, fp);
      fwrite(&chunk_size, 4, 1, fp);
      fwrite(&nw, 4, 1, fp);
      fwrite(window_boundaries.data(), 4, nw, fp);
      fclose(fp);
      fwrite(&chunk_size, 4, 1, fp_state);
    } else {  // data.cpp:272-325
      const int L_chunk = chunk_size + overlap_in_section;
      const uint64_t uL = (uint64_t)L_chunk;
      fwrite(&uL, 8, 1, fp_hap);
      fwrite(&uN, 8, 1, fp_hap);
      const int window_start = window_boundaries_overlap[0];
      for (int w = 0; w < num_windows_overlap; w++) window_boundaries_overlap[w] -= window_start;
      for (int w = 0; w <= num_windows; w++) window_boundaries[w] -= window_start;
      const int nw = num_windows + num_windows_overlap + 1;
      fwrite(&N, 4, 1, fp);
      fwrite(&L_chunk, 4, 1, fp);
      fwrite(&nw, 4, 1, fp);
      fwrite(window_boundaries_overlap.data(), 4, num_windows_overlap, fp);
      fwrite(window_boundaries.data(), 4, num_windows + 1, fp);
      fclose(fp);
      for (int w = 0; w <= num_windows; w++) window_boundaries[w] += window_start;
      fwrite(&L_chunk, 4, 1, fp_state);
      for (int i = 0; i < overlap_in_section; i++) {
        const int sv = transition_state(snp_tmp);
        fwrite(&sv, 4, 1, fp_state);
        snp_tmp++;
        fwrite(p_overlap[i].data(), 1, (size_t)N, fp_hap);
      }
    }
    for (int i = 0; i < chunk_size; i++) {
      const int sv = transition_state(snp_tmp);
      fwrite(&sv, 4, 1, fp_state);
      snp_tmp++;
      fwrite(p_seq[i].data(), 1, (size_t)N, fp_hap);
    }
    fclose(fp_hap);
    fclose(fp_state);
    chunk_index++;
  }
  bp_pos[L] = bp_pos[L - 1] + 1;
  haps.close();
  const int num_chunks = (int)section_boundary_start.size();

  std::cerr << std::setprecision(2) << "Warning: Will use min "
            << 2.0 * (4.0 * N * N * (max_windows_per_section + 2.0)) / 1e9 << "GB of hard disc." << std::endl;

  {  // parameters.bin (data.cpp:361-375)
    FILE *fp = fopen((file_out + "/parameters.bin").c_str(), "w");
    if (!fp) {
      set_error("cannot write parameters.bin");
      return RL_EIO;
    }
    actual_min_memory_size += (2 * N * N + 3 * N);
    actual_min_memory_size *= 4.0 / 1e9;
    fwrite(&N, 4, 1, fp);
    fwrite(&L, 4, 1, fp);
    fwrite(&num_chunks, 4, 1, fp);
    fwrite(&actual_min_memory_size, 8, 1, fp);
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
    for (int s = 0; s < L; s++) {
      fwrite(&s, 4, 1, fp);
      fwrite(&bp_pos[s], 4, 1, fp);
      fwrite(&dist[s], 4, 1, fp);
      put_field(fp, rsid[s]);
      put_field(fp, ancestral[s]);
      put_field(fp, alternative[s]);
    }
    fclose(fp);
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
