// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A small driver that links the *reference's own* sources (compiled where
// they lie under /root/reference by oracle/Makefile; nothing is copied) and
// dumps intermediate results of the Paint -> BuildTopology path so that the
// oracle restatement and the product can be pinned against the real thing:
//
//   ref_harness repaint  <outdir> <chunk> <window> <dump.bin>
//       RePaintSection for every target of one window
//       dump: int N; per target: int D; float logscales[D]; float top[D*N]
//   ref_harness matrix   <outdir> <chunk> <window> <dump.bin> snp [snp ...]
//       DistanceMeasure::GetMatrix at the window start and at each listed SNP
//       (cursors advanced the way AncesTreeBuilder::BuildTopology does)
//       dump: int N; per matrix: int snp; float d[N*N]
//   ref_harness quickbuild <N> <d.bin> <parents.out> [<prior.bin>]
//   ref_harness quickbuild_seq <N> <parents.out> (<d.bin> <prior.bin|->)...
//       MinMatch::QuickBuild on a raw N*N float matrix (with optional prior)
//       output: int parent[2N-1]
//   ref_harness treeseq  <outdir> <chunk> <window> <dump.bin>
//       runs AncesTreeBuilder::BuildTopology and dumps, for every tree,
//       int pos; int parent[2N-1]
//   ref_harness paint_targets <outdir> <chunk> <dump.bin> k [k ...]
//       FastPainting::PaintSteppingStones for the listed targets only (the
//       loop body of pipeline/Paint.cpp:81-87), every window's record kept
//       dump: per target, per window: int k; int w; int len; bytes[len]
//   ref_harness paint_window <outdir> <chunk> <window> <k0> <k1> <part.bin>
//       PaintSteppingStones for targets [k0, k1); only <window>'s records
//       are kept (the others go to /dev/null): the parts of consecutive
//       ranges concatenated ARE the reference's relate_<window>.bin
//   ref_harness paint_windows <outdir> <chunk> <k0> <k1> <prefix> w [w ...]
//       as paint_window for SEVERAL windows from one painting of the targets
//       (a target is painted over the whole chunk whatever is kept):
//       window w's records go to <prefix>_<w>.bin
//   ref_harness repaint_targets <outdir> <chunk> <window> <dump.bin> k [k ...]
//       as repaint, for the listed targets only (ascending)
//
// This file only exists in this container's workflow: /root/reference does
// not travel to the GPU box; the fixtures it produces do (tests/golden/).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "anc.hpp"
#include "anc_builder.hpp"
#include "data.hpp"
#include "fast_painting.hpp"
#include "tree_builder.hpp"

static void read_params(const std::string &out, int chunk, int &N, int &L,
                        std::vector<int> &wb) {
  FILE *fp = fopen((out + "/parameters_c" + std::to_string(chunk) + ".bin").c_str(), "rb");
  if (!fp) { fprintf(stderr, "cannot open parameters\n"); exit(1); }
  int nw;
  if (fread(&N, 4, 1, fp) != 1 || fread(&L, 4, 1, fp) != 1 || fread(&nw, 4, 1, fp) != 1) exit(1);
  wb.resize(nw);
  if (fread(&wb[0], 4, nw, fp) != (size_t)nw) exit(1);
  fclose(fp);
}

static Data *load(const std::string &out, int chunk) {
  std::string b = out + "/chunk_" + std::to_string(chunk);
  Data *d = new Data((b + ".hap").c_str(), (b + ".bp").c_str(), (b + ".dist").c_str(),
                     (b + ".r").c_str(), (b + ".rpos").c_str(), (b + ".state").c_str());
  d->name = b + "/paint/relate";
  d->N = d->sequence.size() ? (int)d->sequence.subVectorSize(0) : 0;
  d->L = (int)d->sequence.size();
  return d;
}

static void apply_painting(Data &data) {
  const char *p = getenv("REF_PAINTING");  // "theta,rho" like --painting
  if (!p) return;
  std::string s(p), a, b;
  size_t c = s.find(',');
  a = s.substr(0, c);
  b = s.substr(c + 1);
  data.theta = std::stof(a);
  data.ntheta = 1.0 - data.theta;
  double rho = std::stof(b);
  for (auto &x : data.r) x *= rho;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  std::string mode = argv[1];

  if (mode == "quickbuild_seq") {
    // ref_harness quickbuild_seq <N> <parents.out> (<d.bin> <prior.bin|->)...
    // ONE MinMatch for the whole sequence, as AncesTreeBuilder::BuildTopology keeps one per section
    // (anc_builder.cpp:436): what it carries from build to build is part of the result.
    int N = atoi(argv[2]);
    Data data(N, 1);
    MinMatch tb(data);
    std::vector<double> ages;
    FILE *fo = fopen(argv[3], "wb");
    if (!fo) return 1;
    for (int a = 4; a + 1 < argc; a += 2) {
      CollapsedMatrix<float> d, prior;
      d.resize(N, N);
      FILE *fp = fopen(argv[a], "rb");
      if (!fp || fread(&d[0][0], 4, (size_t)N * N, fp) != (size_t)N * N) return 1;
      fclose(fp);
      Tree tree;
      if (std::string(argv[a + 1]) != "-") {
        prior.resize(N, N);
        fp = fopen(argv[a + 1], "rb");
        if (!fp || fread(&prior[0][0], 4, (size_t)N * N, fp) != (size_t)N * N) return 1;
        fclose(fp);
        tb.QuickBuild(d, tree, ages, prior);
      } else {
        tb.QuickBuild(d, tree, ages);
      }
      for (int i = 0; i < 2 * N - 1; i++) {
        int p = tree.nodes[i].parent ? (*tree.nodes[i].parent).label : -1;
        fwrite(&p, 4, 1, fo);
      }
    }
    fclose(fo);
    return 0;
  }

  if (mode == "quickbuild_seq_ages") {
    // ref_harness quickbuild_seq_ages <N> <parents.out> <ages.bin: N doubles> (<d.bin> <prior.bin|->)...
    // as quickbuild_seq, with MinMatch::QuickBuild's sample_ages (ancient samples; Ne as BuildTopology.cpp:36 sets it)
    int N = atoi(argv[2]);
    Data data(N, 1);
    data.Ne = std::max(17.5f * data.N, 30000.0f);
    MinMatch tb(data);
    std::vector<double> ages(N);
    FILE *fa = fopen(argv[4], "rb");
    if (!fa || fread(&ages[0], 8, (size_t)N, fa) != (size_t)N) return 1;
    fclose(fa);
    FILE *fo = fopen(argv[3], "wb");
    if (!fo) return 1;
    for (int a = 5; a + 1 < argc; a += 2) {
      CollapsedMatrix<float> d, prior;
      d.resize(N, N);
      FILE *fp = fopen(argv[a], "rb");
      if (!fp || fread(&d[0][0], 4, (size_t)N * N, fp) != (size_t)N * N) return 1;
      fclose(fp);
      Tree tree;
      if (std::string(argv[a + 1]) != "-") {
        prior.resize(N, N);
        fp = fopen(argv[a + 1], "rb");
        if (!fp || fread(&prior[0][0], 4, (size_t)N * N, fp) != (size_t)N * N) return 1;
        fclose(fp);
        tb.QuickBuild(d, tree, ages, prior);
      } else {
        tb.QuickBuild(d, tree, ages);
      }
      for (int i = 0; i < 2 * N - 1; i++) {
        int p = tree.nodes[i].parent ? (*tree.nodes[i].parent).label : -1;
        fwrite(&p, 4, 1, fo);
      }
    }
    fclose(fo);
    return 0;
  }

  if (mode == "quickbuild") {
    int N = atoi(argv[2]);
    Data data(N, 1);
    CollapsedMatrix<float> d, prior;
    d.resize(N, N);
    FILE *fp = fopen(argv[3], "rb");
    if (!fp || fread(&d[0][0], 4, (size_t)N * N, fp) != (size_t)N * N) return 1;
    fclose(fp);
    MinMatch tb(data);
    Tree tree;
    std::vector<double> ages;
    if (argc > 5) {
      prior.resize(N, N);
      fp = fopen(argv[5], "rb");
      if (!fp || fread(&prior[0][0], 4, (size_t)N * N, fp) != (size_t)N * N) return 1;
      fclose(fp);
      tb.QuickBuild(d, tree, ages, prior);
    } else {
      tb.QuickBuild(d, tree, ages);
    }
    fp = fopen(argv[4], "wb");
    for (int i = 0; i < 2 * N - 1; i++) {
      int p = tree.nodes[i].parent ? (*tree.nodes[i].parent).label : -1;
      fwrite(&p, 4, 1, fp);
    }
    fclose(fp);
    return 0;
  }

  if (mode == "paint_targets" || mode == "paint_window" || mode == "paint_windows") {
    std::string out = argv[2];
    int chunk = atoi(argv[3]);
    int N, L;
    std::vector<int> wb;
    read_params(out, chunk, N, L, wb);
    int W = (int)wb.size() - 1;
    Data *pd = load(out, chunk);
    Data &data = *pd;
    apply_painting(data);
    if (mode == "paint_targets") {
      FILE *fo = fopen(argv[4], "wb");
      if (!fo) return 1;
      std::vector<char> buf;
      for (int a = 5; a < argc; a++) {
        int k = atoi(argv[a]);
        std::vector<FILE *> pfiles(W);
        for (int w = 0; w < W; w++) {
          pfiles[w] = tmpfile();
          if (!pfiles[w]) return 1;
        }
        FastPainting painter(data);
        painter.PaintSteppingStones(data, wb, pfiles, k);
        for (int w = 0; w < W; w++) {
          int len = (int)ftell(pfiles[w]);
          buf.resize(len);
          rewind(pfiles[w]);
          if (fread(buf.data(), 1, len, pfiles[w]) != (size_t)len) return 1;
          fclose(pfiles[w]);
          fwrite(&k, 4, 1, fo);
          fwrite(&w, 4, 1, fo);
          fwrite(&len, 4, 1, fo);
          fwrite(buf.data(), 1, len, fo);
        }
        fflush(fo);
      }
      fclose(fo);
      return 0;
    }
    if (mode == "paint_windows") {
      const int k0 = atoi(argv[4]), k1 = atoi(argv[5]);
      const std::string prefix = argv[6];
      FILE *fnull = fopen("/dev/null", "wb");
      if (!fnull || argc < 8) return 1;
      std::vector<FILE *> pfiles(W, fnull), mine;
      for (int a = 7; a < argc; a++) {
        const int w = atoi(argv[a]);
        if (w < 0 || w >= W || pfiles[w] != fnull) return 1;
        pfiles[w] = fopen((prefix + "_" + std::to_string(w) + ".bin").c_str(), "wb");
        if (!pfiles[w]) return 1;
        mine.push_back(pfiles[w]);
      }
      for (int k = k0; k < k1; k++) {
        FastPainting painter(data);
        painter.PaintSteppingStones(data, wb, pfiles, k);
        if ((k - k0) % 25 == 0) {
          for (FILE *f : mine) fflush(f);
          fprintf(stderr, "[paint_windows] %d/%d\n", k - k0, k1 - k0);
        }
      }
      for (FILE *f : mine) fclose(f);
      fclose(fnull);
      return 0;
    }
    int window = atoi(argv[4]), k0 = atoi(argv[5]), k1 = atoi(argv[6]);
    FILE *fo = fopen(argv[7], "wb");
    FILE *fnull = fopen("/dev/null", "wb");
    if (!fo || !fnull || window < 0 || window >= W) return 1;
    std::vector<FILE *> pfiles(W, fnull);
    pfiles[window] = fo;
    for (int k = k0; k < k1; k++) {
      FastPainting painter(data);
      painter.PaintSteppingStones(data, wb, pfiles, k);
      if ((k - k0) % 25 == 0) { fflush(fo); fprintf(stderr, "[paint_window %d] %d/%d\n", window, k - k0, k1 - k0); }
    }
    fclose(fo);
    fclose(fnull);
    return 0;
  }

  std::string out = argv[2];
  int chunk = atoi(argv[3]);
  int window = atoi(argv[4]);
  int N, L;
  std::vector<int> wb;
  read_params(out, chunk, N, L, wb);
  Data *pd = load(out, chunk);
  Data &data = *pd;
  apply_painting(data);
  FILE *fo = fopen(argv[5], "wb");
  if (!fo) return 1;

  if (mode == "repaint" || mode == "repaint_targets") {
    std::vector<char> wanted(N, mode == "repaint");
    for (int a = 6; a < argc && mode == "repaint_targets"; a++) wanted[atoi(argv[a])] = 1;
    FastPainting painter(data);
    char fn[2048];
    snprintf(fn, sizeof fn, "%s_%i.bin", data.name.c_str(), window);
    FILE *fp = fopen(fn, "rb");
    if (!fp) return 1;
    fwrite(&N, 4, 1, fo);
    for (int n = 0; n < N; n++) {
      int s0, s1, bb, be;
      float la, lb;
      CollapsedMatrix<float> ab, bend, top;
      std::vector<float> ls;
      if (fread(&s0, 4, 1, fp) != 1 || fread(&s1, 4, 1, fp) != 1) return 1;
      ab.ReadFromFile(fp, bb, la);
      bend.ReadFromFile(fp, be, lb);
      if (!wanted[n]) continue;
      if (mode == "repaint_targets") fwrite(&n, 4, 1, fo);
      painter.RePaintSection(data, top, ls, ab, bend, bb, be, la, lb, n);
      int D = (int)ls.size();
      fwrite(&D, 4, 1, fo);
      fwrite(&ls[0], 4, D, fo);
      for (int i = 0; i < D; i++) fwrite(&top[i][0], 4, N, fo);
    }
    fclose(fp);
  } else if (mode == "matrix") {
    int start = wb[window];
    DistanceMeasure d(data, window);
    fwrite(&N, 4, 1, fo);
    d.GetMatrix(start);
    fwrite(&start, 4, 1, fo);
    fwrite(&d.matrix[0][0], 4, (size_t)N * N, fo);
    int cur = start;
    for (int a = 6; a < argc; a++) {
      int snp = atoi(argv[a]);
      for (int s = cur + 1; s <= snp; s++) {
        for (int i = 0; i < N; i++) {
          if (data.sequence[s][i] == '1') {
            d.v_snp_prev[i]++;
            d.v_rpos_prev[i] = data.rpos[s];
          }
        }
      }
      cur = snp;
      d.GetMatrix(snp);
      fwrite(&snp, 4, 1, fo);
      fwrite(&d.matrix[0][0], 4, (size_t)N * N, fo);
    }
  } else if (mode == "treeseq") {
    data.Ne = std::max(17.5f * data.N, 30000.0f);
    std::vector<double> ages;
    AncesTree anc;
    AncesTreeBuilder ab(data, ages, 1);
    int start = wb[window];
    int end = (window < (int)wb.size() - 2) ? wb[window + 1] - 1 : data.L - 1;
    ab.BuildTopology(window, start, end, data, anc, 1, true, 0);
    fwrite(&N, 4, 1, fo);
    for (auto &mt : anc.seq) {
      fwrite(&mt.pos, 4, 1, fo);
      for (int i = 0; i < 2 * N - 1; i++) {
        int p = mt.tree.nodes[i].parent ? (*mt.tree.nodes[i].parent).label : -1;
        fwrite(&p, 4, 1, fo);
      }
    }
  } else {
    return 2;
  }
  fclose(fo);
  return 0;
}
