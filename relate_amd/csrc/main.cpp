// main.cpp -- `Relate` drop-in for the stages of this path (plus their input producer):
//   Relate --mode FindEquivalentBranches --chunk_index c -o out
//   Relate --mode MakeChunks    --haps x.haps --sample x.sample --map x.map [--memory 5] [--dist f] [--transversion] -o out
//   Relate --mode Paint         --chunk_index c -o out [--painting theta,rho]
//   Relate --mode BuildTopology --chunk_index c --first_section a --last_section b -o out
//          [--painting theta,rho] [--seed s] [--fb x] [--no_consistency]
// Same options, files and stderr banners as include/pipeline/Relate.cpp:19-115,
// Paint.cpp, BuildTopology.cpp of the reference; every other --mode is refused
// (use the reference binary for them).  Extra options: --device n,
// --sum_mode exact|lanes|lanes32, --find_equivalent_branches (with PaintBuildTopology / BuildTopology over all
// sections of a chunk: the stage downstream fused in, every .anc written once).
#include <sys/resource.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <map>
#include <string>

#include "relate_amd.h"

static void usage_line() {
  rusage usage;
  getrusage(RUSAGE_SELF, &usage);
  std::cerr << "CPU Time spent: " << usage.ru_utime.tv_sec << "." << std::setfill('0') << std::setw(6)
            << usage.ru_utime.tv_usec << "s; Max Memory usage: " << usage.ru_maxrss / 1000.0 << "Mb." << std::endl;
  std::cerr << "---------------------------------------------------------" << std::endl << std::endl;
}

int main(int argc, char **argv) {
  // BuildTopology keeps several tree-builder launches and window kernels in flight from its section threads: more
  // hardware queues than HIP's default four (read when the runtime starts; an explicit setting wins) -- but not more
  // than the device keeps resident: past 16 the queues are time-sliced and every build kernel, preempted with its
  // 150 KB of LDS, ran 1.37x longer (C3, 80 sections open: 135 s with 24 queues, 121 s with 20, 91 s with 12 or 16,
  // 96 s with 4 to 8)
  setenv("GPU_MAX_HW_QUEUES", "12", 0);
  // This process runs one stage and ends: the blocks its contexts release stay in the library's cache until the
  // process is gone instead of going back to the driver one hipFree at a time (C3: ~3000 blocks, 4 s)
  rl_keep_cache_until_exit(1);
  // option table of Relate.cpp:19-45 restricted to what the two modes read
  const std::map<std::string, bool> known = {  // name -> takes a value
      {"mode", true}, {"chunk_index", true}, {"first_section", true}, {"last_section", true},
      {"output", true}, {"painting", true}, {"seed", true}, {"fb", true}, {"sample_ages", true},
      {"no_consistency", false}, {"device", true}, {"sum_mode", true}, {"help", false},
      {"find_equivalent_branches", false},  // (PaintBuildTopology / BuildTopology of a whole chunk: the next stage fused in)
      // accepted and ignored by these two modes in the reference as well
      {"haps", true}, {"sample", true}, {"map", true}, {"mutation_rate", true}, {"effectiveN", true},
      {"memory", true}, {"dist", true}, {"annot", true}, {"coal", true}, {"transversion", false}};
  std::map<std::string, std::string> opt;
  for (int a = 1; a < argc; a++) {
    std::string s = argv[a], name;
    if (s == "-o") name = "output";
    else if (s == "-m") name = "mutation_rate";
    else if (s == "-N") name = "effectiveN";
    else if (s == "-h") name = "help";
    else if (s.rfind("--", 0) == 0) name = s.substr(2);
    else {
      std::cerr << "Unexpected argument " << s << std::endl;
      return 1;
    }
    auto it = known.find(name);
    if (it == known.end()) {  // cxxopts throws option_not_exists_exception (cxxopts.hpp:1041)
      std::cerr << "Option '" << name << "' does not exist" << std::endl;
      return 1;
    }
    if (it->second) {
      if (a + 1 >= argc) {
        std::cerr << "Option '" << name << "' is missing an argument" << std::endl;
        return 1;
      }
      opt[name] = argv[++a];
    } else {
      opt[name] = "1";
    }
  }
  if (opt.count("help") || !opt.count("mode")) {
    std::cerr << "Usage: Relate --mode Paint|BuildTopology --chunk_index c -o out [options]" << std::endl;
    return opt.count("help") ? 0 : 1;
  }
  const std::string mode = opt["mode"];
  if (!opt.count("output")) {
    std::cerr << "Not enough arguments supplied." << std::endl;
    std::cerr << "Needed: output." << std::endl;
    return 1;
  }
  const std::string out = opt["output"];
  if (out.find('/') != std::string::npos) {  // Relate.cpp:50-58
    std::cerr << "Output needs to be in working directory." << std::endl;
    return 1;
  }
  if (mode == "MakeChunks") {  // pipeline/MakeChunks.cpp:13-114
    if (!opt.count("haps") || !opt.count("sample") || !opt.count("map")) {
      std::cerr << "Not enough arguments supplied." << std::endl;
      std::cerr << "Needed: haps, sample, map, output. Optional: memory, dist, transversion." << std::endl;
      return 1;
    }
    std::cerr << "---------------------------------------------------------" << std::endl;
    std::cerr << "Parsing data.." << std::endl;
    const float memory = opt.count("memory") ? std::stof(opt["memory"]) : 5.0f;
    const int rc = rl_stage_make_chunks(opt["haps"].c_str(), opt["sample"].c_str(), opt["map"].c_str(),
                                        opt.count("dist") ? opt["dist"].c_str() : nullptr, out.c_str(),
                                        opt.count("transversion") ? 1 : 0, memory);
    if (rc != 0) {
      std::cerr << rl_last_error() << std::endl;
      return 1;
    }
    usage_line();
    return 0;
  }
  if (!opt.count("chunk_index")) {
    std::cerr << "Not enough arguments supplied." << std::endl;
    std::cerr << "Needed: chunk_index, output." << std::endl;
    return 1;
  }
  if (mode == "FindEquivalentBranches") {  // pipeline/FindEquivalentBranches.cpp:13-167
    std::cerr << "---------------------------------------------------------" << std::endl;
    std::cerr << "Propagating mutations across AncesTrees..." << std::endl;
    if (rl_stage_find_equivalent_branches(out.c_str(), std::stoi(opt["chunk_index"])) != 0) {
      std::cerr << rl_last_error() << std::endl;
      return 1;
    }
    usage_line();
    return 0;
  }
  const int chunk = atoi(opt["chunk_index"].c_str());
  const int device = opt.count("device") ? atoi(opt["device"].c_str()) : 0;
  int sum_mode = RL_SUM_EXACT;
  if (opt.count("sum_mode")) {
    if (opt["sum_mode"] == "lanes") sum_mode = RL_SUM_LANES;
    else if (opt["sum_mode"] == "lanes32") sum_mode = RL_SUM_LANES32;
    else if (opt["sum_mode"] != "exact") {
      std::cerr << "--sum_mode must be exact, lanes or lanes32" << std::endl;
      return 1;
    }
  }
  int use_painting = 0;
  double theta = 0.001, rho = 1.0;
  if (opt.count("painting")) {  // Paint.cpp:38-61: "theta,rho" parsed with std::stof
    const std::string p = opt["painting"];
    const size_t c = p.find(',');
    if (c == std::string::npos) {
      std::cerr << "--painting expects theta,rho" << std::endl;
      return 1;
    }
    theta = std::stof(p.substr(0, c));
    rho = std::stof(p.substr(c + 1));
    use_painting = 1;
  }
  int rc;
  // every option of the stage in one struct, per call (include/relate_amd.h rl_stage_opts)
  rl_stage_opts so;
  rl_stage_opts_init(&so);
  so.sum_mode = sum_mode;
  so.device = device;
  so.use_painting = use_painting;
  so.theta = theta;
  so.rho = rho;
  so.flags = opt.count("no_consistency") ? 1 : 0;
  so.fb = opt.count("fb") ? (int)std::stof(opt["fb"]) : 0;  // BuildTopology.cpp:111-114
  const std::string ages = opt.count("sample_ages") ? opt["sample_ages"] : std::string();
  so.sample_ages_path = ages.empty() ? nullptr : ages.c_str();  // BuildTopology.cpp:93-108
  so.find_equivalent_branches = opt.count("find_equivalent_branches") ? 1 : 0;
  if (mode == "Paint") {
    std::cerr << "---------------------------------------------------------" << std::endl;
    std::cerr << "Painting sequences..." << std::endl;
    rc = rl_stage_paint_ex(out.c_str(), chunk, &so);
    if (rc == 0) usage_line();
  } else if (mode == "PaintBuildTopology") {
    // Paint + BuildTopology of the chunk in one process, the stepping stones kept in HBM: what `--mode All` does per
    // chunk (Relate.cpp:257-283) without the paint files.  Sections default to all of the chunk's.
    rc = rl_stage_paint_build_topology_ex(out.c_str(), chunk, opt.count("first_section") ? atoi(opt["first_section"].c_str()) : 0,
                                          opt.count("last_section") ? atoi(opt["last_section"].c_str()) : 1 << 30, &so);
    if (rc == 1) return 1;
  } else if (mode == "BuildTopology") {
    if (!opt.count("first_section") || !opt.count("last_section")) {
      std::cerr << "Not enough arguments supplied." << std::endl;
      std::cerr << "Needed: first_section, last_section." << std::endl;
      return 1;
    }
    rc = rl_stage_build_topology_ex(out.c_str(), chunk, atoi(opt["first_section"].c_str()),
                                    atoi(opt["last_section"].c_str()), &so);
    if (rc == 1) return 1;  // first_section >= num_windows (BuildTopology.cpp:45)
  } else {
    std::cerr << "Mode " << mode << " is not part of this build: it replaces --mode MakeChunks, Paint, BuildTopology and FindEquivalentBranches "
              << "only; run the reference Relate for the other stages." << std::endl;
    return 1;
  }
  if (rc != 0) {
    std::cerr << "Error: " << rl_last_error() << std::endl;
    return 1;
  }
  return 0;
}
