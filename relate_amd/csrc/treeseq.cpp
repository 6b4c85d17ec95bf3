#include "common.h"
using namespace rl;
extern "C" int rl_stage_build_topology(const char *, int, int, int, int, double, double, int, int, int, int) { set_error("not implemented"); return RL_ESTATE; }
