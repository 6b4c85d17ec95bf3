#include "common.h"
using namespace rl;
extern "C" int rl_quickbuild(int, double, float *, const float *, int *, int *, int *) { set_error("not implemented"); return RL_ESTATE; }
