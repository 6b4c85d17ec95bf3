// window.cpp -- placeholder, replaced below
#include "common.h"
using namespace rl;
extern "C" {
rl_window *rl_window_open(rl_ctx *, int, const char *, int, int, float *) { set_error("not implemented"); return nullptr; }
void rl_window_close(rl_window *) {}
int rl_window_bounds(const rl_window *, int *, int *) { return RL_ESTATE; }
int rl_window_rows(const rl_window *, int) { return RL_ESTATE; }
int rl_window_get_topology(rl_window *, int, float *, float *) { return RL_ESTATE; }
int rl_window_advance(rl_window *, int) { return RL_ESTATE; }
int rl_window_matrix(rl_window *, int, float *, float *) { return RL_ESTATE; }
}
