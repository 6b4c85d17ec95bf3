#include "paint_device.h"
#include "launch.h"
namespace rl { hipError_t launch_matrix(const MatrixParams &, const Layout &, int, hipStream_t) { return hipErrorNotSupported; } }
