"""relate_amd -- MI355X-native Paint -> BuildTopology path of Relate.

Host-side mirror of the C ABI in include/relate_amd.h; see relate_amd.api.
"""
import os

# BuildTopology keeps several tree-builder launches and window kernels in flight: more hardware queues than HIP's
# default four, but no more than the device keeps resident (relate_amd/csrc/main.cpp has the measurements).  Read by
# the HIP runtime when it starts, so it only counts if nothing has touched the GPU yet; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

from . import api  # noqa: F401,E402
