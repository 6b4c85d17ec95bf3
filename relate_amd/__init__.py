"""relate_amd -- MI355X-native Paint -> BuildTopology path of Relate.

Host-side mirror of the C ABI in include/relate_amd.h; see relate_amd.api.
"""
from . import api  # noqa: F401
