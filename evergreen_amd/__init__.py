"""evergreen_amd -- MI355X-native implementation of Evergreen's per-distro scheduling hot path.

Scope: the tunable TaskPlanner (unit construction, scoring, rank sort,
first-occurrence dedup), GetDistroQueueInfo and UtilizationBasedHostAllocator, for many distros per
launch, behind a C ABI (include/evg_sched.h). Hand-written HIP for gfx950 in csrc/; this package only
holds the host-side mirror of the reference interface and the loader.
"""
from . import abi  # noqa: F401

__all__ = ["abi", "native", "scheduler", "gen"]
