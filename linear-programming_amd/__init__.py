"""MI355X-native dense simplex-tableau engine (host-side Python binding).

The compute path is the HIP library `lib/liblpamd.so` (C ABI in
include/lpamd.h).  There is no CPU fallback: importing `engine` without the
built library, or creating an engine without a GPU, raises.
"""
from .engine import (  # noqa: F401
    ERROR, UNBOUNDED, NOSOLUTION, SOLVED, LIMIT, MAX_COST, MAX_REDUCTION, STRATEGY,
    LpamdError, RevisedTableauEngine, ShardGroup, TableauEngine, comm_unique_id, library, library_path,
)
