"""Utilities: profiling helpers (rocprofv3 command builder, CSV summarisers, step timer)."""
from .profiling import (PMC_SLOTS, StepTimer, kernel_time_by_name, pivot_counters, rocprof_command,  # noqa: F401
                        split_counter_passes, summarize_kernel_trace)
