"""Profiling helpers: a rocprofv3 command builder that respects the counter-slot limits of gfx950, summarisers for
its CSV output, and a host-side step timer.

The reference has no profiler usage; its only timing artifact is the ``global_step/sec`` summary that
``MonitoredTrainingSession`` installs when ``checkpoint_dir`` is set (/root/reference/example.py:189-192), which
``session.StepCounterHook`` reproduces.  This module adds what a kernel-level framework needs on top: per-kernel
time and hardware counters.
"""
from __future__ import annotations

import csv
import time
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Sequence

# Hardware counter slots per block in ONE rocprofv3 pass (it does not split a request over passes; asking for more
# makes it fail).  TCC_*FETCH_SIZE* costs 3 of the 4 TCC slots and *WRITE_SIZE* 2, so those two need separate passes.
PMC_SLOTS = {"SQ": 8, "TCC": 4, "TCP": 4, "TA": 2, "TD": 2, "GRBM": 2}
_TCC_COST = (("FETCH_SIZE", 3), ("WRITE_SIZE", 2))

# Tracing domains that must not be combined with counter collection in one run.
_NO_PMC_WITH = ("--sys-trace", "-s", "--runtime-trace", "-r", "--hip-trace", "--hsa-trace", "--memory-copy-trace",
                "--scratch-memory-trace", "--marker-trace")


def _base_name(counter: str) -> str:
    for suffix in ("_sum", "_avr", "_min", "_max"):
        if counter.endswith(suffix):
            return counter[:-len(suffix)]
    return counter


def split_counter_passes(counters: Sequence[str]) -> List[List[str]]:
    """Partition ``counters`` into passes that each fit the per-block slot limits (first-fit, order kept)."""
    passes: List[Dict] = []
    for c in counters:
        block = c.split("_", 1)[0]
        if block not in PMC_SLOTS:
            raise ValueError("unknown counter block in %r (have %s)" % (c, ", ".join(PMC_SLOTS)))
        cost = 1
        if block == "TCC":
            for key, k in _TCC_COST:
                if key in c:
                    cost = k
        for p in passes:
            if _base_name(c) in p["names"]:
                p["list"].append(c)          # _sum/_avr/... of a counter already in the pass are free
                break
            if p["used"].get(block, 0) + cost <= PMC_SLOTS[block]:
                p["used"][block] = p["used"].get(block, 0) + cost
                p["names"].add(_base_name(c))
                p["list"].append(c)
                break
        else:
            if cost > PMC_SLOTS[block]:
                raise ValueError("%s alone exceeds the %s slots" % (c, block))
            passes.append({"used": {block: cost}, "names": {_base_name(c)}, "list": [c]})
    return [p["list"] for p in passes]


def rocprof_command(program: Sequence[str], out_dir: str, name: str = "run", counters: Optional[Sequence[str]] = None,
                    kernel_trace: bool = True, stats: bool = True, extra: Iterable[str] = ()) -> List[str]:
    """``rocprofv3`` argv for ONE pass.  With ``counters`` the pass collects exactly those (they must fit one pass,
    see :func:`split_counter_passes`) together with kernel tracing only.  ``program`` must be the executable itself
    (``["python3", "bench.py", ...]``): wrappers that re-exec after the profiler's library initialised the GPU are
    not allowed."""
    extra = list(extra)
    if counters:
        if len(split_counter_passes(counters)) != 1:
            raise ValueError("counters do not fit one pass: %s" % (split_counter_passes(counters),))
        bad = [e for e in extra if e in _NO_PMC_WITH]
        if bad:
            raise ValueError("counter collection cannot be combined with %s" % bad)
    if program and program[0] in ("env", "taskset", "numactl", "bash", "sh"):
        raise ValueError("put the program itself after '--', not a launcher (%r)" % program[0])
    cmd = ["rocprofv3"]
    if kernel_trace:
        cmd.append("--kernel-trace")
    if stats and not counters:
        cmd.append("--stats")
    if counters:
        cmd += ["--pmc", *counters]
    cmd += ["-d", out_dir, "-o", name, "--output-format", "csv", *extra, "--", *program]
    return cmd


def summarize_kernel_trace(path: str, last_steps_marker: Optional[str] = None) -> List[Dict]:
    """Rows of a ``*_kernel_trace.csv`` sorted by start time: name, start offset (us), duration (us)."""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if not rows:
        return []
    t0 = int(rows[0]["Start_Timestamp"])
    out = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.append({"name": r["Kernel_Name"], "start_us": (s - t0) / 1e3, "dur_us": (e - s) / 1e3})
    return out


def kernel_time_by_name(path: str) -> "OrderedDict[str, Dict[str, float]]":
    """Total / mean time per kernel name from a ``*_kernel_trace.csv``, largest total first."""
    agg: Dict[str, List[float]] = {}
    for r in summarize_kernel_trace(path):
        agg.setdefault(r["name"], []).append(r["dur_us"])
    items = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    return OrderedDict((k, {"calls": len(v), "total_us": sum(v), "mean_us": sum(v) / len(v)}) for k, v in items)


def pivot_counters(path: str) -> List[Dict]:
    """One dict per dispatch of a ``*_counter_collection.csv``: kernel name plus one key per counter."""
    disp: "OrderedDict[int, Dict]" = OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(disp.values())


class StepTimer:
    """Host-side ``steps/sec`` and ``examples/sec`` over windows of ``every`` steps (no device synchronisation: on
    an asynchronous device the rate is that of the launch stream, which equals the device rate once queues fill)."""

    def __init__(self, every: int = 100, batch_size: Optional[int] = None):
        self.every = int(every)
        self.batch_size = batch_size
        self._t = time.perf_counter()
        self._n = 0
        self.last: Dict[str, float] = {}

    def step(self, n: int = 1) -> Optional[Dict[str, float]]:
        self._n += n
        if self._n < self.every:
            return None
        now = time.perf_counter()
        rate = self._n / max(now - self._t, 1e-12)
        self.last = {"global_step/sec": rate}
        if self.batch_size:
            self.last["examples/sec"] = rate * self.batch_size
        self._t, self._n = now, 0
        return self.last
