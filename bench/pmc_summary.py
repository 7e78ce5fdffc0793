"""Pivot a rocprofv3 counter_collection CSV: one row per dispatch (last N), one column per counter."""
import csv
import sys
from collections import OrderedDict

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(csv.DictReader(open(path)))
disp = OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", "")})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
keys = sorted({k for d in disp.values() for k in d if k not in ("name", "grid")})
print("%-60s %s %14s" % ("kernel", " ".join("%14s" % k[-14:] for k in keys), "mfma_pipe_busy"))
for did in list(disp)[-n:]:
    d = disp[did]
    nm = d["name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58]
    # Share of the kernel's duration during which a matrix pipe was executing, averaged over the chip's 1024 pipes
    # (256 CUs x 4 SIMDs): SQ_VALU_MFMA_BUSY_CYCLES is summed over all pipes; GRBM_GUI_ACTIVE is the kernel's active
    # cycles summed over the 8 XCDs.  Both tick with the graphics clock the counters run on, which is not necessarily the
    # clock the shader sustains under this load, so the figure compares kernels rather than measuring peak fraction.
    act = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    share = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * act) if act else float("nan")
    print("%-60s %s %13.1f%%" % (nm, " ".join("%14.4g" % d.get(k, 0) for k in keys), 100.0 * share))
