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
print("%-60s %s" % ("kernel", " ".join("%14s" % k[-14:] for k in keys)))
for did in list(disp)[-n:]:
    d = disp[did]
    nm = d["name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58]
    print("%-60s %s" % (nm, " ".join("%14.4g" % d.get(k, 0) for k in keys)))
