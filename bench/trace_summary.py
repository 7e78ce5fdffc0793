"""Print the kernel timeline of the last training step from a rocprofv3 kernel-trace CSV."""
import csv
import sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
base = None
tot = 0.0
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if base is None:
        base = s
    tot += (e - s) / 1000
    print("%9.1f %8.1fus grid=%s,%s,%s wg=%s vgpr=%s agpr=%s lds=%s %s" % (
        (s - base) / 1000, (e - s) / 1000, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"],
        r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"),
        r["Kernel_Name"].replace("(anonymous namespace)::", "")[:80]))
print("sum of kernel time: %.1f us" % tot)
