"""Group a rocprofv3 kernel trace (csv) by kernel name and grid size: count, median, mean duration."""
import csv, glob, collections, sys
root, pat = sys.argv[1], sys.argv[2]
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        g = r.get("Grid_Size_X") or r.get("Grid_Size")
        d[(r["Kernel_Name"][:48], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v = sorted(v)
    print(k, len(v), "median %.2f us  mean %.2f us" % (v[len(v) // 2] / 1e3, sum(v) / len(v) / 1e3))
