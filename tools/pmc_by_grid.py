"""Summarise a rocprofv3 --pmc counter_collection csv: mean counter value per (kernel, grid size)."""
import csv, glob, collections, sys
root, pat = sys.argv[1], sys.argv[2]
d = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:44], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print("%-46s grid=%-9s %-24s n=%-4d mean=%.4g" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
