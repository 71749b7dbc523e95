"""Summarise a rocprofv3 --pmc counter_collection CSV: per (kernel, grid) mean counter values.
usage: python tools/pmc_summary.py <dir> [name-filter]"""
import csv, glob, os, sys, collections

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
agg = collections.OrderedDict()
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "")
            if flt and flt not in name:
                continue
            short = name.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
            key = (short, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
            a = agg.setdefault(key, {})
            c = a.setdefault(r["Counter_Name"], [0.0, 0])
            c[0] += float(r["Counter_Value"]); c[1] += 1
for (k, g, w), a in agg.items():
    print("%s grid=%s wg=%s" % (k, g, w))
    for cn, (s, n) in sorted(a.items()):
        print("    %-28s %14.0f  (n=%d)" % (cn, s / n, n))
