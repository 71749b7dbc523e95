"""From a rocprofv3 --kernel-trace csv: for a kernel-name pattern, the names of the kernels dispatched right before and
after each match (which op of the step issues it)."""
import csv, sys, collections, glob
pat = sys.argv[2]
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
c = collections.Counter()
for i, n in enumerate(names):
    if pat in n:
        prev = names[i - 1][:60] if i else "-"
        nxt = names[i + 1][:60] if i + 1 < len(names) else "-"
        c[(prev, nxt)] += 1
for (p, n), k in c.most_common(25):
    print(k, "|", p, "|", n)
print("total", sum(c.values()), "of", len(names), "dispatches")
