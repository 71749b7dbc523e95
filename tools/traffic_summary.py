"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per kernel class per step.
usage: traffic_summary.py <fetch_dir> <write_dir> <steps_in_run> [csrc_sha (bench.py --print-csrc-sha)]"""
import csv, glob, json, os, sys, collections

def load(d, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
                a = agg[n]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg

rd, wr, steps = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
out = {"csrc_sha": sys.argv[4] if len(sys.argv) > 4 else None, "formula": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes, per training step", "kernels": {}}
tot = 0.0
for k in sorted(set(rd) | set(wr)):
    r, nr = rd.get(k, [0.0, 0]); w, nw = wr.get(k, [0.0, 0])
    b = (2 * r + w) * 1024 / steps
    tot += b
    out["kernels"][k] = {"read_GB": 2 * r * 1024 / steps / 1e9, "write_GB": w * 1024 / steps / 1e9, "GB_per_step": b / 1e9,
                         "launches_per_step": max(nr, nw) / steps}
out["total_GB_per_step"] = tot / 1e9
# the convolution entry points' own launches: the implicit-GEMM kernels and the split-K fold of the weight gradients
conv = [v for k, v in out["kernels"].items() if k.startswith("conv_") or k.startswith("wgrad_reduce")]
out["conv_GB_per_step"] = sum(v["GB_per_step"] for v in conv)
out["conv_launches_per_step"] = sum(v["launches_per_step"] for v in conv)
print(json.dumps(out, indent=1))
