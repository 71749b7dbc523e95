"""PMC traffic of the distance-head kernels (tools/bench_dist.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE).
usage: traffic_dist_summary.py <fetch_dir> <write_dir> <csrc_sha>"""
import csv, glob, json, os, sys

ALG = {"proto_dist_fwd_c16_kernel": 192, "upsample4_dist_fwd_c16_kernel": 132, "head_bwd_fused_c16_kernel": 74}
PX = 16 * 768 * 768


def load(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for k in ALG:
                if k in r["Kernel_Name"]:
                    a = out.setdefault(k, [0.0, 0])
                    a[0] += float(r["Counter_Value"]); a[1] += 1
    return out


rd, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
res = {"csrc_sha": sys.argv[3], "formula": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes", "kernels": {}}
for k in ALG:
    r, nr = rd.get(k, [0.0, 0]); w, nw = wr.get(k, [0.0, 0])
    if not nr or not nw:
        continue
    b = 2 * r * 1024 / nr + w * 1024 / nw
    res["kernels"][k] = {"launches": [nr, nw], "read_bytes_per_launch": 2 * r * 1024 / nr, "write_bytes_per_launch": w * 1024 / nw,
                         "bytes_per_launch": b, "algorithmic_bytes_per_launch": ALG[k] * PX, "ratio": b / (ALG[k] * PX)}
if "proto_dist_fwd_c16_kernel" in res["kernels"]:
    res["bytes_per_launch"] = res["kernels"]["proto_dist_fwd_c16_kernel"]["bytes_per_launch"]       # bench.py hbm_kernel.traffic
print(json.dumps(res, indent=1))
