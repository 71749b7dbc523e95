"""Per-class / per-kernel-template summaries of rocprofv3 --pmc passes over a SERIAL bench.py run (DML_OVERLAP_WGRAD=0).

  traffic : python tools/pmc_by_class.py traffic <fetch_dir> <write_dir> <oplog.json> <csrc_sha>
            HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM; separate passes) per kernel name
            AND per conv shape class: the profiler's dispatch sequence of conv kernels is zipped with the plan's conv launch
            list that bench.py wrote (DML_BENCH_OPLOG); a weight gradient's split-K fold (wgrad_reduce*) is charged to the
            weight-gradient launch before it.
  sq      : python tools/pmc_by_class.py sq <sq_dir> <oplog.json> <kernel_stats.csv> <csrc_sha>
            SQ counters per kernel TEMPLATE (full instantiation name) per step, next to the kernel's time from the
            --kernel-trace --stats run of the same command, and per conv shape class.
            mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES): both counters come back summed per shader
            engine (32 engines x 32 SIMDs; SQ_BUSY_CYCLES per launch = the launch's duration in cycles), MFMA busy cycles
            are summed over an engine's 32 SIMDs.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].strip()


def dispatches(d, counters):
    """[(dispatch id, kernel name, {counter: value})] in dispatch order"""
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] not in counters:
                    continue
                key = int(r["Dispatch_Id"])
                e = rows.setdefault(key, [short(r["Kernel_Name"]), {}])
                e[1][r["Counter_Name"]] = e[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [(k, v[0], v[1]) for k, v in sorted(rows.items())]


def zip_with_ops(disp, ops, steps):
    """conv dispatches (one 'conv_*' kernel per plan conv launch, followed by its optional wgrad_reduce*) -> per op index"""
    per_op = [collections.defaultdict(float) for _ in ops]
    conv = [(i, n, c) for i, (_, n, c) in enumerate(disp) if n.startswith("conv_")]
    if len(conv) != steps * len(ops):
        raise SystemExit("dispatch sequence does not match the plan: %d conv dispatches, %d steps x %d launches"
                         % (len(conv), steps, len(ops)))
    for j, (pos, name, cnt) in enumerate(conv):
        op = per_op[j % len(ops)]
        for k, v in cnt.items():
            op[k] += v / steps
        op["_n"] += 1.0 / steps
        nxt = pos + 1
        while nxt < len(disp) and disp[nxt][1].startswith("wgrad_reduce"):
            for k, v in disp[nxt][2].items():
                op[k] += v / steps
            nxt += 1
    return per_op


def per_class(per_op, ops, keys):
    rows = {}
    for op, cnt in zip(ops, per_op):
        ftot = sum(m["flops"] for m in op["members"])
        for m in op["members"]:
            r = rows.setdefault((m["kind"], m["label"]), collections.defaultdict(float))
            share = m["flops"] / ftot
            for k in keys:
                r[k] += cnt.get(k, 0.0) * share
            r["launches"] += 1
            r["alg_bytes"] += m["alg_bytes"]
            r["flops"] += m["flops"]
    return rows


def main():
    mode = sys.argv[1]
    if mode == "traffic":
        fetch_dir, write_dir, oplog, sha = sys.argv[2:6]
        log = json.load(open(oplog))
        ops, steps = log["ops"], log["steps_in_run"]
        rd, wr = dispatches(fetch_dir, {"FETCH_SIZE"}), dispatches(write_dir, {"WRITE_SIZE"})
        out = {"csrc_sha": sha, "dtype": log["dtype"],
               "formula": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes, per training step", "kernels": {}}
        agg = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
        for _, n, c in rd:
            a = agg[n.split("<")[0]]
            a[0] += c.get("FETCH_SIZE", 0.0); a[2] += 1
        for _, n, c in wr:
            a = agg[n.split("<")[0]]
            a[1] += c.get("WRITE_SIZE", 0.0); a[3] += 1
        tot = 0.0
        for k in sorted(agg):
            r, w, nr, nw = agg[k]
            b = (2 * r + w) * 1024 / steps
            tot += b
            out["kernels"][k] = {"read_GB": 2 * r * 1024 / steps / 1e9, "write_GB": w * 1024 / steps / 1e9,
                                 "GB_per_step": b / 1e9, "launches_per_step": max(nr, nw) / steps}
        out["total_GB_per_step"] = tot / 1e9
        conv = [v for k, v in out["kernels"].items() if k.startswith("conv_") or k.startswith("wgrad_reduce")]
        out["conv_GB_per_step"] = sum(v["GB_per_step"] for v in conv)
        out["conv_launches_per_step"] = sum(v["launches_per_step"] for v in conv)
        pr, pw = zip_with_ops(rd, ops, steps), zip_with_ops(wr, ops, steps)
        merged = []
        for a, b in zip(pr, pw):
            m = dict(a)
            m.update({k: v for k, v in b.items() if k != "_n"})
            merged.append(m)
        cls = per_class(merged, ops, ("FETCH_SIZE", "WRITE_SIZE"))
        table = []
        for (kind, lab), r in cls.items():
            pmc = (2 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024
            table.append({"kind": kind, "shape": lab, "launches": int(r["launches"]), "pmc_MB": round(pmc / 1e6, 1),
                          "alg_MB": round(r["alg_bytes"] / 1e6, 1), "pmc_over_alg": round(pmc / r["alg_bytes"], 3)})
        table.sort(key=lambda t: -t["pmc_MB"])
        out["classes"] = table
        print(json.dumps(out, indent=1))
    elif mode == "sq":
        sq_dir, oplog, stats_csv, sha = sys.argv[2:6]
        log = json.load(open(oplog))
        ops, steps = log["ops"], log["steps_in_run"]
        names = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                 "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS")
        disp = dispatches(sq_dir, set(names))
        # kernel durations of the stats run of the same command (total ns per kernel name over the run)
        dur = {}
        if os.path.exists(stats_csv):
            with open(stats_csv) as fh:
                for r in csv.DictReader(fh):
                    dur[short(r["Name"])] = (float(r["TotalDurationNs"]), int(r["Calls"]))

        def derived(c):
            busy, wave = c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_WAVE_CYCLES", 0.0)
            return {"mfma_busy_frac": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * busy), 4) if busy else None,
                    "wait_any_frac_of_wave_cycles": round(c.get("SQ_WAIT_ANY", 0.0) / wave, 4) if wave else None,
                    "wait_inst_any_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4) if wave else None,
                    "active_inst_any_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4) if wave else None,
                    "active_inst_valu_frac": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wave, 4) if wave else None,
                    "active_inst_lds_frac": round(c.get("SQ_ACTIVE_INST_LDS", 0.0) / wave, 4) if wave else None}

        tmpl = collections.defaultdict(lambda: collections.defaultdict(float))
        for _, n, c in disp:
            if not (n.startswith("conv_") or n.startswith("wgrad_") or n.startswith("bn_") or "head" in n or "dist" in n):
                continue
            t = tmpl[n]
            for k, v in c.items():
                t[k] += v / steps
            t["launches_per_step"] += 1.0 / steps
        out = {"csrc_sha": sha, "dtype": log["dtype"], "steps_in_run": steps,
               "normalisation": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES); *_frac = counter / SQ_WAVE_CYCLES "
                                "(quad-cycle units on both sides)", "kernel_templates": [], "classes": []}
        for n, t in sorted(tmpl.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0)):
            e = {"kernel": n, "launches_per_step": round(t["launches_per_step"], 2)}
            if n in dur:
                e["ms_per_step"] = round(dur[n][0] / 1e6 / steps, 4)
            e.update(derived(t))
            e["counters_per_step"] = {k: round(t[k]) for k in names if k in t}
            out["kernel_templates"].append(e)
        per_op = zip_with_ops(disp, ops, steps)
        cls = per_class(per_op, ops, names)
        for (kind, lab), r in sorted(cls.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0)):
            e = {"kind": kind, "shape": lab, "launches": int(r["launches"]), "gflop": round(r["flops"] / 1e9, 1)}
            e.update(derived(r))
            out["classes"].append(e)
        print(json.dumps(out, indent=1))
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
