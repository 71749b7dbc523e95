"""GPU box: reads a rocprofv3 --kernel-trace csv and reports, for the timed steps, how much of the wall time the
device had at least one kernel running (union of intervals), and the idle-gap histogram on the busiest queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows))
# window = the last NSTEP train steps; a step ends with the fused SGD kernel
NSTEP = 4
sgd = [e for e in ev if "sgd_kernel" in e[2]]
lo, hi = sgd[-NSTEP - 1][1], sgd[-1][1]
ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
print("window: %d steps, %.2f ms per step" % (NSTEP, (hi - lo) / 1e6 / NSTEP))
wall = max(e[1] for e in ev) - ev[0][0]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
gaps = []
for s, e, n, q in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("kernels %d wall %.2f ms busy(union) %.2f ms idle %.2f ms (%.1f %%)" % (len(ev), wall / 1e6, busy / 1e6, (wall - busy) / 1e6, 100.0 * (wall - busy) / wall))
h = collections.Counter()
for g, n in gaps:
    b = 1 if g < 1000 else 2 if g < 2000 else 5 if g < 5000 else 10 if g < 10000 else 50 if g < 50000 else 1000
    h[b] += g
print("idle by gap size (us bucket upper bound -> total ms):", {k: round(v / 1e6, 3) for k, v in sorted(h.items())}, "n gaps", len(gaps))
byk = collections.Counter()
for g, n in gaps:
    byk[n[:60]] += g
print("idle before kernel (top):")
for k, v in byk.most_common(12):
    print("  %-62s %.3f ms" % (k, v / 1e6))
per_q = collections.Counter()
for s, e, n, q in ev:
    per_q[q] += e - s
print("busy per queue (ms):", {k: round(v / 1e6, 2) for k, v in per_q.items()})
