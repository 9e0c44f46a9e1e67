"""One replayed step of a rocprofv3 kernel trace (dev helper): wall time, sum of kernel time, time with >= 1 kernel running,
and the kernels in start order.  python tools/step_timeline.py <kernel_trace.csv> [anchor-kernel-substring]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "clip_adam_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
best = None
for a, b in zip(idx, idx[1:]):                                  # the shortest anchor-to-anchor interval: a replayed step
    w = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    if b - a >= 10 and (best is None or w < best[0]): best = (w, a, b)        # at least 10 kernels: not a back-to-back timing loop
w, a, b = best
seg = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
ivs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cur_s, cur_e = 0, None, None
for s, e in ivs:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"step wall {w / 1e3:.1f} us, {len(seg)} kernels, sum of kernel time {sum(e - s for s, e in ivs) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
prev = t0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  gap {(s - prev) / 1e3:6.1f}  q{r['Queue_Id']}  {r['Kernel_Name'][:70]}")
    prev = max(prev, e)
