"""Per-kernel summary of a rocprofv3 run (its results .db, or a directory holding one): calls, median / mean duration, total.
   python tools/prof_db.py gpurun_out/prof [--timeline N]   (--timeline: N consecutive kernels from the middle of the run)"""
import glob, os, sqlite3, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(path)
rows = c.execute("select name, start, end, queue_id from kernels order by start").fetchall()
by = {}
for n, s, e, q in rows:
    by.setdefault(n, []).append((e - s) / 1e3)
tot = sum(sum(v) for v in by.values())
print(f"{'kernel':72s} {'calls':>6s} {'med_us':>8s} {'mean_us':>8s} {'tot_ms':>8s} {'pct':>5s}")
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:28]:
    v2 = sorted(v)
    print(f"{n[:72]:72s} {len(v):6d} {v2[len(v)//2]:8.1f} {sum(v)/len(v):8.1f} {sum(v)/1e3:8.2f} {100*sum(v)/tot:5.1f}")
if "--timeline" in sys.argv:
    k = int(sys.argv[sys.argv.index("--timeline") + 1])
    i = len(rows) // 2 if "--from" not in sys.argv else int(float(sys.argv[sys.argv.index("--from") + 1]) * len(rows))
    while i < len(rows) and not rows[i][0].startswith("__amd_rocclr_copyBuffer"):
        i += 1
    t0 = rows[i][1]
    for n, s, e, q in rows[i:i + k]:
        print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} dur={(e - s) / 1e3:6.1f} q={q} {n[:70]}")
