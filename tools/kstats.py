"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (a kernel_stats.csv, or a directory holding one)."""
import csv, glob, os, sys
f = sys.argv[1] if os.path.isfile(sys.argv[1]) else glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} pct={r['Percentage']}")
