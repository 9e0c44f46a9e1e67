"""Dev: mean value of every counter in a rocprofv3 --pmc CSV directory for the kernels whose name contains a substring.
python tools/pmc_counters.py <substring> <dir> [<dir> ...]"""
import csv, glob, os, sys
sub, dirs = sys.argv[1], sys.argv[2:]
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                a = acc.setdefault(r["Counter_Name"], [0.0, 0])
                a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (s, n) in sorted(acc.items()):
            print(f"{k:32s} mean {s / n:16.1f}  over {n} dispatches")
