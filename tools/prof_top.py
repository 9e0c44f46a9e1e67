"""Prints the top rows of a rocprofv3 kernel_stats.csv (dev helper): python tools/prof_top.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(r["Name"][:52].ljust(52), r["Calls"].rjust(5), f'{float(r["AverageNs"]) / 1e3:9.1f}', r["Percentage"].rjust(6), r["MinNs"].rjust(8), r["MaxNs"].rjust(8))
