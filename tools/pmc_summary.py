"""Summarise rocprofv3 --pmc passes for one kernel: python tools/pmc_summary.py <kernel substring> <out.json> <dir:FETCH> <dir:WRITE> [<dir:other>...]"""
import collections, csv, glob, json, sys
name, out = sys.argv[1], sys.argv[2]
res = {"kernel_substring": name}
for d in sys.argv[3:]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(f)):
        if name in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in acc.items():
        res[k] = sum(v) / len(v)
    if dur:
        res.setdefault("kernel_us", {})[d.rstrip("/").split("/")[-1]] = sum(dur) / len(dur)
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    res["hbm_bytes_per_launch_corrected"] = (2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024
    res["correction"] = "gfx950: FETCH_SIZE (KB) counts 1/2 of wide reads -> doubled; WRITE_SIZE (KB) exact, float atomics included"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
