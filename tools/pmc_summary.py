"""Summarise rocprofv3 --pmc csv output per kernel (mean per dispatch)."""
import csv, glob, os, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, d in acc.items():
    if "qmpc" not in k: continue
    res[k] = {c: sum(v) / len(v) for c, v in d.items()}
    res[k]["dispatches"] = max(len(v) for v in d.values())
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:28s} {v:16.1f}")
