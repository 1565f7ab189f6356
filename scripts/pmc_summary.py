"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "exa::" not in k:
            continue
        short = k.split("(")[0].replace("void ", "")
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k in sorted(agg):
    lines.append(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        lines.append("   %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
txt = "\n".join(lines)
print(txt)
open(os.path.join(root, "summary.txt"), "w").write(txt + "\n")
