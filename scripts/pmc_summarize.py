"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and counter (the per-dispatch CSV of a run
that builds an index is too large to carry around): python scripts/pmc_summarize.py <dir> <out.csv>"""
import collections, csv, glob, os, sys
d, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
dur = collections.defaultdict(float)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        calls[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
with open(out, "w") as w:
    w.write("kernel,calls,total_ms,counter,sum_over_dispatches\n")
    for k, v in agg.items():
        for c, x in v.items():
            w.write(f'"{k}",{calls[k]},{dur[k]:.3f},{c},{x:.0f}\n')
print(open(out).read())
