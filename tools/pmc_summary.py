"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: python tools/pmc_summary.py <dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    dur[(k, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in agg.items():
    n = len(cnt[k]); d = [x for (kk, _), x in dur.items() if kk == k]
    print("%-50s n=%d avg %.1f us | " % (k, n, sum(d) / len(d)) + " ".join("%s=%d" % (a, round(b / n)) for a, b in sorted(v.items())))
