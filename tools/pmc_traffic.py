"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; unit KB per dispatch):
python tools/pmc_traffic.py <fetch dir> <write dir> <out.json> <out.txt>.  FETCH_SIZE is doubled (gfx950 wide-read
correction, MI355X_MICROARCH.md HBM section)."""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, ids = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"]); ids[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (tot[k] * 1024.0 / len(ids[k]), len(ids[k])) for k in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline` "
                 "(tools/pmc_traffic.sh); FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md); bytes per launch, averaged over the kernel's launches",
       "kernels": {}}
rows = []
for k in sorted(fetch, key=lambda k: -(fetch[k][0] * fetch[k][1])):
    w = write.get(k, (0.0, 0))[0]
    out["kernels"][k] = {"fetch_bytes_corrected": 2 * fetch[k][0], "write_bytes": w, "calls": fetch[k][1]}
    rows.append("%-58s %5d %14.1f %14.1f %14.1f" % (k[:56], fetch[k][1], fetch[k][0] / 1e6, 2 * fetch[k][0] / 1e6, w / 1e6))
json.dump(out, open(sys.argv[3], "w"), indent=1)
open(sys.argv[4], "w").write("%-58s %5s %14s %14s %14s\n" % ("kernel", "calls", "FETCH avg MB", "FETCHx2 avg MB", "WRITE avg MB") + "\n".join(rows[:24]) + "\n")
print("\n".join(rows[:12]))
