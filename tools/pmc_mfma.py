"""MFMA pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass:
python tools/pmc_mfma.py <dir> [out.txt].  SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of the 1024 matrix pipes (32 per
v_mfma_f32_16x16x4_f32), GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs: utilisation = busy / (1024 * active / 8)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); ids = collections.defaultdict(set); dur = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); ids[k].add(r["Dispatch_Id"])
    dur[(k, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = []
for k, v in agg.items():
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    n = len(ids[k]); d = sum(x for (kk, _), x in dur.items() if kk == k) / n
    util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * v["GRBM_GUI_ACTIVE"] / 8.0)
    rows.append((v["SQ_VALU_MFMA_BUSY_CYCLES"], "%-62s %4d launches  avg %8.1f us (under PMC)  MFMA pipes busy %5.1f %%" % (k[:60], n, d, 100 * util)))
out = "\n".join(t for _, t in sorted(rows, reverse=True))
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline\n"
                                 "(tools/pmc_mfma.sh; utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 pipes x GRBM_GUI_ACTIVE / 8 XCDs); fp32 MFMA 16x16x4 = 32 busy cycles)\n\n" + out + "\n")
