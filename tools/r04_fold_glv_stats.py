"""Prints the fold kernels' rows of the two kernel-stats files tools/r04_fold_glv_ab.sh leaves behind."""
import csv
import glob
import sys

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04f"
for m in (1, 2):
    for f in glob.glob(f"{out}/fold{m}/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "multifold" in r["Name"] or "odd_multiples<16" in r["Name"]:
                print(f"fold_wnaf={m}  {r['Name'][:34]:34s} calls {r['Calls']:>3s}  avg {float(r['AverageNs']) / 1e6:.3f} ms")
