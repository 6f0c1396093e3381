"""Per-kernel averages of rocprofv3 --pmc counters (csv output): python tools/pmc_summary.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[0]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("vt::", "")[:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
names = sorted({c for k in acc for c in acc[k]})
print("%-40s %5s " % ("kernel", "n") + " ".join("%14s" % c[-14:] for c in names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", acc[k].get(names[0], 0))):
    n = max(cnt[k].values())
    print("%-40s %5d " % (k, n) + " ".join("%14.4g" % (acc[k][c] / max(cnt[k][c], 1)) for c in names))
