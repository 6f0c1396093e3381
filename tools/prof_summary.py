"""Summarises a rocprofv3 --kernel-trace --stats csv directory: per-kernel calls / average / share."""
import csv, glob, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-30s %6s %11s %7s" % ("kernel", "calls", "avg_us", "share"))
for r in rows:
    name = r["Name"].split("(")[0].replace("vt::", "")
    print("%-30s %6s %11.2f %6.1f%%" % (name[:30], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("sum of kernel time per step: %.1f us (%d steps)" % (tot / steps / 1e3, steps))
