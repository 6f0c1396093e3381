"""Summarises a rocprofv3 --kernel-trace --stats output directory (csv or rocpd .db): per-kernel
calls / average duration / share, plus launch geometry and register use when the .db holds them.

    python tools/prof_summary.py <dir> [steps]
"""
import csv
import glob
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("vt::", "")[:34]


def from_csv(path):
    rows = list(csv.DictReader(open(path)))
    return [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, "") for r in rows]


def from_db(path):
    c = sqlite3.connect(path)
    q = ("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(grid_x), max(grid_x), max(workgroup_x), "
         "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
         "from kernels group by name order by sum(duration) desc")
    out = []
    for n, calls, tot, avg, g0, g1, wg, vg, ag, sg, lds, scr in c.execute(q):
        geo = "grid %s wg %s vgpr %s+%s sgpr %s lds %s scratch %s" % (g1 if g0 == g1 else "%s..%s" % (g0, g1), wg, vg, ag, sg, lds, scr)
        out.append((n, calls, tot, avg, geo))
    return out


def main():
    d = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dbs = sorted(glob.glob(d + "/**/*.db", recursive=True))
    csvs = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))
    rows = from_db(dbs[0]) if dbs else from_csv(csvs[0])
    tot = sum(r[2] for r in rows)
    print("%-34s %6s %10s %7s  %s" % ("kernel", "calls", "avg_us", "share", "launch"))
    for n, calls, t, avg, geo in rows:
        print("%-34s %6d %10.2f %6.1f%%  %s" % (short(n), calls, avg, 100 * t / tot, geo))
    # kernels launched fewer times than there are steps (set-up fills, the copy yardstick) are not part of a step
    step_tot = sum(r[2] for r in rows if r[1] >= steps)
    print("sum of kernel time per step: %.1f us (%d steps; kernels with at least one launch per step)" % (step_tot / steps, steps))


if __name__ == "__main__":
    main()
