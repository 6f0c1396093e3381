"""Prints the kernels of the last step in a rocprofv3 --kernel-trace database in launch order (those above a duration
floor), with launch geometry: python tools/step_trace.py <results.db> [min_us]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    floor = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
    rows = list(c.execute("select name,start,duration,grid_x,workgroup_x,vgpr_count,lds_size from kernels order by start"))
    ends = [i for i, r in enumerate(rows) if "k_sgd" in r[0] or "k_reduce_multi_sgd" in r[0]]
    s, e = ends[-2] + 1, ends[-1]
    t0 = rows[s][1]
    tot = 0.0
    for r in rows[s:e + 1]:
        tot += r[2] / 1e3
        if r[2] / 1e3 < floor:
            continue
        n = r[0].replace("vt::", "").replace("void ", "").split("(")[0][:48]
        print("%9.1f %9.1f  %-48s grid %d vgpr %s lds %s" % ((r[1] - t0) / 1e3, r[2] / 1e3, n, r[3] // r[4], r[5], r[6]))
    print("sum of kernel durations of the step: %.1f us" % tot)


if __name__ == "__main__":
    main()
