"""Every kernel of the last complete general-shape step in a rocprofv3 --kernel-trace output directory
(rocpd .db), in launch order, with its grid (in workgroups) and duration: which product or streaming pass
the time of a wide configuration goes to.  A step is delimited by two consecutive launches of the marker
kernel (default k_im2col, the first kernel of the general-shape forward pass).

    python tools/step_calls.py <dir> [marker] [min_us]
"""
import glob
import sqlite3
import sys


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_im2col"
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    db = sorted(glob.glob(d + "/**/*.db", recursive=True))[0]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    want = ["grid_x", "grid_y", "grid_z", "workgroup_x"]
    alt = {"grid_x": "grid_size_x", "grid_y": "grid_size_y", "grid_z": "grid_size_z", "workgroup_x": "workgroup_size_x"}
    pick = [w if w in cols else alt[w] for w in want]
    if any(p not in cols for p in pick):
        print("columns:", cols)
        return
    rows = list(c.execute("select name, start, end, %s from kernels order by start" % ", ".join(pick)))
    firsts = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(firsts) < 2:
        print("fewer than two %s launches" % marker)
        return
    a, b = firsts[-2], firsts[-1]
    t0 = rows[a][1]
    for name, st, en, gx, gy, gz, wx in rows[a:b]:
        us = (en - st) / 1e3
        if us < min_us:
            continue
        short = name.replace("vt::", "").replace("void ", "")
        short = short.split("(")[0][:44]
        print("%9.1f +%8.1f  wg %6d x %4d x %5d  %s" % ((st - t0) / 1e3, us, gx // max(wx, 1), gy, gz, short))
    print("step span %.1f us, sum of kernel durations %.1f us" % ((rows[b][1] - t0) / 1e3, sum(r[2] - r[1] for r in rows[a:b]) / 1e3))


if __name__ == "__main__":
    main()
