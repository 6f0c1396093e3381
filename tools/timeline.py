"""Timeline of the last complete step in a rocprofv3 --kernel-trace output directory (rocpd .db):
start offset, duration and queue of every kernel between two consecutive k_conv_fwd launches, so
that overlap between the two backward chains (and idle gaps) can be read off.

    python tools/timeline.py <dir> [step_from_end]
"""
import glob
import sqlite3
import sys


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    db = sorted(glob.glob(d + "/**/*.db", recursive=True))[0]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = "name, start, end" + (", " + qcol if qcol else ", 0")
    rows = list(c.execute("select %s from kernels order by start" % sel))
    firsts = [i for i, r in enumerate(rows) if "k_conv_fwd" in r[0]]
    if len(firsts) < back + 1:
        print("columns:", cols)
        print("only %d steps in the trace" % len(firsts))
        return
    a, b = firsts[-back - 1], firsts[-back]
    t0 = rows[a][1]
    busy_end = t0
    idle = 0
    for name, st, en, q in rows[a:b]:
        gap = st - busy_end
        if gap > 0:
            idle += gap
        busy_end = max(busy_end, en)
        print("%8.1f +%6.1f  q%-3s %s%s" % ((st - t0) / 1e3, (en - st) / 1e3, q, name.split("(")[0].replace("vt::", "")[:40],
                                           "   [idle %.1f us before]" % (gap / 1e3) if gap > 500 else ""))
    print("step span %.1f us, device idle inside it %.1f us, sum of kernel durations %.1f us" % (
        (rows[b][1] - t0) / 1e3, idle / 1e3, sum(r[2] - r[1] for r in rows[a:b]) / 1e3))


if __name__ == "__main__":
    main()
