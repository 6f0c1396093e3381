"""Turns the outputs of tools/profile_shapes.sh (gpurun_out/shapes/) into the committed summaries under profiles/:
kernel traces, bench lines, per-kernel traffic tables and the whole-step traffic totals.

    python tools/shape_profiles_to_repo.py <tag> [shape ...]     e.g. r02_k, or r03_e wide
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "shapes")
CFG = {"scaled": 1024, "wide": 16}


def step_rows(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
    rs = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    ids = [i for i, r in enumerate(rs) if "k_sgd" in r["Kernel_Name"]]
    return rs[ids[0] + 1:ids[-1] + 1]  # the launches of the measured step (after the warm-up step's update)


def main():
    tag = sys.argv[1]
    only = sys.argv[2:]  # shapes to take (default: both)
    totals = {"unit": "bytes per training step (every launch of one step)",
              "formula": "sum over the step's launches of 1024 * (2 * FETCH_SIZE + WRITE_SIZE), separate rocprofv3 --pmc passes "
                         "of bench.py --steps 1 --warmup 1 (tools/profile_shapes.sh)", "steps": {}}
    for sh, b in CFG.items():
        if only and sh not in only:
            continue
        line = [l for l in open("%s/%s_bench.json" % (SRC, sh)) if l.startswith("{")][-1]
        d = json.loads(line)
        open("%s/profiles/%s_%s_bench_line.json" % (ROOT, tag, sh), "w").write(line)
        out = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --shape %s --batch %d --steps 3 --warmup 1 --no-cpu-baseline"
               "   (MI355X; general-shape step)" % (sh, b),
               "# bench line of the same tree without the profiler: %.3f ms/step, %.0f images/s, whole-step HBM roofline fraction %.3f"
               % (d["ms_per_step"], d["value"], d["roofline"]["frac"])]
        out += [l[:190] for l in subprocess.check_output([sys.executable, ROOT + "/tools/prof_summary.py", "%s/%s_trace" % (SRC, sh), "4"],
                                                         text=True).splitlines()]
        out += ["", "# kernels of one step in launch order (>= 20 us)"]
        db = glob.glob("%s/%s_trace/**/*.db" % (SRC, sh), recursive=True)[0]
        out += subprocess.check_output([sys.executable, ROOT + "/tools/step_trace.py", db, "20"], text=True).splitlines()
        open("%s/profiles/%s_%s_shape_b%d_kernel_trace.txt" % (ROOT, tag, sh, b), "w").write("\n".join(out) + "\n")
        subprocess.check_output([sys.executable, ROOT + "/tools/pmc_traffic.py", "%s/%s_fetch" % (SRC, sh), "%s/%s_write" % (SRC, sh),
                                 "%s/profiles/%s_hbm_traffic_%s_b%d.json" % (ROOT, tag, sh, b),
                                 "bench.py --shape %s --batch %d --steps 1 --warmup 1 (general-shape step): average per launch over the "
                                 "kernels of that name" % (sh, b)])
        F, W = step_rows("%s/%s_fetch" % (SRC, sh), "FETCH_SIZE"), step_rows("%s/%s_write" % (SRC, sh), "WRITE_SIZE")
        fb = sum(2 * 1024 * float(r["Counter_Value"]) for r in F)
        wb = sum(1024 * float(r["Counter_Value"]) for r in W)
        totals["steps"]["%s_b%d_k8" % (sh, b)] = {"traffic_bytes": int(fb + wb), "fetch_bytes": int(fb), "write_bytes": int(wb),
                                                  "launches": len(F)}
        print(sh, "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], "traffic %.1f GB (read %.1f, write %.1f)" % ((fb + wb) / 1e9, fb / 1e9, wb / 1e9))
    json.dump(totals, open("%s/profiles/%s_step_hbm_traffic_shapes.json" % (ROOT, tag), "w"), indent=1)


if __name__ == "__main__":
    main()
