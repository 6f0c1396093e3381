"""Prints the per-tensor distances of the trained-weights step check instead of asserting (debug aid)."""
import inspect
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_train_step as T  # noqa: E402

src = inspect.getsource(T._step_vs_f64_oracle)
src = src.replace("            if not e < (loose or {}).get(name, TOL):",
                  "            print('%-22s err %.3e  tol %.3e' % (name, e, (loose or {}).get(name, TOL)))\n"
                  "            if not e < (loose or {}).get(name, TOL):")
src = src.replace("    assert not bad, bad", "    print('BAD', bad)")
ns = dict(T.__dict__)
exec(src, ns)
ns["_step_vs_f64_oracle"](int(sys.argv[1]) if len(sys.argv) > 1 else 32, 8, 1, 28, pretrain_steps=200)
