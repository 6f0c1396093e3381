#!/usr/bin/env python3
"""VTCNN training throughput on MI355X: BloodMNIST-shaped synthetic data, data-parallel.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one reference training iteration (utils/trainer.hpp:39-54) per GPU: forward, backward,
gradient all-reduce (N > 1), SGD update, on one resident mini-batch of `--batch` 1x28x28 images,
8 classes.  Weak scaling: every rank trains its own mini-batches (one reference mini-batch is a
semantic unit because of BatchNorm), gradients are summed over ranks with one RCCL all-reduce of
the flat gradient arena and divided by world*batch.  Rank 0 prints ONE JSON line: the contract's fields, `roofline`
(whole-step HBM form: algorithmic layer-boundary bytes per step over the measured step time, with the per-launcher
table and the event-timed dominant launcher beside it) and `cpu_baseline` (the real reference on the host cores).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process never touches a GPU; it starts N ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` on a free port of 127.0.0.1), relays rank 0's
JSON line and exits with the launcher's status.  `n_gpus` in the line is the number of ranks that joined the
process group (an all-reduce of ones), not the flag.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT_ = os.path.dirname(os.path.abspath(__file__))


def launch_ranks(n, argv):
    """Parent of a multi-rank run: starts one process per GPU before anything here has touched a device, passes
    the children's stderr through, prints the one JSON line rank 0 wrote and returns the launcher's exit status."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT_, "bench.py")] + list(argv)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")  # what torch.distributed.run would set itself, without its warning
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = []
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            lines.append(line.rstrip("\n"))
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        rc = 1
    return rc


def _launch_if_parent():
    """`--gpus N > 1` outside a process group: become the parent (before numpy / torch are even imported)."""
    if "WORLD_SIZE" in os.environ:
        return
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    known, _ = ap.parse_known_args()
    if known.gpus > 1:
        sys.exit(launch_ranks(known.gpus, sys.argv[1:]))


if __name__ == "__main__":
    _launch_if_parent()

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, spec
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (v_mfma_f32_32x32x2_f32), spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16), spec
# the attention backward forms every fp32 product from bf16 hi / lo halves: 19 matrix instructions of
# 32x32x16 per 32 x 32 tile pair where the fp32 formulation counts 5 contractions of 32x32x16
MHA_EXECUTED_PER_ALGORITHMIC = 19.0 / 5.0


def attention_bwd_flops(d, batch):
    """Algorithmic FLOPs of one attention-backward launch in the reference's own formulation
    (layers/multi_head_attention.hpp:155-169): per head and group five T x T x head_dim
    contractions (scores, delta_attn, delta_value, delta_key, delta_query), T = mini-batch."""
    hd = d.token_channels // d.heads
    return 5 * 2 * batch * batch * hd * d.heads * d.tokens


def algorithmic_bytes_per_image(d, forward_only=False, generic=False):
    """Layer-boundary fp32 traffic per image apportioned to the launcher that performs it.
    Each tensor that crosses a top-level reference Layer interface counts once per write and once
    per read; tensors internal to a composite layer and parameters do not count.  The sum over
    launchers is 4*[2*Ci*P + 14*C1*P + 5*C2*P + 4*Q + 11*L*D + 5*K] bytes per trained image."""
    Ci, P, C2, L, D, K = d.image_channels, d.input_dim ** 2, d.vt_channels, d.tokens, d.token_channels, d.num_classes
    C1, Q = C2 // 2, C2 * 196
    t = {
        "conv3x3_fwd": Ci * P + C1 * P,
        "tokenizer_fwd": 3 * C1 * P + L * D,
        "transformer_ffn_fwd": 2 * L * D,
        "projector_tokens_fwd": L * D,
        "projector_fwd": C1 * P + C2 * P,
        "head_fwd": C2 * P + 2 * Q + 3 * K,
        "head_bwd": 2 * Q + C2 * P + 2 * K,
        "projector_bwd": 2 * C1 * P + 2 * C2 * P + 2 * L * D,
        "transformer_bwd_b": 4 * L * D,
        "tokenizer_bwd": 4 * C1 * P + L * D,
        "conv3x3_wgrad": 3 * C1 * P + Ci * P,
    }
    fwd_total = 4 * (Ci * P + 5 * C1 * P + 2 * C2 * P + 2 * Q + 4 * L * D + 3 * K)
    if generic:  # the general-shape step is profiled as two launchers: the whole forward, the whole backward
        total = 4 * (2 * Ci * P + 14 * C1 * P + 5 * C2 * P + 4 * Q + 11 * L * D + 5 * K)
        return {"generic_forward": fwd_total, "generic_backward": total - fwd_total}, (fwd_total if forward_only else total)
    t = {k: 4 * v for k, v in t.items()}
    total = 4 * (2 * Ci * P + 14 * C1 * P + 5 * C2 * P + 4 * Q + 11 * L * D + 5 * K)
    assert sum(t.values()) == total, (sum(t.values()), total)
    assert sum(v for k, v in t.items() if k.endswith("_fwd")) == fwd_total
    return t, (fwd_total if forward_only else total)


SHAPES = {  # hyper-parameters besides the class count
    "blood": dict(image_channels=1, input_dim=28, tokens=16, token_channels=32, vt_channels=16, heads=2),
    "scaled": dict(image_channels=3, input_dim=224, tokens=16, token_channels=64, vt_channels=128, heads=8),
    "wide": dict(image_channels=3, input_dim=224, tokens=196, token_channels=512, vt_channels=128, heads=8),
}

LAUNCHER_KERNEL = {  # launcher name (vt_model_profile) -> device kernel it launches
    "conv3x3_fwd": "k_conv_fwd", "tokenizer_fwd": "k_tokenizer_fwd", "transformer_ffn_fwd": "k_tr_ffn_fwd",
    "projector_tokens_fwd": "k_proj_tokens_fwd", "projector_fwd": "k_proj_fwd", "head_fwd": "k_head_fwd",
    "head_bwd": "k_head_bwd_win", "projector_bwd": "k_proj_bwd", "transformer_bwd_b": "k_tr_bwd_b",
    "tokenizer_bwd": "k_tokenizer_bwd", "conv3x3_wgrad": "k_conv_wgrad", "mha_bwd_rows": "k_mha_rows",
}


def measured_traffic(launcher, batch, classes):
    """HBM bytes per launch of the launcher's kernel from the committed PMC passes
    (profiles/r02_hbm_traffic_b1024_k8.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    runs of this same command, 2*FETCH_SIZE + WRITE_SIZE as the gfx950 notes prescribe).  Only
    valid for the configuration it was measured on; None otherwise."""
    t = measured_traffic_table(batch, classes)
    k = LAUNCHER_KERNEL.get(launcher)
    return t.get(k) if (t and k) else None


TRAFFIC_FILES = ("r02_hbm_traffic_b1024_k8.json", "r01_hbm_traffic_b1024_k8.json")  # newest first
STEP_LAUNCHES = {"k_bn_stats_finalize": 4, "k_bn_bwd_finalize": 4}  # launches per step where it is not one


def measured_step_traffic(shape, batch, classes):
    """HBM bytes of one whole general-shape training step from the committed PMC passes; None off the measured
    configurations."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r03_h_step_hbm_traffic_shapes.json")))
        return int(t["steps"]["%s_b%d_k%d" % (shape, batch, classes)]["traffic_bytes"])
    except (OSError, KeyError, ValueError):
        return None


def measured_traffic_table(batch, classes):
    """kernel -> HBM bytes per launch from the committed PMC passes; None off the measured configuration."""
    if batch != 1024 or classes != 8:
        return None
    for fn in TRAFFIC_FILES:
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", fn)))
            return {k: int(v["traffic_bytes"]) for k, v in t["kernels"].items()}
        except (OSError, KeyError, ValueError):
            continue
    return None


def cpu_baseline(batch, classes):
    """Reference CPU path timed on this box's host cores (rank 0, N=1 only): the real reference
    compiled from its own headers (oracle/_ref/ref_harness_o3, built in the build container) when
    present, else the oracle port.  Bounded sample: five training steps of one mini-batch after one
    warm-up step (about 25 s at 1024 images on 16 threads).  `cores` is the number of threads used (the
    box's CPU share for one GPU is 16 cores); `host_cpu_count` is what the machine reports."""
    host_cpus = os.cpu_count() or 1
    cores = min(16, host_cpus)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness_o3")
    if os.access(exe, os.X_OK):
        try:
            env = dict(os.environ, OMP_NUM_THREADS=str(cores))
            out = subprocess.run([exe, "bench", str(batch), str(classes), "5", "1"], env=env, capture_output=True,
                                 text=True, timeout=300)
            r = json.loads(out.stdout.strip().splitlines()[-1])
            res = {"value": round(r["images_per_sec"], 2), "unit": "images/sec", "cores": int(r["threads"]),
                   "host_cpu_count": host_cpus, "kind": "reference",
                   "sample": "5 training steps (after 1 warm-up) of one %d-image mini-batch, reference headers "
                             "compiled -O3 -march=x86-64-v3 -fopenmp, OMP_NUM_THREADS=%d" % (batch, cores)}
            # the same binary on one core, at the reference's own batch size (a 1024-image step would take a minute)
            one = subprocess.run([exe, "bench", "32", str(classes), "3", "0"], env=dict(os.environ, OMP_NUM_THREADS="1"),
                                 capture_output=True, text=True, timeout=300)
            r1 = json.loads(one.stdout.strip().splitlines()[-1])
            res["one_core"] = {"value": round(r1["images_per_sec"], 2), "unit": "images/sec", "cores": 1,
                               "sample": "3 training steps of one 32-image mini-batch"}
            # and what `run.sh main <dataset> train` itself runs: the reference's loader batch of 32 (main.cpp:77) on all cores
            b32 = subprocess.run([exe, "bench", "32", str(classes), "20", "2"], env=env, capture_output=True, text=True, timeout=300)
            r32 = json.loads(b32.stdout.strip().splitlines()[-1])
            res["batch32_all_cores"] = {"value": round(r32["images_per_sec"], 2), "unit": "images/sec", "cores": int(r32["threads"]),
                                        "sample": "20 training steps (after 2 warm-up) of one 32-image mini-batch, the reference's own "
                                                  "loader batch size"}
            return res
        except Exception as e:  # fall through to the port
            sys.stderr.write("reference harness failed (%s); timing the oracle port instead\n" % e)
    from oracle.pyoracle import Oracle, make_dims
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    n = min(batch, 256)
    rng = np.random.default_rng(1)
    x = (rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0))
    y = rng.integers(0, classes, size=n).astype(np.int32)
    o = Oracle(make_dims(num_classes=classes), n)
    arena = np.zeros(o.layout.n_total, dtype=np.float32)
    arena[:] = 0.01
    t0 = time.time()
    o.forward(arena, x, True)
    g = o.backward(arena, y)
    o.sgd(arena, g)
    dt = time.time() - t0
    return {"value": round(n / dt, 2), "unit": "images/sec", "cores": cores, "host_cpu_count": host_cpus, "kind": "port",
            "sample": "1 training step of one %d-image mini-batch, CPU restatement (oracle/vtcnn_oracle.c)" % n}


def algorithmic_flops_per_image(d, batch, forward_only=False):
    """Matrix-product and convolution FLOPs of one image in the reference's own formulation: forward
    2*[conv + tokenizer + transformer (q/k/v, scores and attention.values over the mini-batch axis, output projection,
    feed-forward) + projector + fc], training = 3 x forward (the reference's recomputations and its unused
    convolution input delta are not counted)."""
    Ci, P, C2, L, D, K = d.image_channels, d.input_dim ** 2, d.vt_channels, d.tokens, d.token_channels, d.num_classes
    C1, Q = C2 // 2, C2 * 196
    fwd = 2 * (9 * Ci * C1 * P + 2 * P * C1 * L + L * C1 * D + 3 * L * D * D + 2 * L * batch * D + L * D * D + 2 * L * D * D
               + P * C1 * D + L * D * D + P * D * L + L * D * C2 + P * L * C2 + P * C1 * C2 + Q * K)
    return fwd if forward_only else 3 * fwd


GENERIC_EMULATION = ("fp32 storage and accumulation; products of depth >= 32 that do not feed an exponential (activation x "
                     "weight, delta x weight, gradient contractions) run as bf16x3 hi/lo split products with fp32 accumulate "
                     "(about 2^-17 relative per product), the others on exact fp32 MFMA; VT_GEMM_EXACT=1 keeps every product exact")
BF16_EMULATION = ("--bf16: the projector's value projection, the classifier and the backward's recomputed q/k/v projections round "
                  "their operands to bf16 (fp32 accumulate); products upstream of an exponential (feed-forward block, keys, "
                  "scores) and everything else as in the fp32 mode of the general step (VT_BF16_PRODUCTS=31 adds them, "
                  "outside the 1e-2 bar)")


def emulation_note(generic, bf16):
    if not generic:
        return ("fp32 storage and accumulation; the 16- and 32-deep contractions "
                "(projector value/downsample and delta products, tokenizer input delta, all attention-backward products) "
                "run as bf16x3 hi/lo split products with fp32 accumulate (about 2^-17 relative per product), "
                "the narrow ones on exact fp32 MFMA")
    if os.environ.get("VT_GEMM_EXACT", "") not in ("", "0"):
        return BF16_EMULATION if bf16 else None
    return (BF16_EMULATION + "; " + GENERIC_EMULATION) if bf16 else GENERIC_EMULATION


def time_extra_config(shape, batch, bf16, classes, steps=3, warmup=1):
    """One of the other BASELINE configurations under the same clock, in this process (after the headline model is
    gone): `steps` training steps of the general-shape step on one resident synthetic mini-batch.  The scaled shape
    is HBM-bound (layer-boundary byte model), the wide shape matrix-bound (algorithmic FLOPs against the dense
    matrix peak of the arithmetic it runs in: exact-fp32 MFMA 157.3 TF, bf16 2.5 PF)."""
    from vit_amd import VTCNN, Dims
    dims = Dims(num_classes=classes, **SHAPES[shape])
    m = VTCNN(dims, max_batch=batch, seed=42)
    if bf16:
        m.set_matmul_precision(True)
    g = torch.Generator(device="cuda")
    g.manual_seed(123)
    x = torch.rand((batch, dims.image_channels, dims.input_dim, dims.input_dim), device="cuda", generator=g)
    y = torch.randint(0, classes, (batch,), device="cuda", generator=g).to(torch.int32)
    for _ in range(warmup):
        m.train_step(x, y, float(batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train_step(x, y, float(batch))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    finite = bool(torch.isfinite(m.arena).all().item())
    _, total_bytes = algorithmic_bytes_per_image(dims, False, True)
    flops = algorithmic_flops_per_image(dims, batch)
    gbs = total_bytes * batch / dt / 1e9
    tfs = flops * batch / dt / 1e12
    if shape == "wide":
        peak = MFMA_BF16_PEAK_TFLOPS if bf16 else MFMA_F32_PEAK_TFLOPS
        roof = {"bound": "mfma", "achieved": round(tfs, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tfs / peak, 4),
                "algorithmic_flops_per_step": flops * batch, "hbm_gbs_on_the_byte_model": round(gbs, 1)}
    else:
        roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_step": total_bytes * batch, "tflops_on_the_flop_model": round(tfs, 2)}
    roof["traffic"] = measured_step_traffic(shape, batch, classes) if not bf16 else None
    roof["traffic_source"] = "committed PMC passes of this configuration (profiles/), not this run" if roof["traffic"] else None
    del m, x, y
    torch.cuda.empty_cache()
    return {"ms_per_step": round(dt * 1e3, 3), "images_per_sec": round(batch / dt, 1), "batch": batch, "steps": steps,
            "warmup": warmup, "dtype": "bf16" if bf16 else "f32", "emulation": emulation_note(True, bf16),
            "weights_finite_after_run": finite, "roofline": roof}


def time_blood_general_path(batch, classes, exact, steps=30, warmup=5):
    """The headline shape on the general-shape launchers (VT_GENERAL_PATH=1 at create), with the deep contractions as
    bf16 hi/lo split products (exact False: the arithmetic of the tuned step) or with every product on exact fp32 matrix
    instructions (exact True): what exactness costs on this shape, measured on the one set of kernels that has both."""
    from vit_amd import VTCNN, Dims
    from vit_amd._lib import lib as L
    dims = Dims(num_classes=classes, **SHAPES["blood"])
    os.environ["VT_GENERAL_PATH"] = "1"
    before = L.vt_set_exact_products(1 if exact else 0)
    try:
        m = VTCNN(dims, max_batch=batch, seed=42)
    finally:
        os.environ.pop("VT_GENERAL_PATH", None)
    try:
        g = torch.Generator(device="cuda")
        g.manual_seed(321)
        x = (torch.randint(0, 256, (batch, 1, 28, 28), device="cuda", generator=g).to(torch.float32) / 255.0).contiguous()
        y = torch.randint(0, classes, (batch,), device="cuda", generator=g).to(torch.int32)
        for _ in range(warmup):
            m.train_step(x, y, float(batch))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.train_step(x, y, float(batch))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        finite = bool(torch.isfinite(m.arena).all().item())
    finally:
        L.vt_set_exact_products(before)
    del m, x, y
    torch.cuda.empty_cache()
    return {"ms_per_step": round(dt * 1e3, 4), "images_per_sec": round(batch / dt, 1), "batch": batch, "steps": steps,
            "warmup": warmup, "dtype": "f32", "kernels": "general-shape launchers (not the tuned step)",
            "emulation": None if exact else GENERIC_EMULATION, "weights_finite_after_run": finite}


def time_blood_config(batch, group, classes, steps=100, warmup=20):
    """The headline shape at another batching, under the same clock: `group` > 0 trains batch / group independent
    reference-sized mini-batches side by side with one summed update (vt_model_train_step_grouped: the semantics of
    that many data-parallel ranks of the reference's own loader batch on one device), 0 the plain step."""
    from vit_amd import VTCNN, Dims
    dims = Dims(num_classes=classes, **SHAPES["blood"])
    m = VTCNN(dims, max_batch=batch, seed=42)
    g = torch.Generator(device="cuda")
    g.manual_seed(321)
    x = (torch.randint(0, 256, (2, batch, 1, 28, 28), device="cuda", generator=g).to(torch.float32) / 255.0).contiguous()
    y = torch.randint(0, classes, (2, batch), device="cuda", generator=g).to(torch.int32)

    def one(i):
        if group > 0:
            m.train_step_grouped(x[i & 1], y[i & 1], group, float(group))
        else:
            m.train_step(x[i & 1], y[i & 1], float(batch))
    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        one(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    finite = bool(torch.isfinite(m.arena).all().item())
    _, total_bytes = algorithmic_bytes_per_image(dims, False, False)
    gbs = total_bytes * batch / dt / 1e9
    del m, x, y
    torch.cuda.empty_cache()
    return {"ms_per_step": round(dt * 1e3, 4), "images_per_sec": round(batch / dt, 1), "batch": batch,
            "statistics_group": group, "steps": steps, "warmup": warmup, "dtype": "f32", "emulation": emulation_note(False, False),
            "weights_finite_after_run": finite,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_step": total_bytes * batch, "traffic": None}}


def dry_run(a, world, rank):
    """VT_BENCH_DRY=1: rehearsal of everything around the device work on a machine without a GPU (CPU tests): the
    ranks rendezvous over gloo, count themselves, exchange a gradient-sized buffer, and rank 0 prints the line with
    `value` null."""
    import torch.distributed as dist
    from vit_amd.engine import Dims, layout_sizes
    dims = Dims(num_classes=a.classes, **SHAPES[a.shape])
    n_total, n_train = layout_sizes(dims)
    joined = 1
    if world > 1:
        dist.init_process_group("gloo")
        ones = torch.ones(1)
        dist.all_reduce(ones)
        joined = int(ones.item())
        g = torch.full((n_train,), float(rank + 1))
        dist.all_reduce(g)
        assert float(g[0]) == world * (world + 1) / 2 and float(g[-1]) == float(g[0])
    if rank == 0:
        print(json.dumps({"metric": "train images/sec (whole node) VTCNN BloodMNIST", "value": None, "unit": "images/sec",
                          "n_gpus": joined, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "rehearsal without a device: rendezvous and exchange only",
                                     "per_gpu_batch": a.batch, "global_batch": a.batch * joined, "parallelism": "dp%d" % joined,
                                     "grad_allreduce_floats": int(n_train) if world > 1 else 0}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="images per mini-batch per GPU")
    ap.add_argument("--classes", type=int, default=8)
    ap.add_argument("--shape", choices=tuple(SHAPES), default="blood",
                    help="blood: the headline 1x28x28 shape; scaled / wide: the 224x224x3 shapes (general-shape step)")
    ap.add_argument("--mode", choices=("train", "predict"), default="train",
                    help="train: the headline metric; predict: forward-only images/sec (eval mode)")
    ap.add_argument("--group", type=int, default=0,
                    help="the batch is consecutive reference mini-batches of this many images, each with its own BatchNorm "
                         "statistics: predict mode = what run.sh predict evaluates; train mode = that many independent "
                         "mini-batches trained side by side with one summed update (vt_model_train_step_grouped)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one HIP graph (single GPU, train mode; for small batches)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: data-parallel gradient exchange hidden behind compute (vit_amd/parallel.py)")
    ap.add_argument("--bf16", action="store_true",
                    help="general shapes only: forward-type products through bf16 operands (parity 1e-2, not 1e-4)")
    ap.add_argument("--exchange-alone", action="store_true",
                    help="rehearsal on one GPU: a one-rank RCCL group whose all-reduces are really launched")
    ap.add_argument("--concurrent", type=int, default=1,
                    help="0: issue the whole backward on one stream (default: two independent chains on two streams)")
    ap.add_argument("--defer-attention", type=int, default=0,
                    help="1: the attention block's backward and weight update run on a stream of their own "
                         "beside the next step's forward, which never reads those weights (same bytes; "
                         "vt_model_set_deferred_attention); only without a gradient exchange (one GPU)")
    ap.add_argument("--fused-reductions", type=int, default=-1, help="BatchNorm reductions closed inside their producer kernels: 1 / 0, default: the library's choice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the three short runs of the scaled and wide configurations behind the headline timing")
    ap.add_argument("--kernel", default=None, help="launcher to time with HIP events (default: the slowest)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the process group decides\n" % (a.gpus, world))
    if os.environ.get("VT_BENCH_DRY", "") not in ("", "0"):
        return dry_run(a, world, rank)
    if os.environ.get("VT_BENCH_FAULT_TIMEOUT"):  # debugging aid: dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VT_BENCH_FAULT_TIMEOUT"]), exit=True)
    # rehearsal aid: with fewer GPUs than ranks (one-GPU box) ranks share devices, which RCCL refuses, so the
    # exchange can then be rehearsed over gloo (VT_BENCH_BACKEND=gloo); a real run has one GPU per rank and RCCL
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("VT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
            _all_reduce = dist.all_reduce

            def staged_all_reduce(t, *args, **kw):  # gloo rehearsal: device tensors travel through the host
                if not t.is_cuda:
                    return _all_reduce(t, *args, **kw)
                h = t.cpu()
                _all_reduce(h, *args, **{k: v for k, v in kw.items() if k != "async_op"})
                t.copy_(h)
                return None
            dist.all_reduce = staged_all_reduce
    elif a.exchange_alone:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))

    joined, distinct = 1, 1
    if dist is not None and world > 1:
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)  # every rank that joined the group adds one: this, not the flag, is n_gpus
        joined = int(ones.item())
        uu = torch.zeros(world, dtype=torch.int64, device="cuda")
        import zlib
        ident = str(getattr(torch.cuda.get_device_properties(local), "uuid", local))
        uu[rank] = zlib.crc32(ident.encode()) + 1
        dist.all_reduce(uu)
        distinct = len(set(uu.tolist()))

    from vit_amd import VTCNN, Dims
    dims = Dims(num_classes=a.classes, **SHAPES[a.shape])
    generic = a.shape != "blood"
    B = a.batch
    m = VTCNN(dims, max_batch=B, seed=42)
    if a.bf16:
        m.set_matmul_precision(True)
    m.set_concurrency(bool(a.concurrent))
    if a.fused_reductions >= 0 and not generic:
        m.set_fused_reductions(bool(a.fused_reductions))
    deferred = bool(a.defer_attention) and world == 1 and not generic and not a.exchange_alone and a.mode == "train"
    if deferred:
        m.set_deferred_attention(True)
    nb = 4  # resident mini-batches cycled through
    rng = np.random.default_rng(123 + rank)
    xs = torch.from_numpy(rng.integers(0, 256, size=(nb, B, dims.image_channels, dims.input_dim, dims.input_dim), dtype=np.uint8).astype(np.float32) / np.float32(255.0)).cuda()
    ys = torch.from_numpy(rng.integers(0, a.classes, size=(nb, B)).astype(np.int32)).cuda()
    from vit_amd.parallel import DataParallel
    # world 1: plain forward / backward / SGD; world > 1: one blocking all-reduce of the gradient arena
    # (--overlap 1: exchange hidden behind the attention backward and the next step's forward)
    dp = DataParallel(m, world=world, rank=rank, overlap=bool(a.overlap))
    dp.always_exchange = a.exchange_alone

    if a.mode == "predict":
        m.set_training(False)

    use_graph = a.graph and world == 1 and a.group == 0
    if use_graph:
        xbuf, ybuf = torch.empty_like(xs[0]), torch.empty_like(ys[0])
        if a.mode == "train":
            m.capture_step(xbuf, ybuf, float(B))
        else:
            m.capture_forward(xbuf)

    def step(i):
        if use_graph:
            xbuf.copy_(xs[i % nb], non_blocking=True)
            if a.mode == "train":
                ybuf.copy_(ys[i % nb], non_blocking=True)
            m.step_graph()
        elif a.mode == "predict" and a.group > 0:
            m.forward_grouped(xs[i % nb], a.group)
        elif a.mode == "predict":
            m.forward(xs[i % nb])
        elif a.group > 0 and world == 1 and not a.exchange_alone:
            m.train_step_grouped(xs[i % nb], ys[i % nb], a.group, float(a.group))  # divisor (B / group) * group = B
        elif a.group > 0:  # grouped forward / backward, the ranks' sums exchanged, one update
            m.forward_grouped(xs[i % nb], a.group)
            m.backward(ys[i % nb])
            dp.all_reduce_grads()
            m.update_weights(float(B) * world)
        else:
            dp.train_step(xs[i % nb], ys[i % nb], B)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # warm-up: also the per-launcher breakdown that picks the dominant kernel
    # (the breakdown uses only the last warm-up steps, after one-time module loading is over)
    nprof = min(3, max(a.warmup - 1, 1))
    for i in range(a.warmup):
        if i == a.warmup - nprof:
            sync()
            m.profile("*")
        step(i)
    dp.finish()
    sync()
    breakdown = m.profile_read()
    per_launch_ms = {k: v[0] / v[1] for k, v in breakdown.items()}
    tbl, total_bytes = algorithmic_bytes_per_image(dims, a.mode == "predict", generic)
    candidates = {k: per_launch_ms[k] for k in breakdown if k in tbl or k == "mha_bwd_rows"}
    # the dominant launcher = the slowest one of this run's warm-up breakdown (among those a roofline is defined for:
    # the byte model's launchers and the attention backward); `most_bytes_kernel` names the byte model's largest
    most_bytes = max(tbl, key=lambda k: tbl[k])
    dominant = a.kernel or (max(candidates, key=lambda k: candidates[k]) if candidates else most_bytes)
    # timed with HIP events on one step in eight: the two queue markers of an event pair stall the launch
    # stream for about 4.4 us each time, which would otherwise be part of every measured step
    m.profile(dominant + "/8")

    # Untimed settle steps between the warm-up's per-launcher breakdown (which ends in a device synchronisation and a
    # host read-back: the GPU idles and drops its clocks) and the timed region, so that a short timed region (the
    # driver's 20 steps are 5 ms) measures the steady state and not the ramp after an idle gap.
    # (the general shapes run milliseconds per step: their clocks are settled by the warm-up itself)
    settle = max(0, int(os.environ.get("VT_BENCH_SETTLE", "0" if generic else "60")))
    for i in range(settle):
        step(a.warmup + a.steps + i)
    m.profile(dominant + "/8")  # restart the event-pair count: only timed steps are counted

    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    dp.finish()  # the last step's trailing attention update belongs to the timed work
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = m.profile_read()
    m.profile(None)
    if hasattr(m, "join_attention"):
        m.join_attention()
    weights_finite = bool(torch.isfinite(m.arena).all().item())  # after every timed update: nothing overflowed

    # what an event pair itself adds to the interval it brackets (the markers are queue packets of
    # their own): the launcher's HIP-event time minus this is what a kernel trace reports
    s0 = torch.cuda.current_stream()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for e0, e1 in pairs:
        e0.record(s0)
        e1.record(s0)
    torch.cuda.synchronize()
    pair_us = float(np.median([e0.elapsed_time(e1) for e0, e1 in pairs])) * 1e3

    # yardstick: what a plain copy with the kernels' own access pattern (one contiguous region per
    # workgroup) reaches on this device once the footprint is beyond the 256 MB Infinity Cache
    copy_gbs = None
    if rank == 0 and a.mode == "train" and not generic:
        import ctypes as C
        from vit_amd._lib import lib as _lib
        blocks, floats = 4096, 50 * 256
        src = torch.rand(blocks * floats, device="cuda")
        dst = torch.empty_like(src)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(2):  # first round warms up
            e0.record()
            for _ in range(5):
                _lib.vt_probe_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), blocks, floats, 0, st)
            e1.record()
            torch.cuda.synchronize()
        copy_gbs = 5 * 2.0 * blocks * floats * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    if rank == 0:
        value = a.steps * B * joined / dt
        step_s = dt / a.steps
        traffic_tbl = measured_traffic_table(B, a.classes) if (a.mode == "train" and not generic) else None
        # per-launcher table from the warm-up breakdown (HIP events around every launcher; an event pair adds
        # `pair_us` to each interval): HBM form for the launchers the byte model charges, matrix form for
        # the attention backward, whose work is O(batch^2) and outside the byte model
        ktable = {}
        for name, ms in per_launch_ms.items():
            us = ms * 1e3
            row = {"us": round(us, 2)}
            if name in tbl:
                gbs = tbl[name] * B / (us * 1e-6) / 1e9
                row.update({"bound": "hbm", "algorithmic_bytes_per_launch": tbl[name] * B, "gbs": round(gbs, 1),
                            "frac": round(gbs / HBM_PEAK_GBS, 4)})
            elif name == "mha_bwd_rows":
                fl = attention_bwd_flops(dims, B)
                tf = fl / (us * 1e-6) / 1e12
                row.update({"bound": "mfma", "algorithmic_flops_per_launch": fl, "tflops": round(tf, 2),
                            "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "peak": MFMA_BF16_PEAK_TFLOPS,
                            "frac_of_f32_matrix_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                            "executed_bf16_flops_per_launch": int(fl * MHA_EXECUTED_PER_ALGORITHMIC)})
            if traffic_tbl and LAUNCHER_KERNEL.get(name) in traffic_tbl:
                row["traffic"] = traffic_tbl[LAUNCHER_KERNEL[name]]
            ktable[name] = row
        # the one launcher timed with HIP events over the measured steps
        k_ms, k_cnt = prof.get(dominant, (0.0, 0))
        avg_s = (k_ms / k_cnt) * 1e-3 if k_cnt else float("nan")
        if dominant == "mha_bwd_rows":  # compute bound on the matrix cores: algorithmic FLOPs over time
            fl = attention_bwd_flops(dims, B)
            ach = fl / avg_s / 1e12 if k_cnt else None
            dom = {"bound": "mfma", "kernel": dominant, "achieved": None if ach is None else round(ach, 2),
                   "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": None if ach is None else round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                   "traffic": measured_traffic(dominant, B, a.classes), "algorithmic_flops_per_launch": fl,
                   "executed_bf16_flops_per_launch": int(fl * MHA_EXECUTED_PER_ALGORITHMIC)}
        else:
            ach = tbl.get(dominant, 0) * B / avg_s / 1e9 if k_cnt else None
            dom = {"bound": "hbm", "kernel": dominant, "achieved": None if ach is None else round(ach, 1),
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None if ach is None else round(ach / HBM_PEAK_GBS, 4),
                   "traffic": measured_traffic(dominant, B, a.classes),
                   "algorithmic_bytes_per_launch": tbl.get(dominant, 0) * B}
        dom.update({"avg_launch_us": None if not k_cnt else round(avg_s * 1e6, 2), "launches_timed": k_cnt, "timed_every_nth_step": 8,
                    "event_pair_overhead_us": round(pair_us, 2),
                    "chosen": "the slowest launcher of this run's warm-up breakdown" if not a.kernel else "--kernel",
                    "most_bytes_kernel": most_bytes})
        # headline roofline: HBM is the bounding roofline of this shape (every contraction is 8 to 32 deep), so
        # the figure is the whole step: algorithmic bytes of one step over the measured time of one step
        # (all launches of the step; `ms_per_step` of this same line)
        step_gbs = total_bytes * B / step_s / 1e9
        roof = {"bound": "hbm", "kernel": "whole step (every launch of forward + backward + update)",
                "achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(step_gbs / HBM_PEAK_GBS, 4),
                "traffic": (measured_step_traffic(a.shape, B, a.classes) if (generic and a.mode == "train") else None)
                           if not traffic_tbl else int(sum(v * STEP_LAUNCHES.get(k, 1) for k, v in traffic_tbl.items()
                                                           if k != "k_probe_copy")),
                "traffic_source": "summed from the committed PMC passes of this configuration (profiles/, separate "
                                  "rocprofv3 --pmc runs of this command), not measured in this run",
                "algorithmic_bytes_per_launch": total_bytes * B, "avg_launch_us": round(step_s * 1e6, 2),
                "launches_timed": a.steps,
                "definition": "algorithmic layer-boundary bytes per image x images per step / measured step time; "
                              "per-launcher figures in `kernels`, the event-timed dominant launcher in `dominant_kernel`",
                "dominant_kernel": dom, "kernels": ktable}
        res = {
            "metric": ("train" if a.mode == "train" else "predict") + " images/sec (whole node) VTCNN " + {"blood": "BloodMNIST"}.get(a.shape, a.shape), "value": round(value, 1), "unit": "images/sec",
            "n_gpus": joined, "steps": a.steps, "warmup": a.warmup, "untimed_settle_steps": settle,
            "ms_per_step": round(1e3 * dt / a.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.bf16 else "f32",
            "emulation": emulation_note(generic, a.bf16),
            "data": "synthetic",
            "config": {"workload": ("VTCNN train step (forward+backward+SGD)" if a.mode == "train" else
                                    "VTCNN predict step (eval-mode forward)") + ", %s shape %dx%dx%d, %d classes, "
                                   "mini-batch %d per GPU, fp32, seed-42 reference init" % (
                                       {"blood": "BloodMNIST"}.get(a.shape, a.shape), dims.image_channels, dims.input_dim,
                                       dims.input_dim, a.classes, B),
                       "hip_graph": bool(use_graph), "two_stream_backward": bool(a.concurrent) and not generic, "deferred_attention_backward": deferred, "statistics_group": a.group, "per_gpu_batch": B, "global_batch": B * joined, "parallelism": "dp%d" % joined,
                       "distinct_devices": distinct,
                       "exchange_overlapped": bool(dp.overlap and a.mode == "train" and not use_graph),
                       "grad_allreduce_floats": int(m.layout.n_trainable) if world > 1 else 0},
            "weights_finite_after_run": weights_finite,
            "roofline": roof,
            "whole_step_hbm_frac": round(value / joined * total_bytes / (HBM_PEAK_GBS * 1e9), 4),  # algorithmic bytes, not measured
            "measured_copy_gbs": None if copy_gbs is None else round(copy_gbs, 1),
            "algorithmic_bytes_per_image": total_bytes,
            "launcher_us": {k: round(1e3 * v, 2) for k, v in sorted(per_launch_ms.items(), key=lambda kv: -kv[1])},
        }
        if world == 1 and not a.no_extra_configs and a.mode == "train" and not generic and not a.exchange_alone:
            # the other BASELINE configurations under the same clock: three steps each, after the headline model is gone
            del m, dp, xs, ys
            torch.cuda.empty_cache()
            extra = {}
            for name, shp, bt, bf in (("scaled_b1024_f32", "scaled", 1024, False), ("wide_b16_f32", "wide", 16, False),
                                      ("wide_b16_bf16", "wide", 16, True)):
                try:
                    extra[name] = time_extra_config(shp, bt, bf, a.classes)
                except Exception as e:  # the headline line must survive a failure here
                    extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                    torch.cuda.empty_cache()
            # the headline shape as the reference batches it: 32 loader batches of 32 side by side (one summed update),
            # twice that, and one plain loader batch
            for name, bt, grp in (("blood_b1024_group32_f32", 1024, 32), ("blood_b4096_group32_f32", 4096, 32),
                                  ("blood_b32_f32", 32, 0)):
                try:
                    extra[name] = time_blood_config(bt, grp, a.classes)
                except Exception as e:
                    extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                    torch.cuda.empty_cache()
            # what exact fp32 products cost on the headline shape: the general-shape launchers with and without them
            for name, ex in (("blood_b1024_general_launchers_split_f32", False), ("blood_b1024_general_launchers_exact_f32", True)):
                try:
                    extra[name] = time_blood_general_path(1024, a.classes, ex)
                except Exception as e:
                    extra[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                    torch.cuda.empty_cache()
            res["configs"] = extra
            gs, ge = extra["blood_b1024_general_launchers_split_f32"], extra["blood_b1024_general_launchers_exact_f32"]
            if "ms_per_step" in gs and "ms_per_step" in ge:
                res["emulation"] += ("; the same step on the general-shape launchers: %.3f ms with split products, %.3f ms with "
                                     "every product on exact fp32 MFMA (vt_set_exact_products)" % (gs["ms_per_step"], ge["ms_per_step"]))
        if world == 1 and not a.no_cpu_baseline and a.mode == "train" and not generic:
            res["cpu_baseline"] = cpu_baseline(B, a.classes)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
