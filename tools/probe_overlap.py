"""One-GPU rehearsal of the data-parallel exchange patterns: a one-rank RCCL group (its collectives are
issued for real: stream hand-offs, work handles) around the phased backward.  Prints ms/step and the
host's issue time per step for each pattern, to see what the pattern itself costs before any link
latency is added."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch.distributed as dist
    from vit_amd import VTCNN, Dims
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = 200
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    m = VTCNN(Dims(num_classes=8), max_batch=B, seed=42)
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.integers(0, 256, size=(B, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda()
    y = torch.from_numpy(rng.integers(0, 8, size=B).astype(np.int32)).cuda()
    first, count = m.attention_range()
    n = m.grads.numel()
    g = m.grads
    div = float(B)
    S = dist.ReduceOp.SUM

    def plain():
        m.forward(x); m.backward(y); m.update_weights(div)

    def plain_sync_ar():
        m.forward(x); m.backward(y); dist.all_reduce(g, op=S); m.update_weights(div)

    def phased():
        m.forward(x); m.backward_main(y); m.backward_attention()
        m.update_weights_range(div, 0, first); m.update_weights_range(div, first + count, n - first - count)
        m.update_weights_range(div, first, count)

    def phased_head2():
        m.forward(x); m.backward_main(y)
        w = [dist.all_reduce(g[:first], op=S, async_op=True), dist.all_reduce(g[first + count:], op=S, async_op=True)]
        m.backward_attention()
        for h in w:
            h.wait()
        m.update_weights_range(div, 0, first); m.update_weights_range(div, first + count, n - first - count)
        m.update_weights_range(div, first, count)

    def phased_head1():
        m.forward(x); m.backward_main(y)
        h = dist.all_reduce(g[first + count:], op=S, async_op=True)
        m.backward_attention()
        h.wait()
        m.update_weights_range(div, 0, first); m.update_weights_range(div, first + count, n - first - count)
        m.update_weights_range(div, first, count)

    state = {"t": None}

    def phased_trailing_only():
        m.forward(x); m.backward_main(y)
        if state["t"] is not None:
            state["t"].wait(); m.update_weights_range(div, first, count)
        m.backward_attention()
        m.update_weights_range(div, 0, first); m.update_weights_range(div, first + count, n - first - count)
        state["t"] = dist.all_reduce(g[first:first + count], op=S, async_op=True)

    def phased_all3():
        m.forward(x); m.backward_main(y)
        w = [dist.all_reduce(g[:first], op=S, async_op=True), dist.all_reduce(g[first + count:], op=S, async_op=True)]
        if state["t"] is not None:
            state["t"].wait(); m.update_weights_range(div, first, count)
        m.backward_attention()
        for h in w:
            h.wait()
        m.update_weights_range(div, 0, first); m.update_weights_range(div, first + count, n - first - count)
        state["t"] = dist.all_reduce(g[first:first + count], op=S, async_op=True)

    def phased_sync_positions():
        m.forward(x); m.backward_main(y)
        dist.all_reduce(g[first + count:], op=S)
        m.backward_attention()
        m.update_weights_range(div, 0, first); m.update_weights_range(div, first + count, n - first - count)
        dist.all_reduce(g[first:first + count], op=S)
        m.update_weights_range(div, first, count)

    side = torch.cuda.Stream()
    stage = torch.zeros(count, dtype=torch.float32, device="cuda")
    e1, e2, e3, etr = (torch.cuda.Event() for _ in range(4))

    def side_stream_2():
        main = torch.cuda.current_stream()
        m.forward(x); m.backward_main(y)
        e1.record(main); side.wait_event(e1)
        with torch.cuda.stream(side):
            dist.all_reduce(g, op=S)
        e2.record(side)
        if state.get("s"):
            main.wait_event(etr); m.update_block(div, first, stage)
        m.backward_attention(stage)
        main.wait_event(e2)
        m.update_weights(div)
        e3.record(main); side.wait_event(e3)
        with torch.cuda.stream(side):
            dist.all_reduce(stage, op=S)
        etr.record(side)
        state["s"] = True

    def side_stream_head_only():
        main = torch.cuda.current_stream()
        m.forward(x); m.backward_main(y)
        e1.record(main); side.wait_event(e1)
        with torch.cuda.stream(side):
            dist.all_reduce(g, op=S)
        e2.record(side)
        m.backward_attention(stage)
        main.wait_event(e2)
        m.update_weights(div)
        m.update_block(div, first, stage)

    for name, fn in (("plain", plain), ("side_stream_head_only", side_stream_head_only), ("side_stream_2", side_stream_2), ("plain_sync_ar", plain_sync_ar), ("phased", phased), ("phased_head1", phased_head1),
                     ("phased_head2", phased_head2), ("phased_trailing_only", phased_trailing_only),
                     ("phased_all3", phased_all3), ("phased_sync_positions", phased_sync_positions)):
        state["t"] = None
        state["s"] = False
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if state["t"] is not None:
            state["t"].wait(); m.update_weights_range(div, first, count)
        if state.get("s"):
            torch.cuda.current_stream().wait_event(etr); m.update_block(div, first, stage)
        torch.cuda.synchronize()
        print("%-24s %.4f ms/step   host issue %.4f ms/step" % (name, (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
