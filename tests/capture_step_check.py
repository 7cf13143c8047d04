"""Body of tests/test_gpu_parity.py::test_training_steps_record_into_a_hip_graph, run in a process of its own: a capture that goes
wrong inside torch / the HIP runtime ends the PROCESS (a segmentation fault at capture_end, not an exception), and must not take
the rest of the GPU suite with it.  Prints CAPTURE-OK <n_ops> on success."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import _lib  # noqa: E402
from sgl_amd import device as dev  # noqa: E402


def main():
    _lib.require_gpu()
    torch.cuda.set_device(0)
    cuda = torch.device("cuda", 0)
    from sgl_amd.operators import message_op as mo
    n, d, H, B = 4000, 24, 4, 700
    rng = np.random.default_rng(8)
    hops = [dev.upload_rows(rng.standard_normal((n, d)).astype(np.float32), cuda) for _ in range(H)]
    idx = torch.from_numpy(rng.integers(0, n, size=B)).to(cuda)
    ops = {"simple": mo.LearnableWeightedMessageOp(0, H, "simple", H - 1), "simple_allow_neg": mo.LearnableWeightedMessageOp(0, H, "simple_allow_neg", H - 1),
           "gate": mo.LearnableWeightedMessageOp(0, H, "gate", d), "ori_ref": mo.LearnableWeightedMessageOp(0, H, "ori_ref", d),
           "jk": mo.LearnableWeightedMessageOp(0, H, "jk", H - 1, d), "iterate": mo.IterateLearnableWeightedMessageOp(0, H, "recursive", d),
           "sum": mo.SumMessageOp(0, H), "mean": mo.MeanMessageOp(0, H + 1), "max": mo.MaxMessageOp(0, H), "min": mo.MinMessageOp(1, H),
           "concat": mo.ConcatMessageOp(0, H), "last": mo.LastMessageOp(), "over_smooth": mo.OverSmoothDistanceWeightedOp(),
           "simple_weighted": mo.SimpleWeightedMessageOp(0, H, "alpha", 0.85),
           "hand_crafted": mo.SimpleWeightedMessageOp(1, H, "hand_crafted", [0.5, 0.25, 0.125])}
    for name, op in ops.items():
        torch.manual_seed(1)
        op = op.to(cuda)
        scale = torch.nn.Parameter(torch.ones(d, device=cuda))          # makes the gathered rows carry a gradient (the stateless ops' case)
        params = [scale] + list(op.parameters())
        learnable = name in ("simple", "simple_allow_neg", "gate", "ori_ref", "jk", "iterate")

        def step():
            for p_ in params:
                p_.grad = None
            rows = dev.gather_hops(hops, idx)
            if name != "over_smooth" and not learnable:                  # (NAFS's op is evaluated without gradients: base_model.py:32)
                rows = [r_ * scale for r_ in rows]
            out = op.aggregate(rows)
            if name != "over_smooth":
                (out * out).mean().backward()
            # (only the detached output leaves the step: a loss kept alive keeps its AccumulateGrad nodes -- created on the eager
            # stream -- alive, and a backward that reaches them from inside a capture takes the process down; torch warns about it)
            return out.detach()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        want = step().clone()
        want_g = [None if p_.grad is None else p_.grad.clone() for p_ in params]
        for p_ in params:
            p_.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            got = step()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want), name
        for p_, g_ in zip(params, want_g):
            assert (p_.grad is None) == (g_ is None) and (g_ is None or torch.allclose(p_.grad, g_, rtol=1e-6, atol=1e-8)), name
        # other rows, other values, same graph
        idx.copy_(torch.from_numpy(rng.integers(0, n, size=B)).to(cuda))
        hops[1].mul_(1.5)
        graph.replay()
        torch.cuda.synchronize()
        replayed = got.clone()
        eager = step()
        assert torch.equal(replayed, eager), name
        # (what lives in this graph's private pool is released before the next capture begins)
        del got, eager, replayed, graph
        torch.cuda.synchronize()
    print("CAPTURE-OK", len(ops), flush=True)


if __name__ == "__main__":
    main()
