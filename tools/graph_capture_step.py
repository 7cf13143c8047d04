#!/usr/bin/env python3
"""The learnable models' training step -- row gather of every hop matrix, the learnable aggregator, the head, and the backward of all
three -- captured ONCE in a HIP graph (torch.cuda.CUDAGraph) and replayed: the library's launches are stream-ordered, allocate through
torch's caching allocator only and never synchronise, so they record into a capture like torch's own kernels.  Wall time per step,
eager vs replay, at the batch sizes of the reference's GAMLP example (examples/gamlp_products.py: 50 000)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import synthetic  # noqa: E402
from sgl_amd.io import DeviceAdjacency  # noqa: E402
from sgl_amd.models.homo import GAMLP, GAMLPRecursive  # noqa: E402


def main():
    wl = synthetic.WORKLOADS[os.environ.get("STEP_WORKLOAD", "S1_small")]
    n, d, K, C = wl["n"], wl["d"], 5, 47
    device = torch.device("cuda", 0)
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    adj = DeviceAdjacency(a_ptr, a_col, a_val, (n, n))
    x = synthetic.features_torch(n, d, seed=0, device=device)
    for cls in (GAMLP, GAMLPRecursive):
        for batch in (10_000, 50_000):
            torch.manual_seed(0)
            model = cls(K, d, C, 256, 2).to(device).eval()            # (eval: no dropout mask, so eager and replay are comparable)
            model.preprocess(adj, x)
            params = [p for p in model.parameters() if p.requires_grad]
            idx = torch.randint(0, n, (batch,), device=device)
            y = torch.randint(0, C, (batch,), device=device)

            def step():
                for p in params:
                    p.grad = None
                out = model.model_forward(idx, device)
                loss = torch.nn.functional.cross_entropy(out, y)
                loss.backward()
                return loss.detach()           # (a loss kept alive would keep its AccumulateGrad nodes, and their eager stream, alive)

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            eager_loss = float(step())
            eager_grads = [p.grad.clone() for p in params]
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                loss = step()
            graph.replay()
            torch.cuda.synchronize()
            same = abs(float(loss) - eager_loss) <= 1e-6 * max(abs(eager_loss), 1.0) and \
                all(torch.allclose(p.grad, g_, rtol=1e-5, atol=1e-7) for p, g_ in zip(params, eager_grads))
            # new indices in place: the replay gathers other rows
            idx.copy_(torch.randint(0, n, (batch,), device=device))
            graph.replay()
            torch.cuda.synchronize()
            replay_loss = float(loss)
            eager2 = float(step())
            moved = abs(replay_loss - eager2) <= 1e-5 * max(abs(eager2), 1.0)

            def wall(fn, reps=30):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps * 1e3
            t_eager, t_graph = wall(step), wall(graph.replay)
            print(f"STEP {cls.__name__:15s} batch={batch:6d} K={K} d={d} eager_ms={t_eager:7.3f} graph_replay_ms={t_graph:7.3f} "
                  f"same_loss_and_grads={same} follows_new_indices={moved}", flush=True)
            assert same and moved


if __name__ == "__main__":
    main()
