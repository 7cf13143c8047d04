#!/usr/bin/env python3
"""Where a mini-batch training step of a learnable-aggregator SGAP model goes on the device (the device-resident
training feed; reference loop: sgl/tasks/utils.py:66-76 `train` -> models/base_model.py:58-66 `forward`).

A zoo model (default GAMLP: 'jk' gate over K + 1 hops; --model picks another aggregator) on the products-shaped graph, d + C = 147, batch B: per phase (HIP events) -- row gather of the
K + 1 hop matrices, the aggregator forward, the MLP forward, loss, backward, optimizer -- and the torch profiler's top device
kernels / host ops of a step.

    python tools/profile_train_step.py [--workload S1_products] [--batch 50000] [--prop-steps 5] [--steps 20] [--model GAMLPRecursive] [--profile]"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgl_amd import synthetic  # noqa: E402
from sgl_amd.io import DeviceAdjacency  # noqa: E402
from sgl_amd.models.base_model import take_rows  # noqa: E402
from sgl_amd.models import homo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S1_products")
    ap.add_argument("--batch", type=int, default=50_000)
    ap.add_argument("--prop-steps", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--classes", type=int, default=47)
    ap.add_argument("--model", default="GAMLP", choices=["GAMLP", "GAMLPRecursive", "PASCA_V1", "PASCA_V2"],
                    help="a zoo model with a LEARNABLE aggregator (the others aggregate once, in preprocess)")
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    device = torch.device("cuda")
    wl = synthetic.WORKLOADS[a.workload]
    n, d, C = wl["n"], 100 + a.classes, a.classes
    rowptr, col, val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    adj = DeviceAdjacency(rowptr, col, val, (n, n))
    g = torch.Generator(device=device).manual_seed(0)
    x = torch.randn((n, d), generator=g, device=device)
    y = torch.randint(0, C, (n,), generator=g, device=device)
    cls = getattr(homo, a.model)
    model = cls(a.prop_steps, d, C, 256, 3).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    model.preprocess(adj, x)
    model.train()
    idx = torch.randperm(n, generator=g, device=device)[: a.batch]
    yb = y[idx]

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def step(timers=None):
        marks = [ev() for _ in range(7)]
        marks[0].record()
        rows = [take_rows(f, idx, device) for f in model._processed_feat_list]
        marks[1].record()
        agg = model._pre_msg_op.aggregate(rows)
        marks[2].record()
        out = model._base_model(agg)
        marks[3].record()
        loss = F.cross_entropy(out, yb)
        marks[4].record()
        opt.zero_grad()
        loss.backward()
        marks[5].record()
        opt.step()
        marks[6].record()
        if timers is not None:
            torch.cuda.synchronize()
            for k, name in enumerate(("gather_rows x H", "aggregate fwd", "MLP fwd", "loss", "backward", "optimizer")):
                timers[name] = timers.get(name, 0.0) + marks[k].elapsed_time(marks[k + 1])

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    timers = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(timers)
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    print(f"TRAIN_STEP model={a.model} workload={a.workload} B={a.batch} d={d} H={a.prop_steps + 1}: {wall:.3f} ms per step (wall, phases synchronised)")
    for k, v in timers.items():
        print(f"TRAIN_STEP   {k:22s} {v / a.steps:8.3f} ms")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    print(f"TRAIN_STEP   free-running: {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms per step")
    if a.profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(5):
                step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=15, max_name_column_width=70))


if __name__ == "__main__":
    main()
