#!/usr/bin/env python3
"""What a community-aware PARTITION would buy the need-aware exchange (a "what comes next" measurement).

The row-sharded job cuts the node ids into G contiguous blocks.  With the need-aware exchange a rank receives only the rows its
block references -- on a graph without structure (the benchmark graph) that is still 84 % of all foreign rows at G = 8.  Real
co-purchase / citation graphs have communities; if the blocks followed them, most references would stay inside the block.
This tool takes a products-sized graph with 80 % of its edges inside communities of `bs` nodes and counts, for G = 2 / 4 / 8,
the rows every rank would have to receive per hop

    (a) with the ids shuffled            (what a dump looks like: the blocks cut through every community)
    (b) in the order the plan-time label propagation finds (sgl_amd.reorder.community_order -> relabel, then cut)
    (c) in the generator's own community order (the upper bound)

against the full all-gather.  It only counts (HaloPlan.offline); nothing is propagated.

    python tools/partition_locality.py  > profiles/r03_partition_locality.log
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import synthetic  # noqa: E402
from sgl_amd.dist import HaloPlan, balanced_bounds  # noqa: E402
from sgl_amd.reorder import community_order, permute_csr  # noqa: E402
from tools.bench_reorder import community_graph  # noqa: E402


def ghost_stats(rp, cc, n, G):
    rp_h = rp.cpu().numpy()
    bounds = balanced_bounds(rp_h, G)
    ghosts, full = [], []
    for r in range(G):
        plan = HaloPlan.offline(r, bounds, n, lambda q: cc[int(rp_h[bounds[q]]):int(rp_h[bounds[q + 1]])])
        ghosts.append(plan.n_ghost)
        full.append(plan.rows_in_full)
    return max(ghosts), float(np.mean(ghosts)), max(full)


def main():
    device = torch.device("cuda", 0)
    wl = synthetic.WORKLOADS["S1_products"]
    n, d = wl["n"], wl["d"]
    print(f"# tools/partition_locality.py: rows a rank must RECEIVE per hop under the need-aware exchange (N = {n}, d = {d}: x {d * 4} bytes), "
          "contiguous nnz-balanced blocks, slowest rank / mean over ranks / full all-gather")
    for bs in (2048, 16384):
        rp, cc, vv = community_graph(n, wl["m"], bs, device)
        g = torch.Generator(device=device).manual_seed(11)
        shuffle = torch.randperm(n, generator=g, device=device)
        rp_s, cc_s, vv_s = permute_csr(rp, cc, vv, shuffle)                     # ids shuffled: what a dump looks like
        order, info = community_order(rp_s, cc_s, n)
        rp_l, cc_l, vv_l = permute_csr(rp_s, cc_s, vv_s, order)                 # relabelled in label-propagation order
        for G in (2, 4, 8):
            for name, (a, b) in (("shuffled ids", (rp_s, cc_s)), ("label-propagation order", (rp_l, cc_l)), ("generator order", (rp, cc))):
                mx, mean, full = ghost_stats(a, b, n, G)
                print(f"PART community={bs} G={G} {name:24s} ghost_rows max={mx:8d} mean={mean:10.0f} full_allgather={full:8d} "
                      f"received_fraction={mx / full:.3f} MB_per_hop={mx * d * 4 / 1e6:7.1f}", flush=True)
        print(f"PART community={bs} ordering: {info}", flush=True)
    # the benchmark graph for reference: nothing to find
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    for G in (2, 4, 8):
        mx, mean, full = ghost_stats(a_ptr, a_col, n, G)
        print(f"PART S1 benchmark graph (no communities) G={G} ghost_rows max={mx} full_allgather={full} received_fraction={mx / full:.3f}", flush=True)


if __name__ == "__main__":
    main()
