#!/usr/bin/env python3
"""A PaSca-style search through the plugin API (BASELINE config 5): every trial builds FRESH operators -- one of
{Laplacian r=0.5, PPR alpha in 0.1/0.2/0.3} x k = 10 and one MessageOp (sgl/search/search_models.py:19-46,
search_config.py:14-15) -- and runs propagate + aggregate on the same graph and features, as the reference's search does.
Timed with sgl_amd.config.share_hops off and on (wall clock per trial incl. every host step; the quantity PaSca's
`time_preprocess` objective sees, auto_search.py:29,54)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgl_amd import config, hopcache, synthetic  # noqa: E402
from sgl_amd.io import DeviceAdjacency  # noqa: E402
from sgl_amd.operators import message_op as M  # noqa: E402
from sgl_amd.operators.graph_op import LaplacianGraphOp, PprGraphOp  # noqa: E402


def main():
    wl = synthetic.WORKLOADS[os.environ.get("SWEEP_WORKLOAD", "S1_products")]
    n, d, K = wl["n"], wl["d"], 10
    trials = int(os.environ.get("SWEEP_TRIALS", "24"))
    device = torch.device("cuda", 0)
    a_ptr, a_col, a_val = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
    adj = DeviceAdjacency(a_ptr, a_col, a_val, (n, n))
    x0 = synthetic.features_torch(n, d, seed=0, device=device)
    graph_ops = [lambda: LaplacianGraphOp(K, r=0.5)] + [(lambda a: (lambda: PprGraphOp(K, r=0.5, alpha=a)))(a) for a in (0.1, 0.2, 0.3)]
    msg_ops = [lambda: M.LastMessageOp(), lambda: M.SumMessageOp(0, K + 1), lambda: M.MeanMessageOp(0, K + 1),
               lambda: M.MaxMessageOp(0, K + 1), lambda: M.MinMessageOp(0, K + 1), lambda: M.ConcatMessageOp(0, K + 1)]
    rng = np.random.default_rng(0)
    plan = [(int(rng.integers(len(graph_ops))), int(rng.integers(len(msg_ops)))) for _ in range(trials)]
    sums = {}
    for share in (False, True):
        config.share_hops = share
        hopcache.SHARED.clear()
        ts = []
        sums[share] = []
        for gi, mi in plan:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hops = graph_ops[gi]().propagate(adj, x0)
            out = msg_ops[mi]().aggregate(hops)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            sums[share].append(float(out.double().abs().mean()))
            del hops, out
        ts = np.array(ts) * 1e3
        print(f"SEARCH share_hops={share!s:5s} trials={trials} total_ms={ts.sum():9.1f} median_ms={np.median(ts):8.2f} "
              f"first_ms={ts[0]:8.2f} min_ms={ts.min():8.2f} max_ms={ts.max():8.2f} store={dict(hopcache.SHARED.stats)}", flush=True)
    dev_ = max(abs(a - b) / max(abs(a), 1e-30) for a, b in zip(sums[False], sums[True]))
    print(f"SEARCH mean |out| of every trial agrees between the two runs to {dev_:.2e} (mixed PPR chains: float32 rounding)")
    assert dev_ < 1e-5


if __name__ == "__main__":
    main()
