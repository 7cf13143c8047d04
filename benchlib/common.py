"""Shared by bench.py and its helper modules: constants, the metric string, the algorithmic byte model, workload texts."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_BYTES = 8.0e12  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def baseline_metric():
    """the metric string of BASELINE.json, verbatim (the file travels with the repo snapshot)"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:  # noqa: BLE001
        return "pre-prop SpMM throughput (edge\u00b7featdim/s), ogbn-products k=3, 1/2/4/8 GPU"


def algorithmic_bytes_per_hop(n, nnz, d):
    """SURVEY.md section 8(d) no-reuse gather model: gathered X rows + (col,val) + rowptr + Y write"""
    return nnz * d * 4 + nnz * 8 + (n + 1) * 4 + n * d * 4


class _QuietStdout:
    """RCCL prints a version banner through C stdio (flushed at exit, i.e. AFTER our JSON line).  The driver reads ONE
    JSON line from stdout, so everything except that line is routed to stderr at the file-descriptor level."""

    def __init__(self):
        self.saved = None

    def mute(self):
        if self.saved is None:
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)

    def unmute(self):
        if self.saved is not None:
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)      # push buffered C-level output out while fd 1 still is stderr
            except Exception:  # noqa: BLE001
                pass
            sys.stdout.flush()
            os.dup2(self.saved, 1)
            os.close(self.saved)
            self.saved = None


WORKLOAD_TEXT = {
    "S0": "SGC prop_steps={K} pre-propagation on a Pubmed-sized Chung-Lu graph (BASELINE config 1), LaplacianGraphOp r=0.5",
    "S1": "SGC prop_steps={K} pre-propagation on an ogbn-products-shaped Chung-Lu graph (BASELINE config 2), LaplacianGraphOp r=0.5",
    "S1_community": "SGC prop_steps={K} pre-propagation on the ogbn-products degree law WITH planted communities (4 096 nodes, p_in = 0.8) and "
                    "shuffled node ids (secondary workload: what a locality ordering can find), LaplacianGraphOp r=0.5",
    "S2": "GAMLP label-reuse sized propagation (d=147, prop_steps={K}) on the ogbn-products-shaped graph (BASELINE config 3), "
          "LaplacianGraphOp r=0.5",
    "S3_papers_shard": "one rank's 1/8 row block of an ogbn-papers100M-shaped hashed graph against the full 111 M x 128 feature "
                       "replica (BASELINE configs 4/5, per-GPU share of the 8-GPU job), {K} hop launch(es) per step",
    "S3": "prop_steps={K} propagation on an ogbn-papers100M-shaped hashed graph (BASELINE configs 4/5; directed, generated per "
          "row block on device, values used as A_hat directly: throughput only)",
}


def workload_text(name, K):
    for key in sorted(WORKLOAD_TEXT, key=len, reverse=True):
        if name.startswith(key):
            return f"{name}: " + WORKLOAD_TEXT[key].format(K=K)
    return f"{name}: prop_steps={K} pre-propagation (test workload)"


def _replayed_profile(workload, world):
    """rocprofv3 figures of the same command kept under profiles/ (PMC counters cannot be collected inside the timed run)"""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tfile))
    except Exception:  # noqa: BLE001
        return None
    if "workload" in tj:                              # round-1 layout: a single entry
        tj = {tj["workload"]: tj}
    return tj.get(workload) if world == 1 else None


_PHASE = ["start"]


def _phase(name):
    _PHASE[0] = name
