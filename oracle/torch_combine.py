"""torch-CPU restatement of the reference's MessageOp._combine bodies -- TEST / BASELINE INFRASTRUCTURE ONLY.

What the reference itself executes for an aggregate is a handful of torch CPU operations (and, for the NAFS weighting, a Python
loop over the nodes).  The numpy functions of ref_ops.py restate the ARITHMETIC (they are the parity oracle); the functions here
restate the EXECUTION -- the same torch calls in the same shape -- so that bench.py's cpu_baseline leg can time "the reference
CPU path" of an aggregate on the GPU node's host cores next to the HIP kernels.  Pinned to ref_ops.py (itself pinned to golden
vectors recorded from the reference) by tests/test_oracle_golden.py::test_torch_combine_equals_the_numpy_oracle.

Nothing under sgl_amd/ imports this module."""
import torch
import torch.nn.functional as F


def combine_mean(feat_list, start, end):
    """mean_message_op.py:9-10: Python sum() of the slice, one division"""
    return sum(feat_list[start:end]) / (end - start)


def combine_sum(feat_list, start, end):
    """sum_message_op.py:9-10"""
    return sum(feat_list[start:end])


def combine_max(feat_list, start, end):
    """max_message_op.py:11-12: the [H, n, d] stack is materialised, then reduced"""
    return torch.stack(feat_list[start:end], dim=0).max(dim=0)[0]


def combine_min(feat_list, start, end):
    """min_message_op.py:11-12"""
    return torch.stack(feat_list[start:end], dim=0).min(dim=0)[0]


def combine_concat(feat_list, start, end):
    """concat_message_op.py:11-12"""
    return torch.hstack(feat_list[start:end])


def nafs_weight(feat_list):
    """over_smooth_distance_op.py:12-22: cosine of every hop's row with hop 0's row (1e-10 added to each norm), soft-max over hops"""
    x0 = feat_list[0]
    n0 = torch.norm(x0, 2, 1).add(1e-10)
    cols = []
    for fea in feat_list:
        nh = torch.norm(fea, 2, 1).add(1e-10)
        cols.append(torch.div(torch.div((x0 * fea).sum(1), nh), n0).unsqueeze(-1))
    return F.softmax(torch.cat(cols, dim=1), dim=1)


def combine_over_smooth_distance(feat_list):
    """over_smooth_distance_op.py:11-33 as the reference runs it: the weights vectorised, the weighted hop sum as a PYTHON LOOP over
    nodes and hops (:27-31) -- which is where its time goes (tens of microseconds per node)"""
    weight = nafs_weight(feat_list)
    hops = len(feat_list)
    out = []
    for i in range(feat_list[0].shape[0]):
        acc = 0.
        for j in range(hops):
            acc = acc + (weight[i][j] * feat_list[j][i]).unsqueeze(0)
        out.append(acc)
    return torch.cat(out, dim=0)


def combine_over_smooth_distance_vectorised(feat_list):
    """the same sums without the Python loop (what a maintainer would write; NOT what the reference runs): hop order, from 0."""
    weight = nafs_weight(feat_list)
    acc = torch.zeros_like(feat_list[0])
    for j, fea in enumerate(feat_list):
        acc = acc + weight[:, j:j + 1] * fea
    return acc
