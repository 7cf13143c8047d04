"""Who owns what: nnz-balanced row blocks and pieces, column chunks / slices, the rows x columns grid of ranks."""
import numpy as np


def balanced_bounds(rowptr, parts, weights=None):
    """Cut rows [0, n) into `parts` contiguous blocks holding ~equal numbers of non-zeros (+1 per row so
    empty rows still count), or shares proportional to `weights`.  rowptr: host int64 array [n+1].
    Returns int64 array [parts+1]."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    n = len(rowptr) - 1
    cost = rowptr + np.arange(n + 1, dtype=np.int64)
    if weights is None:
        share = np.arange(1, parts, dtype=np.float64) / parts
    else:
        w = np.asarray(weights, dtype=np.float64)
        if len(w) != parts or (w <= 0).any():
            raise ValueError("one positive weight per part")
        share = np.cumsum(w)[:-1] / w.sum()
    cuts = np.searchsorted(cost, cost[-1] * share, side="left").astype(np.int64)
    b = np.concatenate([[0], np.clip(cuts, 0, n), [n]]).astype(np.int64)
    return np.maximum.accumulate(b)


def tapered_weights(pieces):
    """piece sizes for a hop whose LAST piece's transfer cannot hide behind compute: equal pieces, the last one half
    as large (4 pieces -> 2:2:2:1, the exposed transfer is 1/7 instead of 1/4 of the hop's traffic)"""
    return [2.0] * (pieces - 1) + [1.0] if pieces > 1 else [1.0]


def piece_bounds(rowptr, lo, hi, pieces, weights=None):
    """split the row block [lo, hi) into `pieces` nnz-balanced sub-blocks -> absolute row boundaries [pieces+1]"""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    local = rowptr[lo:hi + 1] - rowptr[lo]
    return balanced_bounds(local, pieces, weights) + lo


def all_piece_bounds(rowptr_host, world, pieces, weights=None):
    """[world, pieces+1] absolute row boundaries: rank blocks balanced by non-zeros, each cut into `pieces`
    (equal, or in the proportions `weights`)"""
    bounds = balanced_bounds(rowptr_host, world)
    return np.stack([piece_bounds(rowptr_host, int(bounds[g]), int(bounds[g + 1]), pieces, weights) for g in range(world)])


def device_piece_spmms(rowptr, col, val, n_cols, my_bounds, rowptr_host=None, strict=False):
    """One DeviceCSR (rectangular: rows of the piece x all columns) per local row piece, as `f(x_full, out)`
    callables for ShardedPropagator.  rowptr/col/val: the FULL normalised adjacency on this rank's device (only
    views of the local rows are kept alive); my_bounds: this rank's row of all_piece_bounds()."""
    from ..device import DeviceCSR, default_long_row_nnz
    if rowptr_host is None:
        rowptr_host = rowptr.cpu().numpy()
    fns, handles = [], []
    for p in range(len(my_bounds) - 1):
        r0, r1 = int(my_bounds[p]), int(my_bounds[p + 1])
        nb, ne = int(rowptr_host[r0]), int(rowptr_host[r1])
        rp_local = (rowptr[r0:r1 + 1] - rowptr[r0]).contiguous()
        h = DeviceCSR(rp_local, col[nb:ne].contiguous(), val[nb:ne].contiguous(), (r1 - r0, n_cols), strict=strict,
                      long_row_nnz=default_long_row_nnz(int(rowptr_host[-1])))      # cut long rows where the WHOLE matrix would
        handles.append(h)
        fns.append(lambda x, out, h=h: h.spmm(x, out=out))
    return fns, handles


def column_chunks(d, n_chunks=2):
    """Split the feature dimension into (up to) `n_chunks` column ranges for the software-pipelined exchange.  Stored as
    separate contiguous matrices, a chunk row should cover whole 128-byte lines (32 floats) so that the chunks together touch
    no more lines than the unsplit row: the d // 32 full lines are dealt out as evenly as possible (earlier chunks get the
    extra ones) and the d % 32 leftover columns ride with the last chunk (d = 100: 64 + 36 for two chunks, 32 + 32 + 36 for
    three -- 4 lines per gathered row either way).  More chunks than full lines: the leftover columns become a chunk of their
    own (d = 100, four chunks: 32 + 32 + 32 + 4)."""
    d, n_chunks = int(d), int(n_chunks)
    if n_chunks <= 1 or d <= 32:
        return [(0, d)]
    full, tail = divmod(d, 32)
    k = min(n_chunks, full)
    base, extra = divmod(full, k)
    widths = [(base + (1 if q < extra else 0)) * 32 for q in range(k)]
    if tail:
        if k < n_chunks:
            widths.append(tail)
        else:
            widths[-1] += tail
    out, c = [], 0
    for w in widths:
        out.append((c, c + w))
        c += w
    return out


def column_slices(d, parts, line=32):
    """Split d feature columns into `parts` contiguous slices (the column groups of a GridLayout).

    Slices are whole 128-byte lines (`line` floats) wherever the column count allows: d = 100 over 4 groups is
    32 + 32 + 32 + 4, not 4 x 25 -- every group still gathers one line per non-zero, but no group stores or exchanges
    7 pad floats per row (-22 % bytes on the wire for the grid layout).  With more parts than lines (d = 100 over 8)
    the split is simply even.  Slices may be empty when d < parts."""
    d, parts = int(d), int(parts)
    units = -(-d // line)
    if parts <= units:
        base, extra = divmod(units, parts)             # lines per part, the first `extra` parts get one more
        out, c = [], 0
        for q in range(parts):
            w = min(d - c, (base + (1 if q < extra else 0)) * line)
            out.append((c, c + w))
            c += w
        return out
    base, extra = divmod(d, parts)
    out, c = [], 0
    for q in range(parts):
        w = base + (1 if q < extra else 0)
        out.append((c, c + w))
        c += w
    return out


class GridLayout:
    """world = row_groups x col_groups ranks.  Rank g works on row block g % row_groups of column slice
    g // row_groups; the ranks of one column group exchange rows between hops, different column groups never talk
    (except as relays of each other's traffic)."""

    def __init__(self, world, row_groups):
        world, row_groups = int(world), int(row_groups)
        if row_groups < 1 or world % row_groups:
            raise ValueError("row_groups must divide the world size")
        self.world, self.row_groups, self.col_groups = world, row_groups, world // row_groups

    def coords(self, g):
        """(row block index, column group index) of global rank g"""
        return g % self.row_groups, g // self.row_groups

    def members(self, cg):
        """global ranks of column group cg, ordered by row block"""
        return [cg * self.row_groups + r for r in range(self.row_groups)]

    def __repr__(self):
        return f"GridLayout({self.row_groups} row blocks x {self.col_groups} column slices)"
