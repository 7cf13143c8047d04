"""Row-sharded multi-GPU pre-propagation: one process per GPU, adjacency rows partitioned across ranks,
the dense feature block all-gathered between hops over xGMI (RCCL through torch.distributed).

The reference has no multi-GPU propagation at all; this is new capability with the
same mathematical result: rank g owns the contiguous row block [b_g, b_{g+1}) of A_hat (balanced by non-zeros)
and of every hop matrix.  Per hop:

    Y_g = A_hat[b_g:b_{g+1}, :] @ X            local HIP SpMM, computed in `pieces` row pieces
    X'  = concat_g(Y_g)                        direct all-gather: every rank pushes each finished piece to its
                                               peers with grouped point-to-point send/recv (all xGMI links busy
                                               at once, no ring), while the next piece is still being computed

The last hop needs no exchange.  Aggregators are row-wise, so they run on the local shards with zero traffic.
Nothing here touches the data path on the host: buffers stay in HBM; torch.distributed is plumbing.

Grid layouts (GridLayout): SpMM is separable over feature columns, and 288 GB of HBM hold a replica of A_hat on every
GPU, so the G ranks can also be arranged as Gr row blocks x Gc column slices.  A rank then multiplies its row block of
A_hat with ITS column slice only and exchanges rows only inside its column group (Gr ranks):

    Gc = G (feature-sharded): every rank runs the whole k-hop chain on d/G columns -- no exchange at all;
    Gr = G (row-sharded):     the scheme above;
    in between:               in-bound bytes per rank per hop drop to (Gr-1)/Gr * N * d/Gc * 4.

xGMI is a point-to-point mesh, so an exchange inside a small column group would use only Gr-1 of a GPU's 7 links.  The
"relay" transport spreads it over all of them: each row piece is cut into G stripes; phase 1 sends stripe q to rank q,
phase 2 has q forward it to the ranks that need it (two link crossings per byte, but 7 links in parallel)."""
from .layout import (GridLayout, all_piece_bounds, balanced_bounds, column_chunks, column_slices, device_piece_spmms,
                     piece_bounds, tapered_weights)
from .propagator import ShardedPropagator
from .sharded_adj import (RowBlock, allgather_blocks, allgather_rows, balanced_bounds_device, block_piece_spmms, canonicalize_block,
                          exchange_checksums, gather_piece_bounds, local_piece_bounds, scatter_row_blocks)
from .halo import HaloPlan, HaloPropagator, halo_checksums
from .redistribute import exchange_var, fetch_rows, redistribute_rows, sharded_community_order
from .graph_op import ShardedGraphOp

__all__ = ["balanced_bounds", "piece_bounds", "all_piece_bounds", "tapered_weights", "device_piece_spmms", "column_chunks",
           "column_slices", "GridLayout", "ShardedPropagator", "ShardedGraphOp", "RowBlock", "scatter_row_blocks",
           "block_piece_spmms", "gather_piece_bounds", "local_piece_bounds", "allgather_blocks", "allgather_rows",
           "balanced_bounds_device", "exchange_checksums", "canonicalize_block", "HaloPlan", "HaloPropagator", "halo_checksums",
           "exchange_var", "fetch_rows", "redistribute_rows", "sharded_community_order"]
