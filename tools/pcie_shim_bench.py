import sys, time, numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, '.')
from sgl_amd import device as dev, synthetic
from sgl_amd.operators.utils import csr_sparse_dense_matmul
wl = synthetic.WORKLOADS["S1_products"]; n, d = wl["n"], wl["d"]
device = torch.device("cuda", 0)
a = synthetic.chung_lu_torch(n, wl["m"], wl["d_max"], seed=0, device=device)
rp, c, v = dev.normalize_adj(*a, n, 0.5, None)
adj = sp.csr_matrix((v.cpu().numpy(), c.cpu().numpy(), rp.cpu().numpy().astype(np.int32)), shape=(n, n))
x = np.random.default_rng(0).standard_normal((n, d)).astype(np.float32)
csr_sparse_dense_matmul(adj, x)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); y = csr_sparse_dense_matmul(adj, x); ts.append(time.perf_counter() - t0)
t = min(ts)
print(f"PCIE shim FloatCSRMulDenseOMP (host pointers, S1, one hop): {t*1e3:.1f} ms = {adj.nnz*d/t/1e9:.1f}e9 edge*feat/s (kernel alone 8.7 ms)")
