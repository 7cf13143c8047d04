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
print(f"PCIE shim FloatCSRMulDenseOMP (host pointers, S1, one hop, adjacency cached from the previous call): {t*1e3:.1f} ms = "
      f"{adj.nnz*d/t/1e9:.1f}e9 edge*feat/s (kernel alone 8.7 ms; first call incl. upload of A_hat and plan: see below)")
import ctypes
from sgl_amd import _lib
adj2 = adj.copy()
t0 = time.perf_counter(); csr_sparse_dense_matmul(adj2, x); t_first = time.perf_counter() - t0
h, m = ctypes.c_int64(0), ctypes.c_int64(0)
_lib.lib().sgl_shim_cache_stats(ctypes.byref(h), ctypes.byref(m))
print(f"PCIE shim first call on a new matrix (upload 1.5 GB + plan): {t_first*1e3:.1f} ms; cache hits {h.value}, misses {m.value}")

# the C symbol alone (what libsgl_hip.so is responsible for): the wrapper above it mirrors the reference's utils.py:31-35,
# whose temporaries (a float64 zeros matrix, astype / flatten copies) cost host time of their own
ip32, ix32, dv = adj.indptr.astype(np.int32), adj.indices.astype(np.int32), adj.data.astype(np.float32)
ans = np.zeros(n * d, dtype=np.float32)
xm = x.reshape(-1).copy()
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
lib = _lib.lib()
lib.FloatCSRMulDenseOMP(p(ans), p(dv), p(ix32), p(ip32), p(xm), n, d)
ts = []
for _ in range(3):
    ans[:] = 0
    t0 = time.perf_counter(); lib.FloatCSRMulDenseOMP(p(ans), p(dv), p(ix32), p(ip32), p(xm), n, d); ts.append(time.perf_counter() - t0)
print(f"PCIE shim C symbol alone, cached adjacency: {min(ts)*1e3:.1f} ms per hop (content hashes of 1.5 GB + 0.98 GB up + zero check + kernel + 0.98 GB down)")
