"""Label use / label reuse on the device.  Reference: sgl/tasks/node_classification_with_label_use.py:58-137 (the loop around
`model.preprocess`) and sgl/tasks/utils.py:33-36 (`add_labels`).

The reference rebuilds `[x || one-hot(labels of a random part of the training nodes)]` on the host every epoch, pre-propagates it
(`prop_steps` SpMMs over [N, d + C]) and -- from `reuse_start_epoch` on -- `label_iters` more times after writing the model's
own soft predictions into the label columns of the unlabeled nodes:

    for _ in range(label_iters):
        pred = model.model_forward(full_idx, device).detach().cpu()
        features[unlabeled_idx, -C:] = softmax(pred[unlabeled_idx])
        model.preprocess(adj, features)

Here the feature matrix, the hop matrices, the predictions and the write-back all stay in HBM: one [N, d + C] float32 device
matrix in the line-aware row pitch the SpMM gathers from (so `preprocess` takes it as it is, no re-packing), rows gathered by
`sgl_gather_rows_f32` inside `model_forward`, softmax + indexed write-back as device tensor ops.  (On Linux the reference loop
cannot run as shipped: `add_labels` returns float64 and its ctypes SpMM accepts float32 only -- tests/golden/make_goldens.py,
gen_g10 -- the arithmetic it describes is what is reproduced here, in float32 like the SpMM.)"""
import torch
import torch.nn.functional as F

from .. import device as dev


def add_labels(x, labels, idx, num_classes, out=None, device="cuda"):
    """[x || one-hot(labels[idx])] as ONE float32 device matrix [N, d + C] (sgl/tasks/utils.py:33-36 builds it on the host).
    `out`: a matrix of that shape from a previous epoch to be refilled in place (its feature columns are kept, only the label
    columns are rewritten -- the per-epoch cost is then C columns, not d + C)."""
    device = torch.device(device)
    n, d = x.shape
    labels = torch.as_tensor(labels, device=device).reshape(-1).long()
    idx = torch.as_tensor(idx, device=device).reshape(-1).long()
    if out is None:
        out = dev.alloc_rows(n, d + num_classes, device)
        out[:, :d] = torch.as_tensor(x, dtype=torch.float32, device=device) if not torch.is_tensor(x) else x.to(device=device, dtype=torch.float32)
    elif tuple(out.shape) != (n, d + num_classes):
        raise ValueError(f"out must be [{n}, {d + num_classes}]")
    out[:, d:] = 0
    if idx.numel():
        out[idx, d + labels[idx]] = 1.0
    return out


@torch.no_grad()
def predict_all(model, n, device, batch_size=None):
    """model.model_forward over all n nodes (in batches of `batch_size`, the reference's label_reuse_batch_size) -> [n, C] on device"""
    device = torch.device(device)
    # contiguous row ranges: the hop rows are views of the hop matrices, nothing is gathered (models/base_model.py take_rows)
    if batch_size is None or batch_size >= n:
        return model.model_forward(range(n), device)
    return torch.cat([model.model_forward(range(s, min(s + batch_size, n)), device) for s in range(0, n, batch_size)])


@torch.no_grad()
def label_reuse(model, adj, features, unlabeled_idx, num_classes, label_iters, device="cuda", batch_size=None):
    """The label-reuse iterations of one epoch (node_classification_with_label_use.py:88-104), entirely on the device.
    `features` ([N, d + C] float32 device matrix, e.g. from add_labels) is updated IN PLACE like the reference's array and
    `model.preprocess(adj, features)` is re-run after every iteration.  Returns `features`."""
    device = torch.device(device)
    unlabeled_idx = torch.as_tensor(unlabeled_idx, device=device).reshape(-1).long()
    n = features.shape[0]
    for _ in range(int(label_iters)):
        pred = predict_all(model, n, device, batch_size)
        features[unlabeled_idx, -num_classes:] = F.softmax(pred[unlabeled_idx], dim=-1)
        model.preprocess(adj, features)
    return features
