"""TEST INFRASTRUCTURE ONLY — oracle of MF's minibatch/optimiser path (`MF(backend="pytorch")`,
cornac/models/mf/backend_pt.py:29-106, driven from cornac/models/mf/recom_mf.py:211-252).

The arithmetic lives in PyTorch (torch.optim.{SGD, Adam, RMSprop, Adagrad} with weight_decay over dense
nn.Embedding gradients).  torch (CPU) is on every box this repo runs on, so this oracle restates the step
with explicit tensors and the same torch optimiser classes; it is pinned against the live reference
(tests/test_oracle_vs_reference.py::test_mf_minibatch_oracle_matches_live_reference) and against goldens the
real reference produced (tests/golden/mf_minibatch.npz).  dropout = 0 only (dropout would consume torch RNG).
"""
import numpy as np


def fit(U, V, Bu, Bi, mu, rid, cid, val, batches, optimizer="sgd", lr=0.01, reg=0.02, use_bias=True):
    """batches: iterable of index arrays into (rid, cid, val).  Returns (U, V, Bu, Bi, per-batch losses)."""
    import torch

    make = {"sgd": torch.optim.SGD, "adam": torch.optim.Adam, "rmsprop": torch.optim.RMSprop,
            "adagrad": torch.optim.Adagrad}[optimizer]
    P = [torch.tensor(np.array(U, np.float32), requires_grad=True), torch.tensor(np.array(V, np.float32), requires_grad=True)]
    if use_bias:
        P += [torch.tensor(np.array(Bu, np.float32).reshape(-1, 1), requires_grad=True),
              torch.tensor(np.array(Bi, np.float32).reshape(-1, 1), requires_grad=True)]
    opt = make(P, lr=lr, weight_decay=reg)
    rid_t, cid_t = torch.as_tensor(np.asarray(rid, np.int64)), torch.as_tensor(np.asarray(cid, np.int64))
    val_t = torch.as_tensor(np.asarray(val, np.float32))
    losses = []
    for ids in batches:
        ids = torch.as_tensor(np.asarray(ids, np.int64))
        u, i, r = rid_t[ids], cid_t[ids], val_t[ids]
        pred = (P[0][u] * P[1][i]).sum(dim=1, keepdim=True)
        if use_bias:
            pred = pred + P[2][u] + P[3][i] + float(mu)
        loss = ((pred.squeeze(1) - r) ** 2).sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    out = [p.detach().numpy() for p in P]
    if not use_bias:
        out += [np.array(Bu, np.float32), np.array(Bi, np.float32)]
    return out[0], out[1], out[2].reshape(-1), out[3].reshape(-1), losses
