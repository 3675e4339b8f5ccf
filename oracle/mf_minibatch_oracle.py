"""TEST INFRASTRUCTURE ONLY — oracle of MF's minibatch/optimiser path (`MF(backend="pytorch")`,
cornac/models/mf/backend_pt.py:29-106, driven from cornac/models/mf/recom_mf.py:211-252).

The arithmetic lives in PyTorch (torch.optim.{SGD, Adam, RMSprop, Adagrad} with weight_decay over dense
nn.Embedding gradients).  torch (CPU) is on every box this repo runs on, so this oracle restates the step
with explicit tensors and the same torch optimiser classes; it is pinned against the live reference
(tests/test_oracle_vs_reference.py::test_mf_minibatch_oracle_matches_live_reference) and against goldens the
real reference produced (tests/golden/mf_minibatch.npz).

Dropout (backend_pt.py:42,59: `nn.Dropout(p)` on the gathered user and item rows): the keep masks are an INPUT here
(`keep`: per batch a pair of 0/1 arrays [B, k]) — `dropout_masks` below restates how the reference's run draws them from
torch's CPU generator (torch.manual_seed(seed), the four nn.Embedding initialisations that consume it first, then per batch
one bernoulli_(1 - p) tensor for the user rows and one for the item rows, scaled by 1 / (1 - p) in float32), pinned against
the live reference with dropout > 0 (tests/test_oracle_vs_reference.py)."""
import numpy as np


def dropout_masks(p, seed, n_users, n_items, k, use_bias, batch_sizes):
    """the keep masks of a reference run on the CPU (recom_mf.py:221-222 seeds torch; backend_pt.py:45-53 builds two — with
    biases four — nn.Embedding tables whose normal_ initialisation consumes the generator before their weights are
    replaced; backend_pt.py:59 then draws the user rows' mask and the item rows' mask of every batch).  Returns a list of
    (keep_u, keep_i) uint8 arrays [B, k] and the float32 scale 1 / (1 - p)."""
    import torch

    if seed is not None:
        torch.manual_seed(seed)
    for shape in [(n_users, k), (n_items, k)] + ([(n_users, 1), (n_items, 1)] if use_bias else []):
        torch.empty(shape).normal_()
    out = []
    for b in batch_sizes:
        ku = torch.empty(int(b), k).bernoulli_(1.0 - p)
        ki = torch.empty(int(b), k).bernoulli_(1.0 - p)
        out.append((ku.numpy().astype(np.uint8), ki.numpy().astype(np.uint8)))
    return out, np.float32(1.0) / np.float32(1.0 - p)


def fit(U, V, Bu, Bi, mu, rid, cid, val, batches, optimizer="sgd", lr=0.01, reg=0.02, use_bias=True, keep=None,
        keep_scale=1.0):
    """batches: iterable of index arrays into (rid, cid, val).  keep: per batch (keep_u, keep_i) 0/1 arrays [B, k] of the
    dropout on the gathered rows, kept entries scaled by keep_scale; None = no dropout.  Returns (U, V, Bu, Bi, per-batch
    losses)."""
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
    for b, ids in enumerate(batches):
        ids = torch.as_tensor(np.asarray(ids, np.int64))
        u, i, r = rid_t[ids], cid_t[ids], val_t[ids]
        ue, ie = P[0][u], P[1][i]
        if keep is not None:
            ue = ue * (torch.as_tensor(np.asarray(keep[b][0], np.float32)) * float(keep_scale))
            ie = ie * (torch.as_tensor(np.asarray(keep[b][1], np.float32)) * float(keep_scale))
        pred = (ue * ie).sum(dim=1, keepdim=True)
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
