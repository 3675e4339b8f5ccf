"""CPU restatement of the reference's WMF training step — TEST INFRASTRUCTURE ONLY.

PARITY: pinned against the reference's OWN WMF code, not against TensorFlow.  The reference computes this path inside
TensorFlow (`tensorflow==2.12.0`, cornac/models/wmf/requirements.txt), which is absent from this image, and no
reference test touches WMF.  What IS checked (tests/test_oracle_vs_reference.py::test_wmf_oracle_and_host_class_match_the_reference_wmf_code,
tests/golden/wmf_ref.npz): cornac/models/wmf/recom_wmf.py + wmf.py run unmodified over oracle/tf1_shim — torch forward
and autograd of the loss THEIR code builds, their xavier initialisation, their item_iter shuffling and batch_C — and
this restatement (hand-derived gradients, loop, Adam) reproduces the result to 5e-6.  What stays restated on both
sides, from TensorFlow's published source rather than by running it: the IndexedSlices gradient of tf.gather,
clip_by_value on it, and tf.train.AdamOptimizer's dense / sparse update rules (listed in the shim's header).

Follows:
  * cornac/models/wmf/wmf.py:34-55        graph: P = U V_b^T, loss = sum(C (R - P)^2) + lambda_u l2(U) + lambda_v l2(V_b)
                                          (tf.nn.l2_loss(x) = sum(x^2) / 2), gradients clipped to [-5, 5]
  * cornac/models/wmf/recom_wmf.py:160-207  one `sess.run(opt)` per batch of items from `item_iter(shuffle=True)`;
                                          C = b everywhere, a where the batch's rating matrix is non-zero
  * TF1 Adam (beta1 .9, beta2 .999, eps 1e-8): lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t);
    m += (1-beta1)(g - m); v += (1-beta2)(g^2 - v); var -= lr_t m / (sqrt(v) + eps).
    `V` receives an IndexedSlices gradient (tf.gather), for which TF1 Adam decays m, v of ALL rows, adds the
    slice contribution to the gathered rows and then updates ALL rows (`_apply_sparse_shared`).
All arithmetic in float32 like the graph's dtype.
"""
import numpy as np

f32 = np.float32


class WmfOracle:
    def __init__(self, U, V, csc, lambda_u=0.01, lambda_v=0.01, a=1.0, b=0.01, lr=0.001,
                 beta1=0.9, beta2=0.999, eps=1e-8):
        self.U = np.array(U, dtype=f32)
        self.V = np.array(V, dtype=f32)
        self.R = csc.tocsc()
        self.lu, self.lv, self.a, self.b, self.lr = f32(lambda_u), f32(lambda_v), f32(a), f32(b), float(lr)
        self.b1, self.b2, self.eps = beta1, beta2, f32(eps)
        self.mU = np.zeros_like(self.U); self.vU = np.zeros_like(self.U)
        self.mV = np.zeros_like(self.V); self.vV = np.zeros_like(self.V)
        self.t = 0

    def step(self, item_ids):
        ids = np.asarray(item_ids, dtype=np.int64)
        self.t += 1
        b1, b2 = f32(self.b1), f32(self.b2)
        Rb = np.asarray(self.R[:, ids].toarray(), dtype=f32)
        C = np.where(Rb != 0, self.a, self.b).astype(f32)
        Vb = self.V[ids]
        P = self.U @ Vb.T
        E = Rb - P
        loss = float(np.sum(C * E * E, dtype=np.float64) + 0.5 * self.lu * np.sum(self.U.astype(np.float64) ** 2)
                     + 0.5 * self.lv * np.sum(Vb.astype(np.float64) ** 2))
        D = f32(-2.0) * C * E
        gU = np.clip(D @ Vb + self.lu * self.U, f32(-5), f32(5)).astype(f32)
        gV = np.clip(D.T @ self.U + self.lv * Vb, f32(-5), f32(5)).astype(f32)
        lr_t = f32(self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t))
        self.mU += (f32(1) - b1) * (gU - self.mU)
        self.vU += (f32(1) - b2) * (gU * gU - self.vU)
        self.U -= lr_t * self.mU / (np.sqrt(self.vU) + self.eps)
        self.mV *= b1
        self.vV *= b2
        self.mV[ids] += (f32(1) - b1) * gV
        self.vV[ids] += (f32(1) - b2) * (gV * gV)
        self.V -= lr_t * self.mV / (np.sqrt(self.vV) + self.eps)
        return loss

    def fit_batches(self, batches):
        return [self.step(ids) for ids in batches]
