"""Oracle, second opinion: the five model_fn bodies restated op-for-op on PyTorch-CPU with autograd.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ and by bench.py's cpu_baseline leg.

Why it exists.  `oracle/models.py` is numpy with HAND-WRITTEN backward passes; the product's HIP kernels were
written by the same author, so an error of understanding could sit on both sides.  This module follows the
reference scripts literally, TF op by TF op (split / matmul(transpose_b) / reshape / transpose / conv1d for the CIN,
tensordot for the cross layer, tile / reshape for the DIN query, a real [B, R] one-hot times a [R, 1] kernel for
the first-order term when `literal=True`), and lets torch.autograd derive every gradient.  tests/test_oracle_torch.py
checks fp64 agreement of logits, loss, all gradients and one TF-1 Adam step with `oracle.models`; bench.py times
the fp32 DeepFM step of this module on all host cores as the CPU baseline SURVEY.md section 8(d) specifies
(TensorFlow itself cannot be installed here or on the GPU box).

Paths relative to /root/reference:
  fm        fm/fm.py:115-133            deepfm   deepfm/deepfm.py:73-113
  xdeepfm   xdeepfm/xdeepfm.py:123-196  dcn      dcn/dcn.py:117-153      din   din/din.py:83-140
Parameters are the dicts of `oracle.init` converted by `params_to_torch`.
"""
import numpy as np
import torch
import torch.nn.functional as Fn

BN_EPS = 1e-3


def params_to_torch(P, dtype=torch.float64, requires_grad=True):
    out = {}
    for k, v in P.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


# ------------------------------------------------------------------------------ feature_column.input_layer -----
def input_layer_embedding(tables, ids, row_off):
    """feature_column.input_layer over the 39 embedding_columns (fm/fm.py:118): one lookup per column in the
    name-sorted slot order, concatenated -> [B, F*D].  `ids` table-local [B,F] (int64 tensor)."""
    cols = []
    for f in range(ids.shape[1]):
        tab = tables[int(row_off[f]):int(row_off[f + 1])]
        cols.append(Fn.embedding(ids[:, f], tab))
    return torch.cat(cols, 1)


def input_layer_indicator(ids, row_off, dtype):
    """input_layer over indicator columns (fm/fm.py:117): the dense [B, R] multi-hot matrix TF really builds."""
    B, R = ids.shape[0], int(row_off[-1])
    x = torch.zeros(B, R, dtype=dtype)
    rows = ids + torch.as_tensor(np.asarray(row_off[:-1]), dtype=torch.int64)[None, :]
    x.scatter_(1, rows, 1.0)
    return x


def first_order(P, ids, row_off, literal):
    """tf.layers.dense(linear_net, 1, relu) on the one-hot input (fm/fm.py:120-121)."""
    w, b = P["w1"], P["b1"]
    if literal:
        x = input_layer_indicator(ids, row_off, w.dtype)
        return torch.relu(x @ w.reshape(-1, 1) + b)
    rows = ids + torch.as_tensor(np.asarray(row_off[:-1]), dtype=torch.int64)[None, :]
    return torch.relu(w[rows].sum(1, keepdim=True) + b)


def fm_second_order(emb, F, D):
    """fm/fm.py:123-129."""
    fm_net = emb.reshape(-1, F, D)
    sum_square = fm_net.sum(1) ** 2
    square_sum = (fm_net ** 2).sum(1)
    return 0.5 * (sum_square - square_sum).sum(1, keepdim=True)


def batch_norm(x, gamma, beta, train):
    """tf.layers.batch_normalization: batch moments in TRAIN; the never-updated moving stats (0, 1) otherwise
    (no model_fn runs UPDATE_OPS)."""
    if train:
        mean = x.mean(0, keepdim=True)
        var = ((x - mean) ** 2).mean(0, keepdim=True)
    else:
        mean, var = torch.zeros_like(x[:1]), torch.ones_like(x[:1])
    return (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta


def dropout(x, rate, mask, train):
    if not train or mask is None or rate == 0.0:
        return x
    return x * torch.as_tensor(mask, dtype=x.dtype) * (1.0 / (1.0 - rate))


def dnn_tower(P, pre, x, n_layers, rate, masks, train):
    """dense(relu) -> batch_normalization -> dropout per layer (deepfm/deepfm.py:103-107)."""
    for i in range(n_layers):
        x = torch.relu(x @ P[f"{pre}.W{i}"] + P[f"{pre}.b{i}"])
        x = batch_norm(x, P[f"{pre}.gamma{i}"], P[f"{pre}.beta{i}"], train)
        x = dropout(x, rate, None if masks is None else masks[i], train)
    return x


# ----------------------------------------------------------------------------------------------- models --------
def fm_logits(P, ids, row_off, literal=False):
    F, D = ids.shape[1], P["tables"].shape[1]
    emb = input_layer_embedding(P["tables"], ids, row_off)
    y1 = first_order(P, ids, row_off, literal)
    y2 = fm_second_order(emb, F, D)
    logits = torch.cat([y1, y2], -1) @ P["out.W"] + P["out.b"]
    return logits.reshape(-1)


def deepfm_logits(P, ids, row_off, n_layers=2, rate=0.5, masks=None, train=True, literal=False):
    F, D = ids.shape[1], P["tables"].shape[1]
    emb = input_layer_embedding(P["tables"], ids, row_off)
    y1 = first_order(P, ids, row_off, literal)
    y2 = fm_second_order(emb, F, D)
    dnn = dnn_tower(P, "dnn", emb.reshape(-1, F * D), n_layers, rate, masks, train)
    y_dnn = torch.relu(dnn @ P["dnn.Wout"] + P["dnn.bout"])
    logits = torch.cat([y1, y2, y_dnn], -1) @ P["out.W"] + P["out.b"]
    return logits.reshape(-1)


def dcn_logits(P, ids, row_off, n_layers=2, rate=0.5, masks=None, train=True):
    """dcn/dcn.py:117-153: xw = tensordot(reshape(xl, [-1,1,dim]), w, 1); xl = xw * x0 + xl + b."""
    x0 = input_layer_embedding(P["tables"], ids, row_off)
    dim = x0.shape[1]
    xl = x0
    for i in range(P["cross.W"].shape[0]):
        xw = torch.tensordot(xl.reshape(-1, 1, dim), P["cross.W"][i], dims=1)      # [B,1]
        xl = xw * x0 + xl + P["cross.b"][i]
    dnn = dnn_tower(P, "dnn", x0, n_layers, rate, masks, train)
    logits = torch.cat([dnn, xl], -1) @ P["out.W"] + P["out.b"]
    return logits.reshape(-1)


def cin_layer_literal(X0, Xk, W, c):
    """One CIN layer through the reference's own tensor ops (xdeepfm/xdeepfm.py:143-169)."""
    D = X0.shape[2]
    split0 = torch.stack(torch.split(X0, 1, 2), 0)             # tf.split(..., D*[1], 2): D x [B,F,1]
    splitk = torch.stack(torch.split(Xk, 1, 2), 0)             # D x [B,H,1]
    dot_m = split0 @ splitk.transpose(-1, -2)                  # tf.matmul(transpose_b=True): [D,B,F,H]
    dot_o = dot_m.reshape(D, -1, X0.shape[1] * Xk.shape[1])    # [D,B,F*H]
    dot = dot_o.permute(1, 0, 2)                               # [B,D,F*H]
    out = dot @ W + c                                          # conv1d with a width-1 filter + bias_add
    return torch.relu(out).permute(0, 2, 1)                    # [B,N,D]


def xdeepfm_logits(P, ids, logx, row_off, cat_slot, cat_off, cin=(128, 128), n_layers=2, rate=0.5, masks=None,
                   train=True):
    """xdeepfm/xdeepfm.py:123-196.  Linear columns = 13 numeric log-values + 26 indicator blocks, concatenated in
    name-sorted order; the dense(1) kernel is stored split as lin.wnum [13] / lin.wcat [Rc] (oracle.init layout)."""
    F, D = ids.shape[1], P["tables"].shape[1]
    crow = ids[:, torch.as_tensor(cat_slot)] + torch.as_tensor(np.asarray(cat_off[:-1]), dtype=torch.int64)[None, :]
    lin = (logx * P["lin.wnum"]).sum(1, keepdim=True) + P["lin.wcat"][crow].sum(1, keepdim=True) + P["lin.b"]
    linear_y = torch.relu(lin)
    X0 = input_layer_embedding(P["tables"], ids, row_off).reshape(-1, F, D)
    hidden, final = [X0], []
    for k in range(len(cin)):
        nxt = cin_layer_literal(X0, hidden[-1], P[f"cin.W{k}"], P[f"cin.c{k}"])
        final.append(nxt)
        hidden.append(nxt)
    result = torch.cat(final, 1).sum(-1)
    cin_y = torch.relu(result @ P["cin.Wout"] + P["cin.bout"])
    emb2 = input_layer_embedding(P["tables2"], ids, row_off)   # the second input_layer call owns new variables
    dnn = dnn_tower(P, "dnn", emb2.reshape(-1, F * D), n_layers, rate, masks, train)
    dnn_y = torch.relu(dnn @ P["dnn.Wout"] + P["dnn.bout"])
    logits = torch.cat([linear_y, cin_y, dnn_y], -1) @ P["out.W"] + P["out.b"]
    return logits.reshape(-1)


def din_attention(P, pre, table, hist, item_emb, rate, masks, train):
    """din/din.py:103-125."""
    K = table.shape[1]
    dense_emb = Fn.embedding(hist, table)                                   # [B,P,K]
    dense_mask = (hist > 0).to(table.dtype).unsqueeze(-1)
    padded = hist.shape[1]
    hist_emb = dense_emb.reshape(-1, K)
    query_emb = item_emb.repeat(1, padded).reshape(-1, K)                   # tf.tile(item_emb, [1, P])
    att = torch.cat([hist_emb, query_emb, hist_emb * query_emb, hist_emb - query_emb], 1)
    for i in range(2):
        att = torch.relu(att @ P[f"{pre}.W{i}"] + P[f"{pre}.b{i}"])
        att = dropout(att, rate, None if masks is None else masks[i], train)
    wgt = (att @ P[f"{pre}.W2"] + P[f"{pre}.b2"]).reshape(-1, padded, 1)
    return ((dense_emb * wgt) * dense_mask).sum(1)


def din_logits(P, i_id, i_cate, hist_i, hist_c, rate=0.5, masks=None, train=True):
    mk = masks or {}
    i_b = P["item_bias"][i_id]
    pkg = Fn.embedding(i_id, P["item_emb"])
    pkgc = Fn.embedding(i_cate, P["cate_emb"])
    h_i = din_attention(P, "att_i", P["item_emb"], hist_i, pkg, rate, mk.get("att_i"), train)
    h_c = din_attention(P, "att_c", P["cate_emb"], hist_c, pkgc, rate, mk.get("att_c"), train)
    net = torch.cat([pkg, h_i, h_c], 1)
    for i in range(3):
        net = torch.relu(net @ P[f"mlp.W{i}"] + P[f"mlp.b{i}"])
        net = dropout(net, rate, mk["mlp"][i] if "mlp" in mk else None, train)
    return (net @ P["mlp.Wout"] + P["mlp.bout"]).reshape(-1) + i_b


# ----------------------------------------------------------------------------------------- loss + optimizer ----
def sigmoid_ce_mean(logits, labels):
    """tf.reduce_mean(tf.nn.sigmoid_cross_entropy_with_logits) (fm/fm.py:146-149)."""
    return Fn.binary_cross_entropy_with_logits(logits, labels.to(logits.dtype), reduction="mean")


def loss_and_grads(logits_fn, P, labels):
    """-> (loss, logits, {name: dense gradient})."""
    for t in P.values():
        t.grad = None
    z = logits_fn(P)
    loss = sigmoid_ce_mean(z, labels)
    loss.backward()
    return loss.detach(), z.detach(), {k: (t.grad if t.grad is not None else torch.zeros_like(t)) for k, t in P.items()}


class AdamTF1:
    """tf.train.AdamOptimizer, TF 1.13: dense ApplyAdam; IndexedSlices through the NON-lazy
    `_apply_sparse_shared` (whole-variable decay, touched rows get the gradient)."""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.slots = {}

    def step(self, P, grads):
        """`grads[name]` dense.  For the variables TF differentiates to IndexedSlices the dense gradient is zero off
        the touched rows, where both TF formulas reduce to the same expression (the identity the product's pull-based
        sweep relies on) -- so one dense formula serves both here."""
        self.t += 1
        a = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        with torch.no_grad():
            for k, var in P.items():
                g = grads[k]
                if k not in self.slots:
                    self.slots[k] = (torch.zeros_like(var), torch.zeros_like(var))
                m, v = self.slots[k]
                m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
                var.sub_(a * m / (v.sqrt() + self.eps))


# ------------------------------------------------------------------------- the timed CPU baseline (bench.py) ---
class DeepFMCpuBaseline:
    """DeepFM bs-256 TRAIN step on PyTorch-CPU fp32, all cores (SURVEY.md section 8d stand-in for "the reference
    TF CPU path").  Two variants of the first-order term and of the table gradient:
      efficient -- gathers; the table / first-order gradients stay sparse (rows + values) and the TF-1 non-lazy
                   Adam decays the whole variable in place, then index_adds the touched rows;
      literal   -- what the TF graph really does: a dense [B, 840 646] one-hot input_layer times the [R,1] kernel
                   (fm/fm.py:117,121) with a dense kernel gradient, embedding gradients as IndexedSlices.
    Both run the same whole-variable optimizer sweep (345 MB of state traffic per step)."""

    def __init__(self, P_np, row_off, n_layers=2, rate=0.5, literal=False, lr=1e-3):
        self.row_off = np.asarray(row_off)
        self.off = torch.as_tensor(self.row_off[:-1], dtype=torch.int64)[None, :]
        self.n, self.rate, self.literal = n_layers, rate, literal
        self.P = params_to_torch(P_np, torch.float32)
        self.sparse = ("tables", "w1")
        self.slots = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in self.P.items()}
        self.lr, self.b1, self.b2, self.eps, self.t = lr, 0.9, 0.999, 1e-8, 0
        self.gen = torch.Generator().manual_seed(0)

    def step(self, ids_np, labels_np):
        P = self.P
        ids = torch.as_tensor(ids_np, dtype=torch.int64)
        y = torch.as_tensor(labels_np, dtype=torch.float32).reshape(-1)
        rows = ids + self.off
        B, F = rows.shape
        D = P["tables"].shape[1]
        for t in P.values():
            t.grad = None
        emb_rows = P["tables"].detach()[rows].requires_grad_(True)          # [B,F,D] gather
        if self.literal:
            x = torch.zeros(B, int(self.row_off[-1]))
            x.scatter_(1, rows, 1.0)
            y1 = torch.relu(x @ P["w1"].reshape(-1, 1) + P["b1"])
            w1_rows = None
        else:
            w1_rows = P["w1"].detach()[rows].requires_grad_(True)
            y1 = torch.relu(w1_rows.sum(1, keepdim=True) + P["b1"])
        y2 = fm_second_order(emb_rows, F, D)
        h = emb_rows.reshape(B, F * D)
        for i in range(self.n):
            h = torch.relu(h @ P[f"dnn.W{i}"] + P[f"dnn.b{i}"])
            h = batch_norm(h, P[f"dnn.gamma{i}"], P[f"dnn.beta{i}"], True)
            keep = (torch.rand(h.shape, generator=self.gen) >= self.rate).to(h.dtype)
            h = h * keep * (1.0 / (1.0 - self.rate))
        y_dnn = torch.relu(h @ P["dnn.Wout"] + P["dnn.bout"])
        logits = (torch.cat([y1, y2, y_dnn], -1) @ P["out.W"] + P["out.b"]).reshape(-1)
        loss = sigmoid_ce_mean(logits, y)
        loss.backward()
        # ---- TF-1 Adam over EVERY variable -------------------------------------------------------------------
        self.t += 1
        a = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        flat = rows.reshape(-1)
        with torch.no_grad():
            for k, var in P.items():
                m, v = self.slots[k]
                if k == "tables" or (k == "w1" and not self.literal):
                    g_rows = emb_rows.grad.reshape(B * F, D) if k == "tables" else w1_rows.grad.reshape(B * F)
                    uniq, inv = torch.unique(flat, return_inverse=True)      # IndexedSlices: duplicates summed first
                    G = torch.zeros((uniq.shape[0],) + g_rows.shape[1:]).index_add_(0, inv, g_rows)
                    m.mul_(self.b1).index_add_(0, uniq, G, alpha=1.0 - self.b1)
                    v.mul_(self.b2).index_add_(0, uniq, G * G, alpha=1.0 - self.b2)
                else:
                    g = var.grad
                    m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
                    v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
                var.addcdiv_(m, v.sqrt().add_(self.eps), value=-a)
        return float(loss.detach())
