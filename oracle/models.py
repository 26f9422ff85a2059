"""Oracle: numpy forward + hand-written backward of the five model_fn bodies.

Follows (paths relative to /root/reference):
  FM       fm/fm.py:115-133
  DeepFM   deepfm/deepfm.py:73-113 married to fm.py's Criteo columns
  xDeepFM  xdeepfm/xdeepfm.py:123-196
  DCN      dcn/dcn.py:117-153
  DIN      din/din.py:83-140
read together with SURVEY.md Appendix A (TF-1.x semantics).  Parameters are plain
dicts of numpy arrays; `ids` are table-local int ids [B,F] in TF's name-sorted slot
order and `row_off` the per-slot row offsets into the concatenated table.

Each model returns logits [B] and keeps a cache; `backward(dz)` returns
(dense_grads: dict name->array, sparse_grads: dict name->(pair_rows [N], pair_vals [N,...]))
where sparse pairs are listed in ascending (b, f) order *per field-major* layout
p = f*B + b, the order the product's sorted segment-sum uses.

Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np

from . import nn


def _pairs_field_major(rows_bf, vals_bf):
    """rows [B,F], vals [B,F,...] -> field-major flattened pairs (p = f*B + b)."""
    rows = np.ascontiguousarray(rows_bf.T).reshape(-1)
    vals = np.ascontiguousarray(np.swapaxes(vals_bf, 0, 1)).reshape((rows.shape[0],) + vals_bf.shape[2:])
    return rows, vals


# ----------------------------------------------------------------- DNN tower --
def tower_fwd(x, P, pre, n_layers, train, rate, masks):
    """[dense(relu) -> BN -> dropout] * n  (deepfm/deepfm.py:103-107)."""
    caches = []
    h = x
    for i in range(n_layers):
        a = nn.dense_fwd(h, P[f"{pre}.W{i}"], P[f"{pre}.b{i}"], relu=True)
        if train:
            y, bc = nn.bn_train_fwd(a, P[f"{pre}.gamma{i}"], P[f"{pre}.beta{i}"])
        else:
            y, bc = nn.bn_eval_fwd(a, P[f"{pre}.gamma{i}"], P[f"{pre}.beta{i}"]), None
        m = masks[i] if (train and masks is not None) else None
        o = nn.dropout_fwd(y, rate, m)
        caches.append((h, a, bc, m))
        h = o
    return h, caches


def tower_bwd(dh, P, pre, caches, rate, grads):
    for i in reversed(range(len(caches))):
        h_in, a, bc, m = caches[i]
        dy = nn.dropout_bwd(dh, rate, m)
        da, dgam, dbet = nn.bn_train_bwd(bc, dy)
        dh, dW, db = nn.dense_bwd(h_in, P[f"{pre}.W{i}"], a, da, relu=True)
        grads[f"{pre}.W{i}"], grads[f"{pre}.b{i}"] = dW, db
        grads[f"{pre}.gamma{i}"], grads[f"{pre}.beta{i}"] = dgam, dbet
    return dh


# ------------------------------------------------------------ embedding + FM --
def gather_fm_fwd(tables, w1, ids, row_off):
    """input_layer(embedding cols) + first-order one-hot matmul + FM second order.
    fm/fm.py:117-129.  Returns rows [B,F], E [B,F,D], y1pre [B] (no bias), S [B,D], y2 [B]."""
    rows = ids.astype(np.int64) + row_off[None, :-1]
    E = tables[rows]
    dt = tables.dtype
    y1 = np.zeros(ids.shape[0], dt)
    if w1 is not None:
        for f in range(ids.shape[1]):          # ascending column order, like the one-hot matmul
            y1 = y1 + w1[rows[:, f]]
    S = np.zeros((ids.shape[0], tables.shape[1]), dt)
    Q = np.zeros_like(S)
    for f in range(ids.shape[1]):
        S = S + E[:, f]
        Q = Q + E[:, f] * E[:, f]
    y2 = dt.type(0.5) * (S * S - Q).sum(1, dtype=dt)
    return rows, E, y1, S, y2


def fm2_bwd(E, S, g2):
    """d y2 / dE as TF autodiff orders it: g*S - g*E  (fm/fm.py:127-129)."""
    return g2[:, None, None] * S[:, None, :] - g2[:, None, None] * E


class FM:
    """fm/fm.py:115-133. logits = dense([relu(first_order), fm2], 1)."""

    def __init__(self, P, row_off):
        self.P, self.row_off = P, row_off

    def forward(self, ids, train=True, masks=None):
        P = self.P
        rows, E, y1p, S, y2 = gather_fm_fwd(P["tables"], P["w1"], ids, self.row_off)
        y1 = np.maximum(y1p + P["b1"][0], 0)
        cat = np.stack([y1, y2], 1)
        z = (cat @ P["out.W"] + P["out.b"]).reshape(-1)
        self.c = (rows, E, S, y1, cat)
        return z

    def backward(self, dz):
        P = self.P
        rows, E, S, y1, cat = self.c
        g, s = {}, {}
        dcat = dz[:, None] @ P["out.W"].T
        g["out.W"], g["out.b"] = cat.T @ dz[:, None], dz.sum(keepdims=True)
        d1 = dcat[:, 0] * (y1 > 0)
        g["b1"] = d1.sum(keepdims=True)
        dE = fm2_bwd(E, S, dcat[:, 1])
        s["tables"] = _pairs_field_major(rows, dE)
        s["w1"] = _pairs_field_major(rows, np.repeat(d1[:, None], rows.shape[1], 1))
        return g, s


class DeepFM:
    """deepfm/deepfm.py:73-113 on the Criteo columns of fm/fm.py:47-97."""

    def __init__(self, P, row_off, n_layers=2, dropout=0.5):
        self.P, self.row_off, self.n, self.rate = P, row_off, n_layers, dropout

    def forward(self, ids, train=True, masks=None):
        P = self.P
        rows, E, y1p, S, y2 = gather_fm_fwd(P["tables"], P["w1"], ids, self.row_off)
        y1 = np.maximum(y1p + P["b1"][0], 0)
        x = E.reshape(E.shape[0], -1)
        h, tc = tower_fwd(x, P, "dnn", self.n, train, self.rate, masks)
        yd = nn.dense_fwd(h, P["dnn.Wout"], P["dnn.bout"], relu=True)
        cat = np.concatenate([y1[:, None], y2[:, None], yd], 1)
        z = (cat @ P["out.W"] + P["out.b"]).reshape(-1)
        self.c = (rows, E, S, y1, h, tc, yd, cat)
        return z

    def backward(self, dz):
        P = self.P
        rows, E, S, y1, h, tc, yd, cat = self.c
        g, s = {}, {}
        dcat = dz[:, None] @ P["out.W"].T
        g["out.W"], g["out.b"] = cat.T @ dz[:, None], dz.sum(keepdims=True)
        d1 = dcat[:, 0] * (y1 > 0)
        g["b1"] = d1.sum(keepdims=True)
        dh, g["dnn.Wout"], g["dnn.bout"] = nn.dense_bwd(h, P["dnn.Wout"], yd, dcat[:, 2:3], relu=True)
        dx = tower_bwd(dh, P, "dnn", tc, self.rate, g)
        dE = fm2_bwd(E, S, dcat[:, 1]) + dx.reshape(E.shape)
        s["tables"] = _pairs_field_major(rows, dE)
        s["w1"] = _pairs_field_major(rows, np.repeat(d1[:, None], rows.shape[1], 1))
        return g, s


# ---------------------------------------------------------------------- DCN ---
def cross_fwd(x0, W, Bc):
    """dcn/dcn.py:132-142: x_{l+1} = (x_l . w_l) * x0 + x_l + b_l.  W,Bc: [L,dim]."""
    xs, ss = [x0], []
    for l in range(W.shape[0]):
        s = (xs[-1] * W[l]).sum(1, dtype=x0.dtype)
        ss.append(s)
        xs.append(s[:, None] * x0 + xs[-1] + Bc[l])
    return xs, ss


def cross_bwd(xs, ss, W, dxL):
    x0 = xs[0]
    dW, dB = np.zeros_like(W), np.zeros_like(W)
    dx0 = np.zeros_like(x0)
    dx = dxL
    for l in reversed(range(W.shape[0])):
        dB[l] = dx.sum(0)
        ds = (dx * x0).sum(1, dtype=x0.dtype)
        dx0 = dx0 + ss[l][:, None] * dx
        dW[l] = (ds[:, None] * xs[l]).sum(0)
        dx = dx + ds[:, None] * W[l]
    return dx0 + dx, dW, dB


class DCN:
    """dcn/dcn.py:117-153 (no first-order term; deep tower has no final 1-unit layer)."""

    def __init__(self, P, row_off, n_layers=2, dropout=0.5):
        self.P, self.row_off, self.n, self.rate = P, row_off, n_layers, dropout

    def forward(self, ids, train=True, masks=None):
        P = self.P
        rows, E, _, _, _ = gather_fm_fwd(P["tables"], None, ids, self.row_off)
        x0 = E.reshape(E.shape[0], -1)
        xs, ss = cross_fwd(x0, P["cross.W"], P["cross.b"])
        h, tc = tower_fwd(x0, P, "dnn", self.n, train, self.rate, masks)
        cat = np.concatenate([h, xs[-1]], 1)
        z = (cat @ P["out.W"] + P["out.b"]).reshape(-1)
        self.c = (rows, E, xs, ss, tc, cat, h.shape[1])
        return z

    def backward(self, dz):
        P = self.P
        rows, E, xs, ss, tc, cat, nh = self.c
        g, s = {}, {}
        dcat = dz[:, None] @ P["out.W"].T
        g["out.W"], g["out.b"] = cat.T @ dz[:, None], dz.sum(keepdims=True)
        dx_c, g["cross.W"], g["cross.b"] = cross_bwd(xs, ss, P["cross.W"], dcat[:, nh:])
        dx_d = tower_bwd(dcat[:, :nh], P, "dnn", tc, self.rate, g)
        s["tables"] = _pairs_field_major(rows, (dx_c + dx_d).reshape(E.shape))
        return g, s


# ------------------------------------------------------------------ xDeepFM ---
def cin_layer_fwd(X0, Xk, W, c):
    """xdeepfm/xdeepfm.py:145-172.  X0 [B,F,D], Xk [B,H,D], W [F*H,N] (f major, h minor; A-13), c [N]
    -> X_{k+1} [B,N,D] = relu(sum_{f,h} X0[b,f,d] Xk[b,h,d] W[f*H+h,n] + c[n])."""
    F, H = X0.shape[1], Xk.shape[1]
    W3 = W.reshape(F, H, -1)
    pre = np.einsum("bfd,bhd,fhn->bnd", X0, Xk, W3, optimize=True).astype(X0.dtype) + c[None, :, None]
    return np.maximum(pre, 0)


def cin_layer_bwd(X0, Xk, W, out, dout):
    F, H = X0.shape[1], Xk.shape[1]
    W3 = W.reshape(F, H, -1)
    dpre = dout * (out > 0)
    dW = np.einsum("bfd,bhd,bnd->fhn", X0, Xk, dpre, optimize=True).reshape(W.shape).astype(W.dtype)
    dc = dpre.sum((0, 2))
    dXk = np.einsum("bfd,fhn,bnd->bhd", X0, W3, dpre, optimize=True).astype(X0.dtype)
    dX0 = np.einsum("bhd,fhn,bnd->bfd", Xk, W3, dpre, optimize=True).astype(X0.dtype)
    return dX0, dXk, dW, dc


XDFM_LINEAR_ORDER = None  # filled lazily: 39 terms in TF's name-sorted order


def xdeepfm_linear_terms():
    """Sorted linear columns of xdeepfm/xdeepfm.py:82,91 (Appendix A-1):
    list of ('num', j) for numeric _c{j} or ('cat', j) for _c{j}_indicator."""
    names = [("_c%d" % j, ("num", j)) for j in range(1, 14)] + \
            [("_c%d_indicator" % j, ("cat", j)) for j in range(14, 40)]
    names.sort(key=lambda t: t[0])
    return [t[1] for t in names]


class XDeepFM:
    """xdeepfm/xdeepfm.py:123-196.  Two independent table sets (Appendix A-6):
    P['tables'] feeds CIN, P['tables2'] feeds the DNN.  Linear part: 13 log-values and 26
    one-hot indicator blocks -> P['lin.wnum'] [13] (index j-1 for _c{j}), P['lin.wcat'] [R_cat]
    (hashed fields _c14.._c39 in natural order, offsets `cat_off`), P['lin.b'] [1].
    `cat_slot[j-14]` = embedding slot of _c{j}."""

    def __init__(self, P, row_off, cat_slot, cat_off, cin_layers=(128, 128), n_layers=2, dropout=0.5):
        self.P, self.row_off, self.cat_slot, self.cat_off = P, row_off, cat_slot, cat_off
        self.cin, self.n, self.rate = list(cin_layers), n_layers, dropout

    def forward(self, ids, logx, train=True, masks=None):
        """ids [B,39]; logx [B,13] = log(x+shift) fp values of _c1.._c13."""
        P = self.P
        dt = P["tables"].dtype
        B = ids.shape[0]
        lin = np.zeros(B, dt)
        crow = ids[:, self.cat_slot].astype(np.int64) + self.cat_off[None, :-1]   # [B,26]
        for kind, j in xdeepfm_linear_terms():
            if kind == "num":
                lin = lin + logx[:, j - 1] * P["lin.wnum"][j - 1]
            else:
                lin = lin + P["lin.wcat"][crow[:, j - 14]]
        lin_y = np.maximum(lin + P["lin.b"][0], 0)
        rows, X0, _, _, _ = gather_fm_fwd(P["tables"], None, ids, self.row_off)
        Xs = [X0]
        for k in range(len(self.cin)):
            Xs.append(cin_layer_fwd(X0, Xs[-1], P[f"cin.W{k}"], P[f"cin.c{k}"]))
        res = np.concatenate(Xs[1:], 1).sum(2, dtype=dt)                            # [B, sumN]
        cin_y = nn.dense_fwd(res, P["cin.Wout"], P["cin.bout"], relu=True)
        _, E2, _, _, _ = gather_fm_fwd(P["tables2"], None, ids, self.row_off)
        h, tc = tower_fwd(E2.reshape(B, -1), P, "dnn", self.n, train, self.rate, masks)
        yd = nn.dense_fwd(h, P["dnn.Wout"], P["dnn.bout"], relu=True)
        cat = np.concatenate([lin_y[:, None], cin_y, yd], 1)
        z = (cat @ P["out.W"] + P["out.b"]).reshape(-1)
        self.c = (rows, crow, logx, lin_y, Xs, res, cin_y, E2, h, tc, yd, cat)
        return z

    def backward(self, dz):
        P = self.P
        rows, crow, logx, lin_y, Xs, res, cin_y, E2, h, tc, yd, cat = self.c
        g, s = {}, {}
        dcat = dz[:, None] @ P["out.W"].T
        g["out.W"], g["out.b"] = cat.T @ dz[:, None], dz.sum(keepdims=True)
        dl = dcat[:, 0] * (lin_y > 0)
        g["lin.b"] = dl.sum(keepdims=True)
        g["lin.wnum"] = (logx * dl[:, None]).sum(0)
        s["lin.wcat"] = _pairs_field_major(crow, np.repeat(dl[:, None], crow.shape[1], 1))
        dres, g["cin.Wout"], g["cin.bout"] = nn.dense_bwd(res, P["cin.Wout"], cin_y, dcat[:, 1:2], relu=True)
        X0 = Xs[0]
        dX0 = np.zeros_like(X0)
        dXs = []
        o = 0
        for k in range(len(self.cin)):
            n = self.cin[k]
            dXs.append(np.repeat(dres[:, o:o + n, None], X0.shape[2], 2))   # d/dX of sum over d
            o += n
        for k in reversed(range(len(self.cin))):
            d0, dk, g[f"cin.W{k}"], g[f"cin.c{k}"] = cin_layer_bwd(X0, Xs[k], P[f"cin.W{k}"], Xs[k + 1], dXs[k])
            dX0 = dX0 + d0
            if k > 0:
                dXs[k - 1] = dXs[k - 1] + dk
            else:
                dX0 = dX0 + dk
        s["tables"] = _pairs_field_major(rows, dX0)
        dh, g["dnn.Wout"], g["dnn.bout"] = nn.dense_bwd(h, P["dnn.Wout"], yd, dcat[:, 2:3], relu=True)
        dx = tower_bwd(dh, P, "dnn", tc, self.rate, g)
        s["tables2"] = _pairs_field_major(rows, dx.reshape(E2.shape))
        return g, s


# ---------------------------------------------------------------------- DIN ---
def din_attention_fwd(T, hist, q, P, pre, train, rate, masks):
    """din/din.py:103-125.  T [rows,K], hist [B,Pn] int, q [B,K] -> pooled [B,K]."""
    B, Pn = hist.shape
    K = T.shape[1]
    H = T[hist]                                             # [B,P,K]
    mask = (hist > 0).astype(T.dtype)[:, :, None]
    Hf = H.reshape(B * Pn, K)
    Qf = np.repeat(q[:, None, :], Pn, 1).reshape(B * Pn, K)
    feat = np.concatenate([Hf, Qf, Hf * Qf, Hf - Qf], 1)
    a1 = nn.dense_fwd(feat, P[f"{pre}.W0"], P[f"{pre}.b0"], relu=True)
    m1 = masks[0] if (train and masks is not None) else None
    o1 = nn.dropout_fwd(a1, rate, m1)
    a2 = nn.dense_fwd(o1, P[f"{pre}.W1"], P[f"{pre}.b1"], relu=True)
    m2 = masks[1] if (train and masks is not None) else None
    o2 = nn.dropout_fwd(a2, rate, m2)
    w = nn.dense_fwd(o2, P[f"{pre}.W2"], P[f"{pre}.b2"]).reshape(B, Pn, 1)
    out = (H * w * mask).sum(1, dtype=T.dtype)
    return out, (H, mask, Hf, Qf, feat, a1, m1, o1, a2, m2, o2, w)


def din_attention_bwd(cache, dout, P, pre, rate, g):
    H, mask, Hf, Qf, feat, a1, m1, o1, a2, m2, o2, w = cache
    B, Pn, K = H.shape
    dwe = dout[:, None, :] * mask                           # d(wgt_emb) [B,P,K]
    dH = dwe * w
    dw = (H * dwe).sum(2, dtype=H.dtype).reshape(B * Pn, 1)
    do2, g[f"{pre}.W2"], g[f"{pre}.b2"] = nn.dense_bwd(o2, P[f"{pre}.W2"], None, dw)
    da2 = nn.dropout_bwd(do2, rate, m2)
    do1, g[f"{pre}.W1"], g[f"{pre}.b1"] = nn.dense_bwd(o1, P[f"{pre}.W1"], a2, da2, relu=True)
    da1 = nn.dropout_bwd(do1, rate, m1)
    dfeat, g[f"{pre}.W0"], g[f"{pre}.b0"] = nn.dense_bwd(feat, P[f"{pre}.W0"], a1, da1, relu=True)
    d_h, d_q, d_hq, d_hmq = dfeat[:, :K], dfeat[:, K:2 * K], dfeat[:, 2 * K:3 * K], dfeat[:, 3 * K:]
    dHf = d_h + d_hq * Qf + d_hmq
    dQf = d_q + d_hq * Hf - d_hmq
    dH = dH + dHf.reshape(B, Pn, K)
    dq = dQf.reshape(B, Pn, K).sum(1, dtype=H.dtype)
    return dH, dq


class DIN:
    """din/din.py:83-140.  P: item_bias [Ni], item_emb [Ni,K], cate_emb [Nc,K],
    att_i.{W,b}{0,1,2}, att_c.{W,b}{0,1,2}, mlp.{W,b}{0,1,2}, mlp.Wout, mlp.bout."""

    def __init__(self, P, dropout=0.5):
        self.P, self.rate = P, dropout

    def forward(self, i_id, i_cate, hist_i, hist_c, train=True, masks=None):
        P = self.P
        mk = masks or {}
        qi, qc = P["item_emb"][i_id], P["cate_emb"][i_cate]
        hi, ci = din_attention_fwd(P["item_emb"], hist_i, qi, P, "att_i", train, self.rate, mk.get("att_i"))
        hc, cc = din_attention_fwd(P["cate_emb"], hist_c, qc, P, "att_c", train, self.rate, mk.get("att_c"))
        net = np.concatenate([qi, hi, hc], 1)
        acts = []
        x = net
        for i in range(3):
            a = nn.dense_fwd(x, P[f"mlp.W{i}"], P[f"mlp.b{i}"], relu=True)
            m = mk["mlp"][i] if (train and "mlp" in mk) else None
            o = nn.dropout_fwd(a, self.rate, m)
            acts.append((x, a, m))
            x = o
        z = nn.dense_fwd(x, P["mlp.Wout"], P["mlp.bout"]).reshape(-1) + P["item_bias"][i_id]
        self.c = (i_id, i_cate, hist_i, hist_c, ci, cc, acts, x)
        return z

    def backward(self, dz):
        P = self.P
        i_id, i_cate, hist_i, hist_c, ci, cc, acts, xl = self.c
        K = P["item_emb"].shape[1]
        g, s = {}, {}
        dx, g["mlp.Wout"], g["mlp.bout"] = nn.dense_bwd(xl, P["mlp.Wout"], None, dz[:, None])
        for i in reversed(range(3)):
            x, a, m = acts[i]
            da = nn.dropout_bwd(dx, self.rate, m)
            dx, g[f"mlp.W{i}"], g[f"mlp.b{i}"] = nn.dense_bwd(x, P[f"mlp.W{i}"], a, da, relu=True)
        dqi, dhi, dhc = dx[:, :K], dx[:, K:2 * K], dx[:, 2 * K:]
        dHi, dq_i = din_attention_bwd(ci, dhi, P, "att_i", self.rate, g)
        dHc, dq_c = din_attention_bwd(cc, dhc, P, "att_c", self.rate, g)
        B, Pn = hist_i.shape
        # pair order: the direct lookup first, then the history (b major, p minor)
        s["item_emb"] = (np.concatenate([i_id, hist_i.reshape(-1)]),
                         np.concatenate([dqi + dq_i, dHi.reshape(B * Pn, K)]))
        s["cate_emb"] = (np.concatenate([i_cate, hist_c.reshape(-1)]),
                         np.concatenate([dq_c, dHc.reshape(B * Pn, K)]))
        s["item_bias"] = (i_id.copy(), dz.copy())
        return g, s


# ------------------------------------------------------------- train step ----
def train_step(model, opt, fwd_args, labels, fwd_kwargs=None, lazy=False):
    """One Estimator TRAIN step: forward, mean sigmoid-CE, backward, TF-1 Adam on every variable.
    First-order / linear kernels come from tf.layers.dense on a one-hot input, so their gradient is
    DENSE in TF (ApplyAdam formula); embedding tables and tf.gather'd vectors are IndexedSlices
    (non-lazy sparse formula).  Returns (loss, logits)."""
    P = model.P
    z = model.forward(*fwd_args, train=True, **(fwd_kwargs or {}))
    loss, dz = nn.sigmoid_ce_mean(z, labels)
    g, s = model.backward(dz)
    for name, grad in g.items():
        opt.apply_dense(name, P[name], grad.reshape(P[name].shape).astype(P[name].dtype))
    for name, (rows, vals) in s.items():
        uniq, G = nn.segment_sum_rows(rows, vals)
        if name in ("w1", "lin.wcat"):            # dense-gradient variables (one-hot matmul kernels)
            if lazy:
                opt.apply_sparse(name, P[name], uniq, G, lazy=True)
            else:
                dense = np.zeros_like(P[name])
                dense[uniq] = G
                opt.apply_dense(name, P[name], dense)
        else:
            opt.apply_sparse(name, P[name], uniq, G, lazy=lazy)
    opt.finish_step()
    return loss, z


def train_step_dp(model, opt, rank_fwd_args, rank_labels, rank_fwd_kwargs=None, lazy=False):
    """One SYNCHRONOUS data-parallel TRAIN step of N replicas on N different batches, as tf.distribute.MirroredStrategy
    runs it (fm/fm.py:184-194, deepfm/readme.md:24 "each step runs two batches"): every replica runs
    forward / backward on ITS batch with the shared variables (batch-norm statistics and dropout per replica), the replica
    losses are scaled by 1/N, dense gradients are SUMMED over the replicas, the replicas' IndexedSlices are concatenated in
    replica order and de-duplicated by the optimizer (one segment-sum over the global batch), then ONE TF-1 Adam update.
    rank_fwd_args[r]: the forward arguments of replica r; rank_labels[r]; rank_fwd_kwargs[r] (e.g. its dropout masks).
    Returns (per-replica mean losses, per-replica logits)."""
    P = model.P
    N = len(rank_fwd_args)
    g_sum, s_cat, losses, zs = {}, {}, [], []
    for r in range(N):
        kw = (rank_fwd_kwargs[r] if rank_fwd_kwargs is not None else None) or {}
        z = model.forward(*rank_fwd_args[r], train=True, **kw)
        loss, dz = nn.sigmoid_ce_mean(z, rank_labels[r])
        g, s = model.backward(dz / N)
        losses.append(loss)
        zs.append(z)
        for name, grad in g.items():
            grad = grad.reshape(P[name].shape)
            g_sum[name] = grad.copy() if name not in g_sum else g_sum[name] + grad
        for name, (rows, vals) in s.items():
            s_cat.setdefault(name, []).append((rows, vals))
    for name, grad in g_sum.items():
        opt.apply_dense(name, P[name], grad.astype(P[name].dtype))
    for name, parts in s_cat.items():
        rows = np.concatenate([p[0] for p in parts])
        vals = np.concatenate([p[1] for p in parts])
        uniq, G = nn.segment_sum_rows(rows, vals)
        if name in ("w1", "lin.wcat"):
            if lazy:
                opt.apply_sparse(name, P[name], uniq, G, lazy=True)
            else:
                dense = np.zeros_like(P[name])
                dense[uniq] = G
                opt.apply_dense(name, P[name], dense)
        else:
            opt.apply_sparse(name, P[name], uniq, G, lazy=lazy)
    opt.finish_step()
    return losses, zs
