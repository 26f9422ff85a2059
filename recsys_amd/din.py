"""Deep Interest Network on Amazon-Electronics style data -- MI355X-native mirror of `din/din.py`
(model_fn :83-180, feature_description :44-50, flags :12-40; BASELINE config 5).

Variables (din/din.py:88-90): `i_item` bias [63002] (zeros), `i_id` [63002, K] and `i_cate` [802, K] (glorot-normal).
Two attention blocks with separate MLP weights (hard-coded 80-40, :85) pool the two id histories against the target
item / category (:103-128); MLP 100-50-20 on [q_item, pooled_item, pooled_cate] (:130-138); logits += i_item[i_id] (:139).
Hand kernels: row gathers, the attention-weighted masked history sum (fwd/bwd), the sparse-row dedup/segment-sum and
the non-lazy TF-1 Adam sweep.  The attention / head MLP GEMMs (M = B*P rows) are library GEMMs via torch.
"""
import argparse
import os

import torch

from . import layers as L
from .estimator import Estimator, EstimatorSpec, EvalSpec, ModeKeys, RunConfig, TrainSpec, get_variable_store, \
    train_and_evaluate
from . import _lib
from .ops import make_scatter_riders, DinAttnFn, DinAttnPoolFn, DinPoolFn, EmbeddingArena, FusedTower, SparseTable, _ptr, _stream

ATTENTION_LAYERS = [80, 40]      # din/din.py:85 (the din_layers flag is ignored by the reference)
MLP_LAYERS = [100, 50, 20]       # din/din.py:86 (the deep_layers flag is ignored by the reference)
N_ITEM, N_CATE = 63002, 802      # din/din.py:88-90


def _prefer_rocblas():
    """The attention MLP's weight gradients are [K, B*P] x [B*P, N] GEMMs with B*P = 102 400: torch's default fp32 path
    (hipBLASLt) runs them in 270-360 us each, rocBLAS in 38-54 us (scripts/gemm_probe.py; 6 of them per step)."""
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.backends.cuda.preferred_blas_library("cublas")      # "cublas" names rocBLAS on ROCm
    except Exception:                                                  # knob absent in this torch: keep the default
        pass


def build_variables(store, params, B, P):
    _prefer_rocblas()
    K = params["embedding_size"]
    n_item, n_cate = params.get("n_item", N_ITEM), params.get("n_cate", N_CATE)
    world = store.dp.world if store.dp is not None else 1
    cap = (B * (P + 1)) * world
    # id 0 is the history padding (din/din.py:56-57,107): its gradient entries are exactly zero (masked sum), so the
    # sparse path may skip that one huge segment -- but ONLY the histories' entries: a target item / category with id 0
    # (tf.gather / embedding_lookup train row 0 like any other) keeps its gradient.
    # Storage: ONE two-field arena [item rows + 1 | category rows + 1, K] (+ both Adam slots) and a one-field arena for the
    # item bias, stored as column 0 of a 4-wide table.  The extra last row of every field is a DUMMY row: the fused step
    # maps the history padding entries' sort keys to it (their gradients are exactly zero), so that row 0 stays an ordinary
    # row for the target lookups (din/din.py:95-107 masks id 0 in the histories only).  The SparseTable objects the
    # autograd path, the checkpoints and the tests see are VIEWS of the same memory.
    dev = store.device
    row_off = [0, n_item + 1, n_item + 1 + n_cate + 1]
    arena = EmbeddingArena(row_off, K, cap, dev, with_w1=True, w1_field_mask=1)    # (first-order path = the item bias, field 0)
    # The bias rides through the item field's scatter as its "first-order vector" and is therefore indexed by GLOBAL rows of the
    # two-field arena (the item field's dummy row n_item is written by its row owner, the category field's rows are prefetched):
    # it is allocated over the arena's whole row space; the variable i_item [n_item] is the leading slice.
    barena = EmbeddingArena([0, arena.R], 4, B * world, dev, with_w1=False)
    with torch.no_grad():
        arena.tables.zero_()
        barena.tables.zero_()

    def view(tbl_arena, lo, rows, Kw, capacity, null_row):
        t = SparseTable(rows, Kw, capacity, dev, table=tbl_arena.tables[lo:lo + rows], null_row=null_row)
        assert t.table.data_ptr() == tbl_arena.tables[lo:lo + rows].data_ptr()
        t.m, t.v = tbl_arena.m_t[lo:lo + rows], tbl_arena.v_t[lo:lo + rows]
        return t

    item = view(arena, 0, n_item, K, cap, n_item)              # null_row == rows: the dummy key of the history padding
    cate = view(arena, n_item + 1, n_cate, K, cap, n_cate)
    bias = view(barena, 0, n_item, 4, B * world, -1)           # i_item [n_item] stored as column 0 of a 4-wide table
    with torch.no_grad():
        for tbl, rows in ((item, n_item), (cate, n_cate)):
            t = torch.empty(rows, K)
            L.glorot_normal_(t, rows, K, store.gen)             # tf.glorot_normal_initializer over [rows, K]
            tbl.table.copy_(t)
    shapes, init = {}, {}
    zeros = lambda t, g: t.zero_()

    def add(pre, d, widths):
        for i, n in enumerate(widths):
            shapes[f"{pre}.W{i}"], shapes[f"{pre}.b{i}"] = (d, n), (n,)
            init[f"{pre}.W{i}"] = lambda t, g, fi=d, fo=n: L.glorot_uniform_(t, fi, fo, g)
            init[f"{pre}.b{i}"] = zeros
            d = n
        return d

    add("att_i", 4 * K, ATTENTION_LAYERS + [1])
    add("att_c", 4 * K, ATTENTION_LAYERS + [1])
    d = add("mlp", 3 * K, MLP_LAYERS)
    shapes["mlp.Wout"], shapes["mlp.bout"] = (d, 1), (1,)
    init["mlp.Wout"] = lambda t, g, fi=d: L.glorot_uniform_(t, fi, 1, g)
    init["mlp.bout"] = zeros
    # The fused tower reads activation rows as float4: layer widths are STORED padded to a multiple of 4 (50 -> 52); the
    # variables keep their reference shapes as leading slices, the pad units have zero weights, zero bias, zero gradient.
    pad4 = lambda n: (n + 3) & ~3
    widths = [pad4(n) for n in MLP_LAYERS]
    storage, dprev = {}, 3 * K
    for i, n in enumerate(widths):
        storage[f"mlp.W{i}"], storage[f"mlp.b{i}"] = (dprev, n), (n,)
        dprev = n
    storage["mlp.Wout"] = (dprev, 1)
    store.build({"i_id": item, "i_cate": cate, "i_item": bias}, shapes, init, params["learning_rate"], storage)
    store.tower = None
    if (3 * K) % 4 == 0 and widths[-1] <= 256:
        store.tower = FusedTower(store.dense, "mlp", 3 * K, widths, B, store.device, batch_norm=False)
    store.din = None
    if store.tower is not None and len(ATTENTION_LAYERS) == 2 and DinAttnFn.supported(K, *ATTENTION_LAYERS) and \
            store.adam_mode == "tf1_dense":
        store.din = DinFused(store, arena, barena, n_item, n_cate, K, B, P)
        if store.dp is not None:
            # data parallel (round 4): the fused step's collectives are plain calls between graph segments (deepfm.py)
            store.graph_safe_dp = True


class DinFused:
    """din/din.py:83-173 TRAIN as a fixed launch sequence, no autograd (what _train_fused is to deepfm.py):
      keys of both id tables (1 launch) -> ONE dedup sort for both (csrc/sort_large.hip, F = 2) -> the untouched-row sweep
      -> ONE launch for the six lookups (:96-105) -> per history: row list of the non-padding positions, fused attention MLP,
      masked weighted sum written straight into its slice of the 'mlp_layer' input (:131, never concatenated) -> MLP + loss
      forward / backward (FusedTower, no batch-norm) -> per history: pooling backward + attention backward, both writing
      their share of d(history rows) into the scatter's value block [entries, 2, K]; the target rows' gradients (the
      attention's dq + the MLP input slice) land in the same block -> the bias rows -> two scatter + Adam launches.
    The autograd path (`fused_step=False`) needed 68 launches, ~17 of them framework glue (cat / copy / fill)."""

    def __init__(self, store, arena, barena, n_item, n_cate, K, B, P):
        import ctypes as C
        self.C = C
        dev = store.device
        self.arena, self.barena = arena, barena
        self.n_item, self.n_cate, self.K, self.cap_B, self.P = n_item, n_cate, K, B, P
        N = B * (P + 1)
        f32 = dict(device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.keys2 = torch.zeros(N, 2, **i32)
        self.keys_t = torch.zeros(2, arena.stride, **i32)     # the same keys field-major (large sorts: no transpose launch)
        self.labels_f = torch.zeros(B, **f32)
        self.X = torch.empty(B, 3 * K, **f32)                 # [q_item | pooled item history | pooled category history]
        self.qi, self.qc = torch.empty(B, K, **f32), torch.empty(B, K, **f32)
        self.ib = torch.empty(B, **f32)
        self.H = [torch.empty(B * P, K, **f32) for _ in range(2)]
        self.vals = torch.zeros(N, 2 * K, **f32)              # the scatter's value block: entry e -> [item grad | category grad]
        self.gbias = torch.zeros(N, **f32)                    # d loss / d i_item[i_id] of the target entries, 0 for history entries
        self.dp = store.dp
        self.ux = None
        # padding rows of the two-field arena: the last row of every field (include/rsx.h RSX_NULL_LAST_ROW)
        arena.null_last = True
        arena._bind_partials()
        if self.dp is not None:
            from .deepfm import dp_unique_wanted
            if dp_unique_wanted(store, {}) and EmbeddingArena.unique_exchange_ok(arena.row_off_np, self.dp.world):
                # round 5: every rank de-duplicates and sums ITS entries (the single replica's sort + scatter), the ranks exchange
                # unique (row, sum) lists: send block [dense | G [capT, K] | bias sums [capT]], capT = min(entries, item rows + 1)
                # + min(entries, category rows + 1) instead of 2K + 1 floats per ENTRY (8.4 MB instead of 27 MB per rank and step
                # at batch 1 024 x 100)
                self.ux = arena.enable_unique_exchange(self.dp.world, N)
                self.keys_t = torch.zeros(2, self.ux.local.stride, **i32)
                self.dp.make_send_block(store.dense, self.ux.capT, [K, 1])
                store.dp_unique = True
            else:
                # RSX_DP_EXCHANGE=examples: value block and bias gradients live in the persistent send block behind the dense
                # gradient arena ([dense | vals (2K per entry) | gbias (1 per entry)]): ONE all-gather straight from it, no pack
                # copy; the global scatter reads every rank's block in place (dist.DataParallel.gather_send_block)
                self.dp.make_send_block(store.dense, N, [2 * K, 1])
                self.vals = self.gbias = None
        n1, n2 = ATTENTION_LAYERS
        self.a1 = [torch.empty(B * P, n1, **f32) for _ in range(2)]
        self.a2 = [torch.empty(B * P, n2, **f32) for _ in range(2)]
        self.w = [torch.empty(B, P, **f32) for _ in range(2)]
        self.dw = [torch.empty(B, P, **f32) for _ in range(2)]
        self.rows = [torch.empty(B * P + 2 + (B * P + 1023) // 1024, **i32) for _ in range(2)]
        self.ws = [torch.empty(int(_lib.lib().rsx_din_attn_bwd_workspace_floats(B, P, K, n1, n2)), **f32) for _ in range(2)]

    def peer_entry_keys(self, features):
        """EmulatedDataParallel (per-example exchange): the sort keys [entries, 2] a PEER rank holding `features` would send."""
        i_id = features["i_id"].to(torch.int32).contiguous()
        i_cate = features["i_cate"].to(torch.int32).contiguous()
        hist = [features["u_iid_seq"].to(torch.int32).contiguous(), features["u_icat_seq"].to(torch.int32).contiguous()]
        B = i_id.shape[0]
        out = torch.zeros(B * (self.P + 1), 2, dtype=torch.int32, device=i_id.device)
        _lib.check(_lib.lib().rsx_din_keys(_ptr(i_id), _ptr(i_cate), _ptr(hist[0]), _ptr(hist[1]), B, self.P, self.n_item,
                                          self.n_cate, _ptr(out), _stream()), "rsx_din_keys")
        return out

    def ux_peer_keys(self, features):
        """EmulatedDataParallel: the packed key block a PEER rank holding `features` would send -- its sort keys (rsx_din_keys), its
        own dedup sort and pack, on scratch workspaces."""
        a, ux = self.arena, self.ux
        i_id = features["i_id"].to(torch.int32).contiguous()
        i_cate = features["i_cate"].to(torch.int32).contiguous()
        hist = [features["u_iid_seq"].to(torch.int32).contiguous(), features["u_icat_seq"].to(torch.int32).contiguous()]
        B = i_id.shape[0]
        N = B * (self.P + 1)
        if not hasattr(ux, "scratch"):
            ux.scratch = ux.new_local()
            self._peer_keys2 = torch.zeros_like(self.keys2)
        _lib.check(_lib.lib().rsx_din_keys(_ptr(i_id), _ptr(i_cate), _ptr(hist[0]), _ptr(hist[1]), B, self.P, self.n_item,
                                          self.n_cate, _ptr(self._peer_keys2), _stream()), "rsx_din_keys")
        own = (ux.local, ux.keys)
        ux.local, ux.keys = ux.scratch
        try:
            ux.local.select(0)
            ux.local.field_sort(self._peer_keys2[:N])
            out = a.ux_pack(1).reshape(-1).clone()
        finally:
            ux.local, ux.keys = own
        return out

    def _side_stream(self):
        if _lib.form("din_side_sort") != "1":
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        return self._side

    def _gather_jobs(self, B, i_id, i_cate, hist):
        """The step's six lookups (din/din.py:96-105) as rsx_gather_job structs; the category history LAST (the fused step lets
        the first five ride in its first prepare launch and that one in the second)."""
        a = self.arena
        K, P = self.K, self.P
        cate_tab = a.tables[self.n_item + 1:]
        jobs = (_lib.GatherJob * 6)()
        spec = [(a.tables, i_id, self.X, B, K, 3 * K, 0), (a.tables, i_id, self.qi, B, K, K, 0),
                (cate_tab, i_cate, self.qc, B, K, K, 0),
                (self.barena.tables, i_id, self.ib, B, 1, 1, 4),       # tf.gather(i_item, i_id) (:96): column 0 of the 4-wide table
                (a.tables, hist[0], self.H[0], B * P, K, K, 0), (cate_tab, hist[1], self.H[1], B * P, K, K, 0)]
        for j, (tab, ids, out, n, k, ld, ldt) in zip(jobs, spec):
            j.table, j.ids, j.out, j.n, j.K, j.ld_out, j.row_base, j.ld_table = \
                tab.data_ptr(), ids.data_ptr(), out.data_ptr(), n, k, ld, 0, ldt
        return jobs

    def _gather(self, B, i_id, i_cate, hist):
        _lib.check(_lib.lib().rsx_gather_rows_multi(self._gather_jobs(B, i_id, i_cate, hist), 6, _stream()), "rsx_gather_rows_multi")

    def train_step(self, store, features, labels, params, masks):
        L = _lib.lib()
        C, a, K, P = self.C, self.arena, self.K, self.P
        i_id = features["i_id"].to(torch.int32).contiguous()
        i_cate = features["i_cate"].to(torch.int32).contiguous()
        hist = [features["u_iid_seq"].to(torch.int32).contiguous(), features["u_icat_seq"].to(torch.int32).contiguous()]
        B = i_id.shape[0]
        assert B <= self.cap_B and hist[0].shape[1] == P
        N = B * (P + 1)
        st = _stream()
        P_ = store.dense
        rate = params["dropout"]
        step = store.opt.state.view(torch.int32)[3:4]
        n1, n2 = ATTENTION_LAYERS
        with torch.no_grad():
            # ---- ids only: keys, ONE sort for both tables, the bias rows' sort, then the sweep over the untouched rows ----
            # (sort keys of both tables + both histories' lists of non-padding positions: two launches)
            keys2 = self.keys2[:N]
            cnts = [self.rows[t][B * P:] for t in range(2)]
            dp = self.dp
            ux = self.ux is not None
            world = dp.world if dp is not None else 1
            # the large sort takes field-major keys as they are (single replica, and the rank's OWN sort of the unique-list exchange)
            big = N > a.LDS_SORT_MAX_B and (dp is None or ux)
            lab64 = labels.reshape(-1) if labels.dtype == torch.int64 and labels.is_contiguous() else None
            # Round 4: the six lookups ride in the two prepare launches as extra workgroups (they depend on the ids only, like the
            # prepare kernels: the 11 us bandwidth-bound gather runs beside two latency-bound launches); RSX_DIN_GATHER_RIDE=0: its
            # own launch after them
            gride = _lib.form("din_gather_ride") == "1"
            gjobs = self._gather_jobs(B, i_id, i_cate, hist) if gride else None
            _lib.check(L.rsx_din_prepare2_gather(_ptr(i_id), _ptr(i_cate), _ptr(hist[0]), _ptr(hist[1]), B, P, self.n_item,
                                                 self.n_cate, _ptr(self.keys_t) if big else _ptr(keys2), self.keys_t.shape[1] if big else 0,
                                                 _ptr(self.rows[0]), _ptr(cnts[0]), _ptr(self.w[0]), _ptr(self.rows[1]),
                                                 _ptr(cnts[1]), _ptr(self.w[1]), _ptr(lab64),
                                                 _ptr(self.labels_f) if lab64 is not None else None, gjobs, 6 if gride else 0,
                                                 5 if gride else 0, st), "rsx_din_prepare2_gather")
            labels_f = self.labels_f[:B] if lab64 is not None else labels.reshape(-1).to(torch.float32)
            a.select(0)
            # Round 4: the ids-only branch of the step -- the dedup sort (7 launches on 52 workgroups: a chain of launch latencies)
            # and the sweep over the rows it leaves untouched -- runs on a SIDE stream beside the forward / backward launches
            # (which read and write touched rows only) and joins the step's stream before the scatter.  Captured, the fork and
            # the join are two edges of the step's graph.  Data parallel: the keys' all-gather stays on the step's stream (it is
            # ordered with the other collectives there, and ends a graph segment); the fork follows it, inside the next segment
            # (RSX_DIN_SIDE_SORT_DP=0: in line).  RSX_DIN_SIDE_SORT=0: in line.
            keys_g = None
            if dp is not None and not ux:
                # the optimizer sees the GLOBAL batch (TF concatenates the replicas' IndexedSlices): the dedup sort runs over the
                # all-gathered keys, rank blocks in order -- the entry order of the gathered value block
                keys_g = dp.all_gather_entry_keys(keys2, features["i_id"], self.peer_entry_keys)
                vals_full, gbias_full = dp.send_views(N)
            side = self._side_stream() if (dp is None or _lib.form("din_side_sort_dp") == "1") else None
            main = torch.cuda.current_stream()
            if side is not None:
                side.wait_stream(main)
            with torch.cuda.stream(side if side is not None else main):
                if ux:
                    # unique-list exchange: the rank's OWN dedup sort + its key block, beside the forward; the all-gather of the key
                    # blocks, the merge into global lists and the sweep follow the forward (below)
                    loc = self.ux.local
                    loc.select(0)
                    if big:
                        loc.field_sort_t(self.keys_t, N)
                    else:
                        loc.field_sort(keys2)
                    keys_l = a.ux_pack(1)
                elif dp is not None:
                    a.field_sort(keys_g)
                elif big:
                    a.field_sort_t(self.keys_t, N)
                else:
                    a.field_sort(keys2)
                # The item bias (i_item, tf.gather by the target ids: :96,139) rides with the item table: its rows ARE item rows, so the
                # item field's dedup serves it; rows that only a history touches get a zero-gradient update from the same launch.
                def cold_sweep():
                    cold = [a.adam_split_segments()[0][0],
                            dict(kind=_lib.RSX_ADAM_TABLE_TF1_COLD, d=4, n=self.n_item, var=self.barena.tables, m=self.barena.m_t,
                                 v=self.barena.v_t, slot=a.slot, slot_w=None)]
                    store.opt.run_slice(store.opt.cold_slices(cold, [1.0])[0])
                if not ux:
                    cold_sweep()
            # ---- forward --------------------------------------------------------------------------------------------
            if not gride:
                self._gather(B, i_id, i_cate, hist)
            q = (self.qi[:B], self.qc[:B])
            att_m = []
            att_seed = 0xD1A77 + 7919 * (dp.rank if dp is not None else 0)
            att_mk = [None, None]
            if masks is not None:
                att_mk = [masks.get("att_i"), masks.get("att_c")]
            for t, pre in enumerate(("att_i", "att_c")):
                Ws = [P_[f"{pre}.{v}{i}"] for i in range(3) for v in ("W", "b")]
                mk = att_mk[t] if rate > 0.0 else None
                m1, m2 = (None, None) if mk is None else (mk[0].contiguous(), mk[1].contiguous())
                rows, cnt = self.rows[t], cnts[t]
                _lib.check(L.rsx_din_attn_fwd(_ptr(self.H[t]), _ptr(q[t]), *[_ptr(x) for x in Ws], _ptr(self.a1[t]), _ptr(self.a2[t]),
                                              _ptr(self.w[t]), _ptr(m1), _ptr(m2), _ptr(step), att_seed, 2 * t, rate, _ptr(rows),
                                              _ptr(cnt), B, P, K, n1, n2, st), "rsx_din_attn_fwd")
                att_m.append((m1, m2))
            # masked weighted sums of both histories in one launch, written into their slices of the MLP input
            _lib.check(L.rsx_din_pool_fwd_pair(_ptr(self.H[0]), _ptr(self.w[0]), _ptr(hist[0]),
                                               C.c_void_p(self.X.data_ptr() + 4 * K), _ptr(self.H[1]), _ptr(self.w[1]),
                                               _ptr(hist[1]), C.c_void_p(self.X.data_ptr() + 8 * K), B, P, K, 3 * K, st),
                       "rsx_din_pool_fwd_pair")
            tw = store.tower
            mlp_mk = None if masks is None or "mlp" not in masks else \
                [torch.nn.functional.pad(m, (0, w - m.shape[1]), value=1.0) for m, w in zip(masks["mlp"], tw.widths)]
            gbias = self.gbias[:N] if (dp is None or ux) else gbias_full
            rank = dp.rank if dp is not None else 0
            loss, prob, dX, gs0, _ = tw.train_step(
                self.X[:B], labels_f, rate, step, s0=self.ib[:B],
                head=("mlp.Wout", "mlp.bout", None, None), relu0=False, relu2=False, replicas=world, masks=mlp_mk,
                seed=0xD1AD + 7919 * rank,                       # replicas draw independent dropout patterns
                outs=(None, gbias[:B], None),                    # d loss / d bias lands in the scatter's first-order input
                # the weight-gradient reduce of the one-launch mlp_layer rides in the pooling-backward launch below
                # (RSX_MLP_REDUCE_RIDE=0: its own launch)
                reduce_rider=_lib.form("mlp_reduce_ride") == "1")
            rider = getattr(tw, "mlp_reduce_job", None)
            if B < self.cap_B:
                # a batch smaller than an earlier one (the final partial batch of an epoch): entries B.. are HISTORY entries now,
                # whose bias gradient is zero -- not what a larger batch's head (or, in the send block, a larger batch's value
                # block) left there.  Unconditional for this batch size, so that a captured graph of the step carries it.
                gbias[B:(self.cap_B if (dp is None or ux) else N)].zero_()
            # ---- backward of the two attention blocks, straight into the scatter's value block --------------------------
            if ux:
                # ids phase, second half: the key blocks' all-gather (the local sort ran beside the forward), the merge into the
                # global unique-row lists / slot map / src, and the untouched-row sweep, which needs that slot map
                if side is not None:
                    main.wait_stream(side)
                keys_all = dp.all_gather_keys(keys_l, a, [features["i_id"]])
                # ... and back onto the side stream: the merge and the sweep touch the lists / slot map / untouched rows only, the
                # backward launches below read touched rows and the rank's OWN sort; joined before the optimizer launch
                if side is not None:
                    side.wait_stream(main)
                with torch.cuda.stream(side if side is not None else main):
                    a.select(0)
                    a.ux_merge(keys_all, 1)
                    cold_sweep()
            vals = self.vals[:N] if (dp is None or ux) else vals_full
            vbase = vals.data_ptr()
            dHp = [C.c_void_p(vbase + 4 * (B * 2 * K + t * K)) for t in range(2)]        # rows B.. of column block t
            doutp = [C.c_void_p(dX.data_ptr() + 4 * K * (t + 1)) for t in range(2)]       # d(pooled history t) = dX[:, (t+1)K : (t+2)K]
            _lib.check(L.rsx_din_pool_bwd_pair_ride(_ptr(self.H[0]), _ptr(self.w[0]), _ptr(hist[0]), doutp[0], dHp[0], _ptr(self.dw[0]),
                                                    _ptr(self.H[1]), _ptr(self.w[1]), _ptr(hist[1]), doutp[1], dHp[1], _ptr(self.dw[1]),
                                                    0, B, P, K, 3 * K, 2 * K, None if rider is None else C.byref(rider), st),
                       "rsx_din_pool_bwd_pair_ride")
            gouts = []
            for t, pre in enumerate(("att_i", "att_c")):
                Ws = [P_[f"{pre}.W{i}"] for i in range(3)]
                names = [f"{pre}.{v}{i}" for i in range(3) for v in ("W", "b")]
                gout = P_.packed_grad(names)
                assert gout is not None
                gouts.append(gout)
                m1, m2 = att_m[t]
                rows, cnt = self.rows[t], self.rows[t][B * P:]
                _lib.check(L.rsx_din_attn_bwd_nofinish(_ptr(self.H[t]), _ptr(q[t]), *[_ptr(x) for x in Ws], _ptr(self.a1[t]),
                                                       _ptr(self.a2[t]), _ptr(self.dw[t]), dHp[t], _ptr(self.ws[t]), _ptr(m1),
                                                       _ptr(m2), _ptr(step), att_seed, 2 * t, rate, 1, _ptr(rows), _ptr(cnt),
                                                       _ptr(hist[t]), B, P, K, n1, n2, 2 * K, st), "rsx_din_attn_bwd_nofinish")
            # weight gradients + the target rows' gradients (dq, plus the MLP input slice for the item block) of both blocks
            dqp = [C.c_void_p(vbase + 4 * t * K) for t in range(2)]         # rows 0 .. B-1 of column block t
            # (round 4, single replica: the two weight-gradient reduces -- 28 of the launch's 54 MB, read by the optimizer only --
            # come back as jobs and ride in the scatter's stage-A launch; RSX_SCATTER_RIDERS=0: inside this launch)
            ride_fin = dp is None and _lib.form("scatter_riders") == "1"
            vjobs = (_lib.VecReduceJob * 2)() if ride_fin else None
            _lib.check(L.rsx_din_attn_finish_pair_defer(_ptr(self.ws[0]), _ptr(gouts[0]), dqp[0], _ptr(hist[0]), _ptr(dX),
                                                        _ptr(self.ws[1]), _ptr(gouts[1]), dqp[1], _ptr(hist[1]), None,
                                                        B, P, K, n1, n2, 2 * K, 3 * K, vjobs, st), "rsx_din_attn_finish_pair_defer")
            riders = make_scatter_riders(vec_jobs=list(vjobs)) if ride_fin else None
            if side is not None:
                main.wait_stream(side)            # the scatter (train_op) needs the sort; the sweep must precede its Adam state advance
            if ux:      # the rank's own segment-sum (stage A + packed stage B), written as its block of the send buffer
                Gv, gbv = dp.send_views(self.ux.capT)
                a.ux_segsum_local(N, None, vals, gbias, None, Gv, gbv, 0)

        def train_op():                                                    # AdamOptimizer.minimize (:172-173)
            with torch.no_grad():
                ba = self.barena
                if dp is None:
                    a.segsum_adam(N, None, vals, gbias, None, store.opt, store.dense.adam_segments(),
                                  w1_ext=(ba.tables, ba.m_t, ba.v_t, 4, 1), riders=riders)
                elif ux:
                    # ONE collective [dense | G | bias sums], then the touched-row Adam of both tables + the bias off the merged lists
                    (G0, gb0), blocks, dense_segs = dp.gather_send_block(self.ux.capT, fold_dense=True)
                    a.select(0)
                    a.ux_merged_adam(G0, gb0, blocks[1], store.opt, dense_segs or store.dense.adam_segments(),
                                     w1_ext=(ba.tables, ba.m_t, ba.v_t, 4, 1))
                else:
                    # ONE collective: [dense gradient arena | value block | bias gradients] of every rank; the dense arenas are
                    # summed in rank order inside the optimizer launch, the scatter reads the rank blocks in place
                    (vg, gbg), blocks, dense_segs = dp.gather_send_block(N, fold_dense=True)
                    # (dense_segs is None when the arena took the all-reduce branch -- RSX_DP_ALLREDUCE_MIN_BYTES, a large
                    # embedding_size: the gradients were then summed in place and the arena's own segments apply them)
                    a.segsum_adam(N * world, None, vg, gbg, None, store.opt, dense_segs or store.dense.adam_segments(),
                                  blocks=blocks,
                                  w1_ext=(ba.tables, ba.m_t, ba.v_t, 4, 1))

        return EstimatorSpec(ModeKeys.TRAIN, predictions={"prob": prob}, loss=loss[0], train_op=train_op)


class DinHeadFn(torch.autograd.Function):
    """'mlp_layer' + logits + loss of the TRAIN step (din/din.py:131-147) through the fused tower kernels (csrc/tower.hip,
    no batch-norm): forward AND backward run inside forward(); backward() hands the stored input gradients to autograd
    (the attention / pooling / lookups upstream stay autograd Functions).  The loss already carries the 1/(B*replicas)
    scale, so backward() expects an incoming gradient of 1."""

    @staticmethod
    def forward(ctx, X, i_b, labels, store, rate, masks, replicas):
        t = store.tower
        if masks is not None:       # injected keep masks (parity tests) arrive with the reference widths: pad to storage
            masks = [torch.nn.functional.pad(m, (0, w - m.shape[1]), value=1.0) for m, w in zip(masks, t.widths)]
        loss, prob, dX, gs0, _ = t.train_step(
            X.contiguous(), labels.reshape(-1).to(torch.float32), rate, store.opt.state.view(torch.int32)[3:4],
            s0=i_b.contiguous(), head=("mlp.Wout", "mlp.bout", None, None), relu0=False, relu2=False, replicas=replicas,
            masks=masks, seed=0xD1AD)
        ctx.save_for_backward(dX, gs0)
        ctx.mark_non_differentiable(prob)
        return loss[0], prob

    @staticmethod
    def backward(ctx, g_loss, g_prob):
        dX, gs0 = ctx.saved_tensors
        return dX, gs0, None, None, None, None, None


def _attention(tbl, hist, q, P_, pre, training, rate, masks, store=None, layer0=0):
    """din/din.py:103-125."""
    B, Pn = hist.shape
    K = tbl.K
    H = tbl.lookup(hist, pad_id=0)                                         # dense_emb [B,P,K] (:105); id 0 = padding (:107)
    n1, n2 = P_[f"{pre}.W0"].shape[1], P_[f"{pre}.W1"].shape[1]
    if len(ATTENTION_LAYERS) == 2 and DinAttnFn.supported(K, n1, n2) and store is not None:
        # fused MFMA kernel: the [B*P, 4K] concat, the tiled query and the layer outputs never round-trip through HBM;
        # dropout = the counter hash of the fused tower (keyed by the optimizer's device-side step counter)
        r = rate if training else 0.0
        mk2 = masks if (training and r > 0.0) else None
        names = [f"{pre}.{v}{i}" for i in range(3) for v in ("W", "b")]
        gout = P_.packed_grad(names) if training else None
        Ws = [P_[n] for n in names]
        step = store.opt.state.view(torch.int32)[3:4]
        if gout is not None:        # TRAIN: attention + pooling as one node, weight gradients straight into the arena
            return DinAttnPoolFn.apply(H, q, hist, *Ws, r, mk2, step, 0xD1A77, layer0, gout)
        w = DinAttnFn.apply(H, q, *Ws, r, mk2, step, 0xD1A77, layer0)
        return DinPoolFn.apply(H, w, hist)                                  # masked weighted sum (:122-124)
    hist_emb = H.reshape(B * Pn, K)
    query_emb = q[:, None, :].expand(B, Pn, K).reshape(B * Pn, K)          # tile + reshape (:111)
    att = torch.cat([hist_emb, query_emb, hist_emb * query_emb, hist_emb - query_emb], 1)     # (:114)
    for i in range(len(ATTENTION_LAYERS)):
        att = L.dense(att, P_[f"{pre}.W{i}"], P_[f"{pre}.b{i}"], relu=True)
        att = L.dropout(att, rate, training, None if masks is None else masks[i])
    w = L.dense(att, P_[f"{pre}.W2"], P_[f"{pre}.b2"]).reshape(B, Pn)        # att_wgt (:120-121)
    return DinPoolFn.apply(H, w, hist)                                      # masked weighted sum (:122-124)


def model_fn(features, labels, mode, params):
    """din/din.py:83-180.  features: i_id, i_cate int [B]; u_iid_seq, u_icat_seq int [B,P] (zero padded)."""
    store = get_variable_store()
    i_id = features["i_id"].to(torch.int32)
    i_cate = features["i_cate"].to(torch.int32)
    hist_i = features["u_iid_seq"].to(torch.int32).contiguous()
    hist_c = features["u_icat_seq"].to(torch.int32).contiguous()
    if not store.built:
        build_variables(store, params, max(int(params.get("max_batch_size", 0)), i_id.shape[0]), hist_i.shape[1])
    item, cate, bias = (store.embeddings[k] for k in ("i_id", "i_cate", "i_item"))
    P_ = store.dense
    training = mode == ModeKeys.TRAIN
    rate = params["dropout"]
    mk = params.get("_dropout_masks") or {}
    if training and store.din is not None and params.get("fused_step", True) and \
            params.get("fused_attention", True) and params.get("fused_head", True):
        return store.din.train_step(store, features, labels, params, params.get("_dropout_masks"))

    i_b = bias.lookup(i_id)[:, 0]                                          # tf.gather(pkg_w, i_id) (:96)
    pkg_emb = item.lookup(i_id)                                            # (:100)
    pkgc_emb = cate.lookup(i_cate)                                         # (:101)
    fused = store if params.get("fused_attention", True) else None
    pkg_emb_h = _attention(item, hist_i, pkg_emb, P_, "att_i", training, rate, mk.get("att_i"), fused, 0)
    pkgc_emb_h = _attention(cate, hist_c, pkgc_emb, P_, "att_c", training, rate, mk.get("att_c"), fused, 2)
    net = torch.cat([pkg_emb, pkg_emb_h, pkgc_emb_h], 1)                   # 'mlp_layer' (:131)
    fused_head = training and store.tower is not None and params.get("fused_head", True)
    if fused_head:
        dp = store.dp
        loss, pred = DinHeadFn.apply(net, i_b, labels, store, rate, mk.get("mlp"), dp.world if dp is not None else 1)

        def train_op_fused():                                              # AdamOptimizer.minimize (:172-173)
            loss.backward()                                                # (1/world is inside the fused head's loss scale)
            with torch.no_grad():
                for tbl in (item, cate, bias):
                    tbl.finalize(dp)
                if dp is not None:
                    dp.all_reduce_sum(store.dense.grad)
                store.apply_gradients()

        return EstimatorSpec(mode, predictions={"prob": pred}, loss=loss, train_op=train_op_fused)
    if not training and store.tower is not None and params.get("fused_infer", True) and not torch.is_grad_enabled() \
            and net.shape[0] <= store.tower.cap:
        # EVAL / PREDICT head through the TRAIN step's kernels (FusedTower.infer: 3 forward launches + the head, dropout off)
        lab = None if (labels is None or mode == ModeKeys.PREDICT) else labels.reshape(-1).to(torch.float32)
        prob, loss = store.tower.infer(net.contiguous(), store.opt.state.view(torch.int32)[3:4], lab, s0=i_b.contiguous(),
                                       head=("mlp.Wout", "mlp.bout", None, None), relu0=False, relu2=False)
        predictions = {"prob": prob}
        if mode == ModeKeys.PREDICT:
            return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
        return EstimatorSpec(mode, predictions=predictions, loss=loss[0], eval_metric_ops={"AUC": None, "Accuracy": None})
    for i in range(len(MLP_LAYERS)):
        net = L.dense(net, P_[f"mlp.W{i}"], P_[f"mlp.b{i}"], relu=True)
        net = L.dropout(net, rate, training, mk["mlp"][i] if "mlp" in mk else None)
    logits = L.dense(net, P_["mlp.Wout"], P_["mlp.bout"]).reshape(-1) + i_b   # (:138-139)
    pred = torch.sigmoid(logits)
    predictions = {"prob": pred}
    if mode == ModeKeys.PREDICT:
        return EstimatorSpec(mode, predictions=predictions, export_outputs={"serving_default": predictions})
    loss = L.sigmoid_ce_mean(logits, labels)
    if mode == ModeKeys.EVAL:
        return EstimatorSpec(mode, predictions=predictions, loss=loss, eval_metric_ops={"AUC": None, "Accuracy": None})
    dp = store.dp
    # MirroredStrategy scales the replica loss by 1/N for the GRADIENTS (the fused head does the same inside); the spec carries
    # the replica's own mean loss like every other TRAIN path -- the Estimator's log line averages over the replicas
    loss_bwd = loss / dp.world if dp is not None else loss

    def train_op():                                                        # AdamOptimizer.minimize (:172-173)
        loss_bwd.backward()
        with torch.no_grad():
            for tbl in (item, cate, bias):
                tbl.finalize(dp)
            if dp is not None:
                dp.all_reduce_sum(store.dense.grad)
            store.apply_gradients()

    return EstimatorSpec(mode, predictions=predictions, loss=loss, train_op=train_op)


def define_flags():
    p = argparse.ArgumentParser()
    p.add_argument("--embedding_size", type=int, default=32)
    p.add_argument("--learning_rate", type=float, default=0.001)
    p.add_argument("--dropout", type=float, default=0.5)
    p.add_argument("--task_type", default="train")
    p.add_argument("--num_epochs", type=int, default=5)
    p.add_argument("--deep_layers", default="100,100")        # accepted, ignored like the reference (:19,85-86)
    p.add_argument("--din_layers", default="80,40")
    p.add_argument("--train_path", default="/home/wangrc/din_dataset/")
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--log_steps", type=int, default=100)
    p.add_argument("--eval_steps", type=int, default=200)
    p.add_argument("--save_checkpoints_steps", type=int, default=1000)
    p.add_argument("--mirror", type=lambda s: s.lower() in ("1", "true", "yes"), default=True)
    p.add_argument("--model_dir", default="./din_model/")
    p.add_argument("--hist_len", type=int, default=100)
    return p


def input_fn(filenames, batch_size, num_epochs=-1, need_shuffle=False, hist_len=100, shard=None, shard_tail=False):
    from .input_pipeline import din_input_fn
    return din_input_fn(filenames, batch_size, num_epochs, need_shuffle, hist_len=hist_len, ids_int32=True, shard=shard,
                        shard_tail=shard_tail)


def main(argv=None):
    FLAGS = define_flags().parse_args(argv)
    if FLAGS.mirror:                # MirroredStrategy() = every GPU of the host (din/din.py:187-190): become the launcher
        from . import dist
        rc = dist.maybe_spawn_mirror(FLAGS, __name__, argv)
        if rc is not None:
            if rc != 0:
                raise SystemExit(rc)
            return None
    train_files, eval_files = [FLAGS.train_path + "train2"], [FLAGS.train_path + "valid2"]      # din/din.py:197-198
    params = {"embedding_size": FLAGS.embedding_size, "learning_rate": FLAGS.learning_rate, "dropout": FLAGS.dropout,
              "max_batch_size": FLAGS.batch_size, "hist_len": FLAGS.hist_len}
    cfg = RunConfig(save_checkpoints_steps=FLAGS.save_checkpoints_steps, keep_checkpoint_max=5,
                    log_step_count_steps=FLAGS.log_steps)
    est = Estimator(model_fn, FLAGS.model_dir, params, cfg)
    shard = None
    if FLAGS.mirror:
        from . import dist
        dp = dist.attach_if_distributed(est)
        if dp is not None:          # every replica takes its own batches of the one stream (MirroredStrategy, din/din.py:187-190)
            shard = (dp.rank, dp.world)
    if FLAGS.task_type == "train":
        tr = TrainSpec(lambda: input_fn(train_files, FLAGS.batch_size, FLAGS.num_epochs, True, FLAGS.hist_len, shard))
        ev = EvalSpec(lambda: input_fn(eval_files, FLAGS.batch_size, 1, False, FLAGS.hist_len, shard, True), steps=FLAGS.eval_steps)
        return train_and_evaluate(est, tr, ev)
    if FLAGS.task_type == "eval":
        return est.evaluate(lambda: input_fn(eval_files, FLAGS.batch_size, 1, False, FLAGS.hist_len, shard, True), steps=FLAGS.eval_steps)
    return list(zip(range(10), est.predict(lambda: input_fn(eval_files, FLAGS.batch_size, 1, False, FLAGS.hist_len))))


if __name__ == "__main__":
    main()
