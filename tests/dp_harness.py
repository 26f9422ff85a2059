"""Single-GPU stand-ins for an N-rank data-parallel group (TEST and PROFILING harnesses, not product: moved out of
recsys_amd/dist.py in round 6).

  EmulatedDataParallel   bench.py --emulate_world N: one process plays rank 0 of N replicas so that the per-rank COMPUTE of an
                         N-GPU step can be timed on one GPU (collectives replaced by local tiling / peers' real batches).
  LoopbackDataParallel   tests/test_gpu_dp_loopback.py: ONE process plays the N ranks of a synchronous step one after the other on
                         N DIFFERENT batches, held against the oracle's MirroredStrategy step (oracle.models.train_step_dp).
Both subclass recsys_amd.dist.DataParallel and override only its collectives."""
import torch

from recsys_amd.dist import DataParallel


class EmulatedDataParallel(DataParallel):
    always_captured = True      # (dist.dp_capture)
    """PROFILING AID (bench.py --emulate_world N): one process plays rank 0 of N replicas so that the per-rank COMPUTE of an
    N-GPU step can be timed on one GPU.  No collective latency is modelled.
    What the peers contribute decides the optimizer stage's work -- how many DISTINCT rows the global step touches -- so the
    ids phase can be fed with REAL other batches (set_peers: round 5; VERDICT r4 weak #3 -- tiling one batch N times gives the
    global batch the unique-row count of a single replica): the peers' id matrices (per-example exchange) or their packed
    unique-row lists (unique-list exchange; computed once per resident batch by `key_fn`, cached).  Gradient VALUES are still
    this rank's own, repeated (all_gather_rows tiles; all_reduce_sum multiplies by N): values do not change the timing.
    Without set_peers every collective tiles: N identical replicas (the parity tests' use; exact but not representative)."""

    def __init__(self, world):
        self.group, self.rank, self.world, self.comm = None, 0, int(world), None
        self._peers, self._key_fn, self._key_cache = {}, None, {}

    def set_peers(self, peers, key_fn=None):
        """peers: {data_ptr of a resident batch's token tensor (its ids; din.py: i_id): [the N-1 peer ranks' feature dicts]};
        key_fn(features) -> the packed key block [KS] int32 of that batch (EmbeddingArena.ux_peer_keys / DinFused.ux_peer_keys)."""
        self._peers, self._key_fn, self._key_cache = dict(peers), key_fn, {}

    def warm_keys(self):
        """Computes every peer key block NOW (outside graph capture and outside the timed region)."""
        if getattr(self, "_entry_fn", None) is not None:
            for ptr, plist in self._peers.items():
                for r, pf in enumerate(plist):
                    self._key_cache[(ptr, "entry", r)] = self._entry_fn(pf).clone()
        if self._key_fn is None:
            return
        for ptr, plist in self._peers.items():
            for r, pf in enumerate(plist):
                if (ptr, r) not in self._key_cache:
                    self._key_cache[(ptr, r)] = self._key_fn(pf).reshape(-1).clone()

    def all_gather_id_list(self, ids_list, prefetchable=False):
        out = []
        for ids in ids_list:
            pl = self._peers.get(ids.data_ptr())
            out.append(ids.repeat(self.world, 1) if pl is None else torch.cat([ids] + [pf["ids"] for pf in pl], 0))
        return out

    def all_gather_entry_keys(self, keys, token=None, peer_fn=None):
        pl = self._peers.get(token.data_ptr()) if token is not None else None
        if pl is None or peer_fn is None:
            return self.all_gather_rows(keys)
        rows = [keys]
        for r, pf in enumerate(pl):
            c = self._key_cache.get((token.data_ptr(), "entry", r))
            if c is None:
                c = self._key_cache[(token.data_ptr(), "entry", r)] = peer_fn(pf).clone()
            rows.append(c)
        return torch.cat(rows, 0)

    def all_gather_keys(self, keys, arena=None, ids_list=None):
        k = len(ids_list) if ids_list else 1
        KS = keys.numel() // k
        rows = [keys.reshape(1, -1)]
        for r in range(self.world - 1):
            blocks = []
            for i in range(k):
                t = ids_list[i] if ids_list else None
                c = self._key_cache.get((t.data_ptr(), r)) if t is not None else None
                if c is None and t is not None and self._key_fn is not None and t.data_ptr() in self._peers:
                    c = self._key_cache[(t.data_ptr(), r)] = self._key_fn(self._peers[t.data_ptr()][r]).reshape(-1).clone()
                blocks.append(c if c is not None else keys.reshape(-1)[i * KS:(i + 1) * KS])
            rows.append(torch.cat(blocks).reshape(1, -1))
        return torch.cat(rows, 0)

    def all_gather_rows(self, x, prefetchable=False):
        x = x.contiguous()
        return x.repeat((self.world,) + (1,) * (x.dim() - 1))

    def all_reduce_sum(self, flat):
        return flat.mul_(self.world)

    def all_reduce_async(self, flat):
        flat.mul_(self.world)
        return None

    def wait_all(self, handles):
        pass

    def _all_gather_into(self, out, x):
        out.copy_(x.expand_as(out))

    def _overlapped_allreduce_allgather(self, grad, out, x):
        grad.mul_(self.world)
        out.copy_(x.expand_as(out))

    def barrier(self):
        pass


class LoopbackDataParallel(DataParallel):
    always_captured = True      # (dist.dp_capture)
    """TEST HARNESS (tests/test_gpu_dp_loopback.py, VERDICT r4 item 1a): ONE process plays the N ranks of a synchronous
    data-parallel step ONE AFTER THE OTHER on N DIFFERENT batches -- shared variables, per-rank batch-norm statistics and
    dropout seeds, exactly the kernels and the send / gathered buffer layouts of a real N-rank run -- so that a wrong rank
    stride, block offset or replica sum FAILS against the oracle (EmulatedDataParallel tiles one batch N times: its N rank
    blocks are byte-identical and cannot).

    Protocol (loopback_train_step below): ranks 0 .. N-2 run model_fn as SHADOW passes -- forward and backward write the
    rank's send block, which is stashed; everything that changes optimizer state before train_op (untouched-row sweeps,
    AdamTF1.shadow) is skipped -- then rank N-1 runs model_fn with every collective seeing all N ranks' real inputs (its
    dedup sort is the global one) and its train_op runs the optimizer stage once over the N real send blocks.
    Collectives are matched by their call order inside model_fn, which is the same on every rank.  Fused steps with the
    zero-copy send block only (what the data-parallel product path runs)."""

    def __init__(self, world):
        self.group, self.rank, self.world, self.comm = None, 0, int(world), None
        self.shadow = False
        self._calls, self._ci, self._stash, self._phase = [], 0, [], "model"

    # -- driver side --------------------------------------------------------------------------------------
    def begin_step(self):
        self._calls, self._stash = [], []

    def enter_rank(self, r, store):
        self.rank, self._ci, self._phase = r, 0, "model"
        self.shadow = r < self.world - 1
        store.opt.shadow = self.shadow
        # rank-LOCAL state of the unique-list exchange (the rank's own sort workspaces and key blocks live across the steps of an
        # optimizer window): one set per played rank
        arenas = list(store.embeddings.values())
        if getattr(store, "din", None) is not None:
            arenas.append(store.din.arena)           # (din.py: the two-field arena behind its SparseTable views)
        for a in arenas:
            ux = getattr(a, "ux", None)
            if ux is not None:
                if not hasattr(ux, "rank_locals"):
                    ux.rank_locals = [(ux.local, ux.keys)] + [ux.new_local() for _ in range(self.world - 1)]
                ux.local, ux.keys = ux.rank_locals[r]

    def leave_rank(self, store):
        """After model_fn of a shadow rank: keep its send block (dense gradient arena | per-unit gradient block)."""
        if self.shadow:
            self._stash.append(self._send.clone())
        else:
            self._phase = "train_op"
        store.opt.shadow = False

    # -- collectives ----------------------------------------------------------------------------------------
    def _send_offset(self, x):
        s = getattr(self, "_send", None)
        if s is None:
            return None
        o = (x.data_ptr() - s.data_ptr()) // 4
        return o if (0 <= o and o + x.numel() <= s.numel() and x.dtype == s.dtype) else None

    def _gather_from_send(self, x):
        o = self._send_offset(x)
        assert len(self._stash) == self.world - 1 and self.rank == self.world - 1, "loopback: train_op of the LAST rank only"
        rows = [st[o:o + x.numel()].view(x.shape) for st in self._stash] + [x]
        return torch.cat(rows, 0)

    def all_gather_rows(self, x, prefetchable=False):
        x = x.contiguous()
        if self._send_offset(x) is not None:
            return self._gather_from_send(x)
        if self._phase != "model":
            raise RuntimeError("LoopbackDataParallel: a collective outside the send block inside train_op -- the shadow ranks' "
                               "inputs of it do not exist (zero-copy send block paths only)")
        c = self._ci
        self._ci += 1
        if c == len(self._calls):
            self._calls.append([None] * self.world)
        self._calls[c][self.rank] = x.clone()
        # (shadow passes see their own block in place of the ranks that have not run yet: their sort results are overwritten
        # by the last rank's, whose gathered buffer holds every rank's real block)
        return torch.cat([t if t is not None else x for t in self._calls[c]], 0)

    def _all_gather_into(self, out, x):
        out.copy_(self._gather_from_send(x).view_as(out))

    def _overlapped_allreduce_allgather(self, grad, out, x):
        o = self._send_offset(grad)
        tot = self._stash[0][o:o + grad.numel()].clone()
        for st in self._stash[1:]:
            tot += st[o:o + grad.numel()]
        tot += grad                                  # rank order: the live block is the last rank's
        grad.copy_(tot)
        self._all_gather_into(out, x)

    def all_reduce_sum(self, flat):
        raise RuntimeError("LoopbackDataParallel: fused steps only (no autograd-path all-reduce)")

    def all_reduce_async(self, flat):
        raise RuntimeError("LoopbackDataParallel does not play RSX_DP_OVERLAP")

    def wait_all(self, handles):
        pass

    def barrier(self):
        pass


def loopback_train_step(est, rank_features, rank_labels, window=None, before_rank=None):
    """One data-parallel TRAIN step of est.store.dp.world ranks on ONE GPU (LoopbackDataParallel): rank r trains on
    (rank_features[r], rank_labels[r]).  window = (k, pos, rank_window_features) with rank_window_features[r] = the features
    of rank r's k batches: the step is position pos of an optimizer window.  before_rank(r): called before rank r's model_fn
    (e.g. to inject that rank's dropout masks into est.params).  Eager (no HIP graphs).
    -> the ranks' mean losses (floats)."""
    from recsys_amd.estimator import ModeKeys
    st, dp = est.store, est.store.dp
    assert isinstance(dp, LoopbackDataParallel) and len(rank_features) == dp.world
    dp.begin_step()
    losses, spec = [], None
    for r in range(dp.world):
        dp.enter_rank(r, st)
        if before_rank is not None:
            before_rank(r)
        st.window = (window[0], window[1], window[2][r]) if (window is not None and window[0] > 1) else None
        try:
            spec = est._call_model_fn(rank_features[r], rank_labels[r], ModeKeys.TRAIN)
        finally:
            st.window = None
        losses.append(float(spec.loss))
        dp.leave_rank(st)
    spec.train_op()
    dp.rank, dp._phase = 0, "model"
    return losses
