"""Torch-side plumbing over the C ABI (include/rsx.h): device buffers, streams, autograd glue.

PyTorch is only the memory/stream/autograd carrier here; every op of the embedding path and the
optimizer is a hand-written gfx950 kernel in librsx.so, called with raw device pointers.  There is no
CPU fallback: constructing these objects without a GPU + librsx.so raises.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import AdamSeg, check, lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_scatter_riders(dw_jobs=None, cross_job=None, vec_jobs=None):
    """_lib.ScatterRiders from the tower's deferred dW reduce jobs (FusedTower.train_step(defer_dw_reduce=True) ->
    tower.dw_jobs_pending), the cross layers' deferred reduce (CrossLayers.backward(defer_reduce=True)) and deferred
    partial-vector reduces (din.py: rsx_din_attn_finish_pair_defer); None if there is none."""
    dw_jobs = list(dw_jobs or ())
    vec_jobs = [v for v in (vec_jobs or ()) if v.n > 0]
    if not dw_jobs and (cross_job is None or cross_job.n == 0) and not vec_jobs:
        return None
    r = _lib.ScatterRiders()
    # (the job tables are fixed-size C arrays: refuse here, before a backward launch has deferred a reduce nobody would run)
    if len(dw_jobs) > len(r.dw) or len(vec_jobs) > len(r.vec):
        raise _lib.RsxError("make_scatter_riders: %d dW / %d vector reduce jobs exceed the rider tables (%d / %d)"
                            % (len(dw_jobs), len(vec_jobs), len(r.dw), len(r.vec)))
    for i, j in enumerate(dw_jobs):
        r.dw[i] = j
    r.n_dw = len(dw_jobs)
    if cross_job is not None and cross_job.n > 0:
        r.cross = cross_job
    for i, v in enumerate(vec_jobs):
        r.vec[i] = v
    r.n_vec = len(vec_jobs)
    return r


def run_scatter_riders(riders):
    """The riders as their own launches (a scatter without a stage A to carry them)."""
    if riders is None:
        return
    if riders.n_dw > 0:
        check(lib().rsx_tower_reduce_dw_jobs(riders.dw, riders.n_dw, _stream()), "rsx_tower_reduce_dw_jobs")
    if riders.cross.n > 0:
        check(lib().rsx_cross_reduce_run(C.byref(riders.cross), _stream()), "rsx_cross_reduce_run")
    if riders.n_vec > 0:
        check(lib().rsx_vec_reduce_run(riders.vec, riders.n_vec, _stream()), "rsx_vec_reduce_run")


def _require_cuda(dev):
    if not torch.cuda.is_available():
        raise _lib.RsxError("recsys_amd hot path needs an MI355X (torch.cuda.is_available() is False); "
                            "there is no CPU fallback")
    lib()
    return torch.device(dev)


class EmbeddingArena:
    """All embedding tables of one `input_layer` call, concatenated row-wise: tables[R, D] (+ the
    first-order weight vector w1[R] of the indicator columns), their Adam slots and the
    dedup/scatter workspace.  Mirrors the variables created by
    tf.feature_column.input_layer(features, embedding cols) (fm/fm.py:118) and the [R,1] kernel of
    tf.layers.dense(linear_net, 1) (fm/fm.py:121)."""

    def __init__(self, row_off, D, capacity, device="cuda", with_w1=False, w1_field_mask=None,
                 tables=None, w1=None, workspace_only=False):
        dev = _require_cuda(device)
        self.row_off_np = np.asarray(row_off, np.int64)
        self.F = len(self.row_off_np) - 1
        self.R = int(self.row_off_np[-1])
        self.D = int(D)
        self.stride = int(capacity)
        self.max_rows = int(np.diff(self.row_off_np).max())
        self.row_off = torch.tensor(self.row_off_np, dtype=torch.int32, device=dev)
        self.tables = torch.empty(self.R, D, device=dev) if tables is None else \
            torch.as_tensor(tables, dtype=torch.float32).to(dev).contiguous()
        assert self.tables.shape == (self.R, D)
        # workspace_only: the sort / segment-sum workspace of ANOTHER arena's variables (the rank-local half of the data-parallel
        # unique-list exchange, enable_unique_exchange): `tables` / `w1` alias the owner's, there are no Adam slots
        self.m_t = None if workspace_only else torch.zeros_like(self.tables)
        self.v_t = None if workspace_only else torch.zeros_like(self.tables)
        self.with_w1 = with_w1
        self.w1_mask = (1 << self.F) - 1 if w1_field_mask is None else int(w1_field_mask)
        if with_w1:
            self.w1 = torch.empty(self.R, device=dev) if w1 is None else \
                torch.as_tensor(w1, dtype=torch.float32).to(dev).contiguous().reshape(-1)
            self.m_w = None if workspace_only else torch.zeros_like(self.w1)
            self.v_w = None if workspace_only else torch.zeros_like(self.w1)
        else:
            self.w1 = None
        F, st = self.F, self.stride
        i32 = dict(dtype=torch.int32, device=dev)
        self.G = torch.zeros(F * st, D, device=dev)
        self.gw1 = torch.zeros(F * st, device=dev) if with_w1 else None
        self.last_B = 0
        # batches beyond one workgroup's LDS sort through global memory (csrc/sort_large.hip)
        self.sort_ws = None
        if st > self.LDS_SORT_MAX_B:
            self.sort_ws = torch.zeros(int(lib().rsx_field_sort_large_workspace_ints(st, F, st)), **i32)
        # two-stage segment-sum workspace (B > TWO_STAGE_MIN_B): segment index per sorted position + chunk partials
        self.two_stage_ws = st > self.TWO_STAGE_MIN_B
        if self.two_stage_ws:
            nch = (st + 15) // 16
            self.P = torch.zeros(F * nch * 2, D, device=dev)
            self.P1 = torch.zeros(F * nch * 2, device=dev) if with_w1 else None
        # Dedup-sort workspaces: sortbufs[0] is the step's own; an optimizer WINDOW of k steps (rsx_adam_window) sorts
        # the k batches ahead into sortbufs[0..k-1] (allocated on first use) and select(i) makes entry i the current one.
        # (the slot maps of all window positions are slices of ONE allocation, equally spaced: rsx_adam_seg.slot_w)
        self._slot_stride = (self.R + 4 + 3) // 4 * 4
        self._slot_all = torch.full((_lib.ADAM_WINDOW_MAX, self._slot_stride), -1, **i32)
        self.sortbufs = [self._new_sortbuf()]
        self.window_bufs(_lib.default_adam_window(st, True))      # all any exchange may use, NOW: never inside a HIP-graph capture (see window_bufs)
        self.select(0)
        # requires-grad hook so autograd calls GatherFM.backward although the tables are raw buffers
        self.hook = torch.zeros((), device=dev, requires_grad=True)

    # Above this batch size the scatter runs in two stages (uniform 16-position chunks feed the long segments): small
    # batches are latency-bound and one launch wins; large / skewed ones are bound by the longest segment chain.
    # Measured on Criteo-39 (scripts/segsum_modes.py, scatter+Adam): B=1024 15.6 us single vs 28.4 two-stage, 2048: 33.5
    # vs 35.9 stand-alone but 31 us better two-stage inside the data-parallel step, 4096: 72.4 vs 53.9.
    TWO_STAGE_MIN_B = 1024
    LDS_SORT_MAX_B = 8192         # above: rsx_field_sort_large (multi-workgroup; measured crossover ~8192, scripts/sort_time.py;
                                  # rsx_field_sort itself reaches 16384)

    def _new_sortbuf(self):
        F, st = self.F, self.stride
        i32 = dict(dtype=torch.int32, device=self.tables.device)
        b = dict(perm=torch.zeros(F * st, **i32), seg_off=torch.zeros(F * (st + 1), **i32),
                 uniq_row=torch.zeros(F * st, **i32), nuniq=torch.zeros(F, **i32),
                 slot=self._slot_all[len(self.sortbufs) if hasattr(self, "sortbufs") else 0],   # [R + 4]: int4 tail reads of VEC_SLOT
                 segid=torch.zeros(F * st + 2 * F + F * ((st + 15) // 16), **i32) if self.two_stage_ws else None)
        return b

    def _bind_partials(self):
        # (G / gw1: stage A finishes every segment of <= 16 entries straight into the scatter's outputs)
        self.partials = None
        if self.two_stage_ws:
            nl = getattr(self, "null_last", False)     # padding entries keyed to every field's last (dummy) row: never walked
            self.partials = _lib.SegPartials(_ptr(self.segid), _ptr(self.P), _ptr(self.P1), _ptr(self.G), _ptr(self.gw1),
                                             _ptr(self.row_off) if nl else None, _lib.NULL_LAST_ROW if nl else _lib.NULL_NONE, 0)

    def select(self, i):
        """Makes sort workspace i (position i of the optimizer window) the one field_sort / sort_job / segsum* use."""
        owner = getattr(self, "_sort_owner", None)
        bufs = owner.window_bufs(i + 1) if owner is not None else self.window_bufs(i + 1)
        for k, v in bufs[i].items():
            setattr(self, k, v)
        self.cur_buf = i
        self._bind_partials()

    def window_bufs(self, k):
        assert 1 <= k <= _lib.ADAM_WINDOW_MAX
        while len(self.sortbufs) < k:
            # A workspace created while a HIP graph is being captured would live in that graph's memory pool and its
            # zero-fill would be replayed with the graph -- wiping the unique-row count the next sort needs to clear the
            # previous slot-map entries (found the hard way: windows of 7-8 steps diverged on the second replay).
            if torch.cuda.is_current_stream_capturing():
                raise _lib.RsxError("EmbeddingArena.window_bufs(%d): allocate the window's sort workspaces before capture" % k)
            self.sortbufs.append(self._new_sortbuf())
        return self.sortbufs[:k]

    def _two_stage(self, B):
        return self.partials is not None and B > self.TWO_STAGE_MIN_B

    def share_sort_of(self, other):
        """Two table sets looked up with the SAME ids over the same field layout (xDeepFM's two input_layer calls,
        xdeepfm/xdeepfm.py:125,185) have one dedup result: alias `other`'s sort outputs; field_sort here becomes a no-op."""
        assert np.array_equal(self.row_off_np, other.row_off_np) and self.stride == other.stride
        self._sort_owner = other
        self.sort_ws = other.sort_ws
        self.select(other.cur_buf)

    # -- data parallel: exchange of per-rank unique-row lists (csrc/uniq_exchange.hip) ----------------------
    @staticmethod
    def unique_exchange_ok(row_off, world):
        """rsx_uniq_merge keeps a presence bitmap + prefix counts of a field's rows in LDS (rows / 4 bytes of 160 KB) and the
        optimizer launch unrolls over at most RSX_UNIQ_MAX_RANKS ranks."""
        rows = np.diff(np.asarray(row_off, np.int64))
        return world <= _lib.UNIQ_MAX_RANKS and int(rows.max()) <= 640000 and len(rows) <= 64

    def enable_unique_exchange(self, world, b_local, sort_capacity=None):
        """Data parallel, round 5 (include/rsx.h "exchange of per-rank UNIQUE-ROW lists"): this arena (capacity = the GLOBAL
        batch) keeps the global unique-row lists, slot maps and optimizer state; `self.ux.local` -- a workspace-only arena of
        capacity b_local over the SAME variables -- runs the rank's own dedup sort and segment-sum.  goff: field f's rows of a
        rank's block start at goff[f]; cap_f = min(b_local, rows_f) bounds the unique rows one rank can have in field f."""
        from types import SimpleNamespace
        assert self.unique_exchange_ok(self.row_off_np, world) and b_local * world <= self.stride
        dev = self.tables.device
        rows = np.diff(self.row_off_np)
        caps = np.minimum(rows, b_local)
        goff = np.concatenate([[0], np.cumsum(caps)]).astype(np.int32)
        capT = int(goff[-1])
        i32 = dict(dtype=torch.int32, device=dev)
        # row-range parts per field of the merge (rsx_uniq_merge): one workgroup per (field, part, rank) keeps the longest list
        # walk near 8 192 entries -- 1 for deepfm.py at 8 x 256, 4 for dcn.py at 8 x 4 096, 44 for din.py's item table at 8 x 1 024
        parts = int(os.environ.get("RSX_UX_PARTS", "0")) or max(1, min(_lib.UNIQ_MAX_PARTS, -(-int(caps.max()) * int(world) // 8192)))
        KS = (self.F + self.F * (parts + 1) + capT + 3) & ~3

        def new_local():
            loc = EmbeddingArena(self.row_off_np, self.D, int(sort_capacity or b_local), dev, with_w1=self.with_w1,
                                 w1_field_mask=self.w1_mask, tables=self.tables, w1=self.w1, workspace_only=True)
            assert loc.tables.data_ptr() == self.tables.data_ptr()
            if getattr(self, "null_last", False):         # (din.py: the padding entries' dummy rows, never walked)
                loc.null_last = True
                loc._bind_partials()
            return loc, torch.zeros(_lib.ADAM_WINDOW_MAX, KS, **i32)

        local, keys = new_local()
        gpw = 256 // self.D
        gmax = np.minimum(rows, b_local * world)
        self.ux = SimpleNamespace(
            world=int(world), b=int(b_local), goff_np=goff, goff=torch.tensor(goff, **i32), capT=capT, KS=KS, local=local,
            keys=keys, src=[], new_local=new_local, parts=parts,
            part_counts=torch.zeros(_lib.ADAM_WINDOW_MAX * self.F * parts, **i32),
            max_units=int(((gmax + gpw - 1) // gpw).sum()), max_unique=int(gmax.max()), max_entries=int(caps.max()) * int(world))
        self.ux_src_bufs(len(self.sortbufs))
        return self.ux

    def ux_src_bufs(self, k):
        """src [N][F, stride] of window positions 0 .. k-1 (allocated outside graph capture, like the sort workspaces)."""
        ux = self.ux
        while len(ux.src) < k:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.RsxError("EmbeddingArena.ux_src_bufs(%d): allocate before capture" % k)
            ux.src.append(torch.full((ux.world, self.F * self.stride), -1, dtype=torch.int32, device=self.tables.device))
        return ux.src[:k]

    def ux_sort_pack(self, ids_list):
        """The rank's OWN dedup sorts of the k batches of an optimizer window (k = 1: a single step) + their key blocks
        [nuniq | unique rows packed at goff] -> keys [1, k * KS] int32, the input of the ids-phase all-gather."""
        ux, k = self.ux, len(ids_list)
        loc = ux.local
        if k == 1 and ids_list[0].shape[0] > loc.LDS_SORT_MAX_B:
            loc.select(0)
            loc.field_sort(ids_list[0])
        else:
            loc.sort_window(ids_list)
        return self.ux_pack(k)

    def ux_peer_keys(self, ids):
        """EmulatedDataParallel: the packed key block [KS] a PEER rank holding batch `ids` would send (its dedup sort + pack, run
        on a scratch workspace so that this rank's own sort state is untouched)."""
        ux = self.ux
        if not hasattr(ux, "scratch"):
            ux.scratch = ux.new_local()
        own = (ux.local, ux.keys)
        ux.local, ux.keys = ux.scratch
        try:
            out = self.ux_sort_pack([ids]).reshape(-1).clone()
        finally:
            ux.local, ux.keys = own
        return out

    def ux_pack(self, k):
        ux, loc = self.ux, self.ux.local
        jobs = (_lib.UniqPackJob * k)()
        for i, b in enumerate(loc.window_bufs(k)):
            jobs[i].uniq_row, jobs[i].nuniq, jobs[i].keys = b["uniq_row"].data_ptr(), b["nuniq"].data_ptr(), ux.keys[i].data_ptr()
        check(lib().rsx_uniq_pack(jobs, k, _ptr(ux.goff), _ptr(self.row_off), self.F, loc.stride, ux.parts,
                                  ux.max_entries // ux.world, _stream()), "rsx_uniq_pack")
        return ux.keys[:k].view(1, k * ux.KS)

    def ux_merge(self, keys_g, k):
        """keys_g [N, k * KS] (the all-gathered key blocks) -> window positions 0 .. k-1 of THIS arena's sort workspaces:
        global unique rows, slot maps, src (rsx_uniq_merge).  Leaves what rsx_field_sort would leave for the global batch,
        minus the per-example permutation nobody needs any more."""
        ux = self.ux
        assert keys_g.dtype == torch.int32 and keys_g.is_contiguous() and tuple(keys_g.shape) == (ux.world, k * ux.KS)
        jobs = (_lib.UniqMergeJob * k)()
        srcs = self.ux_src_bufs(k)
        for i, b in enumerate(self.window_bufs(k)):
            jobs[i].uniq_row, jobs[i].nuniq, jobs[i].slot = b["uniq_row"].data_ptr(), b["nuniq"].data_ptr(), b["slot"].data_ptr()
            jobs[i].src = srcs[i].data_ptr()
        check(lib().rsx_uniq_merge(_ptr(keys_g), k * ux.KS, ux.KS, jobs, k, _ptr(ux.goff), _ptr(self.row_off),
                                   _ptr(ux.part_counts), ux.parts, self.max_rows, ux.max_entries, self.F, ux.world, self.stride,
                                   _stream()), "rsx_uniq_merge")
        self.last_B = ux.max_unique

    def ux_segsum_local(self, B, S, dX, gy1, gy2, G_out, gw1_out, pos=0, null_row=-1):
        """The rank's own sorted segment-sum (window position pos of the LOCAL workspace) written as its block of the
        exchange: G_out [capT, D], gw1_out [capT] (views of the send block)."""
        loc = self.ux.local
        loc.select(pos)
        assert G_out.is_contiguous() and G_out.shape[0] >= self.ux.capT
        part = loc._stage_a(B, S, dX, gy1, gy2)
        w = gy1 is not None and gw1_out is not None
        check(lib().rsx_segsum_bwd_packed(_ptr(self.tables), _ptr(S), _ptr(dX), _ptr(gy1) if w else None, _ptr(gy2),
                                          _ptr(loc.perm), _ptr(loc.seg_off), _ptr(loc.uniq_row), _ptr(loc.nuniq), _ptr(G_out),
                                          _ptr(gw1_out) if w else None, self.w1_mask, B, self.F, self.D, loc.stride, null_row,
                                          part, _ptr(self.ux.goff), _stream()), "rsx_segsum_bwd_packed")

    def ux_merged_adam(self, G0, gw10, rank_stride, opt, extra_segments, sweep=None, advance=True, second=None, window=None,
                       w1_ext=None):
        """The optimizer launch of the exchange (rsx_merged_adam_rows): G0 / gw10 = rank 0's blocks inside the gathered buffer,
        rank r's rank_stride floats further.  second = (arena2, G2_0): a table set sharing this arena's lists.
        window = (k, cur) as segsum_adam."""
        ux = self.ux
        arr, n = opt._seg_array(extra_segments)
        lr, b1, b2, eps = opt.hp
        win = None
        if window is not None and window[0] > 1:
            assert self.cur_buf == window[1]
            win = self.window(*window)
        sec = None
        if second is not None:
            a2, G20 = second
            sec_obj = _lib.TableSet(a2.tables.data_ptr(), a2.m_t.data_ptr(), a2.v_t.data_ptr(), G20.data_ptr(), None)
            sec = C.byref(sec_obj)
        w = self.with_w1 and gw10 is not None
        w1p, mwp, vwp, w1s, w1sp = (_ptr(self.w1), _ptr(self.m_w), _ptr(self.v_w), 1, 0) if w else (None, None, None, 1, 0)
        if w1_ext is not None:
            w1p, mwp, vwp, w1s, w1sp = _ptr(w1_ext[0]), _ptr(w1_ext[1]), _ptr(w1_ext[2]), int(w1_ext[3]), int(w1_ext[4])
            assert gw10 is not None
        check(lib().rsx_merged_adam_rows(_ptr(self.tables), _ptr(self.m_t), _ptr(self.v_t), w1p, mwp, vwp, _ptr(G0),
                                         _ptr(gw10) if w1p is not None else None, int(rank_stride), ux.world,
                                         _ptr(ux.src[self.cur_buf]), _ptr(ux.goff), _ptr(self.uniq_row), _ptr(self.nuniq),
                                         self.w1_mask, ux.max_units, self.F, self.D, self.stride, arr, n,
                                         None if sweep is None else C.byref(sweep), sec, None if win is None else C.byref(win),
                                         _ptr(opt.state), 1 if advance else 0, lr, b1, b2, eps, w1s, w1sp, _stream()),
              "rsx_merged_adam_rows")

    # -- kernels ---------------------------------------------------------------------------
    def field_sort_t(self, ids_t, B):
        """field_sort for keys that are already FIELD-MAJOR: ids_t int32 [F, stride] (field f's B keys in row f), large sorts
        only (rsx_field_sort_large_t: no transpose launch); ids_t is clobbered."""
        assert ids_t.dtype == torch.int32 and ids_t.is_contiguous() and tuple(ids_t.shape) == (self.F, self.stride)
        assert self.LDS_SORT_MAX_B < B <= self.stride and getattr(self, "_sort_owner", None) is None
        check(lib().rsx_field_sort_large_t(_ptr(ids_t), _ptr(self.row_off), _ptr(self.perm), _ptr(self.seg_off),
                                           _ptr(self.uniq_row), _ptr(self.nuniq), _ptr(self.slot), _ptr(self.segid),
                                           _ptr(self.sort_ws), self.max_rows, B, self.F, self.stride, _stream()),
              "rsx_field_sort_large_t")
        self.last_B = B

    def field_sort(self, ids):
        B = ids.shape[0]
        assert ids.dtype == torch.int32 and ids.is_contiguous() and ids.shape[1] == self.F and B <= self.stride
        owner = getattr(self, "_sort_owner", None)
        if owner is not None:
            assert owner.last_B == B, "shared sort: sort the owning arena first"
            self.last_B = B
            return
        if B > self.LDS_SORT_MAX_B:
            check(lib().rsx_field_sort_large(_ptr(ids), _ptr(self.row_off), _ptr(self.perm), _ptr(self.seg_off),
                                             _ptr(self.uniq_row), _ptr(self.nuniq), _ptr(self.slot), _ptr(self.segid),
                                             _ptr(self.sort_ws), self.max_rows, B, self.F, self.stride, _stream()),
                  "rsx_field_sort_large")
        else:
            check(lib().rsx_field_sort(_ptr(ids), _ptr(self.row_off), _ptr(self.perm), _ptr(self.seg_off),
                                       _ptr(self.uniq_row), _ptr(self.nuniq), _ptr(self.slot),
                                       _ptr(self.segid) if self._two_stage(B) else None, self.max_rows,
                                       B, self.F, self.stride, _stream()), "rsx_field_sort")
        self.last_B = B

    def sort_window(self, ids_list):
        """The dedup sorts of the k batches of an optimizer window in ONE launch (rsx_field_sort_multi), batch i into
        sort workspace i; leaves workspace 0 selected."""
        k = len(ids_list)
        jobs = (_lib.SortJob * k)()
        for i, ids in enumerate(ids_list):
            self.select(i)
            jobs[i] = self.sort_job(ids)
        self.select(0)
        if _lib.form("win_separate_sorts") == "1":      # debugging aid: k launches of the single-sort kernel
            for i, ids in enumerate(ids_list):
                self.select(i)
                self.field_sort(ids)
            self.select(0)
            return
        check(lib().rsx_field_sort_multi(jobs, k, _stream()), "rsx_field_sort_multi")

    def window(self, k, cur):
        """rsx_adam_window of position `cur` in a window of k steps (sort_window ran at its start)."""
        if k <= 1:
            return None
        w = _lib.AdamWindow()
        w.k, w.cur, w.max_unique = k, cur, self.last_B
        owner = getattr(self, "_sort_owner", None) or self
        for i, b in enumerate(owner.window_bufs(k)):
            w.uniq_row[i], w.nuniq[i], w.slot[i] = b["uniq_row"].data_ptr(), b["nuniq"].data_ptr(), b["slot"].data_ptr()
        return w

    def sort_job(self, ids):
        """The rsx_field_sort call of this step as a job another launch can carry (FusedTower.train_step)."""
        B = ids.shape[0]
        assert ids.dtype == torch.int32 and ids.is_contiguous() and ids.shape[1] == self.F and B <= self.stride
        self.last_B = B
        j = _lib.SortJob()
        j.ids, j.row_off, j.perm, j.seg_off = ids.data_ptr(), self.row_off.data_ptr(), self.perm.data_ptr(), self.seg_off.data_ptr()
        j.uniq_row, j.nuniq, j.slot = self.uniq_row.data_ptr(), self.nuniq.data_ptr(), self.slot.data_ptr()
        j.segid = self.segid.data_ptr() if self._two_stage(B) else None
        j.max_rows_per_field, j.B, j.F, j.stride = self.max_rows, B, self.F, self.stride
        j.skip_mask = 0
        return j

    def gather_outputs(self, B, fm=False, first_order=False, S_out=None):
        """The output buffers of gather(): E [B, F*D], S [B,D]|None, y1 [B]|None, y2 [B]|None."""
        dev = self.tables.device
        E = torch.empty(B, self.F * self.D, device=dev)
        S = (S_out if S_out is not None else torch.empty(B, self.D, device=dev)) if fm else None
        y2 = torch.empty(B, device=dev) if fm else None
        y1 = torch.empty(B, device=dev) if first_order else None
        return E, S, y1, y2

    def gather(self, ids, fm=False, first_order=False, S_out=None, sort_job=None):
        """-> E [B, F*D], S [B,D]|None, y1 [B]|None, y2 [B]|None (no autograd).  S_out: caller-owned [B,D] buffer for S
        (e.g. a view of the data-parallel send block).  sort_job: the step's dedup sort (EmbeddingArena.sort_job) rides in
        this launch as extra workgroups (rsx_gather_fm_fwd_sort)."""
        B = ids.shape[0]
        E, S, y1, y2 = self.gather_outputs(B, fm, first_order, S_out)
        if sort_job is not None:
            check(lib().rsx_gather_fm_fwd_sort(_ptr(self.tables), _ptr(self.w1) if first_order else None, _ptr(self.row_off),
                                               _ptr(ids), _ptr(E), _ptr(S), _ptr(y1), _ptr(y2), self.w1_mask,
                                               B, self.F, self.D, C.byref(sort_job), _stream()), "rsx_gather_fm_fwd_sort")
            return E, S, y1, y2
        check(lib().rsx_gather_fm_fwd(_ptr(self.tables), _ptr(self.w1) if first_order else None, _ptr(self.row_off),
                                      _ptr(ids), _ptr(E), _ptr(S), _ptr(y1), _ptr(y2), self.w1_mask,
                                      B, self.F, self.D, _stream()), "rsx_gather_fm_fwd")
        return E, S, y1, y2

    @staticmethod
    def _blocks(blocks):
        """blocks = (examples per rank block, floats between blocks) for inputs read in place from an all-gathered buffer
        (dist.DataParallel.gather_example_grads(blocked=True)); None = contiguous."""
        return None if blocks is None else C.byref(_lib.ExampleBlocks(int(blocks[0]), int(blocks[1])))

    def _stage_a(self, B, S, dX, gy1, gy2, blk=None, riders=None):
        """Stage A of the two-stage scatter; returns the partials handle stage B takes (None: single stage).
        riders (_lib.ScatterRiders or None): reductions of the step's other launches that only the optimizer reads, carried by
        this launch as extra workgroups (rsx_segsum_partials_ride); without a stage A they run as their own launches here."""
        if not self._two_stage(B):
            run_scatter_riders(riders)
            return None
        check(lib().rsx_segsum_partials_ride(_ptr(self.tables), _ptr(S), _ptr(dX), _ptr(gy1), _ptr(gy2), _ptr(self.perm),
                                             _ptr(self.seg_off), _ptr(self.uniq_row), C.byref(self.partials), self.w1_mask,
                                             B, self.F, self.D, self.stride, -1, blk,
                                             None if riders is None else C.byref(riders), _stream()), "rsx_segsum_partials_ride")
        return C.byref(self.partials)

    def segsum(self, B, S, dX, gy1, gy2, blocks=None):
        blk = self._blocks(blocks)
        part = self._stage_a(B, S, dX, gy1, gy2, blk)
        check(lib().rsx_segsum_bwd(_ptr(self.tables), _ptr(S), _ptr(dX), _ptr(gy1), _ptr(gy2), _ptr(self.perm),
                                   _ptr(self.seg_off), _ptr(self.uniq_row), _ptr(self.nuniq), _ptr(self.G),
                                   _ptr(self.gw1) if gy1 is not None else None, self.w1_mask, B, self.F, self.D,
                                   self.stride, part, blk, _stream()), "rsx_segsum_bwd")

    def segsum_adam(self, B, S, dX, gy1, gy2, opt, extra_segments, sweep=None, blocks=None, advance=True, second=None,
                    window=None, w1_ext=None, riders=None):
        """Segment-sum + touched-row Adam in one launch (+ `extra_segments`, e.g. the dense arena, as extra workgroups).
        second = (arena2, dX2): a table set sharing this arena's sort (share_sort_of), updated by the same launch.
        window = (k, cur): this step is position `cur` of an optimizer window of k steps (the current sort workspace must
        be `cur`): the rows the window's other steps touch get this step's untouched-row update in the same launch."""
        blk = self._blocks(blocks)
        sec = None
        if second is not None:
            a2, dX2 = second
            assert getattr(a2, "_sort_owner", None) is self and a2.D == self.D and dX2.is_contiguous()
            two = a2._stage_a(B, None, dX2, None, None, blk) is not None
            sec_obj = _lib.TableSet(a2.tables.data_ptr(), a2.m_t.data_ptr(), a2.v_t.data_ptr(), dX2.data_ptr(),
                                    C.addressof(a2.partials) if two else None)
            sec = C.byref(sec_obj)
        arr, n = opt._seg_array(extra_segments)
        lr, b1, b2, eps = opt.hp
        win = None
        if window is not None and window[0] > 1:
            assert self.cur_buf == window[1], "segsum_adam: select() the window position's sort workspace first"
            win = self.window(*window)
        w = self.with_w1 and gy1 is not None
        w1p, mwp, vwp, w1s, w1sp = (_ptr(self.w1), _ptr(self.m_w), _ptr(self.v_w), 1, 0) if w else (None, None, None, 1, 0)
        if w1_ext is not None:          # (w1, m, v, stride, sparse formula): a first-order vector stored outside this arena
            w1p, mwp, vwp, w1s, w1sp = _ptr(w1_ext[0]), _ptr(w1_ext[1]), _ptr(w1_ext[2]), int(w1_ext[3]), int(w1_ext[4])
            assert gy1 is not None
        part = self._stage_a(B, S, dX, gy1, gy2, blk, riders)
        check(lib().rsx_segsum_adam_rows2(_ptr(self.tables), _ptr(self.m_t), _ptr(self.v_t), w1p, mwp, vwp, _ptr(S), _ptr(dX),
                                          _ptr(gy1), _ptr(gy2), _ptr(self.perm), _ptr(self.seg_off), _ptr(self.uniq_row),
                                          _ptr(self.nuniq), self.w1_mask, B, self.F, self.D, self.stride, arr, n,
                                          None if sweep is None else C.byref(sweep), part, blk, sec,
                                          None if win is None else C.byref(win),
                                          _ptr(opt.state), 1 if advance else 0, lr, b1, b2, eps, w1s, w1sp, _stream()),
              "rsx_segsum_adam_rows2")

    # -- optimizer segments ----------------------------------------------------------------
    def adam_split_segments(self, window_k=1):
        """(cold, hot): the exact TF-1 update split into the untouched-row sweep (depends only on the slot map, may
        overlap the tower) and the touched rows (needs this step's gradients).  window_k > 1: the sweep of a whole optimizer
        window (the rows none of its k steps touches, k updates in one pass; call at window position 0)."""
        slot_w = None
        if window_k > 1:
            assert self.cur_buf == 0
            owner = getattr(self, "_sort_owner", None) or self
            slot_w = [b["slot"] for b in owner.window_bufs(window_k)[1:]]
        cold = [dict(kind=_lib.RSX_ADAM_TABLE_TF1_COLD, d=self.D, n=self.R, var=self.tables, m=self.m_t, v=self.v_t,
                     slot=self.slot, slot_w=slot_w)]
        hot = [dict(kind=_lib.RSX_ADAM_TABLE_ROWS, d=self.D, n=self.F * self.last_B, var=self.tables, m=self.m_t,
                    v=self.v_t, g=self.G, uniq_row=self.uniq_row, nuniq=self.nuniq, B=self.last_B, stride=self.stride)]
        if self.with_w1:
            cold.append(dict(kind=_lib.RSX_ADAM_VEC_COLD, n=self.R, var=self.w1, m=self.m_w, v=self.v_w, slot=self.slot,
                             slot_w=slot_w))
            hot.append(dict(kind=_lib.RSX_ADAM_VEC_ROWS_DENSE, n=self.F * self.last_B, var=self.w1, m=self.m_w, v=self.v_w,
                            g=self.gw1, uniq_row=self.uniq_row, nuniq=self.nuniq, B=self.last_B, stride=self.stride))
        return cold, hot

    def adam_segments(self, lazy=False):
        slot = self.slot
        segs = []
        if lazy:
            segs.append(dict(kind=_lib.RSX_ADAM_TABLE_ROWS, d=self.D, n=self.F * self.last_B, var=self.tables,
                             m=self.m_t, v=self.v_t, g=self.G, uniq_row=self.uniq_row, nuniq=self.nuniq,
                             B=self.last_B, stride=self.stride))
            if self.with_w1:
                segs.append(dict(kind=_lib.RSX_ADAM_VEC_ROWS, n=self.F * self.last_B, var=self.w1, m=self.m_w,
                                 v=self.v_w, g=self.gw1, uniq_row=self.uniq_row, nuniq=self.nuniq,
                                 B=self.last_B, stride=self.stride))
        else:
            segs.append(dict(kind=_lib.RSX_ADAM_TABLE_TF1, d=self.D, n=self.R, var=self.tables, m=self.m_t,
                             v=self.v_t, g=self.G, slot=slot))
            if self.with_w1:
                segs.append(dict(kind=_lib.RSX_ADAM_VEC_SLOT, n=self.R, var=self.w1, m=self.m_w, v=self.v_w,
                                 g=self.gw1, slot=slot))
        return segs


class GatherFM(torch.autograd.Function):
    """Autograd node of the fused gather (+first-order +FM2).  backward = sorted segment-sum into the
    arena's sparse-gradient workspace (the IndexedSlices of TF); returns no tensor gradients."""

    @staticmethod
    def forward(ctx, hook, arena, ids, fm, first_order, dp=None):
        E, S, y1, y2 = arena.gather(ids, fm, first_order)
        ctx.arena, ctx.B, ctx.S, ctx.dp = arena, ids.shape[0], S, dp
        ctx.fm, ctx.fo = fm, first_order
        outs = [E]
        if first_order:
            outs.append(y1)
        if fm:
            outs.append(y2)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        gE = grads[0]
        k = 1
        gy1 = gy2 = None
        if ctx.fo:
            gy1 = grads[k]
            k += 1
        if ctx.fm:
            gy2 = grads[k]
        gE = None if gE is None else gE.contiguous()
        gy1 = None if gy1 is None else gy1.contiguous()
        gy2 = None if gy2 is None else gy2.contiguous()
        if ctx.fo and gy1 is None:
            gy1 = torch.zeros(ctx.B, device=ctx.arena.tables.device)
        S = ctx.S if gy2 is not None else None
        B = ctx.B
        if ctx.dp is not None:
            # data parallel: one packed all-gather of the per-example gradient block, then the same sorted
            # segment-sum over the GLOBAL batch on every rank (arena.field_sort saw the all-gathered ids)
            if gE is None:
                gE = torch.zeros(B, ctx.arena.F * ctx.arena.D, device=ctx.arena.tables.device)
            gE, S, gy1, gy2 = ctx.dp.gather_example_grads(gE, S, gy1, gy2)
            B = gE.shape[0]
        ctx.arena.segsum(B, S, gE, gy1, gy2)
        return None, None, None, None, None, None


def gather_fm(arena, ids, fm=False, first_order=False, dp=None):
    return GatherFM.apply(arena.hook, arena, ids, fm, first_order, dp)


class DenseArena:
    """All dense variables of a model as views into ONE flat fp32 buffer (and their grads / Adam slots
    likewise), so the optimizer sweeps them as a single segment."""

    def __init__(self, shapes, device="cuda", storage_shapes=None):
        """storage_shapes: optional {name: padded shape}: the variable is STORED with that (larger) shape and exposed as the
        leading slice of the logical shape (e.g. a [100, 50] kernel stored as [100, 52] so that a fused kernel sees row
        strides that are multiples of 4).  Pad elements start at zero and, receiving zero gradients, stay zero."""
        dev = _require_cuda(device)
        self.names = list(shapes)
        storage_shapes = storage_shapes or {}
        self.storage = {k: tuple(storage_shapes.get(k, shapes[k])) for k in self.names}
        self.offsets = {}
        n = 0
        for k in self.names:
            self.offsets[k] = n
            n += int(np.prod(self.storage[k])) if len(self.storage[k]) else 1
            n = (n + 3) & ~3                       # keep every variable 16-byte aligned
        self.n = n
        self.flat = torch.zeros(n, device=dev)
        self.grad = torch.zeros(n, device=dev)
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.params = {}
        for k in self.names:
            st = self.storage[k]
            sz = int(np.prod(st)) if len(st) else 1
            o = self.offsets[k]
            sl = tuple(slice(0, d) for d in shapes[k])
            assert len(st) == len(shapes[k]) and all(a >= b for a, b in zip(st, shapes[k]))
            p = self.flat[o:o + sz].view(st)[sl].requires_grad_()
            p.grad = self.grad[o:o + sz].view(st)[sl]
            self.params[k] = p

    def __getitem__(self, k):
        return self.params[k]

    def load(self, values):
        with torch.no_grad():
            for k, v in values.items():
                if k in self.params:
                    self.params[k].copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32).reshape(self.params[k].shape))

    def adam_segments(self):
        return [dict(kind=_lib.RSX_ADAM_DENSE, n=self.n, var=self.flat, m=self.m, v=self.v, g=self.grad,
                     zero_grad=1)]

    def rebind_grad(self, storage):
        """Moves the gradient arena into caller-owned storage (flat fp32, >= n): the data-parallel send block keeps it in
        front of the per-example gradient block so that ONE buffer goes into the collective without a pack copy."""
        assert storage.is_contiguous() and storage.numel() >= self.n and storage.dtype == torch.float32
        g = storage[:self.n]
        g.copy_(self.grad)
        self.grad = g
        for k, p in self.params.items():
            st = self.storage[k]
            sz = int(np.prod(st)) if len(st) else 1
            o = self.offsets[k]
            p.grad = g[o:o + sz].view(st)[tuple(slice(0, d) for d in p.shape)]

    def packed_grad(self, names):
        """The grad-arena slice covering `names` when they sit back to back without padding (every size but the last a
        multiple of 4) -- a kernel that emits their gradients as one flat array can then write straight into it.  None if
        they are not packed."""
        o = self.offsets[names[0]]
        e = o
        for k in names:
            if self.offsets[k] != e or tuple(self.params[k].shape) != self.storage[k]:
                return None
            e += self.params[k].numel()
        return self.grad[o:e]


class AdamTF1:
    """tf.train.AdamOptimizer (fm/fm.py:162): one rsx_adam_tf1_multi launch per step over every segment."""

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, device="cuda"):
        dev = _require_cuda(device)
        self.hp = (float(lr), float(beta1), float(beta2), float(eps))
        st = np.zeros(_lib.ADAM_STATE_WORDS, np.float32)     # words 0..3 + the arrival counters (zero)
        check(lib().rsx_adam_state_init_h(st.ctypes.data_as(C.c_void_p), beta1, beta2))
        self.state = torch.from_numpy(st).to(dev)

    @staticmethod
    def _seg_array(segments):
        n = len(segments)
        arr = (AdamSeg * n)()
        for i, s in enumerate(segments):
            a = arr[i]
            a.kind, a.d, a.n = s["kind"], s.get("d", 0), int(s["n"])
            for f in ("var", "m", "v", "g", "slot", "uniq_row", "nuniq"):
                t = s.get(f)
                setattr(a, f, None if t is None else t.data_ptr())
            a.B, a.stride, a.zero_grad = s.get("B", 0), s.get("stride", 0), s.get("zero_grad", 0)
            a.g_replicas, a.g_replica_stride = int(s.get("g_replicas", 0)), int(s.get("g_replica_stride", 0))
            for j, t in enumerate(s.get("slot_w") or ()):
                a.slot_w[j] = t.data_ptr()
        return arr, n

    def step(self, segments):
        arr, n = self._seg_array(segments)
        lr, b1, b2, eps = self.hp
        check(lib().rsx_adam_tf1_multi(arr, n, _ptr(self.state), lr, b1, b2, eps, _stream()), "rsx_adam_tf1_multi")

    def cold_slices(self, cold_segments, weights, window_block_u=0):
        """Cuts the untouched-row sweep (COLD kinds) into len(weights) consecutive workgroup ranges, proportional to
        `weights`, as rsx_adam_slice structs for the tower launches.  Returns a list (entries may be None).
        window_block_u (2 / 4): a WINDOW sweep (segments with slot_w) in smaller table blocks, one slice per step of the
        window -- slice i is meant to ride in step i's head launch; slices after the first read the window's step sizes from the
        optimizer state (rsx_adam_slice.alphas_from_state)."""
        if getattr(self, "shadow", False):      # dist.LoopbackDataParallel: a shadow rank's pass changes no optimizer state
            return [None] * len(weights)
        arr, n = self._seg_array(cold_segments)
        nb = int(lib().rsx_adam_num_blocks_u(arr, n, int(window_block_u)))
        if nb < 0:
            check(nb, "rsx_adam_num_blocks")
        lr, b1, b2, eps = self.hp
        tot = float(sum(weights))
        out, lo, acc = [], 0, 0.0
        for i, w in enumerate(weights):
            acc += w
            hi = nb if i == len(weights) - 1 else min(nb, int(round(nb * acc / tot)))
            if hi > lo:
                sl = _lib.AdamSlice()
                sl.segs, sl.nseg, sl.lr, sl.beta1, sl.beta2, sl.eps = arr, n, lr, b1, b2, eps
                sl.state, sl.blk_lo, sl.blk_hi = self.state.data_ptr(), lo, hi
                sl.window_block_u = int(window_block_u)
                sl.alphas_from_state = 1 if (window_block_u and lo > 0) else 0
                sl._keep = arr
                out.append(sl)
            else:
                out.append(None)
            lo = hi
        return out

    def run_slice(self, sl):
        if sl is None:
            return
        check(lib().rsx_adam_slice_run(C.byref(sl), _stream()), "rsx_adam_slice_run")

    def window_sweep(self, cold_segments):
        """The ONE untouched-row sweep of an optimizer window (COLD segments with slot_w), a launch of its own on the step's
        stream.  (Measured and rejected, r02: the same launch on a forked stream inside the graph, concurrent with the
        window's steps -- it only touches rows none of them touches.  At full width it stretches the concurrent tower
        launches 5-9x (0.082 ms per step against 0.075); throttled to 128 workgroups it spreads over the whole window and
        gains 3 %, within the box-to-box noise, for a second stream, a snapshot of the beta powers and a join per window.)"""
        self.run_slice(self.cold_slices(cold_segments, [1.0])[0])

    @property
    def global_step(self):
        return int(self.state.view(torch.int32)[3].item()) - 1


class FusedTower:
    """TRAIN-step driver of the fused HIP tower (csrc/tower.hip): L x [dense(relu) -> BN -> dropout], the
    1-unit output layer, the logits head and the mean sigmoid-CE loss, forward AND backward, in 2L+1
    launches.  Gradients of the dense variables are written straight into the DenseArena's flat grad
    buffer; dX / gs0 / gs1 (gradients of the tower inputs) are returned for the embedding scatter.
    Mirrors deepfm/deepfm.py:100-112 (`dnn` scope + logits) with fm/fm.py:146-149 (loss)."""

    @staticmethod
    def supports(k0, widths):
        """The fused kernels' envelope; model code checks it BEFORE creating any variable and otherwise takes the
        autograd path (`tower='torch'`), so every --deep_layers the reference accepts still trains."""
        widths = [int(w) for w in widths]
        return bool(widths) and int(k0) % 4 == 0 and all(w % 4 == 0 for w in widths[:-1]) and widths[-1] <= 256

    def __init__(self, dense, pre, k0, widths, capacity, device="cuda", batch_norm=True):
        """batch_norm=False: L x [dense(relu) -> dropout] (din/din.py:132-137), no gamma/beta variables."""
        dev = _require_cuda(device)
        self.bn_on = bool(batch_norm)
        self.P, self.pre, self.k0, self.widths = dense, pre, int(k0), [int(w) for w in widths]
        if not FusedTower.supports(self.k0, self.widths):
            raise _lib.RsxError("FusedTower envelope: widths multiple of 4, last width <= 256 (use tower='torch')")
        self.cap = int(capacity)
        B, RT = self.cap, (self.cap + 15) // 16
        f64 = dict(dtype=torch.float64, device=dev)
        self.a = [torch.empty(B, n, device=dev) for n in self.widths]
        self.dy = [torch.empty(B, n, device=dev) for n in self.widths]
        self.fstat = [torch.zeros(RT, 2, n, **f64) for n in self.widths]
        self.bstat = [torch.zeros(RT, 2, n, **f64) for n in self.widths]
        # batches > 512 (include/rsx.h RSX_TOWER_FIXED_STATS_MIN_B): fixed-point accumulators per statistics buffer,
        # int64 [8, 4, n rounded up to 16] viewed as doubles, all in one allocation ordered [bstat_0 | fstat_0 .. fstat_{L-1} | bstat_1 .. bstat_{L-1}]:
        # the first layer's forward launch clears bstat_0 for its step, the first layer's backward launch everything else for the
        # next one (two contiguous ranges)
        self.fix = self.fix_eval = None
        if self.cap > 512:
            sz = [32 * ((n + 15) // 16 * 16) for n in self.widths]          # RSX_TOWER_FIXED_STATS_DOUBLES(n)

            def fixed_rows():
                flat = torch.zeros(2 * sum(sz), **f64)
                o, fs, bs = sz[0], [], [flat[:sz[0]]]
                for k in sz:
                    fs.append(flat[o:o + k])
                    o += k
                for k in sz[1:]:
                    bs.append(flat[o:o + k])
                    o += k
                return flat, fs, bs
            self.fix, self.fstat_fix, self.bstat_fix = fixed_rows()
            self.fix_eval, self.fstat_fix_eval, self.bstat_fix_eval = fixed_rows()     # infer's producers add here (never read)
        self.bn = [torch.zeros(2, n, device=dev) for n in self.widths]
        self.mask_flat = torch.ones(B * sum(self.widths), device=dev)
        self.dX = torch.empty(B, self.k0, device=dev)
        self.prob = torch.empty(B, device=dev)
        self.gs0 = torch.empty(B, device=dev)
        self.gs1 = torch.empty(B, device=dev)
        self.dwd_part = torch.zeros(RT, self.widths[-1], device=dev)
        self.hpart = torch.zeros(RT, 8, **f64)
        self._eval_stats = {}                       # batch size -> per-layer statistics that put BN in inference form (infer)
        self._zero_labels = torch.zeros(B, device=dev)
        self.loss = torch.zeros(1, device=dev)
        # large batches: every dW tile's reduction over the batch is split into row blocks (csrc/tower.hip)
        self.dwp = []
        for l, n in enumerate(self.widths):
            K = self.k0 if l == 0 else self.widths[l - 1]
            self.dwp.append(torch.empty(int(lib().rsx_tower_bwd_workspace_floats(self.cap, K, n)), device=dev)
                            if self.cap >= 1024 else None)

    @staticmethod
    def fused_gather_ok(arena, B):
        """The envelope of the gather riding in the first forward launch (RSX_FUSE_GATHER=0: the two launches, A/B runs)."""
        return _lib.form("fuse_gather") != "0" and \
            bool(lib().rsx_gather_tower_fwd0_supported(int(B), int(arena.F), int(arena.D)))

    def _masks(self, B, rate, masks):
        """Injected keep-masks (parity tests) as contiguous [B, N_l] views; None -> the kernels derive the mask
        from the counter-based hash (seed, step, layer, element), no buffer and no launch."""
        if rate == 0.0 or masks is None:
            return [None] * len(self.widths)
        out, o = [], 0
        for i, n in enumerate(self.widths):
            v = self.mask_flat[o:o + B * n].view(B, n)
            v.copy_(masks[i])
            out.append(v)
            o += B * n
        return out

    def train_step(self, X, labels, rate, rng_step, s0=None, c0=None, s1=None,
                   head=("dnn.Wout", "dnn.bout", "out.W", "out.b"), relu0=True, relu2=True, replicas=1, masks=None,
                   seed=0x5eed, sort_job=None, sweeps=None, sort_in_fwd=False, outs=None, layer_done=None, gather=None,
                   reduce_rider=False, defer_dw_reduce=False, cross_rider=None):
        """X [B,k0]; s0/s1 [B] extra scalar inputs of the head (first-order pre-activation, FM term);
        c0 = name of the bias added to s0; rng_step = device uint32 tensor that changes every step;
        sort_job = EmbeddingArena.sort_job(ids): the dedup sort rides in the last layer's backward launch, or in the
        FIRST layer's forward launch when sort_in_fwd (required when sweeps are used: they read its slot map).
        sweeps = 2L+1 rsx_adam_slice structs (or None) for [fwd_0..fwd_{L-1}, head, bwd_{L-1}..bwd_0]: slices of the
        untouched-row optimizer sweep that ride along as extra workgroups (AdamTF1.cold_slices).
        outs = (dX [B,k0], gs0 [B], gs1 [B]): caller-owned contiguous output buffers (views of the data-parallel send block)
        instead of the tower's own.
        gather = (arena, ids, S, y1, y2): X is the arena's (still empty) E buffer and the FIRST forward launch fills it -- the
        input_layer lookup + first-order sum + FM term ride in that launch (rsx_gather_tower_fwd0; fused_gather_ok says when).
        layer_done(l): called after layer l's backward launch (l = L-1 .. 0), when that layer's dW/db (and, for l = L-1,
        the head's gradients) have been launched in full -- the hook of the RSX_DP_OVERLAP all-reduces.  The layers' dW
        reductions are then NOT deferred to the end.
        cross_rider (round 6, dcn.py; CrossLayers.rider_args): the cross layers' backward runs as extra workgroups of the LAST
        layer's backward launch and the first layer's launch accumulates onto the dX it wrote (include/rsx.h rsx_tower_bwd_extra;
        the caller checked cross_ride_ok); its gradient-partials reduce comes back as self.cross_job_pending.
        Returns (loss [1], prob [B], dX [B,k0], gs0 [B], gs1 [B])."""
        L, P, pre = lib(), self.P, self.pre
        self.cross_job_pending = None
        o_dX, o_gs0, o_gs1 = [o if o is not None else d for o, d in zip(outs or (None,) * 3, (self.dX, self.gs0, self.gs1))]
        B = X.shape[0]
        assert B <= self.cap and X.is_contiguous() and X.shape[1] == self.k0
        st = _stream()
        mk = self._masks(B, rate, masks)
        nl = len(self.widths)
        self.mlp_reduce_job = None
        self.dw_jobs_pending = []         # (every path: a caller that defers reads it after ANY step, the one-launch form included)
        # Round 4: a tower WITHOUT batch-norm (din.py's 'mlp_layer') runs forward + loss + backward as ONE launch + one reduce
        # (csrc/mlp_fused.hip, rsx_mlp_nobn_train_step) instead of 2L + 2 launches; RSX_MLP_FUSE=0: the launch-per-layer form
        if (not self.bn_on and self._mlp_fused_ok() and sort_job is None and sweeps is None and gather is None
                and layer_done is None and s1 is None and c0 is None and not relu0 and not relu2
                and head[2] is None and head[3] is None and isinstance(head[0], str) and isinstance(head[1], str)):
            ms = _lib.MlpStep()
            ms.X, ms.B, ms.K0, ms.L = X.data_ptr(), B, self.k0, nl
            for l in range(nl):
                ms.W[l], ms.b[l] = P[f"{pre}.W{l}"].data_ptr(), P[f"{pre}.b{l}"].data_ptr()
                ms.dW[l], ms.db[l] = P[f"{pre}.W{l}"].grad.data_ptr(), P[f"{pre}.b{l}"].grad.data_ptr()
                ms.masks[l] = None if mk[l] is None else mk[l].data_ptr()
                ms.widths[l] = self.widths[l]
            ms.wout, ms.bout = P[head[0]].data_ptr(), P[head[1]].data_ptr()
            ms.dwout, ms.dbout = P[head[0]].grad.data_ptr(), P[head[1]].grad.data_ptr()
            ms.s0 = None if s0 is None else s0.data_ptr()
            ms.labels, ms.rng_step = labels.data_ptr(), rng_step.data_ptr()
            ms.prob, ms.dX, ms.gs0 = self.prob.data_ptr(), o_dX.data_ptr(), None if s0 is None else o_gs0.data_ptr()
            ms.workspace, ms.loss = self._mlp_ws.data_ptr(), self.loss.data_ptr()
            ms.seed, ms.dropout_rate, ms.loss_scale = seed, rate, 1.0 / (B * replicas)
            assert labels.is_contiguous() and o_dX.is_contiguous() and (s0 is None or s0.is_contiguous())
            # reduce_rider: the caller hands self.mlp_reduce_job to a later launch of the step that carries the reduce as extra
            # workgroups (din.py: rsx_din_pool_bwd_pair_ride)
            ms.defer_reduce = 1 if reduce_rider else 0
            check(L.rsx_mlp_nobn_train_step(C.byref(ms), st), "rsx_mlp_nobn_train_step")
            self.mlp_reduce_job = None
            if reduce_rider:
                self.mlp_reduce_job = _lib.MlpReduceJob()
                check(L.rsx_mlp_nobn_reduce_job(C.byref(ms), C.byref(self.mlp_reduce_job)), "rsx_mlp_nobn_reduce_job")
            return self.loss, self.prob[:B], o_dX[:B], o_gs0[:B], o_gs1[:B]
        g = lambda name: P[name].grad
        big = B > 512                      # fixed-point statistics rows instead of row partials (include/rsx.h)
        fst = self.fstat_fix if big else self.fstat
        bst = self.bstat_fix if big else self.bstat
        z_fwd0 = (_ptr(bst[0]), bst[0].numel()) if big else (None, 0)
        z_bwd0 = (_ptr(self.fix[bst[0].numel():]), self.fix.numel() - bst[0].numel()) if big else (None, 0)
        bnp = (lambda name: _ptr(P[name])) if self.bn_on else (lambda name: None)        # gamma / beta (None: no batch-norm)
        bng = (lambda name: _ptr(P[name].grad)) if self.bn_on else (lambda name: None)
        rs = _ptr(rng_step)
        sw = list(sweeps) if sweeps is not None else [None] * (2 * nl + 1)
        ref = lambda x: None if x is None else C.byref(x)
        for l in range(nl):
            K = self.k0 if l == 0 else self.widths[l - 1]
            if l == 0 and gather is not None:
                ar, ids, gS, gy1, gy2 = gather
                assert ar.F * ar.D == self.k0 and ids.shape[0] == B
                check(L.rsx_gather_tower_fwd0(_ptr(ar.tables), _ptr(ar.w1) if gy1 is not None else None, _ptr(ar.row_off),
                                              _ptr(ids), _ptr(X), _ptr(gS), _ptr(gy1), _ptr(gy2), ar.w1_mask, ar.F, ar.D,
                                              _ptr(P[f"{pre}.W0"]), _ptr(P[f"{pre}.b0"]), _ptr(self.a[0]), _ptr(fst[0]),
                                              B, self.widths[0], ref(sort_job) if sort_in_fwd else None, ref(sw[0]),
                                              z_fwd0[0], z_fwd0[1], st),
                      "rsx_gather_tower_fwd0")
                continue
            check(L.rsx_tower_fwd_layer(_ptr(X if l == 0 else self.a[l - 1]), _ptr(P[f"{pre}.W{l}"]), _ptr(P[f"{pre}.b{l}"]),
                                        _ptr(self.a[l]), _ptr(fst[l]),
                                        _ptr(fst[l - 1]) if l else None,
                                        bnp(f"{pre}.gamma{l - 1}") if l else None,
                                        bnp(f"{pre}.beta{l - 1}") if l else None,
                                        _ptr(mk[l - 1]) if l else None, _ptr(self.bn[l - 1]) if (l and self.bn_on) else None,
                                        rs, seed, l, rate, B, K, self.widths[l],
                                        ref(sort_job) if (l == 0 and sort_in_fwd) else None, ref(sw[l]),
                                        z_fwd0[0] if l == 0 else None, z_fwd0[1] if l == 0 else 0, st),
                  "rsx_tower_fwd_layer")
        # head parameters: a variable name, or an explicit (tensor, grad_tensor) pair (e.g. a slice of out.W)
        pv = lambda x: None if x is None else (P[x] if isinstance(x, str) else x[0])
        gv = lambda x: None if x is None else (P[x].grad if isinstance(x, str) else x[1])
        wd, bd, wo, bo = head
        n_last = self.widths[-1]
        check(L.rsx_tower_head(_ptr(self.a[-1]), _ptr(fst[-1]), bnp(f"{pre}.gamma{nl - 1}"),
                               bnp(f"{pre}.beta{nl - 1}"), _ptr(mk[-1]), _ptr(self.bn[-1]), _ptr(pv(wd)), _ptr(pv(bd)),
                               _ptr(s0), _ptr(pv(c0)), _ptr(s1), _ptr(pv(wo)),
                               _ptr(pv(bo)), _ptr(labels), _ptr(self.prob), _ptr(self.dy[-1]),
                               _ptr(bst[-1]), _ptr(self.dwd_part), _ptr(self.hpart), _ptr(o_gs0), _ptr(o_gs1),
                               rs, seed, nl - 1, rate, 1.0 / (B * replicas), int(relu0), int(relu2), B, n_last, ref(sw[nl]), st),
              "rsx_tower_head")
        # (large batches: the layers' dW reductions are handed back as jobs and run as ONE launch after the last layer)
        jobs = (_lib.DwReduceJob * max(nl, 1))()
        defer = nl <= 4 and layer_done is None
        for l in reversed(range(nl)):
            K = self.k0 if l == 0 else self.widths[l - 1]
            last = l == nl - 1
            extra = None
            if cross_rider is not None and (last or l == 0):
                extra = _lib.TowerBwdExtra()
                if last:
                    cr = cross_rider
                    self.cross_job_pending = _lib.CrossReduceJob()
                    extra.x0, extra.cW, extra.cB, extra.s = cr["x0"].data_ptr(), cr["W"].data_ptr(), cr["B"].data_ptr(), cr["s"].data_ptr()
                    extra.gz, extra.wout, extra.dX = o_gs0.data_ptr(), cr["wout"].data_ptr(), o_dX.data_ptr()
                    extra.dcW, extra.dcB, extra.dwout = cr["dW"].data_ptr(), cr["dB"].data_ptr(), cr["dwout"].data_ptr()
                    extra.workspace, extra.dim, extra.L = cr["ws"].data_ptr(), int(cr["dim"]), int(cr["L"])
                    extra.reduce_out = C.pointer(self.cross_job_pending)
                else:
                    extra.accumulate_dx = 1
            check(L.rsx_tower_bwd_layer_defer(
                _ptr(X if l == 0 else self.a[l - 1]), _ptr(P[f"{pre}.W{l}"]), _ptr(self.a[l]), _ptr(self.dy[l]),
                _ptr(bst[l]), _ptr(self.bn[l]), bnp(f"{pre}.gamma{l}"),
                _ptr(g(f"{pre}.W{l}")), _ptr(g(f"{pre}.b{l}")), bng(f"{pre}.gamma{l}"), bng(f"{pre}.beta{l}"),
                _ptr(self.bn[l - 1]) if l else None, bnp(f"{pre}.gamma{l - 1}") if l else None,
                bnp(f"{pre}.beta{l - 1}") if l else None, _ptr(mk[l - 1]) if l else None,
                _ptr(self.dy[l - 1]) if l else _ptr(o_dX), _ptr(bst[l - 1]) if l else None,
                _ptr(self.hpart) if last else None, _ptr(self.dwd_part) if last else None,
                _ptr(gv(wd)) if last else None, _ptr(gv(bd)) if last else None,
                _ptr(gv(wo)) if last else None, _ptr(gv(bo)) if last else None,
                _ptr(gv(c0)) if last else None, _ptr(self.loss) if last else None,
                rs, seed, l, rate, B, K, self.widths[l],
                C.byref(sort_job) if (last and sort_job is not None and not sort_in_fwd) else None,
                ref(sw[nl + 1 + (nl - 1 - l)]), _ptr(self.dwp[l]), C.byref(jobs[l]) if defer else None,
                z_bwd0[0] if l == 0 else None, z_bwd0[1] if l == 0 else 0, None if extra is None else C.byref(extra), st),
                "rsx_tower_bwd_layer_defer")
            if layer_done is not None:
                layer_done(l)
        self.dw_jobs_pending = []
        if defer:
            todo = [jobs[l] for l in range(nl) if jobs[l].sb > 0]
            if todo and defer_dw_reduce:
                # (the caller hands them to the scatter's stage-A launch: make_scatter_riders / segsum_adam(riders=...))
                self.dw_jobs_pending = todo
            elif todo:
                arr = (_lib.DwReduceJob * len(todo))(*todo)
                check(L.rsx_tower_reduce_dw_jobs(arr, len(todo), st), "rsx_tower_reduce_dw_jobs")
        return self.loss, self.prob[:B], o_dX[:B], o_gs0[:B], o_gs1[:B]


    def _mlp_fused_ok(self):
        """The one-launch form of a batch-norm-free tower applies (envelope of csrc/mlp_fused.hip); allocates its workspace."""
        ok = getattr(self, "_mlp_ok", None)
        if ok is None:
            w = (C.c_int32 * len(self.widths))(*self.widths)
            ok = _lib.form("mlp_fuse") == "1" and len(self.widths) <= 3 and \
                bool(lib().rsx_mlp_nobn_supported(self.k0, w, len(self.widths)))
            if ok:
                n = int(lib().rsx_mlp_nobn_workspace_floats(self.cap, self.k0, w, len(self.widths)))
                self._mlp_ws = torch.empty(n, device=self.prob.device)
            self._mlp_ok = ok
        return ok

    def infer(self, X, rng_step, labels=None, s0=None, c0=None, s1=None,
              head=("dnn.Wout", "dnn.bout", "out.W", "out.b"), relu0=True, relu2=True):
        """EVAL / PREDICT forward through the same kernels as the TRAIN step (L forward launches + the head): dropout off and
        batch-norm in inference form.  The reference never updates the moving statistics (EVAL uses mean 0 /
        variance 1), so BN(x) = gamma * x / sqrt(1 + eps) + beta -- exactly what the kernels compute when the statistics they
        are handed say sum(x) = 0, sum(x^2) = B.  -> (prob [B], loss [1] or None when labels is None)."""
        L, P, pre = lib(), self.P, self.pre
        B = X.shape[0]
        assert B <= self.cap and X.is_contiguous() and X.shape[1] == self.k0
        st = _stream()
        nl = len(self.widths)
        bnp = (lambda name: _ptr(P[name])) if self.bn_on else (lambda name: None)
        es = self._eval_stats.get(B)
        if es is None:             # (first call per batch size: eager, before any capture of this signature)
            es = []
            for n in self.widths:
                if B > 512:         # the fixed-point rows (include/rsx.h): sum = 0, sum of squares = B = (hi * 2^32 + lo) * 2^-52
                    t = torch.zeros(8, 4, (n + 15) // 16 * 16, dtype=torch.int64, device=X.device)
                    t[0, 2] = B << 20
                    t = t.view(torch.float64)
                else:
                    t = torch.zeros(self.fstat[0].shape[0], 2, n, dtype=torch.float64, device=X.device)
                    t[0, 1] = float(B)
                es.append(t)
            self._eval_stats[B] = es
        rs = _ptr(rng_step)
        for l in range(nl):
            K = self.k0 if l == 0 else self.widths[l - 1]
            check(L.rsx_tower_fwd_layer(_ptr(X if l == 0 else self.a[l - 1]), _ptr(P[f"{pre}.W{l}"]), _ptr(P[f"{pre}.b{l}"]),
                                        _ptr(self.a[l]), _ptr((self.fstat_fix_eval if B > 512 else self.fstat)[l]),
                                        _ptr(es[l - 1]) if l else None,
                                        bnp(f"{pre}.gamma{l - 1}") if l else None, bnp(f"{pre}.beta{l - 1}") if l else None,
                                        None, _ptr(self.bn[l - 1]) if (l and self.bn_on) else None, rs, 0, l, 0.0, B, K,
                                        self.widths[l], None, None, None, 0, st), "rsx_tower_fwd_layer")
        pv = lambda x: None if x is None else (P[x] if isinstance(x, str) else x[0])
        wd, bd, wo, bo = head
        lab = labels if labels is not None else self._zero_labels[:B]
        check(L.rsx_tower_head(_ptr(self.a[-1]), _ptr(es[-1]), bnp(f"{pre}.gamma{nl - 1}"), bnp(f"{pre}.beta{nl - 1}"), None,
                               _ptr(self.bn[-1]), _ptr(pv(wd)), _ptr(pv(bd)), _ptr(s0), _ptr(pv(c0)), _ptr(s1), _ptr(pv(wo)),
                               _ptr(pv(bo)), _ptr(lab), _ptr(self.prob), _ptr(self.dy[-1]),
                               _ptr((self.bstat_fix_eval if B > 512 else self.bstat)[-1]),
                               _ptr(self.dwd_part), _ptr(self.hpart), _ptr(self.gs0), _ptr(self.gs1), rs, 0, nl - 1, 0.0,
                               1.0 / B, int(relu0), int(relu2), B, self.widths[-1], None, st), "rsx_tower_head")
        loss = None
        if labels is not None:     # the head's per-row-tile partial sums of the CE terms (column 0), in fp64
            loss = (self.hpart[:(B + 15) // 16, 0].sum() / B).to(torch.float32).reshape(1)
        return self.prob[:B].clone(), loss          # (a copy: self.prob is the TRAIN step's output buffer as well)


class CrossLayers:
    """DCN cross layers (csrc/cross.hip) with their backward workspace.  dcn/dcn.py:132-142."""

    def __init__(self, dim, L, capacity, device="cuda"):
        dev = _require_cuda(device)
        self.dim, self.L, self.cap = int(dim), int(L), int(capacity)
        self.s = torch.empty(self.cap, self.L, device=dev)
        self.cz = torch.empty(self.cap, device=dev)
        n = lib().rsx_cross_bwd_workspace_floats(self.cap, self.dim, self.L)
        self.ws = torch.empty(max(n, lib().rsx_cross_bwd_workspace_floats(1024, self.dim, self.L)), device=dev)

    def forward(self, x0, W, Bc, wout=None, want_xL=False):
        B = x0.shape[0]
        xL = torch.empty_like(x0) if want_xL else None
        check(lib().rsx_cross_fwd(_ptr(x0), _ptr(W), _ptr(Bc), _ptr(wout), _ptr(self.s), _ptr(xL),
                                  _ptr(self.cz) if wout is not None else None, B, self.dim, self.L, _stream()),
              "rsx_cross_fwd")
        return self.s[:B], xL, (self.cz[:B] if wout is not None else None)

    def gather_forward(self, arena, ids, W, Bc, wout):
        """The arena's input_layer lookup AND the cross layers' forward in ONE launch (rsx_gather_cross_fwd, round 4): -> x0 [B, dim],
        s [B, L], cz [B].  Applies when arena.D == 16 and arena.F * 16 == dim (fused_gather_ok)."""
        B = ids.shape[0]
        E, _, _, _ = arena.gather_outputs(B, False, False, None)
        check(lib().rsx_gather_cross_fwd(_ptr(arena.tables), _ptr(arena.row_off), _ptr(ids), _ptr(E), _ptr(W), _ptr(Bc), _ptr(wout),
                                         _ptr(self.s), _ptr(self.cz), B, arena.F, arena.D, self.L, _stream()),
              "rsx_gather_cross_fwd")
        return E, self.s[:B], self.cz[:B]

    def fused_gather_ok(self, arena):
        return arena.D == 16 and arena.F <= 64 and arena.F * arena.D == self.dim and self.L <= 8 and \
            _lib.form("gather_cross") == "1"

    def cross_ride_ok(self, tower, B):
        """Can the backward ride in the tower's last backward launch (rsx_tower_bwd_cross_ride_supported; RSX_FORMS cross_ride=0:
        the launch of its own)?"""
        w, nl = tower.widths, len(tower.widths)
        return nl >= 2 and tower.bn_on and _lib.form("cross_ride") == "1" and bool(lib().rsx_tower_bwd_cross_ride_supported(
            int(B), w[nl - 2], w[nl - 1], tower.k0, w[0], self.dim, self.L))

    def rider_args(self, x0, W, Bc, dW, dB, wout, dwout):
        """What FusedTower.train_step(cross_rider=...) needs to run this backward inside its last layer's backward launch."""
        return dict(x0=x0, W=W, B=Bc, s=self.s, wout=wout, dW=dW, dB=dB, dwout=dwout, ws=self.ws, dim=self.dim, L=self.L)

    def backward(self, x0, W, Bc, dW, dB, dX, accumulate, dxL=None, gz=None, wout=None, dwout=None, defer_reduce=False):
        """defer_reduce: the second launch (the gradient partials' sum) comes back as a _lib.CrossReduceJob for the scatter's
        stage-A launch to carry (make_scatter_riders) instead of being launched here."""
        B = x0.shape[0]
        job = _lib.CrossReduceJob() if defer_reduce else None
        check(lib().rsx_cross_bwd_defer(_ptr(x0), _ptr(W), _ptr(Bc), _ptr(self.s), _ptr(dxL), _ptr(gz), _ptr(wout), _ptr(dX),
                                        int(accumulate), _ptr(dW), _ptr(dB), _ptr(dwout), _ptr(self.ws), B, self.dim, self.L,
                                        None if job is None else C.byref(job), _stream()), "rsx_cross_bwd_defer")
        return job


class CrossFn(torch.autograd.Function):
    """Autograd wrapper (used by the torch-tower path and by the op-level parity tests)."""

    @staticmethod
    def forward(ctx, x0, W, Bc, op):
        _, xL, _ = op.forward(x0.contiguous(), W, Bc, None, want_xL=True)
        ctx.op = op
        ctx.save_for_backward(x0, W, Bc)
        return xL

    @staticmethod
    def backward(ctx, g):
        x0, W, Bc = ctx.saved_tensors
        dW, dB, dX = torch.empty_like(W), torch.empty_like(Bc), torch.empty_like(x0)
        ctx.op.backward(x0.contiguous(), W, Bc, dW, dB, dX, False, dxL=g.contiguous())
        return dX, dW, dB, None


class SparseTable:
    """One embedding variable looked up with tf.nn.embedding_lookup / tf.gather several times per step
    (din/din.py:88-105): table[R,K] + Adam slots + the sorted-segment workspace for up to `capacity` entries.
    Lookups register their (ids, grad) pairs during backward; `finalize()` concatenates them in registration
    order, sorts by row (stable, so duplicate rows are summed in entry order like TF's
    _apply_sparse_duplicate_indices), builds the segments and the summed gradient G, which the non-lazy TF-1
    Adam kind pulls through `slot`.  Sort, segments and sums are the same librsx.so kernels as the Criteo fields (F = 1)."""

    def __init__(self, rows, K, capacity, device="cuda", table=None, null_row=-1):
        dev = _require_cuda(device)
        self.R, self.K, self.cap = int(rows), int(K), int(capacity)
        # null_row: the key whose entries carry exactly-zero gradients and need not be summed, or -1.  null_row == rows is a
        # DUMMY key one past the table: lookups declared with `pad_id` (DIN's histories, padding id 0: din/din.py:107) map their
        # padding entries to it, so that the real row `pad_id` stays an ordinary row for every other lookup of the table
        self.null_row = int(null_row)
        self.key_rows = self.R + 1 if self.null_row == self.R else self.R
        self.table = torch.zeros(self.R, self.K, device=dev) if table is None else \
            torch.as_tensor(table, dtype=torch.float32).to(dev).contiguous()
        self.m = torch.zeros_like(self.table)
        self.v = torch.zeros_like(self.table)
        i32 = dict(dtype=torch.int32, device=dev)
        self.row_off = torch.zeros(2, **i32)
        self.row_off[1] = self.key_rows
        self.uniq_row = torch.zeros(self.cap, **i32)
        self.seg_off = torch.zeros(self.cap + 1, **i32)
        self.nuniq = torch.zeros(1, **i32)
        self.slot = torch.full((self.R + 8,), -1, **i32)          # (+ the dummy key, + int4 tail reads)
        self.G = torch.zeros(self.cap, self.K, device=dev)
        self.perm = torch.zeros(self.cap, **i32)
        self.sort_ws = torch.zeros(int(lib().rsx_field_sort_large_workspace_ints(self.cap, 1, self.cap)), **i32) \
            if self.cap > EmbeddingArena.LDS_SORT_MAX_B else None
        # two-stage segment-sum workspace (Zipf-head rows make a few segments thousands of entries long)
        nch = (self.cap + 15) // 16
        self.segid = torch.zeros(self.cap + 2 + nch, **i32)
        self.P = torch.zeros(nch * 2, self.K, device=dev)
        self.partials = _lib.SegPartials(_ptr(self.segid), _ptr(self.P), None, _ptr(self.G), None)
        self.hook = torch.zeros((), device=dev, requires_grad=True)
        self.pending = []
        self._seq = 0

    def gather(self, ids):
        """ids: int32 tensor of any shape -> rows [*ids.shape, K] (no autograd)."""
        flat = ids.reshape(-1, 1).contiguous()
        out = torch.empty(flat.shape[0], self.K, device=self.table.device)
        check(lib().rsx_gather_fm_fwd(_ptr(self.table), None, _ptr(self.row_off), _ptr(flat), _ptr(out), None, None, None,
                                      0, flat.shape[0], 1, self.K, _stream()), "rsx_gather_fm_fwd")
        return out.view(*ids.shape, self.K)

    def lookup(self, ids, pad_id=None):
        """pad_id: this lookup's padding id (its entries are masked downstream and carry exactly-zero gradients): keyed to
        the table's dummy row in the sparse gradient when the table was built with null_row == rows."""
        return _LookupFn.apply(self.hook, self, ids, pad_id)

    def finalize(self, dp=None):
        """Dedup + ordered segment-sum of everything registered since the last call (forward lookup order, which is
        the order TF concatenates the IndexedSlices of one variable in).  dp: all-gather the entries first."""
        if not self.pending:
            return
        self.pending.sort(key=lambda t: t[0])
        ids = torch.cat([i.reshape(-1) for _, i, _ in self.pending])
        vals = torch.cat([g.reshape(-1, self.K) for _, _, g in self.pending]).contiguous()
        self.pending, self._seq = [], 0
        if dp is not None:
            ids, vals = dp.all_gather_rows(ids), dp.all_gather_rows(vals)
        N = ids.shape[0]
        assert N <= self.cap, "SparseTable capacity %d < %d entries" % (self.cap, N)
        # dedup sort of the row keys by the same kernels as the Criteo fields (F = 1): stable by entry order, so duplicate
        # rows are summed in entry order like TF's _apply_sparse_duplicate_indices
        ids2 = ids.to(torch.int32).reshape(N, 1).contiguous()
        two = N > EmbeddingArena.TWO_STAGE_MIN_B
        perm = self.perm
        if N > EmbeddingArena.LDS_SORT_MAX_B:
            check(lib().rsx_field_sort_large(_ptr(ids2), _ptr(self.row_off), _ptr(perm), _ptr(self.seg_off), _ptr(self.uniq_row),
                                             _ptr(self.nuniq), _ptr(self.slot), _ptr(self.segid), _ptr(self.sort_ws), self.key_rows,
                                             N, 1, self.cap, _stream()), "rsx_field_sort_large")
        else:
            check(lib().rsx_field_sort(_ptr(ids2), _ptr(self.row_off), _ptr(perm), _ptr(self.seg_off), _ptr(self.uniq_row),
                                       _ptr(self.nuniq), _ptr(self.slot), _ptr(self.segid) if two else None, self.key_rows, N, 1,
                                       self.cap, _stream()), "rsx_field_sort")
        part = None
        if two:
            part = C.byref(self.partials)
            check(lib().rsx_segsum_partials(None, None, _ptr(vals), None, None, _ptr(perm), _ptr(self.seg_off),
                                            _ptr(self.uniq_row), part, 0, N, 1, self.K, self.cap, self.null_row, None,
                                            _stream()),
                  "rsx_segsum_partials")
        check(lib().rsx_segsum_rows(_ptr(vals), _ptr(perm), _ptr(self.seg_off), _ptr(self.uniq_row), _ptr(self.nuniq),
                                    _ptr(self.G), N, self.K, self.cap, self.null_row, part, _stream()), "rsx_segsum_rows")
        self._keep = (ids2, vals)

    def adam_segments(self, lazy=False):
        if lazy:
            raise _lib.RsxError("lazy_rows is not implemented for SparseTable (DIN); use adam_mode='tf1_dense'")
        return [dict(kind=_lib.RSX_ADAM_TABLE_TF1, d=self.K, n=self.R, var=self.table, m=self.m, v=self.v, g=self.G,
                     slot=self.slot)]


class _LookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hook, tbl, ids, pad_id=None):
        ctx.tbl, ctx.ids, ctx.seq, ctx.pad_id = tbl, ids, tbl._seq, pad_id
        tbl._seq += 1
        return tbl.gather(ids)

    @staticmethod
    def backward(ctx, g):
        ids = ctx.ids
        if ctx.pad_id is not None and ctx.tbl.null_row == ctx.tbl.R:
            ids = torch.where(ids == ctx.pad_id, torch.full_like(ids, ctx.tbl.null_row), ids)
        ctx.tbl.pending.append((ctx.seq, ids, g.contiguous()))
        return None, None, None, None


class DinPoolFn(torch.autograd.Function):
    """sum_p H[b,p,:] * w[b,p] * (ids > 0)   (din/din.py:118-124)."""

    @staticmethod
    def forward(ctx, H, w, ids):
        B, P, K = H.shape
        H, w = H.contiguous(), w.contiguous()
        out = torch.empty(B, K, device=H.device)
        check(lib().rsx_din_pool_fwd(_ptr(H), _ptr(w), _ptr(ids), _ptr(out), B, P, K, _stream()), "rsx_din_pool_fwd")
        ctx.save_for_backward(H, w, ids)
        return out

    @staticmethod
    def backward(ctx, g):
        H, w, ids = ctx.saved_tensors
        B, P, K = H.shape
        dH, dw = torch.empty_like(H), torch.empty_like(w)
        check(lib().rsx_din_pool_bwd(_ptr(H), _ptr(w), _ptr(ids), _ptr(g.contiguous()), _ptr(dH), _ptr(dw), 0, B, P, K,
                                     _stream()), "rsx_din_pool_bwd")
        return dH, dw, None


class DinAttnFn(torch.autograd.Function):
    """The attention MLP of `_attention` (din/din.py:111-121) as one fused launch per direction (csrc/din_attn.hip):
    w[b,p] = W2 . drop(relu(W1^T . drop(relu(W0^T . [h, q, h*q, h-q] + b0)) + b1)) + b2."""

    @staticmethod
    def supported(K, N1, N2):
        return K in (16, 32) and N1 <= 80 and N2 <= 48

    @staticmethod
    def forward(ctx, H, q, W0, b0, W1, b1, W2, b2, rate, masks, rng_step, seed, layer0, grad_out=None):
        """grad_out: optional flat fp32 buffer [dW0|db0|dW1|db1|dW2|db2] (e.g. the dense arena's grad slice of these six
        variables, DenseArena.packed_grad): the backward kernel writes the weight gradients THERE and autograd gets None for
        them -- no per-variable accumulate launches."""
        B, P, K = H.shape
        N1, N2 = W0.shape[1], W1.shape[1]
        H, q = H.contiguous(), q.contiguous()
        dev = H.device
        a1, a2 = torch.empty(B * P, N1, device=dev), torch.empty(B * P, N2, device=dev)
        w = torch.empty(B, P, device=dev)
        m1, m2 = (None, None) if masks is None else (masks[0].contiguous(), masks[1].contiguous())
        check(lib().rsx_din_attn_fwd(_ptr(H), _ptr(q), _ptr(W0), _ptr(b0), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(a1),
                                     _ptr(a2), _ptr(w), _ptr(m1), _ptr(m2), _ptr(rng_step), seed, layer0, rate, None, None,
                                     B, P, K, N1, N2, _stream()), "rsx_din_attn_fwd")
        ctx.save_for_backward(H, q, W0, W1, W2, a1, a2)
        ctx.cfg = (rate, m1, m2, rng_step, seed, layer0)
        ctx.grad_out = grad_out
        return w

    @staticmethod
    def backward(ctx, g):
        H, q, W0, W1, W2, a1, a2 = ctx.saved_tensors
        rate, m1, m2, rng_step, seed, layer0 = ctx.cfg
        B, P, K = H.shape
        N1, N2 = W0.shape[1], W1.shape[1]
        dev = H.device
        dH, dq = torch.empty_like(H), torch.empty_like(q)
        n0, n1 = 4 * K * N1, N1 * N2
        ng = n0 + N1 + n1 + 2 * N2 + 1
        direct = ctx.grad_out is not None
        if direct:
            assert ctx.grad_out.is_contiguous() and ctx.grad_out.numel() >= ng
        grads = ctx.grad_out if direct else torch.empty(ng, device=dev)
        ws = torch.empty(int(lib().rsx_din_attn_bwd_workspace_floats(B, P, K, N1, N2)), device=dev)
        check(lib().rsx_din_attn_bwd(_ptr(H), _ptr(q), _ptr(W0), _ptr(W1), _ptr(W2), _ptr(a1), _ptr(a2),
                                     _ptr(g.contiguous()), _ptr(dH), _ptr(dq), _ptr(grads), _ptr(ws), _ptr(m1), _ptr(m2),
                                     _ptr(rng_step), seed, layer0, rate, 0, None, None, None, B, P, K, N1, N2, _stream()),
              "rsx_din_attn_bwd")
        o = 0
        out = []
        for n, shape in ((n0, W0.shape), (N1, (N1,)), (n1, W1.shape), (N2, (N2,)), (N2, W2.shape), (1, (1,))):
            out.append(grads[o:o + n].view(shape))
            o += n
        dW0, db0, dW1, db1, dW2, db2 = out
        if direct:
            return (dH, dq) + (None,) * 12
        return dH, dq, dW0, db0, dW1, db1, dW2, db2, None, None, None, None, None, None


class DinAttnPoolFn(torch.autograd.Function):
    """DinAttnFn followed by DinPoolFn as ONE autograd node (din/din.py:111-124): the backward runs the pooling backward
    first (dH share + the logits' gradient dw) and lets the attention backward ACCUMULATE into the same dH buffer, so
    autograd never adds two [B,P,K] gradients; weight gradients go straight to `grad_out` (see DinAttnFn)."""

    @staticmethod
    def forward(ctx, H, q, hist, W0, b0, W1, b1, W2, b2, rate, masks, rng_step, seed, layer0, grad_out):
        B, P, K = H.shape
        N1, N2 = W0.shape[1], W1.shape[1]
        H, q = H.contiguous(), q.contiguous()
        dev = H.device
        a1, a2 = torch.empty(B * P, N1, device=dev), torch.empty(B * P, N2, device=dev)
        w, out = torch.empty(B, P, device=dev), torch.empty(B, K, device=dev)
        m1, m2 = (None, None) if masks is None else (masks[0].contiguous(), masks[1].contiguous())
        # only the history positions that are not padding go through the MLP (they are masked out of the sum and get no
        # gradient: din/din.py:118-124); w is zeroed at the padded ones
        rows = torch.empty(B * P + 2 + (B * P + 1023) // 1024, dtype=torch.int32, device=dev)     # row list | count | scratch
        cnt = rows[B * P:]
        check(lib().rsx_din_valid_rows(_ptr(hist), B, P, _ptr(rows), _ptr(cnt), _ptr(w), _stream()), "rsx_din_valid_rows")
        check(lib().rsx_din_attn_fwd(_ptr(H), _ptr(q), _ptr(W0), _ptr(b0), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(a1),
                                     _ptr(a2), _ptr(w), _ptr(m1), _ptr(m2), _ptr(rng_step), seed, layer0, rate, _ptr(rows),
                                     _ptr(cnt), B, P, K, N1, N2, _stream()), "rsx_din_attn_fwd")
        check(lib().rsx_din_pool_fwd(_ptr(H), _ptr(w), _ptr(hist), _ptr(out), B, P, K, _stream()), "rsx_din_pool_fwd")
        ctx.rows = rows
        ctx.save_for_backward(H, q, hist, W0, W1, W2, a1, a2, w)
        ctx.cfg = (rate, m1, m2, rng_step, seed, layer0)
        ctx.grad_out = grad_out
        return out

    @staticmethod
    def backward(ctx, g):
        H, q, hist, W0, W1, W2, a1, a2, w = ctx.saved_tensors
        rate, m1, m2, rng_step, seed, layer0 = ctx.cfg
        B, P, K = H.shape
        N1, N2 = W0.shape[1], W1.shape[1]
        dev = H.device
        dH, dw, dq = torch.empty_like(H), torch.empty_like(w), torch.empty_like(q)
        check(lib().rsx_din_pool_bwd(_ptr(H), _ptr(w), _ptr(hist), _ptr(g.contiguous()), _ptr(dH), _ptr(dw), 0, B, P, K,
                                     _stream()), "rsx_din_pool_bwd")
        ng = 4 * K * N1 + N1 + N1 * N2 + 2 * N2 + 1
        assert ctx.grad_out.is_contiguous() and ctx.grad_out.numel() >= ng
        ws = torch.empty(int(lib().rsx_din_attn_bwd_workspace_floats(B, P, K, N1, N2)), device=dev)
        check(lib().rsx_din_attn_bwd(_ptr(H), _ptr(q), _ptr(W0), _ptr(W1), _ptr(W2), _ptr(a1), _ptr(a2), _ptr(dw), _ptr(dH),
                                     _ptr(dq), _ptr(ctx.grad_out), _ptr(ws), _ptr(m1), _ptr(m2), _ptr(rng_step), seed, layer0,
                                     rate, 1, _ptr(ctx.rows), _ptr(ctx.rows[B * P:]), _ptr(hist), B, P, K, N1, N2, _stream()),
              "rsx_din_attn_bwd")
        return (dH, dq) + (None,) * 13


class CinLayerFn(torch.autograd.Function):
    """One CIN layer (csrc/cin.hip): out[b,n,d] = relu(sum_{f,h} X0[b,f,d] Xk[b,h,d] W[f*H+h,n] + c[n]).
    xdeepfm/xdeepfm.py:145-172."""

    @staticmethod
    def forward(ctx, X0, Xk, W, c, sweep=None):
        """sweep: optional rsx_adam_slice (AdamTF1.cold_slices) carried by this layer's weight-gradient launch."""
        B, F, D = X0.shape
        H, N = Xk.shape[1], W.shape[1]
        X0, Xk = X0.contiguous(), Xk.contiguous()
        out = torch.empty(B, N, D, device=X0.device)
        check(lib().rsx_cin_layer_fwd(_ptr(X0), _ptr(Xk), _ptr(W), _ptr(c), _ptr(out), B, F, H, N, D, None, _stream()),
              "rsx_cin_layer_fwd")
        ctx.save_for_backward(X0, Xk, W, out)
        ctx.sweep = sweep
        return out

    @staticmethod
    def backward(ctx, g):
        X0, Xk, W, out = ctx.saved_tensors
        B, F, D = X0.shape
        H, N = Xk.shape[1], W.shape[1]
        dX0, dXk = torch.empty_like(X0), torch.empty_like(Xk)
        dW, dc = torch.empty_like(W), torch.empty(N, device=W.device)
        ws = torch.empty(int(lib().rsx_cin_bwd_workspace_floats(B, F, H, N)), device=W.device)
        check(lib().rsx_cin_layer_bwd(_ptr(X0), _ptr(Xk), _ptr(W), _ptr(out), _ptr(g.contiguous()), None, None, _ptr(dXk), 0, _ptr(dX0),
                                      0, _ptr(dW), _ptr(dc), _ptr(ws), B, F, H, N, D,
                                      None if ctx.sweep is None else C.byref(ctx.sweep), _stream()), "rsx_cin_layer_bwd")
        return dX0, dXk, dW, dc, None


class CinNet:
    """'cin_net' of the training step (xdeepfm/xdeepfm.py:135-182) without autograd or glue kernels: L rsx_cin_layer_fwd +
    rsx_cin_out_fwd forward; rsx_cin_out_bwd + L rsx_cin_layer_bwd backward.  The concat / reduce_sum / dense head and the
    broadcast of its gradient back into every layer map happen inside those kernels; the gradients of X^0 from all layers
    (and from its second role as X^k of layer 0) accumulate in one buffer; weight gradients land in the dense arena."""

    def __init__(self, F, D, sizes, capacity, device="cuda", bf16=False, split=0):
        dev = _require_cuda(device)
        self.F, self.D, self.sizes, self.L = F, D, [int(n) for n in sizes], len(sizes)
        # split = 3: the contraction on the bf16 matrix cores with every operand kept as three bf16 planes
        # (csrc/cin_split.hip: every product exact to 2^-23 -- fp32-grade, the parity path on those cores);
        # split = 4: forward / data gradients with TWO scaled fp16 planes per operand (three MFMAs per k-step instead of six,
        # products to 2^-22), weight gradients on three bf16 planes
        self.split = int(split or 0)
        if self.split and not (3 <= self.split <= 4 and F <= 40 and D == 16 and max(sizes) <= 128 and self.L <= 4):
            raise _lib.RsxError("CinNet: split operands need split in (3, 4), F <= 40, D = 16, layers <= 128 wide, <= 4 layers")
        if self.split:
            bf16 = False
            hs16 = [F] + self.sizes[:-1]
            ns = self.split
            self.w16 = [torch.empty(int(lib().rsx_cin_split_weight_elems(F, h, n, ns)), dtype=torch.int16, device=dev)
                        for h, n in zip(hs16, self.sizes)]
            self.ws16 = [torch.empty(int(lib().rsx_cin_split_bwd_workspace_bytes(capacity, n, ns)), dtype=torch.uint8, device=dev)
                         for n in self.sizes]
            self._w16_h = (C.c_void_p * self.L)(*[w.data_ptr() for w in self.w16])
            self._H_h = (C.c_int32 * self.L)(*hs16)
            # layer 0 in mode 4 (dXk IS dX0; a few tiles of h only): its data-gradient launch splits the FIELDS over two workgroups
            # per tile when both fit the CUs in one round; the second half's dXk is one more tile partial (RSX_CIN_DX_FSPLIT=0: off)
            ht0 = (hs16[0] + 15) // 16
            self.fsplit0 = (ns == 4 and _lib.form("cin_dx_fsplit") != "0" and
                            2 * ht0 * ((capacity + 7) // 8) <= 256)
            extra = [capacity * F * D if (k == 0 and self.fsplit0) else 0 for k in range(self.L)]
            self.dx0_parts = [torch.empty(int(lib().rsx_cin_bf16_dx0_parts_floats(capacity, F, h)) + x, device=dev)
                              for h, x in zip(hs16, extra)]
            self._tiles_h = (C.c_int32 * self.L)(*[(h + 15) // 16 + (1 if (k == 0 and self.fsplit0) else 0)
                                                   for k, h in enumerate(hs16)][::-1])
        # bf16=True: the contraction runs on the bf16 MFMA path (csrc/cin_bf16.hip: Xk / W / dpre rounded to bf16, fp32
        # accumulation) -- NOT the parity path; fp32 (False) is the default everywhere
        self.bf16 = bool(bf16)
        if self.bf16:
            hs16 = [F] + self.sizes[:-1]
            self.w16 = [torch.empty(int(lib().rsx_cin_bf16_weight_elems(F, h, n)), dtype=torch.int16, device=dev)
                        for h, n in zip(hs16, self.sizes)]
            # one backward workspace PER LAYER: the dX launches fill them layer by layer, ONE launch then computes every
            # layer's weight gradient from them
            self.ws16 = [torch.empty(int(lib().rsx_cin_bf16_bwd_workspace_bytes(capacity, n)), dtype=torch.uint8, device=dev)
                         for n in self.sizes]
            self._w16_h = (C.c_void_p * self.L)(*[w.data_ptr() for w in self.w16])
            self._H_h = (C.c_int32 * self.L)(*hs16)
            # eight examples per workgroup (csrc/cin_bf16_wide.hip): the data-gradient launches leave dX0 as one partial per
            # 16-wide tile of h, ONE reduce launch adds the tiles of all layers
            self.wide = F <= 40 and _lib.form("cin_wide") != "0"
            if self.wide:
                self.dx0_parts = [torch.empty(int(lib().rsx_cin_bf16_dx0_parts_floats(capacity, F, h)), device=dev) for h in hs16]
                self._tiles_h = (C.c_int32 * self.L)(*[(h + 15) // 16 for h in hs16][::-1])
        self.outs = [torch.empty(capacity, n, D, device=dev) for n in self.sizes]
        self.dmap = [torch.empty(capacity, n, D, device=dev) for n in self.sizes[:-1]]   # gradient wrt map k (from layer k+1)
        hs = [F] + self.sizes[:-1]
        self.dpre = torch.empty(max(int(lib().rsx_cin_bwd_workspace_floats(capacity, F, h, n)) for h, n in zip(hs, self.sizes)), device=dev)
        self.dX0 = torch.empty(capacity, F, D, device=dev)
        self.y, self.gs = torch.empty(capacity, device=dev), torch.empty(capacity, device=dev)
        self._sizes_h = (C.c_int32 * self.L)(*self.sizes)
        self._outs_h = (C.c_void_p * self.L)(*[o.data_ptr() for o in self.outs])
        self.offs = [sum(self.sizes[:k]) for k in range(self.L)]

    def gather_ride_ok(self):
        """xdeepfm.py's lookup (rsx_gather_two_fwd) can ride in this net's filter-preparation launch (RSX_CIN_GATHER_RIDE=0: two
        launches, A/B runs)."""
        return bool(self.split) and self.D == 16 and _lib.form("cin_gather_ride") != "0"

    def forward(self, X0, P, sweeps=None, gather_job=None):
        """X0 [B,F,D] contiguous -> cin_y [B] (view of an internal buffer).  sweeps[k]: slice of the untouched-row
        optimizer sweep carried by layer k's forward launch."""
        B = X0.shape[0]
        Xk, H = X0, self.F
        if self.split:    # the split operand images of all layers' filters: one launch (+ the caller's lookup as its rider)
            W_h = (C.c_void_p * self.L)(*[P[f"cin.W{k}"].data_ptr() for k in range(self.L)])
            if gather_job is not None:
                check(lib().rsx_cin_split_prep_gather(W_h, self._w16_h, self._H_h, self._sizes_h, self.L, self.F, self.split,
                                                      C.byref(gather_job), _stream()), "rsx_cin_split_prep_gather")
            else:
                check(lib().rsx_cin_split_prep(W_h, self._w16_h, self._H_h, self._sizes_h, self.L, self.F, self.split, _stream()),
                      "rsx_cin_split_prep")
        else:
            assert gather_job is None, "CinNet.forward: a lookup can only ride in the split-operand prep launch (gather_ride_ok)"
        if self.bf16:     # the bf16 operand images of all layers' filters: one launch
            W_h = (C.c_void_p * self.L)(*[P[f"cin.W{k}"].data_ptr() for k in range(self.L)])
            check(lib().rsx_cin_prep_bf16_multi(W_h, self._w16_h, self._H_h, self._sizes_h, self.L, self.F, _stream()),
                  "rsx_cin_prep_bf16_multi")
        for k, n in enumerate(self.sizes):
            sw = None if sweeps is None or sweeps[k] is None else C.byref(sweeps[k])
            if self.split:
                assert sw is None, "the split-operand CIN launches carry no sweep slices"
                check(lib().rsx_cin_split_fwd(_ptr(X0), _ptr(Xk), _ptr(self.w16[k]), _ptr(P[f"cin.c{k}"]), _ptr(self.outs[k]),
                                              B, self.F, H, n, self.D, self.split, _stream()), "rsx_cin_split_fwd")
                Xk, H = self.outs[k], n
                continue
            if self.bf16:
                check(lib().rsx_cin_layer_fwd_bf16(_ptr(X0), _ptr(Xk), _ptr(self.w16[k]), _ptr(P[f"cin.c{k}"]), _ptr(self.outs[k]),
                                                   B, self.F, H, n, self.D, sw, _stream()), "rsx_cin_layer_fwd_bf16")
                Xk, H = self.outs[k], n
                continue
            check(lib().rsx_cin_layer_fwd(_ptr(X0), _ptr(Xk), _ptr(P[f"cin.W{k}"]), _ptr(P[f"cin.c{k}"]), _ptr(self.outs[k]), B,
                                          self.F, H, n, self.D, sw, _stream()), "rsx_cin_layer_fwd")
            Xk, H = self.outs[k], n
        check(lib().rsx_cin_out_fwd(self._outs_h, self._sizes_h, self.L, _ptr(P["cin.Wout"]), _ptr(P["cin.bout"]), _ptr(self.y),
                                    B, self.D, _stream()), "rsx_cin_out_fwd")
        return self.y[:B]

    def backward(self, X0, P, gy, sweeps=None, dX0_out=None, lin=None):
        """gy [B]: gradient wrt cin_y.  Writes the cin.* gradients into P[...].grad and returns dX0 [B,F,D] (internal
        buffer, or dX0_out: a caller-owned contiguous [B,F,D] buffer).  sweeps[k]: slice of the untouched-row optimizer sweep
        carried by layer k's backward launch; on the bf16 path (L data-gradient launches, then ONE launch with all layers'
        weight gradients) sweeps has L + 1 entries: [dx of layer 0 .. L-1, the weight-gradient launch]."""
        B, L = X0.shape[0], self.L
        dX0 = self.dX0 if dX0_out is None else dX0_out
        # lin = (logx [B, n], g_lin [B], dwnum [n]): the numeric linear_net gradient rides in the head's backward launch
        lx, gl, dwn = lin if lin is not None else (None, None, None)
        check(lib().rsx_cin_out_bwd_lin(self._outs_h, self._sizes_h, L, _ptr(self.y), _ptr(gy), _ptr(self.gs),
                                        _ptr(P["cin.Wout"].grad), _ptr(P["cin.bout"].grad), _ptr(lx), _ptr(gl), _ptr(dwn),
                                        0 if lx is None else lx.shape[1], B, self.D, _stream()), "rsx_cin_out_bwd_lin")
        wout = P["cin.Wout"].data_ptr()
        wide = self.bf16 and self.wide and (sweeps is None or all(x is None for x in sweeps[:L]))
        for k in range(L - 1, -1, -1):
            Xk, H = (X0, self.F) if k == 0 else (self.outs[k - 1], self.sizes[k - 1])
            dout = None if k == L - 1 else _ptr(self.dmap[k])
            if k == 0:   # X0 in both roles: one buffer, accumulating
                dxk, acc_dxk, acc_dx0 = dX0, 1 if L > 1 else 0, 1
            else:
                dxk, acc_dxk, acc_dx0 = self.dmap[k - 1], 0, 0 if k == L - 1 else 1
            if self.split:
                assert sweeps is None or all(x is None for x in sweeps), "the split-operand CIN launches carry no sweep slices"
                check(lib().rsx_cin_split_bwd_dx(_ptr(X0), _ptr(Xk), _ptr(self.w16[k]), _ptr(self.outs[k]), dout, _ptr(self.gs),
                                                 C.c_void_p(wout + 4 * self.offs[k]), _ptr(dxk), 2 if (k == 0 and self.fsplit0) else 0,
                                                 _ptr(self.dx0_parts[k]),
                                                 _ptr(self.ws16[k]), B, self.F, H, self.sizes[k], self.D, self.split, _stream()),
                      "rsx_cin_split_bwd_dx")
                continue
            if self.bf16 and wide:
                # (layer 0: dXk IS dX0 and nothing has been written there yet -- the other layers' shares arrive with the reduce)
                check(lib().rsx_cin_layer_bwd_dx_bf16_parts(_ptr(X0), _ptr(Xk), _ptr(self.w16[k]), _ptr(self.outs[k]), dout,
                                                            _ptr(self.gs), C.c_void_p(wout + 4 * self.offs[k]), _ptr(dxk), 0,
                                                            _ptr(self.dx0_parts[k]), _ptr(self.ws16[k]), B, self.F, H,
                                                            self.sizes[k], self.D, _stream()), "rsx_cin_layer_bwd_dx_bf16_parts")
                continue
            if self.bf16:       # w16[k] was prepared by this step's forward (the filters do not change in between)
                check(lib().rsx_cin_layer_bwd_dx_bf16(_ptr(X0), _ptr(Xk), _ptr(self.w16[k]), _ptr(self.outs[k]), dout, _ptr(self.gs),
                                                      C.c_void_p(wout + 4 * self.offs[k]), _ptr(dxk), acc_dxk, _ptr(dX0), acc_dx0,
                                                      _ptr(self.ws16[k]), B, self.F, H, self.sizes[k], self.D,
                                                      None if sweeps is None or sweeps[k] is None else C.byref(sweeps[k]),
                                                      _stream()), "rsx_cin_layer_bwd_dx_bf16")
                continue
            sw = None if sweeps is None or sweeps[k] is None else C.byref(sweeps[k])
            check(lib().rsx_cin_layer_bwd(_ptr(X0), _ptr(Xk), _ptr(P[f"cin.W{k}"]), _ptr(self.outs[k]), dout, _ptr(self.gs),
                                          C.c_void_p(wout + 4 * self.offs[k]), _ptr(dxk), acc_dxk, _ptr(dX0), acc_dx0,
                                          _ptr(P[f"cin.W{k}"].grad), _ptr(P[f"cin.c{k}"].grad), _ptr(self.dpre), B, self.F, H,
                                          self.sizes[k], self.D, sw, _stream()), "rsx_cin_layer_bwd")
        fuse_red = bool(self.split) and _lib.form("cin_dx0_ride") != "0"     # (0: the reduce as its own launch, A/B)
        if wide or self.split:  # dX0 = layer 0's dXk (already there) + every layer's tile partials, last layer first
            parts_h = (C.c_void_p * L)(*[self.dx0_parts[k].data_ptr() for k in range(L - 1, -1, -1)])
            if not fuse_red:
                check(lib().rsx_cin_dx0_reduce(parts_h, self._tiles_h, L, _ptr(dX0), 1, B, self.F, self.D, _stream()),
                      "rsx_cin_dx0_reduce")
        if self.bf16 or self.split:     # every layer's weight gradient in ONE launch
            jobs = (_lib.CinDwJob * L)()
            for k in range(L):
                Xk, H = (X0, self.F) if k == 0 else (self.outs[k - 1], self.sizes[k - 1])
                jobs[k] = _lib.CinDwJob(Xk.data_ptr(), self.ws16[k].data_ptr(), P[f"cin.W{k}"].grad.data_ptr(),
                                        P[f"cin.c{k}"].grad.data_ptr(), H, self.sizes[k], B if (wide or self.split) else 0)
            assert sweeps is None or len(sweeps) == L + 1, "bf16 CinNet.backward: L + 1 sweep slots"
            if self.split and fuse_red:   # the dX0 reduce rides in the weight-gradient launch (the same sums in the same order)
                check(lib().rsx_cin_split_bwd_dw_dx0(_ptr(X0), jobs, L, B, self.F, self.D, self.split, parts_h, self._tiles_h, L,
                                                     _ptr(dX0), 1, _stream()), "rsx_cin_split_bwd_dw_dx0")
                return dX0[:B]
            if self.split:
                check(lib().rsx_cin_split_bwd_dw(_ptr(X0), jobs, L, B, self.F, self.D, self.split, _stream()), "rsx_cin_split_bwd_dw")
                return dX0[:B]
            sw = None if sweeps is None or sweeps[L] is None else C.byref(sweeps[L])
            check(lib().rsx_cin_bwd_dw_bf16(_ptr(X0), jobs, L, B, self.F, self.D, sw, _stream()), "rsx_cin_bwd_dw_bf16")
        return dX0[:B]
