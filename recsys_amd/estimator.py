"""Estimator-shaped surface of the reference scripts on PyTorch-ROCm.

Mirrors what fm/fm.py:173-224 (and its four twins) use from `tensorflow_estimator`:
`Estimator(model_fn, model_dir, params, config)`, `.train/.evaluate/.predict`,
`train_and_evaluate(est, TrainSpec, EvalSpec)`, `RunConfig`, `EstimatorSpec`, `ModeKeys`.

Differences that are deliberate (MI355X-first):
  * model_fn runs eagerly on `cuda`; its variables live in the Estimator's `VariableStore`
    (flat HBM arenas) instead of a TF graph.
  * TRAIN mode returns `train_op` as a closure (backward + one fused Adam launch).  The Estimator
    captures `model_fn + train_op` ONCE per batch shape into a HIP graph and replays it per step,
    which removes the Python/launch overhead that dominates at batch 256.
  * MirroredStrategy (fm/fm.py:184-194) becomes one process per GPU + RCCL (recsys_amd.dist).
"""
import os
import queue
import threading
import time
from dataclasses import dataclass
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch

from . import _lib
from . import metrics as _metrics
from .ops import AdamTF1, DenseArena, EmbeddingArena


class ModeKeys:
    TRAIN = "train"
    EVAL = "eval"
    PREDICT = "infer"


@dataclass
class EstimatorSpec:
    mode: str
    predictions: Optional[Dict[str, torch.Tensor]] = None
    loss: Optional[torch.Tensor] = None
    train_op: Optional[Callable[[], None]] = None
    eval_metric_ops: Optional[Dict[str, Any]] = None
    export_outputs: Optional[Dict[str, Any]] = None


@dataclass
class RunConfig:
    """estimator.RunConfig fields the scripts set (fm/fm.py:187-194)."""
    save_checkpoints_steps: Optional[int] = None
    keep_checkpoint_max: int = 5
    log_step_count_steps: int = 100
    save_summary_steps: int = 200
    train_distribute: Any = None
    eval_distribute: Any = None
    use_hip_graph: bool = True
    device: str = "cuda"
    seed: int = 0
    adam_mode: str = "tf1_dense"    # 'tf1_dense' (reference semantics) | 'lazy_rows' (NOT TF semantics)


@dataclass
class TrainSpec:
    input_fn: Callable
    max_steps: Optional[int] = None


@dataclass
class EvalSpec:
    input_fn: Callable
    steps: Optional[int] = 100
    start_delay_secs: int = 120
    throttle_secs: int = 600


class PackedBatch:
    """(features, labels) of one batch packed into ONE contiguous byte buffer, so a training step needs a single
    host->device (or device->device) copy into the HIP graph's static input buffer instead of one per tensor."""

    _flat_layouts = {}      # (key of a FlatFeatures batch) -> (layout, nbytes): batches of one stream share them

    def __init__(self, features, labels, device=None, pin=False):
        flat_np = getattr(features, "flat", None)
        if flat_np is not None and device is None and not pin:
            # the input pipeline's reader already assembled the batch in this packing (input_pipeline.FlatFeatures): O(1)
            pkey = features.pack_key
            hit = PackedBatch._flat_layouts.get(pkey)
            if hit is not None and hit[1] == flat_np.nbytes and labels.shape[0] == hit[0][0][2][0] and \
                    labels.__array_interface__["data"][0] == flat_np.__array_interface__["data"][0]:
                self.layout, self.nbytes = hit
                self.flat = torch.from_numpy(flat_np)
                return
        items = [("__label__", labels)] + sorted(features.items())
        self.layout, off = [], 0
        arrs = []
        for k, v in items:
            t = torch.as_tensor(v).contiguous()
            nb = t.numel() * t.element_size()
            self.layout.append((k, t.dtype, tuple(t.shape), off, nb))
            arrs.append(t)
            off = (off + nb + 15) & ~15
        self.nbytes = off
        flat = self._adopt(items) if (device is None and not pin) else None
        if flat is not None and flat_np is not None and flat.data_ptr() == flat_np.ctypes.data and off == flat_np.nbytes:
            PackedBatch._flat_layouts[pkey] = (self.layout, self.nbytes)      # the next batch of this stream skips all of this
        if flat is None:
            # plain memcpy through numpy views (torch's CPU copy_ of > 32 KB wakes its thread pool: ~0.4 ms per part when the
            # pool is asleep, which it is between two batches)
            buf = np.zeros(off, np.uint8)
            for (k, dt, sh, o, nb), t in zip(self.layout, arrs):
                if nb:
                    buf[o:o + nb] = t.cpu().reshape(-1).view(torch.uint8).numpy()
            flat = torch.from_numpy(buf)
        if pin and torch.cuda.is_available():
            flat = flat.pin_memory()
        self.flat = flat if device is None else flat.to(device)

    def _adopt(self, items):
        """Zero-copy: the input pipeline's C++ reader assembles every batch in ONE flat host buffer laid out exactly like
        this packing (input_pipeline._criteo_batches: label | cont_log | ids, 16-byte aligned parts); when the arrays handed
        in are views of such a buffer at the packing's own offsets, wrap it instead of copying 40 KB per step."""
        arrays = [v for _, v in items]
        if not all(isinstance(v, np.ndarray) for v in arrays):
            return None
        root = arrays[0]
        while isinstance(root.base, np.ndarray):
            root = root.base
        if root.dtype != np.uint8 or root.ndim != 1 or not root.flags["C_CONTIGUOUS"]:
            return None
        p0 = arrays[0].ctypes.data
        off0 = p0 - root.ctypes.data
        if off0 < 0 or off0 + self.nbytes > root.nbytes or (p0 & 15):
            return None
        for (k, dt, sh, o, nb), v in zip(self.layout, arrays):
            if not v.flags["C_CONTIGUOUS"] or (nb and v.ctypes.data - p0 != o):
                return None
        return torch.from_numpy(root[off0:off0 + self.nbytes])

    def key(self):
        return tuple((k, str(dt), sh) for k, dt, sh, _, _ in self.layout)

    def to(self, device):
        o = PackedBatch.__new__(PackedBatch)
        o.layout, o.nbytes, o.flat = self.layout, self.nbytes, self.flat.to(device, non_blocking=True)
        return o

    def clone(self):
        o = PackedBatch.__new__(PackedBatch)
        o.layout, o.nbytes, o.flat = self.layout, self.nbytes, self.flat.clone()
        return o

    def views(self):
        feats, labels = {}, None
        for k, dt, sh, o, nb in self.layout:
            v = self.flat[o:o + nb].view(dt).view(sh)
            if k == "__label__":
                labels = v
            else:
                feats[k] = v
        return feats, labels


class LazyMeanLoss:
    """A TRAIN step's mean loss that nobody has asked for yet: the per-example terms sit in a device buffer (column `col` of
    `terms` [n, stride], written by the step's kernels) and are summed when the value is read -- float(), .item(), .tensor().
    The Estimator only touches a step's loss at log lines, so a step that would need a launch of its own just to reduce 256
    numbers (fm.py's fused forward + head, csrc/embedding.hip gather_fm_head_k) does not pay for it."""

    def __init__(self, terms, col, n):
        self.terms, self.col, self.n = terms, int(col), int(n)       # n = examples (the rows may be pre-added groups)

    def detach(self):
        return self

    def tensor(self):
        return (self.terms[:, self.col].double().sum() / self.n).to(torch.float32)

    def clone(self):
        return self.tensor()

    def reshape(self, *shape):
        return self.tensor().reshape(*shape)

    def item(self):
        return float(self.tensor().item())

    def __float__(self):
        return self.item()

    def __format__(self, spec):
        return format(float(self), spec)


class VariableStore:
    """The variables of one model: named embedding arenas + one flat dense arena + the optimizer."""

    def __init__(self, device, seed, adam_mode):
        self.device = torch.device(device)
        self.embeddings: Dict[str, EmbeddingArena] = {}
        self.dense: Optional[DenseArena] = None
        self.opt: Optional[AdamTF1] = None
        self.built = False
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(seed)
        self.adam_mode = adam_mode
        self.extra_segments = []     # model-specific optimizer segments (e.g. DIN tables)
        self.dp = None               # recsys_amd.dist.DataParallel when training data-parallel
        self.graph_safe_dp = False   # set by model code whose DP collectives run outside autograd (segmentable)
        self.dp_unique = False       # data parallel: the sparse exchange carries per-rank unique-row lists (dist.py, round 5)
        # Optimizer window (include/rsx.h rsx_adam_window): window_k = the longest run of consecutive TRAIN steps the
        # model's fused step can treat as one window (set by model code; 1 = every step on its own); `window` =
        # (k, position, [features of the k batches]) while the Estimator runs a step that belongs to one.
        self.window_k = 1
        self.window = None

    def window_of_step(self):
        """-> (k, position, features of the window's batches) of the TRAIN step being built; (1, 0, None) without a window."""
        return self.window if self.window is not None else (1, 0, None)

    def build(self, embeddings: Dict[str, EmbeddingArena], dense_shapes, dense_init, lr, storage_shapes=None):
        self.embeddings = embeddings
        self.dense = DenseArena(dense_shapes, self.device, storage_shapes)
        with torch.no_grad():
            for k, fn in dense_init.items():
                t = torch.empty(tuple(dense_shapes[k]))
                fn(t, self.gen)
                self.dense[k].copy_(t)
        self.opt = AdamTF1(lr=lr, device=self.device)
        self.built = True

    def adam_segments(self):
        lazy = self.adam_mode == "lazy_rows"
        segs = []
        for a in self.embeddings.values():
            segs += a.adam_segments(lazy)
        segs += self.extra_segments
        segs += self.dense.adam_segments()
        return segs

    def sort_ids_for_backward(self, arena, ids):
        """Dedup stage of the sparse gradient (ids only).  Data-parallel: over the all-gathered global batch.  (The fused
        TRAIN steps do not call this: their sort rides in the first tower launch.)"""
        if self.dp is not None:
            ids = self.dp.all_gather_rows(ids)
        arena.field_sort(ids)

    def minimize(self, loss):
        """optimizer.minimize(loss) (fm/fm.py:162-163) incl. MirroredStrategy's 1/N loss scaling and
        cross-replica gradient sum (Appendix A-12)."""
        if self.dp is not None:
            (loss / self.dp.world).backward()
            self.dp.all_reduce_sum(self.dense.grad)
        else:
            loss.backward()
        self.apply_gradients()

    def apply_gradients(self):
        self.opt.step(self.adam_segments())

    # -- checkpoint -------------------------------------------------------
    def state_dict(self):
        sd = {"opt_state": self.opt.state.cpu(), "dense": {k: getattr(self.dense, k).cpu() for k in ("flat", "m", "v")}}
        for name, a in self.embeddings.items():
            sd["emb." + name] = {k: getattr(a, k).cpu() for k in ("tables", "m_t", "v_t", "w1", "m_w", "v_w", "table", "m", "v")
                                 if getattr(a, k, None) is not None}
        return sd

    def load_state_dict(self, sd):
        with torch.no_grad():
            self.opt.state[:4].copy_(sd["opt_state"][:4])      # (beta powers, ticket, step; the arrival counters stay zero)
            for k in ("flat", "m", "v"):
                getattr(self.dense, k).copy_(sd["dense"][k])
            for name, a in self.embeddings.items():
                for k, v in sd["emb." + name].items():
                    getattr(a, k).copy_(v)


_CURRENT_STORE = []


def get_variable_store() -> VariableStore:
    """What tf.get_variable's implicit graph is to the reference: the calling Estimator's store."""
    if not _CURRENT_STORE:
        raise RuntimeError("model_fn must be called by an Estimator (no active VariableStore)")
    return _CURRENT_STORE[-1]


class Estimator:
    MAX_INFER_GRAPHS = 32      # EVAL / PREDICT input signatures that get a captured graph (and an entry in _graphs) at most

    def __init__(self, model_fn, model_dir=None, params=None, config: Optional[RunConfig] = None):
        self._n_infer_graphs = 0
        self.model_fn = model_fn
        self.model_dir = model_dir
        self.params = dict(params or {})
        self.config = config or RunConfig()
        self.store = VariableStore(self.config.device, self.config.seed, self.config.adam_mode)
        self._graphs = {}
        self._ring = {}        # pinned staging buffers for host batches (see _h2d)
        # captured instances of a streaming window that take turns (_train_window_packed): window w + 1 is staged into the
        # other instance while window w is queued or running (3 measured the same as 2: the stream is GPU-bound)
        self._window_sets = max(2, int(_lib.form("window_sets")))
        self._restored = False
        self._log_t = None
        self.dist = None       # recsys_amd.dist.DataParallel, set by attach_distributed()

    # -- plumbing ----------------------------------------------------------------------------
    def _call_model_fn(self, features, labels, mode):
        _CURRENT_STORE.append(self.store)
        try:
            return self.model_fn(features, labels, mode, self.params)
        finally:
            _CURRENT_STORE.pop()

    def _to_device(self, x):
        dev = self.store.device
        if isinstance(x, dict):
            return {k: self._to_device(v) for k, v in x.items()}
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        if isinstance(x, torch.Tensor):
            return x.to(dev, non_blocking=True)
        return x

    def _train_eager(self, features, labels):
        spec = self._call_model_fn(features, labels, ModeKeys.TRAIN)
        spec.train_op()
        # detach: a live loss would keep this step's autograd graph (and its AccumulateGrad nodes, bound to
        # this stream) alive into the next step, which breaks HIP-graph capture on the capture stream
        return spec.loss.detach()

    def _train_window(self, batches):
        """len(batches) consecutive TRAIN steps as ONE optimizer window: the model's fused step sorts the ids of all of
        them at the first step and sweeps the untouched rows of the optimizer state once for the whole window.  After the
        last step the variables equal those of len(batches) single steps bit for bit; between two steps of a window they
        are not a state any single-step run passes through, so nothing else (evaluate, checkpoint) may run in between.
        -> the loss of every step."""
        k = len(batches)
        feats = [f for f, _ in batches]
        losses = []
        for pos, (f, l) in enumerate(batches):
            self.store.window = (k, pos, feats) if k > 1 else None
            try:
                losses.append(self._train_eager(f, l))
            except BaseException:
                if k > 1:
                    # the rows no step of the window touches are already k steps ahead, the touched ones and global_step only
                    # `pos`: not a state any run passes through.  Poison the store: evaluate / predict / checkpoint refuse.
                    self.store.window_broken = True
                raise
            finally:
                self.store.window = None
        return losses

    def _check_consistent(self, what):
        if getattr(self.store, "window_broken", False):
            raise _lib.RsxError("Estimator.%s: an optimizer window was interrupted by an exception; the variables are between two "
                           "window boundaries (untouched rows ahead of the touched ones).  Restore from a checkpoint." % what)

    def _window_len(self):
        """Steps per optimizer window: what the model supports (store.window_k), RSX_ADAM_WINDOW overrides (1 = off)."""
        if not self.store.built or (self.store.dp is not None and not getattr(self.store, "window_dp", False)):
            return 1
        k = int(os.environ.get("RSX_ADAM_WINDOW", self.params.get("adam_window", self.store.window_k)))
        return max(1, min(k, self.store.window_k))

    def _dp_capture(self):
        from .dist import dp_capture
        return dp_capture(self.store.dp)

    def _use_graph(self):
        """HIP graphs are on unless the step has data-parallel collectives issued from inside autograd's backward
        (worker thread: the capture cannot be segmented there); RSX_DP_CAPTURE=1 captures the collectives too."""
        if not self.config.use_hip_graph:
            return False
        return self.store.dp is None or self.store.graph_safe_dp or self._dp_capture()

    def _capture(self, fn):
        """-> (replayable, fn's result).  Data-parallel steps become graph segments with eager RCCL calls between
        them (dist.SegmentedGraph); single-process steps one HIP graph."""
        torch.cuda.synchronize()
        if self.store.dp is not None and not self._dp_capture():
            from .dist import SegmentedGraph
            seg = SegmentedGraph()
            return seg, seg.capture(fn)
        graph = torch.cuda.CUDAGraph()
        # (data parallel with captured collectives: "thread_local" lets ProcessGroupNCCL's watchdog thread poll its events
        # while THIS thread captures -- see dist.SegmentedGraph._begin)
        kw = {"capture_error_mode": "thread_local"} if self.store.dp is not None else {}
        with torch.cuda.graph(graph, **kw):
            out = fn()
        return graph, out

    def _capture_step(self, features, labels):
        """The TRAIN step over static input tensors -> (replayable, static loss).  Data-parallel steps whose train_op is
        plain kernel calls (store.graph_safe_dp): only model_fn is captured; train_op -- pack, the gradient all-gather, 2-3
        launches -- is re-issued eagerly on every replay.  A graph launch costs ~9 us of start-up and ~8 us before the next
        un-captured operation begins (measured through RCCL at world 1): a graph around so few launches loses more than it saves."""
        if self.store.dp is not None and self.store.graph_safe_dp and not self._dp_capture() and \
                _lib.form("dp_eager_tail") == "1":
            seg, spec = self._capture(lambda: self._call_model_fn(features, labels, ModeKeys.TRAIN))
            seg.items.append(spec.train_op)      # re-executed (eagerly) by every replay, after the captured segments
            return seg, spec.loss.detach()
        return self._capture(lambda: self._train_eager(features, labels))

    def _shape_key(self, features, labels):
        return tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in features.items())) + (tuple(labels.shape),)

    def _train_step(self, features, labels=None):
        """One global step; captured into a HIP graph per input signature after 2 eager warm-up steps.
        `features` may be a PackedBatch (then one copy feeds the graph's static input buffer)."""
        if isinstance(features, PackedBatch):
            return self._train_step_packed(features)
        if not self._use_graph():
            return self._train_eager(features, labels)
        key = self._shape_key(features, labels)
        g = self._graphs.get(key)
        if g is None:
            g = {"warm": 0}
            self._graphs[key] = g
        if "graph" not in g:
            if g["warm"] < 2:
                g["warm"] += 1
                return self._train_eager(features, labels)
            g["feat"] = {k: v.clone() for k, v in features.items()}
            g["lab"] = labels.clone()
            graph, g["loss"] = self._capture_step(g["feat"], g["lab"])
            g["graph"] = graph      # capture executes nothing: the static buffers already hold this
            graph.replay()          # batch (cloned above), so replay once to apply its step
            return g["loss"]
        for k, v in features.items():
            g["feat"][k].copy_(v, non_blocking=True)
        g["lab"].copy_(labels, non_blocking=True)
        g["graph"].replay()
        return g["loss"]

    def _h2d(self, static, pb):
        """One batch -> the graph's static input buffer, ONE copy.  A pageable host batch (what an input pipeline hands over)
        goes through a ring of pinned staging buffers first: the H2D copy of pageable memory is synchronous -- the host would
        wait for the GPU to finish the previous window before it could even start packing the next one (measured end to end
        over TFRecord shards: 185 us per DeepFM step against 70 us of GPU time)."""
        src = pb.flat
        if src.device.type == "cpu" and not src.is_pinned() and torch.cuda.is_available():
            ring = self._ring.get(pb.nbytes)
            if ring is None:
                R = 4 * max(2, self.store.window_k)
                bufs = [torch.empty(pb.nbytes, dtype=torch.uint8).pin_memory() for _ in range(R)]
                ring = {"bufs": bufs, "np": [b.numpy() for b in bufs], "evs": [torch.cuda.Event() for _ in range(R)],
                        "used": [False] * R, "i": 0}
                self._ring[pb.nbytes] = ring
            i = ring["i"]
            ring["i"] = (i + 1) % len(ring["bufs"])
            if ring["used"][i]:
                ring["evs"][i].synchronize()         # the copy that last read this staging buffer has run (long ago)
            np.copyto(ring["np"][i], src.numpy())    # (a plain memcpy: torch's CPU copy_ of > 32 KB wakes its thread pool, ~0.4 ms)
            static.flat.copy_(ring["bufs"][i], non_blocking=True)
            ring["evs"][i].record()
            ring["used"][i] = True
            return
        static.flat.copy_(src, non_blocking=True)

    def _train_step_packed(self, pb):
        key = ("packed",) + pb.key()
        g = self._graphs.setdefault(key, {"warm": 0}) if self._use_graph() else None
        if g is not None and "graph" in g:
            # ONE copy per step, host or device -> the graph's static input buffer
            self._h2d(g["static"], pb)
            g["graph"].replay()
            return g["loss"]
        if pb.flat.device != self.store.device:
            pb = pb.to(self.store.device)
        if g is None:
            return self._train_eager(*pb.views())
        if "graph" not in g:
            if g["warm"] < 2:
                g["warm"] += 1
                return self._train_eager(*pb.views())
            g["static"] = pb.clone()
            graph, g["loss"] = self._capture_step(*g["static"].views())
            g["graph"] = graph
            graph.replay()
            return g["loss"]

    def train_resident(self, batches, steps, steps_per_graph=8):
        """Train `steps` steps over a list of HBM-resident PackedBatch objects (cycled in order).  Groups of
        `steps_per_graph` consecutive batches are captured into ONE HIP graph each that reads the resident
        buffers directly: no per-step input copy and one graph launch per group ("capture launch-bound inner
        loops in hipGraphs").  An input pipeline refills the buffers in place between replays."""
        n = len(batches)
        if self._use_graph() and self.store.dp is not None and not self._dp_capture():
            # data-parallel with eager collectives: one segmented step per RESIDENT batch, reading its buffer in place
            # (no per-step input copy); steps cannot be grouped because the collectives sit between the segments
            key = ("resident-dp", id(batches[0]), n)
            g = self._graphs.get(key)
            done = 0
            if g is None:
                for s in range(min(2, steps)):
                    self._train_eager(*batches[s % n].views())
                done = min(2, steps)
                g = {"steps": [self._capture_step(*b.views()) for b in batches], "wins": {}}
                # optimizer windows (model code that supports them under data parallelism sets store.window_dp): K consecutive
                # resident batches as ONE segmented capture -- the ids of all K batches are all-gathered at its start
                K = self._window_len()
                if K > 1 and n % K == 0:
                    for i in range(0, n, K):
                        g["wins"][i] = self._capture(
                            lambda i=i: self._train_window([batches[i + j].views() for j in range(K)]))
                    g["K"] = K
                self._graphs[key] = g
            loss = None
            s = done
            while s < steps:
                K = g.get("K", 0)
                if K and (s % n) % K == 0 and steps - s >= K:
                    seg, losses = g["wins"][s % n]
                    seg.replay()
                    loss = losses[-1]
                    s += K
                    continue
                seg, loss = g["steps"][s % n]
                # RSX_DP_PREFETCH=1: the next batch's ids all-gather is issued underneath this step (not after the last one:
                # nothing would consume it before the buffers may be refilled).  Off by default: through RCCL at world 1 the
                # async launch + stream hand-over costs 7 us more than it hides; whether real xGMI latency reverses that can
                # only be measured on a multi-GPU node.
                nxt = g["steps"][(s + 1) % n][0] if (s + 1 < steps and _lib.form("dp_prefetch") == "1") else None
                seg.replay(nxt)
                s += 1
            return loss if loss is not None else g["steps"][0][1]
        # (data parallel with captured collectives -- dist.dp_capture, the default over RCCL: they sit in the groups' graphs like any
        # other launch)
        dp_eager = self.store.dp is not None and not self._dp_capture()
        if not self._use_graph() or n % steps_per_graph != 0 or steps_per_graph <= 1 or dp_eager:
            for s in range(steps):
                loss = self._train_step(batches[s % n])
            return loss
        key = ("resident", id(batches[0]), n, steps_per_graph)
        g = self._graphs.get(key)
        if g is None:
            g = {"groups": {}, "tails": {}, "loss": None, "warm": 0}
            self._graphs[key] = g
        done = 0
        while g["warm"] < 2 and done < steps:                  # eager warm-up (allocator, lazy init), first call only
            g["loss"] = self._train_eager(*batches[done % n].views())
            g["warm"] += 1
            done += 1
        # The schedule is a function of (start, count) only: a head graph up to the next group boundary (first call: the
        # two eager warm-up steps leave the stream at batch 2), full groups, then ONE tail graph holding the remaining
        # (< steps_per_graph) steps -- batches are consumed strictly in order.  Every graph is captured before anything is
        # replayed, and a caller that times this function calls prepare_resident() first, so no capture ever falls
        # inside a timed region.
        for graph, loss in self._resident_schedule(g, batches, done, steps - done, steps_per_graph):
            graph.replay()
            g["loss"] = loss
        return g["loss"]

    def _resident_schedule(self, g, batches, start, steps, spg):
        n = len(batches)

        def capture(first, count):
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            K = self._window_len()
            # (data parallel with captured collectives: see _capture -- a communicator thread may poll events meanwhile)
            kw = {"capture_error_mode": "thread_local"} if self.store.dp is not None else {}
            with torch.cuda.graph(graph, **kw):
                k = 0
                nwin = -(-count // K)    # windows of the graph, as EVEN as possible (20 steps: 7 + 7 + 6 rather than 8 + 8 + 4:
                while k < count:         # a window's ONE sweep costs 57 us + ~3 us per step, so short windows are the dear ones)
                    w = -(-(count - k) // nwin)      # optimizer windows never cross a graph: the resident batches ARE the look-ahead
                    loss = self._train_window([batches[first + k + j].views() for j in range(w)])[-1]
                    k += w
                    nwin -= 1
            return graph, loss        # `loss`: the static output of the graph's last step

        def partial(first, count):
            if (first, count) not in g["tails"]:
                g["tails"][(first, count)] = capture(first, count)
            return g["tails"][(first, count)]

        sched, pos, left = [], start % n, steps
        if pos % spg and left > 0:
            cnt = min(spg - pos % spg, left)
            sched.append(partial(pos, cnt))
            pos, left = (pos + cnt) % n, left - cnt
        while left >= spg:
            if spg < left < spg + spg // 2 and pos + left <= n:
                # a full group + a short tail (20 steps at 16 per graph): ONE graph for both -- a graph launch costs ~9 us of
                # start-up and ~8 us at its end on this stack
                sched.append(partial(pos, left))
                pos, left = (pos + left) % n, 0
                break
            gi = pos // spg
            if gi not in g["groups"]:
                g["groups"][gi] = capture(pos, spg)
            sched.append(g["groups"][gi])
            pos, left = (pos + spg) % n, left - spg
        if left:
            sched.append(partial(pos, left))
        return sched

    def prepare_resident(self, batches, steps, steps_per_graph=8):
        """Capture every HIP graph train_resident(batches, steps, steps_per_graph) will replay, so that a timed region
        afterwards is pure replay.  Capturing itself executes nothing -- but when the resident graphs are still cold this runs
        the TWO eager warm-up training steps first (lazy initialisation, allocator warm-up): real optimizer steps."""
        n = len(batches)
        dp_eager = self.store.dp is not None and not self._dp_capture()
        if not self._use_graph() or n % steps_per_graph != 0 or steps_per_graph <= 1 or dp_eager:
            return
        g = self._graphs.get(("resident", id(batches[0]), n, steps_per_graph))
        if g is None or g["warm"] < 2:
            self.train_resident(batches, 2, steps_per_graph)       # the two eager warm-up steps
            g = self._graphs[("resident", id(batches[0]), n, steps_per_graph)]
        self._resident_schedule(g, batches, 0, steps, steps_per_graph)

    def _maybe_restore(self):
        if self._restored or not self.model_dir or not self.store.built:
            return
        self._restored = True
        from . import checkpoint
        checkpoint.restore_latest(self.model_dir, self.store)

    @property
    def global_step(self):
        return self.store.opt.global_step if self.store.built else 0

    # -- public API --------------------------------------------------------------------------
    def _train_window_packed(self, pbs, launcher=None):
        """One optimizer window over len(pbs) host (or device) batches -> loss of the last step: ONE graph replay.
        Host batches are memcpy'ed into one pinned staging buffer, and the captured window's FIRST NODE is a kernel that
        fetches that buffer into the window's static inputs (slices of one device allocation; csrc/embedding.hip
        rsx_copy_bytes), so that a window is one hipGraphLaunch (22-45 us of host time) and nothing else on the stream.
        Measured before it (ms per DeepFM step, bench.py --host_input, unprofiled): one hipMemcpyAsync per batch on a copy
        stream + two events per window 0.0786, one copy per window on the copy stream 0.0749, the same copy in line 0.0696,
        the kernel node 0.0680 (resident batches: 0.0655).  Two captured instances of the window (each with its own inputs
        and staging buffer) take turns, so that window w + 1 is staged while window w is queued or running.
        launcher (Estimator.train, opt-in): the replay is handed to the launch thread; -> None (the loss is the launch's)."""
        host = all(pb.flat.device.type == "cpu" for pb in pbs)
        key = ("packedwin", len(pbs), host) + pbs[0].key()
        g = self._graphs.setdefault(key, {"warm": 0, "sets": [], "turn": 0})
        if len(g["sets"]) == self._window_sets:
            if host:
                st = self._stage_window(g, pbs)
                if launcher is not None:
                    launcher.submit(lambda: self._launch_staged(st))
                    return None
                return self._launch_staged(st)
            if launcher is not None:
                launcher.drain()
            st = g["sets"][g["turn"]]
            g["turn"] = (g["turn"] + 1) % len(g["sets"])
            for sb, pb in zip(st["static"], pbs):          # device-resident batches: one D2D copy each, in stream order
                self._h2d(sb, pb)
            st["graph"].replay()
            return st["losses"][-1]
        if launcher is not None:
            launcher.drain()
        dev = [pb if pb.flat.device == self.store.device else pb.to(self.store.device) for pb in pbs]
        if g["warm"] < 1:
            g["warm"] += 1
            return self._train_window([pb.views() for pb in dev])[-1]
        stride = (dev[0].nbytes + 255) & ~255
        st = {"done": torch.cuda.Event(), "stride": stride, "free": threading.Event(),
              "all": torch.empty(len(dev) * stride, dtype=torch.uint8, device=self.store.device)}
        st["free"].set()
        st["static"] = []
        for i, pb in enumerate(dev):
            sb = PackedBatch.__new__(PackedBatch)
            sb.layout, sb.nbytes, sb.flat = pb.layout, pb.nbytes, st["all"][i * stride:i * stride + pb.nbytes]
            sb.flat.copy_(pb.flat)
            st["static"].append(sb)
        if host:
            st["pin"] = torch.empty(len(dev) * stride, dtype=torch.uint8).pin_memory()
            st["pin_np"] = st["pin"].numpy()
            st["pin"].copy_(st["all"])  # (the staging buffer starts as this window's batches: the graph's first node reads it)
            torch.cuda.synchronize()

        # The kernel node reads the pinned staging buffer through the device's mapping of host memory, right after plain host
        # stores: that needs coherent (fine-grained) pinned allocations, torch's default.  RSX_WINDOW_STAGE=memcpy (automatic
        # under HIP_HOST_COHERENT=0) makes the first node a hipMemcpyAsync instead (0.0696 vs 0.0680 ms per DeepFM step).
        stage_memcpy = _lib.form("window_stage") == "memcpy" or os.environ.get("HIP_HOST_COHERENT", "1") == "0"

        def window():
            if host and stage_memcpy:
                st["all"].copy_(st["pin"], non_blocking=True)
            elif host:                  # first node: the staged batches, fetched from pinned host memory by a kernel
                _lib.check(_lib.lib().rsx_copy_bytes(st["all"].data_ptr(), st["pin"].data_ptr(), st["all"].numel(),
                                                     torch.cuda.current_stream().cuda_stream), "rsx_copy_bytes")
            return self._train_window([sb.views() for sb in st["static"]])

        st["graph"], st["losses"] = self._capture(window)
        st["graph"].replay()            # capture executes nothing: the inputs already hold this window's batches
        st["done"].record(torch.cuda.current_stream())
        g["sets"].append(st)
        return st["losses"][-1]

    def _stage_window(self, g, pbs):
        """Host half of a window (the training thread): the next instance's staging buffer <- the window's batches."""
        st = g["sets"][g["turn"]]
        g["turn"] = (g["turn"] + 1) % len(g["sets"])
        st["free"].wait()                  # the launch that last used this instance has been issued (and `done` recorded)
        st["free"].clear()
        st["done"].synchronize()           # ... and the GPU has finished it (two windows ago): its first node read the buffer
        stride, pin = st["stride"], st["pin_np"]
        for i, pb in enumerate(pbs):       # (plain memcpys: torch's CPU copy_ of > 32 KB wakes its thread pool, ~0.4 ms)
            np.copyto(pin[i * stride:i * stride + pb.nbytes], pb.flat.numpy())
        return st

    def _launch_staged(self, st):
        """Device half of a window (the launch thread, or in line): ONE graph replay -- staged batches in, all steps -> loss of
        the last step."""
        st["graph"].replay()
        st["done"].record()
        st["free"].set()
        return st["losses"][-1]

    def train(self, input_fn, steps=None, max_steps=None):
        self._check_consistent("train")       # (a poisoned store must be restored from a checkpoint, not trained on)
        it = iter(input_fn())
        if self._use_graph() and _lib.form("input_thread") != "0":      # (measured: no gain next to the launch thread)
            depth = max(16, 2 * self._window_len())
            # (an iterator that outlives this call keeps its thread and whatever the thread has pulled ahead)
            it = it.threaded(depth) if isinstance(it, _Resumable) else _InputThread(it, depth)
        # optional thread that issues the staged windows (single replica, HIP graphs on).  Off by default: a window launch
        # costs the training thread 22-45 us per 8 steps (scripts/graph_launch_cost.py) and the stream is GPU-bound without
        # it; with a slow input pipeline it takes the launch off the fetching thread (e2e A/B: within the box-to-box noise)
        launcher = None
        if self._use_graph() and self.store.dp is None and _lib.form("launch_thread") != "0":
            launcher = _LaunchThread(self.config.device)
        try:
            self._train_loop(it, steps, max_steps, launcher)
        finally:
            if launcher is not None:
                launcher.close()
            _close_iter(it)
        return self

    def _train_loop(self, it, steps, max_steps, launcher):
        done = 0
        cfg = self.config
        self._log_t, log_step0 = time.time(), None
        held = []                # batches pulled ahead that did not go into the last window
        gs_host = None           # the global step, tracked on the host after one device read
        exhausted = False
        while True:
            if steps is not None and done >= steps:
                break
            if max_steps is not None and self.store.built:
                if gs_host is None:
                    gs_host = self.global_step
                if gs_host >= max_steps:
                    break
            # How many steps the next optimizer window may hold (look-ahead over the input pipeline, which runs ahead of
            # the device anyway): never across the end of training, a log line or a checkpoint -- between the steps of a
            # window the variables are not a state the step-by-step run passes through.
            room = self._window_len() if self._use_graph() else 1
            if steps is not None:
                room = min(room, steps - done)
            if max_steps is not None and gs_host is not None:
                room = min(room, max_steps - gs_host)
            for every in (cfg.log_step_count_steps, cfg.save_checkpoints_steps if self.model_dir else 0):
                if every:
                    room = min(room, every - done % every)
            while len(held) < room and not exhausted:
                try:
                    held.append(next(it))
                except StopIteration:
                    exhausted = True
            if not held:
                break
            features, labels = held[0]
            if not self.store.built:          # variables are created by the first model_fn call
                self._call_model_fn(self._to_device(features), self._to_device(labels), ModeKeys.PREDICT)
                self._maybe_restore()
                room = 1
            # one packed host buffer per batch -> one H2D copy each -> the HIP graph's static inputs
            win = [PackedBatch(*held[0])]
            while len(win) < room and len(win) < len(held):
                pb = PackedBatch(*held[len(win)])
                if pb.key() != win[0].key():   # e.g. the last, smaller batch of an epoch: a step of its own
                    break
                win.append(pb)
            if len(win) == 1:
                if launcher is not None:
                    launcher.drain()
                loss = self._train_step(win[0])
            else:                              # (one graph per window length: the full one, and the shorter ones that
                loss = self._train_window_packed(win, launcher)     # end at a log line / checkpoint / the end of training)
            features, labels = held[len(win) - 1]
            del held[:len(win)]
            done += len(win)
            if gs_host is not None:
                gs_host += len(win)
            gs = None
            if done % cfg.log_step_count_steps == 0 or (cfg.save_checkpoints_steps and done % cfg.save_checkpoints_steps == 0):
                if launcher is not None and loss is None:
                    loss = launcher.drain()
                gs = self.global_step
            if gs is not None and done % cfg.log_step_count_steps == 0:
                now = time.time()
                world = 1
                if self.store.dp is not None:
                    # every TRAIN step returns its REPLICA's mean loss (only the gradients carry MirroredStrategy's 1/N): the
                    # mean over replicas is the mean loss of the global batch, which is what gets logged
                    world = self.store.dp.world
                    loss = self.store.dp.all_reduce_sum(loss.detach().clone().reshape(1)) / world
                if self._is_chief():
                    if log_step0 is not None:
                        rate = (gs - log_step0) / max(now - self._log_t, 1e-9)
                        print("INFO:global_step/sec: %.4g  (examples/sec: %.6g)" % (rate, rate * labels.shape[0] * world), flush=True)
                    print("INFO:loss = %.7g, step = %d" % (float(loss), gs), flush=True)
                self._log_t, log_step0 = now, gs
            if gs is not None and cfg.save_checkpoints_steps and self.model_dir and \
                    done % cfg.save_checkpoints_steps == 0:
                self._save_checkpoint(gs)
        if launcher is not None:
            launcher.drain()
        if self.model_dir and self.store.built and done:
            self._save_checkpoint(self.global_step)

    def _infer_step(self, features, labels, mode):
        """One EVAL / PREDICT forward over a host (or device) batch -> (prob, loss or None, labels on the device).  With HIP
        graphs on, the forward is captured once per input signature over static input buffers (ONE packed copy per batch,
        one graph launch): evaluate() was 283 us per batch of eager launches against a 70 us TRAIN step."""
        has_lab = labels is not None
        if not self.store.built:          # variables are created by the first model_fn call
            self._call_model_fn(self._to_device(features), self._to_device(labels) if has_lab else None, ModeKeys.PREDICT)
        self._maybe_restore()
        if not self._use_graph():
            f, l = self._to_device(features), (self._to_device(labels) if has_lab else None)
            spec = self._call_model_fn(f, l if mode == ModeKeys.EVAL else None, mode)
            return spec.predictions["prob"], spec.loss, l
        pb = PackedBatch(features, labels if has_lab else np.zeros(0, np.float32))
        key = ("infer", mode, has_lab) + pb.key()
        g = self._graphs.get(key)
        if g is None:
            if self._n_infer_graphs >= self.MAX_INFER_GRAPHS:
                # a serving loop with ever-new request sizes: stop capturing (and stop remembering signatures), stay eager
                dev = pb if pb.flat.device == self.store.device else pb.to(self.store.device)
                f, l = dev.views()
                spec = self._call_model_fn(f, l if mode == ModeKeys.EVAL else None, mode)
                return spec.predictions["prob"], spec.loss, (l if has_lab else None)
            g = self._graphs[key] = {"warm": 0}
            self._n_infer_graphs += 1
        if "graph" in g:
            self._h2d(g["static"], pb)
            g["graph"].replay()
            return g["prob"], g["loss"], g["labels"]
        dev = pb if pb.flat.device == self.store.device else pb.to(self.store.device)
        if g["warm"] < 1:                 # first batch of this signature: eager (lazy initialisation, allocator warm-up)
            g["warm"] += 1
            f, l = dev.views()
            spec = self._call_model_fn(f, l if mode == ModeKeys.EVAL else None, mode)
            return spec.predictions["prob"], spec.loss, (l if has_lab else None)
        g["static"] = dev.clone()
        f, l = g["static"].views()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        kw = {"capture_error_mode": "thread_local"} if self.store.dp is not None else {}
        with torch.cuda.graph(graph, **kw):
            spec = self._call_model_fn(f, l if mode == ModeKeys.EVAL else None, mode)
        g["prob"], g["loss"], g["labels"] = spec.predictions["prob"], spec.loss, (l if has_lab else None)
        g["graph"] = graph
        graph.replay()                    # capture executes nothing: the static buffers already hold this batch
        return g["prob"], g["loss"], g["labels"]

    def evaluate(self, input_fn, steps=None):
        """Estimator.evaluate (fm/fm.py:216-221): AUC-200 / Accuracy / mean batch loss accumulated ON DEVICE by one
        launch per batch (metrics.EvalMetrics); the host synchronises once, when the counters are read back.
        Data-parallel: every rank evaluates its own shard of the eval stream and the integer counters are summed."""
        self._check_consistent("evaluate")
        met = _metrics.EvalMetrics(self.store.device)
        n = 0
        it = input_fn()
        try:
            with torch.no_grad():
                for features, labels in it:
                    if steps is not None and n >= steps:
                        break
                    prob, loss, lab = self._infer_step(features, labels, ModeKeys.EVAL)
                    met.update(lab, prob, loss)
                    n += 1
        finally:
            _close_iter(it)
        if self.store.dp is not None:
            met.all_reduce(self.store.dp)
        res = met.result()
        res = {"AUC": res["AUC"], "Accuracy": res["Accuracy"], "loss": res["loss"], "global_step": self.global_step}
        if self._is_chief():
            print("INFO:Saving dict for global step %d: AUC = %.7g, Accuracy = %.7g, global_step = %d, loss = %.7g"
                  % (res["global_step"], res["AUC"], res["Accuracy"], res["global_step"], res["loss"]), flush=True)
        return res

    def _is_chief(self):
        return self.store.dp is None or getattr(self.store.dp, "rank", 0) == 0

    def _save_checkpoint(self, step):
        """Replicas are bit-identical by construction: rank 0 writes, everybody waits (MirroredStrategy: the chief saves)."""
        from . import checkpoint
        self._check_consistent("checkpoint")
        if self._is_chief():
            checkpoint.save(self.model_dir, self.store, step, self.config.keep_checkpoint_max)
        if self.store.dp is not None:
            self.store.dp.barrier()

    def predict(self, input_fn):
        self._check_consistent("predict")
        it = input_fn()
        try:
            with torch.no_grad():
                for features, labels in it:
                    prob, _, _ = self._infer_step(features, labels, ModeKeys.PREDICT)
                    prob = prob.reshape(-1).cpu().numpy()
                    for p in prob:
                        yield {"prob": p}
        finally:
            _close_iter(it)        # the caller usually breaks out after a few results (fm/fm.py:216-219)


    def predict_examples(self, serialized, parse_fn=None, batch_size=4096):
        """Inference entry that takes what a serving client sends: a list of serialized `tf.train.Example` byte strings
        (deepfm/grpc_client.py:50-76 builds exactly these; the reference's exported SavedModel parses them with the
        script's feature_description, deepfm/deepfm.py:213-233) -> {'prob': float32 [n]}.
        parse_fn(list[bytes]) -> (features, labels); default: the Criteo parse spec of the params' embedding columns, or
        the DIN spec (din/din.py:44-57) when the params hold no feature columns."""
        from . import input_pipeline as ip
        self._check_consistent("predict_examples")
        if parse_fn is None:
            cols = self.params.get("embedding_feature_columns")
            if cols is not None:
                from .feature_columns import CriteoLayout
                layout = CriteoLayout.from_columns(cols)
                parse_fn = lambda ex: ip.parse_criteo_examples(ex, layout)          # noqa: E731
            else:
                P = int(self.params.get("hist_len", 100))
                parse_fn = lambda ex: ip.parse_din_examples(ex, P)                  # noqa: E731
        serialized = list(serialized)
        out = []
        with torch.no_grad():
            for s in range(0, len(serialized), batch_size):
                features, labels = parse_fn(serialized[s:s + batch_size])
                prob, _, _ = self._infer_step(features, labels, ModeKeys.PREDICT)
                out.append(prob.reshape(-1).float().cpu().numpy())
        return {"prob": np.concatenate(out) if out else np.zeros(0, np.float32)}


class _LaunchThread:
    """Issues the staged windows of Estimator.train (ONE graph replay each) in order, on a thread of its own (opt-in,
    RSX_LAUNCH_THREAD=1): torch releases the GIL inside replay(), so the training thread fetches and stages window w + 1
    meanwhile.  (It was written when a rocprofv3 --hip-runtime-trace showed hipGraphLaunch holding its caller for 499 us per
    window -- which turned out to be the profiler's own interception of the graph's 58 dispatches; unprofiled the launch
    costs 22-45 us, and the stream is GPU-bound either way.)"""

    def __init__(self, device):
        self._q = queue.Queue(maxsize=1)
        self._err = None
        self.last = None
        self._dev = torch.cuda.current_device() if torch.device(device).type == "cuda" else None
        self._t = threading.Thread(target=self._run, name="rsx-launch", daemon=True)
        self._t.start()

    def _run(self):
        while True:
            fn = self._q.get()
            try:
                if fn is None:
                    return
                if self._err is None:
                    if self._dev is not None and torch.cuda.current_device() != self._dev:
                        torch.cuda.set_device(self._dev)
                    self.last = fn()
            except BaseException as e:             # surfaces at the next submit / drain
                self._err = e
            finally:
                self._q.task_done()

    def _check(self):
        if not self._t.is_alive():
            raise RuntimeError("rsx-launch thread is gone") from self._err

    def submit(self, fn):
        if self._err is not None:
            self.drain()
        while True:
            self._check()
            try:
                self._q.put(fn, timeout=1.0)
                return
            except queue.Full:
                continue

    def drain(self):
        """Everything submitted has been issued to the stream -> result of the last launch."""
        with self._q.all_tasks_done:
            while self._q.unfinished_tasks:
                self._check()
                self._q.all_tasks_done.wait(timeout=1.0)
        if self._err is not None:
            e, self._err = self._err, None
            raise e
        return self.last

    def close(self):
        try:
            self.drain()
        except BaseException:
            pass
        if self._t.is_alive():
            self._q.put(None)
            self._t.join(timeout=5.0)
        self._err = None


class _InputThread:
    """Pulls batches from an input iterator on a thread of its own, a bounded number ahead of the consumer (opt-in,
    RSX_INPUT_THREAD=1).  The reader's `next` is a C call that releases the GIL, but the generator code around it and the
    training thread's own Python share the GIL: end to end it measured within the noise of the default (the C++ reader already
    runs `prefetch` batches ahead on its own threads), so it stays a knob for input_fns that do real Python work per batch."""

    _END = object()

    def __init__(self, it, depth):
        self._it = it
        self._q = queue.Queue(maxsize=max(2, int(depth)))
        self._stop = False
        self._t = threading.Thread(target=self._run, name="rsx-input", daemon=True)
        self._t.start()

    def _run(self):
        try:
            for item in self._it:
                if self._stop:
                    break
                while not self._stop:
                    try:
                        self._q.put((item, None), timeout=0.05)
                        break
                    except queue.Full:
                        continue
                if self._stop:
                    break
            item = (self._END, None)
        except BaseException as e:                 # surfaces in the consumer
            item = (self._END, e)
        _close_iter(self._it)                      # (by the thread that runs the generator)
        while not self._stop:
            try:
                self._q.put(item, timeout=0.05)
                return
            except queue.Full:
                continue

    def __iter__(self):
        return self

    def __next__(self):
        item, extra = self._q.get()
        if item is self._END:
            self._q.put((item, extra))             # (stay exhausted)
            if extra is not None:
                raise extra
            raise StopIteration
        return item

    def close(self):
        self._stop = True
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._t.join(timeout=5.0)


def _close_iter(it):
    """Stop a prefetching input iterator whose consumer leaves early (evaluate(steps=...), predict's caller breaking
    out): generators get close(), which runs their `finally` and stops the producer thread."""
    close = getattr(it, "close", None)
    if close is not None:
        close()


def train_and_evaluate(estimator: Estimator, train_spec: TrainSpec, eval_spec: EvalSpec, eval_every_steps=None):
    """estimator.train_and_evaluate (fm/fm.py:222): train until the input is exhausted / max_steps, evaluating
    at every checkpoint (the TF loop evaluates whenever a new checkpoint appears, throttled)."""
    every = eval_every_steps or estimator.config.save_checkpoints_steps
    res = None
    if not every:
        estimator.train(train_spec.input_fn, max_steps=train_spec.max_steps)
        return estimator.evaluate(eval_spec.input_fn, steps=eval_spec.steps)
    it_fn = _Resumable(train_spec.input_fn)
    while not it_fn.exhausted:
        estimator.train(it_fn, steps=every, max_steps=train_spec.max_steps)
        res = estimator.evaluate(eval_spec.input_fn, steps=eval_spec.steps)
        if train_spec.max_steps is not None and estimator.global_step >= train_spec.max_steps:
            break
    return res


class _Resumable:
    """Keeps one iterator alive across successive Estimator.train(steps=...) calls."""

    def __init__(self, input_fn):
        self.it = None
        self.input_fn = input_fn
        self.exhausted = False

    def __call__(self):
        if self.it is None:
            self.it = iter(self.input_fn())
        return self

    def __iter__(self):
        return self

    def threaded(self, depth):
        """Move the underlying iterator behind an input thread (once): batches pulled ahead stay queued between calls."""
        if not isinstance(self.it, _InputThread):
            self.it = _InputThread(self.it, depth)
        return self

    def __next__(self):
        try:
            return next(self.it)
        except StopIteration:
            self.exhausted = True
            raise
