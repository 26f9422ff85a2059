"""Data parallelism: one process per GPU over RCCL/xGMI (backend "nccl" on ROCm), replacing the
reference's only distribution mechanism, `tf.distribute.MirroredStrategy()` (fm/fm.py:184-194 and
twins).

Semantics kept from MirroredStrategy: every replica holds all variables and takes its own batch of
`--batch_size` examples; the loss is scaled by 1/N and gradients are SUMMED across replicas; BatchNorm
and dropout stay per replica; one global step per synchronous step.

MI355X design (tables are only 54 MB, replicated -- no row sharding / all-to-all):
  * dense grads: ONE flat all-reduce(sum) over the dense arena's gradient buffer (<= 0.4 MB, latency
    bound, so a single bucket) -- in the fused steps folded into the sparse collective below: the arena rides in
    front of the per-example block and the optimizer launch adds the replicas' arenas in rank order.
  * sparse (embedding) grads: TF concatenates the replicas' IndexedSlices and dedups in the optimizer.
    Here each rank all-gathers the batch ids (at step start, they are inputs) and ONE per-example gradient
    block [dense | dX | S | gy2 | gy1] (after backward), sent straight from a persistent send block that the
    kernels write in place; every rank then runs the same per-field sort + sorted segment-sum over the GLOBAL
    batch of N*b examples, reading every rank's block in place from the gathered buffer.  DP(N, b) is therefore
    the single-process computation on N*b examples by construction, in a fixed (rank, example) order,
    and every replica applies bit-identical updates.  On the 8-GPU xGMI full mesh an all-gather sends
    each peer its slice over a dedicated link (7 x ~153 GB/s) instead of a ring.
"""
import os
import warnings

import torch
import torch.distributed as dist


def is_distributed_env():
    return int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("RSX_FORCE_DIST") == "1"


def init_process_group(backend=None):
    """Reads RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torch.distributed.run).  nccl == RCCL on ROCm."""
    if dist.is_initialized():
        return
    if backend is None:
        # RSX_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses duplicate devices) -- used by the world-2 test of
        # the script-level data-parallel path on the single-GPU test box
        backend = os.environ.get("RSX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(backend)


# ---- one command -> all local GPUs ---------------------------------------------------------------------------------------------
# `tf.distribute.MirroredStrategy()` with no arguments (fm/fm.py:184-186; deepfm/readme.md:22-24) makes `python fm.py` train on
# EVERY GPU of the host.  Here a replica is a process, so the first process re-launches itself as N ranks (one per visible GPU)
# through torch.distributed.run on 127.0.0.1 and waits for them; a process that already is a rank (WORLD_SIZE set) never does.

def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def local_replica_count(n_devices=None, env=None):
    """How many ranks `--mirror true` must START from this process: 0 = none (this process already is a rank of a launched job,
    or the host has at most one GPU), else one per visible GPU.  RSX_MIRROR_REPLICAS=N overrides the device count (N <= 1: never
    spawn; N > device count: several ranks per GPU over gloo -- smoke runs on a one-GPU box)."""
    env = os.environ if env is None else env
    if "WORLD_SIZE" in env or env.get("RSX_FORCE_DIST") == "1":
        return 0
    if "RSX_MIRROR_REPLICAS" in env:
        n = int(env["RSX_MIRROR_REPLICAS"])
    else:
        n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) if n_devices is None else int(n_devices)
    return n if n > 1 else 0


def spawn_command(n, args, script=None, module=None, port=None, python=None):
    """The command line that runs `script args` (or `-m module args`) as n ranks on this host."""
    import sys
    assert (script is None) != (module is None)
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port())]
    cmd += ["-m", module] if module is not None else [script]
    return cmd + [str(x) for x in args]


def spawn_env(n, n_devices=None, env=None):
    """Environment of the spawned ranks: more ranks than GPUs (a one-GPU box) cannot use RCCL (it refuses two ranks on one
    device) -> gloo, unless the caller chose a backend."""
    env = dict(os.environ if env is None else env)
    nd = (torch.cuda.device_count() if torch.cuda.is_available() else 0) if n_devices is None else int(n_devices)
    if n > nd and "RSX_DIST_BACKEND" not in env:
        env["RSX_DIST_BACKEND"] = "gloo"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("RSX_MIRROR_REPLICAS", None)
    return env


def spawn_local_ranks(n, args, script=None, module=None):
    """Runs this program as n ranks (one per GPU) and returns their exit code."""
    import subprocess
    return subprocess.call(spawn_command(n, args, script=script, module=module), env=spawn_env(n))


def maybe_spawn_mirror(FLAGS, module, argv=None):
    """`--mirror true` on a multi-GPU host: re-launch as one rank per GPU and return their exit code; None: carry on in-process."""
    import sys
    if not getattr(FLAGS, "mirror", False):
        return None
    n = local_replica_count()
    if not n:
        return None
    if module == "__main__":        # (python -m recsys_amd.<script>: the module's real name is in its spec)
        spec = getattr(sys.modules["__main__"], "__spec__", None)
        module = spec.name if spec is not None else module
    print("INFO:--mirror: %d local GPUs -> %d data-parallel ranks (fm/fm.py:184-186 MirroredStrategy)" % (n, n), flush=True)
    return spawn_local_ranks(n, list(sys.argv[1:] if argv is None else argv), module=module)


class DirectComm:
    """RCCL called through the C ABI (include/rsx.h "Collectives of the data-parallel step", csrc/comm.cpp): ncclAllGather /
    ncclAllReduce enqueued on the CURRENT stream -- a node of the step's HIP graph when the stream is capturing, as
    MirroredStrategy's cross-replica sums are nodes of TF's training graph (fm/fm.py:184-194).  No ProcessGroupNCCL Work object,
    no watchdog thread polling events of a capturing stream.  The unique id travels through torch.distributed's key-value store
    (the launcher's rendezvous); torch's process group stays the CONTROL plane (barriers, the eval counters, bench timing)."""
    _n = 0

    def __init__(self, rank, world):
        import ctypes as C
        from . import _lib
        self.C, self.L = C, _lib.lib()
        L = self.L
        self._check(L.rsx_comm_available_h(None), "rsx_comm_available_h")
        store = dist.distributed_c10d._get_default_store()
        key = "rsx_rccl_unique_id_%d" % DirectComm._n          # (every rank creates its communicators in the same order)
        DirectComm._n += 1
        if rank == 0:
            uid = C.create_string_buffer(128)
            self._check(L.rsx_comm_unique_id_h(uid), "rsx_comm_unique_id_h")
            store.set(key, uid.raw)
            raw = uid.raw
        else:
            raw = bytes(store.get(key))
        h = C.c_void_p()
        self._check(L.rsx_comm_init_h(raw, int(rank), int(world), C.byref(h)), "rsx_comm_init_h")
        self.h, self.rank, self.world = h, int(rank), int(world)
        self._side = None
        # self-check before anything is captured around it: every rank contributes its own number, every rank must read them all
        # back in rank order (a mis-wired communicator -- wrong device, wrong id, a library copy that does not reach its peers --
        # fails HERE, loudly, and DataParallel falls back to torch.distributed)
        probe = torch.full((4,), self.rank, dtype=torch.int32, device="cuda")
        got = torch.empty(self.world * 4, dtype=torch.int32, device="cuda")
        self.all_gather(got, probe)
        torch.cuda.synchronize()
        want = torch.arange(self.world, dtype=torch.int32, device="cuda").repeat_interleave(4)
        if not torch.equal(got, want):
            raise _lib.RsxError("rsx_all_gather self-check: rank %d of %d read %s" % (self.rank, self.world, got.tolist()))

    def _check(self, rc, what):
        from . import _lib
        if rc != 0:
            msg = self.L.rsx_comm_last_error_h()
            raise _lib.RsxError("%s failed: %s (%d) %s" % (what, self.L.rsx_strerror(rc).decode(), rc, (msg or b"").decode()))

    def _st(self):
        return self.C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def all_gather(self, out, x):
        assert out.is_contiguous() and x.is_contiguous() and out.numel() * out.element_size() == self.world * x.numel() * x.element_size()
        self._check(self.L.rsx_all_gather(self.h, x.data_ptr(), out.data_ptr(), x.numel() * x.element_size(), self._st()), "rsx_all_gather")

    def all_reduce_sum(self, t):
        assert t.dtype == torch.float32 and t.is_contiguous()
        self._check(self.L.rsx_all_reduce_sum_f32(self.h, t.data_ptr(), t.data_ptr(), t.numel(), self._st()), "rsx_all_reduce_sum_f32")

    def all_reduce_all_gather(self, grad, out, x):
        assert grad.dtype == torch.float32 and grad.is_contiguous() and out.is_contiguous() and x.is_contiguous()
        self._check(self.L.rsx_all_reduce_all_gather(self.h, grad.data_ptr(), grad.numel(), x.data_ptr(), out.data_ptr(),
                                                     x.numel() * x.element_size(), self._st()), "rsx_all_reduce_all_gather")

    def side_stream(self):
        """The side HIP stream of RSX_DP_OVERLAP: an all-reduce issued from inside backward runs beside the lower layers'
        backward launches (fork / join by events -- legal under capture: the side stream joins the capture)."""
        if self._side is None:
            self._side = torch.cuda.Stream()
        return self._side

    def close(self):
        if self.h is not None and self.h.value:
            self.L.rsx_comm_destroy_h(self.h)
            self.h = None


def direct_wanted(group=None):
    """The data plane goes through RCCL directly (DirectComm) when the ranks own DIFFERENT devices -- torch's group is "nccl" --
    unless RSX_DP_DIRECT=0 (A/B: torch.distributed collectives, rounds 1-5).  Several ranks on one GPU (RSX_DIST_BACKEND=gloo:
    the one-GPU test box) stay on gloo: RCCL refuses two ranks on one device."""
    return torch.cuda.is_available() and os.environ.get("RSX_DP_DIRECT", "1") != "0" and dist.get_backend(group) == "nccl"


class SegmentedGraph:
    """A training step captured as HIP-graph SEGMENTS with the RCCL collectives launched eagerly between them:
    [graph | collective | eager callable]* replayed in order on the current stream.  The compute stays launch-free
    (graphs) while the collectives take RCCL's ordinary, un-captured path -- robust on every RCCL build, and the
    collectives' host cost (a few us each) hides behind the previous segment's GPU time.  The Estimator appends the
    step's train_op as an eager callable (a graph around its 2-3 launches costs more than it saves).
    RSX_DP_CAPTURE=1 captures the collectives into the graph instead (single segment).

    Rule for code running under capture: a collective's input and output tensors must exist BEFORE the break (they
    are then allocated from the graphs' shared private pool, so their addresses are the ones every replay uses)."""
    _active = None

    def __init__(self):
        self.items = []                 # torch.cuda.CUDAGraph | callable
        self.prefetch = None            # the step's _PrefetchableAllGather item, if any
        self._pool = None
        self._ctx = None
        self._graph = None

    def _begin(self):
        self._graph = torch.cuda.CUDAGraph()
        # capture_error_mode "thread_local": ProcessGroupNCCL's watchdog thread polls the events of collectives that are still
        # in flight (hipEventQuery) -- legal while THIS thread captures only in that mode.  With the default ("global") an
        # asynchronous all-reduce left pending across a segment boundary (RSX_DP_OVERLAP) aborted the process from the
        # watchdog: "operation not permitted when stream is capturing".
        self._ctx = torch.cuda.graph(self._graph, pool=self._pool, capture_error_mode="thread_local")
        self._ctx.__enter__()

    def _end(self):
        with warnings.catch_warnings():      # a step that OPENS with a collective ends an empty first segment: legal, and
            warnings.filterwarnings("ignore", message="The CUDA Graph is empty")     # replaying it is a no-op
            self._ctx.__exit__(None, None, None)
        if self._pool is None:
            self._pool = self._graph.pool()
        self.items.append(self._graph)
        self._ctx = self._graph = None

    def capture(self, fn):
        """Runs fn() under capture (fn's collectives call graph_break); returns fn's result."""
        assert SegmentedGraph._active is None
        SegmentedGraph._active = self
        self._begin()
        try:
            out = fn()
        finally:
            self._end()
            SegmentedGraph._active = None
        return out

    def run_eager(self, op):
        self._end()
        op()
        self.items.append(op)
        self._begin()

    def replay(self, next_seg=None):
        """next_seg: the step that will be replayed after this one (resident batches): its prefetchable collective (the ids
        all-gather, whose input depends on nothing this step computes) is issued as soon as this step's own has been
        consumed, so that it runs on RCCL's stream underneath this step's kernels."""
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()
                if it is self.prefetch and next_seg is not None and next_seg.prefetch is not None:
                    next_seg.prefetch.issue()


class _AsyncAllReduce:
    def __init__(self, t, group, comm=None):
        self.t, self.group, self.work, self.comm = t, group, None, comm

    def issue(self):
        if self.comm is not None:            # DirectComm: on the side stream, forked from / joined to the step's by events
            side = self.comm.side_stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.comm.all_reduce_sum(self.t)
            self.work = side
            return
        self.work = dist.all_reduce(self.t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self):
        if self.work is not None:
            if self.comm is not None:
                torch.cuda.current_stream().wait_stream(self.work)
            else:
                self.work.wait()
            self.work = None


def dp_capture(dp):
    """Are the step's collectives CAPTURED into its HIP graphs (one graph per group of steps, like the single-replica schedule)
    or issued eagerly between graph segments (SegmentedGraph)?  Captured is FASTER -- through RCCL at world 1 deepfm.py's step
    costs 0.0598 ms captured against 0.0790 eager (0.0571 without data parallelism): every segment boundary is a graph launch's
    ~9 us start-up + ~8 us tail on this stack -- but NOT the default (RSX_DP_CAPTURE=1 opts in): with this torch / RCCL pair
    ProcessGroupNCCL's watchdog thread sometimes polls an event that was last recorded in the capturing stream and aborts the
    process (xdeepfm.py, whose step has an all-reduce beside the all-gather: 4 of 12 world-1 runs; the four other models: 0 of
    ~40; profiles/r05_z_dp_world1_rccl.txt).  A crash in one run in three is not a default.  The emulation / loopback harnesses (tests/dp_harness.py)
    have no real collective to break a graph for: always captured."""
    if dp is not None and getattr(dp, "always_captured", False):      # (tests/dp_harness.py: no real collective to break a graph for)
        return True
    if dp is None:
        return False
    # Round 6: with the collectives issued through the C ABI on the step's own stream (DirectComm) there is no watchdog and
    # nothing to abort: captured is the DEFAULT.  Through torch.distributed (RSX_DP_DIRECT=0, or gloo) it stays opt-in.
    default = "1" if getattr(dp, "comm", None) is not None else "0"
    return os.environ.get("RSX_DP_CAPTURE", default) == "1"


def dp_overlap_enabled():
    """RSX_DP_OVERLAP=1 (opt-in, default off): the dense-gradient all-reduce is issued per tower layer from inside
    backward instead of riding in the step's single all-gather.  At DeepFM's size the step is latency-bound and one
    collective beats several, so this exists to be A/B-ed on a multi-GPU node, not as a default."""
    return os.environ.get("RSX_DP_OVERLAP", "0") == "1"


def overlap_ranges(dense, layer_names):
    """Splits a DenseArena into the float ranges that become final after each backward launch.
    layer_names: per tower layer l, the names of its variables (W, b, gamma, beta).  Returns (per_layer [(lo, hi)],
    rest [(lo, hi), ...]): layer l's covering range, and the maximal ranges covered by no layer (head, first-order
    bias, ...), which are final after the LAST layer's backward launch (the first one issued).  A layer whose variables
    interleave with another's makes the split impossible -> RsxError (the caller then keeps the default exchange)."""
    from . import _lib
    names = list(dense.names)
    end = {k: (dense.offsets[names[i + 1]] if i + 1 < len(names) else dense.n) for i, k in enumerate(names)}
    owner = {}
    per_layer = []
    for l, ks in enumerate(layer_names):
        ks = [k for k in ks if k in dense.offsets]
        lo, hi = min(dense.offsets[k] for k in ks), max(end[k] for k in ks)
        for k in names:
            if lo <= dense.offsets[k] < hi:
                if k not in ks:
                    raise _lib.RsxError(f"RSX_DP_OVERLAP: dense variable {k} lies inside tower layer {l}'s range")
                owner[k] = l
        per_layer.append((lo, hi))
    rest, cur = [], None
    for k in names:
        if k in owner:
            if cur is not None:
                rest.append(tuple(cur))
                cur = None
        elif cur is None:
            cur = [dense.offsets[k], end[k]]
        else:
            cur[1] = end[k]
    if cur is not None:
        rest.append(tuple(cur))
    return per_layer, rest


class _PrefetchableAllGather:
    """An all-gather whose input is ready before the step starts (the batch ids).  As a SegmentedGraph item it either waits
    for the asynchronous launch a previous step issued for it (`issue`) or runs synchronously."""

    def __init__(self, out, x, group, comm=None):
        self.out, self.x, self.group, self.work, self.comm = out, x, group, None, comm

    def issue(self):
        if self.work is None and self.comm is None:      # (DirectComm: in stream order, nothing to issue ahead)
            self.work = dist.all_gather_into_tensor(self.out, self.x, group=self.group, async_op=True)

    def __call__(self):
        if self.work is not None:
            self.work.wait()            # the current stream waits for RCCL's; the host does not block
            self.work = None
        elif self.comm is not None:
            self.comm.all_gather(self.out, self.x)
        else:
            dist.all_gather_into_tensor(self.out, self.x, group=self.group)


def graph_break(op):
    """Run the collective `op` now; under a SegmentedGraph capture, end the current segment first and record `op` so
    every replay re-issues it between the segments."""
    seg = SegmentedGraph._active
    if seg is None:                     # (eager, or a capture that takes the collectives too: dp_capture)
        op()
    else:
        seg.run_eager(op)


class DataParallel:
    def __init__(self, group=None):
        assert dist.is_initialized(), "call recsys_amd.dist.init_process_group() first"
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.comm = None
        if direct_wanted(group):
            try:
                self.comm = DirectComm(self.rank, self.world)
            except Exception as e:          # no RCCL to bind, or its init failed: torch.distributed carries the data plane
                warnings.warn("recsys_amd.dist: RCCL through the C ABI unavailable (%s); collectives go through torch.distributed "
                              "(eager between graph segments)" % (e,))
                self.comm = None

    def _ag(self, out, x):
        if self.comm is not None:
            self.comm.all_gather(out, x)
        else:
            dist.all_gather_into_tensor(out, x, group=self.group)

    # -- inputs -------------------------------------------------------------------------------
    def all_gather_rows(self, x, prefetchable=False):
        """x [b, ...] on every rank (same b) -> [N*b, ...] in rank order.  prefetchable: x is an INPUT of the step (ready
        before it starts); under a segmented capture the collective is then recorded so that the previous step may issue it
        early (SegmentedGraph.replay(next_seg=...))."""
        x = x.contiguous()
        out = torch.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        seg = SegmentedGraph._active
        if prefetchable and seg is not None and seg.prefetch is None:
            op = _PrefetchableAllGather(out, x, self.group, self.comm)
            seg.prefetch = op
            seg.run_eager(op)
        else:
            graph_break(lambda: self._ag(out, x))
        return out

    def all_gather_id_list(self, ids_list, prefetchable=False):
        """The id matrices of k local batches (one step, or the k steps of an optimizer window) -> the k GLOBAL ones [N*b, F] in
        rank order, with ONE collective (the per-example exchange's ids phase)."""
        if len(ids_list) == 1:
            return [self.all_gather_rows(ids_list[0], prefetchable=prefetchable)]
        b = ids_list[0].shape[0]
        allg = self.all_gather_rows(torch.stack(ids_list).unsqueeze(0))               # [N, k, b, F]
        return [allg[:, i].reshape(self.world * b, -1).contiguous() for i in range(len(ids_list))]

    def all_gather_entry_keys(self, keys, token=None, peer_fn=None):
        """din.py's per-example exchange: the sort keys [entries, 2] of the rank's batch -> [N * entries, 2] in rank order.  (token /
        peer_fn: EmulatedDataParallel derives its peers' keys from other resident batches with them.)"""
        return self.all_gather_rows(keys)

    def all_gather_keys(self, keys, arena=None, ids_list=None):
        """The ids-phase collective of the unique-list exchange: keys [1, k * KS] int32 (EmbeddingArena.ux_sort_pack: the rank's
        packed unique-row lists of the k batches of an optimizer window) -> [N, k * KS].  (arena / ids_list: what the key block
        was made from -- EmulatedDataParallel derives its peers' blocks from other batches with them.)"""
        return self.all_gather_rows(keys)

    # -- sparse gradient block ----------------------------------------------------------------
    @staticmethod
    def pack_widths(F, D, has_fm, has_w1):
        w = [("dX", F * D)]
        if has_fm:
            w += [("S", D), ("gy2", 1)]
        if has_w1:
            w += [("gy1", 1)]
        return w

    # -- zero-copy send block -------------------------------------------------------------------
    def make_send_block(self, dense, b_max, widths):
        """One persistent buffer [dense gradient arena | pad | the rank's block of the sparse exchange: up to b_max units]; the
        dense arena's grad views are moved into it.  widths: per-unit float counts of the block's parts in block order -- the
        unique-list exchange (round 5): units = the capT packed unique rows, parts G [capT, D] (, a second table set's), gw1
        [capT]; the per-example exchange: units = examples, parts e.g. dX F*D, S D, gy2 1, gy1 1."""
        n0 = (dense.n + 3) & ~3
        self._send = torch.zeros(n0 + b_max * sum(widths) + 4, device=dense.grad.device)
        self._send_n0, self._send_widths, self._send_dense = n0, list(widths), dense
        dense.rebind_grad(self._send)
        return self._send

    def send_bytes(self, b=None):
        """Bytes one rank contributes to the step's gradient collective (reported by bench.py); b = units of the per-example
        block (examples; din.py: entries = examples x (1 + history length)), default: those of the last exchange."""
        b = getattr(self, "_last_b", 0) if b is None else b
        return 4 * (self._send_n0 + b * sum(self._send_widths))

    def send_views(self, b):
        """The parts of the example block for a batch of b examples, as views of the send block: [b, w] each (w = 1: [b])."""
        out, o = [], self._send_n0
        for w in self._send_widths:
            v = self._send[o:o + b * w]
            out.append(v if w == 1 else v.view(b, w))
            o += b * w
        return out

    def gather_send_block(self, b, fold_dense=False, dense_done=False):
        """Exchanges [dense | block(b)] straight from the send block (no pack copy) and returns (views of RANK 0's parts
        inside the gathered buffer, blocks descriptor) for EmbeddingArena.segsum*(..., blocks=).
        Small dense arenas (< RSX_DP_ALLREDUCE_MIN_BYTES, default 1 MiB: DeepFM / DCN / FM, ~0.3 MB): the arena rides in
        front of the per-example block in ONE all-gather and the ranks' arenas are summed in rank order -- into the local
        arena by one launch here, or (fold_dense) inside the optimizer launch itself: a third return value (else None) then
        replaces DenseArena.adam_segments() (RSX_ADAM_DENSE with B = world replicas `stride` floats apart).  The step is
        latency-bound there, one collective beats two.
        Large arenas (xDeepFM's CIN filters: 3.3 MB): an all-gather would deliver N x the bytes of an all-reduce (26 MB per
        rank per step at N = 8), so the arena takes a true all-reduce(sum), issued ASYNCHRONOUSLY on RCCL's stream before the
        example block's all-gather and awaited after it -- the two collectives overlap each other (and whatever the caller
        launches before the optimizer).
        dense_done (RSX_DP_OVERLAP, see overlap_ranges / all_reduce_async): the arena was already summed in place by
        all-reduces issued from inside backward; only the example block is exchanged and the third value is None."""
        from . import _lib
        n0, n = self._send_n0, self._send_dense.n
        self._last_b = b
        L = b * sum(self._send_widths)
        d = self._send_dense
        big = n * 4 >= int(os.environ.get("RSX_DP_ALLREDUCE_MIN_BYTES", str(1024 * 1024)))
        if big or dense_done:
            Lp = (L + 3) & ~3
            x = self._send[n0:n0 + Lp].view(1, Lp)
            out = torch.empty((self.world, Lp), dtype=x.dtype, device=x.device)
            grad = self._send[:n]

            if dense_done:
                graph_break(lambda: self._all_gather_into(out, x))
            else:
                graph_break(lambda: self._overlapped_allreduce_allgather(grad, out, x))
            views, o = [], 0
            for w in self._send_widths:
                v = out[0, o:o + b * w]
                views.append(v if w == 1 else v.view(b, w))
                o += b * w
            self._keep = out
            return views, (b, Lp), None
        ln = (n0 + L + 3) & ~3                             # rank blocks stay 16-byte aligned
        out = self.all_gather_rows(self._send[:ln].view(1, ln))  # [N, ln]
        seg = None
        if fold_dense:
            seg = [dict(kind=_lib.RSX_ADAM_DENSE, n=d.n, var=d.flat, m=d.m, v=d.v, g=out[0, :n], B=self.world, stride=ln,
                        zero_grad=0)]
        else:
            torch.sum(out[:, :n], 0, out=d.grad)
        views, o = [], n0
        for w in self._send_widths:
            v = out[0, o:o + b * w]
            views.append(v if w == 1 else v.view(b, w))
            o += b * w
        self._keep = out
        return views, (b, ln), seg

    def gather_example_grads(self, dX, S=None, gy1=None, gy2=None, ids=None, dense=None, blocked=False):
        """Packs the per-example gradient block, all-gathers it once, and returns the global views
        dX_g [N*b, F*D], S_g [N*b, D]|None, gy1_g [N*b]|None, gy2_g [N*b]|None (contiguous).
        ids (int32 [b,F], optional) ride in the same block -> a 5th return value ids_g [N*b,F]: one collective per
        step instead of two.  The block travels as int32 BITS (floats bit-cast to int32, never the reverse): small
        integer ids viewed as fp32 are subnormals, which float copy kernels may flush to zero.
        dense (flat fp32 [n], optional): this replica's dense-gradient arena rides behind the example block and is
        summed over replicas IN PLACE, in rank order on every rank (bit-identical replicas) -- the dense all-reduce
        folded into the same collective: the step is latency-bound, so one collective beats two.
        blocked=True (with dense): nothing is copied after the collective -- the returned tensors are the arrays of RANK
        BLOCK 0 inside the gathered buffer plus a 5th value `blocks` = (b, floats between rank blocks) for
        EmbeddingArena.segsum*(…, blocks=blocks), which read every rank's block in place."""
        b, N = dX.shape[0], self.world
        parts = [dX]
        if gy2 is not None:
            parts += [S, gy2.reshape(b, 1)]
        if gy1 is not None:
            parts += [gy1.reshape(b, 1)]
        if ids is not None:
            assert dense is None
            parts = [p.contiguous().view(torch.int32) for p in parts] + [ids.contiguous()]
            g = self.all_gather_rows(torch.cat(parts, 1))
            gi = g
            g = g.view(torch.float32)
        elif dense is not None:
            L = sum(p.numel() for p in parts)
            pad = (-(L + dense.numel())) % 4                     # rank blocks stay 16-byte aligned for float4 loads
            tail = [dense] if pad == 0 else [dense, dense.new_zeros(pad)]
            out = self.all_gather_rows(torch.cat([p.reshape(-1) for p in parts] + tail).view(1, -1))   # [N, L+n(+pad)]
            torch.sum(out[:, L:L + dense.numel()], 0, out=dense)
            if blocked:
                views, o = [], 0
                for p in parts:
                    views.append(out[0, o:o + p.numel()])
                    o += p.numel()
                S_v = gy2_v = gy1_v = None
                k = 1
                if gy2 is not None:
                    S_v, gy2_v = views[1], views[2]
                    k = 3
                if gy1 is not None:
                    gy1_v = views[k]
                self._keep = out
                return views[0], S_v, gy1_v, gy2_v, (b, out.shape[1])
            # per-rank block = the parts back to back, each [b, w_i] row-major
            res, o = [], 0
            for p in parts:
                w = p.shape[1]
                res.append(out[:, o:o + b * w].reshape(N * b, w))
                o += b * w
            dX_g = res[0]
            S_g = gy2_g = gy1_g = None
            k = 1
            if gy2 is not None:
                S_g, gy2_g = res[1], res[2].reshape(-1)
                k = 3
            if gy1 is not None:
                gy1_g = res[k].reshape(-1)
            return dX_g, S_g, gy1_g, gy2_g
        else:
            pack = torch.cat(parts, 1) if len(parts) > 1 else dX
            g = self.all_gather_rows(pack)
        o = dX.shape[1]
        dX_g = g[:, :o].contiguous()
        S_g = gy2_g = gy1_g = None
        if gy2 is not None:
            D = S.shape[1]
            S_g = g[:, o:o + D].contiguous()
            gy2_g = g[:, o + D].contiguous()
            o += D + 1
        if gy1 is not None:
            gy1_g = g[:, o].contiguous()
            o += 1
        if ids is not None:
            return dX_g, S_g, gy1_g, gy2_g, gi[:, o:o + ids.shape[1]].contiguous()
        return dX_g, S_g, gy1_g, gy2_g

    def _overlapped_allreduce_allgather(self, grad, out, x):
        if self.comm is not None:       # one RCCL group on the step's stream: all-reduce(grad) beside all-gather(x)
            self.comm.all_reduce_all_gather(grad, out, x)
            return
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # captured collectives (dp_capture): synchronous calls only.  An async_op=True collective issued under capture hands
            # ProcessGroupNCCL's watchdog thread a Work whose end event was recorded in the capturing stream; its next poll
            # aborts the process ("operation not permitted on an event last recorded in a capturing stream" -- seen in 2 of 4
            # world-1 runs of xdeepfm.py).  Inside the graph the two collectives are nodes of RCCL's stream either way.
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(out, x, group=self.group)
            return
        work = dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        dist.all_gather_into_tensor(out, x, group=self.group)
        work.wait()                     # stream-level wait: the host does not block

    def _all_gather_into(self, out, x):
        self._ag(out, x)

    # -- dense gradients ------------------------------------------------------------------------
    def all_reduce_async(self, flat):
        """RSX_DP_OVERLAP: all-reduce(sum) of `flat` (a slice of the dense gradient arena whose producers have all been
        launched) issued NOW on RCCL's stream, asynchronously -- it runs underneath whatever the caller launches next
        (the lower layers' backward).  Returns a handle for wait_all().  Under a SegmentedGraph capture the issue is a
        segment break like every other collective, so this mode trades one graph segment for one per tower layer."""
        h = _AsyncAllReduce(flat, self.group, self.comm if (flat.is_cuda and flat.dtype == torch.float32) else None)
        graph_break(h.issue)
        return h

    def wait_all(self, handles):
        """The current stream waits for the collectives of all_reduce_async (no host block)."""
        hs = list(handles)
        graph_break(lambda: [h.wait() for h in hs])

    def all_reduce_sum(self, flat):
        if self.comm is not None and flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous():
            graph_break(lambda: self.comm.all_reduce_sum(flat))
        else:                           # (integer eval counters, float64 loss sums, CPU tensors: the control plane)
            graph_break(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group))
        return flat

    def barrier(self):
        if dist.get_backend(self.group) == "nccl":      # name the device: RCCL otherwise guesses it from the current context
            dist.barrier(group=self.group, device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier(group=self.group)


def window_global_ids(dp, window_features, key="ids"):
    """The ids of the k batches of an optimizer window (estimator.VariableStore.window) as the GLOBAL batches the optimizer
    sees: dp None -> the local ones; else ONE all-gather of the stacked local ids, then k [N*b, F] tensors in rank order."""
    ids = [f[key] for f in window_features]
    if dp is None:
        return ids
    return dp.all_gather_id_list(ids)


def attach_if_distributed(estimator):
    """`--mirror` (fm/fm.py:36,184-186): data-parallel when launched with WORLD_SIZE > 1."""
    if not is_distributed_env():
        return None
    init_process_group()
    dp = DataParallel()
    estimator.store.dp = dp
    estimator.dist = dp
    return dp
