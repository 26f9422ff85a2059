#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: examples/sec of DeepFM training on the Criteo 39-field pipeline
(d=16, DNN 100-100, batch 256 per replica; BASELINE configs[1]) on N MI355X of one node.

A "step" = one full training step (embedding gather + first-order + FM, DNN tower fwd/bwd with
BatchNorm + dropout 0.5, sorted segment-sum scatter, TF-1 non-lazy Adam over every variable) on one
synthetic pre-hashed batch already resident in HBM.  Optimizer semantics are the reference's
(`adam_mode=tf1_dense`): nothing is skipped inside the timed region.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the kernel that moves the bytes: the TF-faithful dense Adam sweep.  With optimizer windows (the
                  default: include/rsx.h rsx_adam_window) that is adam_window_k, ONE pass over the optimizer state per
                  window of k steps: bytes of the pass / mean launch duration, HIP events on the launch stream;
                  `single_step_sweep` = adam_multi_k, the one-step sweep every non-windowed path runs.
                  Step-level fractions, named for what they divide: tf1_equivalent_step_frac = the bytes a
                  step-by-step TF-1 run has to move (345 MB) / measured step time / peak; moved_bytes_step_frac = the bytes
                  THIS path moves per step (one window pass / window length + the model's per-example traffic) / step time /
                  peak -- the honest traffic fraction of the latency-bound step.
  configs      -- (N = 1) the other BASELINE.json configs timed in the same process, a few seconds each: fm bs 256,
                  dcn bs 4096, xdeepfm fp32 and bf16-CIN bs 256, din bs 1024: ms_per_step, examples_per_sec, the dominant
                  kernel by name (rocprofv3 tables under profiles/) and the same two step-level fractions (+ the MFMA
                  fraction of the CIN flops for xdeepfm).
  cpu_baseline -- the same step on PyTorch-CPU fp32 with every host core (oracle/torch_ref.py; N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--adam_mode", default="tf1_dense", choices=["tf1_dense", "lazy_rows"])
    p.add_argument("--no_graph", action="store_true")
    p.add_argument("--emulate_identical", action="store_true", help="--emulate_world with N IDENTICAL replicas (one batch tiled N "
                   "times: rounds 1-4's emulation; the global batch then has a single replica's unique rows)")
    p.add_argument("--emulate_world", type=int, default=0, help="profiling aid: time the per-rank COMPUTE of an N-GPU "
                   "data-parallel step on one GPU (collectives replaced by local tiling; not a throughput claim)")
    p.add_argument("--host_input", action="store_true", help="measurement aid: every step's batch starts in pinned HOST "
                   "memory and crosses PCIe inside the timed region (one packed copy per step, per-step graphs); the "
                   "reported `value` of the default run never includes this")
    p.add_argument("--repeats", type=int, default=5, help="the K-step timed region is run this many times; the median is reported")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--cpu_seconds", type=float, default=25.0)
    p.add_argument("--n_batches", type=int, default=64)
    p.add_argument("--model", default="deepfm", choices=["deepfm", "fm", "dcn", "xdeepfm", "din"],
                   help="deepfm = the BASELINE metric's config (configs[1]); the others are the remaining BASELINE configs "
                        "(dcn: bs 4096, 3 cross layers; xdeepfm: CIN 128,128; din: bs 1024, hist 100, K 32)")
    p.add_argument("--cin_bf16", action="store_true", help="xdeepfm: CIN contraction on the bf16 MFMA path (fp32 accumulate); the "
                   "line then reports dtype 'bf16 CIN operands, f32 accumulate, f32 elsewhere'")
    p.add_argument("--cin_split", type=int, default=None, choices=[0, 3, 4],
                   help="xdeepfm: CIN contraction on the 16-bit MFMA with split operands (csrc/cin_split.hip); 3 = three bf16 planes, every "
                        "product exact to 2^-23; 4 = two scaled fp16 planes forward / data gradients (half the MFMAs) -- both held to "
                        "the fp32 path's 1e-5 parity tests; 0 = the fp32 MFMA kernels; unset = xdeepfm.py's default (4)")
    p.add_argument("--no_overlap", action="store_true", help="profiling aid: plain path (stand-alone sort, segment-sum, ONE full "
                   "optimizer sweep) instead of sweep slices riding in the tower launches -- shows every kernel's own duration")
    p.add_argument("--no_configs", action="store_true", help="skip the `configs` list (the other BASELINE configs)")
    p.add_argument("--no_e2e", action="store_true", help="skip the end-to-end entry of `configs` (deepfm.py from TFRecord shards)")
    p.add_argument("--config_steps", type=int, default=320, help="timed steps per entry of the `configs` list (x 3 repeats)")
    p.add_argument("--steps_per_graph", type=int, default=16, help="training steps captured per HIP graph (1: per-step "
                   "graph fed by one D2D copy of the batch)")
    return p.parse_args()


def cpu_baseline(batches, layout, seconds):
    """the same DeepFM bs-256 TRAIN step on PyTorch-CPU fp32 with every host core
    (oracle/torch_ref.DeepFMCpuBaseline -- TensorFlow itself cannot be installed here): 20 warm-up steps, then 5 timed
    repeats of R steps (R scaled so that the leg stays inside `seconds`), median repeat.  `value` = the efficient
    variant (gathers + sparse gradients); `tf_literal` = the variant that builds the dense [B, 840 646] one-hot
    input_layer and multiplies it with the [R, 1] kernel, which is what fm/fm.py:117,121 makes TensorFlow do."""
    from oracle import init, torch_ref
    host_cores = os.cpu_count() or 1
    B = batches[0][0].shape[0]
    P = init.deepfm_params(0, 16, (100, 100), np.float32, layout.row_off)

    def timed(literal, warm, budget, max_per_rep):
        m = torch_ref.DeepFMCpuBaseline(P, layout.row_off, 2, 0.5, literal=literal)
        n = 0
        t0 = time.perf_counter()
        for _ in range(warm):
            m.step(batches[n % len(batches)][0], batches[n % len(batches)][1])
            n += 1
        per = (time.perf_counter() - t0) / max(warm, 1)
        reps = 5
        r = int(max(1, min(max_per_rep, budget / (reps * max(per, 1e-6)))))
        dts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for _ in range(r):
                m.step(batches[n % len(batches)][0], batches[n % len(batches)][1])
                n += 1
            dts.append(time.perf_counter() - t0)
        return r * B / sorted(dts)[reps // 2], r, warm

    # "All cores" is not the fastest setting on a many-core host: the step is a chain of small ops plus a memory-bound
    # sweep, and 256 intra-op threads measured 35 examples/s on the GPU box (7 s per step of barrier thrash).  So the
    # thread count is swept upwards and the BEST one is the baseline; every point is reported.
    sweep, best_t, best_per = [], 1, float("inf")
    probe = torch_ref.DeepFMCpuBaseline(P, layout.row_off, 2, 0.5, literal=False)
    for t in sorted(set(c for c in (4, 8, 16, 32, 64, 128, host_cores) if c <= host_cores)):
        torch.set_num_threads(t)
        probe.step(batches[0][0], batches[0][1])
        t0 = time.perf_counter()
        for k in range(3):
            probe.step(batches[k % len(batches)][0], batches[k % len(batches)][1])
        per = (time.perf_counter() - t0) / 3
        sweep.append((t, round(B / per, 1)))
        if per < best_per:
            best_t, best_per = t, per
        elif per > 2.0 * best_per:
            break
    del probe
    torch.set_num_threads(best_t)
    eff, r_e, w_e = timed(False, 20, seconds * 0.6, 40)          # 5 x 40 = the 200 timed steps of section 8(d)
    lit, r_l, w_l = timed(True, 2, seconds * 0.25, 4)
    return {"value": eff, "unit": "examples/sec", "cores": best_t, "host_cores": host_cores, "kind": "port",
            "thread_sweep_examples_per_sec": sweep,
            "tf_literal": {"value": lit, "unit": "examples/sec",
                           "what": "dense [B, 840646] one-hot input_layer x [R,1] kernel, dense kernel gradient"},
            "sample": "DeepFM bs%d TRAIN step (fwd + autograd bwd + TF-1 non-lazy Adam over all 345 MB of state), PyTorch-CPU "
                      "fp32, torch.set_num_threads(%d) = the best point of the thread sweep on this %d-core host; efficient "
                      "variant %d warm + 5 x %d timed steps (median repeat), TF-literal one-hot variant %d warm + 5 x %d; "
                      "TensorFlow itself is not installable here" % (B, best_t, host_cores, w_e, r_e, w_l, r_l)}


WORKLOADS = {"deepfm": "Criteo-39 d=16 DNN 100-100", "fm": "Criteo-39 d=16", "dcn": "Criteo-39 d=16 3 cross layers DNN 100-100",
             "xdeepfm": "Criteo-39 d=16 CIN 128,128 DNN 100-100", "din": "Amazon-Electronics-shaped hist_len=100 K=32"}
# the launch that takes the largest share of the step in the committed rocprofv3 tables (profiles/r0N_z_*_kernel_stats.txt, latest round)
DOMINANT = {"deepfm": "segsum_adam_k (scatter + touched-row Adam; latency-bound) / adam_window_k per window",
            "fm": "segsum_adam_k / adam_window_k per window", "dcn": "tower_bwd_k<true> (fp32 MFMA dW/dX tiles at bs 4096)",
            "xdeepfm": "cin_bwd_dw_k / cin_bwd_dx2_k (fp32 MFMA)", "xdeepfm_bf16": "cin_bwd_dw_bf16_k (bf16 MFMA)",
            "xdeepfm_x3": "cin_split_dw_k<3> / cin_split_dx8_k<3,4> (bf16 MFMA, 3 planes per operand)",
            "xdeepfm_x4": "cin_split_dw_k<3> (3 bf16 planes) / cin_split_dx8_k<4,4> (2 fp16 planes per operand)",
            "din": "din_attn_bwd_k (fp32 MFMA attention MLP backward; the forward runs on the bf16 MFMA with split operands)"}


def per_example_bytes(model):
    """Algorithmic HBM bytes per example and step outside the optimizer sweep: gather 5 148 B +
    scatter 5 148 B per Criteo example (x 2 table sets for xdeepfm), + 2 x 2 x 2 496 B for dcn's cross layers (fwd, bwd),
    DIN pool 27 456 B fwd + the same backward."""
    return {"deepfm": 2 * 5148, "fm": 2 * 5148, "dcn": 2 * 5148 + 4 * 2496, "xdeepfm": 4 * 5148, "din": 2 * 27456}[model]


def sweep_bytes(est, wk):
    """(tf1_equivalent bytes per step, bytes of ONE window pass): 24 B per table / first-order element, 32 B per dense
    element; a window pass adds the wk slot maps (4 B per row each)."""
    from recsys_amd.ops import EmbeddingArena
    store = est.store
    segs = store.adam_segments()
    n_sparse = sum(int(sg["n"]) * int(sg.get("d", 1) or 1) for sg in segs if sg["kind"] in (1, 2))
    alg = 24 * n_sparse + 32 * store.dense.n
    arenas = [x for x in store.embeddings.values() if isinstance(x, EmbeddingArena)]
    rows = sum(int(ar.R) for ar in arenas if getattr(ar, "_sort_owner", None) is None)
    return alg, 24 * n_sparse + 4 * wk * rows, n_sparse, arenas


def cin_mode(cin_bf16, cin_split=None):
    """False (fp32 MFMA) | True (bf16 operands) | 'x1'..'x4' (split operands, csrc/cin_split.hip).  --cin_split unset: xdeepfm.py's
    own default (mode 4) unless --cin_bf16; --cin_split 0: the fp32 MFMA kernels."""
    if cin_split is None:
        if cin_bf16:
            return True
        from recsys_amd.xdeepfm import default_cin_split
        d = default_cin_split()
        return ("x%d" % d) if d else False
    return ("x%d" % cin_split) if cin_split else bool(cin_bf16)


def cin_ran(est, asked):
    """The CIN arithmetic that actually RAN (est.store.cin after build_variables: the flag may fall back to the fp32 MFMA kernels
    outside the split kernels' envelope), in cin_mode()'s vocabulary; `asked` for models without a CIN."""
    c = getattr(est.store, "cin", None)
    if c is None:
        return asked
    if getattr(c, "split", 0):
        return "x%d" % c.split
    return bool(getattr(c, "bf16", False))


CIN_DTYPE = {False: "f32", True: "bf16 CIN operands / f32 accumulate, f32 elsewhere",
             "x3": "f32; CIN products as 3 bf16 planes per operand on the bf16 MFMA (6 MFMAs per k-step, exact to 2^-23: fp32-grade), f32 accumulate",
             "x4": "f32; CIN forward / data-gradient products as 2 scaled fp16 planes per operand on the fp16 MFMA (3 MFMAs per k-step, "
                   "2^-22-grade: held to the fp32 path's tolerances), weight gradients as 3 bf16 planes, f32 accumulate"}
CIN_TAG = {False: "", True: " --cin_bf16", "x3": " --cin_split 3", "x4": " --cin_split 4"}
CIN_KEY = {False: "", True: "_bf16", "x3": "_x3", "x4": "_x4"}


def step_fractions(est, model, B, ms_per_step, wk, cin_bf16=False):
    alg, pass_bytes, _, arenas = sweep_bytes(est, wk)
    windowed = wk > 1 and bool(arenas)
    moved = (pass_bytes / wk if windowed else alg) + B * per_example_bytes(model)
    out = {"tf1_equivalent_step_frac": round(alg / (ms_per_step * 1e-3) / 8e12, 4),
           "moved_bytes_step_frac": round(moved / (ms_per_step * 1e-3) / 8e12, 4),
           "moved_bytes_per_step": int(moved), "tf1_equivalent_bytes_per_step": int(alg)}
    if model == "xdeepfm":
        flops = 3 * 2 * B * 16 * (39 * 39 * 128 + 39 * 128 * 128)          # fwd x 3 with backward
        # 16-bit MFMAs issued per algorithmic k-step (x4: 3 in the forward and the data gradients, 6 in the weight gradients)
        terms = {"x3": 6, "x4": 4}.get(cin_bf16, 1)
        peak = 2.5e15 if cin_bf16 else 157.3e12
        # ISSUED: the 16-bit MFMA flops the kernels execute (terms plane products per algorithmic product) against the dense bf16
        # peak; USEFUL: the algorithmic (fp32-equivalent) flops of SURVEY 8(d) against the same peak -- both, side by side
        out["cin_mfma_step_frac"] = round(terms * flops / (ms_per_step * 1e-3) / peak, 4)
        out["cin_mfma_step_frac_useful"] = round(flops / (ms_per_step * 1e-3) / peak, 4)
        out["cin_mfma_step_frac_is"] = "issued MFMA flops / step time / peak" if terms > 1 else "algorithmic flops / step time / peak"
        out["cin_flops_per_step"] = flops
        if terms > 1:
            out["cin_mfma_flops_issued_per_step"] = terms * flops
            out["cin_useful_vs_fp32_mfma_peak"] = round(flops / (ms_per_step * 1e-3) / 157.3e12, 4)
        out["mfma_peak"] = "2.5 PF dense bf16" if cin_bf16 else "157.3 TF fp32"
    return out


def time_config(a, model, batch_size, cin_bf16, dp, emu, rank, dev, steps, warmup, repeats):
    """Builds the Estimator of one BASELINE config over HBM-resident synthetic batches, runs `warmup` untimed steps (graph
    capture included) and `repeats` timed regions of EXACTLY `steps` steps, each bracketed by barrier + synchronize."""
    from recsys_amd import dcn, deepfm, din, fm, synthetic, xdeepfm
    from recsys_amd.estimator import Estimator, PackedBatch, RunConfig
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    B = batch_size
    if model != "deepfm" and batch_size == 256:
        B = {"dcn": 4096, "din": 1024}.get(model, 256)
    linear = {"deepfm": "indicator_all", "fm": "indicator_all", "dcn": "numeric", "xdeepfm": "numeric+indicator"}.get(model)
    lin, emb = build_feature_columns(16, linear) if linear else (None, None)
    params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 32 if model == "din" else 16,
              "learning_rate": 1e-3, "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": B,
              "cross_layers": {"dcn": 3, "xdeepfm": "128,128"}.get(model), "cin_bf16": cin_bf16 is True,
              "cin_split": int(cin_bf16[1]) if isinstance(cin_bf16, str) else 0}     # (explicit: False = the fp32 MFMA kernels)
    if a.no_overlap:
        params["overlap_adam"] = False
    mfn = {"deepfm": deepfm.model_fn, "fm": fm.model_fn, "dcn": dcn.model_fn, "xdeepfm": xdeepfm.model_fn, "din": din.model_fn}[model]
    cfg = RunConfig(use_hip_graph=not a.no_graph, adam_mode=a.adam_mode, device=str(dev), seed=1234)
    est = Estimator(mfn, None, params, cfg)
    if dp is not None or emu is not None:
        est.store.dp = dp or emu
        est.dist = dp or emu
    layout = CriteoLayout.from_columns(emb) if emb else None
    if model == "din":
        rng = np.random.default_rng(synthetic.SEED + rank)
        raw = [synthetic.din_batch(rng, B) for _ in range(a.n_batches)]
        host = None
        # id features narrowed to int32 on the host, as din.input_fn delivers them (ids_int32=True)
        feats = [PackedBatch({k: v.astype(np.int32) for k, v in b.items() if k != "label"}, b["label"], device=dev) for b in raw]
    else:
        host = synthetic.criteo_id_batches(layout, a.n_batches, B, seed=synthetic.SEED + rank)
        # one packed HBM-resident buffer per batch (ids [+ log-values] + labels): no per-step input copy
        feats = [PackedBatch({"ids": i, "cont_log": c} if model == "xdeepfm" else {"ids": i}, y, device=dev) for i, y, c in host]
    # variables are created on the first call; then W untimed warm-up steps (includes graph capture)
    with torch.no_grad():
        est._call_model_fn(feats[0].views()[0], None, "infer")
    if emu is not None and not a.emulate_identical:
        # the peers of resident batch i are REAL other batches (i + r n / N) -- the global step then touches the rows N
        # different batches touch, not one batch's rows N times; their packed unique-row lists are
        # computed here, once, outside the timed region
        n = len(feats)
        tok = "i_id" if model == "din" else "ids"
        views = [f.views()[0] for f in feats]
        peers = {views[i][tok].data_ptr(): [views[(i + (r + 1) * max(1, n // emu.world)) % n] for r in range(emu.world - 1)]
                 for i in range(n)}
        key_fn = None
        if getattr(est.store, "dp_unique", False):
            if model == "din":
                key_fn = est.store.din.ux_peer_keys
            else:
                ar = est.store.embeddings["input_layer"]
                key_fn = lambda pf, ar=ar: ar.ux_peer_keys(pf["ids"])
        emu.set_peers(peers, key_fn)
        if model == "din" and not getattr(est.store, "dp_unique", False):
            emu._entry_fn = est.store.din.peer_entry_keys
        emu.warm_keys()
        torch.cuda.synchronize()
    host_pbs = None
    if a.host_input:
        host_pbs = [PackedBatch(*f.to("cpu").views(), pin=True) for f in feats]

    def run(nsteps):
        if host_pbs is not None:        # what Estimator.train does with host batches: K copies + ONE graph replay per optimizer window
            K, s = est._window_len(), 0
            while s < nsteps:
                k = min(K, nsteps - s)
                if k > 1:
                    loss = est._train_window_packed([host_pbs[(s + j) % len(host_pbs)] for j in range(k)])
                else:
                    loss = est._train_step(host_pbs[s % len(host_pbs)])
                s += k
            return loss
        if a.steps_per_graph > 1:
            return est.train_resident(feats, nsteps, a.steps_per_graph)
        for s in range(nsteps):
            loss = est._train_step(feats[s % len(feats)])
        return loss

    run(warmup)
    if host_pbs is None and a.steps_per_graph > 1:
        # every HIP graph the timed schedule replays is captured HERE (capturing executes nothing), so the timed
        # region below is pure replay whatever `steps % steps_per_graph` is
        est.prepare_resident(feats, steps, a.steps_per_graph)

    def sync():
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
            torch.cuda.synchronize()

    # EXACTLY `steps` steps per timed region, bracketed by barrier + synchronize on both sides, MAX over ranks; the
    # region is repeated `--repeats` times back to back and the MEDIAN repeat is reported (a 20-step region is 2 ms:
    # one repeat is at the mercy of a single clock ramp or host hiccup).  All repeats are listed in config.
    dts, rank_dts = [], []
    for _ in range(max(1, repeats)):
        sync()
        t0 = time.perf_counter()
        loss = run(steps)
        torch.cuda.synchronize()
        if dp is not None:
            dp.barrier()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dp is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            allt = [torch.zeros_like(t) for _ in range(dp.world)]
            torch.distributed.all_gather(allt, t)
            rank_dts.append([float(x.item()) for x in allt])
            dt = max(rank_dts[-1])                      # MAX over ranks
        dts.append(dt)
    med = sorted(range(len(dts)), key=lambda i: dts[i])[len(dts) // 2]
    dt = dts[med]
    return {"est": est, "rank_dts": rank_dts[med] if rank_dts else None, "B": B, "dt": dt, "dts": dts, "cin": cin_ran(est, cin_bf16) if model == "xdeepfm" else False, "final_loss": float(loss), "host": host, "layout": layout, "feats": feats}


def other_configs(a, rank, dev):
    """The remaining BASELINE.json configs, each timed in this process: `config_steps` steps x 3 repeats after a warm-up
    that includes every graph capture (median repeat)."""
    import gc
    out = []
    for model, bf16 in (("fm", False), ("dcn", False), ("xdeepfm", False), ("xdeepfm", "x3"), ("xdeepfm", "x4"), ("xdeepfm", True), ("din", False)):
        if model == a.model and bf16 == cin_mode(a.cin_bf16, a.cin_split):
            continue
        steps = a.config_steps if model != "din" else max(64, a.config_steps // 2)
        try:
            t = time_config(a, model, 256, bf16, None, None, rank, dev, steps, 64, 3)
        except Exception as e:                                   # a config that fails must not take the headline line with it
            out.append({"workload": "%s.py%s" % (model, CIN_TAG[bf16]), "error": repr(e)[:300]})
            continue
        ms = t["dt"] / steps * 1e3
        wk = t["est"]._window_len() if (not a.no_graph and not a.no_overlap) else 1
        if model == "xdeepfm":
            bf16 = t["cin"]
        e = {"workload": "%s.py %s bs=%d%s, full train step (fwd+bwd+TF1 Adam), adam_mode=%s" %
                         (model, WORKLOADS[model], t["B"], CIN_TAG[bf16].replace(" --", ", "), a.adam_mode),
             "dtype": CIN_DTYPE[bf16],
             "ms_per_step": round(ms, 5), "examples_per_sec": round(t["B"] * steps / t["dt"], 1), "steps": steps,
             "timed_repeats_ms_per_step": [round(x / steps * 1e3, 5) for x in t["dts"]], "adam_window": wk,
             "final_loss": round(t["final_loss"], 5), "dominant_kernel": DOMINANT.get(model + CIN_KEY[bf16])}
        e.update(step_fractions(t["est"], model, t["B"], ms, wk, bf16))
        try:
            e["launches_per_step"] = launches_per_step(t["est"], t["feats"], wk)
        except Exception as ex:
            e["launches_per_step"] = None
            e["launches_per_step_error"] = repr(ex)[:200]
        dk = dominant_kernel_fraction(model + CIN_KEY[bf16])
        if dk is not None:
            e["dominant_kernel_roofline"] = dk
        out.append(e)
        del t
        gc.collect()
        torch.cuda.empty_cache()
    return out


def e2e_config(a, dev, resident_ms):
    """SURVEY 8(f-1) in front of the driver: deepfm.py bs 256 END TO END -- TFRecord shards on disk (written to
    a temp dir before the clock starts) -> the C++ reader (framing, CRC-32C, Example parse, FarmHash / bucketize, batching) ->
    Estimator.train (optimizer windows, one H2D copy per batch, HIP graphs): host parse and PCIe are INSIDE the timed region.
    A `configs` entry, never `value`."""
    import tempfile
    from recsys_amd import deepfm, synthetic
    from recsys_amd import input_pipeline as ip
    from recsys_amd.estimator import Estimator, RunConfig
    from recsys_amd.feature_columns import CriteoLayout, build_feature_columns
    bs, n = 256, 262144
    lin, emb = build_feature_columns(16, "indicator_all")
    layout = CriteoLayout.from_columns(emb)
    with tempfile.TemporaryDirectory() as d:
        rng = np.random.default_rng(0)
        files = []
        for k in range(4):
            label, cont, cat = synthetic.criteo_raw_batch(rng, n // 4)
            pth = os.path.join(d, "part-r-%05d" % k)
            ip.write_criteo_shard(pth, label, cont, cat)
            files.append(pth)
        params = {"linear_feature_columns": lin, "embedding_feature_columns": emb, "embedding_size": 16, "learning_rate": 1e-3,
                  "dropout": 0.5, "deep_layers": "100,100", "max_batch_size": bs}
        est = Estimator(deepfm.model_fn, None, params, RunConfig(device=str(dev), seed=1, log_step_count_steps=10 ** 9, adam_mode=a.adam_mode))
        threads = min(32, os.cpu_count() or 1)
        fn = lambda: ip.criteo_input_fn(files, bs, num_epochs=-1, need_shuffle=True, layout=layout, num_parallel=threads)
        est.train(fn, steps=304)                        # variables, warm-up, graph captures
        torch.cuda.synchronize()
        t0 = time.perf_counter()                        # the pipeline's own start-up (reader threads, the 1 000-batch shuffle buffer)
        it = iter(fn())
        next(it)
        t_start = time.perf_counter() - t0
        it.close()
        steps = 12 * ((n // bs) // 8 * 8)               # (~0.8 s per repeat: the 23 ms of pipeline start-up per train() call amortised)
        dts = []
        for _ in range(3):
            t0 = time.perf_counter()
            est.train(fn, steps=steps)
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        dt = sorted(dts)[1]
        t0 = time.perf_counter()                        # the input pipeline alone: what the host side can deliver
        it = iter(fn())
        for _ in range(steps):
            next(it)
        dt_in = time.perf_counter() - t0
        it.close()
    ms = dt / steps * 1e3
    return {"workload": "deepfm.py Criteo-39 d=16 DNN 100-100 bs=256 END TO END from TFRecord shards (Estimator.train: host parse + "
                        "hash + H2D inside the timed region), adam_mode=%s" % a.adam_mode,
            "dtype": "f32", "data": "synthetic TFRecord shards (%d records, 4 files, written before the clock starts)" % n,
            "ms_per_step": round(ms, 5), "examples_per_sec": round(bs * steps / dt, 1), "steps": steps,
            "timed_repeats_ms_per_step": [round(x / steps * 1e3, 5) for x in dts],
            "ms_per_step_without_pipeline_startup": round((dt - t_start) / steps * 1e3, 5),
            "pipeline_startup_ms": round(t_start * 1e3, 1),
            "input_pipeline_alone_examples_per_sec": round(bs * steps / dt_in, 1), "reader_threads": threads,
            "host_cores": os.cpu_count(), "resident_batch_ms_per_step": round(resident_ms, 5),
            "ratio_to_resident_batches": round(resident_ms / ms, 4)}


def dp_exchange_info(store, B):
    """What one rank contributes to the step's collectives (data parallel / emulated), bytes per step: the gradient collective
    [dense | the rank's block of the sparse exchange] + the ids-phase collective (the packed unique-row lists of the unique-list
    exchange, or the batch ids of the per-example one), and which exchange runs (recsys_amd/dist.py)."""
    d = getattr(store, "dp", None)
    if d is None or not hasattr(d, "_send"):
        return {}
    ux = bool(getattr(store, "dp_unique", False))
    ids_bytes = 0
    for ar in store.embeddings.values():
        if getattr(ar, "ux", None) is not None and getattr(ar, "_sort_owner", None) is None:
            ids_bytes += 4 * ar.ux.KS
        elif not ux and hasattr(ar, "F") and getattr(ar, "_sort_owner", None) is None:
            ids_bytes += 4 * B * ar.F
    return {"dp_exchange": "unique_rows" if ux else "examples",
            "dp_send_bytes_per_rank_per_step": int(d.send_bytes()) + ids_bytes,
            "dp_gradient_bytes": int(d.send_bytes()), "dp_ids_phase_bytes": ids_bytes}


def launches_per_step(est, feats, wk):
    """librsx kernel launches per training step (rsx_dbg_launch_count around one EAGER optimizer window of wk steps, after the
    timed region): at batch 256 the step is the sum of its dependent launches, so this is the number the latency budget is
    counted in.  Launches of the window's first step that serve the whole window (multi-sort, sweep) are shared over wk."""
    from recsys_amd import _lib
    L = _lib.lib()
    k = max(1, int(wk))
    batches = [feats[i % len(feats)].views() for i in range(k)]
    torch.cuda.synchronize()
    c0 = int(L.rsx_dbg_launch_count())
    if k > 1:
        est._train_window(batches)
    else:
        est._train_eager(*batches[0])
    torch.cuda.synchronize()
    return round((int(L.rsx_dbg_launch_count()) - c0) / k, 3)


# The launch that takes the largest share of each config's step, with the work ONE launch does (SURVEY 8(d) figures) and the
# peak that bounds it; its duration comes from the committed rocprofv3 table of that config (profiles/r0N_z_<model>_kernel_stats.txt,
# newest round present) -- bench.py recomputes the fraction from those two, it does not measure the kernel itself.
DOMINANT_WORK = {
    "dcn": ("tower_bwd_big_k", 4 * 4096 * 624 * 100, 157.3e12, "flop", "fp32 MFMA: d(input) + dW of the 624-wide layer at batch 4096"),
    "xdeepfm": ("cin_bwd_dw_k", 2 * 256 * 16 * 39 * 128 * 128, 157.3e12, "flop", "fp32 MFMA: dW of the [39*128, 128] CIN layer"),
    "xdeepfm_bf16": ("cin_bwd_dw_bf16_k", 2 * 256 * 16 * (39 * 39 * 128 + 39 * 128 * 128), 2.5e15, "flop", "bf16 MFMA: dW of both CIN layers (one launch)"),
    # (split operands: the work is the 16-bit MFMA flops the launch ISSUES -- 6 products of planes per algorithmic product)
    "xdeepfm_x3": ("cin_split_dw_k", 6 * 2 * 256 * 16 * (39 * 39 * 128 + 39 * 128 * 128), 2.5e15, "flop (issued)",
                   "bf16 MFMA, 3 planes per operand: dW of both CIN layers (one launch), 6 MFMAs per k-step"),
    "xdeepfm_x4": ("cin_split_dw_k", 6 * 2 * 256 * 16 * (39 * 39 * 128 + 39 * 128 * 128), 2.5e15, "flop (issued)",
                   "bf16 MFMA, 3 planes per operand: dW of both CIN layers (one launch), 6 MFMAs per k-step (mode 4 keeps the weight "
                   "gradients on three bf16 planes)"),
    "din": ("din_attn_bwd_k", 2 * 2 * 51200 * (128 * 80 + 80 * 40 + 40), 157.3e12, "flop", "fp32 MFMA: attention MLP backward over ~51 200 valid positions (half of 102 400)"),
}


def dominant_kernel_fraction(key):
    import glob
    import re
    if key not in DOMINANT_WORK:
        return None
    kern, work, peak, unit, what = DOMINANT_WORK[key]
    model = key.replace("_bf16", "_bf16")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_z_%s_kernel_stats.txt" % ("xdeepfm_f32" if key == "xdeepfm" else model))))
    if not files:
        return None
    for line in open(files[-1]):
        if kern in line[:72]:
            f = line[72:].split()        # (scripts/rocpd_summary.py: the name in 72 columns, then calls / avg / min / max ...)
            if len(f) >= 4 and f[0].isdigit():
                avg_us = float(f[1])
                out = {"kernel": kern, "what": what, "work_per_launch": work, "unit": unit, "avg_us": avg_us,
                       "source": os.path.basename(files[-1]), "peak": peak,
                       "frac": round(work / (avg_us * 1e-6) / peak, 4)}
                if "issued" in unit:     # six plane products per algorithmic product: the USEFUL fraction beside the issued one
                    out["frac_is"] = "ISSUED 16-bit MFMA flops / launch time / dense bf16 peak"
                    out["frac_useful"] = round(work / 6 / (avg_us * 1e-6) / peak, 4)
                    out["useful_vs_fp32_mfma_peak"] = round(work / 6 / (avg_us * 1e-6) / 157.3e12, 4)
                return out
    return None


def headline_step_kernels():
    """The OTHER launches of the headline (deepfm bs 256) step next to the sweep, each against the roof that bounds it, recomputed
    from the committed evidence: avg duration from the latest profiles/r0N_z_deepfm_kernel_stats.txt, HBM-side traffic from the
    latest profiles/r0N_*_pmc_{FETCH,WRITE}_SIZE_deepfm.txt (separate --pmc passes; FETCH_SIZE / WRITE_SIZE in KB per launch, at
    face value: these launches read 64-byte rows, not the wide streams the guide's x2 correction is for).  VERDICT r4 item 7."""
    import glob
    import re
    ks = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_z_deepfm_kernel_stats.txt")))
    if not ks:
        return None

    def table(path):
        out = {}
        for line in open(path):
            f = line[72:].split()            # (scripts/rocpd_summary.py: the name in 72 columns, then calls / avg / min / max ...)
            if len(f) >= 2 and f[0].isdigit() and int(f[0]) > 50:
                out.setdefault(line[:72].strip(), float(f[1]))
        return out

    def pmc(name):
        fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_%s_deepfm.txt" % name)))
        out = {}
        if fs:
            for line in open(fs[-1]):
                m = re.match(r"^(.*?)\s+%s\s+calls=\s*(\d+)\s+mean=([0-9.]+)" % name, line)
                if m and int(m.group(2)) > 50:
                    out.setdefault(m.group(1).strip(), float(m.group(3)))
        return out, (os.path.basename(fs[-1]) if fs else None)

    avg = table(ks[-1])
    fetch, fsrc = pmc("FETCH_SIZE")
    write, _ = pmc("WRITE_SIZE")
    B, K0, N = 256, 624, 100
    spec = [("tower_bwd_k", "fp32 MFMA 16x16x4: d(input) + dW tiles of one tower layer (two launches per step: 100x100 and 624x100)",
             (4 * B * N * N + 4 * B * K0 * N) / 2, 157.3e12, "flop",
             ((B * N + N * N + B * N) * 4 + (B * K0 + K0 * N + B * N) * 4) / 2),
            ("segsum_adam_k", "scatter + touched-row Adam + dense Adam + lazy window pass (latency-bound; bytes against HBM)",
             None, 8e12, "B", 4.4e6)]
    rows = []
    for kern, what, flop, peak, unit, alg_read in spec:
        k = next((n for n in avg if n.startswith("void " + kern) or n.startswith(kern)), None)
        if k is None:
            continue
        us = avg[k]
        e = {"kernel": kern, "what": what, "avg_us": us, "source": os.path.basename(ks[-1])}
        if flop is not None:
            e.update(work_per_launch=int(flop), unit="flop", peak=peak, frac=round(flop / (us * 1e-6) / peak, 4))
        kf = next((n for n in fetch if kern in n), None)
        if kf is not None:
            fb, wb = fetch[kf] * 1024, write.get(kf, 0.0) * 1024
            e.update(pmc_fetch_bytes=int(fb), pmc_write_bytes=int(wb), alg_bytes=int(alg_read), pmc_source=fsrc,
                     fetch_over_alg=round(fb / alg_read, 2) if flop is not None else round((fb + wb) / alg_read, 2))
            if flop is None:
                e.update(unit="B", peak=peak, frac=round((fb + wb) / (us * 1e-6) / peak, 4))
        rows.append(e)
    return rows


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    from recsys_amd import build as _build
    from recsys_amd import dist
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher of N ranks (one per GPU; the driver's own form,
        # `python -m torch.distributed.run ... bench.py --gpus N`, arrives here with WORLD_SIZE set and skips this)
        _build.build(verbose=False)
        raise SystemExit(dist.spawn_local_ranks(a.gpus, sys.argv[1:], script=os.path.abspath(__file__)))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("bench.py --gpus %d inside a job of WORLD_SIZE=%d: launch N ranks for --gpus N" % (a.gpus, world))
    dp = None
    if world > 1 or os.environ.get("RSX_FORCE_DIST") == "1":
        dist.init_process_group()            # nccl (= RCCL); RSX_DIST_BACKEND=gloo lets several ranks share ONE GPU (smoke runs)
        dp = dist.DataParallel()
    if rank == 0:
        _build.build(verbose=False)          # no-op when the in-tree librsx.so is current
    if dp is not None:
        dp.barrier()                         # the other ranks load the library only after rank 0's build check
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    emu = None
    if a.emulate_world > 1 and dp is None:
        from tests.dp_harness import EmulatedDataParallel          # (profiling harness, not product)
        emu = EmulatedDataParallel(a.emulate_world)

    dp_captured = dist.dp_capture(dp or emu)
    t = time_config(a, a.model, a.batch_size, cin_mode(a.cin_bf16, a.cin_split), dp, emu, rank, dev, a.steps, a.warmup, a.repeats)
    est, B, dt, dts, final_loss, host, layout = t["est"], t["B"], t["dt"], t["dts"], t["final_loss"], t["host"], t["layout"]
    cin_run = t["cin"]

    # ---- roofline leg: the dominant kernel, HIP events on the launch stream (torch's current stream) -----
    store = est.store
    segs = store.adam_segments()
    n_dense = store.dense.n
    # algorithmic bytes of the TF-faithful sweep: 24 B per table / first-order element (var, m, v read + written),
    # 32 B per dense element (+ gradient read and zeroed)
    n_sparse = sum(int(sg["n"]) * int(sg.get("d", 1) or 1) for sg in segs if sg["kind"] in (1, 2))
    alg_bytes = 24 * n_sparse + 32 * n_dense if a.adam_mode == "tf1_dense" else None
    # 20 launches captured back to back in a HIP graph and bracketed by ONE event pair per replay, so the event /
    # launch gap (~7 us around a lone launch) does not inflate the per-launch duration; the step's own state (slot
    # map + sparse grads of the last batch) is live, exactly as in the timed region.
    per, reps = 20, 10
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(per):
            store.opt.step(segs)
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    adam_ms = 0.0
    for r in range(reps):
        e0.record()
        gr.replay()
        e1.record()
        e1.synchronize()
        adam_ms += e0.elapsed_time(e1)
    adam_ms /= reps * per
    roof = None
    wk = est._window_len() if (not a.no_graph and not a.no_overlap) else 1      # (data-parallel runs use windows too)

    def pmc_traffic(kernel):
        """PMC-derived HBM bytes per launch: collected offline (scripts/pmc.sh, separate --pmc passes) and committed."""
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_%s.json" % kernel)))
            if pm.get("model") == a.model and a.adam_mode == "tf1_dense":
                return int((2 * pm["FETCH_SIZE_kb_per_launch"] + pm["WRITE_SIZE_kb_per_launch"]) * 1024)
        except Exception:
            pass
        return None

    src = ("offline: separate rocprofv3 --pmc passes of this command (scripts/pmc.sh), 2 x FETCH_SIZE + WRITE_SIZE, "
           "committed as profiles/pmc_%s.json")
    if alg_bytes is not None:
        ach = alg_bytes / (adam_ms * 1e-3) / 1e9
        step_ach = alg_bytes / (dt / a.steps) / 1e9
        traffic = pmc_traffic("adam_multi_k")
        single = {"kernel": "adam_multi_k", "achieved": round(ach, 1), "frac": round(ach / 8000.0, 4),
                  "cache_assisted": True,
                  "cache_assisted_note": "20 back-to-back replays over 345 MB of state with the 256 MiB Infinity Cache underneath: "
                                         "the figure can exceed the guide's achievable HBM rate (6.3 TB/s) and is NOT an HBM number",
                  "traffic": traffic,
                  "traffic_source": src % "adam_multi_k" if traffic else None, "alg_bytes_per_launch": alg_bytes,
                  "launch_ms": round(adam_ms, 5)}
        from recsys_amd.ops import EmbeddingArena
        arenas = [x for x in store.embeddings.values() if isinstance(x, EmbeddingArena)]
        if wk > 1 and arenas:
            # The timed step runs ONE sweep over the untouched rows per optimizer window of wk steps (include/rsx.h
            # rsx_adam_window): the launch that moves the bytes.  Timed the same way, with the last window's slot maps live.
            cold = []
            for ar in arenas:
                ar.select(0)
                cold += ar.adam_split_segments(window_k=wk)[0][::-1]
            sl = store.opt.cold_slices(cold, [1.0])[0]
            gw = torch.cuda.CUDAGraph()
            store.opt.run_slice(sl)
            torch.cuda.synchronize()
            with torch.cuda.graph(gw):
                for _ in range(per):
                    store.opt.run_slice(sl)
            gw.replay()
            torch.cuda.synchronize()
            win_ms = 0.0
            for r in range(reps):
                e0.record()
                gw.replay()
                e1.record()
                e1.synchronize()
                win_ms += e0.elapsed_time(e1)
            win_ms /= reps * per
            # bytes of ONE pass: 24 B per table / first-order element + the wk slot maps (4 B per row each)
            rows = sum(int(ar.R) for ar in arenas if getattr(ar, "_sort_owner", None) is None)
            pass_bytes = 24 * n_sparse + 4 * wk * rows
            wach = pass_bytes / (win_ms * 1e-3) / 1e9
            traffic = pmc_traffic("adam_window_k")
            roof = {"bound": "hbm", "kernel": "adam_window_k<%d> (ONE untouched-row sweep per optimizer window)" % (wk - 1), "window_steps": wk,
                    "achieved": round(wach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(wach / 8000.0, 4),
                    "achievable_peak": 6300.0, "frac_of_achievable": round(wach / 6300.0, 4),
                    "traffic": traffic, "traffic_source": src % "adam_window_k" if traffic else None,
                    "alg_bytes_per_launch": pass_bytes, "launch_ms": round(win_ms, 5),
                    "note": "achieved = the bytes this launch has to move (one pass over the optimizer state) / its duration; "
                            "it applies wk TF-1 updates per element, i.e. SURVEY 8(d)'s 345 MB/step figure x wk steps = "
                            "tf1_equivalent_bytes, moved in one pass",
                    "tf1_equivalent_bytes": wk * alg_bytes,
                    "tf1_equivalent_GBps": round(wk * alg_bytes / (win_ms * 1e-3) / 1e9, 1),
                    "single_step_sweep": single,
                    # the whole step against the bytes a step-by-step TF-1 run has to move (345 MB each)
                    "tf1_equivalent_step_GBps": round(step_ach, 1), "step_floor_ms": round(pass_bytes / wk / 8e12 * 1e3, 5)}
        else:
            # step-level figure beside the stand-alone kernel: in the timed step the sweep does not run as adam_multi_k
            # but rides, slice by slice, in the tower / head / scatter launches (same per-workgroup code); the bytes it has
            # to move per step are the same, so `tf1_equivalent_step_GBps` = those bytes / the measured step time.
            roof = dict(single, bound="hbm", peak=8000.0, unit="GB/s", tf1_equivalent_step_GBps=round(step_ach, 1),
                        step_floor_ms=round(alg_bytes / 8e12 * 1e3, 5))

    if roof is not None:
        # step-level fractions, named for what they divide (see the module docstring)
        roof.update(step_fractions(est, a.model, B, dt / a.steps * 1e3, wk, cin_run))
        roof.setdefault("achievable_peak", 6300.0)
        roof.setdefault("frac_of_achievable", round(roof["achieved"] / 6300.0, 4))
        if a.model == "deepfm" and B == 256:
            try:
                roof["other_step_kernels"] = headline_step_kernels()
            except Exception as ex:
                roof["other_step_kernels_error"] = repr(ex)[:200]
    try:
        n_launch = launches_per_step(est, t["feats"], wk)
    except Exception as ex:
        n_launch = None
    world_seen = torch.distributed.get_world_size() if dp is not None else 1
    backend = torch.distributed.get_backend() if dp is not None else None
    rank_dts = t["rank_dts"]
    if dp is not None:
        dp.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    N = max(world, 1) if dp is not None else 1
    out = {"metric": "examples/sec", "value": round(N * B * a.steps / dt, 1), "unit": "examples/sec", "n_gpus": a.gpus,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 5), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if a.model != "xdeepfm" else CIN_DTYPE[cin_run], "data": "synthetic" if not a.host_input else "synthetic, host-resident batches (PCIe inside the timed region)",
           "world_size_seen": world_seen,
           **({"rank_ms_per_step": [round(x / a.steps * 1e3, 5) for x in rank_dts], "backend": backend,
               "ranks_per_gpu": -(-world_seen // max(torch.cuda.device_count(), 1))} if dp is not None else {}),
           "config": {"workload": "%s.py %s bs=%d/replica, full train step "
                                  "(fwd+bwd+TF1 Adam), adam_mode=%s, hip_graph=%s, %s"
                                  % (a.model, WORKLOADS[a.model], B,
                                     a.adam_mode, not a.no_graph,
                                     ("steps_per_graph=%d" % a.steps_per_graph) if (dp is None and emu is None) else
                                     ("one graph-segment chain per optimizer window (one ids all-gather per window, one gradient "
                                      "all-gather per step), RCCL collectives %s" %
                                      ("captured into the graphs (steps_per_graph=%d)" % a.steps_per_graph if dp_captured else "eager between graph segments"))),
                      "global_batch": N * B, "parallelism": ("dp%d" % N) if emu is None else "EMULATED per-rank compute of dp%d (not a throughput claim)" % emu.world, "final_loss": round(final_loss, 5),
                      "adam_window": wk, "launches_per_step": n_launch,
                      **dp_exchange_info(store, B),
                      "timed_repeats_ms_per_step": [round(x / a.steps * 1e3, 5) for x in dts], "reported": "median repeat"},
           "roofline": roof}
    if N == 1 and emu is None and not a.no_configs and a.model == "deepfm" and not a.host_input and a.adam_mode == "tf1_dense":
        del t, est, store
        out["configs"] = other_configs(a, rank, dev)
        if not a.no_e2e:
            try:
                out["configs"].append(e2e_config(a, dev, out["ms_per_step"]))
            except Exception as e:                               # (must not take the headline line with it)
                out["configs"].append({"workload": "deepfm.py end to end from TFRecord shards", "error": repr(e)[:300]})
    if N == 1 and not a.no_cpu_baseline and a.model == "deepfm":
        out["cpu_baseline"] = cpu_baseline(host, layout, a.cpu_seconds)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
