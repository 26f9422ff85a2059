"""Argument validation of the device entry points (include/rsx.h): every call below is rejected
BEFORE any HIP call is made, so the checks run without a GPU.  Pointers are fake non-NULL addresses -- they are never read."""
import ctypes as C

import pytest

EINVAL, EUNSUPPORTED, OK = -1, -3, 0
P = C.c_void_p(0x1000)          # "some non-NULL pointer"


@pytest.fixture(scope="module")
def L():
    from recsys_amd import _lib, build
    build.build(verbose=False)
    return _lib.lib()


def test_gather_and_sort_reject_bad_shapes(L):
    assert L.rsx_gather_fm_fwd(None, None, P, P, P, None, None, None, 0, 8, 4, 16, None) == EINVAL          # no tables
    assert L.rsx_gather_fm_fwd(P, None, P, P, P, None, None, None, 0, 8, 4, 12, None) in (EINVAL, EUNSUPPORTED)   # D = 12
    assert L.rsx_field_sort(P, P, P, P, P, P, P, None, 100, 8, 4, 4, None) == EINVAL                      # stride < B
    assert L.rsx_field_sort(P, P, P, P, P, P, P, None, 100, 1 << 20, 4, 1 << 20, None) == EUNSUPPORTED    # one workgroup's LDS
    assert L.rsx_field_sort_large(P, P, P, P, P, P, P, None, None, 100, 8, 4, 8, None) == EINVAL          # no workspace
    assert L.rsx_field_sort_large(P, P, P, P, P, P, P, None, P, 1 << 19, 8, 4, 8, None) == EUNSUPPORTED   # > 2^18 rows per field
    assert L.rsx_field_sort_large_workspace_ints(100, 3, 128) > 3 * 3 * 128
    # xDeepFM's fused input side: both table sets, the first-order vector and the numeric part are all required
    assert L.rsx_gather_two_fwd(P, P, None, P, P, P, P, P, P, P, 0, 8, 4, 16, 13, None) == EINVAL
    assert L.rsx_gather_two_fwd(P, P, P, P, P, P, P, P, P, P, 0, 8, 4, 16, 65, None) == EINVAL      # > 64 numeric features
    assert L.rsx_gather_two_fwd(P, P, P, P, P, P, P, P, P, P, 0, 8, 4, 12, 13, None) == EINVAL      # D = 12
    assert L.rsx_gather_two_fwd(None, None, None, None, None, None, None, None, None, None, 0, 0, 4, 16, 13, None) == OK   # empty batch


def test_cin_entry_points_reject_bad_arguments(L):
    sizes = (C.c_int32 * 9)(*([16] * 9))
    outs = (C.c_void_p * 9)(*([0x1000] * 9))
    assert L.rsx_cin_out_fwd(outs, sizes, 9, P, P, P, 4, 16, None) == EUNSUPPORTED                        # > 8 layers
    assert L.rsx_cin_out_fwd(outs, sizes, 2, P, P, P, 4, 8, None) == EUNSUPPORTED                         # D != 16
    assert L.rsx_cin_out_fwd(None, sizes, 2, P, P, P, 4, 16, None) == EINVAL
    assert L.rsx_cin_out_fwd(outs, sizes, 2, None, P, P, 4, 16, None) == EINVAL
    assert L.rsx_cin_out_fwd(outs, sizes, 2, P, P, P, 0, 16, None) == OK                                  # empty batch
    assert L.rsx_cin_out_bwd(outs, sizes, 2, P, P, None, P, P, 4, 16, None) == EINVAL                     # no gs
    assert L.rsx_cin_out_bwd_lin(outs, sizes, 2, P, P, P, P, P, None, P, P, 13, 4, 16, None) == EINVAL    # dwnum without logx
    assert L.rsx_cin_out_bwd_lin(outs, sizes, 2, P, P, P, P, P, P, P, P, 0, 4, 16, None) == EINVAL        # dwnum with nnum 0
    assert L.rsx_cin_layer_fwd(P, P, P, P, P, 4, 39, 200, 16, 16, None, None) == EUNSUPPORTED             # H > 128
    # backward: neither dout nor the direct-connect pair
    assert L.rsx_cin_layer_bwd(P, P, P, P, None, None, None, P, 0, C.c_void_p(0x2000), 0, P, P, P, 4, 39, 39, 16, 16, None, None) == EINVAL
    # gs without wout
    assert L.rsx_cin_layer_bwd(P, P, P, P, None, P, None, P, 0, C.c_void_p(0x2000), 0, P, P, P, 4, 39, 39, 16, 16, None, None) == EINVAL
    # one gradient buffer for both roles of X0 is legal only for the first layer (Xk == X0) and accumulating
    assert L.rsx_cin_layer_bwd(P, C.c_void_p(0x3000), P, P, P, None, None, P, 0, P, 1, P, P, P, 4, 39, 39, 16, 16, None, None) == EINVAL
    assert L.rsx_cin_layer_bwd(P, P, P, P, P, None, None, P, 0, P, 0, P, P, P, 4, 39, 39, 16, 16, None, None) == EINVAL
    assert L.rsx_cin_bwd_workspace_floats(4, 39, 128, 128) == 4 * 128 * 16 + 8 * 4 * 39 * 16


def test_tower_batch_norm_free_mode_is_all_or_nothing(L):
    u32 = C.c_uint32
    # forward: previous layer present (fstat_prev) with gamma but no beta / bn_prev_out
    assert L.rsx_tower_fwd_layer(P, P, P, P, P, P, P, None, None, None, P, u32(1), 1, 0.0, 8, 16, 16, None, None, None, 0, None) == EINVAL
    # forward: gamma NULL but beta given
    assert L.rsx_tower_fwd_layer(P, P, P, P, P, P, None, P, None, None, P, u32(1), 1, 0.0, 8, 16, 16, None, None, None, 0, None) == EINVAL
    assert L.rsx_tower_fwd_layer(P, P, P, P, P, None, None, None, None, None, P, u32(1), 0, 0.0, 8, 18, 16, None, None, None, 0, None) == EUNSUPPORTED  # K % 4
    assert L.rsx_tower_fwd_layer(P, P, P, P, P, None, None, None, None, None, P, u32(1), 0, 1.0, 8, 16, 16, None, None, None, 0, None) == EINVAL       # rate
    # head: gamma NULL with beta non-NULL
    args = [P, P, None, P, None, P, P, P] + [None] * 5 + [P, P, P, P, P, P, None, None, P]
    assert L.rsx_tower_head(*args, u32(1), 0, 0.0, 1.0, 0, 0, 8, 16, None, None) == EINVAL
    args[2], args[3] = None, None                                                                        # no batch-norm: fine up to ...
    assert L.rsx_tower_head(*args, u32(1), 0, 0.0, 1.0, 0, 0, 8, 300, None, None) == EUNSUPPORTED        # ... the width limit


def test_optimizer_segments_are_validated(L):
    from recsys_amd._lib import AdamSeg, RSX_ADAM_DENSE, RSX_ADAM_TABLE_TF1
    seg = (AdamSeg * 1)()
    seg[0].kind, seg[0].n = RSX_ADAM_DENSE, 100
    for f in ("var", "m", "v", "g"):
        setattr(seg[0], f, 0x1000)
    seg[0].B, seg[0].stride = 4, 64                     # replica sum: stride shorter than the arena
    assert L.rsx_adam_tf1_multi(seg, 1, P, 1e-3, 0.9, 0.999, 1e-8, None) == EINVAL
    seg[0].stride = 102                                 # not a multiple of 4 floats
    assert L.rsx_adam_tf1_multi(seg, 1, P, 1e-3, 0.9, 0.999, 1e-8, None) == EINVAL
    seg[0].kind, seg[0].d, seg[0].B, seg[0].stride = RSX_ADAM_TABLE_TF1, 6, 0, 0     # row width not a multiple of 4
    seg[0].slot = 0x1000
    assert L.rsx_adam_tf1_multi(seg, 1, P, 1e-3, 0.9, 0.999, 1e-8, None) == EINVAL
    assert L.rsx_adam_tf1_multi(seg, 13, P, 1e-3, 0.9, 0.999, 1e-8, None) == EINVAL   # > RSX_ADAM_MAX_SEGS
    assert L.rsx_adam_num_blocks(seg, 1) < 0


def test_scatter_second_table_set_is_validated(L):
    from recsys_amd._lib import TableSet
    ts = TableSet(0x1000, 0x1000, 0, 0x1000, None)      # no v
    r = L.rsx_segsum_adam_rows(P, P, P, None, None, None, None, P, None, None, P, P, P, P, 0, 8, 4, 16, 8, None, 0, None, None,
                               None, C.byref(ts), None, P, 1, 1e-3, 0.9, 0.999, 1e-8, None)
    assert r == EINVAL
    # first-order gradient without the first-order vector
    r = L.rsx_segsum_adam_rows(P, P, P, None, None, None, None, P, P, None, P, P, P, P, 0, 8, 4, 16, 8, None, 0, None, None,
                               None, None, None, P, 1, 1e-3, 0.9, 0.999, 1e-8, None)
    assert r == EINVAL


def test_optimizer_window_is_validated(L):
    from recsys_amd._lib import AdamSeg, AdamWindow, SortJob, RSX_ADAM_TABLE_TF1, RSX_ADAM_TABLE_TF1_COLD
    w = AdamWindow()
    w.k, w.cur, w.max_unique = 2, 0, 8
    for i in range(2):
        w.uniq_row[i], w.nuniq[i], w.slot[i] = 0x1000 + 0x100 * i, 0x1000, 0x1000
    call = lambda win: L.rsx_segsum_adam_rows(P, P, P, None, None, None, None, P, None, None, P, P, C.c_void_p(0x1000), P, 0, 8, 4,
                                              16, 8, None, 0, None, None, None, None, C.byref(win), P, 1, 1e-3, 0.9, 0.999,
                                              1e-8, None)
    w.cur = 2
    assert call(w) == EINVAL                                   # position outside the window
    w.cur = 1
    assert call(w) == EINVAL                                   # entry `cur` is not this step's sort workspace
    w.cur, w.k = 0, 9
    assert call(w) == EINVAL                                   # more than RSX_ADAM_WINDOW_MAX steps
    w.k = 2
    w.slot[1] = None
    assert call(w) == EINVAL
    # the extra slot maps are a prefix, and only the COLD kinds take them
    seg = (AdamSeg * 1)()
    seg[0].kind, seg[0].d, seg[0].n = RSX_ADAM_TABLE_TF1_COLD, 16, 64
    seg[0].var = seg[0].m = seg[0].v = seg[0].slot = 0x1000
    seg[0].slot_w[1] = 0x1000
    assert L.rsx_adam_num_blocks(seg, 1) == EINVAL
    seg[0].slot_w[0] = 0x1000
    assert L.rsx_adam_num_blocks(seg, 1) > 0
    seg[0].kind, seg[0].g = RSX_ADAM_TABLE_TF1, 0x1000
    assert L.rsx_adam_num_blocks(seg, 1) == EINVAL
    # multi-sort: 1..8 jobs of one shape, each with a workspace of its own
    jobs = (SortJob * 2)()
    for i in range(2):
        j = jobs[i]
        j.ids = j.row_off = j.perm = j.seg_off = j.uniq_row = j.nuniq = j.slot = 0x1000
        j.max_rows_per_field, j.B, j.F, j.stride = 100, 8, 4, 8
    assert L.rsx_field_sort_multi(jobs, 2, None) == EINVAL     # shared workspace
    assert L.rsx_field_sort_multi(jobs, 9, None) == EINVAL
    jobs[1].slot, jobs[1].perm, jobs[1].B = 0x2000, 0x2000, 4
    assert L.rsx_field_sort_multi(jobs, 2, None) == EINVAL     # different shapes


def test_attention_envelope(L):
    u32 = C.c_uint32
    assert L.rsx_din_attn_fwd(P, P, P, P, P, P, P, P, P, P, P, None, None, P, u32(1), 0, 0.0, None, None, 4, 10, 24, 80, 40, None) == EUNSUPPORTED  # K
    assert L.rsx_din_attn_fwd(P, P, P, P, P, P, P, P, P, P, P, None, None, P, u32(1), 0, 0.0, None, None, 4, 10, 32, 96, 40, None) == EUNSUPPORTED  # N1
    assert L.rsx_din_attn_fwd(P, P, P, P, P, P, P, P, P, P, P, None, None, P, u32(1), 0, 0.0, P, None, 4, 10, 32, 80, 40, None) == EINVAL     # rows without count
    assert L.rsx_din_attn_bwd(P, P, P, P, P, P, P, P, None, P, P, P, None, None, P, u32(1), 0, 0.0, 0, None, None, None, 4, 10, 32, 80, 40, None) == EINVAL
    # a row list needs the ids it was made from (the dq reduction masks by them)
    assert L.rsx_din_attn_bwd(P, P, P, P, P, P, P, P, P, P, P, P, None, None, P, u32(1), 0, 0.0, 0, P, P, None, 4, 10, 32, 80, 40, None) == EINVAL
    assert L.rsx_din_valid_rows(None, 4, 10, P, P, None, None) == EINVAL
    assert L.rsx_din_valid_rows(P, 1 << 20, 1 << 10, P, P, None, None) == EUNSUPPORTED
    assert L.rsx_din_attn_bwd_workspace_floats(4, 10, 32, 80, 40) > 0


def test_copy_and_selftest_entry_points_reject_bad_arguments(L):
    """rsx_copy_bytes (the streaming windows' first graph node) moves whole 16-byte words between 16-byte aligned buffers."""
    buf = (C.c_char * 64)()
    a = C.addressof(buf)
    a16 = (a + 15) & ~15
    assert L.rsx_copy_bytes(None, a16, 16, None) == EINVAL
    assert L.rsx_copy_bytes(a16, None, 16, None) == EINVAL
    assert L.rsx_copy_bytes(a16, a16 + 16, 8, None) == EINVAL          # not a multiple of 16 bytes
    assert L.rsx_copy_bytes(a16 + 4, a16 + 16, 16, None) == EINVAL     # misaligned destination
    assert L.rsx_copy_bytes(a16, a16 + 20, 16, None) == EINVAL         # misaligned source
    assert L.rsx_copy_bytes(a16, a16 + 16, 0, None) == OK              # nothing to do
    assert L.rsx_adam_fast_math_selftest(None, 1, 1, 0, None) == EINVAL


def test_round4_entry_points_reject_bad_arguments(L):
    """The one-launch batch-norm-free tower, the riders and the fused lookup + cross forward (include/rsx.h, round 4)."""
    from recsys_amd import _lib
    w = (C.c_int32 * 3)(100, 52, 20)
    assert L.rsx_mlp_nobn_supported(96, w, 3) == 1
    assert L.rsx_mlp_nobn_supported(96, w, 4) == 0                                   # > 3 hidden layers
    assert L.rsx_mlp_nobn_supported(98, w, 3) == 0                                   # input width not a multiple of 4
    assert L.rsx_mlp_nobn_supported(96, (C.c_int32 * 3)(128, 52, 20), 3) == 0        # a layer wider than 112
    assert L.rsx_mlp_nobn_workspace_floats(1024, 96, w, 3) == 64 * (112 * 112 + 112 * 64 + 64 * 32 + 24)
    assert L.rsx_mlp_nobn_train_step(None, None) == EINVAL
    ms = _lib.MlpStep()
    ms.B, ms.K0, ms.L = 16, 96, 3
    for i, n in enumerate((100, 52, 20)):
        ms.widths[i] = n
    assert L.rsx_mlp_nobn_train_step(C.byref(ms), None) == EINVAL                    # no buffers
    ms.L = 4
    assert L.rsx_mlp_nobn_train_step(C.byref(ms), None) == EINVAL
    ms.L, ms.B = 3, 0
    assert L.rsx_mlp_nobn_train_step(C.byref(ms), None) == OK                        # empty batch
    job = _lib.MlpReduceJob()
    assert L.rsx_mlp_nobn_reduce_job(C.byref(ms), None) == EINVAL
    assert L.rsx_mlp_nobn_reduce_job(C.byref(ms), C.byref(job)) == OK and job.e4_last == 0
    # riders
    assert L.rsx_vec_reduce_run(None, 1, None) == EINVAL
    assert L.rsx_vec_reduce_run(None, 0, None) == OK
    vj = (_lib.VecReduceJob * 2)()
    assert L.rsx_vec_reduce_run(vj, 3, None) == EINVAL                               # > 2 jobs
    assert L.rsx_vec_reduce_run(vj, 2, None) == OK                                   # both empty (n == 0)
    vj[0].n, vj[0].G = 64, 4
    assert L.rsx_vec_reduce_run(vj, 1, None) == EINVAL                               # no partials / output
    cj = _lib.CrossReduceJob()
    assert L.rsx_cross_reduce_run(None, None) == EINVAL
    assert L.rsx_cross_reduce_run(C.byref(cj), None) == OK                           # n == 0: nothing to do
    cj.n = 10
    assert L.rsx_cross_reduce_run(C.byref(cj), None) == EINVAL
    # the fused lookup + cross forward: D = 16 only, wout and cz together
    assert L.rsx_gather_cross_fwd(P, P, P, P, P, P, P, P, P, 8, 39, 8, 3, None) == EUNSUPPORTED
    assert L.rsx_gather_cross_fwd(P, P, P, P, P, P, P, P, P, 8, 39, 16, 9, None) == EUNSUPPORTED       # > 8 layers
    assert L.rsx_gather_cross_fwd(P, P, P, P, P, P, None, P, P, 8, 39, 16, 3, None) == EINVAL         # cz without wout
    assert L.rsx_gather_cross_fwd(None, P, P, P, P, P, P, P, P, 8, 39, 16, 3, None) == EINVAL
    assert L.rsx_gather_cross_fwd(P, P, P, P, P, P, P, P, P, 0, 39, 16, 3, None) == OK                 # empty batch
    # din.py's prepare launches carrying lookups: the job split must lie inside the job list
    gj = (_lib.GatherJob * 2)()
    assert L.rsx_din_prepare2_gather(P, P, P, P, 4, 10, 100, 10, P, 0, P, P, None, P, P, None, None, None, gj, 2, 3, None) == EINVAL
    assert L.rsx_din_prepare2_gather(P, P, P, P, 4, 10, 100, 10, P, 0, P, P, None, P, P, None, None, None, None, 2, 1, None) == EINVAL
    assert L.rsx_din_prepare2_gather(P, P, P, P, 4, 10, 100, 10, P, 0, P, P, None, P, P, None, None, None, gj, 2, 1, None) == EINVAL   # empty jobs: no table


def test_round5_cin_entry_points_reject_bad_arguments(L):
    """csrc/cin_split.hip (split operands on the bf16 matrix cores) and csrc/cin_bf16_wide.hip (dX0 as tile partials)."""
    from recsys_amd import _lib
    one = lambda v: (C.c_void_p * 1)(v)
    i1 = lambda v: (C.c_int32 * 1)(v)
    # sizes
    assert L.rsx_cin_split_weight_elems(39, 128, 128, 3) == 3 * 2 * 39 * 128 * 128
    assert L.rsx_cin_split_weight_elems(39, 39, 128, 1) == 39 * 48 * 128 + 39 * 128 * 64       # H padded to 16 / to 32
    assert L.rsx_cin_split_weight_elems(39, 128, 128, 5) == 0 and L.rsx_cin_split_weight_elems(39, 128, 128, 0) == 0
    assert L.rsx_cin_split_weight_elems(39, 128, 128, 4) == 2 * 2 * 39 * 128 * 128 + 80      # mode 4: two fp16 planes + 40 fp32 scales
    assert L.rsx_cin_split_bwd_workspace_bytes(256, 128, 4) == L.rsx_cin_split_bwd_workspace_bytes(256, 128, 3)   # dW: three bf16 planes
    assert L.rsx_cin_split_bwd_workspace_bytes(256, 128, 3) == 3 * 256 * 128 * 16 * 2 + 256 * 128 * 4
    assert L.rsx_cin_split_bwd_workspace_bytes(7, 20, 2) == 2 * 8 * 32 * 16 * 2 + 7 * 32 * 4   # odd batch: whole example pairs
    assert L.rsx_cin_bf16_dx0_parts_floats(256, 39, 128) == 8 * 256 * 39 * 16
    assert L.rsx_cin_bf16_dx0_parts_floats(256, 39, 39) == 3 * 256 * 39 * 16
    # prep
    assert L.rsx_cin_split_prep(one(0x1000), one(0x2000), i1(128), i1(128), 1, 39, 5, None) == EINVAL         # ns out of range
    assert L.rsx_cin_split_prep(one(0x1000), one(0x2000), i1(129), i1(128), 1, 39, 3, None) == EUNSUPPORTED   # H > 128
    assert L.rsx_cin_split_prep(one(0), one(0x2000), i1(128), i1(128), 1, 39, 3, None) == EINVAL
    assert L.rsx_cin_split_prep(None, one(0x2000), i1(128), i1(128), 1, 39, 3, None) == EINVAL
    assert L.rsx_cin_split_prep(one(0x1000), one(0x2000), i1(128), i1(128), 5, 39, 3, None) == EUNSUPPORTED   # > 4 layers
    # forward
    assert L.rsx_cin_split_fwd(P, P, P, P, P, 4, 41, 128, 128, 16, 3, None) == EUNSUPPORTED                   # F > 40
    assert L.rsx_cin_split_fwd(P, P, P, P, P, 4, 39, 128, 128, 8, 3, None) == EUNSUPPORTED                    # D != 16
    assert L.rsx_cin_split_fwd(P, P, P, P, P, 4, 39, 128, 128, 16, 0, None) == EINVAL                         # ns
    assert L.rsx_cin_split_fwd(P, P, None, P, P, 4, 39, 128, 128, 16, 3, None) == EINVAL
    assert L.rsx_cin_split_fwd(None, None, None, None, None, 0, 39, 128, 128, 16, 3, None) == OK              # empty batch
    # data gradients: dout or the direct-connect pair, never neither
    assert L.rsx_cin_split_bwd_dx(P, P, P, P, None, None, None, P, 0, P, P, 4, 39, 128, 128, 16, 3, None) == EINVAL
    assert L.rsx_cin_split_bwd_dx(P, P, P, P, None, P, None, P, 0, P, P, 4, 39, 128, 128, 16, 3, None) == EINVAL      # gs without wout
    assert L.rsx_cin_split_bwd_dx(P, P, P, P, P, None, None, P, 0, None, P, 4, 39, 128, 128, 16, 3, None) == EINVAL   # no dx0_parts
    assert L.rsx_cin_split_bwd_dx(P, P, P, P, P, None, None, P, 0, P, P, 4, 39, 128, 130, 16, 3, None) == EUNSUPPORTED
    assert L.rsx_cin_split_bwd_dx(P, P, P, P, P, None, None, P, 0, P, P, 0, 39, 128, 128, 16, 3, None) == OK
    assert L.rsx_cin_layer_bwd_dx_bf16_parts(P, P, P, P, None, None, None, P, 0, P, P, 4, 39, 128, 128, 16, None) == EINVAL
    assert L.rsx_cin_layer_bwd_dx_bf16_parts(P, P, P, P, P, None, None, P, 0, P, P, 4, 41, 128, 128, 16, None) == EUNSUPPORTED
    # the reduce of the tile partials
    assert L.rsx_cin_dx0_reduce(one(0x1000), i1(8), 1, None, 0, 4, 39, 16, None) == EINVAL
    assert L.rsx_cin_dx0_reduce(one(0x1000), i1(0), 1, P, 0, 4, 39, 16, None) == EINVAL                       # a job without tiles
    assert L.rsx_cin_dx0_reduce(one(0x1000), i1(8), 5, P, 0, 4, 39, 16, None) == EUNSUPPORTED                 # > 4 jobs
    assert L.rsx_cin_dx0_reduce(one(0x1000), i1(8), 1, P, 0, 0, 39, 16, None) == OK
    # weight gradients
    jobs = (_lib.CinDwJob * 1)(_lib.CinDwJob(0x1000, 0x2000, 0x3000, 0x4000, 128, 128, 0))
    assert L.rsx_cin_split_bwd_dw(P, jobs, 1, 4, 39, 16, 5, None) == EINVAL
    assert L.rsx_cin_split_bwd_dw(P, jobs, 5, 4, 39, 16, 3, None) == EUNSUPPORTED
    assert L.rsx_cin_split_bwd_dw(None, jobs, 1, 4, 39, 16, 3, None) == EINVAL
    assert L.rsx_cin_split_bwd_dw(P, jobs, 1, 0, 39, 16, 3, None) == OK
    bad = (_lib.CinDwJob * 1)(_lib.CinDwJob(0x1000, 0, 0x3000, 0x4000, 128, 128, 0))
    assert L.rsx_cin_split_bwd_dw(P, bad, 1, 4, 39, 16, 3, None) == EINVAL                                    # no workspace
    parts, tl = (C.c_void_p * 1)(0x1000), (C.c_int32 * 1)(8)
    assert L.rsx_cin_split_bwd_dw_dx0(P, jobs, 1, 0, 39, 16, 3, parts, tl, 1, P, 1, None) == OK               # empty batch
    assert L.rsx_cin_split_bwd_dw_dx0(P, jobs, 1, 4, 39, 16, 3, None, tl, 1, P, 1, None) == EINVAL
    assert L.rsx_cin_split_bwd_dw_dx0(P, jobs, 1, 4, 39, 16, 3, parts, tl, 1, None, 1, None) == EINVAL        # no dX0
    assert L.rsx_cin_split_bwd_dw_dx0(P, jobs, 1, 4, 39, 16, 3, parts, tl, 5, P, 1, None) == EUNSUPPORTED     # > 4 layers of partials
    assert L.rsx_cin_split_bwd_dw_dx0(P, jobs, 1, 4, 39, 16, 3, (C.c_void_p * 1)(0), tl, 1, P, 1, None) == EINVAL
    # the bf16 weight-gradient launch checks the workspace convention it is told
    j2 = (_lib.CinDwJob * 1)(_lib.CinDwJob(0x1000, 0x2000, 0x3000, 0x4000, 128, 128, 3))
    assert L.rsx_cin_bwd_dw_bf16(P, j2, 1, 4, 39, 16, None, None) == EINVAL                                   # dc_rows neither 0 nor B


def test_round6_collective_entry_points_reject_bad_arguments(L):
    """include/rsx.h "Collectives of the data-parallel step": argument checks come before any RCCL call; the binding itself
    (dlopen + dlsym of the RCCL copy this process holds) works without a GPU."""
    ECOMM = -5
    v = C.c_int(0)
    rc = L.rsx_comm_available_h(C.byref(v))
    assert rc in (OK, EUNSUPPORTED)                      # EUNSUPPORTED: a host without RCCL -- the rest of the library works
    if rc == OK:
        assert v.value >= 20000                           # an NCCL_VERSION_CODE
        uid = C.create_string_buffer(128)
        assert L.rsx_comm_unique_id_h(uid) == OK and any(uid.raw)
    assert L.rsx_comm_unique_id_h(None) == EINVAL
    h = C.c_void_p()
    assert L.rsx_comm_init_h(None, 0, 1, C.byref(h)) == EINVAL
    assert L.rsx_comm_init_h(P, 0, 1, None) == EINVAL
    assert L.rsx_comm_init_h(P, 2, 2, C.byref(h)) == EINVAL           # rank outside 0 .. world-1
    assert L.rsx_comm_init_h(P, 0, 0, C.byref(h)) == EINVAL
    assert L.rsx_comm_destroy_h(None) == EINVAL
    assert L.rsx_comm_rank_world_h(None, None, None) == EINVAL
    assert L.rsx_all_gather(None, P, P, 16, None) == EINVAL           # no communicator
    assert L.rsx_all_gather(P, None, P, 16, None) == EINVAL
    assert L.rsx_all_reduce_sum_f32(None, P, P, 4, None) == EINVAL
    assert L.rsx_all_reduce_sum_f32(P, P, None, 4, None) == EINVAL
    assert L.rsx_all_reduce_all_gather(None, P, 4, P, P, 16, None) == EINVAL
    assert L.rsx_all_reduce_all_gather(P, P, 4, P, P, 18, None) == EINVAL        # block not a multiple of 4 bytes
    assert L.rsx_strerror(ECOMM).decode().startswith("collective library call failed")
    assert isinstance(L.rsx_comm_last_error_h(), bytes)


def test_round6_cross_rider_envelope(L):
    """rsx_tower_bwd_cross_ride_supported (include/rsx.h rsx_tower_bwd_extra): dcn.py's shapes at batch >= 4096 only."""
    ok = L.rsx_tower_bwd_cross_ride_supported
    assert ok(4096, 100, 100, 624, 100, 624, 3) == 1           # dcn.py bs 4096: 624 -> 100 -> 100, 3 cross layers
    assert ok(8192, 100, 100, 624, 100, 624, 3) == 1
    assert ok(1024, 100, 100, 624, 100, 624, 3) == 0           # below 4 096 a 100-wide layer's backward takes the tile kernels
    assert ok(256, 100, 100, 624, 100, 624, 3) == 0            # small batches: the tile kernels do not know the roles
    assert ok(4096, 100, 100, 624, 100, 624, 2) == 0           # the rider body is instantiated for L = 3
    assert ok(4096, 624, 100, 624, 100, 624, 3) == 0           # a WIDE carrying layer (two d(input) passes) is not
    assert ok(4096, 100, 100, 624, 100, 626, 3) == 0           # dim % 4
    assert ok(0, 100, 100, 624, 100, 624, 3) == 0
