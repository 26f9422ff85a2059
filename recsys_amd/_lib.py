"""ctypes binding of librsx.so (include/rsx.h).  Fails loudly when the library is missing: there is
no CPU fallback anywhere in this package."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RSX_LIB_PATH: an alternative build of the same library (profiling variants, scripts/stamp_probe.py); never a fallback
LIB_PATH = os.environ.get("RSX_LIB_PATH") or os.path.join(_HERE, "librsx.so")

(RSX_ADAM_DENSE, RSX_ADAM_TABLE_TF1, RSX_ADAM_VEC_SLOT, RSX_ADAM_TABLE_ROWS, RSX_ADAM_VEC_ROWS, RSX_ADAM_TABLE_TF1_COLD,
 RSX_ADAM_VEC_COLD, RSX_ADAM_VEC_ROWS_DENSE) = range(8)
RSX_ADAM_MAX_SEGS = 12


class RsxError(RuntimeError):
    pass



# ---- launch-form toggles (round 6: ONE environment variable instead of two dozen) ---------------------------------------------
# Every fused / riding launch form of the steps keeps the form it replaced for A/B runs and for tests/test_gpu_knobs.py, which
# holds each alternative against the default schedule's bits.  The defaults are the measured best on the only hardware
# available; RSX_FORMS="name=value,name=value" overrides them (names below, case-insensitive).
FORMS = {
    "scatter_riders": ("1", "reductions only the optimizer reads ride in the scatter's stage-A launch (0: launches of their own)"),
    "sort_in_gather": ("0", "deepfm.py: the dedup sort rides in the gather launch instead of the first tower-forward launch"),
    "sort_ride_max": ("2048", "largest global batch whose dedup sort rides in another launch"),
    "din_side_sort": ("1", "din.py: the ids-only branch (sort + sweep) on a side stream of the step's graph"),
    "din_side_sort_dp": ("1", "the same under data parallelism"),
    "din_gather_ride": ("1", "din.py: the history gather rides in the prepare launch"),
    "mlp_reduce_ride": ("1", "din.py: the fused MLP's gradient reduce rides in the pooling backward launch"),
    "mlp_fuse": ("1", "a batch-norm-free tower as ONE launch (csrc/mlp_fused.hip) instead of 2L + 2"),
    "fm_fuse": ("1", "fm.py: forward + head in one launch"),
    "fuse_gather": ("1", "the input_layer lookup rides in the first tower-forward launch"),
    "gather_cross": ("1", "dcn.py: lookup + cross forward as one launch"),
    "cross_ride": ("1", "dcn.py: the cross layers' backward rides in the last tower layer's backward launch (0: a launch of its own)"),
    "win_separate_sorts": ("0", "debugging aid: an optimizer window's k sorts as k launches"),
    "cin_dx_fsplit": ("1", "xdeepfm.py: the first CIN layer's data gradients split their fields over two workgroups per tile"),
    "cin_wide": ("1", "xdeepfm.py --cin_bf16: the wide-tile kernels"),
    "cin_gather_ride": ("1", "xdeepfm.py: the lookup rides in the filter-preparation launch"),
    "cin_dx0_ride": ("1", "xdeepfm.py: the dX0 tile reduce rides in the weight-gradient launch"),
    "xdfm_sort_ride": ("1", "xdeepfm.py: the dedup sort rides in a tower launch"),
    "dp_eager_tail": ("1", "segmented data-parallel graphs: the step's train_op as an eager tail"),
    "dp_prefetch": ("0", "segmented data-parallel graphs: the next step's ids all-gather issued early"),
    "window_stage": ("", "'memcpy': staged windows fetched with hipMemcpyAsync instead of the copy kernel"),
    "window_sets": ("2", "captured instances of a streaming window that take turns"),
    "input_thread": ("0", "the input iterator on a thread of its own"),
    "launch_thread": ("0", "streaming windows issued by a thread of their own"),
    "adam_window_large": ("4", "steps per optimizer window above the small-batch threshold"),
}


def form(name):
    """Value (a string) of launch-form toggle `name`: RSX_FORMS's override, else the default."""
    name = name.lower()
    default = FORMS[name][0]
    spec = os.environ.get("RSX_FORMS")
    if spec:
        for item in spec.split(","):
            k, _, v = item.partition("=")
            k = k.strip().lower()
            if k and k not in FORMS:
                raise RsxError("RSX_FORMS: unknown launch form %r (known: %s)" % (k, ", ".join(sorted(FORMS))))
            if k == name:
                return v.strip()
    return default


class AdamSeg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("d", C.c_int32), ("n", C.c_int64),
                ("var", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("g", C.c_void_p),
                ("slot", C.c_void_p), ("uniq_row", C.c_void_p), ("nuniq", C.c_void_p),
                ("B", C.c_int32), ("stride", C.c_int32), ("zero_grad", C.c_int32), ("slot_w", C.c_void_p * 7),
                ("g_replicas", C.c_int32), ("g_replica_stride", C.c_int64)]


class AdamSlice(C.Structure):
    _fields_ = [("segs", C.POINTER(AdamSeg)), ("nseg", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("state", C.c_void_p), ("blk_lo", C.c_uint32),
                ("blk_hi", C.c_uint32), ("window_block_u", C.c_int32), ("alphas_from_state", C.c_int32)]


class ExampleBlocks(C.Structure):
    _fields_ = [("examples", C.c_int32), ("stride_floats", C.c_int64)]


class SegPartials(C.Structure):
    _fields_ = [("segid", C.c_void_p), ("P", C.c_void_p), ("P1", C.c_void_p), ("G", C.c_void_p), ("gw1", C.c_void_p),
                ("row_off", C.c_void_p), ("null_row", C.c_int32), ("skip_mask", C.c_uint64)]

    def __init__(self, segid=None, P=None, P1=None, G=None, gw1=None, row_off=None, null_row=-1, skip_mask=0):
        super().__init__(segid, P, P1, G, gw1, row_off, null_row, skip_mask)


NULL_NONE, NULL_LAST_ROW = -1, -2          # include/rsx.h RSX_NULL_*


class MlpStep(C.Structure):               # include/rsx.h rsx_mlp_step
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p * 3), ("b", C.c_void_p * 3), ("masks", C.c_void_p * 3),
                ("wout", C.c_void_p), ("bout", C.c_void_p), ("s0", C.c_void_p), ("labels", C.c_void_p),
                ("rng_step", C.c_void_p), ("prob", C.c_void_p), ("dX", C.c_void_p), ("gs0", C.c_void_p),
                ("workspace", C.c_void_p), ("dW", C.c_void_p * 3), ("db", C.c_void_p * 3), ("dwout", C.c_void_p),
                ("dbout", C.c_void_p), ("loss", C.c_void_p), ("seed", C.c_uint32), ("dropout_rate", C.c_float),
                ("loss_scale", C.c_float), ("B", C.c_int32), ("K0", C.c_int32), ("L", C.c_int32), ("widths", C.c_int32 * 3),
                ("defer_reduce", C.c_int32), ("reserved", C.c_int32)]


class MlpReduceJob(C.Structure):          # include/rsx.h rsx_mlp_reduce_job
    _fields_ = [("part", C.c_void_p), ("poff", C.c_longlong * 4), ("dW", C.c_void_p * 3), ("db", C.c_void_p * 3),
                ("dwout", C.c_void_p), ("dbout", C.c_void_p), ("loss", C.c_void_p), ("K", C.c_int32 * 3), ("N", C.c_int32 * 3),
                ("e4_end", C.c_uint32 * 3), ("e4_last", C.c_uint32), ("L", C.c_int32), ("nwg", C.c_int32), ("NPo", C.c_int32),
                ("NL", C.c_int32), ("inv_B", C.c_double)]


class DwReduceJob(C.Structure):
    _fields_ = [("partials", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p), ("sb", C.c_int32), ("K", C.c_int32),
                ("N", C.c_int32), ("layout", C.c_int32)]


class CrossReduceJob(C.Structure):        # include/rsx.h rsx_cross_reduce_job
    _fields_ = [("part", C.c_void_p), ("dW", C.c_void_p), ("dB", C.c_void_p), ("dwout", C.c_void_p), ("RT", C.c_int32),
                ("n", C.c_int32), ("L", C.c_int32), ("dim", C.c_int32)]


class TowerBwdExtra(C.Structure):         # include/rsx.h rsx_tower_bwd_extra
    _fields_ = [("accumulate_dx", C.c_int32), ("x0", C.c_void_p), ("cW", C.c_void_p), ("cB", C.c_void_p), ("s", C.c_void_p),
                ("gz", C.c_void_p), ("wout", C.c_void_p), ("dX", C.c_void_p), ("dcW", C.c_void_p), ("dcB", C.c_void_p),
                ("dwout", C.c_void_p), ("workspace", C.c_void_p), ("dim", C.c_int32), ("L", C.c_int32),
                ("reduce_out", C.POINTER(CrossReduceJob))]


class VecReduceJob(C.Structure):          # include/rsx.h rsx_vec_reduce_job
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("G", C.c_int32), ("n", C.c_int32)]


class ScatterRiders(C.Structure):         # include/rsx.h rsx_scatter_riders
    _fields_ = [("dw", DwReduceJob * 4), ("n_dw", C.c_int32), ("n_vec", C.c_int32), ("cross", CrossReduceJob),
                ("vec", VecReduceJob * 2)]


class GatherJob(C.Structure):
    _fields_ = [("table", C.c_void_p), ("ids", C.c_void_p), ("out", C.c_void_p), ("n", C.c_int64), ("K", C.c_int32),
                ("ld_out", C.c_int32), ("row_base", C.c_int32), ("ld_table", C.c_int32)]


class TableSet(C.Structure):
    _fields_ = [("tables", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("dX", C.c_void_p), ("partials", C.c_void_p)]


class CinDwJob(C.Structure):
    _fields_ = [("Xk", C.c_void_p), ("ws", C.c_void_p), ("dW", C.c_void_p), ("dc", C.c_void_p), ("H", C.c_int32), ("N", C.c_int32),
                ("dc_rows", C.c_int32)]


class GatherTwoJob(C.Structure):          # include/rsx.h rsx_gather_two_job
    _fields_ = [("tables1", C.c_void_p), ("w1", C.c_void_p), ("tables2", C.c_void_p), ("row_off", C.c_void_p), ("ids", C.c_void_p),
                ("num_x", C.c_void_p), ("num_w", C.c_void_p), ("E1", C.c_void_p), ("E2", C.c_void_p), ("y1", C.c_void_p),
                ("w1_field_mask", C.c_uint64), ("B", C.c_int32), ("F", C.c_int32), ("D", C.c_int32), ("ND", C.c_int32)]


class UniqPackJob(C.Structure):           # include/rsx.h rsx_uniq_pack_job
    _fields_ = [("uniq_row", C.c_void_p), ("nuniq", C.c_void_p), ("keys", C.c_void_p)]


class UniqMergeJob(C.Structure):          # include/rsx.h rsx_uniq_merge_job
    _fields_ = [("uniq_row", C.c_void_p), ("nuniq", C.c_void_p), ("slot", C.c_void_p), ("src", C.c_void_p)]


UNIQ_MAX_RANKS, UNIQ_MAX_PARTS = 8, 64


class SortJob(C.Structure):
    _fields_ = [("ids", C.c_void_p), ("row_off", C.c_void_p), ("perm", C.c_void_p), ("seg_off", C.c_void_p),
                ("uniq_row", C.c_void_p), ("nuniq", C.c_void_p), ("slot", C.c_void_p), ("segid", C.c_void_p),
                ("max_rows_per_field", C.c_int32), ("B", C.c_int32), ("F", C.c_int32), ("stride", C.c_int32),
                ("skip_mask", C.c_uint64)]


ADAM_WINDOW_MAX = 8
ADAM_STATE_WORDS = 4 + 32 * 32        # include/rsx.h RSX_ADAM_STATE_WORDS


def default_adam_window(capacity, unique_exchange=False):
    """Steps per optimizer window (include/rsx.h rsx_adam_window) for a sort workspace of `capacity` examples: the one
    sweep per window costs 57-79 us for 1-8 steps (DeepFM-size state), every step walks the other steps' unique-row lists
    (their length grows with the batch).  Measured on MI355X: 8 steps up to batch 1024, 4 above (dcn.py at 4096 with the
    packed sweep: 0.229 / 0.231 / 0.239 ms per step with windows of 4 / 6 / 8, scripts/gpu_round2_ax.sh)."""
    # unique-list exchange (the only configuration the 2048 point was measured in -- round 5: deepfm.py as 8 ranks x 256, 0.1032
    # ms per step with windows of 4, 0.0977 with 8): 8 steps up to a global batch of 2048; everything else keeps the
    # single-replica measurements' 1024
    if capacity <= (2048 if unique_exchange else 1024):
        return ADAM_WINDOW_MAX
    return max(1, min(ADAM_WINDOW_MAX, int(form("adam_window_large"))))     # (A/B knob for the large-batch choice)


class AdamWindow(C.Structure):
    _fields_ = [("k", C.c_int32), ("cur", C.c_int32), ("max_unique", C.c_int32), ("uniq_row", C.c_void_p * ADAM_WINDOW_MAX),
                ("nuniq", C.c_void_p * ADAM_WINDOW_MAX), ("slot", C.c_void_p * ADAM_WINDOW_MAX)]


_P, _I, _U64, _F = C.c_void_p, C.c_int, C.c_uint64, C.c_float
_SIGS = {
    "rsx_version": (C.c_int, []),
    "rsx_dbg_launch_count": (C.c_uint64, []),
    "rsx_strerror": (C.c_char_p, [_I]),
    "rsx_gather_fm_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _U64, _I, _I, _I, _P]),
    "rsx_gather_fm_fwd_sort": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _U64, _I, _I, _I, _P, _P]),
    "rsx_gather_two_fwd": (_I, [_P] * 10 + [_U64, _I, _I, _I, _I, _P]),
    "rsx_gather_fm_head": (_I, [_P] * 5 + [_U64] + [_P] * 8 + [_I, _I, _I, _I, _I, _F, _I, _I, _I, _P]),
    "rsx_field_sort": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "rsx_field_sort_large": (_I, [_P] * 9 + [_I, _I, _I, _I, _P]),
    "rsx_field_sort_large_t": (_I, [_P] * 9 + [_I, _I, _I, _I, _P]),
    "rsx_field_sort_large_workspace_ints": (C.c_size_t, [_I, _I, _I]),
    "rsx_segsum_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U64, _I, _I, _I, _I, _P, _P, _P]),
    "rsx_segsum_partials": (_I, [_P] * 9 + [_U64, _I, _I, _I, _I, _I, _P, _P]),
    "rsx_segsum_partials_ride": (_I, [_P] * 9 + [_U64, _I, _I, _I, _I, _I, _P, C.POINTER(ScatterRiders), _P]),
    "rsx_adam_state_init_h": (_I, [_P, _F, _F]),
    "rsx_adam_tf1_multi": (_I, [C.POINTER(AdamSeg), _I, _P, _F, _F, _F, _F, _P]),
    "rsx_tower_fwd_layer": (_I, [_P] * 11 + [C.c_uint32, _I, _F, _I, _I, _I, _P, _P, _P, _I, _P]),
    "rsx_gather_tower_fwd0": (_I, [_P] * 8 + [_U64, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P]),
    "rsx_gather_tower_fwd0_supported": (_I, [_I, _I, _I]),
    "rsx_tower_head": (_I, [_P] * 22 + [C.c_uint32, _I, _F, _F, _I, _I, _I, _I, _P, _P]),
    "rsx_fm_head": (_I, [_P] * 13 + [_F, _I, _P, _P]),
    "rsx_fm_head_terms": (_I, [_P] * 14 + [_I, _I, _I, _I, _I, _F, _I, _P, _P]),
    "rsx_tower_bwd_layer": (_I, [_P] * 26 + [C.c_uint32, _I, _F, _I, _I, _I, _P, _P, _P, _P]),
    "rsx_tower_bwd_layer_defer": (_I, [_P] * 26 + [C.c_uint32, _I, _F, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "rsx_tower_bwd_cross_ride_supported": (_I, [_I] * 7),
    "rsx_tower_reduce_dw_jobs": (_I, [_P, _I, _P]),
    "rsx_tower_bwd_workspace_floats": (C.c_size_t, [_I, _I, _I]),
    "rsx_segsum_adam_rows": (_I, [_P] * 14 + [_U64, _I, _I, _I, _I, C.POINTER(AdamSeg), _I, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _P]),
    "rsx_segsum_adam_rows2": (_I, [_P] * 14 + [_U64, _I, _I, _I, _I, C.POINTER(AdamSeg), _I, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F,
                                   _I, _I, _P]),
    "rsx_field_sort_multi": (_I, [C.POINTER(SortJob), _I, _P]),
    "rsx_segsum_bwd_packed": (_I, [_P] * 11 + [_U64, _I, _I, _I, _I, _I, _P, _P, _P]),
    "rsx_uniq_pack": (_I, [C.POINTER(UniqPackJob), _I, _P, _P, _I, _I, _I, _I, _P]),
    "rsx_uniq_merge": (_I, [_P, C.c_longlong, _I, C.POINTER(UniqMergeJob), _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "rsx_merged_adam_rows": (_I, [_P] * 8 + [C.c_longlong, _I, _P, _P, _P, _P, _U64, _I, _I, _I, _I, C.POINTER(AdamSeg), _I,
                                  _P, _P, _P, _P, _I, _F, _F, _F, _F, _I, _I, _P]),
    "rsx_adam_num_blocks": (C.c_int64, [C.POINTER(AdamSeg), _I]),
    "rsx_adam_num_blocks_u": (C.c_int64, [C.POINTER(AdamSeg), _I, _I]),
    "rsx_adam_slice_run": (_I, [_P, _P]),
    "rsx_copy_bytes": (_I, [_P, _P, C.c_size_t, _P]),
    "rsx_comm_available_h": (_I, [C.POINTER(C.c_int)]),
    "rsx_comm_last_error_h": (C.c_char_p, []),
    "rsx_comm_unique_id_h": (_I, [_P]),
    "rsx_comm_init_h": (_I, [_P, _I, _I, C.POINTER(C.c_void_p)]),
    "rsx_comm_destroy_h": (_I, [_P]),
    "rsx_comm_rank_world_h": (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rsx_all_gather": (_I, [_P, _P, _P, C.c_size_t, _P]),
    "rsx_all_reduce_sum_f32": (_I, [_P, _P, _P, C.c_size_t, _P]),
    "rsx_all_reduce_all_gather": (_I, [_P, _P, C.c_size_t, _P, _P, C.c_size_t, _P]),
    "rsx_adam_fast_math_selftest": (_I, [_P, C.c_uint32, _I, _I, _P]),
    "rsx_cross_fwd": (_I, [_P] * 7 + [_I, _I, _I, _P]),
    "rsx_cross_bwd_workspace_floats": (C.c_size_t, [_I, _I, _I]),
    "rsx_gather_cross_fwd": (_I, [_P] * 9 + [_I, _I, _I, _I, _P]),
    "rsx_cross_bwd": (_I, [_P] * 8 + [_I] + [_P] * 4 + [_I, _I, _I, _P]),
    "rsx_cross_bwd_defer": (_I, [_P] * 8 + [_I] + [_P] * 4 + [_I, _I, _I, C.POINTER(CrossReduceJob), _P]),
    "rsx_cross_reduce_run": (_I, [C.POINTER(CrossReduceJob), _P]),
    "rsx_din_pool_fwd": (_I, [_P] * 4 + [_I, _I, _I, _P]),
    "rsx_din_pool_fwd_ld": (_I, [_P] * 4 + [_I, _I, _I, _I, _P]),
    "rsx_din_pool_bwd_ld": (_I, [_P] * 6 + [_I, _I, _I, _I, _I, _I, _P]),
    "rsx_gather_rows_multi": (_I, [_P, _I, _P]),
    "rsx_din_keys": (_I, [_P] * 4 + [_I, _I, _I, _I, _P, _P]),
    "rsx_din_prepare": (_I, [_P] * 4 + [_I, _I, _I, _I] + [_P] * 8),
    "rsx_din_attn_bwd_nofinish": (_I, [_P] * 13 + [C.c_uint32, _I, C.c_float, _I] + [_P] * 3 + [_I] * 6 + [_P]),
    "rsx_din_attn_finish_pair": (_I, [_P] * 10 + [_I] * 7 + [_P]),
    "rsx_din_attn_finish_pair_defer": (_I, [_P] * 10 + [_I] * 7 + [_P, _P]),
    "rsx_vec_reduce_run": (_I, [_P, _I, _P]),
    "rsx_din_pool_fwd_pair": (_I, [_P] * 8 + [_I, _I, _I, _I, _P]),
    "rsx_din_pool_bwd_pair": (_I, [_P] * 12 + [_I, _I, _I, _I, _I, _I, _P]),
    "rsx_din_pool_bwd_pair_ride": (_I, [_P] * 12 + [_I, _I, _I, _I, _I, _I, C.POINTER(MlpReduceJob), _P]),
    "rsx_din_prepare2": (_I, [_P] * 4 + [_I, _I, _I, _I] + [_P, _I] + [_P] * 6 + [_P, _P, _P]),
    "rsx_din_prepare2_gather": (_I, [_P] * 4 + [_I, _I, _I, _I] + [_P, _I] + [_P] * 6 + [_P, _P, C.POINTER(GatherJob), _I, _I, _P]),
    "rsx_din_attn_bwd_ld": (_I, [_P] * 15 + [C.c_uint32, _I, _F, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "rsx_din_pool_bwd": (_I, [_P] * 6 + [_I, _I, _I, _I, _P]),
    "rsx_segsum_rows": (_I, [_P] * 6 + [_I, _I, _I, _I, _P, _P]),
    "rsx_din_attn_fwd": (_I, [_P] * 14 + [C.c_uint32, _I, _F, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rsx_din_valid_rows": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "rsx_din_attn_bwd": (_I, [_P] * 15 + [C.c_uint32, _I, _F, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rsx_din_attn_bwd_workspace_floats": (C.c_size_t, [_I, _I, _I, _I, _I]),
    "rsx_sorted_segments": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "rsx_cin_layer_fwd": (_I, [_P] * 5 + [_I] * 5 + [_P, _P]),
    "rsx_cin_layer_bwd": (_I, [_P] * 8 + [_I, _P, _I, _P, _P, _P] + [_I] * 5 + [_P, _P]),
    "rsx_cin_bwd_workspace_floats": (C.c_size_t, [_I, _I, _I, _I]),
    "rsx_cin_bf16_weight_elems": (C.c_size_t, [_I, _I, _I]),
    "rsx_cin_bf16_bwd_workspace_bytes": (C.c_size_t, [_I, _I]),
    "rsx_cin_prep_bf16": (_I, [_P, _P, _I, _I, _I, _P]),
    "rsx_cin_layer_fwd_bf16": (_I, [_P] * 5 + [_I] * 5 + [_P, _P]),
    "rsx_cin_layer_bwd_bf16": (_I, [_P] * 8 + [_I, _P, _I, _P, _P, _P] + [_I] * 5 + [_P, _P]),
    "rsx_cin_layer_bwd_dx_bf16": (_I, [_P] * 8 + [_I, _P, _I, _P] + [_I] * 5 + [_P, _P]),
    "rsx_cin_bwd_dw_bf16": (_I, [_P, C.POINTER(CinDwJob), _I, _I, _I, _I, _P, _P]),
    "rsx_cin_prep_bf16_multi": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "rsx_cin_bf16_dx0_parts_floats": (C.c_size_t, [_I, _I, _I]),
    "rsx_cin_split_weight_elems": (C.c_size_t, [_I, _I, _I, _I]),
    "rsx_cin_split_prep": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "rsx_cin_split_fwd": (_I, [_P] * 5 + [_I] * 6 + [_P]),
    "rsx_cin_split_bwd_workspace_bytes": (C.c_size_t, [_I, _I, _I]),
    "rsx_cin_split_bwd_dx": (_I, [_P] * 8 + [_I, _P, _P] + [_I] * 6 + [_P]),
    "rsx_cin_split_prep_gather": (_I, [_P, _P, _P, _P, _I, _I, _I, C.POINTER(GatherTwoJob), _P]),
    "rsx_cin_split_bwd_dw": (_I, [_P, C.POINTER(CinDwJob), _I, _I, _I, _I, _I, _P]),
    "rsx_cin_split_bwd_dw_dx0": (_I, [_P, C.POINTER(CinDwJob), _I, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P]),
    "rsx_cin_layer_bwd_dx_bf16_parts": (_I, [_P] * 8 + [_I, _P, _P] + [_I] * 5 + [_P]),
    "rsx_cin_dx0_reduce": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "rsx_cin_out_fwd": (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _P]),
    "rsx_cin_out_bwd": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P]),
    "rsx_cin_out_bwd_lin": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rsx_hash_fp64_h": (_I, [_P, _P, C.c_int64, _P]),
    "rsx_fingerprint64_h": (C.c_uint64, [_P, C.c_size_t]),
    "rsx_bucketize_log_h": (_I, [_P, C.c_int64, _P, _I, _F, _P]),
    "rsx_tfrecord_index_h": (C.c_int64, [_P, C.c_size_t, _P, _P, C.c_int64, _I]),
    "rsx_criteo_parse_h": (_I, [_P, _P, _P, C.c_int64, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I]),
    "rsx_din_parse_h": (_I, [_P, _P, _P, C.c_int64, _I, _P, _P, _P, _P, _P, _I]),
    "rsx_criteo_encode_h": (C.c_int64, [_P, _P, _P, _P, C.c_int64, _P, C.c_int64]),
    "rsx_din_encode_h": (C.c_int64, [_P, _P, _P, _P, _P, C.c_int64, _I, _I, _P, C.c_int64]),
    "rsx_crc32c_h": (C.c_uint32, [_P, C.c_size_t]),
    "rsx_masked_crc32c_h": (C.c_uint32, [_P, C.c_size_t]),
    "rsx_int64_features_parse_h": (_I, [_P, _P, _P, C.c_int64, _P, _I, _P, _I]),
    "rsx_hash_int64_keys_h": (_I, [_P, C.c_int64, C.c_uint64, _P]),
    "rsx_crc32c_table_h": (C.c_uint32, [_P, C.c_size_t]),
    "rsx_criteo_reader_open_h": (_P, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "rsx_criteo_reader_next_h": (_I, [_P, _P, _P, _P]),
    "rsx_din_reader_open_h": (_P, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "rsx_din_reader_next_h": (_I, [_P, _P, _P, _P, _P, _P]),
    "rsx_reader_records_parsed_h": (C.c_int64, [_P]),
    "rsx_reader_close_h": (None, [_P]),
    "rsx_mlp_nobn_supported": (_I, [_I, _P, _I]),
    "rsx_mlp_nobn_workspace_floats": (C.c_size_t, [_I, _I, _P, _I]),
    "rsx_mlp_nobn_train_step": (_I, [C.POINTER(MlpStep), _P]),
    "rsx_mlp_nobn_reduce": (_I, [C.POINTER(MlpStep), _P]),
    "rsx_mlp_nobn_reduce_job": (_I, [C.POINTER(MlpStep), C.POINTER(MlpReduceJob)]),
    "rsx_eval_metrics_state_words": (_I, [_I]),
    "rsx_eval_metrics_update": (_I, [_P, _P, _P, _I, _P, _P, _I, _P]),
}

_lib = None


def lib():
    """Load librsx.so once.  Raises RsxError (never falls back) when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RsxError("librsx.so not built: run `python -m recsys_amd.build` (hipcc --offload-arch=gfx950); "
                       "there is no CPU fallback for the hot path")
    # torch first: it bundles its own libamdhip64; librsx.so must bind to THAT runtime (the streams and device pointers it
    # receives belong to it).  Loaded the other way round, /opt/rocm's copy comes in as a second HIP runtime and the first
    # kernel launch fails.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        if not hasattr(L, name):
            raise RsxError("librsx.so does not export %s (stale build?)" % name)
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def exported_symbols():
    return sorted(_SIGS)


def check(status, what="rsx call"):
    if status != 0:
        raise RsxError("%s failed: %s (%d)" % (what, lib().rsx_strerror(status).decode(), status))
