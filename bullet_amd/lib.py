"""ctypes binding of libbullet_amd.so (include/bullet_amd.h). Fails loudly if the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BT_LIB_PATH: developer override to time experimental builds of the same library (never a fallback)
SO_PATH = os.environ.get("BT_LIB_PATH") or os.path.join(_HERE, "libbullet_amd.so")

ACT = {"identity": 0, "relu": 1, "crelu": 2, "screlu": 3, "sqrrelu": 4}
INPUT = {"chess768": 0, "chess_buckets": 1, "chess_buckets_mirrored": 2, "ataxx147": 3, "ataxx98": 4}
MAX_DENSE = 4
ADAMW_MAX_TENSORS = 16


class BtError(RuntimeError):
    pass


class FtDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("batch", "rows", "cols", "stride", "offset", "nnz", "act", "pairwise", "n_inputs", "out_ld",
                 "segment")]


class DenseDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("batch", "in_", "out", "buckets", "act")]


class AdamwTensor(C.Structure):
    _fields_ = [("grad", C.c_void_p), ("param", C.c_void_p), ("momentum", C.c_void_p), ("velocity", C.c_void_p),
                ("n", C.c_size_t), ("decay", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("min_weight", C.c_float), ("max_weight", C.c_float)]


class NetDesc(C.Structure):
    _fields_ = [
        ("num_inputs", C.c_int), ("nnz", C.c_int), ("ft_out", C.c_int), ("ft_act", C.c_int),
        ("pairwise", C.c_int), ("perspectives", C.c_int), ("num_buckets", C.c_int), ("n_dense", C.c_int),
        ("dense_out", C.c_int * MAX_DENSE), ("dense_act", C.c_int * MAX_DENSE),
        ("dense_bucketed", C.c_int * MAX_DENSE), ("use_entry_weights", C.c_int),
        ("ft_factoriser", C.c_int), ("dense_dual", C.c_int * MAX_DENSE), ("dense_skip", C.c_int * MAX_DENSE),
        ("ft_segment", C.c_int), ("loss_kind", C.c_int), ("loss_power", C.c_float),
    ]


LOSS = {"sigmoid_mse": 0, "sigmoid_power": 1, "softmax_ce": 2}


SAVE_TRANSFORM = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_size_t)
QUANT = {"float": 0, "i8": 1, "i16": 2, "i32": 3}


class SavedFormat(C.Structure):
    _fields_ = [("id", C.c_char_p), ("custom", C.c_void_p), ("custom_len", C.c_size_t), ("quant", C.c_int),
                ("multiplier", C.c_int), ("round", C.c_int), ("transpose", C.c_int), ("merge_factoriser", C.c_int),
                ("transform", SAVE_TRANSFORM), ("user", C.c_void_p)]


def saved_formats(entries):
    """entries: list of dicts {id|custom, quant=("i16", 255), round, transpose, merge_factoriser, transform}.
    Returns (ctypes array, keepalive list)."""
    arr = (SavedFormat * len(entries))()
    keep = []
    for f, e in zip(arr, entries):
        if "custom" in e:
            b = C.create_string_buffer(bytes(e["custom"]), len(e["custom"]))
            keep.append(b)
            f.custom, f.custom_len = C.cast(b, C.c_void_p), len(e["custom"])
            continue
        f.id = e["id"].encode()
        q = e.get("quant")
        if q:
            f.quant, f.multiplier = QUANT[q[0]], int(q[1])
        f.round, f.transpose = int(e.get("round", False)), int(e.get("transpose", False))
        f.merge_factoriser = int(e.get("merge_factoriser", False))
        if e.get("transform"):
            cb = SAVE_TRANSFORM(e["transform"])
            keep.append(cb)
            f.transform = cb
    return arr, keep


class Batch(C.Structure):
    _fields_ = [("batch", C.c_int), ("stm", C.c_void_p), ("nstm", C.c_void_p), ("buckets", C.c_void_p),
                ("targets", C.c_void_p), ("entry_weights", C.c_void_p)]


class InputDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("factorised", C.c_int), ("buckets", C.c_int * 64), ("num_buckets", C.c_int)]


class LrSched(C.Structure):
    _fields_ = [("kind", C.c_int), ("a", C.c_float), ("b", C.c_float), ("n", C.c_int), ("warmup_batches", C.c_int)]


class WdlSched(C.Structure):
    _fields_ = [("kind", C.c_int), ("start", C.c_float), ("end", C.c_float), ("warmup_batches", C.c_int)]


class TrainConfig(C.Structure):
    _fields_ = [("batch_size", C.c_int), ("batches_per_superbatch", C.c_int), ("start_superbatch", C.c_int),
                ("end_superbatch", C.c_int), ("lr", LrSched), ("wdl", WdlSched), ("eval_scale", C.c_float),
                ("out_buckets", C.c_int), ("use_win_rate_model", C.c_int), ("threads", C.c_int),
                ("log_rate", C.c_int), ("wdl_output", C.c_int), ("device_mapper", C.c_int),
                ("dp_rank", C.c_int), ("dp_world", C.c_int), ("grad_sync", C.c_void_p),
                ("grad_sync_user", C.c_void_p), ("dp_loss_slot", C.c_int), ("stop_superbatch", C.c_int), ("reader", C.c_void_p),
                ("lr_fn", C.c_void_p), ("wdl_fn", C.c_void_p), ("sched_user", C.c_void_p),
                ("blend_fn", C.c_void_p), ("weight_fn", C.c_void_p), ("fn_user", C.c_void_p),
                ("comm", C.c_void_p), ("dp_shard_min_floats", C.c_size_t)]


COMM_ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
COMM_SCATTER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class ViriBuiltinFilter(C.Structure):  # bt_viri_builtin_filter
    _fields_ = [("min_ply", C.c_uint32), ("min_pieces", C.c_uint32), ("max_eval", C.c_uint32),
                ("filter_tactical", C.c_int), ("filter_check", C.c_int), ("filter_castling", C.c_int),
                ("max_eval_incorrectness", C.c_uint32), ("random_fen_skipping", C.c_int),
                ("random_fen_skip_probability", C.c_double)]


class CommHooks(C.Structure):
    _fields_ = [("all_reduce", COMM_ALLREDUCE_CB), ("reduce_scatter", COMM_SCATTER_CB), ("all_gather", COMM_SCATTER_CB),
                ("user", C.c_void_p)]


class DpPlan(C.Structure):
    _fields_ = [("sharded", C.c_int), ("table_n", C.c_size_t), ("per_rank", C.c_size_t), ("slice_lo", C.c_size_t),
                ("tail_off", C.c_size_t), ("tail_n", C.c_size_t)]


BATCH_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_float)
GRAD_SYNC_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
SUPERBATCH_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_float, C.c_float)
LR = {"constant": 0, "drop": 1, "step": 2, "linear": 3, "cosine": 4, "exp": 5}
WDL = {"constant": 0, "linear": 1}


class MapOpts(C.Structure):
    _fields_ = [("out_buckets", C.c_int), ("blend", C.c_float), ("eval_scale", C.c_float),
                ("use_win_rate_model", C.c_int), ("wdl_output", C.c_int), ("threads", C.c_int),
                ("blend_fn", C.c_void_p), ("weight_fn", C.c_void_p), ("fn_user", C.c_void_p)]


# every symbol include/bullet_amd.h declares (tests/test_abi.py checks the .so exports them all)
SYMBOLS = [
    "bt_version", "bt_last_error",
    "bt_device_count", "bt_device_set", "bt_device_props", "bt_device_sync", "bt_malloc", "bt_free",
    "bt_host_alloc_pinned", "bt_host_free_pinned", "bt_stream_create", "bt_stream_destroy", "bt_stream_sync",
    "bt_event_create", "bt_event_destroy", "bt_event_record", "bt_stream_wait_event", "bt_event_sync", "bt_event_query",
    "bt_memset_async", "bt_memcpy_h2d_async", "bt_memcpy_d2h_async", "bt_memcpy_d2d_async",
    "bt_comm_use_own_stream", "bt_comm_available", "bt_profile_start", "bt_profile_set_filter", "bt_profile_only", "bt_profile_set_sampling", "bt_profile_stop", "bt_profile_num_kernels", "bt_profile_kernel_name",
    "bt_ft_forward", "bt_ft_backward_workspace_bytes", "bt_ft_backward", "bt_ft_build_index",
    "bt_ft_backward_prebuilt", "bt_ft_forward_prebuilt", "bt_ft_segment_violations",
    "bt_dense_forward", "bt_dense_backward_workspace_bytes", "bt_dense_backward", "bt_dense_backward_static",
    "bt_sigmoid_mse", "bt_loss", "bt_adamw", "bt_radam", "bt_lookahead", "bt_weight_clip", "bt_weight_decay",
    "bt_clip_pass_through", "bt_faux_quantise",
    "bt_net_set_optimiser",
    "bt_net_param_count", "bt_net_num_tensors", "bt_net_tensor_size", "bt_net_tensor_offset", "bt_net_create",
    "bt_net_destroy", "bt_net_params", "bt_net_grads", "bt_net_momentum", "bt_net_velocity", "bt_net_set_adamw",
    "bt_net_flat_size", "bt_net_max_batch", "bt_net_grads_owned", "bt_chessboard_from_str", "bt_measure_loader_throughput", "bt_reader_open_viribinpack", "bt_reader_open_sfbinpack",
    "bt_reader_open_viribinpack_builtin", "bt_viri_filter_default", "bt_viri_filter_unrestricted", "bt_reader_open_montybinpack", "bt_net_use_graphs", "bt_net_graph_launches",
    "bt_net_forward_backward", "bt_net_optimiser_step", "bt_net_optimiser_step_event", "bt_net_train_step", "bt_net_forward", "bt_net_hint_next_batch",
    "bt_input_init", "bt_input_num_inputs", "bt_input_max_active", "bt_map_batch", "bt_map_batch_device", "bt_map_batch_device_errors",
    "bt_net_init_weights_host", "bt_lr_at", "bt_wdl_at", "bt_reader_open", "bt_reader_next", "bt_reader_close",
    "bt_train_run",
    "bt_net_tensor_name", "bt_net_tensor_shape", "bt_checkpoint_write", "bt_checkpoint_read", "bt_save_quantised",
    "bt_save_raw", "bt_net_save_checkpoint", "bt_net_load_checkpoint",
    "bt_comm_unique_id", "bt_comm_init_rank", "bt_comm_from_hooks", "bt_comm_destroy", "bt_comm_rank", "bt_comm_world",
    "bt_allreduce_sum_f32", "bt_reduce_scatter_sum_f32", "bt_allgather_f32",
    "bt_net_dp_plan", "bt_net_dp_step", "bt_net_dp_gather_state",
]

_lib = None


def load():
    """Load libbullet_amd.so; raises BtError (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise BtError(f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(SO_PATH)
    L.bt_last_error.restype = C.c_char_p
    L.bt_profile_kernel_name.restype = C.c_char_p
    L.bt_profile_set_filter.argtypes = [C.c_ulonglong]
    L.bt_profile_stop.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int]
    for f in ("bt_ft_backward_workspace_bytes", "bt_dense_backward_workspace_bytes", "bt_net_param_count",
              "bt_net_tensor_size", "bt_net_tensor_offset", "bt_net_flat_size"):
        getattr(L, f).restype = C.c_size_t
    for f in ("bt_net_params", "bt_net_grads", "bt_net_momentum", "bt_net_velocity"):
        getattr(L, f).restype = C.c_void_p
        getattr(L, f).argtypes = [C.c_void_p]
    L.bt_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    L.bt_chessboard_from_str.argtypes = [C.c_char_p, C.c_void_p]
    L.bt_reader_open_viribinpack.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_uint64, C.POINTER(C.c_void_p)]
    L.bt_reader_open_sfbinpack.argtypes = L.bt_reader_open_viribinpack.argtypes
    L.bt_reader_open_montybinpack.argtypes = L.bt_reader_open_viribinpack.argtypes
    L.bt_reader_open_viribinpack_builtin.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.POINTER(ViriBuiltinFilter),
                                                     C.c_uint64, C.POINTER(C.c_void_p)]
    for f in ("bt_viri_filter_default", "bt_viri_filter_unrestricted"):
        getattr(L, f).argtypes = [C.POINTER(ViriBuiltinFilter)]
        getattr(L, f).restype = None
    L.bt_reader_next.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.bt_reader_close.argtypes = [C.c_void_p]
    L.bt_measure_loader_throughput.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.bt_free.argtypes = [C.c_void_p]
    L.bt_host_alloc_pinned.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    L.bt_host_free_pinned.argtypes = [C.c_void_p]
    L.bt_stream_sync.argtypes = [C.c_void_p]
    L.bt_stream_create.argtypes = [C.POINTER(C.c_void_p)]
    L.bt_stream_destroy.argtypes = [C.c_void_p]
    L.bt_event_create.argtypes = [C.POINTER(C.c_void_p)]
    L.bt_event_destroy.argtypes = [C.c_void_p]
    L.bt_event_record.argtypes = [C.c_void_p, C.c_void_p]
    L.bt_event_sync.argtypes = [C.c_void_p]
    L.bt_event_query.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.bt_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p]
    L.bt_memset_async.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    for f in ("bt_memcpy_h2d_async", "bt_memcpy_d2h_async", "bt_memcpy_d2d_async"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.bt_ft_forward.argtypes = [C.c_void_p, C.POINTER(FtDesc), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                C.c_void_p, C.c_void_p]
    L.bt_ft_backward_workspace_bytes.argtypes = [C.POINTER(FtDesc)]
    L.bt_ft_backward.argtypes = [C.c_void_p, C.POINTER(FtDesc), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.bt_ft_build_index.argtypes = [C.c_void_p, C.POINTER(FtDesc), C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t]
    L.bt_ft_backward_prebuilt.argtypes = [C.c_void_p, C.POINTER(FtDesc)] + [C.c_void_p] * 6 + [C.c_size_t]
    L.bt_ft_forward_prebuilt.argtypes = [C.c_void_p, C.POINTER(FtDesc), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.bt_ft_segment_violations.argtypes = [C.c_void_p, C.POINTER(FtDesc), C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    L.bt_dense_forward.argtypes = [C.c_void_p, C.POINTER(DenseDesc)] + [C.c_void_p] * 5
    L.bt_dense_backward_workspace_bytes.argtypes = [C.POINTER(DenseDesc)]
    L.bt_dense_backward.argtypes = [C.c_void_p, C.POINTER(DenseDesc)] + [C.c_void_p] * 9 + [C.c_size_t]
    L.bt_dense_backward_static.argtypes = [C.c_void_p, C.POINTER(DenseDesc)] + [C.c_void_p] * 8
    L.bt_sigmoid_mse.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    L.bt_loss.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                          C.c_void_p, C.c_void_p]
    L.bt_weight_clip.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float]
    L.bt_weight_decay.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]
    L.bt_clip_pass_through.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float]
    L.bt_faux_quantise.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_int]
    L.bt_adamw.argtypes = [C.c_void_p, C.c_int, C.POINTER(AdamwTensor), C.c_void_p, C.c_void_p]
    L.bt_radam.argtypes = [C.c_void_p, C.c_int, C.POINTER(AdamwTensor), C.c_void_p, C.c_void_p, C.c_int, C.c_float]
    L.bt_lookahead.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                               C.c_float]
    L.bt_net_set_optimiser.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int]
    L.bt_net_param_count.argtypes = [C.POINTER(NetDesc)]
    L.bt_net_num_tensors.argtypes = [C.POINTER(NetDesc)]
    L.bt_net_tensor_size.argtypes = [C.POINTER(NetDesc), C.c_int]
    L.bt_net_tensor_offset.argtypes = [C.POINTER(NetDesc), C.c_int]
    L.bt_net_tensor_name.argtypes = [C.POINTER(NetDesc), C.c_int]
    L.bt_net_tensor_name.restype = C.c_char_p
    L.bt_net_tensor_shape.argtypes = [C.POINTER(NetDesc), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.bt_checkpoint_write.argtypes = [C.POINTER(NetDesc), C.c_void_p, C.c_char_p]
    L.bt_checkpoint_read.argtypes = [C.POINTER(NetDesc), C.c_void_p, C.c_char_p]
    L.bt_save_quantised.argtypes = [C.POINTER(NetDesc), C.c_void_p, C.POINTER(SavedFormat), C.c_int, C.c_char_p]
    L.bt_save_raw.argtypes = [C.POINTER(NetDesc), C.c_void_p, C.POINTER(SavedFormat), C.c_int, C.c_char_p]
    L.bt_net_save_checkpoint.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(SavedFormat), C.c_int, C.c_char_p]
    L.bt_net_load_checkpoint.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
    L.bt_net_create.argtypes = [C.POINTER(NetDesc), C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.bt_net_destroy.argtypes = [C.c_void_p]
    L.bt_net_set_adamw.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 5
    L.bt_net_forward_backward.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.c_void_p]
    L.bt_net_hint_next_batch.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p]
    L.bt_net_optimiser_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bt_net_optimiser_step_event.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bt_net_train_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bt_net_forward.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p]
    L.bt_input_init.argtypes = [C.POINTER(InputDesc), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.bt_input_num_inputs.argtypes = [C.POINTER(InputDesc)]
    L.bt_input_max_active.argtypes = [C.POINTER(InputDesc)]
    L.bt_map_batch.argtypes = [C.POINTER(InputDesc), C.POINTER(MapOpts), C.c_void_p, C.c_int] + [C.c_void_p] * 5
    L.bt_map_batch_device.argtypes = [C.c_void_p, C.POINTER(InputDesc), C.POINTER(MapOpts), C.c_void_p, C.c_int] + \
        [C.c_void_p] * 5
    L.bt_map_batch_device_errors.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.bt_net_init_weights_host.argtypes = [C.POINTER(NetDesc), C.c_uint64, C.c_int, C.c_void_p]
    L.bt_lr_at.restype = C.c_float
    L.bt_lr_at.argtypes = [C.POINTER(LrSched), C.c_int, C.c_int]
    L.bt_wdl_at.restype = C.c_float
    L.bt_wdl_at.argtypes = [C.POINTER(WdlSched), C.c_int, C.c_int, C.c_int]
    L.bt_reader_open.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
    L.bt_reader_next.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.bt_reader_close.argtypes = [C.c_void_p]
    L.bt_train_run.argtypes = [C.c_void_p, C.POINTER(InputDesc), C.POINTER(TrainConfig), C.POINTER(C.c_char_p), C.c_int,
                               BATCH_CB, SUPERBATCH_CB, C.c_void_p]
    L.bt_comm_unique_id.argtypes = [C.c_void_p]
    L.bt_comm_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.bt_comm_from_hooks.argtypes = [C.POINTER(CommHooks), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    for f in ("bt_comm_destroy", "bt_comm_rank", "bt_comm_world"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.bt_allreduce_sum_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.bt_reduce_scatter_sum_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.bt_allgather_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.bt_net_dp_plan.argtypes = [C.POINTER(NetDesc), C.c_int, C.c_int, C.c_size_t, C.POINTER(DpPlan)]
    L.bt_net_dp_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.bt_net_dp_gather_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise BtError(f"[{rc}] {load().bt_last_error().decode()}")


def profile_start(max_launches=100000, only=None, every=1):
    """only: iterable of kernel names (or parts of names) to time (None = all); every: time every n-th launch of each"""
    L = load()
    check(L.bt_profile_start(max_launches))
    if every > 1:
        check(L.bt_profile_set_sampling(every))
    for n in only or ():
        check(L.bt_profile_only(n.encode()))


def profile_stop():
    """-> {kernel_name: (launches, total_ms)} for kernels that ran"""
    L = load()
    n = L.bt_profile_num_kernels()
    counts, total = (C.c_int * n)(), (C.c_double * n)()
    check(L.bt_profile_stop(counts, total, n))
    return {L.bt_profile_kernel_name(i).decode(): (counts[i], total[i]) for i in range(n) if counts[i]}


def make_net_desc(num_inputs, nnz, ft_out, ft_act, dense, pairwise=False, perspectives=2, num_buckets=1,
                  use_entry_weights=False, ft_factoriser=0, loss="sigmoid_mse", loss_power=2.0,
                  ft_segment=0):
    """dense: list of (out_per_bucket, act_name, bucketed[, dual[, skip]]); ft_factoriser = NB repeats of l0f;
    loss: "sigmoid_mse" | "sigmoid_power" (with loss_power) | "softmax_ce" (last layer = classes per bucket)"""
    d = NetDesc()
    d.num_inputs, d.nnz, d.ft_out, d.ft_act = num_inputs, nnz, ft_out, ACT[ft_act]
    d.pairwise, d.perspectives, d.num_buckets, d.n_dense = int(pairwise), perspectives, num_buckets, len(dense)
    for k, spec in enumerate(dense):
        o, a, bk = spec[:3]
        d.dense_out[k], d.dense_act[k], d.dense_bucketed[k] = o, ACT[a], int(bk)
        d.dense_dual[k] = int(spec[3]) if len(spec) > 3 else 0
        d.dense_skip[k] = int(spec[4]) if len(spec) > 4 else 0
    d.use_entry_weights = int(use_entry_weights)
    d.ft_factoriser = ft_factoriser
    d.loss_kind, d.loss_power = LOSS[loss], float(loss_power)
    d.ft_segment = ft_segment
    return d


def make_input_desc(kind, factorised=False, layout=None):
    d = InputDesc()
    lay = (C.c_int * len(layout))(*layout) if layout is not None else None
    check(load().bt_input_init(C.byref(d), INPUT[kind], int(factorised), lay, len(layout) if layout else 0))
    return d


def map_batch(idesc, records, out_buckets=1, blend=0.0, eval_scale=400.0, win_rate_model=False, wdl=False,
              threads=1, out=None):
    """Host batch preparation through the C ABI. records: numpy uint8 [count*32]."""
    import numpy as np
    L = load()
    count = records.size // 32
    nnz = L.bt_input_max_active(C.byref(idesc))
    if out is None:
        out = (np.zeros((count, nnz), np.int32), np.zeros((count, nnz), np.int32), np.zeros(count, np.int32),
               np.zeros(count * (3 if wdl else 1), np.float32), np.zeros(count, np.float32))
    o = MapOpts(out_buckets, blend, eval_scale, int(win_rate_model), int(wdl), threads)
    check(L.bt_map_batch(C.byref(idesc), C.byref(o), records.ctypes.data, count, *[a.ctypes.data for a in out]))
    return out


class Net:
    """Thin owner of a bt_net plus torch views of its flat buffers (device memory plumbing only)."""

    def __init__(self, desc, max_batch, device="cuda"):
        import torch
        self.L = load()
        self.desc = desc
        self.max_batch = max_batch
        self.n_tensors = self.L.bt_net_num_tensors(C.byref(desc))
        self.sizes = [self.L.bt_net_tensor_size(C.byref(desc), t) for t in range(self.n_tensors)]
        self.offsets = [self.L.bt_net_tensor_offset(C.byref(desc), t) for t in range(self.n_tensors)]
        self.n_flat = self.L.bt_net_param_count(C.byref(desc))
        # flat params / grads live in torch tensors so a harness can all-reduce / copy them
        self.params = torch.zeros(self.n_flat, dtype=torch.float32, device=device)
        # gradient and loss sum share one allocation (n_flat is a multiple of 4 floats), so data-parallel
        # training all-reduces `step_buffer` once per step
        self._gbuf = torch.zeros(self.n_flat + 4, dtype=torch.float32, device=device)
        self.grads = self._gbuf[:self.n_flat]
        self.loss = self._gbuf[self.n_flat:self.n_flat + 1]
        self.step_buffer = self._gbuf[:self.n_flat + 1]
        h = C.c_void_p()
        check(self.L.bt_net_create(C.byref(desc), max_batch, self.params.data_ptr(), self.grads.data_ptr(),
                                   C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.L.bt_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tensor(self, flat, t):
        return flat[self.offsets[t]:self.offsets[t] + self.sizes[t]]

    def init_weights(self, seed=198273612, ft_effective_inputs=0):
        """ModelWeights::new with the reference's RNG stream, uploaded to the device."""
        import numpy as np
        import torch
        host = np.zeros(self.n_flat, dtype=np.float32)
        check(self.L.bt_net_init_weights_host(C.byref(self.desc), seed, ft_effective_inputs, host.ctypes.data))
        self.params.copy_(torch.from_numpy(host).to(self.params.device))

    def save_checkpoint(self, path, entries, stream=None):
        """save_to_checkpoint (value/save.rs:24-43)"""
        import torch
        arr, keep = saved_formats(entries)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.L.bt_net_save_checkpoint(self.h, s, arr, len(entries), str(path).encode()))

    def load_checkpoint(self, path, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.L.bt_net_load_checkpoint(self.h, s, str(path).encode()))

    def train(self, input_desc, paths, batch_size, batches_per_superbatch, end_superbatch, lr, wdl, eval_scale=400.0,
              start_superbatch=1, out_buckets=1, threads=4, log_rate=0, win_rate_model=False, on_batch=None,
              on_superbatch=None, wdl_output=False, device_mapper=False, dist=None, reader=None, comm=None,
              shard_min_floats=0):
        """run::train through the C ABI. lr / wdl: LrSched / WdlSched. Returns [(superbatch, running_loss, pos/s)].
        dist: an initialised torch.distributed module -> data-parallel training (bt_train_config.dp_*): `batch_size`
        is the global batch, this rank trains on its shard, and the step buffer (flat gradient + loss sum) is
        all-reduced once per step on the training loop's compute stream (RCCL when the backend is "nccl").
        comm: a parallel.Comm (bt_comm: native RCCL, or hooks) instead of `dist` -- the loop then exchanges through
        bt_net_dp_step (no Python on the step path; a table of >= shard_min_floats floats is sharded)."""
        import torch
        torch.cuda.synchronize()
        cfg = TrainConfig(batch_size, batches_per_superbatch, start_superbatch, end_superbatch, lr, wdl, eval_scale,
                          out_buckets, int(win_rate_model), threads, log_rate, int(wdl_output), int(device_mapper))
        sync_cb, sync_err = None, []
        if comm is not None:  # (a communicator of one rank still goes through bt_net_dp_step)
            cfg.dp_rank, cfg.dp_world, cfg.dp_loss_slot = comm.rank, comm.world, 1
            cfg.comm, cfg.dp_shard_min_floats = comm.h, shard_min_floats
            sync_err = comm.errors
        elif dist is not None and dist.get_world_size() > 1:
            def grad_sync(_u, buf, n, stream):
                try:
                    if buf != self._gbuf.data_ptr() or n > self._gbuf.numel():
                        # (only the loss when the caller's buffer has no slot for it: never the case for Net)
                        raise RuntimeError("grad_sync: not the step buffer of this net")
                    with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                        if dist.get_backend() == "gloo":  # CPU collective (tests): staged through the host
                            host = self._gbuf[:n].cpu()
                            dist.all_reduce(host)
                            self._gbuf[:n].copy_(host)
                        else:
                            dist.all_reduce(self._gbuf[:n])
                    return 0
                except Exception as e:  # an exception must not unwind through the C frames
                    sync_err.append(e)
                    return 1
            sync_cb = GRAD_SYNC_CB(grad_sync)
            cfg.dp_rank, cfg.dp_world, cfg.dp_loss_slot = dist.get_rank(), dist.get_world_size(), 1
            cfg.grad_sync = C.cast(sync_cb, C.c_void_p)
        report = []

        def sb_cb(_u, sb, loss, pps):
            report.append((sb, loss, pps))
            if on_superbatch:
                on_superbatch(sb, loss, pps)

        def b_cb(_u, sb, b, loss):
            if on_batch:
                on_batch(sb, b, loss)

        if reader is not None:  # a bt_reader the caller opened (bt_reader_open_viribinpack), instead of `paths`
            cfg.reader = reader
            paths = paths or []
        cpaths = (C.c_char_p * max(len(paths), 1))(*[p.encode() for p in paths])
        rc = self.L.bt_train_run(self.h, C.byref(input_desc), C.byref(cfg), cpaths, len(paths), BATCH_CB(b_cb),
                                 SUPERBATCH_CB(sb_cb), None)
        if sync_err:
            raise sync_err[0]
        check(rc)
        return report

    def eval_raw_output(self, fen, input_desc, out_buckets=1, wdl_output=False):
        """ValueTrainer::eval_raw_output (value.rs:213-235): "{fen} | 0 | 0.0" -> ChessBoard -> mapper -> forward of a
        batch of one through bt_net_forward (ModelEvaluator)."""
        import numpy as np
        import torch
        rec = np.zeros(32, dtype=np.uint8)
        check(self.L.bt_chessboard_from_str(f"{fen} | 0 | 0.0".encode(), rec.ctypes.data))
        stm, ntm, bk, tg, w = map_batch(input_desc, rec, out_buckets=out_buckets, blend=1.0, wdl=wdl_output, threads=1)
        t = [torch.from_numpy(a).cuda() for a in (stm, ntm, bk, tg, w)]
        width = self.desc.dense_out[self.desc.n_dense - 1]
        out = torch.zeros(width, dtype=torch.float32, device="cuda")
        loss = torch.zeros(1, dtype=torch.float32, device="cuda")
        check(self.L.bt_net_forward(self.h, torch.cuda.current_stream().cuda_stream, C.byref(self.make_batch(*t)),
                                    out.data_ptr(), loss.data_ptr()))
        return out.cpu().numpy()

    def eval(self, fen, input_desc, out_buckets=1, wdl_output=False):
        """ValueTrainer::eval (value.rs:237-255): the raw score, or the expected result of a W/D/L head"""
        import numpy as np
        vals = self.eval_raw_output(fen, input_desc, out_buckets, wdl_output)
        if vals.size == 3:
            e = np.exp(vals - vals.max())  # [loss, draw, win]
            return float((e[2] + e[1] / np.float32(2.0)) / e.sum())
        if vals.size == 1:
            return float(vals[0])
        raise BtError("Invalid output size!")

    def set_params(self, arrays):
        import torch
        for t, a in enumerate(arrays):
            self.tensor(self.params, t).copy_(torch.from_numpy(a).to(self.params.device))

    def get_tensors(self, flat):
        return [self.tensor(flat, t).detach().cpu().numpy().copy() for t in range(self.n_tensors)]

    def state_view(self, which):
        """momentum / velocity as numpy lists (copies)"""
        import numpy as np
        import torch
        ptr = (self.L.bt_net_momentum if which == "momentum" else self.L.bt_net_velocity)(self.h)
        host = np.zeros(self.n_flat, dtype=np.float32)
        torch.cuda.synchronize()
        check(self.L.bt_memcpy_d2h_async(host.ctypes.data, ptr, host.nbytes, None))
        check(self.L.bt_device_sync())
        return [host[o:o + s].copy() for o, s in zip(self.offsets, self.sizes)]

    def make_batch(self, stm, nstm, buckets, targets, weights=None):
        b = Batch()
        b.batch = stm.shape[0] if stm.dim() == 2 else targets.shape[0]
        b.stm = stm.data_ptr()
        b.nstm = nstm.data_ptr() if nstm is not None else None
        b.buckets = buckets.data_ptr() if buckets is not None else None
        b.targets = targets.data_ptr()
        b.entry_weights = weights.data_ptr() if weights is not None else None
        b._keep = (stm, nstm, buckets, targets, weights)
        return b

    def forward_backward(self, batch, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.L.bt_net_forward_backward(self.h, s, C.byref(batch), self.loss.data_ptr()))

    def train_step(self, batch, adj_dev, lr_dev, stream=None, done_event=None):
        """bt_net_train_step: forward + backward + optimiser update of one batch (single device)"""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.L.bt_net_train_step(self.h, s, C.byref(batch), self.loss.data_ptr(), adj_dev.data_ptr(),
                                       lr_dev.data_ptr(), done_event))

    def hint_next(self, batch, ready=None):
        """bt_net_hint_next_batch: `batch` is what the step after the coming one uses; its tensors are complete now, or
        once `ready` (a recorded event: a hipEvent_t handle or a torch.cuda.Event) has fired"""
        ev = None if ready is None else ready if isinstance(ready, (int, C.c_void_p)) else C.c_void_p(ready.cuda_event)
        check(self.L.bt_net_hint_next_batch(self.h, C.byref(batch) if batch is not None else None, ev))

    def dp_step(self, comm, adj_dev, lr_dev, stream=None, shard_min_floats=0):
        """bt_net_dp_step: gradient exchange through `comm` (a parallel.Comm) + the optimiser step"""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        rc = self.L.bt_net_dp_step(self.h, comm.h, s, adj_dev.data_ptr(), lr_dev.data_ptr(), shard_min_floats)
        if comm.errors:
            raise comm.errors[0]
        check(rc)

    def optimiser_step(self, adj_dev, lr_dev, stream=None, done_event=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.L.bt_net_optimiser_step_event(self.h, s, adj_dev.data_ptr(), lr_dev.data_ptr(), done_event))
