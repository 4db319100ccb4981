"""ctypes binding of the CPU oracle (oracle/liboracle.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "liboracle.so")

ACT = {"identity": 0, "relu": 1, "crelu": 2, "screlu": 3, "sqrrelu": 4}
INPUT = {"chess768": 0, "chess_buckets": 1, "chess_buckets_mirrored": 2, "ataxx147": 3, "ataxx98": 4}
MAX_DENSE = 4


class NetDesc(C.Structure):
    _fields_ = [
        ("num_inputs", C.c_int), ("nnz", C.c_int), ("ft_out", C.c_int), ("ft_act", C.c_int),
        ("pairwise", C.c_int), ("perspectives", C.c_int), ("num_buckets", C.c_int), ("n_dense", C.c_int),
        ("dense_out", C.c_int * MAX_DENSE), ("dense_act", C.c_int * MAX_DENSE),
        ("dense_bucketed", C.c_int * MAX_DENSE), ("use_entry_weights", C.c_int),
        ("ft_factoriser", C.c_int), ("dense_dual", C.c_int * MAX_DENSE), ("dense_skip", C.c_int * MAX_DENSE),
        ("loss_kind", C.c_int), ("loss_power", C.c_float),
    ]


LOSS = {"sigmoid_mse": 0, "sigmoid_power": 1, "softmax_ce": 2}


class InputDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("factorised", C.c_int), ("buckets", C.c_int * 64), ("num_buckets", C.c_int)]


class Rng(C.Structure):
    _fields_ = [("s0", C.c_uint64), ("s1", C.c_uint64)]


def build():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_ROOT, "oracle", "bullet_oracle.c")):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_net_forward_backward.restype = C.c_float
        _lib.orc_net_forward.restype = C.c_float
        _lib.orc_sigmoid.restype = C.c_float
        _lib.orc_sigmoid.argtypes = [C.c_float]
        _lib.orc_loss.restype = C.c_float
        _lib.orc_net_tensor_size.restype = C.c_size_t
        _lib.orc_rng_next_u64.restype = C.c_uint64
        _lib.orc_rng_standard_normal.restype = C.c_double
        _lib.orc_material_bucket.argtypes = [C.c_uint64, C.c_int]
    return _lib


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def make_net(num_inputs, nnz, ft_out, ft_act, dense, pairwise=False, perspectives=2, num_buckets=1,
             use_entry_weights=False, ft_factoriser=0, loss="sigmoid_mse", loss_power=2.0, ft_segment=0):
    """dense: list of (out_per_bucket, act_name, bucketed[, dual[, skip]]); ft_segment is a layout promise to the
    device kernels and changes nothing in the arithmetic, so the oracle ignores it"""
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
    return d


def tensor_sizes(d):
    L = lib()
    return [L.orc_net_tensor_size(C.byref(d), t) for t in range(L.orc_net_num_tensors(C.byref(d)))]


def ptr_array(arrs):
    return (C.POINTER(C.c_float) * len(arrs))(*[fp(a) for a in arrs])


def net_init(d, seed=198273612, ft_eff=0):
    params = [np.zeros(n, dtype=np.float32) for n in tensor_sizes(d)]
    lib().orc_net_init(C.byref(d), C.c_uint64(seed), ft_eff, ptr_array(params))
    return params


def loss(kind, y, targets, weights=None, power=2.0):
    """-> (loss sum, grad_y); y, targets: [batch] or [batch, width]"""
    y = np.ascontiguousarray(y, dtype=np.float32)
    B = y.shape[0]
    width = y.size // B
    gy = np.zeros_like(y)
    w = fp(weights) if weights is not None else None
    l = lib().orc_loss(LOSS[kind], B, width, fp(y), fp(targets), w, C.c_float(power), fp(gy))
    return float(l), gy


def weight_clip(w, lo, hi):
    lib().orc_weight_clip(fp(w), C.c_size_t(w.size), C.c_float(lo), C.c_float(hi))


def clip_pass_through(x, lo, hi):
    y = np.empty_like(x)
    lib().orc_clip_pass_through(fp(x), fp(y), C.c_size_t(x.size), C.c_float(lo), C.c_float(hi))
    return y


def faux_quantise(x, value, round_):
    y = np.empty_like(x)
    lib().orc_faux_quantise(fp(x), fp(y), C.c_size_t(x.size), C.c_float(value), C.c_int(int(round_)))
    return y


def weight_decay(w, decay, lr):
    lib().orc_weight_decay(fp(w), C.c_size_t(w.size), C.c_float(decay), C.c_float(lr))


def net_fwd_bwd(d, params, stm, nstm, buckets, targets, weights=None):
    B = stm.shape[0]
    grads = [np.zeros_like(p) for p in params]
    out = np.zeros(B * d.dense_out[d.n_dense - 1], dtype=np.float32)
    w = weights if weights is not None else np.ones(B, dtype=np.float32)
    loss = lib().orc_net_forward_backward(C.byref(d), B, ptr_array(params), ptr_array(grads), ip(stm), ip(nstm),
                                          ip(buckets), fp(targets), fp(w), fp(out))
    return float(loss), grads, out


def radam_step(step, b1=0.9, b2=0.999, thr=5.0):
    ss, dn = C.c_float(), C.c_int()
    lib().orc_radam_step(step, C.c_float(b1), C.c_float(b2), C.c_float(thr), C.byref(ss), C.byref(dn))
    return ss.value, dn.value


def radam(p, m, v, g, adj, lr, step_size, denom, decay=0.0, b1=0.9, b2=0.999, wmin=-3.4028235e38, wmax=3.4028235e38):
    lib().orc_radam(fp(p), fp(m), fp(v), fp(g), C.c_size_t(p.size), C.c_float(adj), C.c_float(lr), C.c_float(step_size),
                    C.c_int(denom), C.c_float(decay), C.c_float(b1), C.c_float(b2), C.c_float(wmin), C.c_float(wmax))


def lookahead(w, slow, alpha):
    lib().orc_lookahead(fp(w), fp(slow), C.c_size_t(w.size), C.c_float(alpha))


def adamw(p, m, v, g, adj, lr, decay=0.01, b1=0.9, b2=0.999, wmin=-1.98, wmax=1.98):
    lib().orc_adamw(fp(p), fp(m), fp(v), fp(g), C.c_size_t(p.size), C.c_float(adj), C.c_float(lr), C.c_float(decay),
                    C.c_float(b1), C.c_float(b2), C.c_float(wmin), C.c_float(wmax))


def input_desc(kind, factorised=False, layout=None):
    d = InputDesc()
    d.kind = INPUT[kind]
    d.factorised = int(factorised)
    d.num_buckets = 1
    if kind == "chess_buckets":
        lib().orc_input_buckets_set((C.c_int * 64)(*layout), C.byref(d))
    elif kind == "chess_buckets_mirrored":
        lib().orc_input_mirrored_expand((C.c_int * 32)(*layout), C.byref(d))
    d.factorised = int(factorised)
    return d


def map_batch(idesc, records, n_out_buckets=1, blend=0.0, eval_scale=400.0, win_rate_model=False, wdl=False,
              stm=None, ntm=None):
    """records: uint8 array [count*32]. Returns stm, ntm, buckets, targets, weights."""
    L = lib()
    count = records.size // 32
    nnz = L.orc_input_max_active(C.byref(idesc))
    if stm is None:
        stm = np.zeros((count, nnz), dtype=np.int32)
        ntm = np.zeros((count, nnz), dtype=np.int32)
    buckets = np.zeros(count, dtype=np.int32)
    targets = np.zeros(count * (3 if wdl else 1), dtype=np.float32)
    weights = np.zeros(count, dtype=np.float32)
    L.orc_map_batch(C.byref(idesc), vp(records), count, n_out_buckets, C.c_float(blend), C.c_float(eval_scale),
                    int(win_rate_model), int(wdl), ip(stm), ip(ntm), ip(buckets), fp(targets), fp(weights))
    return stm, ntm, buckets, targets, weights
