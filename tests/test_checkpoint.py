"""Checkpoint / saved-network formats (include/bullet_amd.h section 6) against a Python restatement of the reference's
writers. The reference holds no golden files for these formats ("parity unpinned" by data); the restatement below
follows crates/trainer/src/model/weights.rs line by line and the tests compare bytes."""
import ctypes as C
import struct

import numpy as np
import pytest

from bullet_amd import lib

CFG = dict(num_inputs=768 * 2, nnz=32, ft_out=32, ft_act="screlu", num_buckets=8, ft_factoriser=2,
           dense=[(1, "identity", True)])


def flat_and_tensors(desc, seed=5, scale=0.3):
    L = lib.load()
    rng = np.random.default_rng(seed)
    flat = np.zeros(L.bt_net_param_count(C.byref(desc)), np.float32)
    tensors = {}
    for t in range(L.bt_net_num_tensors(C.byref(desc))):
        n, off = L.bt_net_tensor_size(C.byref(desc), t), L.bt_net_tensor_offset(C.byref(desc), t)
        v = (rng.standard_normal(n) * scale).astype(np.float32)
        flat[off:off + n] = v
        r, c = C.c_int(), C.c_int()
        lib.check(L.bt_net_tensor_shape(C.byref(desc), t, C.byref(r), C.byref(c)))
        tensors[L.bt_net_tensor_name(C.byref(desc), t).decode()] = (v, r.value, c.value)
    return flat, tensors


def ref_record(name, v):  # utils::write_to_byte_buffer, weights.rs:404-431
    return name.encode() + b"\n" + struct.pack("<Q", v.size) + v.astype("<f4").tobytes()


def ref_quantise(v, target, q, rnd):  # QuantTarget::quantise, weights.rs:288-331
    if target == "float":
        return v.astype("<f4").tobytes()
    x = np.float64(q) * v.astype(np.float64)
    x = np.where(x >= 0, np.floor(np.abs(x) + 0.5), -np.floor(np.abs(x) + 0.5)) if rnd else np.trunc(x)  # f64::round
    dt = {"i8": np.int8, "i16": np.int16, "i32": np.int32}[target]
    info = np.iinfo(dt)
    if (x < info.min).any() or (x > info.max).any():
        raise OverflowError
    return x.astype(dt).astype("<" + np.dtype(dt).str[1:]).tobytes()


def ref_transpose(v, rows, cols):  # SavedFormat::transpose_impl, weights.rs:262-275
    return v.reshape(cols, rows).T.reshape(-1).copy()


def test_tensor_names_and_shapes():
    L = lib.load()
    d = lib.make_net_desc(**CFG)
    names = [L.bt_net_tensor_name(C.byref(d), t).decode() for t in range(L.bt_net_num_tensors(C.byref(d)))]
    assert names == ["l0w", "l0b", "l1w", "l1b", "l0f"]
    _, tensors = flat_and_tensors(d)
    assert tensors["l0w"][1:] == (32, 1536) and tensors["l0f"][1:] == (32, 768)
    assert tensors["l1w"][1:] == (8, 64) and tensors["l1b"][1:] == (8, 1)


def test_weights_bin_records_in_btreemap_order_and_round_trip(tmp_path):
    L = lib.load()
    d = lib.make_net_desc(**CFG)
    flat, tensors = flat_and_tensors(d)
    p = str(tmp_path / "weights.bin").encode()
    lib.check(L.bt_checkpoint_write(C.byref(d), flat.ctypes.data, p))
    want = b"".join(ref_record(k, tensors[k][0]) for k in sorted(tensors))  # BTreeMap<String, _>: l0b, l0f, l0w, l1b, l1w
    assert open(p, "rb").read() == want
    back = np.full_like(flat, 7.0)
    lib.check(L.bt_checkpoint_read(C.byref(d), back.ctypes.data, p))
    for t in range(L.bt_net_num_tensors(C.byref(d))):
        n, off = L.bt_net_tensor_size(C.byref(d), t), L.bt_net_tensor_offset(C.byref(d), t)
        assert np.array_equal(back[off:off + n], flat[off:off + n])
    # any record order is accepted; unknown ids and wrong sizes are errors (weights.rs:119-127)
    open(p, "wb").write(b"".join(ref_record(k, tensors[k][0]) for k in reversed(sorted(tensors))))
    lib.check(L.bt_checkpoint_read(C.byref(d), back.ctypes.data, p))
    open(p, "wb").write(ref_record("nope", np.zeros(3, np.float32)))
    assert L.bt_checkpoint_read(C.byref(d), back.ctypes.data, p) != 0 and b"No weight with ID found!" in L.bt_last_error()
    open(p, "wb").write(ref_record("l0b", np.zeros(3, np.float32)))
    assert L.bt_checkpoint_read(C.byref(d), back.ctypes.data, p) != 0 and b"Invalid buffer size!" in L.bt_last_error()


def test_quantised_and_raw_files_match_the_reference_writers(tmp_path):
    """The save format of examples/progression/3_input_buckets.rs:45-57 plus a custom header and a truncating entry."""
    L = lib.load()
    d = lib.make_net_desc(**CFG)
    flat, tensors = flat_and_tensors(d)
    entries = [dict(custom=b"NNUE\x01"),
               dict(id="l0w", merge_factoriser=True, round=True, quant=("i16", 255)),
               dict(id="l0b", round=True, quant=("i16", 255)),
               dict(id="l1w", round=True, quant=("i16", 64), transpose=True),
               dict(id="l1b", quant=("i32", 255 * 64)),
               dict(id="l0f", quant=("i8", 64)),
               dict(id="l1b")]
    arr, keep = lib.saved_formats(entries)
    q, r = str(tmp_path / "quantised.bin").encode(), str(tmp_path / "raw.bin").encode()
    lib.check(L.bt_save_quantised(C.byref(d), flat.ctypes.data, arr, len(entries), q))
    lib.check(L.bt_save_raw(C.byref(d), flat.ctypes.data, arr, len(entries), r))
    w0, f0 = tensors["l0w"][0], tensors["l0f"][0]
    merged = w0 + np.tile(f0, 2)  # weights + factoriser.repeat(NUM_INPUT_BUCKETS)
    l1w, rows, cols = tensors["l1w"]
    want = (b"NNUE\x01" + ref_quantise(merged, "i16", 255, True) + ref_quantise(tensors["l0b"][0], "i16", 255, True) +
            ref_quantise(ref_transpose(l1w, rows, cols), "i16", 64, True) +
            ref_quantise(tensors["l1b"][0], "i32", 255 * 64, False) + ref_quantise(f0, "i8", 64, False) +
            ref_quantise(tensors["l1b"][0], "float", 0, False))
    pad = (64 - len(want) % 64) % 64
    want += bytes(b"bullet"[i % 6] for i in range(pad))  # weights.rs:166-173
    got = open(q, "rb").read()
    assert len(got) % 64 == 0 and got == want
    assert open(r, "rb").read() == b"".join(tensors[k][0].astype("<f4").tobytes()
                                             for k in ["l0w", "l0b", "l1w", "l1b", "l0f", "l1b"])


def test_quantisation_overflow_is_an_error_and_callbacks_run(tmp_path):
    L = lib.load()
    d = lib.make_net_desc(**CFG)
    flat, tensors = flat_and_tensors(d, scale=3.0)
    q = str(tmp_path / "q.bin").encode()
    arr, keep = lib.saved_formats([dict(id="l0w", round=True, quant=("i8", 127))])
    assert L.bt_save_quantised(C.byref(d), flat.ctypes.data, arr, 1, q) != 0
    assert b"Failed quantisation from f32 to i8!" in L.bt_last_error()

    def clip(user, name, values, count):  # a SavedFormat::transform closure
        a = np.ctypeslib.as_array(values, shape=(count,))
        np.clip(a, -0.99, 0.99, out=a)
        return 0

    arr, keep = lib.saved_formats([dict(id="l0w", transform=clip, round=True, quant=("i8", 127))])
    lib.check(L.bt_save_quantised(C.byref(d), flat.ctypes.data, arr, 1, q))
    got = open(q, "rb").read()
    want = ref_quantise(np.clip(tensors["l0w"][0], -0.99, 0.99), "i8", 127, True)
    assert got[:len(want)] == want
