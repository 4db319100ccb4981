"""CPU tests of the host side of the training loop (section 5 of the C ABI): weight init, LR / WDL
schedules and the looping fixed-size reader. Expected values are re-derived here from the reference formulas
(crates/bullet_lib/src/trainer/schedule/{lr,wdl}.rs, crates/trainer/src/reader/fixed_size.rs)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle_binding as ob
from bullet_amd import lib

L = lib.load()
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "batch.bf")
f32 = np.float32


@pytest.mark.parametrize("cfg,eff", [
    (dict(num_inputs=768, nnz=32, ft_out=32, ft_act="screlu", dense=[(1, "identity", False)]), 0),
    (dict(num_inputs=768, nnz=32, ft_out=1024, ft_act="screlu", num_buckets=8, dense=[(1, "identity", True)]), 0),
    (dict(num_inputs=7680, nnz=32, ft_out=128, ft_act="crelu", pairwise=True, num_buckets=8,
          dense=[(16, "screlu", True), (32, "screlu", True), (1, "identity", True)]), 32),
])
def test_weight_init_bit_exact_vs_oracle(cfg, eff):
    """ModelWeights::new: product implementation (trainer.cpp) vs the oracle's independent one; the oracle's is
    pinned by the reference e2e losses (tests/test_e2e_reference_losses.py)."""
    dp, do = lib.make_net_desc(**cfg), ob.make_net(**cfg)
    n = L.bt_net_param_count(C.byref(dp))
    flat = np.full(n, np.nan, dtype=np.float32)
    lib.check(L.bt_net_init_weights_host(C.byref(dp), 198273612, eff, flat.ctypes.data))
    ref = ob.net_init(do, 198273612, eff)
    for t, r in enumerate(ref):
        off = L.bt_net_tensor_offset(C.byref(dp), t)
        assert np.array_equal(flat[off:off + r.size], r), f"tensor {t}"


def test_lr_schedulers():
    def lr(kind, a, b, n, batch, sb, warm=0):
        s = lib.LrSched(lib.LR[kind], a, b, n, warm)
        return L.bt_lr_at(C.byref(s), batch, sb)
    assert lr("constant", 0.001, 0, 0, 5, 7) == f32(0.001)
    assert lr("drop", 0.001, 0.1, 8, 0, 8) == f32(0.001) and lr("drop", 0.001, 0.1, 8, 0, 9) == f32(0.001) * f32(0.1)
    assert lr("step", 0.001, 0.3, 4, 0, 4) == f32(0.001)                      # (4-1)/4 = 0 steps
    assert abs(lr("step", 0.001, 0.3, 4, 0, 9) - 0.001 * 0.3 ** 2) < 1e-10    # (9-1)/4 = 2 steps
    assert abs(lr("linear", 0.001, 0.0001, 10, 0, 5) - (0.001 + 0.5 * (0.0001 - 0.001))) < 1e-9
    assert lr("linear", 0.001, 0.0001, 10, 0, 10) == f32(0.0001)
    for sb in (1, 100, 319, 320, 400):
        exp = 0.001 * 0.3 ** 5 if sb >= 320 else 0.001 + (1 - 0.5 * (1 + math.cos(math.pi * sb / 320))) * (0.001 * 0.3 ** 5 - 0.001)
        assert abs(lr("cosine", 0.001, 0.001 * 0.3 ** 5, 320, 0, sb) - exp) < 2e-9
    assert abs(lr("exp", 0.001, 0.00001, 10, 0, 5) - 0.001 * (0.01) ** 0.5) < 1e-9
    # Warmup: base / (warmup - batch) during superbatch 1 only (lr.rs:174-187)
    assert lr("constant", 0.001, 0, 0, 0, 1, warm=10) == f32(0.001) / f32(10)
    assert lr("constant", 0.001, 0, 0, 9, 1, warm=10) == f32(0.001)
    assert lr("constant", 0.001, 0, 0, 3, 2, warm=10) == f32(0.001)


def test_wdl_schedulers():
    def wdl(kind, s, e, batch, sb, mx, warm=0):
        w = lib.WdlSched(lib.WDL[kind], s, e, warm)
        return L.bt_wdl_at(C.byref(w), batch, sb, mx)
    assert wdl("constant", 0.75, 0, 3, 4, 10) == f32(0.75)
    # LinearWDL: start + (end-start)/max(max-1,1) * (superbatch-1)  (wdl.rs:40-45)
    assert wdl("linear", 0.2, 0.0, 0, 1, 10) == f32(0.2)
    assert abs(wdl("linear", 0.2, 0.0, 0, 10, 10)) < 1e-8
    assert abs(wdl("linear", 0.2, 0.0, 0, 4, 10) - (0.2 - 0.2 / 9 * 3)) < 1e-7
    assert wdl("linear", 0.2, 0.0, 0, 1, 1) == f32(0.2)


def test_reader_loops_files_and_honours_start(tmp_path):
    rec = np.fromfile(GOLDEN, dtype=np.uint8).reshape(-1, 32)
    a, b = rec[:1000], rec[1000:1700]
    pa, pb = tmp_path / "a.bf", tmp_path / "b.bf"
    a.tofile(pa), b.tofile(pb)
    stream = np.concatenate([a, b])  # one epoch = 1700 records
    paths = (C.c_char_p * 2)(str(pa).encode(), str(pb).encode())
    for start in (0, 999, 1000, 1234, 1700 * 3 + 5):
        h = C.c_void_p()
        lib.check(L.bt_reader_open(paths, 2, start, C.byref(h)))
        pos = start % 1700
        for count in (300, 1700, 4321, 1):
            out = np.zeros((count, 32), dtype=np.uint8)
            lib.check(L.bt_reader_next(h, out.ctypes.data, count))
            exp = stream[(pos + np.arange(count)) % 1700]
            assert np.array_equal(out, exp), (start, count)
            pos = (pos + count) % 1700
        L.bt_reader_close(h)
    h = C.c_void_p()
    bad = (C.c_char_p * 1)(b"/nonexistent/file.bf")
    assert L.bt_reader_open(bad, 1, 0, C.byref(h)) == 4 and b"File not found" in L.bt_last_error()
