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
            # a NULL destination skips records (across file ends too): how a data-parallel rank passes foreign shards
            skip = (count * 7) % 2500
            lib.check(L.bt_reader_next(h, None, skip))
            pos = (pos + skip) % 1700
        L.bt_reader_close(h)
    h = C.c_void_p()
    bad = (C.c_char_p * 1)(b"/nonexistent/file.bf")
    assert L.bt_reader_open(bad, 1, 0, C.byref(h)) == 4 and b"File not found" in L.bt_last_error()


# ------------------------------------------------------------------------------------------ text positions
def _fen_of(record, as_black):
    """Renders a fixture record (side-to-move relative) as `<fen> | score | result` text, either with white to move
    (the record as it stands) or as the same position seen with black to move (ranks mirrored, colours swapped,
    score and result negated) -- parsing either must give the record back."""
    occ = int(record[:8].copy().view(np.uint64)[0])
    board, n = {}, 0
    for sq in range(64):
        if occ >> sq & 1:
            pc = (int(record[8 + n // 2]) >> (4 * (n & 1))) & 15
            n += 1
            colour, kind = pc >> 3, pc & 7
            if as_black:
                board[sq ^ 56] = "pnbrqk"[kind] if colour == 0 else "PNBRQK"[kind]
            else:
                board[sq] = "PNBRQK"[kind] if colour == 0 else "pnbrqk"[kind]
    rows = []
    for rank in range(7, -1, -1):
        row, gap = "", 0
        for file in range(8):
            ch = board.get(8 * rank + file)
            if ch is None:
                gap += 1
            else:
                row += (str(gap) if gap else "") + ch
                gap = 0
        rows.append(row + (str(gap) if gap else ""))
    score = int(record[24:26].copy().view(np.int16)[0])
    res = int(record[26])
    if as_black:
        score, res = -score, 2 - res
    return "%s %s - - 0 1 | %d | %s" % ("/".join(rows), "b" if as_black else "w", score, ["0.0", "0.5", "1.0"][res])


def test_chessboard_from_str_round_trips_the_reference_fixture():
    """bulletformat's FromStr is not in the reference tree (parity unpinned, see the C source); what IS pinned is the
    record layout of examples/tests/batch.bf, so: every fixture record, written out as text from either side's
    point of view, must parse back to the identical 32 bytes."""
    rec = np.fromfile(GOLDEN, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros(32, dtype=np.uint8)
    for i in range(0, rec.shape[0], 7):
        for as_black in (False, True):
            lib.check(L.bt_chessboard_from_str(_fen_of(rec[i], as_black).encode(), out.ctypes.data))
            assert np.array_equal(out, rec[i]), (i, as_black)


def test_chessboard_from_str_known_position_and_errors():
    out = np.zeros(32, dtype=np.uint8)
    lib.check(L.bt_chessboard_from_str(b"rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1 | 37 | [0.5]",
                                       out.ctypes.data))
    assert int(out[:8].copy().view(np.uint64)[0]) == 0xFFFF00000000FFFF
    assert list(out[8:12]) == [0x13, 0x42, 0x25, 0x31]           # R N B Q K B N R, low nibble first
    assert list(out[12:16]) == [0x00] * 4 and list(out[16:20]) == [0x88] * 4
    assert list(out[20:24]) == [0x9B, 0xCA, 0xAD, 0xB9]
    assert int(out[24:26].copy().view(np.int16)[0]) == 37 and out[26] == 1 and out[27] == 4 and out[28] == 60 ^ 56
    # black to move: score and result flip, the board is seen from black's side (same bytes for the start position)
    white = out.copy()
    lib.check(L.bt_chessboard_from_str(b"rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR b KQkq - 0 1 | -37 | 0.5",
                                       out.ctypes.data))
    assert np.array_equal(out, white)
    for bad in (b"8/8/8/8/8/8/8/8 w - - 0 1 | 0 | 0.5", b"k7/8/8/8/8/8/8/K7 w - - 0 1 | 0", b"k7/8/8/8/8/8/8/K7 w | x | 1.0",
                b"k7/8/8/8/8/8/8/K7 w | 0 | 0.7"):
        assert L.bt_chessboard_from_str(bad, out.ctypes.data) == 1 and L.bt_last_error()


def test_text_reader_serves_the_lines_round_and_round(tmp_path):
    rec = np.fromfile(GOLDEN, dtype=np.uint8).reshape(-1, 32)[:53]
    path = tmp_path / "positions.txt"
    path.write_text("".join(_fen_of(r, i % 2 == 1) + ("\r\n" if i % 3 == 0 else "\n") for i, r in enumerate(rec)))
    h = C.c_void_p()
    paths = (C.c_char_p * 1)(b"text:" + str(path).encode())
    lib.check(L.bt_reader_open(paths, 1, 1000, C.byref(h)))  # start is ignored by the text loader (text.rs:25)
    out = np.zeros((130, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, out.ctypes.data, 130))
    assert np.array_equal(out, rec[np.arange(130) % 53])
    lib.check(L.bt_reader_next(h, out.ctypes.data, 10))
    assert np.array_equal(out[:10], rec[(130 + np.arange(10)) % 53])
    lib.check(L.bt_reader_next(h, None, 60))  # skipped
    lib.check(L.bt_reader_next(h, out.ctypes.data, 10))
    assert np.array_equal(out[:10], rec[(200 + np.arange(10)) % 53])
    L.bt_reader_close(h)
    path.write_text("not a position\n")
    assert L.bt_reader_open(paths, 1, 0, C.byref(h)) != 0 and b"line 1" in L.bt_last_error()


def test_loader_throughput_runs_the_schedule_without_a_device():
    cfg = lib.TrainConfig(4096, 3, 1, 2, lib.LrSched(lib.LR["constant"], 0.001, 0, 0, 0),
                          lib.WdlSched(lib.WDL["constant"], 0.5, 0, 0), 400.0, 8, 0, 2, 0, 0, 0)
    idesc = lib.make_input_desc("chess768")
    pps = C.c_double(0)
    paths = (C.c_char_p * 1)(GOLDEN.encode())
    lib.check(L.bt_measure_loader_throughput(C.byref(idesc), C.byref(cfg), paths, 1, C.byref(pps)))
    assert pps.value > 1e4


def test_mapper_closures_wdl_adjust_and_datapoint_weight():
    """value/builder.rs:102-111 + value.rs:94,113-115: per-position blend adjustment and entry weights"""
    rec = np.fromfile(GOLDEN, dtype=np.uint8)[:32 * 300]
    idesc = lib.make_input_desc("chess768")
    base = lib.map_batch(idesc, rec, out_buckets=8, blend=0.4, threads=1)
    BLEND = C.CFUNCTYPE(C.c_float, C.c_void_p, C.c_void_p, C.c_float)
    WEIGHT = C.CFUNCTYPE(C.c_float, C.c_void_p, C.c_void_p)
    pieces = lambda p: bin(int.from_bytes(C.string_at(p, 8), "little")).count("1")  # noqa: E731
    blend_fn = BLEND(lambda _u, p, b: b if pieces(p) > 16 else 1.0)                 # endgames: pure game result
    weight_fn = WEIGHT(lambda _u, p: pieces(p) / 32.0)
    o = lib.MapOpts(8, 0.4, 400.0, 0, 0, 3, C.cast(blend_fn, C.c_void_p), C.cast(weight_fn, C.c_void_p), None)
    out = [np.zeros_like(a) for a in base]
    lib.check(L.bt_map_batch(C.byref(idesc), C.byref(o), rec.ctypes.data, 300, *[a.ctypes.data for a in out]))
    r = rec.reshape(-1, 32)
    n_pieces = np.array([bin(int(x)).count("1") for x in r[:, :8].copy().view(np.uint64).reshape(-1)])
    assert np.array_equal(out[0], base[0]) and np.array_equal(out[2], base[2])
    assert np.array_equal(out[4], (n_pieces / 32.0).astype(np.float32))
    end = n_pieces <= 16
    assert np.array_equal(out[3][~end], base[3][~end])
    assert np.array_equal(out[3][end], (r[end, 26] / 2.0).astype(np.float32)) and end.any() and (~end).any()
    bad = BLEND(lambda _u, p, b: 1.5)
    o = lib.MapOpts(8, 0.4, 400.0, 0, 0, 2, C.cast(bad, C.c_void_p), None, None)
    assert L.bt_map_batch(C.byref(idesc), C.byref(o), rec.ctypes.data, 300, *[a.ctypes.data for a in out]) == 1
    assert b"WDL proportion" in L.bt_last_error()
