"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol that
include/bullet_amd.h declares, reports errors the documented way, and its host batch-preparation
path (section 4) is bit-exact against the oracle on the reference's own fixture."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_binding as ob
from bullet_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "batch.bf")
MIRROR_LAYOUT = [0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9]


def test_library_exports_every_declared_symbol():
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "bullet_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_error_convention_without_compute():
    L = lib.load()
    d = lib.FtDesc(batch=4, rows=8, cols=16, stride=4, offset=0, nnz=4, act=0, pairwise=0, n_inputs=1, out_ld=8)
    assert L.bt_ft_backward_workspace_bytes(C.byref(d)) == 0  # rows + offset > stride (sparse.rs:30-32)
    assert b"8 + 0 > 4" in L.bt_last_error()
    with pytest.raises(lib.BtError):
        lib.make_input_desc("chess_buckets", layout=[0] * 10)


def test_net_tensor_layout_matches_oracle():
    L = lib.load()
    for cfg in [dict(num_inputs=768, nnz=32, ft_out=1024, ft_act="screlu", dense=[(1, "identity", True)], num_buckets=8),
                dict(num_inputs=12288, nnz=32, ft_out=1024, ft_act="crelu", pairwise=True, num_buckets=8,
                     dense=[(16, "screlu", True), (32, "screlu", True), (1, "identity", True)]),
                dict(num_inputs=147, nnz=49, ft_out=256, ft_act="screlu", dense=[(1, "identity", False)])]:
        dp, do = lib.make_net_desc(**cfg), ob.make_net(**cfg)
        sizes = [L.bt_net_tensor_size(C.byref(dp), t) for t in range(L.bt_net_num_tensors(C.byref(dp)))]
        assert sizes == ob.tensor_sizes(do)
    assert sum(ob.tensor_sizes(ob.make_net(768, 32, 1024, "screlu", [(1, "identity", True)], num_buckets=8))) == 803848


@pytest.mark.parametrize("kind,layout,fact", [
    ("chess768", None, False), ("chess768", None, True),
    ("chess_buckets", [(i * 7) % 16 for i in range(64)], False),
    ("chess_buckets_mirrored", MIRROR_LAYOUT, False), ("chess_buckets_mirrored", MIRROR_LAYOUT, True),
])
def test_host_mapper_bit_exact_on_reference_fixture(kind, layout, fact):
    rec = np.fromfile(GOLDEN, dtype=np.uint8)
    ip_, io_ = lib.make_input_desc(kind, fact, layout), ob.input_desc(kind, fact, layout)
    for kw_p, kw_o in [(dict(out_buckets=8, blend=0.2, threads=4), dict(n_out_buckets=8, blend=0.2)),
                       (dict(out_buckets=1, blend=1.0, win_rate_model=True, threads=3),
                        dict(n_out_buckets=1, blend=1.0, win_rate_model=True)),
                       (dict(out_buckets=4, wdl=True, threads=1), dict(n_out_buckets=4, wdl=True))]:
        a, b = lib.map_batch(ip_, rec, **kw_p), ob.map_batch(io_, rec, **kw_o)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)  # bit-exact, floats included
    stm, ntm = a[0], a[1]
    F = lib.load().bt_input_num_inputs(C.byref(ip_))
    assert stm.max() < F and ntm.max() < F and stm.min() >= -1


def test_fixture_invariants_chess768():
    """Source-derived invariants on all 16 273 fixture records."""
    rec = np.fromfile(GOLDEN, dtype=np.uint8)
    stm, ntm, bk, tg, w = lib.map_batch(lib.make_input_desc("chess768"), rec, out_buckets=8, blend=0.0)
    occ = rec.reshape(-1, 32)[:, :8].copy().view(np.uint64).reshape(-1)
    pop = np.array([bin(int(o)).count("1") for o in occ])
    has_term = (stm < 0).any(1)
    cnt = np.where(has_term, (stm < 0).argmax(1), 32)  # slots after the -1 terminator are untouched
    assert np.array_equal(cnt, pop) and np.array_equal(cnt, np.where((ntm < 0).any(1), (ntm < 0).argmax(1), 32))
    assert np.array_equal(bk, (pop - 2) // 4)
    valid = np.arange(32)[None, :] < cnt[:, None]
    # stm/ntm symmetry of chess768.rs:26-27: colour swapped, square ^ 56
    s, n = stm[valid], ntm[valid]
    assert np.array_equal(n, (1 - s // 384) * 384 + (s % 384) // 64 * 64 + ((s % 64) ^ 56))
    assert np.all((tg > 0) & (tg < 1)) and np.all(w == 1.0)


def test_ataxx_mapper_bit_exact_synthetic():
    rng = np.random.default_rng(5)
    n = 2000
    rec = np.zeros((n, 32), dtype=np.uint8)
    for i in range(n):
        sq = rng.permutation(49)
        a, b, c = rng.integers(1, 20), rng.integers(1, 20), rng.integers(0, 6)
        bbs = [sum(1 << int(s) for s in sq[:a]), sum(1 << int(s) for s in sq[a:a + b]),
               sum(1 << int(s) for s in sq[a + b:a + b + c])]
        rec[i, :24] = np.array(bbs, dtype=np.uint64).view(np.uint8)
        rec[i, 24:26] = np.array([rng.integers(-2000, 2000)], dtype=np.int16).view(np.uint8)
        rec[i, 26] = rng.integers(0, 3)
    rec = rec.reshape(-1)
    for kind in ("ataxx147", "ataxx98"):
        a = lib.map_batch(lib.make_input_desc(kind), rec, blend=0.5, threads=2)
        b = ob.map_batch(ob.input_desc(kind), rec, blend=0.5)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def _startpos_record():
    rec = np.zeros(32, dtype=np.uint8)
    lib.check(lib.load().bt_chessboard_from_str(b"rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1 | 10 | 0.5",
                                                rec.ctypes.data))
    return rec


@pytest.mark.parametrize("what,reason", [
    ("too_many", "More inputs provided"), ("piece_type", "Input feature index exceeded"), ("ksq", "king square"),
    ("result", "game result"), ("too_few", "fewer than two pieces"),
])
def test_host_mapper_rejects_corrupt_records(what, reason):
    """What the reference's mapper panics on (value.rs:80,91,101, chess_buckets.rs:30, outputs.rs:29) fails the call
    with the index of the first bad record; nothing is written past the record's own slots."""
    good = _startpos_record()
    bad = good.copy()
    kind, layout, kw = "chess768", None, dict(out_buckets=8)
    if what == "too_many":
        bad[0:8] = 0xFF  # 64 occupied squares
    elif what == "piece_type":
        bad[8] = (bad[8] & 0xF0) | 6  # first piece: type 6 does not exist
    elif what == "ksq":
        kind, layout = "chess_buckets", [0] * 64
        bad[27] = 200
    elif what == "result":
        bad[26] = 7
        kw = dict(out_buckets=8, wdl=True)
    elif what == "too_few":
        bad[0:8] = 0
        bad[0] = 1  # one piece
    recs = np.concatenate([good, good, bad, good])
    idesc = lib.make_input_desc(kind, False, layout)
    nnz = 32
    guard = 1234567
    out = (np.full((5, nnz), guard, np.int32), np.full((5, nnz), guard, np.int32), np.full(5, guard, np.int32),
           np.full(5 * 3, 9.0, np.float32), np.full(5, 9.0, np.float32))
    with pytest.raises(lib.BtError, match="record 2: " + reason):
        lib.map_batch(idesc, recs, threads=2, out=out, **kw)
    assert (out[0][4] == guard).all() and (out[1][4] == guard).all() and out[2][4] == guard  # row past the batch untouched
    assert out[0][2, 0] == -1 and out[4][2] == 0.0  # the rejected record is an empty list with zero weight
    lib.map_batch(idesc, np.concatenate([good] * 4), threads=2, out=out, **kw)  # and the good ones still map
