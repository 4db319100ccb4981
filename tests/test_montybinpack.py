"""MontyBinpackLoader restatement (bullet_amd/csrc/montybinpack.cpp; reference pipeline
crates/bullet_lib/src/value/loader/montybinpack.rs). The reference holds no montyformat file, so the wire format is
**parity unpinned**; what these tests pin is (a) the decoder and the move maker against an encoder written here,
independently, from the same published layout -- every position of the two hand-written games of
tests/test_viribinpack.py (en passant, both castlings, a promotion, captures, a double push) must equal the record the FEN
text of that position parses to, (b) the reference's in-tree pieces: buffer size rule, xorshift Fisher-Yates shuffle,
looping over the files, the filter contract `Fn(&Position, Move, i16, f32) -> bool` (true = keep)."""
import ctypes as C
import struct

import numpy as np
import pytest

from bullet_amd import lib
from test_viribinpack import EV1, EV2, GAME1, GAME2, expected_records, py_shuffle

L = lib.load()
CODE = {"P": 0b001, "N": 0b010, "B": 0b011, "R": 0b100, "Q": 0b101, "K": 0b110}


def squares(fen_board):
    out, rank, file = {}, 7, 0
    for ch in fen_board:
        if ch == "/":
            rank, file = rank - 1, 0
        elif ch.isdigit():
            file += int(ch)
        else:
            out[8 * rank + file] = ch
            file += 1
    return out


def monty_move(fen, frm, to, promo, flag):
    """the viriformat-style tuple of the shared game tables -> montyformat's from << 10 | to << 4 | flag"""
    sq = squares(fen.split()[0])
    capture = to in sq
    if flag == 1:
        mflag = 5                                  # en passant
    elif flag == 2:
        mflag = 2 if to > frm else 3               # castling: the tables give the rook's square, monty the king's target
        to = (frm & 56) | (6 if mflag == 2 else 2)
    elif flag == 3:
        mflag = 8 + promo + (4 if capture else 0)
    elif capture:
        mflag = 4
    elif sq[frm].upper() == "P" and abs(to - frm) == 16:
        mflag = 1
    else:
        mflag = 0
    return (frm << 10) | (to << 4) | mflag


def encode_game(game, result2, scores, fullmove=1):
    board, side = game[0][0].split()
    planes = [0, 0, 0, 0]
    for s, ch in squares(board).items():
        if ch.islower():
            planes[0] |= 1 << s
        c = CODE[ch.upper()]
        for bit, k in ((4, 1), (2, 2), (1, 3)):
            if c & bit:
                planes[k] |= 1 << s
    out = struct.pack("<4Q", *planes) + struct.pack("<BBBBH", int(side == "b"), 0, 0, 0, fullmove)
    out += bytes([7, 0, 7, 0]) + bytes([result2])
    for (fen, frm, to, promo, flag), sc in zip(game, scores):
        out += struct.pack("<Hh", monty_move(fen, frm, to, promo, flag), sc)
    return out + b"\0\0\0\0"


@pytest.fixture
def binpack(tmp_path):
    a, b = tmp_path / "a.mf", tmp_path / "b.mf"
    a.write_bytes(encode_game(GAME1, 2, EV1))
    b.write_bytes(encode_game(GAME2, 0, EV2) + b"\x01\x02\x03")  # trailing garbage: a truncated game is dropped
    exp = np.concatenate([expected_records(GAME1, 2, EV1), expected_records(GAME2, 0, EV2)])
    return [str(a), str(b)], exp


def open_reader(paths, mb, threads, seed, filt=None):
    h = C.c_void_p()
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    lib.check(L.bt_reader_open_montybinpack(arr, len(paths), mb, threads, filt, None, seed, C.byref(h)))
    return h


@pytest.mark.parametrize("threads", [1, 3])
def test_replay_matches_fen_positions_and_reference_shuffle(binpack, threads):
    paths, exp = binpack
    n_epoch = exp.shape[0]
    buf = 1 * 1024 * 1024 // 32 // 2  # buffer_size_mb = 1 (montybinpack.rs:35)
    h = open_reader(paths, 1, threads, 4242)
    got = np.zeros((2 * buf, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, got.ctypes.data, buf + 7))
    lib.check(L.bt_reader_next(h, got[buf + 7:].ctypes.data, buf - 7))
    L.bt_reader_close(h)
    stream = exp[np.arange(2 * buf) % n_epoch]  # files in order, round and round
    perm1, s = py_shuffle(buf, 4242)
    perm2, _ = py_shuffle(buf, s)
    assert np.array_equal(got[:buf], stream[:buf][perm1])
    assert np.array_equal(got[buf:], stream[buf:][perm2])


def test_filter_contract(binpack):
    paths, exp = binpack
    seen = []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint16, C.c_int16, C.c_float)
    def keep_quiet_moves_of_won_games(_u, rec, mv, score, result):
        seen.append((mv & 15, score, result))
        return int(result == 1.0 and (mv & 15) in (0, 1))  # true = keep (montybinpack.rs:169)

    h = open_reader(paths, 1, 1, 7, keep_quiet_moves_of_won_games)
    got = np.zeros((8192, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, got.ctypes.data, 8192))
    L.bt_reader_close(h)
    flags1 = [monty_move(*g) & 15 for g in GAME1]
    assert flags1 == [1, 1, 0, 1, 5, 4, 0, 0, 0, 0, 2, 2, 1]
    assert [monty_move(*g) & 15 for g in GAME2] == [3, 8, 4, 0]
    want = {exp[i].tobytes() for i, f in enumerate(flags1) if f in (0, 1)}
    assert {r.tobytes() for r in got} == want
    assert seen[:len(GAME1)] == [(f, e, 1.0) for f, e in zip(flags1, EV1)]
    assert seen[len(GAME1):len(GAME1) + 4] == [(f, e, 0.0) for f, e in zip([3, 8, 4, 0], EV2)]


def test_errors(tmp_path, binpack):
    h = C.c_void_p()
    bad = (C.c_char_p * 1)(b"/nonexistent.mf")
    assert L.bt_reader_open_montybinpack(bad, 1, 1, 1, None, None, 1, C.byref(h)) == 4
    reject_all = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint16, C.c_int16, C.c_float)(lambda *a: 0)
    h = open_reader(binpack[0], 1, 2, 1, reject_all)
    out = np.zeros((4, 32), dtype=np.uint8)
    assert L.bt_reader_next(h, out.ctypes.data, 4) != 0 and b"no position passes" in L.bt_last_error()
    L.bt_reader_close(h)
    for name, blob in (("code.mf", b"\xff" * 43 + b"\0\0\0\0"),                      # piece code 7 on every square
                       ("move.mf", encode_game(GAME1[:1], 1, [5])[:43] + struct.pack("<Hh", (30 << 10) | (38 << 4), 1) + b"\0" * 4),
                       ("rook.mf", encode_game([("4k3/8/8/8/8/8/8/4K3 w", 4, 7, 0, 2)], 1, [0]))):  # castling without a rook
        f = tmp_path / name
        f.write_bytes(blob)
        h = open_reader([str(f)], 1, 1, 1)
        assert L.bt_reader_next(h, out.ctypes.data, 4) != 0 and b"malformed" in L.bt_last_error(), name
        L.bt_reader_close(h)
