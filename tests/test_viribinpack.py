"""ViriBinpackLoader restatement (bullet_amd/csrc/viribinpack.cpp; reference pipeline
crates/bullet_lib/src/value/loader/viribinpack.rs). The reference holds no viriformat file, so the wire format is
**parity unpinned**; what these tests pin is (a) the board replay -- every position of two hand-written games
(en passant, both castlings written as king-takes-rook, a promotion, unmoved-rook codes) must equal the record the
FEN text of that position parses to, (b) the reference's in-tree pieces: buffer size rule, xorshift Fisher-Yates
shuffle (value/loader/rng.rs), looping over the files, the custom-filter contract, (c) the restated built-in `Filter`
(ViriFilter::Builtin, viribinpack.rs:178-187) rule by rule on the same games plus one-move games that put the side to
move in check by each kind of piece."""
import ctypes as C
import struct

import numpy as np
import pytest

from bullet_amd import lib

L = lib.load()

START = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR"
GAME1 = [  # (FEN board + side of the position, move from, to, promo, flag) ; flags: 1 ep, 2 castle, 3 promo
    (START + " w", 12, 28, 0, 0),                                              # e4
    ("rnbqkbnr/pppppppp/8/8/4P3/8/PPPP1PPP/RNBQKBNR b", 51, 35, 0, 0),         # d5
    ("rnbqkbnr/ppp1pppp/8/3p4/4P3/8/PPPP1PPP/RNBQKBNR w", 28, 36, 0, 0),       # e5
    ("rnbqkbnr/ppp1pppp/8/3pP3/8/8/PPPP1PPP/RNBQKBNR b", 53, 37, 0, 0),        # f5
    ("rnbqkbnr/ppp1p1pp/8/3pPp2/8/8/PPPP1PPP/RNBQKBNR w", 36, 45, 0, 1),       # exf6 e.p.
    ("rnbqkbnr/ppp1p1pp/5P2/3p4/8/8/PPPP1PPP/RNBQKBNR b", 62, 45, 0, 0),       # Nxf6
    ("rnbqkb1r/ppp1p1pp/5n2/3p4/8/8/PPPP1PPP/RNBQKBNR w", 6, 21, 0, 0),        # Nf3
    ("rnbqkb1r/ppp1p1pp/5n2/3p4/8/5N2/PPPP1PPP/RNBQKB1R b", 52, 44, 0, 0),     # e6
    ("rnbqkb1r/ppp3pp/4pn2/3p4/8/5N2/PPPP1PPP/RNBQKB1R w", 5, 12, 0, 0),       # Be2
    ("rnbqkb1r/ppp3pp/4pn2/3p4/8/5N2/PPPPBPPP/RNBQK2R b", 61, 52, 0, 0),       # Be7
    ("rnbqk2r/ppp1b1pp/4pn2/3p4/8/5N2/PPPPBPPP/RNBQK2R w", 4, 7, 0, 2),        # O-O  (king takes h1 rook)
    ("rnbqk2r/ppp1b1pp/4pn2/3p4/8/5N2/PPPPBPPP/RNBQ1RK1 b", 60, 63, 0, 2),     # O-O  (king takes h8 rook)
    ("rnbq1rk1/ppp1b1pp/4pn2/3p4/8/5N2/PPPPBPPP/RNBQ1RK1 w", 11, 27, 0, 0),    # d4
]
GAME2 = [
    ("r3k3/P7/8/8/8/8/8/4K3 w", 48, 56, 3, 3),                                 # axb.. a7xa8=Q? no: a7-a8 is blocked
]
# promotion with capture and a queenside castling, black to move first
GAME2 = [
    ("r3k3/1P6/8/8/8/8/8/4K3 b", 60, 56, 0, 2),                                # ...O-O-O (king takes a8 rook)
    ("2kr4/1P6/8/8/8/8/8/4K3 w", 49, 57, 0, 3),                                # b8=N+ (promotion piece 0 = knight)
    ("1Nkr4/8/8/8/8/8/8/4K3 b", 58, 57, 0, 0),                                 # Kxb8
    ("1k1r4/8/8/8/8/8/8/4K3 w", 4, 12, 0, 0),                                  # Ke2
]


def pack_board(fen_board, black_to_move, castling_rooks, eval_, wdl, fullmove=1):
    """marlinformat PackedBoard of a position given as FEN board text"""
    sq_piece = {}
    rank, file = 7, 0
    for ch in fen_board:
        if ch == "/":
            rank, file = rank - 1, 0
        elif ch.isdigit():
            file += int(ch)
        else:
            sq_piece[8 * rank + file] = ("PNBRQK".index(ch.upper())) | (0 if ch.isupper() else 8)
            file += 1
    occ, nibbles = 0, []
    for s in sorted(sq_piece):
        occ |= 1 << s
        code = sq_piece[s]
        if s in castling_rooks:
            code = (code & 8) | 6
        nibbles.append(code)
    nibbles += [0] * (32 - len(nibbles))
    pcs = bytes(nibbles[2 * i] | (nibbles[2 * i + 1] << 4) for i in range(16))
    return struct.pack("<Q", occ) + pcs + struct.pack("<BBHhBB", (0x80 if black_to_move else 0) | 64, 0, fullmove, eval_, wdl, 0)


def encode_game(game, castling_rooks, wdl, evals, fullmove=1):
    board, side = game[0][0].split()
    out = pack_board(board, side == "b", castling_rooks, 0, wdl, fullmove)
    for (_, frm, to, promo, flag), ev in zip(game, evals):
        out += struct.pack("<Hh", frm | (to << 6) | (promo << 12) | (flag << 14), ev)
    return out + b"\0\0\0\0"


def expected_records(game, wdl, evals):
    res = ["0.0", "0.5", "1.0"][wdl]
    out = np.zeros((len(game), 32), dtype=np.uint8)
    for i, ((fen, *_), ev) in enumerate(zip(game, evals)):
        lib.check(L.bt_chessboard_from_str(f"{fen} - - 0 1 | {ev} | {res}".encode(), out[i].ctypes.data))
    return out


EV1 = [10 * k - 30 for k in range(len(GAME1))]
EV2 = [-500, 77, 0, 3]


@pytest.fixture
def binpack(tmp_path):
    a, b = tmp_path / "a.vf", tmp_path / "b.vf"
    a.write_bytes(encode_game(GAME1, {0, 7, 56, 63}, 2, EV1))
    b.write_bytes(encode_game(GAME2, {56}, 0, EV2) + b"\x01\x02\x03")  # trailing garbage: a truncated game is dropped
    exp = np.concatenate([expected_records(GAME1, 2, EV1), expected_records(GAME2, 0, EV2)])
    return [str(a), str(b)], exp


def open_reader(paths, mb, threads, seed, filt=None):
    h = C.c_void_p()
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    lib.check(L.bt_reader_open_viribinpack(arr, len(paths), mb, threads, filt, None, seed, C.byref(h)))
    return h


def py_shuffle(n, seed):
    """value/loader/{viribinpack.rs:212-219, rng.rs:13-18}: the permutation applied to a buffer of n records"""
    perm, s, m = list(range(n)), seed, (1 << 64) - 1
    for i in range(n - 1, -1, -1):
        s ^= (s << 13) & m
        s ^= s >> 7
        s ^= (s << 17) & m
        j = s % (i + 1)
        perm[i], perm[j] = perm[j], perm[i]
    return perm, s


@pytest.mark.parametrize("threads", [1, 3])
def test_replay_matches_fen_positions_and_reference_shuffle(binpack, threads):
    paths, exp = binpack
    n_epoch = exp.shape[0]  # 13 + 4 positions per pass over the two files
    buf = 1 * 1024 * 1024 // 32 // 2  # buffer_size_mb = 1 (viribinpack.rs:50)
    h = open_reader(paths, 1, threads, 12345)
    got = np.zeros((2 * buf, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, got.ctypes.data, buf + 100))       # crosses into the second shuffle buffer
    lib.check(L.bt_reader_next(h, got[buf + 100:].ctypes.data, buf - 100))
    L.bt_reader_close(h)
    stream = exp[np.arange(2 * buf) % n_epoch]                        # files in order, round and round
    perm1, s = py_shuffle(buf, 12345)
    perm2, _ = py_shuffle(buf, s)                                     # the generator state carries over
    assert np.array_equal(got[:buf], stream[:buf][perm1])
    assert np.array_equal(got[buf:], stream[buf:][perm2])


def test_custom_filter_contract(binpack):
    paths, exp = binpack
    seen = []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint16, C.c_int16, C.c_float, C.c_int)
    def keep_quiet_white_wins(_u, rec, mv, ev, wdl, ply):
        seen.append((mv >> 14, ev, wdl, ply))
        return int(wdl == 1.0 and (mv >> 14) == 0 and ply >= 2)  # true = keep (viribinpack.rs:188-206)

    h = open_reader(paths, 1, 1, 7, keep_quiet_white_wins)
    got = np.zeros((16384, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, got.ctypes.data, 16384))
    L.bt_reader_close(h)
    keep = [i for i, g in enumerate(GAME1) if g[4] == 0 and i >= 2]
    want = {exp[i].tobytes() for i in keep}
    assert {r.tobytes() for r in got} == want
    assert seen[:len(GAME1)] == [(g[4], e, 1.0, i) for i, (g, e) in enumerate(zip(GAME1, EV1))]
    assert seen[len(GAME1):len(GAME1) + 4] == [(g[4], e, 0.0, i) for i, (g, e) in enumerate(zip(GAME2, EV2))]


def test_errors(tmp_path, binpack):
    h = C.c_void_p()
    bad = (C.c_char_p * 1)(b"/nonexistent.vf")
    assert L.bt_reader_open_viribinpack(bad, 1, 1, 1, None, None, 1, C.byref(h)) == 4
    reject_all = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint16, C.c_int16, C.c_float, C.c_int)(lambda *a: 0)
    h = open_reader(binpack[0], 1, 2, 1, reject_all)
    out = np.zeros((4, 32), dtype=np.uint8)
    assert L.bt_reader_next(h, out.ctypes.data, 4) != 0 and b"no position passes" in L.bt_last_error()
    L.bt_reader_close(h)
    junk = tmp_path / "junk.vf"
    junk.write_bytes(bytes(range(1, 41)) + b"\0\0\0\0")  # a "game" whose first move starts on an empty square
    h = open_reader([str(junk)], 1, 1, 1)
    assert L.bt_reader_next(h, out.ctypes.data, 4) != 0 and b"malformed" in L.bt_last_error()
    L.bt_reader_close(h)


# ---- ViriFilter::Builtin: the crate's `Filter`, restated (parity unpinned; see csrc/viribinpack.cpp)
CHECKS = [  # one-move games, black to move: (FEN board, in check?)
    ("4k3/8/8/8/8/8/4R3/4K3 b", True),    # rook on the open file
    ("4k3/4p3/8/8/8/8/4R3/4K3 b", False),  # the same file, blocked
    ("4k3/3P4/8/8/8/8/8/4K3 b", True),    # pawn
    ("4k3/4P3/8/8/8/8/8/4K3 b", False),   # a pawn straight ahead attacks nothing
    ("4k3/8/3N4/8/8/8/8/4K3 b", True),    # knight
    ("4k3/8/8/1B6/8/8/8/4K3 b", True),    # bishop on the long way round
    ("4k3/8/8/1R6/8/8/8/4K3 b", False),   # a rook on that diagonal does not
    ("4k3/8/8/8/Q7/8/8/4K3 b", True),     # queen, diagonal
    ("3Kk3/8/8/8/8/8/8/8 b", True),       # (kings side by side: the walk sees the king too)
]


def builtin_reader(paths, filt, threads=2, seed=5):
    h = C.c_void_p()
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    rc = L.bt_reader_open_viribinpack_builtin(arr, len(paths), 1, threads, C.byref(filt), seed, C.byref(h))
    return rc, h


def kept_set(paths, filt, n=4096):
    rc, h = builtin_reader(paths, filt)
    lib.check(rc)
    got = np.zeros((n, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, got.ctypes.data, n))
    L.bt_reader_close(h)
    return {r.tobytes() for r in got}


def test_builtin_filter_rules(binpack, tmp_path):
    paths, exp = binpack
    n1 = len(GAME1)
    rec = lambda idx: {exp[i].tobytes() for i in idx}
    f = lib.ViriBuiltinFilter()
    L.bt_viri_filter_unrestricted(C.byref(f))
    assert kept_set(paths, f) == rec(range(len(exp)))  # Filter::UNRESTRICTED drops nothing
    L.bt_viri_filter_default(C.byref(f))
    assert (f.min_ply, f.min_pieces, f.max_eval, f.filter_tactical, f.filter_check, f.filter_castling) == (16, 4, 20000, 1, 1, 0)
    assert f.max_eval_incorrectness == 0xFFFFFFFF and not f.random_fen_skipping
    # GAME1 starts at fullmove 1, white to move: ply = index; GAME2 at fullmove 1, black to move: ply = 1 + index
    f.min_ply = 3
    quiet1 = [i for i, g in enumerate(GAME1) if i >= 3 and g[4] in (0, 2) and i != 5]  # 4: e.p., 5: Nxf6 -- tactical
    # GAME2: 0 castles (kept, ply 1 < 3 though), 1 promotes, 2 Kxb8 captures, 3 has three pieces
    assert kept_set(paths, f) == rec(quiet1)
    f.min_ply = 0
    assert kept_set(paths, f) == rec([i for i, g in enumerate(GAME1) if g[4] in (0, 2) and i != 5] + [n1])
    f.filter_castling = 1
    assert kept_set(paths, f) == rec([i for i, g in enumerate(GAME1) if g[4] == 0 and i != 5])
    f.filter_castling, f.filter_tactical, f.min_pieces = 0, 0, 3
    assert kept_set(paths, f) == rec(range(len(exp)))
    f.min_pieces = 32  # only the positions before the first capture (the e.p. at index 4) have all 32 pieces
    assert kept_set(paths, f) == rec(range(5))
    # evals: EV1 = -30 ... 90 (white wins), EV2 = -500, 77, 0, 3 (black wins)
    f.min_pieces, f.max_eval = 0, 77
    assert kept_set(paths, f) == rec([i for i, e in enumerate(EV1) if abs(e) < 77] + [n1 + 2, n1 + 3])
    f.max_eval, f.max_eval_incorrectness = 20000, 20
    wrong1 = [i for i, e in enumerate(EV1) if e < -20]          # white won: a negative eval is wrong by |eval|
    wrong2 = [n1 + i for i, e in enumerate(EV2) if e > 20]      # black won: a positive one
    assert kept_set(paths, f) == rec(set(range(len(exp))) - set(wrong1) - set(wrong2))
    # checks: nine one-move games (the king steps aside; three or four pieces on the board)
    L.bt_viri_filter_unrestricted(C.byref(f))
    f.filter_check = 1
    games, want = b"", set()
    for k, (fen, in_check) in enumerate(CHECKS):
        g = [(fen, 60, 59 if k != 8 else 61, 0, 0)]
        games += encode_game(g, set(), 1, [k])
        if not in_check:
            want.add(expected_records(g, 1, [k])[0].tobytes())
    cf = tmp_path / "checks.vf"
    cf.write_bytes(games)
    assert kept_set([str(cf)], f) == want and len(want) == 3
    # ply counts on from the header's fullmove
    L.bt_viri_filter_unrestricted(C.byref(f))
    f.min_ply = 21
    late = tmp_path / "late.vf"
    late.write_bytes(encode_game(GAME1, {0, 7, 56, 63}, 2, EV1, fullmove=10))  # plies 18 ... 30
    assert kept_set([str(late)], f) == rec(range(3, n1))


def test_builtin_filter_random_skipping(binpack):
    paths, exp = binpack
    f = lib.ViriBuiltinFilter()
    L.bt_viri_filter_unrestricted(C.byref(f))
    f.random_fen_skipping, f.random_fen_skip_probability = 1, 1.0
    rc, h = builtin_reader(paths, f)
    lib.check(rc)
    out = np.zeros((4, 32), dtype=np.uint8)
    assert L.bt_reader_next(h, out.ctypes.data, 4) != 0 and b"no position passes" in L.bt_last_error()
    L.bt_reader_close(h)
    f.random_fen_skip_probability = 0.5
    rc, h = builtin_reader(paths, f, threads=1)
    lib.check(rc)
    n = 16384  # one shuffle buffer: whole epochs of 17 positions, about half of each kept
    got = np.zeros((n, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, got.ctypes.data, n))
    L.bt_reader_close(h)
    first = exp[0].tobytes()
    share = sum(r.tobytes() == first for r in got) / (n / len(exp))
    assert 0.8 < share < 1.2  # every position is kept with probability 1/2, so each is ~1/17 of the stream
    f.random_fen_skip_probability = 1.5
    rc, h = builtin_reader(paths, f)
    assert rc != 0 and b"outside [0, 1]" in L.bt_last_error()
