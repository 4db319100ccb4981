"""SfBinpackLoader restatement (bullet_amd/csrc/sfbinpack.cpp; reference pipeline
crates/bullet_lib/src/value/loader/sfbinpack.rs). The binpack wire format lives in the un-vendored crate
`sfbinpack 0.6.4` and the reference holds no binpack file: **parity unpinned**. What these tests do pin: an
independent Python encoder of the published format (own move generation, own bit writer) produces chains for two
hand-written games -- en passant (also as a stem with the en-passant nibble), both castlings, a promotion, a king
capture-step -- and every decoded position must equal the record its FEN text parses to, with the right
side-to-move-relative score and result; plus the reference's in-tree pieces (shuffle, filter contract, errors)."""
import ctypes as C
import struct

import numpy as np
import pytest

import test_viribinpack as tv
from bullet_amd import lib

L = lib.load()
P, N, B, R, Q, K = range(6)


def parse(fen):
    board, side = fen.split()[:2]
    sq, rank, file = {}, 7, 0
    for ch in board:
        if ch == "/":
            rank, file = rank - 1, 0
        elif ch.isdigit():
            file += int(ch)
        else:
            sq[8 * rank + file] = (0 if ch.isupper() else 1, "PNBRQK".index(ch.upper()))
            file += 1
    return sq, (1 if side == "b" else 0)


def ray(sq, s, dirs):
    out = []
    for df, dr in dirs:
        f, r = (s & 7) + df, (s >> 3) + dr
        while 0 <= f < 8 and 0 <= r < 8:
            out.append(8 * r + f)
            if 8 * r + f in sq:
                break
            f, r = f + df, r + dr
    return out


def steps(s, deltas):
    return [8 * ((s >> 3) + dr) + (s & 7) + df for df, dr in deltas if 0 <= (s & 7) + df < 8 and 0 <= (s >> 3) + dr < 8]


DIAG, ORTH = [(1, 1), (1, -1), (-1, 1), (-1, -1)], [(1, 0), (-1, 0), (0, 1), (0, -1)]
KN = [(1, 2), (2, 1), (2, -1), (1, -2), (-1, -2), (-2, -1), (-2, 1), (-1, 2)]


def nbits(n):
    return 0 if n <= 1 else (n - 1).bit_length()


class Bits:
    def __init__(self):
        self.bits = []

    def put(self, value, count):  # most significant bit first
        self.bits += [(value >> (count - 1 - i)) & 1 for i in range(count)]

    def vle(self, v, block=4):
        while True:
            self.put((v & 15) | ((1 if v > 15 else 0) << block), block + 1)
            v >>= block
            if v == 0:
                break

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def zz(v):
    return ((v << 1) ^ (v >> 15)) & 0xFFFF


def encode_move(bits, sq, stm, castle, ep, frm, to, promo, flag):
    """movetext encoding of one move (flag as in the viri test: 1 ep, 2 castle = king takes rook, 3 promotion)"""
    ours = sorted(s for s, (c, _) in sq.items() if c == stm)
    bits.put(ours.index(frm), nbits(len(ours)))
    t = sq[frm][1]
    own = set(ours)
    if t == P:
        up = -1 if stm else 1
        dest = {d for d in steps(frm, [(-1, up), (1, up)]) if (d in sq and sq[d][0] != stm) or d == ep}
        f1 = frm + 8 * up
        if f1 not in sq:
            dest.add(f1)
            if (frm >> 3) == (6 if stm else 1) and f1 + 8 * up not in sq:
                dest.add(f1 + 8 * up)
        dest = sorted(dest)
        if (frm >> 3) == (1 if stm else 6):
            bits.put(dest.index(to) * 4 + promo, nbits(4 * len(dest)))
        else:
            bits.put(dest.index(to), nbits(len(dest)))
    elif t == K:
        att = sorted(set(steps(frm, DIAG + ORTH)) - own)
        rights = [r for r in (("long", "short")) if (stm, r) in castle]
        if flag == 2:
            kind = "short" if to > frm else "long"
            bits.put(len(att) + rights.index(kind), nbits(len(att) + len(rights)))
        else:
            bits.put(att.index(to), nbits(len(att) + len(rights)))
    else:
        cand = steps(frm, KN) if t == N else ray(sq, frm, (DIAG if t in (B, Q) else []) + (ORTH if t in (R, Q) else []))
        att = sorted(set(cand) - own)
        bits.put(att.index(to), nbits(len(att)))


def stem(sq, stm, castle, ep, move, score, ply, result):
    occ, nib = 0, []
    for s in sorted(sq):
        c, t = sq[s]
        occ |= 1 << s
        code = 2 * t + c
        if t == P and ep is not None and s == (ep + 8 if c == 0 else ep - 8):
            code = 12
        if t == R and ((c, "short") in castle and s == (63 if c else 7) or (c, "long") in castle and s == (56 if c else 0)):
            code = 13 + c
        if t == K and c == 1 and stm == 1:
            code = 15
        nib.append(code)
    nib += [0] * (32 - len(nib))
    frm, to, promo, flag = move
    mtype = {0: 0, 3: 1, 2: 2, 1: 3}[flag]
    return (struct.pack(">Q", occ) + bytes(nib[2 * i] | (nib[2 * i + 1] << 4) for i in range(16)) +
            struct.pack(">HHHH", (mtype << 14) | (frm << 8) | (to << 2) | promo, zz(score),
                        ply | (zz(result) << 14), 0))


def chain(game, states, scores_stm, ply0, result_first):
    """one stem + movetext for `game` (list of (fen, from, to, promo, flag)); states[i] = (castle, ep) of position i"""
    sq, stm = parse(game[0][0])
    out = stem(sq, stm, states[0][0], states[0][1], game[0][1:], scores_stm[0], ply0, result_first)
    out += struct.pack(">H", len(game) - 1)
    bits = Bits()
    for i in range(1, len(game)):
        sq, stm = parse(game[i][0])
        encode_move(bits, sq, stm, states[i][0], states[i][1], *game[i][1:])
        bits.vle(zz(scores_stm[i] + scores_stm[i - 1]))  # score_i = -score_(i-1) + delta
    return out + (bits.bytes() if len(game) > 1 else b"")


ALL = {(0, "short"), (0, "long"), (1, "short"), (1, "long")}
STATES1 = [(ALL, None)] * 4 + [(ALL, 45)] + [(ALL, None)] * 6 + [({(1, "short"), (1, "long")}, None), (set(), None)]
STATES2 = [({(1, "long")}, None), (set(), None), (set(), None), (set(), None)]


def rel(game, white_scores):
    return [w if parse(g[0])[1] == 0 else -w for g, w in zip(game, white_scores)]


def build_file(split_game1):
    s1, s2 = rel(tv.GAME1, tv.EV1), rel(tv.GAME2, tv.EV2)
    res1 = lambda i: 1 if parse(tv.GAME1[i][0])[1] == 0 else -1   # white wins game 1  # noqa: E731
    res2 = lambda i: -1 if parse(tv.GAME2[i][0])[1] == 0 else 1   # black wins game 2  # noqa: E731
    if split_game1:  # the second chain starts on the position with the en-passant square: nibble 12 in its stem
        body = chain(tv.GAME1[:4], STATES1[:4], s1[:4], 0, res1(0)) + chain(tv.GAME1[4:], STATES1[4:], s1[4:], 4, res1(4))
    else:
        body = chain(tv.GAME1, STATES1, s1, 0, res1(0))
    body2 = chain(tv.GAME2, STATES2, s2, 30, res2(0))
    solo = chain(tv.GAME2[3:], STATES2[3:], s2[3:], 33, res2(3))  # a stem with no movetext
    chunk1, chunk2 = body + body2, solo
    return b"BINP" + struct.pack("<I", len(chunk1)) + chunk1 + b"BINP" + struct.pack("<I", len(chunk2)) + chunk2


def expected():
    e1, e2 = tv.expected_records(tv.GAME1, 2, tv.EV1), tv.expected_records(tv.GAME2, 0, tv.EV2)
    return np.concatenate([e1, e2, e2[3:]])


def open_reader(paths, mb, threads, seed, filt=None):
    h = C.c_void_p()
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    lib.check(L.bt_reader_open_sfbinpack(arr, len(paths), mb, threads, filt, None, seed, C.byref(h)))
    return h


@pytest.mark.parametrize("split,threads", [(False, 1), (True, 2)])
def test_decoded_chains_match_fen_positions_and_reference_shuffle(tmp_path, split, threads):
    path = tmp_path / "games.binpack"
    path.write_bytes(build_file(split))
    exp = expected()
    n_epoch, buf = exp.shape[0], 16384
    h = open_reader([str(path)], 1, threads, 4242)
    got = np.zeros((buf, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, got.ctypes.data, buf))
    L.bt_reader_close(h)
    stream = exp[np.arange(buf + n_epoch) % n_epoch]
    perm, _ = tv.py_shuffle(buf, 4242)
    # (the decoder works through whole files: what it holds beyond the buffer is carried over, not dropped)
    assert np.array_equal(got, stream[:buf][perm])


class SfEntry(C.Structure):
    _fields_ = [("chessboard", C.c_void_p), ("score", C.c_int16), ("ply", C.c_uint16), ("result", C.c_int16),
                ("move_from", C.c_uint8), ("move_to", C.c_uint8), ("move_type", C.c_uint8), ("promotion", C.c_uint8),
                ("captured_piece", C.c_uint8), ("in_check", C.c_uint8)]


def test_filter_sees_the_training_data_entry(tmp_path):
    path = tmp_path / "games.binpack"
    path.write_bytes(build_file(False))
    seen = []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(SfEntry))
    def filt(_u, e):
        e = e.contents
        seen.append((e.score, e.ply, e.result, e.move_from, e.move_to, e.move_type, e.promotion, e.captured_piece, e.in_check))
        # the filter of examples/simple.rs:96-102 (with a lower ply bound)
        return int(e.ply >= 2 and not e.in_check and abs(e.score) <= 10000 and e.move_type == 0 and e.captured_piece == 7)

    h = open_reader([str(path)], 1, 1, 5, filt)
    got = np.zeros((16384, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, got.ctypes.data, 16384))
    L.bt_reader_close(h)
    s1 = rel(tv.GAME1, tv.EV1)
    mt = {0: 0, 3: 1, 2: 2, 1: 3}
    for i, g in enumerate(tv.GAME1):
        sq, stm = parse(g[0])
        cap = sq[g[2]][1] if g[2] in sq and g[4] != 2 else 7
        assert seen[i] == (s1[i], i, 1 if stm == 0 else -1, g[1], g[2], mt[g[4]], g[3], cap, 0), i
    # game 2: "1Nkr4/8/8/8/8/8/8/4K3 b" -- the black king on c8 is in check from the new knight on b8? no: a knight on
    # b8 attacks a6, c6, d7; the position before it, "2kr4/1P6/... w", has no check either; ply counts on from 30
    assert [s[1] for s in seen[13:17]] == [30, 31, 32, 33] and seen[15][7] == N and seen[14][5:7] == (1, 0)
    exp = expected()
    keep = [i for i, g in enumerate(tv.GAME1) if i >= 2 and g[4] == 0 and g[2] not in parse(g[0])[0]]
    keep2 = [13 + i for i, g in enumerate(tv.GAME2) if g[4] == 0 and g[2] not in parse(g[0])[0]] + [13 + 4]
    assert {r.tobytes() for r in got} == {exp[i].tobytes() for i in keep + keep2}


def test_in_check_flag_and_errors(tmp_path):
    # a single stem: white king on e1 attacked by a black rook on e8
    sq, stm = parse("4r1k1/8/8/8/8/8/8/4K3 w")
    body = stem(sq, stm, set(), None, (4, 3, 0, 0), -120, 50, -1) + struct.pack(">H", 0)
    path = tmp_path / "check.binpack"
    path.write_bytes(b"BINP" + struct.pack("<I", len(body)) + body)
    flags = []
    filt = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(SfEntry))(lambda _u, e: flags.append(e.contents.in_check) or 1)
    h = open_reader([str(path)], 1, 1, 1, filt)
    out = np.zeros((3, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, out.ctypes.data, 3))
    L.bt_reader_close(h)
    assert flags[0] == 1
    ref = np.zeros(32, dtype=np.uint8)
    lib.check(L.bt_chessboard_from_str(b"4r1k1/8/8/8/8/8/8/4K3 w - - 0 1 | -120 | 0.0", ref.ctypes.data))
    assert all(np.array_equal(r, ref) for r in out)
    bad = tmp_path / "bad.binpack"
    bad.write_bytes(b"NOPE" + struct.pack("<I", 4) + b"abcd")
    h = open_reader([str(bad)], 1, 1, 1)
    assert L.bt_reader_next(h, out.ctypes.data, 1) != 0 and b"chunk header" in L.bt_last_error()
    L.bt_reader_close(h)
    trunc = tmp_path / "trunc.binpack"
    trunc.write_bytes(b"BINP" + struct.pack("<I", len(body) + 5) + body + b"\1\2\3\4\5")
    h = open_reader([str(trunc)], 1, 1, 1)
    assert L.bt_reader_next(h, out.ctypes.data, 1) != 0 and b"malformed" in L.bt_last_error()
    L.bt_reader_close(h)


# ------------------------------------------------------------------------------------------ random playouts
def fen_of(sq, stm):
    rows = []
    for rank in range(7, -1, -1):
        row, gap = "", 0
        for file in range(8):
            pc = sq.get(8 * rank + file)
            if pc is None:
                gap += 1
            else:
                row += (str(gap) if gap else "") + ("PNBRQK"[pc[1]] if pc[0] == 0 else "pnbrqk"[pc[1]])
                gap = 0
        rows.append(row + (str(gap) if gap else ""))
    return "/".join(rows) + (" b" if stm else " w")


def slider_hits(sq, target, by):
    for s in ray(sq, target, DIAG):
        if s in sq and sq[s][0] == by and sq[s][1] in (B, Q):
            return True
    for s in ray(sq, target, ORTH):
        if s in sq and sq[s][0] == by and sq[s][1] in (R, Q):
            return True
    return False


LOST = {0: (0, "long"), 7: (0, "short"), 56: (1, "long"), 63: (1, "short")}


def play(sq, stm, castle, frm, to, promo, flag):
    """-> (board, side, castling rights, en-passant square) after the move; the en-passant square exists only if a
    pawn can take there without exposing its king to a bishop, rook or queen"""
    sq, castle = dict(sq), set(castle)
    pc = sq.pop(frm)
    ep = None
    if flag == 2:
        rook = sq.pop(to)
        rank, short = frm & 56, to > frm
        sq[rank | (6 if short else 2)], sq[rank | (5 if short else 3)] = pc, rook
    else:
        if flag == 1:
            sq.pop(to + (8 if stm else -8))
        sq[to] = (stm, N + promo) if flag == 3 else pc
    for s in (frm, to):
        castle.discard(LOST.get(s))
        if s in (4, 60):
            castle -= {(s // 56, "long"), (s // 56, "short")}
    if pc[1] == P and abs(to - frm) == 16 and flag == 0:
        cand, taker = (frm + to) // 2, stm ^ 1
        king = [s for s, v in sq.items() if v == (taker, K)]
        for q in steps(cand, [(-1, 1 if taker else -1), (1, 1 if taker else -1)]):
            if sq.get(q) == (taker, P):
                after = {k: v for k, v in sq.items() if k not in (q, to)}
                after[cand] = (taker, P)
                if not king or not slider_hits(after, king[0], stm):
                    ep = cand
    return sq, stm ^ 1, castle, ep


def pseudo_moves(sq, stm, castle, ep):
    out = []
    own = {s for s, v in sq.items() if v[0] == stm}
    for frm in sorted(own):
        t = sq[frm][1]
        if t == P:
            up = -1 if stm else 1
            dest = [d for d in steps(frm, [(-1, up), (1, up)]) if (d in sq and sq[d][0] != stm) or d == ep]
            f1 = frm + 8 * up
            if f1 not in sq:
                dest.append(f1)
                if (frm >> 3) == (6 if stm else 1) and f1 + 8 * up not in sq:
                    dest.append(f1 + 8 * up)
            for d in dest:
                if (frm >> 3) == (1 if stm else 6):
                    out += [(frm, d, pr, 3) for pr in range(4)]
                else:
                    out.append((frm, d, 0, 1 if d == ep and d not in sq else 0))
        elif t == K:
            out += [(frm, d, 0, 0) for d in steps(frm, DIAG + ORTH) if d not in own]
            for kind, rook, between in (("short", frm | 7, (5, 6)), ("long", frm & 56, (1, 2, 3))):
                if (stm, kind) in castle and all((frm & 56) | f not in sq for f in between):
                    out.append((frm, rook, 0, 2))
        else:
            cand = steps(frm, KN) if t == N else ray(sq, frm, (DIAG if t in (B, Q) else []) + (ORTH if t in (R, Q) else []))
            out += [(frm, d, 0, 0) for d in cand if d not in own]
    return [m for m in out if sq.get(m[1], (0, 0))[1] != K or m[3] == 2]  # kings stay on the board


def test_random_playouts_round_trip(tmp_path):
    """40 random pseudo-legal games of up to 80 plies (the format indexes pseudo-legal candidate sets, so legality is
    not needed): encoded by the Python side, decoded by the library, every position compared with the record its
    FEN parses to. Exercises sliders, promotions, en passant (with the capture-possible rule), lost castling rights."""
    rng = np.random.default_rng(2024)
    body, want, viri = b"", [], b""
    for g in range(40):
        sq, stm = parse(tv.START + " w")
        castle, ep = set(ALL), None
        game, states, scores = [], [], []
        for ply in range(int(rng.integers(1, 81))):
            moves = pseudo_moves(sq, stm, castle, ep)
            if not moves or len(sq) < 3:
                break
            special = [m for m in moves if m[3] != 0]
            mv = special[int(rng.integers(len(special)))] if special and rng.random() < 0.5 else moves[int(rng.integers(len(moves)))]
            game.append((fen_of(sq, stm),) + mv)
            states.append((set(castle), ep))
            scores.append(int(rng.integers(-3000, 3001)))
            sq, stm, castle, ep = play(sq, stm, castle, *mv)
        res_white = int(rng.integers(-1, 2))
        first_stm = parse(game[0][0])[1]
        body += chain(game, states, scores, 0, res_white if first_stm == 0 else -res_white)
        whites = []
        for (fen, *_), s in zip(game, scores):
            white = s if parse(fen)[1] == 0 else -s
            whites.append(white)
            want.append(f"{fen} - - 0 1 | {white} | {['0.0', '0.5', '1.0'][res_white + 1]}")
        viri += tv.encode_game(game, {0, 7, 56, 63}, res_white + 1, whites)  # the same games in viriformat
    path = tmp_path / "random.binpack"
    path.write_bytes(b"BINP" + struct.pack("<I", len(body)) + body)
    exp = np.zeros((len(want), 32), dtype=np.uint8)
    for i, line in enumerate(want):
        lib.check(L.bt_chessboard_from_str(line.encode(), exp[i].ctypes.data))
    assert len(want) > 800
    h = open_reader([str(path)], 1, 3, 77)
    got = np.zeros((16384, 32), dtype=np.uint8)
    lib.check(L.bt_reader_next(h, got.ctypes.data, 16384))
    L.bt_reader_close(h)
    perm, _ = tv.py_shuffle(16384, 77)
    assert np.array_equal(got, exp[np.arange(16384) % len(want)][perm])
    # and through the viriformat replay (a move with eval 0 from square 0 to square 0 would read as a terminator:
    # the generator never makes one -- a1a1 is not a move)
    vpath = tmp_path / "random.vf"
    vpath.write_bytes(viri)
    h = tv.open_reader([str(vpath)], 1, 3, 77)
    lib.check(L.bt_reader_next(h, got.ctypes.data, 16384))
    L.bt_reader_close(h)
    assert np.array_equal(got, exp[np.arange(16384) % len(want)][perm])


def test_garbage_chunks_fail_cleanly(tmp_path):
    """random bytes behind a valid chunk header: the decoder must report a malformed binpack (or, by luck, decode
    something), never crash -- stems, bit reader and move replay are all bounds-checked"""
    rng = np.random.default_rng(5)
    out = np.zeros((4, 32), dtype=np.uint8)
    for trial in range(200):
        n = int(rng.integers(34, 400))
        body = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        if trial % 2:  # a valid stem followed by garbage movetext
            sq, stm = parse(tv.START + " w")
            body = stem(sq, stm, ALL, None, (12, 28, 0, 0), 10, 0, 0) + struct.pack(">H", 40) + body
        path = tmp_path / f"g{trial}.binpack"
        path.write_bytes(b"BINP" + struct.pack("<I", len(body)) + body)
        h = open_reader([str(path)], 1, 1, 1)
        rc = L.bt_reader_next(h, out.ctypes.data, 4)
        assert rc in (0, 4)
        L.bt_reader_close(h)
        path.unlink()
