// MontyBinpackLoader (crates/bullet_lib/src/value/loader/montybinpack.rs:17-186): games in montyformat -> positions in
// bulletformat -> shuffle buffer -> the record stream bt_reader_next serves.
//
// What the reference's file states and this one follows: the pipeline (game_source.h), the replay loop (:160-176: for
// every move, `filter(&pos, best_move, score, result)` decides whether the position BEFORE the move is emitted as
// `ChessBoard::from_raw(pos.bbs(), pos.stm(), score, result)`, then the move is made), filter = true keeps.
//
// What it does NOT state: the wire format and the move maker live in the crates.io dependency `montyformat 0.9.2`, which
// is not vendored. They are restated here from that crate's published format -- **parity unpinned**: the reference holds
// no montyformat file or test vector. Layout used (`MontyValueFormat`):
//   game   = compressed board (38 B) | rook files u8 x 4 (white, black; two each) | result u8 (game result x 2, for
//            white) | { best move u16 LE, score i16 LE }* | 4 zero bytes
//   board  = four u64 LE planes: black's pieces, R^Q^K, N^B^K, P^B^Q (a piece's type is the 3-bit code its square has in
//            the last three: P 001, N 010, B 011, R 100, Q 101, K 110) | stm u8 (1 = black) | en-passant square u8 |
//            castling rights u8 | halfmove u8 | fullmove u16 LE
//   move   = from << 10 | to << 4 | flag; flags: 0 quiet, 1 double push, 2 / 3 castle king- / queen-side (`to` = the
//            king's destination; the rook comes from its start file on that side), 4 capture, 5 en passant,
//            8..11 promotion to N B R Q, 12..15 the same with capture
// Scores and the result are white-relative in the file; from_raw turns them to the side to move.
#include <cstring>
#include <memory>

#include "common.h"
#include "game_source.h"

namespace {

constexpr size_t MONTY_HEADER = 43;

bool unpack(const uint8_t *p, bt_game_board *b, uint8_t rook_files[2][2], int *result) {
    uint64_t pl[4];
    std::memcpy(pl, p, 32);
    std::memset(b->sq, 0xFF, 64);
    const uint64_t occ = pl[1] | pl[2] | pl[3];
    for (uint64_t o = occ; o; o &= o - 1) {
        const int s = __builtin_ctzll(o);
        const int code = (int)((pl[1] >> s) & 1) << 2 | (int)((pl[2] >> s) & 1) << 1 | (int)((pl[3] >> s) & 1);
        if (code < 1 || code > 6) return false;
        b->sq[s] = (uint8_t)((((pl[0] >> s) & 1) ? 8 : 0) | (code - 1));
    }
    if (p[32] > 1) return false;
    b->black_to_move = p[32] == 1;
    uint16_t fullmove;
    std::memcpy(&fullmove, p + 36, 2);
    b->fullmove = fullmove;
    for (int side = 0; side < 2; side++)
        for (int k = 0; k < 2; k++) rook_files[side][k] = p[38 + 2 * side + k];
    *result = p[42];
    return *result <= 2;
}

// Position::make of the crate, as far as a bulletformat record sees it (placement and side to move only)
bool make_move(bt_game_board *b, uint16_t mv, const uint8_t rook_files[2][2]) {
    const int from = mv >> 10, to = (mv >> 4) & 63, flag = mv & 15;
    const uint8_t piece = b->sq[from];
    if (piece == 0xFF) return false;
    const uint8_t colour = piece & 8;
    const int side = colour ? 1 : 0;
    if (flag == 2 || flag == 3) {  // the rook of that side: the start file on the king's side of the king
        const int rank = from & 56, kfile = from & 7;
        int rfile = -1;
        for (int k = 0; k < 2; k++) {
            const int f = rook_files[side][k];
            if (f < 8 && (flag == 2 ? f > kfile : f < kfile)) rfile = f;
        }
        if (rfile < 0 || b->sq[rank | rfile] != (colour | 3)) return false;
        b->sq[from] = 0xFF, b->sq[rank | rfile] = 0xFF;
        b->sq[rank | (flag == 2 ? 6 : 2)] = piece;
        b->sq[rank | (flag == 2 ? 5 : 3)] = (uint8_t)(colour | 3);
    } else {
        if (flag == 5) b->sq[to ^ 8] = 0xFF;  // en passant: the pawn taken stands behind `to`
        b->sq[to] = (flag & 8) ? (uint8_t)(colour | (1 + (flag & 3))) : piece;
        b->sq[from] = 0xFF;
    }
    if (b->black_to_move) b->fullmove++;
    b->black_to_move = !b->black_to_move;
    return true;
}

// deserialise_fast_into_buffer: header + moves up to and including the 4-byte terminator; false at EOF
bool read_game(FILE *fp, std::vector<uint8_t> *games, std::vector<size_t> *starts) {
    uint8_t head[MONTY_HEADER];
    if (std::fread(head, 1, MONTY_HEADER, fp) != MONTY_HEADER) return false;
    const size_t at = games->size();
    games->insert(games->end(), head, head + MONTY_HEADER);
    for (;;) {
        uint8_t m[4];
        if (std::fread(m, 1, 4, fp) != 4) {
            games->resize(at);  // truncated game: dropped (the `while let Ok(())` of montybinpack.rs:64 ends)
            return false;
        }
        games->insert(games->end(), m, m + 4);
        if (m[0] == 0 && m[1] == 0 && m[2] == 0 && m[3] == 0) break;
    }
    starts->push_back(at);
    return true;
}

void splat(const uint8_t *g, bt_monty_filter filter, void *user, std::vector<uint8_t> *out, std::atomic<bool> *bad) {
    bt_game_board board;
    uint8_t rook_files[2][2];
    int result;
    if (!unpack(g, &board, rook_files, &result)) {
        bad->store(true);
        return;
    }
    for (const uint8_t *m = g + MONTY_HEADER;; m += 4) {
        uint16_t mv;
        int16_t score;
        std::memcpy(&mv, m, 2), std::memcpy(&score, m + 2, 2);
        if (mv == 0 && score == 0) break;
        uint8_t rec[32];
        if (!bt_internal_board_to_record(&board, score, result, rec)) {
            bad->store(true);
            return;
        }
        if (!filter || filter(user, rec, mv, score, (float)result / 2.0f)) out->insert(out->end(), rec, rec + 32);
        if (!make_move(&board, mv, rook_files)) {
            bad->store(true);
            return;
        }
    }
}

}  // namespace

int bt_internal_monty_source(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                             bt_monty_filter filter, void *user, uint64_t shuffle_seed,
                             std::function<int(std::vector<uint8_t> &)> *out_refill) {
    BT_REQUIRE(out_refill, "bt_reader_open_montybinpack: bad arguments");
    auto src = std::make_shared<bt_game_source>();
    if (int rc = bt_internal_game_source_open(src.get(), paths, n_paths, buffer_size_mb, threads, shuffle_seed,
                                              "bt_reader_open_montybinpack"))
        return rc;
    src->what = "montybinpack";
    src->read_game = read_game;
    src->splat = [filter, user](const uint8_t *g, std::vector<uint8_t> *out, std::atomic<bool> *bad, uint64_t *) {
        splat(g, filter, user, out, bad);
    };
    *out_refill = [src](std::vector<uint8_t> &out) { return bt_internal_game_refill(src.get(), &out); };
    return BT_OK;
}
