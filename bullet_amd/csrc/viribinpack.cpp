// ViriBinpackLoader (crates/bullet_lib/src/value/loader/viribinpack.rs:57-219): games in viriformat -> positions in
// bulletformat -> shuffle buffer -> the record stream bt_reader_next serves.
//
// What the reference's file states and this one follows: the pipeline (files read in order and looped forever; games
// decoded on `threads` workers, :154-174; a shuffle buffer of buffer_size_mb * 2^20 / 32 / 2 positions, :50, shuffled by
// a Fisher-Yates walk driven by the xorshift of value/loader/rng.rs:13-18 when it is full, :121-139 and :212-219;
// the custom filter contract `fn(&Board, Move, i16, f32) -> bool`, true = keep, :188-206).
//
// What it does NOT state: the wire format and the board replay live in the crates.io dependency `viriformat 2.0.1`,
// which is not vendored. They are restated here from that crate's published format -- **parity unpinned**: the
// reference holds no viriformat file or test vector. Layout used:
//   game   = marlinformat PackedBoard (32 B) | { move u16 LE, eval i16 LE }* | 4 zero bytes
//   board  = occupancy u64 LE | 32 piece nibbles in ascending-square order of the occupancy, low nibble first:
//            colour << 3 | type, types P N B R Q K = 0..5 and 6 = a rook that may still castle |
//            stm_ep u8 (bit 7 = black to move, low 7 bits = en-passant square or 64) | halfmove u8 | fullmove u16 LE |
//            eval i16 LE | wdl u8 (2 = white win, 1 = draw, 0 = black win) | extra u8
//   move   = from (6 bits) | to << 6 | promotion piece (N B R Q = 0..3) << 12 | flag << 14
//            (1 = en passant, 2 = castling written as king-takes-own-rook, 3 = promotion)
// Every position BEFORE each move is emitted (Game::splat_to_bulletformat) with that move's eval (white-relative)
// and the game result, through ChessBoard::from_raw (side-to-move relative).
//
// ViriFilter::Builtin (viribinpack.rs:178-187, `Filter::default()` in examples/simple.rs:84): the crate's own `Filter`
// is restated too, from its published fields -- **parity unpinned** like the format. A position is dropped when
//   ply < min_ply (ply = 2 * (fullmove - 1) + black to move, counted on through the game) | |eval| >= max_eval |
//   pieces on the board < min_pieces | filter_tactical and the move captures (en passant too) or promotes |
//   filter_check and the side to move is in check | filter_castling and the move castles |
//   max_eval_incorrectness set and the eval points away from the game result by more than it (a draw: |eval|) |
//   random_fen_skipping and a uniform draw < random_fen_skip_probability.
// Default: min_ply 16, min_pieces 4, max_eval = the tablebase-win bound 20000, tactical and check filters on, the
// rest off; `UNRESTRICTED` drops nothing. The crate's WDL-model filter (`wdl_filtered`, off by default) is not restated.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "game_source.h"

typedef bt_game_board ViriBoard;

namespace {

void unpack(const uint8_t *p, ViriBoard *b, int16_t *eval, int *wdl) {
    uint64_t occ;
    std::memcpy(&occ, p, 8);
    std::memset(b->sq, 0xFF, 64);
    int i = 0;
    for (uint64_t o = occ; o; o &= o - 1, i++) {
        const int s = __builtin_ctzll(o);
        const uint8_t code = (p[8 + i / 2] >> (4 * (i & 1))) & 15;
        uint8_t type = code & 7;
        if (type == 6) type = 3;  // unmoved rook
        b->sq[s] = (uint8_t)((code & 8) | type);
    }
    b->black_to_move = (p[24] & 0x80) != 0;
    uint16_t fullmove;
    std::memcpy(&fullmove, p + 26, 2);
    b->fullmove = fullmove;
    std::memcpy(eval, p + 28, 2);
    *wdl = p[30];
}

// Board::make_move_simple: no legality checks, no clocks / castling rights (not part of a bulletformat record)
void make_move(ViriBoard *b, uint16_t mv) {
    const int from = mv & 63, to = (mv >> 6) & 63, flag = mv >> 14;
    const uint8_t piece = b->sq[from];
    const uint8_t colour = piece & 8;
    if (flag == 2) {  // castling: `to` is the rook's square; king to the g / c file, rook to f / d
        const uint8_t rook = b->sq[to];
        const int rank = from & 56;
        const bool kingside = to > from;
        b->sq[from] = 0xFF, b->sq[to] = 0xFF;
        b->sq[rank | (kingside ? 6 : 2)] = piece;
        b->sq[rank | (kingside ? 5 : 3)] = rook;
    } else {
        if (flag == 1) b->sq[to ^ 8] = 0xFF;  // en passant: the pawn taken stands beside the origin, behind `to`
        b->sq[to] = flag == 3 ? (uint8_t)(colour | (1 + ((mv >> 12) & 3))) : piece;
        b->sq[from] = 0xFF;
    }
    b->black_to_move = !b->black_to_move;
}

}  // namespace

// ChessBoard::from_raw(bbs, stm, score, result) of bulletformat (see host_inputs.cpp for the same convention)
bool bt_internal_board_to_record(const bt_game_board *b, int16_t eval_white, int wdl_white, uint8_t *out) {
    uint64_t occ = 0;
    uint8_t pcs[16] = {0};
    int n = 0, own_k = -1, opp_k = -1;
    const uint8_t own = b->black_to_move ? 8 : 0;
    for (int s = 0; s < 64; s++) {  // ascending squares of the side-to-move-relative board
        const int src = b->black_to_move ? (s ^ 56) : s;
        const uint8_t pc = b->sq[src];
        if (pc == 0xFF) continue;
        if (n >= 32) return false;
        const uint8_t type = pc & 7, rel = (uint8_t)(((pc & 8) == own ? 0 : 8) | type);
        occ |= 1ull << s;
        pcs[n / 2] |= (uint8_t)(rel << (4 * (n & 1)));
        n++;
        if (type == 5) ((pc & 8) == own ? own_k : opp_k) = s;
    }
    if (own_k < 0 || opp_k < 0) return false;
    const int16_t score = b->black_to_move ? (int16_t)(-eval_white) : eval_white;
    const uint8_t result = (uint8_t)(b->black_to_move ? 2 - wdl_white : wdl_white);
    std::memcpy(out, &occ, 8);
    std::memcpy(out + 8, pcs, 16);
    std::memcpy(out + 24, &score, 2);
    out[26] = result, out[27] = (uint8_t)own_k, out[28] = (uint8_t)(opp_k ^ 56);
    out[29] = out[30] = out[31] = 0;
    return true;
}

namespace {

// Is `sq` attacked by a piece of colour `by` (0 white, 8 black)? Plain walks over the 64-square array.
bool attacked(const ViriBoard *b, int sq, uint8_t by) {
    const int r = sq >> 3, f = sq & 7;
    auto at = [&](int rr, int ff) -> int { return (rr < 0 || rr > 7 || ff < 0 || ff > 7) ? -1 : b->sq[rr * 8 + ff]; };
    const int pr = by ? r + 1 : r - 1;  // a white pawn attacks upwards: it stands one rank below its target
    for (int df = -1; df <= 1; df += 2)
        if (at(pr, f + df) == (by | 0)) return true;
    static const int kn[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
    for (auto &d : kn)
        if (at(r + d[0], f + d[1]) == (by | 1)) return true;
    for (int dr = -1; dr <= 1; dr++)
        for (int df = -1; df <= 1; df++) {
            if (!dr && !df) continue;
            if (at(r + dr, f + df) == (by | 5)) return true;
            const bool diag = dr && df;
            for (int rr = r + dr, ff = f + df;; rr += dr, ff += df) {
                const int pc = at(rr, ff);
                if (pc < 0) break;
                if (pc == 0xFF) continue;
                if (pc == (by | 4) || pc == (by | (diag ? 2 : 3))) return true;
                break;
            }
        }
    return false;
}

// Filter::should_filter of the crate, as restated in the header of this file: true = drop the position
bool builtin_drops(const bt_viri_builtin_filter *f, const ViriBoard *b, uint16_t mv, int eval, int wdl_white, int ply,
                   uint64_t *rng) {
    if ((uint32_t)ply < f->min_ply) return true;
    const uint32_t abs_eval = (uint32_t)(eval < 0 ? -eval : eval);
    if (abs_eval >= f->max_eval) return true;
    uint32_t pieces = 0;
    int king = -1;
    const uint8_t own = b->black_to_move ? 8 : 0;
    for (int s = 0; s < 64; s++) {
        if (b->sq[s] == 0xFF) continue;
        pieces++;
        if (b->sq[s] == (own | 5)) king = s;
    }
    if (pieces < f->min_pieces) return true;
    const int to = (mv >> 6) & 63, flag = mv >> 14;
    if (f->filter_tactical && (flag == 1 || flag == 3 || (flag != 2 && b->sq[to] != 0xFF))) return true;
    if (f->filter_check && king >= 0 && attacked(b, king, own ^ 8)) return true;
    if (f->filter_castling && flag == 2) return true;
    if (f->max_eval_incorrectness != UINT32_MAX) {
        const uint32_t err = wdl_white == 2 ? (eval < 0 ? abs_eval : 0) : wdl_white == 0 ? (eval > 0 ? abs_eval : 0) : abs_eval;
        if (err > f->max_eval_incorrectness) return true;
    }
    if (f->random_fen_skipping) {
        *rng ^= *rng << 13, *rng ^= *rng >> 7, *rng ^= *rng << 17;
        if ((double)(*rng >> 11) * (1.0 / 9007199254740992.0) < f->random_fen_skip_probability) return true;
    }
    return false;
}

struct ViriFilterSet {
    bt_viri_filter filter;
    void *user;
    bool has_builtin = false;
    bt_viri_builtin_filter builtin;
};

// Game::deserialise_fast_into_buffer: header + moves up to and including the 4-byte terminator; false at EOF
bool read_game(FILE *fp, std::vector<uint8_t> *games, std::vector<size_t> *starts) {
    uint8_t head[32];
    if (std::fread(head, 1, 32, fp) != 32) return false;
    const size_t at = games->size();
    games->insert(games->end(), head, head + 32);
    for (;;) {
        uint8_t m[4];
        if (std::fread(m, 1, 4, fp) != 4) {
            games->resize(at);  // truncated game: dropped, like the `is_err()` break of viribinpack.rs:79-81
            return false;
        }
        games->insert(games->end(), m, m + 4);
        if (m[0] == 0 && m[1] == 0 && m[2] == 0 && m[3] == 0) break;
    }
    starts->push_back(at);
    return true;
}

void splat(const uint8_t *g, const ViriFilterSet *src, std::vector<uint8_t> *out, std::atomic<bool> *bad, uint64_t *skip_rng) {
    ViriBoard board;
    int16_t eval0;
    int wdl;
    unpack(g, &board, &eval0, &wdl);
    if (wdl > 2) {
        bad->store(true);
        return;
    }
    const int ply0 = 2 * (std::max(board.fullmove, 1) - 1) + (board.black_to_move ? 1 : 0);
    int ply = 0;
    for (const uint8_t *m = g + 32;; m += 4, ply++) {
        uint16_t mv;
        int16_t eval;
        std::memcpy(&mv, m, 2), std::memcpy(&eval, m + 2, 2);
        if (mv == 0 && eval == 0) break;
        uint8_t rec[32];
        if (!bt_internal_board_to_record(&board, eval, wdl, rec)) {
            bad->store(true);
            return;
        }
        const bool keep = src->has_builtin ? !builtin_drops(&src->builtin, &board, mv, eval, wdl, ply0 + ply, skip_rng)
                                           : (!src->filter || src->filter(src->user, rec, mv, eval, (float)wdl / 2.0f, ply));
        if (keep) out->insert(out->end(), rec, rec + 32);
        if (board.sq[mv & 63] == 0xFF) {  // a move from an empty square: not a game of this format
            bad->store(true);
            return;
        }
        make_move(&board, mv);
    }
}

uint64_t xorshift(uint64_t *s) {  // SimpleRand::rng, value/loader/rng.rs:13-18
    *s ^= *s << 13, *s ^= *s >> 7, *s ^= *s << 17;
    return *s;
}

}  // namespace

// One full shuffle buffer into `out`. 0 = ok. (shared with montybinpack.cpp: the same pipeline around another decoder)
int bt_internal_game_refill(bt_game_source *src, std::vector<uint8_t> *out) {
    const size_t want = src->buffer_positions * 32;
    out->clear();
    out->swap(src->carry);
    size_t size_at_wrap = (size_t)-1;  // positions held when the file list last wrapped: an epoch must add some
    while (out->size() < want) {
        // read a group of games (the reference ships 8192 * threads at a time, viribinpack.rs:85)
        std::vector<uint8_t> games;
        std::vector<size_t> starts;
        const size_t group = (size_t)8192 * (size_t)src->threads;
        bool wrapped = false;
        while (starts.size() < group) {
            if (!src->fp) {
                src->fp = std::fopen(src->paths[src->file_idx].c_str(), "rb");
                if (!src->fp) {
                    bt_set_error("File not found: %s", src->paths[src->file_idx].c_str());
                    return BT_ERR_IO;
                }
            }
            if (!src->read_game(src->fp, &games, &starts)) {  // end of this file: next one, round and round (:73-94)
                std::fclose(src->fp);
                src->fp = nullptr;
                src->file_idx = (src->file_idx + 1) % src->paths.size();
                if (src->file_idx == 0) {  // decode what we have before starting the next epoch
                    wrapped = true;
                    break;
                }
            }
        }
        // convert_buffer (:154-174): contiguous groups of games per worker, results appended in worker order
        const int workers = (int)std::max<size_t>(1, std::min<size_t>((size_t)src->threads, starts.size()));
        std::vector<std::vector<uint8_t>> parts(workers);
        std::atomic<bool> bad(false);
        const size_t per = (starts.size() + workers - 1) / workers;
        const uint64_t skip_base = src->skip_counter;
        src->skip_counter += (uint64_t)workers;
        auto work = [&](int w) {
            // (random skipping: one xorshift stream per worker and group of games, seeded by a counter the shuffle seed starts)
            uint64_t z = skip_base + 0x9E3779B97F4A7C15ull * (uint64_t)(w + 1);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull, z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            uint64_t skip_rng = (z ^ (z >> 31)) | 1;
            for (size_t i = w * per; i < std::min(starts.size(), (w + 1) * per); i++)
                src->splat(games.data() + starts[i], &parts[w], &bad, &skip_rng);
        };
        std::vector<std::thread> pool;
        for (int w = 1; w < workers; w++) pool.emplace_back(work, w);
        work(0);
        for (auto &t : pool) t.join();
        if (bad.load()) {
            bt_set_error("%s: malformed game in %s", src->what, src->paths[src->file_idx].c_str());
            return BT_ERR_IO;
        }
        for (auto &p : parts) out->insert(out->end(), p.begin(), p.end());
        if (wrapped) {
            if (out->size() == size_at_wrap) {
                bt_set_error("%s: no position passes the filter (or the files hold no games)", src->what);
                return BT_ERR_IO;
            }
            size_at_wrap = out->size();
        }
    }
    if (out->size() > want) {  // keep the overshoot for the next buffer (:124-138)
        src->carry.assign(out->begin() + want, out->end());
        out->resize(want);
    }
    // shuffle (:212-219): for i in (0..len).rev() { swap(rng % (i + 1), i) }
    const size_t n = out->size() / 32;
    uint8_t tmp[32];
    for (size_t i = n; i-- > 0;) {
        const size_t j = (size_t)(xorshift(&src->rng) % (uint64_t)(i + 1));
        if (j == i) continue;
        uint8_t *a = out->data() + 32 * i, *b = out->data() + 32 * j;
        std::memcpy(tmp, a, 32), std::memcpy(a, b, 32), std::memcpy(b, tmp, 32);
    }
    return BT_OK;
}

int bt_internal_game_source_open(bt_game_source *src, const char *const *paths, int n_paths, size_t buffer_size_mb,
                                 int threads, uint64_t shuffle_seed, const char *who) {
    BT_REQUIRE(paths && n_paths > 0, "%s: bad arguments", who);
    BT_REQUIRE(buffer_size_mb > 0, "%s: buffer_size_mb must be positive", who);
    for (int i = 0; i < n_paths; i++) {
        FILE *f = std::fopen(paths[i], "rb");
        if (!f) {
            bt_set_error("File not found: %s", paths[i]);
            return BT_ERR_IO;
        }
        std::fclose(f);
        src->paths.push_back(paths[i]);
    }
    src->buffer_positions = buffer_size_mb * 1024 * 1024 / 32 / 2;  // viribinpack.rs:50, montybinpack.rs:35
    src->threads = threads > 0 ? threads : 1;
    if (shuffle_seed == 0) {  // SimpleRand::with_seed: microseconds since the epoch & 0xFFFF_FFFF (rng.rs:6-11)
        const auto us = std::chrono::duration_cast<std::chrono::microseconds>(
                            std::chrono::system_clock::now().time_since_epoch()).count();
        shuffle_seed = (uint64_t)us & 0xFFFFFFFFull;
        if (shuffle_seed == 0) shuffle_seed = 1;
    }
    src->rng = shuffle_seed, src->skip_counter = shuffle_seed * 0x2545F4914F6CDD1Dull;
    return BT_OK;
}

extern "C" void bt_viri_filter_default(bt_viri_builtin_filter *f) {
    if (!f) return;
    *f = bt_viri_builtin_filter{};
    f->min_ply = 16, f->min_pieces = 4, f->max_eval = 20000;
    f->filter_tactical = 1, f->filter_check = 1, f->filter_castling = 0;
    f->max_eval_incorrectness = UINT32_MAX;
    f->random_fen_skipping = 0, f->random_fen_skip_probability = 0.0;
}

extern "C" void bt_viri_filter_unrestricted(bt_viri_builtin_filter *f) {
    if (!f) return;
    *f = bt_viri_builtin_filter{};
    f->max_eval = UINT32_MAX, f->max_eval_incorrectness = UINT32_MAX;
}

int bt_internal_viri_source(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                            bt_viri_filter filter, void *user, const bt_viri_builtin_filter *builtin,
                            uint64_t shuffle_seed, std::function<int(std::vector<uint8_t> &)> *out_refill) {
    BT_REQUIRE(out_refill, "bt_reader_open_viribinpack: bad arguments");
    auto src = std::make_shared<bt_game_source>();
    if (int rc = bt_internal_game_source_open(src.get(), paths, n_paths, buffer_size_mb, threads, shuffle_seed,
                                              "bt_reader_open_viribinpack"))
        return rc;
    ViriFilterSet fs;
    fs.filter = filter, fs.user = user;
    if (builtin) {
        BT_REQUIRE(builtin->random_fen_skip_probability >= 0.0 && builtin->random_fen_skip_probability <= 1.0,
                   "bt_reader_open_viribinpack: random_fen_skip_probability outside [0, 1]");
        fs.has_builtin = true, fs.builtin = *builtin;
    }
    src->what = "viribinpack";
    src->read_game = read_game;
    src->splat = [fs](const uint8_t *g, std::vector<uint8_t> *out, std::atomic<bool> *bad, uint64_t *skip_rng) {
        splat(g, &fs, out, bad, skip_rng);
    };
    *out_refill = [src](std::vector<uint8_t> &out) { return bt_internal_game_refill(src.get(), &out); };
    return BT_OK;
}
