// SfBinpackLoader (crates/bullet_lib/src/value/loader/sfbinpack.rs:23-225): Stockfish "binpack" training data ->
// positions in bulletformat -> shuffle buffer -> the record stream bt_reader_next serves.
//
// From the reference's file: the pipeline (files in order, round and round; chunks decoded on `threads` workers,
// :175-215; every entry that passes the user's filter converted with convert_to_bulletformat, :23-48; a shuffle
// buffer of buffer_size_mb * 2^20 / 32 / 2 positions shuffled by the xorshift Fisher-Yates of rng.rs, :143-164,
// :217-224).
//
// NOT in the reference tree: the binpack wire format and the chess rules needed to replay it; they live in the
// crates.io dependency `sfbinpack 0.6.4` (a port of the Stockfish tools' nnue_training_data_formats.h), not vendored.
// They are restated here from that published format -- **parity unpinned**: the reference holds no binpack file or
// vector. Format as restated:
//   file    = { "BINP" | u32 LE chunk size | chunk }*
//   chunk   = { stem (32 B) | u16 BE number of following plies | movetext }*
//   stem    = occupancy u64 BE | 32 piece nibbles (ascending squares, low nibble first) | move u16 BE |
//             score u16 BE (zig-zag) | ply (14 bits) + result (zig-zag, 2 bits) u16 BE | rule50 u16 BE
//   nibble  = 0..11: P p N n B b R r Q q K k (white, black) | 12: pawn whose double push left an en-passant square
//             behind it | 13 / 14: white / black rook that may still castle | 15: black king, black to move
//   move    = type << 14 (0 normal, 1 promotion, 2 castling as king-takes-rook, 3 en passant) | from << 8 | to << 2 |
//             promotion piece (N B R Q = 0..3)
//   movetext, MSB-first bit stream, per ply: index of the moving piece among the mover's pieces (ascending squares),
//             index of the destination among that piece's candidate squares (pawn: captures incl. en passant, push,
//             double push; x4 + piece on the promotion rank; king: king steps not onto own pieces, then the castlings
//             still allowed, long first; others: attacks under the current occupancy not onto own pieces), each in the
//             minimal number of bits for the count; then score = -previous + zig-zag(VLE, 4-bit blocks with a
//             continuation flag); ply + 1, result negated. The stream is padded to a byte at the end of a chain.
// Scores and results are side-to-move relative, as a bulletformat record wants them.
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

namespace {

enum { PAWN = 0, KNIGHT, BISHOP, ROOK, QUEEN, KING, NONE = 7 };

struct Pos {
    uint8_t sq[64];   // colour << 3 | type, 0xFF empty (colour 0 = white)
    int stm;          // 0 white, 1 black
    int ep;           // en-passant square or -1
    uint8_t castle;   // bit 0 white short (h1), 1 white long (a1), 2 black short (h8), 3 black long (a8)
};

inline uint64_t bit(int s) { return 1ull << s; }

uint64_t colour_bb(const Pos &p, int colour) {
    uint64_t b = 0;
    for (int s = 0; s < 64; s++)
        if (p.sq[s] != 0xFF && (p.sq[s] >> 3) == colour) b |= bit(s);
    return b;
}
uint64_t occupied(const Pos &p) {
    uint64_t b = 0;
    for (int s = 0; s < 64; s++)
        if (p.sq[s] != 0xFF) b |= bit(s);
    return b;
}

uint64_t step_attacks(int from, const int (*d)[2], int n) {
    uint64_t b = 0;
    const int f = from & 7, r = from >> 3;
    for (int i = 0; i < n; i++) {
        const int nf = f + d[i][0], nr = r + d[i][1];
        if (nf >= 0 && nf < 8 && nr >= 0 && nr < 8) b |= bit(nr * 8 + nf);
    }
    return b;
}
const int KNIGHT_D[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
const int KING_D[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};

uint64_t slide(int from, uint64_t occ, bool diag, bool ortho) {
    uint64_t b = 0;
    static const int D[8][2] = {{1, 1}, {1, -1}, {-1, 1}, {-1, -1}, {1, 0}, {-1, 0}, {0, 1}, {0, -1}};
    for (int i = 0; i < 8; i++) {
        if ((i < 4 && !diag) || (i >= 4 && !ortho)) continue;
        int f = (from & 7) + D[i][0], r = (from >> 3) + D[i][1];
        while (f >= 0 && f < 8 && r >= 0 && r < 8) {
            b |= bit(r * 8 + f);
            if (occ & bit(r * 8 + f)) break;
            f += D[i][0], r += D[i][1];
        }
    }
    return b;
}
uint64_t piece_attacks(int type, int from, uint64_t occ) {
    switch (type) {
    case KNIGHT: return step_attacks(from, KNIGHT_D, 8);
    case BISHOP: return slide(from, occ, true, false);
    case ROOK: return slide(from, occ, false, true);
    case QUEEN: return slide(from, occ, true, true);
    case KING: return step_attacks(from, KING_D, 8);
    }
    return 0;
}
uint64_t pawn_attacks(int from, int colour) {  // squares a pawn of `colour` on `from` attacks
    const int d[2][2] = {{-1, colour ? -1 : 1}, {1, colour ? -1 : 1}};
    return step_attacks(from, d, 2);
}

// Is `s` attacked by a slider of `by` under occupancy `occ`, with the squares in `gone` emptied first?
bool slider_attacks_square(const Pos &p, int s, int by, uint64_t occ, uint64_t gone) {
    const uint64_t diag = slide(s, occ, true, false), orth = slide(s, occ, false, true);
    for (int q = 0; q < 64; q++) {
        if (p.sq[q] == 0xFF || (p.sq[q] >> 3) != by || (gone & bit(q))) continue;
        const int t = p.sq[q] & 7;
        if ((t == BISHOP || t == QUEEN) && (diag & bit(q))) return true;
        if ((t == ROOK || t == QUEEN) && (orth & bit(q))) return true;
    }
    return false;
}

bool square_attacked(const Pos &p, int s, int by) {
    const uint64_t occ = occupied(p);
    for (int q = 0; q < 64; q++) {
        if (p.sq[q] == 0xFF || (p.sq[q] >> 3) != by) continue;
        const int t = p.sq[q] & 7;
        const uint64_t a = t == PAWN ? pawn_attacks(q, by) : piece_attacks(t, q, occ);
        if (a & bit(s)) return true;
    }
    return false;
}

int king_square(const Pos &p, int colour) {
    for (int s = 0; s < 64; s++)
        if (p.sq[s] == (uint8_t)((colour << 3) | KING)) return s;
    return -1;
}

// An en-passant square is only recorded when a pawn of `taker` could really capture on it: it attacks the square and
// taking would not expose its own king to a slider (the only pieces the vanished pawns can unmask).
bool ep_possible(const Pos &p, int ep, int pushed_to, int taker) {
    const int ks = king_square(p, taker);
    const uint64_t occ = occupied(p);
    for (int q = 0; q < 64; q++) {
        if (p.sq[q] != (uint8_t)((taker << 3) | PAWN) || !(pawn_attacks(q, taker) & bit(ep))) continue;
        if (ks < 0) return true;
        const uint64_t occ_after = (occ & ~bit(q) & ~bit(pushed_to)) | bit(ep);
        if (!slider_attacks_square(p, ks, taker ^ 1, occ_after, bit(pushed_to))) return true;
    }
    return false;
}

struct Move {
    int from, to, type, promo;  // type: 0 normal, 1 promotion, 2 castle (to = rook square), 3 en passant
};

inline uint8_t rights_lost(int s) {  // castling rights that end when a piece leaves or lands on `s`
    switch (s) {
    case 0: return 2;    // a1: white long
    case 4: return 3;    // e1
    case 7: return 1;    // h1: white short
    case 56: return 8;   // a8: black long
    case 60: return 12;  // e8
    case 63: return 4;   // h8: black short
    }
    return 0;
}

void do_move(Pos *p, const Move &m) {
    const uint8_t piece = p->sq[m.from];
    const int colour = piece >> 3, type = piece & 7;
    p->ep = -1;
    if (m.type == 2) {
        const uint8_t rook = p->sq[m.to];
        const int rank = m.from & 56;
        const bool shortc = m.to > m.from;
        p->sq[m.from] = 0xFF, p->sq[m.to] = 0xFF;
        p->sq[rank | (shortc ? 6 : 2)] = piece;
        p->sq[rank | (shortc ? 5 : 3)] = rook;
    } else {
        const int taken = m.to + (colour ? 8 : -8);
        if (m.type == 3 && taken >= 0 && taken < 64) p->sq[taken] = 0xFF;  // (range check: malformed input only)
        p->sq[m.to] = m.type == 1 ? (uint8_t)((colour << 3) | (KNIGHT + m.promo)) : piece;
        p->sq[m.from] = 0xFF;
    }
    p->castle &= (uint8_t)~(rights_lost(m.from) | rights_lost(m.to));
    p->stm ^= 1;
    if (m.type == 0 && type == PAWN && (m.to - m.from == 16 || m.from - m.to == 16)) {
        const int ep = (m.from + m.to) / 2;
        if (ep_possible(*p, ep, m.to, colour ^ 1)) p->ep = ep;
    }
}

struct BitReader {
    const uint8_t *data;
    size_t size, offset = 0;
    int bits_left = 8;
    bool overrun = false;
    uint8_t byte_at(size_t i) {
        if (i >= size) {
            overrun = true;
            return 0;
        }
        return data[i];
    }
    unsigned take8(int count) {  // up to 8 bits, most significant first
        if (count == 0) return 0;
        if (bits_left == 0) offset++, bits_left = 8;
        const uint8_t byte = (uint8_t)(byte_at(offset) << (8 - bits_left));
        unsigned bits = (unsigned)(byte >> (8 - count));
        if (count > bits_left) {
            const int spill = count - bits_left;
            bits |= (unsigned)(byte_at(offset + 1) >> (8 - spill));
            bits_left += 8, offset++;
        }
        bits_left -= count;
        return bits;
    }
    unsigned vle16(int block) {
        const unsigned mask = (1u << block) - 1;
        unsigned v = 0;
        for (int off = 0; off < 32; off += block) {
            const unsigned b = take8(block + 1);
            v |= (b & mask) << off;
            if (!(b >> block)) break;
        }
        return v & 0xFFFF;
    }
    size_t bytes_read() const { return offset + (bits_left != 8 ? 1 : 0); }
};

int used_bits_safe(unsigned n) {  // bits needed to index n alternatives
    if (n <= 1) return 0;
    return 32 - __builtin_clz(n - 1);
}
int nth_set(uint64_t b, unsigned n) {
    for (unsigned i = 0; i < n && b; i++) b &= b - 1;
    return b ? __builtin_ctzll(b) : -1;
}
inline int unzigzag(unsigned u) { return (int)(u >> 1) ^ -(int)(u & 1); }

bool read_move(BitReader *br, const Pos &p, Move *out) {
    const uint64_t ours = colour_bb(p, p.stm), theirs = colour_bb(p, p.stm ^ 1), occ = ours | theirs;
    const unsigned piece_id = br->take8(used_bits_safe((unsigned)__builtin_popcountll(ours)));
    const int from = nth_set(ours, piece_id);
    if (from < 0) return false;
    const int type = p.sq[from] & 7;
    Move m{from, -1, 0, 0};
    if (type == PAWN) {
        const int fwd = p.stm ? -8 : 8, promo_rank = p.stm ? 1 : 6, start_rank = p.stm ? 6 : 1;
        uint64_t targets = theirs;
        if (p.ep >= 0) targets |= bit(p.ep);
        uint64_t dest = pawn_attacks(from, p.stm) & targets;
        const int f1 = from + fwd;
        if (f1 >= 0 && f1 < 64 && !(occ & bit(f1))) {
            dest |= bit(f1);
            if ((from >> 3) == start_rank && !(occ & bit(f1 + fwd))) dest |= bit(f1 + fwd);
        }
        const unsigned n = (unsigned)__builtin_popcountll(dest);
        if ((from >> 3) == promo_rank) {
            const unsigned id = br->take8(used_bits_safe(n * 4));
            m.type = 1, m.promo = (int)(id % 4), m.to = nth_set(dest, id / 4);
        } else {
            m.to = nth_set(dest, br->take8(used_bits_safe(n)));
            if (m.to >= 0 && m.to == p.ep) m.type = 3;
        }
    } else if (type == KING) {
        const uint64_t att = piece_attacks(KING, from, occ) & ~ours;
        const unsigned n = (unsigned)__builtin_popcountll(att);
        const unsigned rights = (p.castle >> (p.stm ? 2 : 0)) & 3;  // bit 0 short, bit 1 long
        const unsigned id = br->take8(used_bits_safe(n + (unsigned)__builtin_popcount(rights)));
        if (id >= n) {
            const bool longc = (id - n == 0) && (rights & 2);
            m.type = 2, m.to = (p.stm ? 56 : 0) + (longc ? 0 : 7);
        } else {
            m.to = nth_set(att, id);
        }
    } else {
        const uint64_t att = piece_attacks(type, from, occ) & ~ours;
        m.to = nth_set(att, br->take8(used_bits_safe((unsigned)__builtin_popcountll(att))));
    }
    if (m.to < 0) return false;
    *out = m;
    return true;
}

bool unpack_stem(const uint8_t *s, Pos *p, Move *mv, int *score, int *ply, int *result) {
    uint64_t occ = 0;
    for (int i = 0; i < 8; i++) occ = (occ << 8) | s[i];
    std::memset(p->sq, 0xFF, 64);
    p->stm = 0, p->ep = -1, p->castle = 0;
    int i = 0;
    for (uint64_t o = occ; o; o &= o - 1, i++) {
        if (i >= 32) return false;
        const int q = __builtin_ctzll(o);
        const unsigned nib = (s[8 + i / 2] >> (4 * (i & 1))) & 15;
        if (nib < 12) {
            p->sq[q] = (uint8_t)(((nib & 1) << 3) | (nib >> 1));
        } else if (nib == 12) {  // pawn that has just made a double push: rank 4 = white, rank 5 = black
            if ((q >> 3) != 3 && (q >> 3) != 4) return false;
            const int colour = (q >> 3) == 3 ? 0 : 1;
            p->sq[q] = (uint8_t)((colour << 3) | PAWN);
            p->ep = colour ? q + 8 : q - 8;
        } else if (nib == 13 || nib == 14) {
            const int colour = nib - 13;
            p->sq[q] = (uint8_t)((colour << 3) | ROOK);
            if (q == (colour ? 63 : 7)) p->castle |= (uint8_t)(1 << (colour * 2));
            if (q == (colour ? 56 : 0)) p->castle |= (uint8_t)(2 << (colour * 2));
        } else {
            p->sq[q] = (uint8_t)((1 << 3) | KING);
            p->stm = 1;
        }
    }
    const unsigned m = ((unsigned)s[24] << 8) | s[25];
    mv->type = (int)(m >> 14), mv->from = (int)((m >> 8) & 63), mv->to = (int)((m >> 2) & 63), mv->promo = (int)(m & 3);
    *score = unzigzag(((unsigned)s[26] << 8) | s[27]);
    const unsigned pr = ((unsigned)s[28] << 8) | s[29];
    *ply = (int)(pr & 0x3FFF), *result = unzigzag(pr >> 14);
    return true;
}

// convert_to_bulletformat (sfbinpack.rs:23-48) + ChessBoard::from_raw: a side-to-move relative record
bool to_record(const Pos &p, int score, int result, uint8_t *out) {
    uint64_t occ = 0;
    uint8_t pcs[16] = {0};
    int n = 0, own_k = -1, opp_k = -1;
    for (int s = 0; s < 64; s++) {
        const int src = p.stm ? (s ^ 56) : s;
        const uint8_t pc = p.sq[src];
        if (pc == 0xFF) continue;
        if (n >= 32) return false;
        const int type = pc & 7, theirs = (pc >> 3) != p.stm;
        occ |= bit(s);
        pcs[n / 2] |= (uint8_t)(((theirs << 3) | type) << (4 * (n & 1)));
        n++;
        if (type == KING) (theirs ? opp_k : own_k) = s;
    }
    if (own_k < 0 || opp_k < 0 || result < -1 || result > 1) return false;
    const int16_t sc = (int16_t)score;
    std::memcpy(out, &occ, 8);
    std::memcpy(out + 8, pcs, 16);
    std::memcpy(out + 24, &sc, 2);
    out[26] = (uint8_t)(1 + result), out[27] = (uint8_t)own_k, out[28] = (uint8_t)(opp_k ^ 56);
    out[29] = out[30] = out[31] = 0;
    return true;
}

struct SfSource {
    std::vector<std::string> paths;
    size_t buffer_positions;
    int threads;
    bt_sf_filter filter;
    void *user;
    uint64_t rng;
    size_t file_idx = 0;
    FILE *fp = nullptr;
    std::vector<uint8_t> carry;
    ~SfSource() {
        if (fp) std::fclose(fp);
    }
};

// every entry of one chunk (ChunkReader::has_next / next, sfbinpack.rs:190-198)
bool decode_chunk(const std::vector<uint8_t> &c, const SfSource *src, std::vector<uint8_t> *out) {
    size_t off = 0;
    uint8_t rec[32];
    auto emit = [&](const Pos &p, const Move &mv, int score, int ply, int result) -> bool {
        if (!to_record(p, score, result, rec)) return false;
        if (src->filter) {
            bt_sf_entry e;
            e.chessboard = rec, e.score = (int16_t)score, e.ply = (uint16_t)ply, e.result = (int16_t)result;
            e.move_from = (uint8_t)mv.from, e.move_to = (uint8_t)mv.to, e.move_type = (uint8_t)mv.type;
            e.promotion = (uint8_t)mv.promo;
            const uint8_t target = mv.type == 2 ? 0xFF : p.sq[mv.to];  // (castling: `to` holds the own rook)
            e.captured_piece = target == 0xFF ? (uint8_t)NONE : (uint8_t)(target & 7);
            const int ks = king_square(p, p.stm);
            e.in_check = (uint8_t)(ks >= 0 && square_attacked(p, ks, p.stm ^ 1));
            if (!src->filter(src->user, &e)) return true;
        }
        out->insert(out->end(), rec, rec + 32);
        return true;
    };
    while (off + 34 <= c.size()) {
        Pos p;
        Move mv;
        int score, ply, result;
        if (!unpack_stem(c.data() + off, &p, &mv, &score, &ply, &result)) return false;
        off += 32;
        const unsigned plies = ((unsigned)c[off] << 8) | c[off + 1];
        off += 2;
        if (!emit(p, mv, score, ply, result)) return false;
        if (plies == 0) continue;
        BitReader br{c.data() + off, c.size() - off};
        int last = -score;
        for (unsigned k = 0; k < plies; k++) {
            if (p.sq[mv.from] == 0xFF) return false;
            do_move(&p, mv);
            if (!read_move(&br, p, &mv)) return false;
            score = last + unzigzag(br.vle16(4));
            last = -score;
            ply++, result = -result;
            if (br.overrun) return false;
            if (!emit(p, mv, score, ply, result)) return false;
        }
        off += br.bytes_read();
    }
    return off == c.size();
}

// read_chunk_into: "BINP" + u32 LE size + data; false at a clean end of file
int read_chunk(FILE *fp, std::vector<uint8_t> *chunk) {
    uint8_t head[8];
    const size_t got = std::fread(head, 1, 8, fp);
    if (got == 0) return 0;
    if (got != 8 || std::memcmp(head, "BINP", 4) != 0) return -1;
    const uint32_t size = (uint32_t)head[4] | ((uint32_t)head[5] << 8) | ((uint32_t)head[6] << 16) | ((uint32_t)head[7] << 24);
    chunk->resize(size);
    return std::fread(chunk->data(), 1, size, fp) == size ? 1 : -1;
}

uint64_t xorshift(uint64_t *s) {  // SimpleRand::rng, value/loader/rng.rs:13-18
    *s ^= *s << 13, *s ^= *s >> 7, *s ^= *s << 17;
    return *s;
}

int refill(SfSource *src, std::vector<uint8_t> *out) {
    const size_t want = src->buffer_positions * 32;
    out->clear();
    out->swap(src->carry);
    size_t size_at_wrap = (size_t)-1;
    while (out->size() < want) {
        std::vector<std::vector<uint8_t>> chunks;  // `threads` chunks per hand-over (sfbinpack.rs:82,96-105)
        bool wrapped = false;
        while ((int)chunks.size() < src->threads) {
            if (!src->fp) {
                src->fp = std::fopen(src->paths[src->file_idx].c_str(), "rb");
                if (!src->fp) {
                    bt_set_error("File not found: %s", src->paths[src->file_idx].c_str());
                    return BT_ERR_IO;
                }
            }
            std::vector<uint8_t> chunk;
            const int rc = read_chunk(src->fp, &chunk);
            if (rc < 0) {
                bt_set_error("sfbinpack: bad chunk header in %s", src->paths[src->file_idx].c_str());
                return BT_ERR_IO;
            }
            if (rc == 0) {
                std::fclose(src->fp);
                src->fp = nullptr;
                src->file_idx = (src->file_idx + 1) % src->paths.size();
                if (src->file_idx == 0) {
                    wrapped = true;
                    break;
                }
                continue;
            }
            chunks.push_back(std::move(chunk));
        }
        const int workers = (int)std::max<size_t>(1, std::min<size_t>((size_t)src->threads, chunks.size()));
        std::vector<std::vector<uint8_t>> parts(workers);
        std::atomic<bool> bad(false);
        const size_t per = (chunks.size() + workers - 1) / workers;
        auto work = [&](int w) {
            for (size_t i = w * per; i < std::min(chunks.size(), (w + 1) * per); i++)
                if (!decode_chunk(chunks[i], src, &parts[w])) bad.store(true);
        };
        std::vector<std::thread> pool;
        for (int w = 1; w < workers; w++) pool.emplace_back(work, w);
        work(0);
        for (auto &t : pool) t.join();
        if (bad.load()) {
            bt_set_error("Binpack must be malformed!");  // sfbinpack.rs:47
            return BT_ERR_IO;
        }
        for (auto &p : parts) out->insert(out->end(), p.begin(), p.end());
        if (wrapped) {
            if (out->size() == size_at_wrap) {
                bt_set_error("sfbinpack: no position passes the filter (or the files hold no entries)");
                return BT_ERR_IO;
            }
            size_at_wrap = out->size();
        }
    }
    if (out->size() > want) {
        src->carry.assign(out->begin() + want, out->end());
        out->resize(want);
    }
    const size_t n = out->size() / 32;
    uint8_t tmp[32];
    for (size_t i = n; i-- > 0;) {  // sfbinpack.rs:217-224
        const size_t j = (size_t)(xorshift(&src->rng) % (uint64_t)(i + 1));
        if (j == i) continue;
        uint8_t *a = out->data() + 32 * i, *b = out->data() + 32 * j;
        std::memcpy(tmp, a, 32), std::memcpy(a, b, 32), std::memcpy(b, tmp, 32);
    }
    return BT_OK;
}

}  // namespace

int bt_internal_sf_source(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads, bt_sf_filter filter,
                          void *user, uint64_t shuffle_seed, std::function<int(std::vector<uint8_t> &)> *out_refill) {
    BT_REQUIRE(paths && n_paths > 0 && out_refill, "bt_reader_open_sfbinpack: bad arguments");
    BT_REQUIRE(buffer_size_mb > 0, "bt_reader_open_sfbinpack: buffer_size_mb must be positive");
    auto src = std::make_shared<SfSource>();
    for (int i = 0; i < n_paths; i++) {
        FILE *f = std::fopen(paths[i], "rb");
        if (!f) {
            bt_set_error("File not found: %s", paths[i]);
            return BT_ERR_IO;
        }
        std::fclose(f);
        src->paths.push_back(paths[i]);
    }
    src->buffer_positions = buffer_size_mb * 1024 * 1024 / 32 / 2;  // sfbinpack.rs:66
    src->threads = threads > 0 ? threads : 1;
    src->filter = filter, src->user = user;
    if (shuffle_seed == 0) {
        const auto us = std::chrono::duration_cast<std::chrono::microseconds>(
                            std::chrono::system_clock::now().time_since_epoch()).count();
        shuffle_seed = (uint64_t)us & 0xFFFFFFFFull;
        if (shuffle_seed == 0) shuffle_seed = 1;
    }
    src->rng = shuffle_seed;
    *out_refill = [src](std::vector<uint8_t> &out) { return refill(src.get(), &out); };
    return BT_OK;
}
