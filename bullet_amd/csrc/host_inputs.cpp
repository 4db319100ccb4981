// Host batch preparation: 32-byte position records -> the five device batch tensors.
// Mirrors SparseInputType::map_features (crates/bullet_lib/src/game/inputs/{chess768,chess_buckets,
// ataxx147,factorised}.rs), OutputBuckets::bucket (game/outputs.rs:24-31) and the make_mapper closure
// (crates/bullet_lib/src/value.rs:77-117), fanned out over worker threads like
// ModelInputsMapper::map (crates/trainer/src/model/inputs.rs:86-109).  Integer outputs must be bit-exact.
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace {

struct ChessRecord {  // bulletformat ChessBoard; layout as examples/tests/test4.rs:8-18
    uint64_t occ;
    uint8_t pcs[16];
    int16_t score;
    uint8_t result, ksq, opp_ksq, extra[3];
};
struct AtaxxRecord {  // bulletformat AtaxxBoard; layout as examples/ataxx.rs:8-18
    uint64_t bbs[3];
    int16_t score;
    uint8_t result, origstm;
    uint16_t fullm;
    uint8_t halfm, extra;
};
static_assert(sizeof(ChessRecord) == 32 && sizeof(AtaxxRecord) == 32, "records are 32 bytes");

inline bool is_ataxx(const bt_input_desc *d) { return d->kind == BT_INPUT_ATAXX147 || d->kind == BT_INPUT_ATAXX98; }

inline float sigmoidf_host(float x) { return 1.0f / (1.0f + std::exp(-x)); }  // value.rs:66-68

// What the reference's mapper panics on (value.rs:80 "Input feature index exceeded input size!", value.rs:91 "More
// inputs provided than the specified maximum!", the bounds checks of `buckets[ksq]` in chess_buckets.rs:30 and of
// `targets[result]` in value.rs:101, the u8 underflow of MaterialCount in outputs.rs:29) is reported per record here.
enum RecError { REC_TOO_MANY = 1, REC_FEATURE = 2, REC_KSQ = 3, REC_RESULT = 4, REC_TOO_FEW = 5 };
const char *rec_error_text(int code) {
    switch (code) {
    case REC_TOO_MANY: return "More inputs provided than the specified maximum!";
    case REC_FEATURE: return "Input feature index exceeded input size!";
    case REC_KSQ: return "king square out of range";
    case REC_RESULT: return "game result out of range";
    case REC_TOO_FEW: return "fewer than two pieces (output bucket underflows)";
    default: return "invalid record";
    }
}

// returns the number of features written, or -RecError (nothing is ever written past slot nnz - 1)
inline int map_one(const bt_input_desc *d, const uint8_t *rec, int32_t *stm, int32_t *ntm, int nnz, int num) {
    int cnt = 0;
    if (is_ataxx(d)) {
        AtaxxRecord r;
        std::memcpy(&r, rec, 32);
        for (int pc = 0; pc < 3; pc++) {
            for (uint64_t bb = r.bbs[pc]; bb; bb &= bb - 1) {
                const int sq = __builtin_ctzll(bb);
                const int s = 49 * pc + sq;
                if (d->kind == BT_INPUT_ATAXX98 && s >= 98) continue;  // ataxx147.rs:52-56
                const int n = (pc == 2) ? s : 49 * (1 - pc) + sq;
                if (s >= num || n >= num) return -REC_FEATURE;
                if (cnt >= nnz) return -REC_TOO_MANY;
                stm[cnt] = s;
                ntm[cnt] = n;
                cnt++;
            }
        }
        return cnt;
    }
    ChessRecord r;
    std::memcpy(&r, rec, 32);
    int s_flip = 0, n_flip = 0, s_base = 0, n_base = 0;
    if (d->kind != BT_INPUT_CHESS768) {
        if (r.ksq > 63 || r.opp_ksq > 63) return -REC_KSQ;
        s_base = 768 * d->buckets[r.ksq];
        n_base = 768 * d->buckets[r.opp_ksq];
        if (d->kind == BT_INPUT_CHESS_BUCKETS_MIRRORED) {
            s_flip = (r.ksq & 7) > 3 ? 7 : 0;
            n_flip = (r.opp_ksq & 7) > 3 ? 7 : 0;
        }
    }
    if (__builtin_popcountll(r.occ) > 32) return -REC_TOO_MANY;  // the piece list holds 32 nibbles
    const int shift = d->factorised ? 768 : 0;
    int piece_no = 0;
    for (uint64_t occ = r.occ; occ; occ &= occ - 1, piece_no++) {
        const int sq = __builtin_ctzll(occ);
        const int nib = (r.pcs[piece_no >> 1] >> ((piece_no & 1) * 4)) & 15;
        const int side = nib >> 3, type_off = (nib & 7) << 6;
        if (type_off >= 384) return -REC_FEATURE;  // piece types 6, 7 do not exist: the index would leave its 768 block
        const int s = s_base + ((side * 384 + type_off + sq) ^ s_flip);
        const int n = n_base + (((1 - side) * 384 + type_off + (sq ^ 56)) ^ n_flip);
        if (shift + s >= num || shift + n >= num) return -REC_FEATURE;
        if (cnt + (d->factorised ? 2 : 1) > nnz) return -REC_TOO_MANY;
        stm[cnt] = shift + s;
        ntm[cnt] = shift + n;
        cnt++;
        if (d->factorised) {
            stm[cnt] = s % 768;
            ntm[cnt] = n % 768;
            cnt++;
        }
    }
    return cnt;
}

// first bad record of a batch: (index << 8) | code, kept as the minimum over the worker threads
void report_bad(std::atomic<long long> *bad, int index, int code) {
    const long long v = ((long long)index << 8) | code;
    long long cur = bad->load();
    while ((cur < 0 || v < cur) && !bad->compare_exchange_weak(cur, v)) {
    }
}

void map_range(const bt_input_desc *d, const bt_map_opts *o, const uint8_t *records, int lo, int hi, int nnz, int num,
               int32_t *stm, int32_t *ntm, int32_t *buckets, float *targets, float *weights, std::atomic<int> *bad_blend,
               std::atomic<long long> *bad_record) {
    const float rscale = 1.0f / o->eval_scale;
    const bool ataxx = is_ataxx(d);
    for (int k = lo; k < hi; k++) {
        const uint8_t *rec = records + (size_t)k * 32;
        int32_t *s = stm + (size_t)k * nnz, *n = ntm + (size_t)k * nnz;
        const int cnt = map_one(d, rec, s, n, nnz, num);
        // a bad record is reported (the call fails) and left as an empty list with zero weight
        auto reject = [&](int code) {
            report_bad(bad_record, k, code);
            s[0] = n[0] = -1;
            buckets[k] = 0, weights[k] = 0.0f;
            const int tw = o->wdl_output ? 3 : 1;
            for (int j = 0; j < tw; j++) targets[tw * (size_t)k + j] = 0.0f;
        };
        if (cnt < 0) {
            reject(-cnt);
            continue;
        }
        if (cnt < nnz) s[cnt] = n[cnt] = -1;
        int16_t score_i;
        uint8_t result;
        uint64_t occ = 0;
        if (ataxx) {
            AtaxxRecord r;
            std::memcpy(&r, rec, 32);
            score_i = r.score, result = r.result;
        } else {
            ChessRecord r;
            std::memcpy(&r, rec, 32);
            score_i = r.score, result = r.result, occ = r.occ;
        }
        int bucket = 0;
        if (o->out_buckets > 1 && !ataxx) {
            const int divisor = (32 + o->out_buckets - 1) / o->out_buckets;
            const int pieces = __builtin_popcountll(occ);
            if (pieces < 2) {  // outputs.rs:29 computes `popcount - 2` in u8
                reject(REC_TOO_FEW);
                continue;
            }
            bucket = (uint8_t)((uint8_t)pieces - 2) / (uint8_t)divisor;
        }
        // WDL targets: value.rs:101 indexes targets[result] (a panic). Scalar targets: the reference keeps training with
        // result / 2 > 1 (value.rs:112); rejected here too, on purpose (stricter than the reference, corrupt data only)
        if (result > 2) {
            reject(REC_RESULT);
            continue;
        }
        buckets[k] = bucket;
        weights[k] = o->weight_fn ? o->weight_fn(o->fn_user, rec) : 1.0f;  // value.rs:94
        if (o->wdl_output) {
            targets[3 * (size_t)k] = targets[3 * (size_t)k + 1] = targets[3 * (size_t)k + 2] = 0.0f;
            targets[3 * (size_t)k + result] = 1.0f;
        } else {
            float score = (float)score_i;
            if (o->use_win_rate_model) {
                const float p = (score - 270.0f) / 380.0f, pm = (-score - 270.0f) / 380.0f;
                score = 0.5f * (1.0f + sigmoidf_host(p) - sigmoidf_host(pm));
            } else {
                score = sigmoidf_host(rscale * score);
            }
            const float res = (float)result / 2.0f;
            float blend = o->blend;
            if (o->blend_fn) {  // value.rs:113-114
                blend = o->blend_fn(o->fn_user, rec, blend);
                if (!(blend >= 0.0f && blend <= 1.0f)) bad_blend->store(1);
            }
            targets[k] = blend * res + (1.0f - blend) * score;
        }
    }
}

}  // namespace

extern "C" {

int bt_input_init(bt_input_desc *d, int kind, int factorised, const int *layout, int layout_len) {
    BT_REQUIRE(d, "bt_input_init: null descriptor");
    BT_REQUIRE(kind >= BT_INPUT_CHESS768 && kind <= BT_INPUT_ATAXX98, "bt_input_init: unknown input kind %d", kind);
    std::memset(d, 0, sizeof(*d));
    d->kind = kind;
    d->num_buckets = 1;
    const bool ataxx = (kind == BT_INPUT_ATAXX147 || kind == BT_INPUT_ATAXX98);
    BT_REQUIRE(!(factorised && ataxx), "bt_input_init: factoriser is defined for chess inputs only");
    d->factorised = factorised ? 1 : 0;
    if (kind == BT_INPUT_CHESS_BUCKETS) {
        BT_REQUIRE(layout && layout_len == 64, "bt_input_init: ChessBuckets needs a 64-entry layout");
        int mx = 0;
        for (int i = 0; i < 64; i++) {
            BT_REQUIRE(layout[i] >= 0, "bt_input_init: negative bucket");
            d->buckets[i] = layout[i];
            if (layout[i] > mx) mx = layout[i];
        }
        d->num_buckets = mx + 1;
    } else if (kind == BT_INPUT_CHESS_BUCKETS_MIRRORED) {
        BT_REQUIRE(layout && layout_len == 32, "bt_input_init: ChessBucketsMirrored needs a 32-entry layout");
        static const int mirror[8] = {0, 1, 2, 3, 3, 2, 1, 0};
        int mx = 0;
        for (int i = 0; i < 32; i++) {
            BT_REQUIRE(layout[i] >= 0, "bt_input_init: negative bucket");
            if (layout[i] > mx) mx = layout[i];
        }
        for (int sq = 0; sq < 64; sq++) d->buckets[sq] = layout[(sq >> 3) * 4 + mirror[sq & 7]];
        d->num_buckets = mx + 1;
    }
    return BT_OK;
}

int bt_input_num_inputs(const bt_input_desc *d) {
    if (!d) return 0;
    int base = 768;
    switch (d->kind) {
    case BT_INPUT_CHESS768: base = 768; break;
    case BT_INPUT_CHESS_BUCKETS:
    case BT_INPUT_CHESS_BUCKETS_MIRRORED: base = 768 * d->num_buckets; break;
    case BT_INPUT_ATAXX147: base = 147; break;
    case BT_INPUT_ATAXX98: base = 98; break;
    }
    return base + (d->factorised ? 768 : 0);
}

int bt_input_max_active(const bt_input_desc *d) {
    if (!d) return 0;
    const int m = (d->kind == BT_INPUT_ATAXX147 || d->kind == BT_INPUT_ATAXX98) ? 49 : 32;
    return d->factorised ? 2 * m : m;
}

int bt_map_batch(const bt_input_desc *d, const bt_map_opts *o, const void *records, int count, int32_t *stm,
                 int32_t *nstm, int32_t *buckets, float *targets, float *weights) {
    BT_REQUIRE(d && o && records && stm && nstm && buckets && targets && weights, "bt_map_batch: null argument");
    BT_REQUIRE(count >= 0, "bt_map_batch: negative count");
    // value.rs:114: assert!((0.0..=1.0).contains(&blend))
    BT_REQUIRE(o->wdl_output || (o->blend >= 0.0f && o->blend <= 1.0f), "WDL proportion must be in [0, 1]");
    const int nnz = bt_input_max_active(d), num = bt_input_num_inputs(d);
    int threads = o->threads < 1 ? 1 : o->threads;
    if (threads > count) threads = count > 0 ? count : 1;
    const uint8_t *rec = static_cast<const uint8_t *>(records);
    std::atomic<int> bad_blend(0);
    std::atomic<long long> bad_record(-1);
    if (threads == 1) {
        map_range(d, o, rec, 0, count, nnz, num, stm, nstm, buckets, targets, weights, &bad_blend, &bad_record);
    } else {
        // chunks of ceil(count/threads), model/inputs.rs:88-103
        const int chunk = (count + threads - 1) / threads;
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) {
            const int lo = t * chunk, hi = std::min(count, lo + chunk);
            if (lo >= hi) break;
            pool.emplace_back(map_range, d, o, rec, lo, hi, nnz, num, stm, nstm, buckets, targets, weights, &bad_blend,
                              &bad_record);
        }
        for (auto &th : pool) th.join();
    }
    BT_REQUIRE(!bad_blend.load(), "WDL proportion must be in [0, 1]");
    const long long bad = bad_record.load();
    BT_REQUIRE(bad < 0, "bt_map_batch: record %lld: %s", bad >> 8, rec_error_text((int)(bad & 0xff)));
    return BT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ text positions
// `<fen> | <score> | <result>` -> ChessBoard, the FromStr impl of bulletformat 1.8.0 (a crates.io dependency that is
// not in the reference tree: restated from its published behaviour, **parity unpinned** except for what the
// reference's own fixture pins -- side-to-move-relative boards, low-nibble-first piece list, ksq = own king,
// opp_ksq = opponent king ^ 56, result 0/1/2 -- see SURVEY.md Appendix B). Call sites: InMemoryTextLoader
// (crates/bullet_lib/src/value/loader/text.rs:21-35) and eval (crates/bullet_lib/src/value.rs:217).
namespace {
const char *skip_ws(const char *p) {
    while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
    return p;
}
std::string trimmed(const char *a, const char *b) {
    while (a < b && (*a == ' ' || *a == '\t' || *a == '\r' || *a == '\n')) a++;
    while (b > a && (b[-1] == ' ' || b[-1] == '\t' || b[-1] == '\r' || b[-1] == '\n')) b--;
    return std::string(a, b);
}
}  // namespace

extern "C" int bt_chessboard_from_str(const char *line, void *record32) {
    BT_REQUIRE(line && record32, "bt_chessboard_from_str: null argument");
    const char *bar1 = std::strchr(line, '|');
    const char *bar2 = bar1 ? std::strchr(bar1 + 1, '|') : nullptr;
    BT_REQUIRE(bar1 && bar2, "bt_chessboard_from_str: expected `<fen> | <score> | <result>`");
    const char *bar3 = std::strchr(bar2 + 1, '|');
    const std::string score_s = trimmed(bar1 + 1, bar2);
    const std::string res_s = trimmed(bar2 + 1, bar3 ? bar3 : bar2 + 1 + std::strlen(bar2 + 1));
    // score: i16 in centipawns, white-relative
    char *end = nullptr;
    const long score_l = std::strtol(score_s.c_str(), &end, 10);
    BT_REQUIRE(!score_s.empty() && *end == '\0' && score_l >= -32768 && score_l <= 32767,
               "bt_chessboard_from_str: bad score `%s`", score_s.c_str());
    int result2;  // 2 * result, white-relative
    if (res_s == "1.0" || res_s == "[1.0]" || res_s == "1") result2 = 2;
    else if (res_s == "0.5" || res_s == "[0.5]" || res_s == "1/2") result2 = 1;
    else if (res_s == "0.0" || res_s == "[0.0]" || res_s == "0") result2 = 0;
    else {
        bt_set_error("bt_chessboard_from_str: bad game result `%s`", res_s.c_str());
        return BT_ERR_INVALID;
    }
    // board field + side to move
    uint64_t colour_bb[2] = {0, 0}, piece_bb[6] = {0, 0, 0, 0, 0, 0};
    const char *p = skip_ws(line);
    int row = 7, col = 0;
    for (; *p && *p != ' ' && *p != '\t' && p < bar1; p++) {
        const char ch = *p;
        if (ch == '/') {
            row--, col = 0;
        } else if (ch >= '1' && ch <= '8') {
            col += ch - '0';
        } else {
            static const char names[] = "PNBRQKpnbrqk";
            const char *f = std::strchr(names, ch);
            BT_REQUIRE(f && ch != '\0', "bt_chessboard_from_str: bad piece `%c`", ch);
            BT_REQUIRE(row >= 0 && col >= 0 && col < 8, "bt_chessboard_from_str: board field overflows");
            const int id = (int)(f - names);
            const uint64_t bit = 1ull << (8 * row + col);
            colour_bb[id / 6] |= bit;
            piece_bb[id % 6] |= bit;
            col++;
        }
    }
    p = skip_ws(p);
    const bool black_to_move = (*p == 'b');
    int score = (int)score_l;
    if (black_to_move) {  // side-to-move relative: mirror the ranks, swap the colours, negate score and result
        for (auto &bb : colour_bb) bb = __builtin_bswap64(bb);
        for (auto &bb : piece_bb) bb = __builtin_bswap64(bb);
        std::swap(colour_bb[0], colour_bb[1]);
        score = -score;
        result2 = 2 - result2;
    }
    ChessRecord r;
    std::memset(&r, 0, sizeof(r));
    r.occ = colour_bb[0] | colour_bb[1];
    BT_REQUIRE(__builtin_popcountll(r.occ) <= 32, "bt_chessboard_from_str: more than 32 pieces");
    int n = 0;
    for (uint64_t occ = r.occ; occ; occ &= occ - 1, n++) {
        const uint64_t bit = occ & (0 - occ);
        int type = 0;
        while (type < 6 && !(piece_bb[type] & bit)) type++;
        const uint8_t pc = (uint8_t)(((colour_bb[1] & bit) ? 8 : 0) | type);
        r.pcs[n / 2] |= (uint8_t)(pc << (4 * (n & 1)));
    }
    const uint64_t own_k = colour_bb[0] & piece_bb[5], opp_k = colour_bb[1] & piece_bb[5];
    BT_REQUIRE(own_k && opp_k, "bt_chessboard_from_str: a king is missing");
    r.score = (int16_t)score;  // (i16 negation of -32768 wraps in the crate; out of any real range)
    r.result = (uint8_t)result2;
    r.ksq = (uint8_t)__builtin_ctzll(own_k);
    r.opp_ksq = (uint8_t)(__builtin_ctzll(opp_k) ^ 56);
    std::memcpy(record32, &r, 32);
    return BT_OK;
}
