// The pipeline the reference's binpack loaders share (value/loader/viribinpack.rs:57-152, montybinpack.rs:42-131): files
// read in order, round and round; groups of 8192 * threads games decoded on worker threads; positions collected into a
// shuffle buffer of buffer_size_mb * 2^20 / 32 / 2 records, shuffled by the xorshift Fisher-Yates walk of
// value/loader/rng.rs when it is full. A format plugs in two functions: read one game's bytes, replay one game.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

struct bt_game_board {
    uint8_t sq[64];  // colour << 3 | type (P N B R Q K = 0..5), 0xFF = empty
    bool black_to_move;
    int fullmove;
};
// ChessBoard::from_raw: eval and result white-relative in, side-to-move-relative in the 32-byte record; false = not a
// legal record (more than 32 pieces, a missing king)
bool bt_internal_board_to_record(const bt_game_board *b, int16_t eval_white, int wdl_white, uint8_t *out);

struct bt_game_source {
    std::vector<std::string> paths;
    size_t buffer_positions = 0;
    int threads = 1;
    uint64_t rng = 1, skip_counter = 0;
    size_t file_idx = 0;
    FILE *fp = nullptr;
    std::vector<uint8_t> carry;  // positions decoded beyond the last full buffer
    const char *what = "binpack";
    // appends one game's bytes to `games` and its start to `starts`; false at the end of the file (a truncated game is dropped)
    bool (*read_game)(FILE *fp, std::vector<uint8_t> *games, std::vector<size_t> *starts) = nullptr;
    // replays one game, appending the 32-byte records that pass the filter; sets `bad` for a malformed game
    std::function<void(const uint8_t *game, std::vector<uint8_t> *out, std::atomic<bool> *bad, uint64_t *skip_rng)> splat;
    ~bt_game_source() {
        if (fp) std::fclose(fp);
    }
};
int bt_internal_game_source_open(bt_game_source *src, const char *const *paths, int n_paths, size_t buffer_size_mb,
                                 int threads, uint64_t shuffle_seed, const char *who);
int bt_internal_game_refill(bt_game_source *src, std::vector<uint8_t> *out);
