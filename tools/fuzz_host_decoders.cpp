#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <functional>
#include <cstdarg>
#include "/root/repo/bullet_amd/csrc/common.h"
// error plumbing normally provided by runtime.hip
static thread_local char g_err[512];
void bt_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
extern "C" const char *bt_last_error(void) { return g_err; }
int main(int argc, char **argv) {
    std::mt19937_64 rng(123);
    int ok = 0, bad = 0;
    auto slurp = [](const char *p) { std::vector<uint8_t> v; FILE *f = fopen(p, "rb"); int c; while ((c = fgetc(f)) != EOF) v.push_back((uint8_t)c); fclose(f); return v; };
    const std::vector<uint8_t> seed_sf = slurp("/tmp/bt_fuzz/seed.binpack"), seed_vf = slurp("/tmp/bt_fuzz/seed.vf"),
                               seed_mf = slurp("/tmp/bt_fuzz/seed.mf");
    for (int trial = 0; trial < 9000; trial++) {
        const int which = trial % 3;  // 0 viriformat, 1 Stockfish binpack, 2 montyformat
        const bool sf = which == 1;
        std::vector<uint8_t> body = sf ? seed_sf : which == 2 ? seed_mf : seed_vf;
        const int flips = trial < 2 ? 0 : 1 + rng() % 4;
        for (int i = 0; i < flips; i++) {
            size_t at = rng() % body.size();
            if (sf && at < 8) at += 8;  // keep the first chunk header
            body[at] = (trial & 4) ? (uint8_t)rng() : (uint8_t)(body[at] ^ (1u << (rng() % 8)));
        }
        std::string path = "/tmp/bt_fuzz/f.bin";
        FILE *f = fopen(path.c_str(), "wb");
        fwrite(body.data(), 1, body.size(), f);
        fclose(f);
        const char *paths[1] = {path.c_str()};
        std::function<int(std::vector<uint8_t> &)> refill;
        bt_viri_builtin_filter vf;
        bt_viri_filter_default(&vf);
        vf.min_ply = 0;
        int rc = sf ? bt_internal_sf_source(paths, 1, 1, 2, nullptr, nullptr, 1, &refill)
                 : which == 2 ? bt_internal_monty_source(paths, 1, 1, 2, nullptr, nullptr, 1, &refill)
                              : bt_internal_viri_source(paths, 1, 1, 2, nullptr, nullptr, (trial & 8) ? &vf : nullptr, 1, &refill);
        if (rc) { bad++; continue; }
        std::vector<uint8_t> out;
        rc = refill(out);
        (rc ? bad : ok)++;
    }
    // text parser fuzz
    for (int trial = 0; trial < 20000; trial++) {
        char line[128];
        int n = 1 + rng() % 100;
        const char alphabet[] = "rnbqkpRNBQKP12345678/ wb-|.[]0-9x";
        for (int i = 0; i < n; i++) line[i] = alphabet[rng() % (sizeof(alphabet) - 1)];
        line[n] = 0;
        uint8_t rec[32];
        bt_chessboard_from_str(line, rec);
    }
    // plain bulletformat records (bt_map_batch): random 32-byte records and bit-flipped copies of a valid one, every
    // input type, output buffers of exactly count * nnz slots (ASan catches a write past them)
    int rec_ok = 0, rec_bad = 0;
    {
        const int layout64[64] = {0};
        const int layout32[32] = {0};
        uint8_t valid[32] = {0};
        bt_chessboard_from_str("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1 | 10 | 0.5", valid);
        for (int trial = 0; trial < 40000; trial++) {
            const int kind = trial % 5, fact = (kind <= 2) ? (trial / 5) % 2 : 0;
            bt_input_desc d;
            if (bt_input_init(&d, kind, fact, kind == 1 ? layout64 : kind == 2 ? layout32 : nullptr, kind == 1 ? 64 : kind == 2 ? 32 : 0)) return 2;
            const int count = 3, nnz = bt_input_max_active(&d);
            std::vector<uint8_t> recs(32 * count);
            for (int r = 0; r < count; r++) {
                if (trial & 8) {
                    for (int i = 0; i < 32; i++) recs[32 * r + i] = (uint8_t)rng();
                } else {
                    memcpy(&recs[32 * r], valid, 32);
                    for (int i = 0, n = rng() % 4; i < n; i++) recs[32 * r + rng() % 32] ^= (uint8_t)(1u << (rng() % 8));
                }
            }
            bt_map_opts o = {};
            o.out_buckets = (trial & 16) ? 8 : 1, o.blend = 0.5f, o.eval_scale = 400.0f, o.wdl_output = (trial >> 5) & 1, o.threads = 1 + trial % 3;
            std::vector<int32_t> stm((size_t)count * nnz), ntm((size_t)count * nnz), bk(count);
            std::vector<float> tg((size_t)count * 3), w(count);
            const int rc = bt_map_batch(&d, &o, recs.data(), count, stm.data(), ntm.data(), bk.data(), tg.data(), w.data());
            (rc ? rec_bad : rec_ok)++;
            if (!rc) {  // accepted: every listed index and every bucket is in range
                const int num = bt_input_num_inputs(&d);
                for (int r = 0; r < count; r++) {
                    if (bk[r] < 0 || bk[r] >= o.out_buckets) return 3;
                    for (int i = 0; i < nnz && stm[(size_t)r * nnz + i] >= 0; i++)
                        if (stm[(size_t)r * nnz + i] >= num || ntm[(size_t)r * nnz + i] < 0 || ntm[(size_t)r * nnz + i] >= num) return 4;
                }
            }
        }
    }
    printf("ok=%d bad=%d records ok=%d rejected=%d\n", ok, bad, rec_ok, rec_bad);
    return 0;
}
