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
    const std::vector<uint8_t> seed_sf = slurp("/tmp/bt_fuzz/seed.binpack"), seed_vf = slurp("/tmp/bt_fuzz/seed.vf");
    for (int trial = 0; trial < 6000; trial++) {
        const bool sf = trial & 1;
        std::vector<uint8_t> body = sf ? seed_sf : seed_vf;
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
        int rc = sf ? bt_internal_sf_source(paths, 1, 1, 2, nullptr, nullptr, 1, &refill)
                    : bt_internal_viri_source(paths, 1, 1, 2, nullptr, nullptr, 1, &refill);
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
    printf("ok=%d bad=%d\n", ok, bad);
    return 0;
}
