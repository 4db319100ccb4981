// Host side of the training loop: weight init, LR / WDL schedules, the looping fixed-size record reader and
// run::train for one device.  Mirrors crates/trainer/src/{run.rs,reader.rs,reader/fixed_size.rs,
// model/weights.rs} and crates/bullet_lib/src/trainer/schedule/{lr,wdl}.rs.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

// ------------------------------------------------------------------------------------------ weight init
namespace {

// rand_xoshiro 0.8.1: Xoroshiro128Plus, seed_from_u64 = two SplitMix64 outputs
struct Xoroshiro128Plus {
    uint64_t s0, s1;
    explicit Xoroshiro128Plus(uint64_t seed) {
        auto sm = [&seed]() {
            seed += 0x9E3779B97F4A7C15ull;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        s0 = sm();
        s1 = sm();
    }
    uint64_t next() {
        const uint64_t a = s0;
        uint64_t b = s1;
        const uint64_t r = a + b;
        b ^= a;
        s0 = ((a << 24) | (a >> 40)) ^ b ^ (b << 16);
        s1 = (b << 37) | (b >> 27);
        return r;
    }
};

// rand_distr 0.6.0 StandardNormal: 256-layer ziggurat (tables per its ziggurat_tables.py recipe)
struct Ziggurat {
    static constexpr double R = 3.654152885361008796;
    double x[257], f[257];
    Ziggurat() {
        const double v = 0.00492867323399;
        x[0] = v / std::exp(-R * R / 2.0);
        x[1] = R;
        for (int i = 2; i < 256; i++) x[i] = std::sqrt(-2.0 * std::log(v / x[i - 1] + std::exp(-x[i - 1] * x[i - 1] / 2.0)));
        x[256] = 0.0;
        for (int i = 0; i < 257; i++) f[i] = std::exp(-x[i] * x[i] / 2.0);
    }
    static double from_bits(uint64_t mant, int e) {
        const uint64_t b = ((uint64_t)(1023 + e) << 52) | mant;
        double d;
        std::memcpy(&d, &b, 8);
        return d;
    }
    double sample(Xoroshiro128Plus &rng) const {
        const double eps_half = 2.220446049250313e-16 / 2.0;
        for (;;) {
            const uint64_t bits = rng.next();
            const int i = (int)(bits & 0xff);
            const double u = from_bits(bits >> 12, 1) - (3.0 - eps_half);
            const double xv = u * x[i];
            if (std::fabs(xv) < x[i + 1]) return xv;
            if (i == 0) {
                double a = 1.0, b = 0.0;
                while (-2.0 * b < a * a) {
                    const double ua = from_bits(rng.next() >> 12, 0) - (1.0 - eps_half);
                    const double ub = from_bits(rng.next() >> 12, 0) - (1.0 - eps_half);
                    a = std::log(ua) / R;
                    b = std::log(ub);
                }
                return u < 0.0 ? a - R : R - a;
            }
            const double u01 = (double)(rng.next() >> 11) * (1.0 / 9007199254740992.0);
            if (f[i + 1] + (f[i] - f[i + 1]) * u01 < std::exp(-xv * xv / 2.0)) return xv;
        }
    }
};

int dense_in_of(const bt_net_desc *d, int k) {
    if (k == 0) return d->perspectives * (d->pairwise ? d->ft_out / 2 : d->ft_out);
    return (d->dense_out[k - 1] - d->dense_skip[k - 1]) * (d->dense_dual[k - 1] ? 2 : 1);
}

}  // namespace

extern "C" int bt_net_init_weights_host(const bt_net_desc *d, uint64_t seed, int ft_eff, float *flat) {
    BT_REQUIRE(d && flat, "bt_net_init_weights_host: null argument");
    static const Ziggurat zig;
    Xoroshiro128Plus rng(seed);
    const int nt = bt_net_num_tensors(d);
    std::memset(flat, 0, bt_net_param_count(d) * sizeof(float));
    const int nw = 2 + 2 * d->n_dense;  // a factoriser (last tensor) is InitSettings::Zeroed like the biases
    for (int t = 0; t < nw && t < nt; t += 2) {  // even tensors = weights; biases (odd) stay zero and consume no RNG
        const size_t n = bt_net_tensor_size(d, t);
        float *dst = flat + bt_net_tensor_offset(d, t);
        const int in = (t == 0) ? (ft_eff > 0 ? ft_eff : d->num_inputs) : dense_in_of(d, (t - 2) / 2);
        const float stdev = std::sqrt(2.0f / ((float)in * 1.0f));
        for (size_t i = 0; i < n; i++) dst[i] = 0.0f + stdev * (float)zig.sample(rng);
    }
    return BT_OK;
}

// ------------------------------------------------------------------------------------------ schedules
extern "C" float bt_lr_at(const bt_lr_sched *s, int batch, int superbatch) {
    float lr = 0.0f;
    switch (s->kind) {
    case BT_LR_CONSTANT: lr = s->a; break;
    case BT_LR_DROP: lr = superbatch > s->n ? s->a * s->b : s->a; break;
    case BT_LR_STEP: {
        const int steps = (superbatch > 0 ? superbatch - 1 : 0) / s->n;
        lr = s->a * std::pow(s->b, (float)steps);  // powi
    } break;
    case BT_LR_LINEAR:
        if (superbatch >= s->n) lr = s->b;
        else {
            const float lambda = (float)superbatch / (float)s->n;
            lr = s->a + lambda * (s->b - s->a);
        }
        break;
    case BT_LR_COSINE:
        if (superbatch >= s->n) lr = s->b;
        else {
            const float progress = (float)superbatch / (float)s->n;
            const float lambda = 1.0f - 0.5f * (1.0f + std::cos(3.14159265358979323846f * progress));
            lr = s->a + lambda * (s->b - s->a);
        }
        break;
    case BT_LR_EXP:
        if (superbatch >= s->n) lr = s->b;
        else {
            const float lambda = (float)superbatch / (float)s->n;
            lr = s->a * std::pow(s->b / s->a, lambda);
        }
        break;
    }
    if (s->warmup_batches > 0 && superbatch == 1 && batch < s->warmup_batches) lr = lr / (float)(s->warmup_batches - batch);
    return lr;
}

extern "C" float bt_wdl_at(const bt_wdl_sched *s, int batch, int superbatch, int max) {
    float w = s->start;
    if (s->kind == BT_WDL_LINEAR) {
        const int den = (max - 1) > 1 ? (max - 1) : 1;
        const float grad = (s->end - s->start) / (float)den;
        w = s->start + grad * (float)(superbatch - 1);
    }
    if (s->warmup_batches > 0 && superbatch == 1 && batch < s->warmup_batches) w = w / (float)(s->warmup_batches - batch);
    return w;
}

// ------------------------------------------------------------------------------------------ reader
struct bt_reader {
    std::vector<std::string> paths;  // already rotated so that paths[0] is the start file
    size_t file_idx = 0;
    FILE *fp = nullptr;
    uint64_t to_skip = 0;
    // "text:<path>": InMemoryTextLoader (value/loader/text.rs): every line parsed once, then served round and round
    std::vector<uint8_t> mem;
    size_t mem_pos = 0;
    // set: `mem` is refilled from a decoding source when it runs out (viribinpack shuffle buffers) instead of wrapping
    std::function<int(std::vector<uint8_t> &)> refill;
    bool clock_seeded = false;  // a shuffling source opened with shuffle_seed = 0: its order differs from process to process
};

namespace {
bool file_records(const std::string &p, uint64_t *n) {
    FILE *f = std::fopen(p.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fclose(f);
    *n = (uint64_t)sz / 32;
    return true;
}
}  // namespace

extern "C" int bt_reader_open(const char *const *paths, int n_paths, uint64_t start, bt_reader **out) {
    BT_REQUIRE(paths && n_paths > 0 && out, "bt_reader_open: bad arguments");
    if (std::strncmp(paths[0], "text:", 5) == 0) {
        BT_REQUIRE(n_paths == 1, "bt_reader_open: a text loader takes one file");  // InMemoryTextLoader::new(file_path)
        FILE *f = std::fopen(paths[0] + 5, "r");
        if (!f) {
            bt_set_error("File not found: %s", paths[0] + 5);
            return BT_ERR_IO;
        }
        bt_reader *r = new bt_reader();
        std::string line;
        int c, rc = BT_OK;
        size_t lineno = 0;
        auto flush = [&]() {
            lineno++;
            if (!line.empty() && line.back() == '\r') line.pop_back();  // BufRead::lines strips "\r\n"
            r->mem.resize(r->mem.size() + 32);
            rc = bt_chessboard_from_str(line.c_str(), r->mem.data() + r->mem.size() - 32);
            if (rc) {
                const std::string why = bt_last_error();
                bt_set_error("%s: line %zu: %s", paths[0] + 5, lineno, why.c_str());  // the reference unwraps (panics)
            }
            line.clear();
        };
        while (rc == BT_OK && (c = std::fgetc(f)) != EOF) {
            if (c == '\n') flush();
            else line.push_back((char)c);
        }
        if (rc == BT_OK && !line.empty()) flush();
        std::fclose(f);
        if (rc == BT_OK && r->mem.empty()) {
            bt_set_error("bt_reader_open: no positions in %s", paths[0] + 5);
            rc = BT_ERR_IO;
        }
        if (rc) {
            delete r;
            return rc;
        }
        *out = r;  // `start` is ignored, as read_chunks(&self, _: usize, ..) does (text.rs:25)
        return BT_OK;
    }
    std::vector<std::string> ps;
    std::vector<uint64_t> counts;
    uint64_t epoch = 0;
    for (int i = 0; i < n_paths; i++) {
        uint64_t n = 0;
        if (!file_records(paths[i], &n)) {
            bt_set_error("File not found: %s", paths[i]);  // fixed_size.rs:28
            return BT_ERR_IO;
        }
        ps.push_back(paths[i]);
        counts.push_back(n);
        epoch += n;
    }
    if (epoch == 0) {
        bt_set_error("bt_reader_open: no records in the given files");
        return BT_ERR_IO;
    }
    start %= epoch;  // fixed_size.rs:48
    size_t first = 0;
    uint64_t before = 0;
    for (size_t i = 0; i < ps.size(); i++) {  // fixed_size.rs:50-64
        if (start < before + counts[i]) {
            first = i;
            break;
        }
        before += counts[i];
    }
    bt_reader *r = new bt_reader();
    for (size_t i = 0; i < ps.size(); i++) r->paths.push_back(ps[(first + i) % ps.size()]);  // rotate_left
    r->to_skip = start - before;
    *out = r;
    return BT_OK;
}

extern "C" int bt_reader_next(bt_reader *r, void *dst, int count) {
    BT_REQUIRE(r && count >= 0, "bt_reader_next: bad arguments");
    uint8_t *p = static_cast<uint8_t *>(dst);  // NULL: the records are skipped (a data-parallel rank's foreign shards)
    size_t need = (size_t)count;
    if (r->refill) {  // one shuffle buffer after the other (viribinpack.rs:143-148)
        while (need > 0) {
            if (r->mem_pos * 32 >= r->mem.size()) {
                if (int rc = r->refill(r->mem)) return rc;
                r->mem_pos = 0;
                if (r->mem.empty()) {
                    bt_set_error("bt_reader_next: the source delivered no positions");
                    return BT_ERR_IO;
                }
            }
            const size_t take = std::min(need, r->mem.size() / 32 - r->mem_pos);
            if (p) std::memcpy(p, r->mem.data() + r->mem_pos * 32, take * 32), p += take * 32;
            need -= take, r->mem_pos += take;
        }
        return BT_OK;
    }
    if (!r->mem.empty()) {  // in-memory text positions: the whole set again and again (text.rs:30-34)
        while (need > 0) {
            const size_t take = std::min(need, r->mem.size() / 32 - r->mem_pos);
            if (p) std::memcpy(p, r->mem.data() + r->mem_pos * 32, take * 32), p += take * 32;
            need -= take;
            r->mem_pos = (r->mem_pos + take) % (r->mem.size() / 32);
        }
        return BT_OK;
    }
    int empty_opens = 0;
    while (need > 0) {
        if (!r->fp) {
            r->fp = std::fopen(r->paths[r->file_idx].c_str(), "rb");
            if (!r->fp) {
                bt_set_error("File not found: %s", r->paths[r->file_idx].c_str());
                return BT_ERR_IO;
            }
            if (r->to_skip > 0) {  // fixed_size.rs:80-84: only the first file opened is seeked
                std::fseek(r->fp, (long)(r->to_skip * 32), SEEK_CUR);
                r->to_skip = 0;
            }
        }
        size_t got;
        if (p) {
            got = std::fread(p, 32, need, r->fp);
            p += got * 32;
        } else {  // skip: whole records up to the end of this file
            const long cur = std::ftell(r->fp);
            std::fseek(r->fp, 0, SEEK_END);
            const long end = std::ftell(r->fp);
            got = std::min(need, (size_t)((end - cur) / 32));
            std::fseek(r->fp, cur + (long)(got * 32), SEEK_SET);
        }
        need -= got;
        if (need > 0) {  // end of this file: next file, wrapping forever (fixed_size.rs:73-106)
            std::fclose(r->fp);
            r->fp = nullptr;
            r->file_idx = (r->file_idx + 1) % r->paths.size();
            empty_opens = got == 0 ? empty_opens + 1 : 0;
            if (empty_opens > (int)r->paths.size() + 1) {
                bt_set_error("bt_reader_next: data files became empty");
                return BT_ERR_IO;
            }
        }
    }
    return BT_OK;
}

extern "C" int bt_reader_open_viribinpack(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                                          bt_viri_filter filter, void *user, uint64_t shuffle_seed, bt_reader **out) {
    BT_REQUIRE(out, "bt_reader_open_viribinpack: bad arguments");
    bt_reader *r = new bt_reader();
    if (int rc = bt_internal_viri_source(paths, n_paths, buffer_size_mb, threads, filter, user, nullptr, shuffle_seed,
                                         &r->refill)) {
        delete r;
        return rc;
    }
    r->clock_seeded = shuffle_seed == 0;
    *out = r;
    return BT_OK;
}

extern "C" int bt_reader_open_viribinpack_builtin(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                                                  const bt_viri_builtin_filter *filter, uint64_t shuffle_seed,
                                                  bt_reader **out) {
    BT_REQUIRE(out && filter, "bt_reader_open_viribinpack_builtin: bad arguments");
    bt_reader *r = new bt_reader();
    if (int rc = bt_internal_viri_source(paths, n_paths, buffer_size_mb, threads, nullptr, nullptr, filter, shuffle_seed,
                                         &r->refill)) {
        delete r;
        return rc;
    }
    r->clock_seeded = shuffle_seed == 0;
    *out = r;
    return BT_OK;
}

extern "C" int bt_reader_open_montybinpack(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                                           bt_monty_filter filter, void *user, uint64_t shuffle_seed, bt_reader **out) {
    BT_REQUIRE(out, "bt_reader_open_montybinpack: bad arguments");
    bt_reader *r = new bt_reader();
    if (int rc = bt_internal_monty_source(paths, n_paths, buffer_size_mb, threads, filter, user, shuffle_seed, &r->refill)) {
        delete r;
        return rc;
    }
    r->clock_seeded = shuffle_seed == 0;
    *out = r;
    return BT_OK;
}

extern "C" int bt_reader_open_sfbinpack(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                                        bt_sf_filter filter, void *user, uint64_t shuffle_seed, bt_reader **out) {
    BT_REQUIRE(out, "bt_reader_open_sfbinpack: bad arguments");
    bt_reader *r = new bt_reader();
    if (int rc = bt_internal_sf_source(paths, n_paths, buffer_size_mb, threads, filter, user, shuffle_seed, &r->refill)) {
        delete r;
        return rc;
    }
    r->clock_seeded = shuffle_seed == 0;
    *out = r;
    return BT_OK;
}

extern "C" int bt_reader_close(bt_reader *r) {
    if (!r) return BT_OK;
    if (r->fp) std::fclose(r->fp);
    delete r;
    return BT_OK;
}

// ------------------------------------------------------------------------------------------ loader throughput
// run::measure_max_cpu_throughput (crates/trainer/src/run.rs:47-73, ValueTrainer::measure_max_cpu_throughput,
// crates/bullet_lib/src/value.rs:257-272): the reader + threaded host mapper alone, over the whole schedule.
extern "C" int bt_measure_loader_throughput(const bt_input_desc *input, const bt_train_config *cfg,
                                            const char *const *paths, int n_paths, double *positions_per_sec) {
    BT_REQUIRE(input && cfg && paths, "bt_measure_loader_throughput: null argument");
    BT_REQUIRE(cfg->batch_size > 0 && cfg->batches_per_superbatch > 0 && cfg->start_superbatch >= 1 &&
                   cfg->end_superbatch >= cfg->start_superbatch,
               "bt_measure_loader_throughput: invalid TrainingSteps");
    const int B = cfg->batch_size, nnz = bt_input_max_active(input), bps = cfg->batches_per_superbatch;
    const size_t tw = cfg->wdl_output ? 3 : 1;
    bt_reader *reader = nullptr;
    const uint64_t start_pos = (uint64_t)B * (uint64_t)bps * (uint64_t)(cfg->start_superbatch - 1);
    if (int rc = bt_reader_open(paths, n_paths, start_pos, &reader)) return rc;
    std::vector<uint8_t> records((size_t)B * 32);
    std::vector<int32_t> stm((size_t)B * nnz), nstm((size_t)B * nnz), buckets(B);
    std::vector<float> targets((size_t)B * tw), weights(B);
    using clock = std::chrono::steady_clock;
    auto secs = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t0 = clock::now();
    auto sb_timer = t0;
    long long total = 0;
    if (cfg->log_rate > 0) std::printf("Measuring CPU Throughput\n");
    for (int sb = cfg->start_superbatch; sb <= cfg->end_superbatch; sb++) {
        for (int batch = 0; batch < bps; batch++) {
            bt_map_opts o = {};
            o.blend_fn = cfg->blend_fn, o.weight_fn = cfg->weight_fn, o.fn_user = cfg->fn_user;
            o.out_buckets = cfg->out_buckets, o.blend = (cfg->wdl_fn ? cfg->wdl_fn(cfg->sched_user, batch, sb, cfg->end_superbatch) : bt_wdl_at(&cfg->wdl, batch, sb, cfg->end_superbatch));
            o.eval_scale = cfg->eval_scale, o.use_win_rate_model = cfg->use_win_rate_model;
            o.wdl_output = cfg->wdl_output, o.threads = cfg->threads;
            int rc = bt_reader_next(reader, records.data(), B);
            if (!rc)
                rc = bt_map_batch(input, &o, records.data(), B, stm.data(), nstm.data(), buckets.data(), targets.data(),
                                  weights.data());
            if (rc) {
                bt_reader_close(reader);
                return rc;
            }
            total += B;
        }
        const auto now = clock::now();
        if (cfg->log_rate > 0)
            std::printf("superbatch %d | time %.1fs | %.0f pos/sec | total time %.1fs\n", sb, secs(sb_timer, now),
                        (double)B * bps / secs(sb_timer, now), secs(t0, now));
        sb_timer = now;
    }
    if (positions_per_sec) *positions_per_sec = (double)total / secs(t0, clock::now());
    bt_reader_close(reader);
    return BT_OK;
}

// ------------------------------------------------------------------------------------------ training loop
namespace {

struct HostSlot {  // one prepared batch in pinned memory (PreparedBatchHost)
    int32_t *stm = nullptr, *nstm = nullptr, *buckets = nullptr;
    float *targets = nullptr, *weights = nullptr;
    uint8_t *records = nullptr;  // device_mapper: the raw records instead of the mapped tensors
    float blend = 0.0f;
    int superbatch = 0, batch = 0;
};

struct DevSet {
    int32_t *stm = nullptr, *nstm = nullptr, *buckets = nullptr;
    float *targets = nullptr, *weights = nullptr;
    uint8_t *records = nullptr;
};

struct Loader {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> ready, free_slots;
    bool stop = false, failed = false;
    std::string error;
};

int pinned(void **p, size_t bytes) { return bt_host_alloc_pinned(p, bytes); }

}  // namespace

extern "C" int bt_train_run(bt_net *net, const bt_input_desc *input, const bt_train_config *cfg,
                            const char *const *paths, int n_paths, bt_batch_callback on_batch,
                            bt_superbatch_callback on_superbatch, void *user) {
    BT_REQUIRE(net && input && cfg, "bt_train_run: null argument");
    BT_REQUIRE(cfg->batch_size > 0 && cfg->batches_per_superbatch > 0 && cfg->start_superbatch >= 1 &&
                   cfg->end_superbatch >= cfg->start_superbatch,
               "bt_train_run: invalid TrainingSteps");
    // data parallel (bt_train_config.dp_*): BG = the global batch every rank reads, B = this rank's shard of it
    const int BG = cfg->batch_size, nnz = bt_input_max_active(input);
    const int dp_world = cfg->dp_world > 1 ? cfg->dp_world : 1, dp_rank = dp_world > 1 ? cfg->dp_rank : 0;
    BT_REQUIRE(dp_rank >= 0 && dp_rank < dp_world, "bt_train_run: dp_rank out of range");
    bt_comm *comm = static_cast<bt_comm *>(cfg->comm);
    BT_REQUIRE(dp_world == 1 || cfg->grad_sync || comm, "bt_train_run: data-parallel training needs comm or grad_sync");
    BT_REQUIRE(!comm || (bt_comm_world(comm) == dp_world && bt_comm_rank(comm) == dp_rank),
               "bt_train_run: comm has rank %d of %d, the configuration says %d of %d", bt_comm_rank(comm),
               bt_comm_world(comm), dp_rank, dp_world);
    const int shard_base = BG / dp_world, shard_rem = BG % dp_world;
    const int lo = dp_rank * shard_base + (dp_rank < shard_rem ? dp_rank : shard_rem);
    const int B = shard_base + (dp_rank < shard_rem ? 1 : 0);
    BT_REQUIRE(B > 0, "bt_train_run: batch_size smaller than dp_world");
    BT_REQUIRE(B <= bt_net_max_batch(net), "bt_train_run: Mismatched sizes! (net max_batch < batch shard)");
    const bool dp = dp_world > 1 || comm != nullptr;  // (a communicator of one rank still goes through the exchange)
    const bool loss_slot = dp && (cfg->dp_loss_slot || bt_net_grads_owned(net));
    BT_REQUIRE(!(dp && comm) || loss_slot, "bt_train_run: comm needs the loss slot behind the gradient (dp_loss_slot)");
    const size_t tw = cfg->wdl_output ? 3 : 1;  // targets per position (value.rs:74)
    const int bps = cfg->batches_per_superbatch;
    BT_REQUIRE(cfg->stop_superbatch <= 0 ||
                   (cfg->stop_superbatch >= cfg->start_superbatch && cfg->stop_superbatch <= cfg->end_superbatch),
               "bt_train_run: stop_superbatch outside [start_superbatch, end_superbatch]");
    const int last_superbatch = cfg->stop_superbatch > 0 ? cfg->stop_superbatch : cfg->end_superbatch;
    const long long total_steps = (long long)(last_superbatch - cfg->start_superbatch + 1) * bps;

    BT_REQUIRE(!(cfg->device_mapper && (cfg->blend_fn || cfg->weight_fn)),
               "bt_train_run: per-position closures need the host mapper (device_mapper = 0)");
    const bool dev_map = cfg->device_mapper != 0;

    // reader.rs:40: skip batch_size * total_batches(start) records (resume)
    bt_reader *reader = static_cast<bt_reader *>(cfg->reader);  // caller's reader (e.g. a binpack source), kept open
    const bool own_reader = reader == nullptr;
    // data parallel: the ranks' shards partition ONE global batch only if every rank walks the same record stream
    BT_REQUIRE(dp_world == 1 || own_reader || !reader->clock_seeded,
               "bt_train_run: data-parallel training over a shuffling reader needs an explicit shuffle_seed "
               "(the same on every rank), not 0 = seeded from the clock");
    const uint64_t start_pos = (uint64_t)BG * (uint64_t)bps * (uint64_t)(cfg->start_superbatch - 1);
    if (own_reader) {
        BT_REQUIRE(paths, "bt_train_run: no data files and no reader");
        if (int rc = bt_reader_open(paths, n_paths, start_pos, &reader)) return rc;
    }

    constexpr int RING = 4;
    HostSlot slots[RING];
    // device batch sets: one being read by the step, one being filled, one spare -- the set a copy overwrites was
    // last read TWO steps ago, and the host (never more than LOSS_RING - 1 steps ahead anyway) waits for that step
    // itself. Making the copy stream wait for the compute stream on the device instead cost ~0.05-0.12 ms per step.
    constexpr int SETS = 3;
    DevSet dev[SETS];
    void *copy_stream = nullptr, *compute_stream = nullptr;
    hipEvent_t uploaded[SETS] = {nullptr}, consumed[SETS] = {nullptr}, slot_copied[RING] = {nullptr};
    constexpr int LOSS_RING = 4;
    hipEvent_t loss_ready[LOSS_RING] = {nullptr};
    float lr_slot[LOSS_RING] = {0.0f};
    bool lr_slot_set[LOSS_RING] = {false};
    float *loss_host = nullptr, *lr_host = nullptr, *loss_dev = nullptr, *lr_dev = nullptr, *adj_dev = nullptr;
    std::vector<uint8_t> records((size_t)B * 32);  // (host mapper: this rank's shard of a batch)
    int rc = BT_OK;
    Loader L;
    std::thread loader_thread;

    auto cleanup = [&]() {
        {
            std::lock_guard<std::mutex> g(L.mu);
            L.stop = true;
        }
        L.cv.notify_all();
        if (loader_thread.joinable()) loader_thread.join();
        (void)hipDeviceSynchronize();
        for (auto &s : slots) {
            bt_host_free_pinned(s.stm), bt_host_free_pinned(s.nstm), bt_host_free_pinned(s.buckets);
            bt_host_free_pinned(s.targets), bt_host_free_pinned(s.weights), bt_host_free_pinned(s.records);
        }
        for (auto &d : dev)
            bt_free(d.stm), bt_free(d.nstm), bt_free(d.buckets), bt_free(d.targets), bt_free(d.weights), bt_free(d.records);
        for (auto e : uploaded) if (e) (void)hipEventDestroy(e);
        for (auto e : consumed) if (e) (void)hipEventDestroy(e);
        for (auto e : slot_copied) if (e) (void)hipEventDestroy(e);
        for (auto e : loss_ready) if (e) (void)hipEventDestroy(e);
        bt_host_free_pinned(loss_host), bt_host_free_pinned(lr_host);
        bt_free(loss_dev), bt_free(lr_dev), bt_free(adj_dev);
        bt_stream_destroy(copy_stream), bt_stream_destroy(compute_stream);
        if (own_reader) bt_reader_close(reader);
    };
#define TRY(x)                \
    do {                      \
        if ((rc = (x))) {     \
            cleanup();        \
            return rc;        \
        }                     \
    } while (0)
#define TRY_HIP(x)                                                                   \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            bt_set_error("%s: %s", hipGetErrorName(e_), hipGetErrorString(e_));      \
            cleanup();                                                               \
            return BT_ERR_HIP;                                                       \
        }                                                                            \
    } while (0)

    for (auto &s : slots) {
        if (dev_map) {
            TRY(pinned((void **)&s.records, (size_t)B * 32));
            continue;
        }
        TRY(pinned((void **)&s.stm, sizeof(int32_t) * (size_t)B * nnz));
        TRY(pinned((void **)&s.nstm, sizeof(int32_t) * (size_t)B * nnz));
        TRY(pinned((void **)&s.buckets, sizeof(int32_t) * (size_t)B));
        TRY(pinned((void **)&s.targets, sizeof(float) * (size_t)B * tw));
        TRY(pinned((void **)&s.weights, sizeof(float) * (size_t)B));
        std::memset(s.stm, 0, sizeof(int32_t) * (size_t)B * nnz);
        std::memset(s.nstm, 0, sizeof(int32_t) * (size_t)B * nnz);
    }
    for (auto &d : dev) {
        TRY(bt_malloc((void **)&d.stm, sizeof(int32_t) * (size_t)B * nnz));
        TRY(bt_malloc((void **)&d.nstm, sizeof(int32_t) * (size_t)B * nnz));
        TRY(bt_malloc((void **)&d.buckets, sizeof(int32_t) * (size_t)B));
        TRY(bt_malloc((void **)&d.targets, sizeof(float) * (size_t)B * tw));
        TRY(bt_malloc((void **)&d.weights, sizeof(float) * (size_t)B));
        if (dev_map) {
            TRY(bt_malloc((void **)&d.records, (size_t)B * 32));
            TRY_HIP(hipMemset(d.stm, 0, sizeof(int32_t) * (size_t)B * nnz));  // (slots behind a terminator are never written)
            TRY_HIP(hipMemset(d.nstm, 0, sizeof(int32_t) * (size_t)B * nnz));
            TRY_HIP(hipStreamSynchronize(nullptr));  // NULL-stream fills do not order against the non-blocking copy stream
        }
    }
    TRY(bt_stream_create(&copy_stream));
    TRY(bt_stream_create(&compute_stream));
    for (auto &e : uploaded) TRY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : consumed) TRY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : slot_copied) TRY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : loss_ready) TRY_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    TRY(pinned((void **)&loss_host, sizeof(float) * LOSS_RING));
    TRY(pinned((void **)&lr_host, sizeof(float) * LOSS_RING));
    TRY(bt_malloc((void **)&loss_dev, sizeof(float) * LOSS_RING));
    TRY(bt_malloc((void **)&lr_dev, sizeof(float) * LOSS_RING));
    TRY(bt_malloc((void **)&adj_dev, sizeof(float)));
    {
        const float adj = 1.0f / (float)BG;  // run.rs:130 (the GLOBAL batch when data parallel)
        TRY(bt_memcpy_h2d_async(adj_dev, &adj, sizeof(float), compute_stream));
        TRY(bt_stream_sync(compute_stream));
    }

    auto map_opts = [&](float blend) {
        bt_map_opts o = {};
        o.blend_fn = cfg->blend_fn, o.weight_fn = cfg->weight_fn, o.fn_user = cfg->fn_user;
        o.out_buckets = cfg->out_buckets;
        o.blend = blend;
        o.eval_scale = cfg->eval_scale;
        o.use_win_rate_model = cfg->use_win_rate_model;
        o.wdl_output = cfg->wdl_output;
        o.threads = cfg->threads;
        return o;
    };
    // ---- loader thread (run.rs:94-103): reader -> mapper -> queue of prepared batches
    for (int i = 0; i < RING; i++) L.free_slots.push_back(i);
    loader_thread = std::thread([&]() {
        int sb = cfg->start_superbatch, batch = 0;
        for (long long step = 0; step < total_steps; step++) {
            int slot;
            {
                std::unique_lock<std::mutex> g(L.mu);
                L.cv.wait(g, [&] { return L.stop || !L.free_slots.empty(); });
                if (L.stop) return;
                slot = L.free_slots.front();
                L.free_slots.pop_front();
            }
            bt_map_opts o = map_opts((cfg->wdl_fn ? cfg->wdl_fn(cfg->sched_user, batch, sb, cfg->end_superbatch) : bt_wdl_at(&cfg->wdl, batch, sb, cfg->end_superbatch)));
            HostSlot &s = slots[slot];
            s.blend = o.blend;
            int lrc;
            // data parallel: every rank walks the same record stream and keeps its shard [lo, lo + B) of each global
            // batch; the other ranks' records are skipped (a seek for plain files), not read
            auto read_shard = [&](uint8_t *dst) -> int {
                if (!dp) return bt_reader_next(reader, dst, B);
                if (int r2 = bt_reader_next(reader, nullptr, lo)) return r2;
                if (int r2 = bt_reader_next(reader, dst, B)) return r2;
                return bt_reader_next(reader, nullptr, BG - lo - B);
            };
            if (dev_map) {  // raw records straight into the pinned slot; the device maps them after the upload
                lrc = read_shard(s.records);
                if (!lrc && !o.wdl_output && !(o.blend >= 0.0f && o.blend <= 1.0f)) {
                    bt_set_error("WDL proportion must be in [0, 1]");  // value.rs:114
                    lrc = BT_ERR_INVALID;
                }
            } else {
                lrc = read_shard(records.data());
                if (!lrc) lrc = bt_map_batch(input, &o, records.data(), B, s.stm, s.nstm, s.buckets, s.targets, s.weights);
            }
            s.superbatch = sb, s.batch = batch;
            {
                std::lock_guard<std::mutex> g(L.mu);
                if (lrc) {
                    L.failed = true;
                    L.error = bt_last_error();
                } else {
                    L.ready.push_back(slot);
                }
            }
            L.cv.notify_all();
            if (lrc) return;
            if (++batch == bps) batch = 0, sb++;  // Step::step, run/schedule.rs:70-77
        }
    });

    auto take_ready = [&](int *slot) -> int {
        std::unique_lock<std::mutex> g(L.mu);
        L.cv.wait(g, [&] { return L.failed || !L.ready.empty(); });
        if (L.failed) {
            bt_set_error("data loading failed: %s", L.error.c_str());
            return BT_ERR_IO;
        }
        *slot = L.ready.front();
        L.ready.pop_front();
        return BT_OK;
    };
    auto upload = [&](int slot, int set) -> int {  // PreparedBatchHost::copy_to_device_async, dataloader.rs:41-54
        HostSlot &s = slots[slot];
        DevSet &d = dev[set];
        if (dev_map) {
            if (int r2 = bt_memcpy_h2d_async(d.records, s.records, (size_t)B * 32, copy_stream)) return r2;
            BT_HIP(hipEventRecord(slot_copied[slot], bt_stream(copy_stream)));
            const bt_map_opts o = map_opts(s.blend);
            if (int r2 = bt_map_batch_device(copy_stream, input, &o, d.records, B, d.stm, d.nstm, d.buckets, d.targets,
                                             d.weights))
                return r2;
            BT_HIP(hipEventRecord(uploaded[set], bt_stream(copy_stream)));
            return BT_OK;
        }
        if (int r2 = bt_memcpy_h2d_async(d.stm, s.stm, sizeof(int32_t) * (size_t)B * nnz, copy_stream)) return r2;
        if (int r2 = bt_memcpy_h2d_async(d.nstm, s.nstm, sizeof(int32_t) * (size_t)B * nnz, copy_stream)) return r2;
        if (int r2 = bt_memcpy_h2d_async(d.buckets, s.buckets, sizeof(int32_t) * (size_t)B, copy_stream)) return r2;
        if (int r2 = bt_memcpy_h2d_async(d.targets, s.targets, sizeof(float) * (size_t)B * tw, copy_stream)) return r2;
        if (int r2 = bt_memcpy_h2d_async(d.weights, s.weights, sizeof(float) * (size_t)B, copy_stream)) return r2;
        BT_HIP(hipEventRecord(slot_copied[slot], bt_stream(copy_stream)));
        BT_HIP(hipEventRecord(uploaded[set], bt_stream(copy_stream)));
        return BT_OK;
    };
    auto release_slot = [&](int slot) -> int {
        BT_HIP(hipEventSynchronize(slot_copied[slot]));
        {
            std::lock_guard<std::mutex> g(L.mu);
            L.free_slots.push_back(slot);
        }
        L.cv.notify_all();
        return BT_OK;
    };

    using clock = std::chrono::steady_clock;
    const auto t_start = clock::now();
    auto sb_timer = clock::now();
    auto secs = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<float>(b - a).count(); };

    TRY(bt_internal_net_enable_hints(net));  // the loop below hints every next batch: allocate its buffer set up front
    int cur_slot = -1, prev_slot = -1;
    TRY(take_ready(&cur_slot));
    TRY(upload(cur_slot, 0));
    int sb = cfg->start_superbatch, batch = 0;
    float running_loss = 0.0f;
    long long sb_positions = 0;
    struct Pending {
        int sb, batch;
    };
    std::deque<Pending> pending;  // losses in flight, delivered in order
    auto deliver = [&](bool flush) -> int {
        while (!pending.empty() && (flush || (int)pending.size() >= LOSS_RING - 1)) {
            const Pending p = pending.front();
            const long long k = (long long)(p.sb - cfg->start_superbatch) * bps + p.batch;
            BT_HIP(hipEventSynchronize(loss_ready[k % LOSS_RING]));
            const float error = loss_host[k % LOSS_RING] / (float)BG;  // run.rs:221
            pending.pop_front();
            running_loss += error;
            sb_positions += BG;
            if (on_batch) on_batch(user, p.sb, p.batch, error);
            if (cfg->log_rate > 0 && p.batch % cfg->log_rate == 0 && p.batch > 0) {
                const float t = secs(sb_timer, clock::now());
                std::printf("superbatch %d [%.1f%% (%d/%d batches, %.0f pos/sec)]\n", p.sb, 100.0f * p.batch / bps, p.batch,
                            bps, sb_positions / t);
            }
            if (p.batch == bps - 1) {  // run.rs:232-247
                const float t = secs(sb_timer, clock::now());
                const float err = running_loss / (float)bps;
                if (cfg->log_rate > 0)
                    std::printf("superbatch %d | time %.1fs | running loss %.6f | %.0f pos/sec | total time %.1fs\n", p.sb,
                                t, err, sb_positions / t, secs(t_start, clock::now()));
                if (on_superbatch) on_superbatch(user, p.sb, err, sb_positions / t);
                running_loss = 0.0f, sb_positions = 0, sb_timer = clock::now();
                if (dev_map) {  // records the device mapper had to reject (the host mapper fails the batch itself)
                    int bad = 0;
                    if (int r2 = bt_map_batch_device_errors(copy_stream, &bad)) return r2;
                    BT_REQUIRE(bad == 0, "bt_train_run: %d corrupt record(s) in superbatch %d (device mapper)", bad, p.sb);
                }
            }
        }
        return BT_OK;
    };

    auto dev_batch = [&](int set) {
        bt_batch bb;
        bb.batch = B;
        bb.stm = dev[set].stm, bb.nstm = dev[set].nstm, bb.buckets = dev[set].buckets;
        bb.targets = dev[set].targets, bb.entry_weights = dev[set].weights;
        return bb;
    };
    int next_slot = -1;
    for (long long step = 0; step < total_steps; step++) {
        const int set = (int)(step % SETS), nset = (int)((step + 1) % SETS);
        const int ring = (int)(step % LOSS_RING);
        // next batch: upload into the following device set while this step computes (run.rs:194-207). It is enqueued
        // BEFORE this step, so that the net can be told about it (its index structures are then built during this
        // step); that set was last read by the step two before this one (`consumed`), which the host waits for.
        if (step + 1 < total_steps) {
            TRY(take_ready(&next_slot));
            if (step >= SETS - 1) TRY_HIP(hipEventSynchronize(consumed[nset]));
            TRY(upload(next_slot, nset));
            const bt_batch nb = dev_batch(nset);
            TRY(bt_net_hint_next_batch(net, &nb, uploaded[nset]));
        }
        // compute waits for this batch's upload -- no wait packet when it has landed already (the usual case: it was
        // enqueued a whole step ago); the learning rate goes through a pinned ring (run.rs:171-173)
        if (hipEventQuery(uploaded[set]) != hipSuccess)
            TRY_HIP(hipStreamWaitEvent(bt_stream(compute_stream), uploaded[set], 0));
        // (a copy only when this ring slot does not hold the value yet: every packet on the compute stream costs a
        // few microseconds of bubble, and most schedules change the rate once per superbatch or never)
        const float lr_now = (cfg->lr_fn ? cfg->lr_fn(cfg->sched_user, batch, sb) : bt_lr_at(&cfg->lr, batch, sb));
        if (!lr_slot_set[ring] || std::memcmp(&lr_slot[ring], &lr_now, sizeof(float)) != 0) {
            lr_host[ring] = lr_slot[ring] = lr_now, lr_slot_set[ring] = true;
            TRY(bt_memcpy_h2d_async(lr_dev + ring, lr_host + ring, sizeof(float), compute_stream));
        }
        const bt_batch bb = dev_batch(set);
        // data parallel: the loss sum sits right behind the flat gradient when there is room, so the exchange of a
        // step is one collective over n + 1 floats
        // single device: the step writes its loss sum straight into the pinned (device-visible) host ring -- one
        // 4-byte store over PCIe instead of a copy packet on the compute stream
        float *loss_at = loss_slot ? bt_net_grads(net) + bt_net_flat_size(net) : dp ? loss_dev + ring : loss_host + ring;
        if (!dp) {  // single device: one call, the update fused into the backward where the net allows it
            TRY(bt_net_train_step(net, compute_stream, &bb, loss_at, adj_dev, lr_dev + ring, consumed[set]));
        } else {
            TRY(bt_net_forward_backward(net, compute_stream, &bb, loss_at));
        }
        if (dp && comm) {  // exchange + optimiser in one call: RCCL from C, the table sharded when it is large
            TRY(bt_net_dp_step(net, comm, compute_stream, adj_dev, lr_dev + ring, cfg->dp_shard_min_floats));
        } else if (dp) {
            auto sync = [&](float *buf, size_t n) -> int {
                if (cfg->grad_sync(cfg->grad_sync_user, buf, n, compute_stream)) {
                    bt_set_error("bt_train_run: grad_sync failed");
                    return BT_ERR_STATE;
                }
                return BT_OK;
            };
            TRY(sync(bt_net_grads(net), bt_net_flat_size(net) + (loss_slot ? 1 : 0)));
            if (!loss_slot) TRY(sync(loss_at, 1));
        }
        if (dp && !comm) TRY(bt_net_optimiser_step_event(net, compute_stream, adj_dev, lr_dev + ring, consumed[set]));
        else if (dp) TRY_HIP(hipEventRecord(consumed[set], bt_stream(compute_stream)));
        if (dp) TRY(bt_memcpy_d2h_async(loss_host + ring, loss_at, sizeof(float), compute_stream));
        TRY_HIP(hipEventRecord(loss_ready[ring], bt_stream(compute_stream)));
        pending.push_back({sb, batch});

        if (prev_slot >= 0) TRY(release_slot(prev_slot));
        prev_slot = cur_slot;
        cur_slot = next_slot;
        TRY(deliver(false));
        if (++batch == bps) batch = 0, sb++;
    }
    TRY(deliver(true));
    if (dp && comm) TRY(bt_net_dp_gather_state(net, comm, compute_stream, cfg->dp_shard_min_floats));
    TRY(bt_stream_sync(compute_stream));
    if (cfg->log_rate > 0) {
        const unsigned total = (unsigned)secs(t_start, clock::now());
        std::printf("Total Training Time: %uh %um %us\n", total / 3600, (total / 60) % 60, total % 60);
    }
    cleanup();
    return BT_OK;
#undef TRY
#undef TRY_HIP
}
