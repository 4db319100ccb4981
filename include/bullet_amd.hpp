// bullet_amd.hpp -- C++17 host-side mirror of bullet_lib's value-trainer API over the C ABI of libbullet_amd.so.
//
// The reference's host is Rust (no toolchain in this image), so the layer a user of bullet touches --
// `ValueTrainerBuilder` -> `ValueTrainer::{run, eval, save_to_checkpoint, load_from_checkpoint}`, the schedules,
// `LocalSettings`, the data loaders, `SavedFormat` and the graph-builder closure (`builder.new_affine`,
// `l0.forward(stm).screlu()`, `.concat`, `.select(buckets)`, ...) -- is mirrored here with the same names, argument
// meaning and error behaviour, so that examples/*.cpp read like the reference's examples/*.rs
// (crates/bullet_lib/src/value/builder.rs:21-394, value.rs:134-272, trainer/schedule.rs:12-19,
// trainer/settings.rs:17-29, crates/compiler/src/model/builder.rs:45-522).
//
// The graph closure is recorded symbolically and matched against the net family the hand-written kernels cover
// (sparse feature transformer with optional pairwise product and factoriser -> up to BT_MAX_DENSE affine layers with
// optional output-bucket select, activation, the dual activation and skip neurons of examples/tests/test2.rs);
// a graph outside that family is rejected with an error, never approximated. Header-only; link with -lbullet_amd.
#ifndef BULLET_AMD_HPP
#define BULLET_AMD_HPP

#include <sys/stat.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "bullet_amd.h"

namespace bullet_lib {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline void check(int rc) {  // G::Error: From<String> (crates/gpu/src/runtime/bindings.rs:67)
    if (rc != BT_OK) throw Error(bt_last_error());
}

// ------------------------------------------------------------------------------------------ game::inputs / outputs
namespace inputs {
template <size_t N>
int get_num_buckets(const std::array<int, N> &layout) {  // game/inputs.rs: max + 1
    int m = 0;
    for (int v : layout) m = v > m ? v : m;
    return m + 1;
}
struct Chess768 {};
struct ChessBuckets {  // chess_buckets.rs:11-33
    std::array<int, 64> layout;
    explicit ChessBuckets(const std::array<int, 64> &l) : layout(l) {}
};
struct ChessBucketsMirrored {  // chess_buckets.rs:40-88
    std::array<int, 32> layout;
    explicit ChessBucketsMirrored(const std::array<int, 32> &l) : layout(l) {}
    static ChessBucketsMirrored make(const std::array<int, 32> &l) { return ChessBucketsMirrored(l); }
};
struct Ataxx147 {};
struct Ataxx98 {};
inline bt_input_desc describe(Chess768) {
    bt_input_desc d;
    check(bt_input_init(&d, BT_INPUT_CHESS768, 0, nullptr, 0));
    return d;
}
inline bt_input_desc describe(const ChessBuckets &b) {
    bt_input_desc d;
    check(bt_input_init(&d, BT_INPUT_CHESS_BUCKETS, 0, b.layout.data(), 64));
    return d;
}
inline bt_input_desc describe(const ChessBucketsMirrored &b) {
    bt_input_desc d;
    check(bt_input_init(&d, BT_INPUT_CHESS_BUCKETS_MIRRORED, 0, b.layout.data(), 32));
    return d;
}
inline bt_input_desc describe(Ataxx147) {
    bt_input_desc d;
    check(bt_input_init(&d, BT_INPUT_ATAXX147, 0, nullptr, 0));
    return d;
}
inline bt_input_desc describe(Ataxx98) {
    bt_input_desc d;
    check(bt_input_init(&d, BT_INPUT_ATAXX98, 0, nullptr, 0));
    return d;
}
}  // namespace inputs

namespace outputs {
template <int N>
struct MaterialCount {  // game/outputs.rs:24-31
    static constexpr int BUCKETS = N;
};
}  // namespace outputs

// ------------------------------------------------------------------------------------------ nn::optimiser
namespace optimiser {
struct AdamW {};
struct RAdam {};
struct Ranger {};
struct AdamWParams {  // crates/trainer/src/optimiser/adam.rs:21-33
    float decay = 0.01f, beta1 = 0.9f, beta2 = 0.999f, min_weight = -1.98f, max_weight = 1.98f;
};
}  // namespace optimiser

// ------------------------------------------------------------------------------------------ trainer::schedule
namespace lr {  // trainer/schedule/lr.rs -- any type with `float lr(int batch, int superbatch) const` is an LrScheduler
namespace detail {
inline float at(int kind, float a, float b, int n, int batch, int superbatch) {
    const bt_lr_sched s{kind, a, b, n, 0};
    return bt_lr_at(&s, batch, superbatch);
}
}  // namespace detail
struct ConstantLR {
    float value;
    float lr(int b, int sb) const { return detail::at(BT_LR_CONSTANT, value, 0.0f, 0, b, sb); }
};
struct DropLR {
    float start, gamma;
    int drop;
    float lr(int b, int sb) const { return detail::at(BT_LR_DROP, start, gamma, drop, b, sb); }
};
struct StepLR {
    float start, gamma;
    int step;
    float lr(int b, int sb) const { return detail::at(BT_LR_STEP, start, gamma, step, b, sb); }
};
struct LinearDecayLR {
    float initial_lr, final_lr;
    int final_superbatch;
    float lr(int b, int sb) const { return detail::at(BT_LR_LINEAR, initial_lr, final_lr, final_superbatch, b, sb); }
};
struct CosineDecayLR {
    float initial_lr, final_lr;
    int final_superbatch;
    float lr(int b, int sb) const { return detail::at(BT_LR_COSINE, initial_lr, final_lr, final_superbatch, b, sb); }
};
struct ExponentialDecayLR {
    float initial_lr, final_lr;
    int final_superbatch;
    float lr(int b, int sb) const { return detail::at(BT_LR_EXP, initial_lr, final_lr, final_superbatch, b, sb); }
};
template <class LR>
struct Warmup {  // lr.rs:166-187
    LR inner;
    int warmup_batches;
    float lr(int batch, int superbatch) const {
        const float base = inner.lr(batch, superbatch);
        return (superbatch == 1 && batch < warmup_batches) ? base / (float)(warmup_batches - batch) : base;
    }
};
template <class First, class Second>
struct Sequence {  // lr.rs:193-210
    First first;
    Second second;
    int first_scheduler_final_superbatch;
    float lr(int batch, int superbatch) const {
        const int mid = first_scheduler_final_superbatch;
        return superbatch <= mid ? first.lr(batch, superbatch) : second.lr(batch, superbatch - mid);
    }
};
}  // namespace lr

namespace wdl {  // trainer/schedule/wdl.rs -- `float blend(int batch, int superbatch, int max) const`
struct ConstantWDL {
    float value;
    float blend(int, int, int) const { return value; }
};
struct LinearWDL {
    float start, end;
    float blend(int batch, int superbatch, int max) const {
        const bt_wdl_sched s{BT_WDL_LINEAR, start, end, 0};
        return bt_wdl_at(&s, batch, superbatch, max);
    }
};
template <class W>
struct Warmup {  // wdl.rs:51-72
    W inner;
    int warmup_batches;
    float blend(int batch, int superbatch, int max) const {
        const float base = inner.blend(batch, superbatch, max);
        return (superbatch == 1 && batch < warmup_batches) ? base / (float)(warmup_batches - batch) : base;
    }
};
template <class First, class Second>
struct Sequence {  // wdl.rs:78-95
    First first;
    Second second;
    int first_scheduler_final_superbatch;
    float blend(int batch, int superbatch, int max) const {
        const int mid = first_scheduler_final_superbatch;
        return superbatch <= mid ? first.blend(batch, superbatch, mid) : second.blend(batch, superbatch - mid, max - mid);
    }
};
}  // namespace wdl

struct TrainingSteps {  // crates/trainer/src/run/schedule.rs:5-11
    int batch_size, batches_per_superbatch, start_superbatch, end_superbatch;
};
template <class LR, class WDL>
struct TrainingSchedule {  // trainer/schedule.rs:12-19
    std::string net_id;
    float eval_scale;
    TrainingSteps steps;
    WDL wdl_scheduler;
    LR lr_scheduler;
    int save_rate;
    bool should_save(int superbatch) const { return superbatch % save_rate == 0 || superbatch == steps.end_superbatch; }
};
struct LocalSettings {  // trainer/settings.rs:17-29
    int threads;
    const char *test_set;  // validation is "not currently implemented" in the reference (value.rs:153-158)
    std::string output_directory;
    int batch_queue_size;
};

// ------------------------------------------------------------------------------------------ value::loader
namespace loader {
struct DirectSequentialDataLoader {  // value/loader/direct.rs
    std::vector<std::string> paths;
    explicit DirectSequentialDataLoader(std::vector<std::string> p) : paths(std::move(p)) {}
    bt_reader *open() const { return nullptr; }  // read by bt_train_run itself, with resume-by-position
};
struct InMemoryTextLoader {  // value/loader/text.rs
    std::vector<std::string> paths;
    explicit InMemoryTextLoader(const std::string &p) : paths{"text:" + p} {}
    bt_reader *open() const { return nullptr; }
};
// value/loader/viribinpack.rs:17-27. ViriFilter is either the crate's built-in `Filter` (restated, parity unpinned: see
// csrc/viribinpack.cpp; Filter::standard() is `Filter::default()`, Filter::unrestricted() is `Filter::UNRESTRICTED`) or
// a custom function (position as a 32-byte ChessBoard, move, eval, wdl) -> keep?; an empty ViriFilter keeps everything.
struct Filter : bt_viri_builtin_filter {
    static Filter standard() {
        Filter f;
        bt_viri_filter_default(&f);
        return f;
    }
    static Filter unrestricted() {
        Filter f;
        bt_viri_filter_unrestricted(&f);
        return f;
    }
};
using ViriCustomFilter = std::function<bool(const uint8_t *chessboard, uint16_t mv, int16_t eval, float wdl)>;
struct ViriFilter {
    bool builtin = false;
    Filter filter{};
    ViriCustomFilter custom;
    ViriFilter() = default;
    ViriFilter(const Filter &f) : builtin(true), filter(f) {}  // `impl From<Filter> for ViriFilter`
    template <typename F, typename = decltype(std::declval<F>()((const uint8_t *)nullptr, uint16_t(), int16_t(), 0.0f))>
    ViriFilter(F f) : custom(std::move(f)) {}
};
struct ViriBinpackLoader {
    std::vector<std::string> paths;
    size_t buffer_size_mb;
    int threads;
    ViriFilter filter;
    ViriBinpackLoader(const std::string &path, size_t buffer_mb, int threads_, ViriFilter f = {})
        : paths{path}, buffer_size_mb(buffer_mb), threads(threads_), filter(std::move(f)) {}
    static ViriBinpackLoader new_concat_multiple(std::vector<std::string> p, size_t buffer_mb, int threads_,
                                                 ViriFilter f = {}) {
        ViriBinpackLoader l(p.at(0), buffer_mb, threads_, std::move(f));
        l.paths = std::move(p);
        return l;
    }
    bt_reader *open() const {  // `this` must outlive the reader (a custom filter is called through it)
        std::vector<const char *> c;
        for (const auto &p : paths) c.push_back(p.c_str());
        bt_reader *r = nullptr;
        if (filter.builtin) {
            check(bt_reader_open_viribinpack_builtin(c.data(), (int)c.size(), buffer_size_mb, threads, &filter.filter, 0, &r));
            return r;
        }
        bt_viri_filter cb = nullptr;
        if (filter.custom)
            cb = [](void *u, const void *rec, uint16_t mv, int16_t eval, float wdl, int) -> int {
                return static_cast<const ViriBinpackLoader *>(u)->filter.custom(static_cast<const uint8_t *>(rec), mv, eval, wdl);
            };
        check(bt_reader_open_viribinpack(c.data(), (int)c.size(), buffer_size_mb, threads, cb,
                                         const_cast<ViriBinpackLoader *>(this), 0, &r));
        return r;
    }
};
// value/loader/sfbinpack.rs. The filter sees a TrainingDataEntry's score / ply / result / move / destination piece /
// in-check flag (bt_sf_entry), as the closure of examples/simple.rs:96-102 does; an empty function keeps everything.
using SfFilter = std::function<bool(const bt_sf_entry &entry)>;
struct SfBinpackLoader {
    std::vector<std::string> paths;
    size_t buffer_size_mb;
    int threads;
    SfFilter filter;
    SfBinpackLoader(const std::string &path, size_t buffer_mb, int threads_, SfFilter f = {})
        : paths{path}, buffer_size_mb(buffer_mb), threads(threads_), filter(std::move(f)) {}
    static SfBinpackLoader new_concat_multiple(std::vector<std::string> p, size_t buffer_mb, int threads_,
                                               SfFilter f = {}) {
        SfBinpackLoader l(p.at(0), buffer_mb, threads_, std::move(f));
        l.paths = std::move(p);
        return l;
    }
    bt_reader *open() const {  // `this` must outlive the reader
        std::vector<const char *> c;
        for (const auto &p : paths) c.push_back(p.c_str());
        bt_reader *r = nullptr;
        bt_sf_filter cb = nullptr;
        if (filter)
            cb = [](void *u, const bt_sf_entry *e) -> int { return static_cast<const SfBinpackLoader *>(u)->filter(*e); };
        check(bt_reader_open_sfbinpack(c.data(), (int)c.size(), buffer_size_mb, threads, cb,
                                       const_cast<SfBinpackLoader *>(this), 0, &r));
        return r;
    }
};
// value/loader/montybinpack.rs. The filter is the reference's `Fn(&Position, Move, i16, f32) -> bool`: (position as a
// 32-byte ChessBoard, move in montyformat's encoding, score, game result) -> keep?; an empty function keeps everything.
using MontyFilter = std::function<bool(const uint8_t *chessboard, uint16_t mv, int16_t score, float result)>;
struct MontyBinpackLoader {
    std::vector<std::string> paths;
    size_t buffer_size_mb;
    int threads;
    MontyFilter filter;
    MontyBinpackLoader(const std::string &path, size_t buffer_mb, int threads_, MontyFilter f = {})
        : paths{path}, buffer_size_mb(buffer_mb), threads(threads_), filter(std::move(f)) {}
    static MontyBinpackLoader new_concat_multiple(std::vector<std::string> p, size_t buffer_mb, int threads_,
                                                  MontyFilter f = {}) {
        MontyBinpackLoader l(p.at(0), buffer_mb, threads_, std::move(f));
        l.paths = std::move(p);
        return l;
    }
    bt_reader *open() const {  // `this` must outlive the reader (the filter is called through it)
        std::vector<const char *> c;
        for (const auto &p : paths) c.push_back(p.c_str());
        bt_reader *r = nullptr;
        bt_monty_filter cb = nullptr;
        if (filter)
            cb = [](void *u, const void *rec, uint16_t mv, int16_t score, float result) -> int {
                return static_cast<const MontyBinpackLoader *>(u)->filter(static_cast<const uint8_t *>(rec), mv, score, result);
            };
        check(bt_reader_open_montybinpack(c.data(), (int)c.size(), buffer_size_mb, threads, cb,
                                          const_cast<MontyBinpackLoader *>(this), 0, &r));
        return r;
    }
};
}  // namespace loader

// ------------------------------------------------------------------------------------------ trainer::save
class SavedFormat {  // crates/trainer/src/model/weights.rs:183-275
  public:
    using Transform = std::function<void(const char *id, float *values, size_t count)>;
    static SavedFormat id(const std::string &name) {
        SavedFormat f;
        f.id_ = name;
        return f;
    }
    static SavedFormat custom(std::vector<uint8_t> bytes) {
        SavedFormat f;
        f.custom_ = std::move(bytes);
        f.is_custom_ = true;
        return f;
    }
    SavedFormat round() && { return round_ = true, std::move(*this); }
    SavedFormat transpose() && { return transpose_ = true, std::move(*this); }
    template <class T>
    SavedFormat quantise(int multiplier) && {
        static_assert(std::is_same<T, int8_t>::value || std::is_same<T, int16_t>::value || std::is_same<T, int32_t>::value,
                      "quantise::<i8 | i16 | i32>");
        quant_ = sizeof(T) == 1 ? BT_QUANT_I8 : sizeof(T) == 2 ? BT_QUANT_I16 : BT_QUANT_I32;
        multiplier_ = multiplier;
        return std::move(*this);
    }
    SavedFormat transform(Transform f) && { return transform_ = std::move(f), std::move(*this); }
    // the closure of examples/progression/3_input_buckets.rs:47-51: weights + l0f.repeat(NUM_INPUT_BUCKETS)
    SavedFormat merge_factoriser() && { return merge_ = true, std::move(*this); }

    bt_saved_format c() const {
        bt_saved_format f;
        std::memset(&f, 0, sizeof(f));
        f.id = is_custom_ ? nullptr : id_.c_str();
        f.custom = is_custom_ ? custom_.data() : nullptr;
        f.custom_len = is_custom_ ? custom_.size() : 0;
        f.quant = quant_, f.multiplier = multiplier_, f.round = round_, f.transpose = transpose_;
        f.merge_factoriser = merge_;
        if (transform_) {
            f.transform = [](void *user, const char *id, float *values, size_t count) -> int {
                (*static_cast<const Transform *>(user))(id, values, count);
                return 0;
            };
            f.user = const_cast<Transform *>(&transform_);
        }
        return f;
    }

  private:
    std::string id_;
    std::vector<uint8_t> custom_;
    bool is_custom_ = false, round_ = false, transpose_ = false, merge_ = false;
    int quant_ = BT_QUANT_FLOAT, multiplier_ = 1;
    Transform transform_;
};

// ------------------------------------------------------------------------------------------ graph builder (symbolic)
struct Shape {
    int rows, cols;
    static Shape make(int r, int c) { return {r, c}; }
};
enum class InitSettings { Zeroed };

namespace graph {
struct AffineData {
    std::string id;
    int in = 0, out = 0, order = 0, effective_inputs = 0;
};
struct WeightsData {  // a weight matrix, optionally `+ factoriser.repeat(n)` (3_input_buckets.rs:62-67)
    std::string id;
    int rows = 0, cols = 0;
    std::shared_ptr<WeightsData> factoriser;
    int repeat = 0;
};
enum class Kind { Stm, Ntm, Buckets, AffineFwd, Act, Mul, Add, Concat, Select, SliceRows, PairwiseMul };
struct NodeData {
    Kind kind;
    int rows = 0;
    std::shared_ptr<NodeData> a, b;
    int act = BT_ACT_IDENTITY, lo = 0, hi = 0, buckets = 1;
    std::shared_ptr<AffineData> layer;
    std::shared_ptr<WeightsData> weights;
};
using NodePtr = std::shared_ptr<NodeData>;
}  // namespace graph

class Weights {
  public:
    std::shared_ptr<graph::WeightsData> d;
    Weights repeat(int n) const {  // Broadcast(Cols), crates/compiler/src/model/operations/linear.rs:37-55
        auto w = std::make_shared<graph::WeightsData>();
        w->id = d->id, w->rows = d->rows, w->cols = d->cols * n, w->factoriser = d, w->repeat = n;
        return Weights{w};
    }
    friend Weights operator+(const Weights &base, const Weights &rep) {
        if (!rep.d->factoriser || rep.d->repeat <= 0 || base.d->factoriser)
            throw Error("weights may only be combined as `l0.weights + l0f.repeat(n)`");
        if (rep.d->rows != base.d->rows || rep.d->cols != base.d->cols) throw Error("Mismatched sizes!");
        auto w = std::make_shared<graph::WeightsData>(*base.d);
        w->factoriser = rep.d->factoriser, w->repeat = rep.d->repeat;
        return Weights{w};
    }
};

class Node {
  public:
    graph::NodePtr p;
    Node relu() const { return act(BT_ACT_RELU); }
    Node crelu() const { return act(BT_ACT_CRELU); }
    Node screlu() const { return act(BT_ACT_SCRELU); }
    Node sqrrelu() const { return act(BT_ACT_SQRRELU); }
    Node pairwise_mul() const {
        if (p->rows % 2) throw Error("pairwise_mul needs an even number of rows");
        return make(graph::Kind::PairwiseMul, p->rows / 2, p, nullptr);
    }
    Node concat(const Node &o) const { return make(graph::Kind::Concat, p->rows + o.p->rows, p, o.p); }
    Node select(const Node &buckets) const {  // model/operations/index.rs:110-114
        if (buckets.p->kind != graph::Kind::Buckets) throw Error("select takes the output-bucket input");
        if (p->rows % buckets.p->buckets) throw Error("Mismatched sizes!");
        return make(graph::Kind::Select, p->rows / buckets.p->buckets, p, buckets.p);
    }
    Node slice_rows(int start, int end) const {
        if (start < 0 || end > p->rows || start >= end) throw Error("Invalid slice indices!");
        Node n = make(graph::Kind::SliceRows, end - start, p, nullptr);
        n.p->lo = start, n.p->hi = end;
        return n;
    }
    friend Node operator*(const Node &x, const Node &y) {
        if (x.p->rows != y.p->rows) throw Error("Mismatched sizes!");
        return make(graph::Kind::Mul, x.p->rows, x.p, y.p);
    }
    friend Node operator+(const Node &x, const Node &y) {
        if (x.p->rows != y.p->rows) throw Error("Mismatched sizes!");
        return make(graph::Kind::Add, x.p->rows, x.p, y.p);
    }
    static Node make(graph::Kind k, int rows, graph::NodePtr a, graph::NodePtr b) {
        auto n = std::make_shared<graph::NodeData>();
        n->kind = k, n->rows = rows, n->a = std::move(a), n->b = std::move(b);
        return Node{n};
    }

  private:
    Node act(int id) const {
        Node n = make(graph::Kind::Act, p->rows, p, nullptr);
        n.p->act = id;
        return n;
    }
};

class Affine {  // crates/compiler/src/model/builder.rs:103-168
  public:
    Weights weights, bias;
    std::shared_ptr<graph::AffineData> d;
    int lo = 0, hi = 0;
    Node forward(const Node &input) const {
        const bool sparse = input.p->kind == graph::Kind::Stm || input.p->kind == graph::Kind::Ntm;
        if (!sparse && input.p->rows != d->in) throw Error("Mismatched sizes!");
        Node n = Node::make(graph::Kind::AffineFwd, hi - lo, input.p, nullptr);
        n.p->layer = d, n.p->weights = weights.d, n.p->lo = lo, n.p->hi = hi;
        return n;
    }
    Affine slice(int start, int end) const {  // rows [start, end) of the layer (4_multi_layer.rs:79)
        if (start < lo || end > hi || start >= end) throw Error("Invalid slice indices!");
        Affine a = *this;
        a.lo = start, a.hi = end;
        return a;
    }
    void init_with_effective_input_size(int n) { d->effective_inputs = n; }  // builder.rs:149-152
};

class ModelBuilder {
  public:
    Affine new_affine(const std::string &id, int in, int out) {  // builder.rs:108-119
        auto a = std::make_shared<graph::AffineData>();
        a->id = id, a->in = in, a->out = out, a->order = next_++;
        auto w = std::make_shared<graph::WeightsData>();
        w->id = id + "w", w->rows = out, w->cols = in;
        auto b = std::make_shared<graph::WeightsData>();
        b->id = id + "b", b->rows = out, b->cols = 1;
        Affine l;
        l.weights = Weights{w}, l.bias = Weights{b}, l.d = a, l.lo = 0, l.hi = out;
        return l;
    }
    Weights new_weights(const std::string &id, Shape shape, InitSettings) {
        auto w = std::make_shared<graph::WeightsData>();
        w->id = id, w->rows = shape.rows, w->cols = shape.cols;
        return Weights{w};
    }

  private:
    int next_ = 0;
};

// loss_fn closure: |output, target| output.sigmoid().squared_error(target) etc. (model/builder.rs:345-363)
struct LossTarget {};
struct LossValue {
    int kind;
    float power;
};
struct LossSigmoid {
    LossValue squared_error(LossTarget) const { return {BT_LOSS_SIGMOID_MSE, 2.0f}; }
    LossValue power_error(LossTarget, float power) const { return {BT_LOSS_SIGMOID_POWER, power}; }
};
struct LossOutput {
    LossSigmoid sigmoid() const { return {}; }
    LossValue softmax_crossentropy_loss(LossTarget) const { return {BT_LOSS_SOFTMAX_CE, 0.0f}; }
};

namespace graph {
// Matches the recorded graph against the supported family and fills a bt_net_desc (the op schedule that
// ModelIR::lower + the canonicaliser would produce for it is hard-wired in csrc/net.hip).
struct Matched {
    bt_net_desc desc;
    int ft_effective_inputs = 0;
};

inline NodePtr strip_act(NodePtr n, int *act) {
    *act = BT_ACT_IDENTITY;
    if (n->kind == Kind::Act) {
        *act = n->act;
        return n->a;
    }
    return n;
}

// one perspective of the feature transformer; returns the sparse input kind
inline void match_ft_side(const NodePtr &n, Kind want, bt_net_desc *d, std::shared_ptr<AffineData> *l0,
                          std::shared_ptr<WeightsData> *w0, bool first) {
    auto bad = [] { return Error("unsupported feature-transformer sub-graph (expected l0.forward(input).<act>() "
                                 "[.pairwise_mul()] or the two-slice product form)"); };
    int act = BT_ACT_IDENTITY, pairwise = 0;
    NodePtr fwd;
    if (n->kind == Kind::Mul) {  // ft(input, 0, h/2) * ft(input, h/2, h)
        int a0, a1;
        NodePtr x = strip_act(n->a, &a0), y = strip_act(n->b, &a1);
        if (x->kind != Kind::AffineFwd || y->kind != Kind::AffineFwd || a0 != a1 || x->layer != y->layer ||
            x->a != y->a || x->lo != 0 || x->hi != y->lo || y->hi != x->layer->out || x->hi * 2 != y->hi)
            throw bad();
        act = a0, pairwise = 1, fwd = x;
    } else if (n->kind == Kind::PairwiseMul) {
        fwd = strip_act(n->a, &act);
        pairwise = 1;
        if (fwd->kind != Kind::AffineFwd || fwd->lo != 0 || fwd->hi != fwd->layer->out) throw bad();
    } else {
        fwd = strip_act(n, &act);
        if (fwd->kind != Kind::AffineFwd || fwd->lo != 0 || fwd->hi != fwd->layer->out) throw bad();
    }
    if (fwd->a->kind != want) throw bad();
    if (first) {
        *l0 = fwd->layer, *w0 = fwd->weights;
        d->ft_act = act, d->pairwise = pairwise, d->ft_out = fwd->layer->out, d->num_inputs = fwd->layer->in;
    } else if (*l0 != fwd->layer || d->ft_act != act || d->pairwise != pairwise ||
               (*w0)->factoriser != fwd->weights->factoriser) {
        throw Error("both perspectives must go through the same feature transformer");
    }
}

inline Matched match(const NodePtr &output, int perspectives, int num_buckets, int nnz, LossValue loss) {
    Matched m;
    bt_net_desc &d = m.desc;
    std::memset(&d, 0, sizeof(d));
    d.perspectives = perspectives, d.num_buckets = num_buckets, d.nnz = nnz;
    d.loss_kind = loss.kind, d.loss_power = loss.power;
    struct Layer {
        std::shared_ptr<AffineData> l;
        int act, bucketed, dual, skip;
    };
    std::vector<Layer> rev;  // last layer first
    NodePtr cur = output, skip_src;
    int skip_lo = 0, skip_hi = 0;
    if (cur->kind == Kind::Add) {  // `out + skip_neuron` (test2.rs:57-65)
        NodePtr a = cur->a, b = cur->b;
        if (a->kind == Kind::SliceRows && b->kind != Kind::SliceRows) std::swap(a, b);
        if (b->kind != Kind::SliceRows) throw Error("unsupported graph: `+` is only matched as `out + skip_neuron`");
        skip_src = b->a, skip_lo = b->lo, skip_hi = b->hi, cur = a;
    }
    bool skip_used = false;
    for (;;) {
        // what a layer's output goes through before the next layer sees it: [slice_rows] [concat(x, x*x)] [act]
        int act, dual = 0, skip = 0;
        NodePtr n = strip_act(cur, &act);
        if (n->kind == Kind::Concat && n->b->kind == Kind::Mul && n->b->a == n->a && n->b->b == n->a) {
            dual = 1, n = n->a;
        }
        if (n->kind == Kind::SliceRows && skip_src && n->a == skip_src) {
            if (n->lo != 0 || n->hi != skip_lo || skip_hi != skip_src->rows)
                throw Error("skip neurons must be the last rows of a layer output");
            skip = skip_hi - skip_lo, skip_used = true, n = n->a;
        }
        int bucketed = 0;
        if (n->kind == Kind::Select) bucketed = 1, n = n->a;
        if (n->kind != Kind::AffineFwd || n->lo != 0 || n->hi != n->layer->out) {
            // not a dense layer: must be the feature-transformer output, reached with nothing stripped
            if (n != cur) throw Error("unsupported graph between the layers");
            break;
        }
        if (n->a->kind == Kind::Stm || n->a->kind == Kind::Ntm) {  // this affine IS the feature transformer
            if (bucketed || dual || skip) throw Error("unsupported graph after the feature transformer");
            break;
        }
        rev.push_back({n->layer, act, bucketed, dual, skip});
        cur = n->a;
    }
    if (skip_src && !skip_used) throw Error("skip neuron is not sliced from a layer of this net");
    if (rev.empty() || rev.size() > BT_MAX_DENSE) throw Error("the net needs 1..4 dense layers after the feature transformer");
    // feature transformer
    std::shared_ptr<AffineData> l0;
    std::shared_ptr<WeightsData> w0;
    if (perspectives == 2) {
        if (cur->kind != Kind::Concat) throw Error("dual_perspective: expected stm_hidden.concat(ntm_hidden)");
        match_ft_side(cur->a, Kind::Stm, &d, &l0, &w0, true);
        match_ft_side(cur->b, Kind::Ntm, &d, &l0, &w0, false);
    } else {
        match_ft_side(cur, Kind::Stm, &d, &l0, &w0, true);
    }
    if (w0->factoriser) {
        d.ft_factoriser = w0->repeat;
        if (w0->factoriser->id != "l0f") throw Error("the factoriser weights must be called l0f");
    }
    m.ft_effective_inputs = l0->effective_inputs;
    // the C ABI names and orders its tensors l0, l1, ... in creation order (weights.rs:57)
    if (l0->id != "l0" || l0->order != 0) throw Error("the feature transformer must be the first affine and be called l0");
    d.n_dense = (int)rev.size();
    int in = (d.pairwise ? d.ft_out / 2 : d.ft_out) * perspectives;
    for (int k = 0; k < d.n_dense; k++) {
        const Layer &L = rev[rev.size() - 1 - k];
        if (L.l->id != "l" + std::to_string(k + 1) || L.l->order != k + 1)
            throw Error("dense layers must be created in order and be called l1, l2, ...");
        if (L.l->in != in) throw Error("Mismatched sizes!");
        const int nb = L.bucketed ? num_buckets : 1;
        if (L.l->out % nb) throw Error("Mismatched sizes!");
        d.dense_out[k] = L.l->out / nb, d.dense_act[k] = L.act, d.dense_bucketed[k] = L.bucketed;
        d.dense_dual[k] = L.dual, d.dense_skip[k] = L.skip;
        in = (d.dense_out[k] - L.skip) * (L.dual ? 2 : 1);
    }
    return m;
}
}  // namespace graph

// ------------------------------------------------------------------------------------------ ValueTrainer
class ValueTrainer;

class OptimiserHandle {  // `trainer.optimiser.*` (crates/trainer/src/optimiser.rs:196-236)
  public:
    void set_params_for_weight(const std::string &id, optimiser::AdamWParams p) {
        params_[id] = p;
        if (apply_) apply_();
    }
    void set_params(optimiser::AdamWParams p) {  // every weight (optimiser.rs:208-212)
        all_ = p, has_all_ = true, params_.clear();
        if (apply_) apply_();
    }
    inline void reset_state();                                   // optimiser.rs:196-202
    inline void write_to_checkpoint(const std::string &dir);     // optimiser.rs:218-224: weights / momentum / velocity
    inline void load_weights_from_file(const std::string &path); // optimiser.rs:226-230: a weights.bin, state untouched
    inline void load_from_checkpoint(const std::string &dir);    // optimiser.rs:232-236

  private:
    friend class ValueTrainer;
    std::map<std::string, optimiser::AdamWParams> params_;
    optimiser::AdamWParams all_;
    bool has_all_ = false;
    std::function<void()> apply_;
    ValueTrainer *owner_ = nullptr;
};

// wdl_adjust_function / datapoint_weight_function (value/builder.rs:102-111): closures over the position record
using BlendFn = std::function<float(const uint8_t *position, float blend)>;
using WeightFn = std::function<float(const uint8_t *position)>;

class ValueTrainer {
  public:
    OptimiserHandle optimiser;
    BlendFn wdl_adjust;     // set by the builder
    WeightFn datapoint_weight;

    ValueTrainer(graph::Matched m, bt_input_desc input, int out_buckets, int opt_kind, std::vector<SavedFormat> fmt,
                 uint64_t seed, int device, bool wdl_output, bool win_rate_model)
        : m_(m), input_(input), out_buckets_(out_buckets), opt_kind_(opt_kind), fmt_(std::move(fmt)), device_(device),
          wdl_output_(wdl_output), win_rate_model_(win_rate_model) {
        check(bt_device_set(device_));
        const size_t n = bt_net_param_count(&m_.desc);
        weights_.assign(n, 0.0f);  // ModelWeights::new(&definition, seed) (value/builder.rs:145)
        check(bt_net_init_weights_host(&m_.desc, seed, m_.ft_effective_inputs, weights_.data()));
        optimiser.apply_ = [this] { apply_params(); };
        optimiser.owner_ = this;
    }
    ValueTrainer(const ValueTrainer &) = delete;
    ValueTrainer &operator=(const ValueTrainer &) = delete;
    ~ValueTrainer() {
        if (net_) bt_net_destroy(net_);
    }

    const bt_net_desc &desc() const { return m_.desc; }

    // Data-parallel training (no counterpart in the reference): one process per GPU, each created with
    // use_device(local rank); `sync` must sum `n` floats at `buf` over all ranks, in place, in `stream` order (e.g.
    // ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, comm, (hipStream_t)stream)). The schedule's batch_size stays
    // the global batch; every rank must be given the same files (and, for binpack loaders, the same shuffle seed).
    using GradSync = int (*)(void *user, float *buf_dev, size_t n, void *stream);
    void data_parallel(int rank, int world, GradSync sync, void *user) {
        dp_rank_ = rank, dp_world_ = world, dp_sync_ = sync, dp_user_ = user;
    }
    // The same through the library's own RCCL communicator (bt_comm_unique_id on rank 0, the 128-byte id handed to the
    // other ranks, bt_comm_init_rank on each with its GPU current): the loop then exchanges with bt_net_dp_step -- a
    // large feature-transformer table reduce-scattered, stepped on 1/world of it per rank and all-gathered, the rest
    // all-reduced -- and nothing but C++ runs on the step path.
    void data_parallel(bt_comm *comm) {
        dp_comm_ = comm, dp_rank_ = bt_comm_rank(comm), dp_world_ = bt_comm_world(comm), dp_sync_ = nullptr, dp_user_ = nullptr;
    }

    // value.rs:141-211: superbatch loop, checkpoints every save_rate superbatches + at the end, log.txt
    template <class LR, class WDL, class Loader>
    void run(const TrainingSchedule<LR, WDL> &schedule, const LocalSettings &settings, const Loader &data) {
        if (settings.test_set) std::printf("Warning: Validation data not currently implemented!\n");
        ensure_net(schedule.steps.batch_size);
        (void)mkdir(settings.output_directory.c_str(), 0777);
        bt_train_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.batch_size = schedule.steps.batch_size, cfg.batches_per_superbatch = schedule.steps.batches_per_superbatch;
        set_schedulers(&cfg, schedule);
        cfg.out_buckets = out_buckets_, cfg.use_win_rate_model = win_rate_model_, cfg.threads = settings.threads;
        cfg.log_rate = 128, cfg.wdl_output = wdl_output_, cfg.device_mapper = 0;
        cfg.dp_rank = dp_rank_, cfg.dp_world = dp_world_, cfg.grad_sync = dp_sync_, cfg.grad_sync_user = dp_user_;
        cfg.comm = dp_comm_;
        cfg.fn_user = this;
        if (wdl_adjust)
            cfg.blend_fn = [](void *u, const void *rec, float blend) -> float {
                return static_cast<ValueTrainer *>(u)->wdl_adjust(static_cast<const uint8_t *>(rec), blend);
            };
        if (datapoint_weight)
            cfg.weight_fn = [](void *u, const void *rec) -> float {
                return static_cast<ValueTrainer *>(u)->datapoint_weight(static_cast<const uint8_t *>(rec));
            };
        std::vector<const char *> paths;
        for (const auto &p : data.paths) paths.push_back(p.c_str());
        struct Record {
            std::vector<std::array<float, 3>> rows;  // (superbatch, batch, loss)
            float loss_sum = 0.0f, ticks = 0.0f;
            int bps = 0;
        } rec;
        rec.bps = cfg.batches_per_superbatch;
        auto on_batch = [](void *u, int sb, int batch, float error) {  // value.rs:182-196
            auto *r = static_cast<Record *>(u);
            r->loss_sum += error, r->ticks += 1.0f;
            if (batch % 32 == 0 || (r->bps < 32 && batch == r->bps)) {
                r->rows.push_back({(float)sb, (float)batch, r->loss_sum / std::fmin(r->ticks, (float)r->bps)});
                r->loss_sum = 0.0f, r->ticks = 0.0f;
            }
        };
        // The loop runs from checkpoint to checkpoint: bt_train_run drains its pipeline when it returns, so the state
        // saved is exactly the one after the last batch of that superbatch (the reference saves inside its
        // superbatch callback, value.rs:197-208); the reader resumes by position like a restarted run (reader.rs:40).
        struct ReaderGuard {  // loaders that decode (binpacks) hand over one reader for the whole run
            bt_reader *r;
            ~ReaderGuard() { bt_reader_close(r); }
        } guard{data.open()};
        cfg.reader = guard.r;
        int sb = schedule.steps.start_superbatch;
        while (sb <= schedule.steps.end_superbatch) {
            int stop = sb;
            while (!schedule.should_save(stop)) stop++;
            cfg.start_superbatch = sb, cfg.end_superbatch = stop;
            // schedules see the real final superbatch (LinearWDL, wdl.rs:40-45): keep it for bt_wdl_at
            last_loss_ = run_span(cfg, schedule.steps.end_superbatch, paths, on_batch, &rec);
            const std::string name = schedule.net_id + "-" + std::to_string(stop);
            const std::string path = settings.output_directory + "/" + name;
            save_to_checkpoint(path);
            if (FILE *f = std::fopen((path + "/log.txt").c_str(), "w")) {  // save::write_losses
                for (const auto &r : rec.rows) std::fprintf(f, "%d,%d,%g\n", (int)r[0], (int)r[1], r[2]);
                std::fclose(f);
            }
            std::printf("Saved [%s]\n", name.c_str());
            sb = stop + 1;
        }
    }

    float last_superbatch_loss() const { return last_loss_; }

    // value.rs:213-255
    std::vector<float> eval_raw_output(const std::string &fen) {
        ensure_net(1);
        uint8_t rec[32];
        check(bt_chessboard_from_str((fen + " | 0 | 0.0").c_str(), rec));
        const int nnz = bt_input_max_active(&input_), tw = wdl_output_ ? 3 : 1;
        std::vector<int32_t> stm(nnz), ntm(nnz);
        int32_t bucket = 0;
        float target[3] = {0, 0, 0}, weight = 1.0f;
        bt_map_opts o;
        std::memset(&o, 0, sizeof(o));
        o.out_buckets = out_buckets_, o.blend = 1.0f, o.eval_scale = 1.0f;  // make_mapper(1.0, ConstantWDL { value: 1.0 })
        o.use_win_rate_model = win_rate_model_, o.wdl_output = wdl_output_, o.threads = 1;
        check(bt_map_batch(&input_, &o, rec, 1, stm.data(), ntm.data(), &bucket, target, &weight));
        const int width = m_.desc.dense_out[m_.desc.n_dense - 1];
        char *dev = nullptr;
        const size_t isz = sizeof(int32_t) * nnz, total = 2 * isz + 64 + sizeof(float) * (tw + 1 + width);
        check(bt_malloc((void **)&dev, total));
        int32_t *d_stm = (int32_t *)dev, *d_ntm = (int32_t *)(dev + isz), *d_b = (int32_t *)(dev + 2 * isz);
        float *d_t = (float *)(dev + 2 * isz + 64), *d_w = d_t + tw, *d_out = d_w + 1;
        int rc = bt_memcpy_h2d_async(d_stm, stm.data(), isz, nullptr);
        if (!rc) rc = bt_memcpy_h2d_async(d_ntm, ntm.data(), isz, nullptr);
        if (!rc) rc = bt_memcpy_h2d_async(d_b, &bucket, sizeof(bucket), nullptr);
        if (!rc) rc = bt_memcpy_h2d_async(d_t, target, sizeof(float) * tw, nullptr);
        if (!rc) rc = bt_memcpy_h2d_async(d_w, &weight, sizeof(float), nullptr);
        bt_batch b{1, d_stm, m_.desc.perspectives == 2 ? d_ntm : nullptr, m_.desc.num_buckets > 1 ? d_b : nullptr, d_t,
                   d_w};
        std::vector<float> out(width);
        if (!rc) rc = bt_net_forward(net_, nullptr, &b, d_out, nullptr);
        if (!rc) rc = bt_memcpy_d2h_async(out.data(), d_out, sizeof(float) * width, nullptr);
        if (!rc) rc = bt_stream_sync(nullptr);
        bt_free(dev);
        check(rc);
        return out;
    }
    float eval(const std::string &fen) {
        const std::vector<float> v = eval_raw_output(fen);
        if (v.size() == 3) {  // [loss, draw, win]
            const float mx = std::fmax(v[2], std::fmax(v[1], v[0]));
            const float win = std::exp(v[2] - mx), draw = std::exp(v[1] - mx), loss = std::exp(v[0] - mx);
            return (win + draw / 2.0f) / (win + draw + loss);
        }
        if (v.size() == 1) return v[0];
        throw Error("Invalid output size!");
    }

    void save_to_checkpoint(const std::string &path) {  // value/save.rs:24-43
        ensure_net(1);
        std::vector<bt_saved_format> f;
        for (const auto &s : fmt_) f.push_back(s.c());
        check(bt_net_save_checkpoint(net_, nullptr, f.data(), (int)f.size(), path.c_str()));
    }
    void load_from_checkpoint(const std::string &path) {  // value/save.rs:93-101
        ensure_net(1);
        check(bt_net_load_checkpoint(net_, nullptr, path.c_str()));
    }
    void save_quantised(const std::string &path) {  // value/save.rs:107-109
        download();
        std::vector<bt_saved_format> f;
        for (const auto &s : fmt_) f.push_back(s.c());
        check(bt_save_quantised(&m_.desc, weights_.data(), f.data(), (int)f.size(), path.c_str()));
    }
    void save_unquantised(const std::string &path) {  // value/save.rs:111-113
        download();
        std::vector<bt_saved_format> f;
        for (const auto &s : fmt_) f.push_back(s.c());
        check(bt_save_raw(&m_.desc, weights_.data(), f.data(), (int)f.size(), path.c_str()));
    }
    template <class LR, class WDL, class Loader>
    double measure_max_cpu_throughput(const TrainingSchedule<LR, WDL> &schedule, const LocalSettings &settings,
                                      const Loader &data) {  // value.rs:257-272
        bt_train_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.batch_size = schedule.steps.batch_size, cfg.batches_per_superbatch = schedule.steps.batches_per_superbatch;
        cfg.start_superbatch = schedule.steps.start_superbatch, cfg.end_superbatch = schedule.steps.end_superbatch;
        set_schedulers(&cfg, schedule);
        cfg.out_buckets = out_buckets_, cfg.use_win_rate_model = win_rate_model_, cfg.threads = settings.threads;
        cfg.log_rate = 1, cfg.wdl_output = wdl_output_;
        std::vector<const char *> paths;
        for (const auto &p : data.paths) paths.push_back(p.c_str());
        double pps = 0.0;
        check(bt_measure_loader_throughput(&input_, &cfg, paths.data(), (int)paths.size(), &pps));
        return pps;
    }

  private:
    friend class OptimiserHandle;
    void download() {  // device weights -> weights_
        ensure_net(1);
        check(bt_memcpy_d2h_async(weights_.data(), bt_net_params(net_), weights_.size() * sizeof(float), nullptr));
        check(bt_stream_sync(nullptr));
    }
    void upload() {
        ensure_net(1);
        check(bt_memcpy_h2d_async(bt_net_params(net_), weights_.data(), weights_.size() * sizeof(float), nullptr));
        check(bt_stream_sync(nullptr));
    }
    // The schedulers are called back from the training loop (the reference's are traits): `schedule` must outlive it.
    template <class LR, class WDL>
    static void set_schedulers(bt_train_config *cfg, const TrainingSchedule<LR, WDL> &schedule) {
        using S = TrainingSchedule<LR, WDL>;
        cfg->eval_scale = schedule.eval_scale;
        cfg->sched_user = const_cast<S *>(&schedule);
        cfg->lr_fn = [](void *u, int batch, int sb) -> float { return static_cast<const S *>(u)->lr_scheduler.lr(batch, sb); };
        cfg->wdl_fn = [](void *u, int batch, int sb, int max) -> float {
            return static_cast<const S *>(u)->wdl_scheduler.blend(batch, sb, max);
        };
    }
    // (Re)creates the device net for at least `batch` positions, carrying weights and optimiser state across.
    void ensure_net(int batch) {
        check(bt_device_set(device_));
        if (net_ && bt_net_max_batch(net_) >= batch) return;
        const size_t n = weights_.size(), bytes = n * sizeof(float);
        std::vector<float> mom, vel;
        if (net_) {
            mom.resize(n), vel.resize(n);
            check(bt_memcpy_d2h_async(weights_.data(), bt_net_params(net_), bytes, nullptr));
            check(bt_memcpy_d2h_async(mom.data(), bt_net_momentum(net_), bytes, nullptr));
            check(bt_memcpy_d2h_async(vel.data(), bt_net_velocity(net_), bytes, nullptr));
            check(bt_stream_sync(nullptr));
            check(bt_net_destroy(net_));
            net_ = nullptr;
        }
        check(bt_net_create(&m_.desc, batch, nullptr, nullptr, &net_));
        if (opt_kind_ != BT_OPT_ADAMW) check(bt_net_set_optimiser(net_, opt_kind_, 5.0f, 0.5f, 6));
        check(bt_memcpy_h2d_async(bt_net_params(net_), weights_.data(), bytes, nullptr));
        if (!mom.empty()) {
            check(bt_memcpy_h2d_async(bt_net_momentum(net_), mom.data(), bytes, nullptr));
            check(bt_memcpy_h2d_async(bt_net_velocity(net_), vel.data(), bytes, nullptr));
        }
        check(bt_stream_sync(nullptr));
        apply_params();
    }
    void apply_params() {
        if (!net_) return;
        if (optimiser.has_all_)
            for (int i = 0; i < bt_net_num_tensors(&m_.desc); i++) {
                const auto &p = optimiser.all_;
                check(bt_net_set_adamw(net_, i, p.decay, p.beta1, p.beta2, p.min_weight, p.max_weight));
            }
        for (const auto &kv : optimiser.params_) {
            int t = -1;
            for (int i = 0; i < bt_net_num_tensors(&m_.desc); i++)
                if (kv.first == bt_net_tensor_name(&m_.desc, i)) t = i;
            if (t < 0) throw Error("No weight with ID found!");
            const auto &p = kv.second;
            check(bt_net_set_adamw(net_, t, p.decay, p.beta1, p.beta2, p.min_weight, p.max_weight));
        }
    }
    float run_span(bt_train_config cfg, int final_superbatch, const std::vector<const char *> &paths,
                   bt_batch_callback on_batch, void *rec) {
        // bt_wdl_at / schedules take the schedule's final superbatch from cfg.end_superbatch, so a span that stops
        // early for a checkpoint still has to present the whole schedule: run it as [start, stop] of [.., final]
        struct Ctx {
            void *rec;
            bt_batch_callback on_batch;
            float last = 0.0f;
        } ctx{rec, on_batch};
        auto b_cb = [](void *u, int sb, int batch, float e) { static_cast<Ctx *>(u)->on_batch(static_cast<Ctx *>(u)->rec, sb, batch, e); };
        auto sb_cb = [](void *u, int, float loss, float) { static_cast<Ctx *>(u)->last = loss; };
        const int stop = cfg.end_superbatch;
        cfg.end_superbatch = final_superbatch;
        cfg.stop_superbatch = stop;
        check(bt_train_run(net_, &input_, &cfg, paths.data(), (int)paths.size(), b_cb, sb_cb, &ctx));
        return ctx.last;
    }

    graph::Matched m_;
    bt_input_desc input_;
    int out_buckets_, opt_kind_;
    std::vector<SavedFormat> fmt_;
    int device_;
    bool wdl_output_, win_rate_model_;
    std::vector<float> weights_;
    bt_net *net_ = nullptr;
    float last_loss_ = 0.0f;
    int dp_rank_ = 0, dp_world_ = 1;
    GradSync dp_sync_ = nullptr;
    void *dp_user_ = nullptr;
    bt_comm *dp_comm_ = nullptr;
};

inline void OptimiserHandle::reset_state() {
    owner_->ensure_net(1);
    check(bt_net_set_optimiser(owner_->net_, owner_->opt_kind_, 5.0f, 0.5f, 6));  // zeroes the state and the step count
}
inline void OptimiserHandle::write_to_checkpoint(const std::string &dir) {
    ValueTrainer &t = *owner_;
    t.ensure_net(1);
    (void)mkdir(dir.c_str(), 0777);
    std::vector<float> host(t.weights_.size());
    const struct {
        const float *dev;
        const char *file;
    } parts[3] = {{bt_net_params(t.net_), "/weights.bin"}, {bt_net_momentum(t.net_), "/momentum.bin"},
                  {bt_net_velocity(t.net_), "/velocity.bin"}};
    for (const auto &p : parts) {
        check(bt_memcpy_d2h_async(host.data(), p.dev, host.size() * sizeof(float), nullptr));
        check(bt_stream_sync(nullptr));
        check(bt_checkpoint_write(&t.m_.desc, host.data(), (dir + p.file).c_str()));
    }
}
inline void OptimiserHandle::load_weights_from_file(const std::string &path) {
    owner_->download();  // ids missing from the file keep their values
    check(bt_checkpoint_read(&owner_->m_.desc, owner_->weights_.data(), path.c_str()));
    owner_->upload();
}
inline void OptimiserHandle::load_from_checkpoint(const std::string &dir) {  // `dir` = .../optimiser_state
    ValueTrainer &t = *owner_;
    t.ensure_net(1);
    std::vector<float> host(t.weights_.size());
    const struct {
        float *dev;
        const char *file;
    } parts[3] = {{bt_net_params(t.net_), "/weights.bin"}, {bt_net_momentum(t.net_), "/momentum.bin"},
                  {bt_net_velocity(t.net_), "/velocity.bin"}};
    for (const auto &p : parts) {
        std::fill(host.begin(), host.end(), 0.0f);
        check(bt_checkpoint_read(&t.m_.desc, host.data(), (dir + p.file).c_str()));
        check(bt_memcpy_h2d_async(p.dev, host.data(), host.size() * sizeof(float), nullptr));
        check(bt_stream_sync(nullptr));
    }
}

// ------------------------------------------------------------------------------------------ ValueTrainerBuilder
class ValueTrainerBuilder {  // crates/bullet_lib/src/value/builder.rs:21-394
  public:
    ValueTrainerBuilder() = default;
    static ValueTrainerBuilder default_() { return ValueTrainerBuilder(); }

    ValueTrainerBuilder &dual_perspective() { return perspectives_ = 2, *this; }
    ValueTrainerBuilder &single_perspective() { return perspectives_ = 1, *this; }
    ValueTrainerBuilder &optimiser(optimiser::AdamW) { return opt_ = BT_OPT_ADAMW, *this; }
    ValueTrainerBuilder &optimiser(optimiser::RAdam) { return opt_ = BT_OPT_RADAM, *this; }
    ValueTrainerBuilder &optimiser(optimiser::Ranger) { return opt_ = BT_OPT_RANGER, *this; }
    template <class I>
    ValueTrainerBuilder &inputs(const I &i) {
        input_ = inputs::describe(i);
        has_input_ = true;
        // king-bucketed sets keep every position inside one 768-column block: LDS-resident kernels per bucket
        segment_ = (std::is_same<I, inputs::ChessBuckets>::value || std::is_same<I, inputs::ChessBucketsMirrored>::value)
                       ? 768 : 0;
        return *this;
    }
    template <int N>
    ValueTrainerBuilder &output_buckets(outputs::MaterialCount<N>) { return buckets_ = N, *this; }
    ValueTrainerBuilder &save_format(std::vector<SavedFormat> f) { return fmt_ = std::move(f), *this; }
    template <class F>
    ValueTrainerBuilder &loss_fn(F f) { return loss_ = f(LossOutput{}, LossTarget{}), has_loss_ = true, *this; }
    ValueTrainerBuilder &wdl_output() { return wdl_output_ = true, *this; }
    ValueTrainerBuilder &use_win_rate_model() { return win_rate_ = true, *this; }
    ValueTrainerBuilder &wdl_adjust_function(BlendFn f) { return blend_fn_ = std::move(f), *this; }
    ValueTrainerBuilder &datapoint_weight_function(WeightFn f) { return weight_fn_ = std::move(f), *this; }
    ValueTrainerBuilder &seed(uint64_t s) { return seed_ = s, *this; }
    ValueTrainerBuilder &use_device(int ordinal) { return device_ = ordinal, *this; }

    // build(|builder, stm_inputs[, ntm_inputs][, output_buckets]| -> output)
    template <class F>
    std::unique_ptr<ValueTrainer> build(F f) {
        graph::Matched m = lower(f);
        auto t = std::make_unique<ValueTrainer>(m, input_, buckets_, opt_, fmt_, seed_, device_, wdl_output_, win_rate_);
        t->wdl_adjust = blend_fn_, t->datapoint_weight = weight_fn_;
        return t;
    }
    // The host half of build(): runs the closure and matches the graph; touches no device (tests use it).
    template <class F>
    graph::Matched lower(F f) {
        if (perspectives_ == 0) throw Error("Must set single_perspective or dual_perspective!");
        if (!has_input_) throw Error("Must set inputs!");
        if (!has_loss_) throw Error("Must set loss function!");
        ModelBuilder builder;
        const int nnz = bt_input_max_active(&input_), fin = bt_input_num_inputs(&input_);
        Node stm = Node::make(graph::Kind::Stm, fin, nullptr, nullptr);
        Node ntm = Node::make(graph::Kind::Ntm, fin, nullptr, nullptr);
        Node bk = Node::make(graph::Kind::Buckets, 1, nullptr, nullptr);
        bk.p->buckets = buckets_;
        Node out = call(f, builder, stm, ntm, bk);
        graph::Matched m = graph::match(out.p, perspectives_, buckets_, nnz, loss_);
        if (m.desc.num_inputs != fin) throw Error("Mismatched sizes!");  // l0 inputs vs inputs.num_inputs()
        if (wdl_output_ && m.desc.loss_kind != BT_LOSS_SOFTMAX_CE) throw Error("wdl_output needs softmax_crossentropy_loss");
        m.desc.ft_segment = segment_;
        m.desc.use_entry_weights = weight_fn_ ? 1 : 0;  // `loss * entry_weights` only then (value/builder.rs:137-140)
        return m;
    }

  private:
    template <class F>
    Node call(F &f, ModelBuilder &b, Node stm, Node ntm, Node bk) {
        if constexpr (std::is_invocable<F, ModelBuilder &, Node, Node, Node>::value) {
            if (perspectives_ != 2 || buckets_ <= 1) throw Error("this closure shape needs dual_perspective + output_buckets");
            return f(b, stm, ntm, bk);
        } else if constexpr (std::is_invocable<F, ModelBuilder &, Node, Node>::value) {
            if (perspectives_ == 2) {
                if (buckets_ > 1) throw Error("output_buckets set: the closure must take the buckets input");
                return f(b, stm, ntm);
            }
            if (buckets_ <= 1) throw Error("single_perspective without output buckets: the closure takes (builder, stm)");
            return f(b, stm, bk);
        } else {
            static_assert(std::is_invocable<F, ModelBuilder &, Node>::value, "build closure: (builder, stm[, ntm][, buckets])");
            if (perspectives_ != 1 || buckets_ > 1) throw Error("the closure must take every input of the net");
            return f(b, stm);
        }
    }
    int perspectives_ = 0, opt_ = BT_OPT_ADAMW, buckets_ = 1, segment_ = 0, device_ = 0;
    bt_input_desc input_{};
    bool has_input_ = false, has_loss_ = false, wdl_output_ = false, win_rate_ = false;
    std::vector<SavedFormat> fmt_;
    LossValue loss_{BT_LOSS_SIGMOID_MSE, 2.0f};
    BlendFn blend_fn_;
    WeightFn weight_fn_;
    uint64_t seed_ = 198273612;  // value/builder.rs:52
};

}  // namespace bullet_lib

#endif
