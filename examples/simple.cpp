// The C++ counterpart of the reference's examples/simple.rs, line for line:
//     (768 -> HIDDEN_SIZE)x2 -> 1, SCReLU, AdamW, StepLR, ConstantWDL, checkpoints every 10 superbatches.
// Build: g++ -std=c++17 -O2 -Iinclude examples/simple.cpp -Lbullet_amd -lbullet_amd -Wl,-rpath,$PWD/bullet_amd
// Usage: simple <data.bf> [superbatches] [batches_per_superbatch]
#include <cstdlib>

#include "bullet_amd.hpp"

using namespace bullet_lib;

constexpr int HIDDEN_SIZE = 128;
constexpr int SCALE = 400;
constexpr int QA = 255;
constexpr int QB = 64;

int main(int argc, char **argv) {
    const char *data = argc > 1 ? argv[1] : "data/baseline.data";
    const int superbatches = argc > 2 ? std::atoi(argv[2]) : 40;
    const int batches_per_superbatch = argc > 3 ? std::atoi(argv[3]) : 6104;
    try {
        auto trainer = ValueTrainerBuilder()
                           // makes `ntm_inputs` available below
                           .dual_perspective()
                           // the default AdamW params include clipping to range [-1.98, 1.98]
                           .optimiser(optimiser::AdamW{})
                           // basic piece-square chessboard inputs
                           .inputs(inputs::Chess768{})
                           // chosen such that inference may be efficiently implemented in-engine
                           .save_format({
                               SavedFormat::id("l0w").round().quantise<int16_t>(QA),
                               SavedFormat::id("l0b").round().quantise<int16_t>(QA),
                               SavedFormat::id("l1w").round().quantise<int16_t>(QB),
                               SavedFormat::id("l1b").round().quantise<int16_t>(QA * QB),
                           })
                           // `target` == wdl * game_result + (1 - wdl) * sigmoid(search score in centipawns / SCALE)
                           .loss_fn([](auto output, auto target) { return output.sigmoid().squared_error(target); })
                           // the basic `(768 -> N)x2 -> 1` inference
                           .build([](ModelBuilder &builder, Node stm_inputs, Node ntm_inputs) {
                               // weights
                               auto l0 = builder.new_affine("l0", 768, HIDDEN_SIZE);
                               auto l1 = builder.new_affine("l1", 2 * HIDDEN_SIZE, 1);
                               // inference
                               auto stm_hidden = l0.forward(stm_inputs).screlu();
                               auto ntm_hidden = l0.forward(ntm_inputs).screlu();
                               auto hidden_layer = stm_hidden.concat(ntm_hidden);
                               return l1.forward(hidden_layer);
                           });

        TrainingSchedule<lr::StepLR, wdl::ConstantWDL> schedule{
            "simple",
            (float)SCALE,
            TrainingSteps{16384, batches_per_superbatch, 1, superbatches},
            wdl::ConstantWDL{0.75f},
            lr::StepLR{0.001f, 0.1f, 18},
            10,
        };
        LocalSettings settings{4, nullptr, "checkpoints", 64};
        // loading from a Viriformat binpack (custom filters only: the crate's built-in `Filter` is not restated)
        loader::ViriBinpackLoader data_loader_viri("data/run.binpack", 1024, 4);
        // loading from a SF binpack, with the filter of the reference's example
        loader::SfBinpackLoader data_loader_sf("data/test80.binpack", 1024, 4, [](const bt_sf_entry &entry) {
            return entry.ply >= 16 && !entry.in_check && std::abs(entry.score) <= 10000 && entry.move_type == 0 &&
                   entry.captured_piece == 7;
        });
        (void)data_loader_viri, (void)data_loader_sf;
        // loading directly from a `BulletFormat` file
        loader::DirectSequentialDataLoader data_loader({data});
        trainer->run(schedule, settings, data_loader);
    } catch (const Error &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
