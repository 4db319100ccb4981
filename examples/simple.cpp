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
                           // two perspectives: the builder callback below receives `ntm_inputs` too
                           .dual_perspective()
                           // AdamW with its defaults (weights are clipped to [-1.98, 1.98])
                           .optimiser(optimiser::AdamW{})
                           // 768 inputs: piece type x colour x square
                           .inputs(inputs::Chess768{})
                           // QA and QB: scales an engine can run in 16-bit integers
                           .save_format({
                               SavedFormat::id("l0w").round().quantise<int16_t>(QA),
                               SavedFormat::id("l0b").round().quantise<int16_t>(QA),
                               SavedFormat::id("l1w").round().quantise<int16_t>(QB),
                               SavedFormat::id("l1b").round().quantise<int16_t>(QA * QB),
                           })
                           // the training target blends the game result (weight wdl) with sigmoid(score in centipawns / SCALE)
                           .loss_fn([](auto output, auto target) { return output.sigmoid().squared_error(target); })
                           // (768 -> HIDDEN) per perspective, both halves concatenated, -> 1
                           .build([](ModelBuilder &builder, Node stm_inputs, Node ntm_inputs) {
                               // the two layers
                               auto l0 = builder.new_affine("l0", 768, HIDDEN_SIZE);
                               auto l1 = builder.new_affine("l1", 2 * HIDDEN_SIZE, 1);
                               // forward pass
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
        // loading from a Viriformat binpack: the built-in filter with its default settings, or any function
        // (board, move, eval, wdl) -> keep? in its place, as for SF binpacks
        loader::ViriBinpackLoader data_loader_viri("data/run.binpack", 1024, 4, loader::Filter::standard());
        // loading from a SF binpack, with the filter of the reference's example
        loader::SfBinpackLoader data_loader_sf("data/test80.binpack", 1024, 4, [](const bt_sf_entry &entry) {
            return entry.ply >= 16 && !entry.in_check && std::abs(entry.score) <= 10000 && entry.move_type == 0 &&
                   entry.captured_piece == 7;
        });
        (void)data_loader_viri, (void)data_loader_sf;
        // ... or records already in bulletformat, read in file order
        loader::DirectSequentialDataLoader data_loader({data});
        trainer->run(schedule, settings, data_loader);
    } catch (const Error &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
