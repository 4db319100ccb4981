// The net of the reference's examples/ataxx.rs through the value-trainer mirror: Ataxx147 inputs
// (147 features, up to 49 active) -> 256x2 SCReLU -> 1, on 32-byte AtaxxBoard records.
// Usage: ataxx <data file of AtaxxBoard records> [output dir] [superbatches] [batches_per_superbatch]
#include <cstdlib>

#include "bullet_amd.hpp"

using namespace bullet_lib;

int main(int argc, char **argv) {
    const char *data = argc > 1 ? argv[1] : "data/ataxx.data";
    const std::string out_dir = argc > 2 ? argv[2] : "checkpoints";
    const int superbatches = argc > 3 ? std::atoi(argv[3]) : 40;
    const int batches_per_superbatch = argc > 4 ? std::atoi(argv[4]) : 1024;
    constexpr int HIDDEN_SIZE = 256;
    try {
        auto trainer = ValueTrainerBuilder()
                           .dual_perspective()
                           .optimiser(optimiser::AdamW{})
                           .inputs(inputs::Ataxx147{})
                           .save_format({
                               SavedFormat::id("l0w").round().quantise<int16_t>(255),
                               SavedFormat::id("l0b").round().quantise<int16_t>(255),
                               SavedFormat::id("l1w").round().quantise<int16_t>(64),
                               SavedFormat::id("l1b").round().quantise<int16_t>(255 * 64),
                           })
                           .loss_fn([](auto output, auto target) { return output.sigmoid().squared_error(target); })
                           .build([](ModelBuilder &builder, Node stm, Node ntm) {
                               auto l0 = builder.new_affine("l0", 147, HIDDEN_SIZE);
                               auto l1 = builder.new_affine("l1", 2 * HIDDEN_SIZE, 1);
                               return l1.forward(l0.forward(stm).screlu().concat(l0.forward(ntm).screlu()));
                           });
        TrainingSchedule<lr::StepLR, wdl::ConstantWDL> schedule{
            "ataxx", 400.0f, TrainingSteps{16384, batches_per_superbatch, 1, superbatches},
            wdl::ConstantWDL{0.5f}, lr::StepLR{0.001f, 0.1f, 15}, 10};
        LocalSettings settings{4, nullptr, out_dir, 64};
        loader::DirectSequentialDataLoader data_loader({data});
        trainer->run(schedule, settings, data_loader);
        std::printf("Final loss: %.6f\n", trainer->last_superbatch_loss());
    } catch (const Error &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
