// The C++ counterpart of the reference's examples/progression/3_input_buckets.rs:
// mirrored king buckets (10) with a factoriser -> 1024x2 SCReLU -> 1, 8 output buckets.
// Usage: 3_input_buckets <data.bf> [output dir] [superbatches] [batches_per_superbatch]
#include <cmath>
#include <cstdlib>

#include "bullet_amd.hpp"

using namespace bullet_lib;

int main(int argc, char **argv) {
    // the knobs of this run
    const int hl_size = 1024;
    const char *dataset_path = argc > 1 ? argv[1] : "data/baseline.data";
    const std::string out_dir = argc > 2 ? argv[2] : "checkpoints";
    const float initial_lr = 0.001f;
    const float final_lr = 0.001f * std::pow(0.3f, 5.0f);
    const int superbatches = argc > 3 ? std::atoi(argv[3]) : 640;
    const int batches_per_superbatch = argc > 4 ? std::atoi(argv[4]) : 6104;
    const float wdl_proportion = 0.75f;
    constexpr int NUM_OUTPUT_BUCKETS = 8;
    const std::array<int, 32> BUCKET_LAYOUT = {0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7,
                                               8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9};
    const int NUM_INPUT_BUCKETS = inputs::get_num_buckets(BUCKET_LAYOUT);
    try {
        auto trainer =
            ValueTrainerBuilder()
                .dual_perspective()
                .optimiser(optimiser::AdamW{})
                .inputs(inputs::ChessBucketsMirrored(BUCKET_LAYOUT))
                .output_buckets(outputs::MaterialCount<NUM_OUTPUT_BUCKETS>{})
                .save_format({
                    // what the engine reads: each bucket's columns with the shared (factoriser) columns added in
                    SavedFormat::id("l0w").merge_factoriser().round().quantise<int16_t>(255),
                    SavedFormat::id("l0b").round().quantise<int16_t>(255),
                    SavedFormat::id("l1w").round().quantise<int16_t>(64).transpose(),
                    SavedFormat::id("l1b").round().quantise<int16_t>(255 * 64),
                })
                .loss_fn([](auto output, auto target) { return output.sigmoid().squared_error(target); })
                .build([&](ModelBuilder &builder, Node stm_inputs, Node ntm_inputs, Node output_buckets) {
                    // l0f: the columns every king bucket shares
                    auto l0f = builder.new_weights("l0f", Shape{hl_size, 768}, InitSettings::Zeroed);
                    auto expanded_factoriser = l0f.repeat(NUM_INPUT_BUCKETS);

                    // l0: one set of columns per king bucket, and the bias
                    auto l0 = builder.new_affine("l0", 768 * NUM_INPUT_BUCKETS, hl_size);
                    l0.weights = l0.weights + expanded_factoriser;

                    // l1: one output per material bucket
                    auto l1 = builder.new_affine("l1", 2 * hl_size, NUM_OUTPUT_BUCKETS);

                    // forward pass
                    auto stm_hidden = l0.forward(stm_inputs).screlu();
                    auto ntm_hidden = l0.forward(ntm_inputs).screlu();
                    auto hidden_layer = stm_hidden.concat(ntm_hidden);
                    return l1.forward(hidden_layer).select(output_buckets);
                });

        // the effective input weight is l0w + l0f, so each of the two gets half of the clipping range
        optimiser::AdamWParams stricter_clipping;
        stricter_clipping.max_weight = 0.99f, stricter_clipping.min_weight = -0.99f;
        trainer->optimiser.set_params_for_weight("l0w", stricter_clipping);
        trainer->optimiser.set_params_for_weight("l0f", stricter_clipping);

        TrainingSchedule<lr::CosineDecayLR, wdl::ConstantWDL> schedule{
            "3_input_buckets",
            400.0f,
            TrainingSteps{16384, batches_per_superbatch, 1, superbatches},
            wdl::ConstantWDL{wdl_proportion},
            lr::CosineDecayLR{initial_lr, final_lr, superbatches},
            10,
        };
        LocalSettings settings{2, nullptr, out_dir, 32};
        loader::DirectSequentialDataLoader dataloader({dataset_path});
        trainer->run(schedule, settings, dataloader);
        std::printf("Final loss: %.6f\n", trainer->last_superbatch_loss());
    } catch (const Error &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
