// The C++ counterpart of the reference's examples/progression/4_multi_layer.rs:
// mirrored king buckets (10) with a factoriser -> 1024x2 pairwise CReLU -> 16 -> 32 -> 1, 8 output buckets.
// Usage: 4_multi_layer <data.bf> [output dir] [superbatches] [batches_per_superbatch]
#include <cmath>
#include <cstdlib>

#include "bullet_amd.hpp"

using namespace bullet_lib;

int main(int argc, char **argv) {
    // the knobs of this run
    const int hl_size = 1024;
    const int l2_size = 16;
    const char *dataset_path = argc > 1 ? argv[1] : "data/baseline.data";
    const std::string out_dir = argc > 2 ? argv[2] : "checkpoints";
    const float initial_lr = 0.001f;
    const float final_lr = 0.001f * std::pow(0.3f, 5.0f);
    const int superbatches = argc > 3 ? std::atoi(argv[3]) : 800;
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
                    // weights + l0f.repeat(NUM_INPUT_BUCKETS): the `.transform` closure of the reference
                    SavedFormat::id("l0w").merge_factoriser().round().quantise<int16_t>(255),
                    SavedFormat::id("l0b").round().quantise<int16_t>(255),
                    SavedFormat::id("l1w").transpose().round().quantise<int8_t>(64),
                    SavedFormat::id("l1b"),
                    SavedFormat::id("l2w").transpose(),
                    SavedFormat::id("l2b"),
                    SavedFormat::id("l3w").transpose(),
                    SavedFormat::id("l3b"),
                })
                .loss_fn([](auto output, auto target) { return output.sigmoid().squared_error(target); })
                .build([&](ModelBuilder &builder, Node stm_inputs, Node ntm_inputs, Node output_buckets) {
                    auto l0f = builder.new_weights("l0f", Shape{hl_size, 768}, InitSettings::Zeroed);
                    auto l0 = builder.new_affine("l0", 768 * NUM_INPUT_BUCKETS, hl_size);
                    l0.init_with_effective_input_size(32);
                    l0.weights = l0.weights + l0f.repeat(NUM_INPUT_BUCKETS);

                    auto l1 = builder.new_affine("l1", hl_size, NUM_OUTPUT_BUCKETS * l2_size);
                    auto l2 = builder.new_affine("l2", l2_size, NUM_OUTPUT_BUCKETS * 32);
                    auto l3 = builder.new_affine("l3", 32, NUM_OUTPUT_BUCKETS);

                    // l0.forward(inputs).crelu().pairwise_mul(), written over the two halves
                    auto ft = [&](Node input, int start, int end) { return l0.slice(start, end).forward(input).crelu(); };
                    auto stm_hidden = ft(stm_inputs, 0, hl_size / 2) * ft(stm_inputs, hl_size / 2, hl_size);
                    auto ntm_hidden = ft(ntm_inputs, 0, hl_size / 2) * ft(ntm_inputs, hl_size / 2, hl_size);

                    auto hl1 = stm_hidden.concat(ntm_hidden);
                    auto hl2 = l1.forward(hl1).select(output_buckets).screlu();
                    auto hl3 = l2.forward(hl2).select(output_buckets).screlu();
                    return l3.forward(hl3).select(output_buckets);
                });

        // the effective input weight is l0w + l0f, so each of the two gets half of the clipping range
        optimiser::AdamWParams stricter_clipping;
        stricter_clipping.max_weight = 0.99f, stricter_clipping.min_weight = -0.99f;
        trainer->optimiser.set_params_for_weight("l0w", stricter_clipping);
        trainer->optimiser.set_params_for_weight("l0f", stricter_clipping);

        TrainingSchedule<lr::CosineDecayLR, wdl::ConstantWDL> schedule{
            "4_multi_layer",
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
