// examples/tests/test1.rs of the reference in C++: 768 -> 32x2 -> 1 SCReLU, 10 superbatches of one batch of 16384
// positions from examples/tests/batch.bf (here tests/golden/batch.bf), ConstantLR 1e-3, ConstantWDL 0.2.
// The reference prints "Expected loss: 0.056420" (test1.rs:54); this program prints the loss it reached and exits
// non-zero when the six digits differ.
#include <cstdlib>

#include "bullet_amd.hpp"

using namespace bullet_lib;

int main(int argc, char **argv) {
    const char *data = argc > 1 ? argv[1] : "tests/golden/batch.bf";
    const std::string out_dir = argc > 2 ? argv[2] : "checkpoints";
    // examples/tests/test3.rs is the same program with a 128-wide hidden layer: `test1 <data> <dir> 128 0.044886`
    const int hidden = argc > 3 ? std::atoi(argv[3]) : 32;
    const std::string expected = argc > 4 ? argv[4] : "0.056420";
    try {
        auto trainer = ValueTrainerBuilder()
                           .dual_perspective()
                           .optimiser(optimiser::AdamW{})
                           .inputs(inputs::Chess768{})
                           .save_format({
                               SavedFormat::id("l0w").round().quantise<int16_t>(255),
                               SavedFormat::id("l0b").round().quantise<int16_t>(255),
                               SavedFormat::id("l1w").round().quantise<int16_t>(64),
                               SavedFormat::id("l1b").round().quantise<int16_t>(255 * 64),
                           })
                           .loss_fn([](auto output, auto target) { return output.sigmoid().squared_error(target); })
                           .build([&](ModelBuilder &builder, Node stm_inputs, Node ntm_inputs) {
                               auto l0 = builder.new_affine("l0", 768, hidden);
                               auto l1 = builder.new_affine("l1", 2 * hidden, 1);
                               auto stm_hidden = l0.forward(stm_inputs).screlu();
                               auto ntm_hidden = l0.forward(ntm_inputs).screlu();
                               return l1.forward(stm_hidden.concat(ntm_hidden));
                           });
        TrainingSchedule<lr::ConstantLR, wdl::ConstantWDL> schedule{
            "test1", 400.0f, TrainingSteps{16384, 1, 1, 10}, wdl::ConstantWDL{0.2f}, lr::ConstantLR{0.001f}, 10};
        LocalSettings settings{4, nullptr, out_dir, 512};
        loader::DirectSequentialDataLoader data_loader({data});
        const float before = 400.0f * trainer->eval("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1");
        std::printf("Eval: %.3fcp\n", before);
        trainer->run(schedule, settings, data_loader);
        std::printf("Expected loss: %s\n", expected.c_str());
        std::printf("Final loss: %.6f\n", trainer->last_superbatch_loss());
        std::printf("Eval: %.3fcp\n", 400.0f * trainer->eval("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"));
        // optimiser API round trip: state written, weights scrambled, state loaded back -> same eval; standalone writers
        const float eval_trained = trainer->eval("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1");
        trainer->optimiser.write_to_checkpoint(out_dir + "/manual_state");
        trainer->optimiser.load_weights_from_file(out_dir + "/test1-10/optimiser_state/weights.bin");
        trainer->optimiser.reset_state();
        trainer->optimiser.load_from_checkpoint(out_dir + "/manual_state");
        trainer->save_quantised(out_dir + "/manual_quantised.bin");
        trainer->save_unquantised(out_dir + "/manual_raw.bin");
        if (trainer->eval("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1") != eval_trained) return 3;
        char got[32];
        std::snprintf(got, sizeof(got), "%.6f", trainer->last_superbatch_loss());
        return std::string(got) == expected ? 0 : 2;
    } catch (const Error &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
