// examples/tests/test2.rs of the reference in C++: mirrored king buckets (2), factoriser, pairwise CReLU feature
// transformer, l1 with a skip neuron and the dual activation concat(x, x*x).crelu(), bucketed l2 / l3,
// LinearWDL 0.2 -> 0, DropLR. The reference prints "Expected loss: 0.052189" (test2.rs:86) from its GPU backend,
// whose GEMM summation order is not pinned: the bar here is 1.5e-6 absolute (oracle and parity).
#include <cmath>

#include "bullet_amd.hpp"

using namespace bullet_lib;

int main(int argc, char **argv) {
    const char *data = argc > 1 ? argv[1] : "tests/golden/batch.bf";
    const std::string out_dir = argc > 2 ? argv[2] : "checkpoints";
    const std::array<int, 32> bucket_layout = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                               1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    const int hl_size = 128;
    const int num_input_buckets = inputs::get_num_buckets(bucket_layout);
    const int num_inputs = 768 * num_input_buckets;
    const int num_output_buckets = 8;
    try {
        auto trainer =
            ValueTrainerBuilder()
                .dual_perspective()
                .optimiser(optimiser::AdamW{})
                .inputs(inputs::ChessBucketsMirrored(bucket_layout))
                .output_buckets(outputs::MaterialCount<8>{})
                .save_format({})
                .loss_fn([](auto output, auto target) { return output.sigmoid().squared_error(target); })
                .build([&](ModelBuilder &builder, Node stm_inputs, Node ntm_inputs, Node buckets) {
                    auto l0f = builder.new_weights("l0f", Shape{hl_size, 768}, InitSettings::Zeroed);
                    auto expanded_factoriser = l0f.repeat(num_input_buckets);

                    auto l0 = builder.new_affine("l0", num_inputs, hl_size);
                    l0.weights = l0.weights + expanded_factoriser;

                    auto l1 = builder.new_affine("l1", hl_size, num_output_buckets * 16);
                    auto l2 = builder.new_affine("l2", 30, num_output_buckets * 32);
                    auto l3 = builder.new_affine("l3", 32, num_output_buckets);

                    auto ft = [&](Node input, int start, int end) { return l0.slice(start, end).forward(input).crelu(); };
                    auto stm_subnet = ft(stm_inputs, 0, hl_size / 2) * ft(stm_inputs, hl_size / 2, hl_size);
                    auto ntm_subnet = ft(ntm_inputs, 0, hl_size / 2) * ft(ntm_inputs, hl_size / 2, hl_size);
                    auto out = stm_subnet.concat(ntm_subnet);

                    out = l1.forward(out).select(buckets);

                    auto skip_neuron = out.slice_rows(15, 16);

                    out = out.slice_rows(0, 15);
                    out = out.concat(out * out).crelu();

                    out = l2.forward(out).select(buckets).screlu();
                    out = l3.forward(out).select(buckets);

                    return out + skip_neuron;
                });
        TrainingSchedule<lr::DropLR, wdl::LinearWDL> schedule{
            "test2", 400.0f, TrainingSteps{16384, 1, 1, 10}, wdl::LinearWDL{0.2f, 0.0f}, lr::DropLR{0.001f, 0.1f, 5}, 10};
        LocalSettings settings{4, nullptr, out_dir, 512};
        loader::DirectSequentialDataLoader data_loader({data});
        std::printf("Eval: %.3fcp\n", 400.0f * trainer->eval("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1 | 0 | 0.0"));
        trainer->run(schedule, settings, data_loader);
        std::printf("Expected loss: 0.052189\n");
        std::printf("Final loss: %.7f\n", trainer->last_superbatch_loss());
        std::printf("Eval: %.3fcp\n", 400.0f * trainer->eval("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1 | 0 | 0.0"));
        return std::fabs(trainer->last_superbatch_loss() - 0.052189f) <= 1.5e-6f ? 0 : 2;
    } catch (const Error &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
