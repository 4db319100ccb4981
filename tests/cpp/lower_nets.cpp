// Host-only check of the graph matcher in include/bullet_amd.hpp: the closures of the reference's examples
// (written as in examples/*.rs) must lower to the bt_net_desc the kernels were parity-tested with. Prints one line
// per net: name and the descriptor fields; tests/test_cpp_api.py compares them with bullet_amd.lib.make_net_desc.
#include "bullet_amd.hpp"

using namespace bullet_lib;

static void show(const char *name, const graph::Matched &m) {
    const bt_net_desc &d = m.desc;
    std::printf("%s num_inputs=%d nnz=%d ft_out=%d ft_act=%d pairwise=%d perspectives=%d num_buckets=%d n_dense=%d", name,
                d.num_inputs, d.nnz, d.ft_out, d.ft_act, d.pairwise, d.perspectives, d.num_buckets, d.n_dense);
    for (int k = 0; k < d.n_dense; k++)
        std::printf(" L%d=%d/%d/%d/%d/%d", k, d.dense_out[k], d.dense_act[k], d.dense_bucketed[k], d.dense_dual[k],
                    d.dense_skip[k]);
    std::printf(" ft_factoriser=%d ft_segment=%d loss_kind=%d loss_power=%g eff=%d\n", d.ft_factoriser, d.ft_segment,
                d.loss_kind, d.loss_power, m.ft_effective_inputs);
}

static auto mse = [](auto output, auto target) { return output.sigmoid().squared_error(target); };

int main() {
    // examples/simple.rs
    show("simple", ValueTrainerBuilder().dual_perspective().inputs(inputs::Chess768{}).loss_fn(mse).lower(
                       [](ModelBuilder &b, Node stm, Node ntm) {
                           auto l0 = b.new_affine("l0", 768, 128);
                           auto l1 = b.new_affine("l1", 256, 1);
                           return l1.forward(l0.forward(stm).screlu().concat(l0.forward(ntm).screlu()));
                       }));
    // examples/progression/2_output_buckets.rs with hl = 1024 (BASELINE configs[1])
    show("output_buckets", ValueTrainerBuilder().dual_perspective().inputs(inputs::Chess768{})
                               .output_buckets(outputs::MaterialCount<8>{}).loss_fn(mse).lower(
                                   [](ModelBuilder &b, Node stm, Node ntm, Node buckets) {
                                       auto l0 = b.new_affine("l0", 768, 1024);
                                       auto l1 = b.new_affine("l1", 2048, 8);
                                       auto h = l0.forward(stm).screlu().concat(l0.forward(ntm).screlu());
                                       return l1.forward(h).select(buckets);
                                   }));
    // examples/progression/4_multi_layer.rs
    const std::array<int, 32> layout = {0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7,
                                        8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9};
    const int NB = inputs::get_num_buckets(layout), hl = 1024;
    show("multi_layer", ValueTrainerBuilder().dual_perspective().inputs(inputs::ChessBucketsMirrored(layout))
                            .output_buckets(outputs::MaterialCount<8>{}).loss_fn(mse).lower(
                                [&](ModelBuilder &b, Node stm, Node ntm, Node buckets) {
                                    auto l0f = b.new_weights("l0f", Shape{hl, 768}, InitSettings::Zeroed);
                                    auto l0 = b.new_affine("l0", 768 * NB, hl);
                                    l0.init_with_effective_input_size(32);
                                    l0.weights = l0.weights + l0f.repeat(NB);
                                    auto l1 = b.new_affine("l1", hl, 8 * 16);
                                    auto l2 = b.new_affine("l2", 16, 8 * 32);
                                    auto l3 = b.new_affine("l3", 32, 8);
                                    auto ft = [&](Node in, int s, int e) { return l0.slice(s, e).forward(in).crelu(); };
                                    auto sh = ft(stm, 0, hl / 2) * ft(stm, hl / 2, hl);
                                    auto nh = ft(ntm, 0, hl / 2) * ft(ntm, hl / 2, hl);
                                    auto hl2 = l1.forward(sh.concat(nh)).select(buckets).screlu();
                                    auto hl3 = l2.forward(hl2).select(buckets).screlu();
                                    return l3.forward(hl3).select(buckets);
                                }));
    // the `.pairwise_mul()` spelling of the same feature transformer, single layer, power loss
    show("pairwise_mul", ValueTrainerBuilder().dual_perspective().inputs(inputs::Chess768{})
                             .loss_fn([](auto o, auto t) { return o.sigmoid().power_error(t, 2.5f); }).lower(
                                 [](ModelBuilder &b, Node stm, Node ntm) {
                                     auto l0 = b.new_affine("l0", 768, 64);
                                     auto l1 = b.new_affine("l1", 64, 1);
                                     auto h = l0.forward(stm).crelu().pairwise_mul().concat(l0.forward(ntm).crelu().pairwise_mul());
                                     return l1.forward(h);
                                 }));
    // examples/tests/test2.rs
    const std::array<int, 32> l2b = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    show("test2", ValueTrainerBuilder().dual_perspective().inputs(inputs::ChessBucketsMirrored(l2b))
                      .output_buckets(outputs::MaterialCount<8>{}).loss_fn(mse).lower(
                          [&](ModelBuilder &b, Node stm, Node ntm, Node buckets) {
                              auto l0f = b.new_weights("l0f", Shape{128, 768}, InitSettings::Zeroed);
                              auto l0 = b.new_affine("l0", 768 * 2, 128);
                              l0.weights = l0.weights + l0f.repeat(2);
                              auto l1 = b.new_affine("l1", 128, 8 * 16);
                              auto l2 = b.new_affine("l2", 30, 8 * 32);
                              auto l3 = b.new_affine("l3", 32, 8);
                              auto ft = [&](Node in, int s, int e) { return l0.slice(s, e).forward(in).crelu(); };
                              auto out = (ft(stm, 0, 64) * ft(stm, 64, 128)).concat(ft(ntm, 0, 64) * ft(ntm, 64, 128));
                              out = l1.forward(out).select(buckets);
                              auto skip = out.slice_rows(15, 16);
                              out = out.slice_rows(0, 15);
                              out = out.concat(out * out).crelu();
                              out = l2.forward(out).select(buckets).screlu();
                              out = l3.forward(out).select(buckets);
                              return out + skip;
                          }));
    // single perspective + W/D/L head (examples/ataxx.rs net shape through the value builder)
    show("single_wdl", ValueTrainerBuilder().single_perspective().inputs(inputs::Ataxx147{}).wdl_output()
                           .loss_fn([](auto o, auto t) { return o.softmax_crossentropy_loss(t); }).lower(
                               [](ModelBuilder &b, Node stm) {
                                   auto l0 = b.new_affine("l0", 147, 256);
                                   auto l1 = b.new_affine("l1", 256, 3);
                                   return l1.forward(l0.forward(stm).screlu());
                               }));
    // graphs outside the family are rejected, not approximated
    int rejected = 0;
    try {
        ValueTrainerBuilder().dual_perspective().inputs(inputs::Chess768{}).loss_fn(mse).lower(
            [](ModelBuilder &b, Node stm, Node ntm) {
                auto l0 = b.new_affine("l0", 768, 32);
                auto l1 = b.new_affine("l1", 64, 1);
                auto h = l0.forward(stm).screlu().concat(l0.forward(ntm).crelu());  // different activations
                return l1.forward(h);
            });
    } catch (const Error &) { rejected++; }
    try {
        ValueTrainerBuilder().dual_perspective().inputs(inputs::Chess768{}).loss_fn(mse).lower(
            [](ModelBuilder &b, Node stm, Node ntm) {
                auto l0 = b.new_affine("l0", 768, 32);
                auto l1 = b.new_affine("l1", 60, 1);                                 // wrong input size
                return l1.forward(l0.forward(stm).screlu().concat(l0.forward(ntm).screlu()));
            });
    } catch (const Error &) { rejected++; }
    try {
        ValueTrainerBuilder().dual_perspective().inputs(inputs::Chess768{}).loss_fn(mse).lower(
            [](ModelBuilder &b, Node stm, Node ntm) {
                auto l0 = b.new_affine("l0", 768, 32);
                auto l1 = b.new_affine("l1", 64, 1);
                auto h = l0.forward(stm).screlu().concat(l0.forward(ntm).screlu());
                return l1.forward(h * h);                                            // an op the kernels do not have
            });
    } catch (const Error &) { rejected++; }
    // schedulers (trainer/schedule/{lr,wdl}.rs): Sequence switches at the midpoint and re-bases the superbatch;
    // Warmup divides by (warmup_batches - batch) during superbatch 1
    lr::Sequence<lr::Warmup<lr::ConstantLR>, lr::LinearDecayLR> seq{{{0.001f}, 10}, {0.001f, 0.0001f, 10}, 4};
    wdl::Sequence<wdl::ConstantWDL, wdl::LinearWDL> wseq{{0.3f}, {0.2f, 0.0f}, 4};
    std::printf("sched %.9g %.9g %.9g %.9g %.9g | %.9g %.9g %.9g\n", seq.lr(0, 1), seq.lr(9, 1), seq.lr(3, 4), seq.lr(0, 5),
                seq.lr(0, 14), wseq.blend(0, 2, 14), wseq.blend(0, 5, 14), wseq.blend(0, 14, 14));
    std::printf("rejected=%d\n", rejected);
    return 0;
}
