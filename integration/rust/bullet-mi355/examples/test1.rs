//! The reference's examples/tests/test1.rs on the MI355X backend: 768 -> 32x2 -> 1 SCReLU, 10 superbatches of one
//! batch of 16384 positions from examples/tests/batch.bf; the reference prints "Expected loss: 0.056420".
use bullet_mi355::{ffi, train, Net};

fn main() {
    let mut desc = ffi::bt_net_desc::default();
    desc.num_inputs = 768;
    desc.nnz = 32;
    desc.ft_out = 32;
    desc.ft_act = 3; // BT_ACT_SCRELU
    desc.perspectives = 2;
    desc.num_buckets = 1;
    desc.n_dense = 1;
    desc.dense_out[0] = 1;
    let mut net = Net::new(desc, 16_384).unwrap();
    net.init_weights(198273612, 0).unwrap();

    let mut input: ffi::bt_input_desc = unsafe { std::mem::zeroed() };
    assert_eq!(unsafe { ffi::bt_input_init(&mut input, 0, 0, std::ptr::null(), 0) }, 0); // BT_INPUT_CHESS768

    let mut cfg: ffi::bt_train_config = unsafe { std::mem::zeroed() };
    cfg.batch_size = 16_384;
    cfg.batches_per_superbatch = 1;
    cfg.start_superbatch = 1;
    cfg.end_superbatch = 10;
    cfg.lr = ffi::bt_lr_sched { kind: 0, a: 0.001, b: 0.0, n: 0, warmup_batches: 0 }; // ConstantLR
    cfg.wdl = ffi::bt_wdl_sched { kind: 0, start: 0.2, end: 0.2, warmup_batches: 0 }; // ConstantWDL
    cfg.eval_scale = 400.0;
    cfg.out_buckets = 1;
    cfg.threads = 4;
    cfg.log_rate = 128;

    let losses = train(&mut net, &input, &cfg, &["examples/tests/batch.bf"]).unwrap();
    println!("Final loss: {:.6}", losses.last().unwrap());
    println!("Expected loss: 0.056420");
}
