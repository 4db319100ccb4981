/*
 * bullet_oracle.h -- CPU restatement of the jw1912/bullet NNUE training hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: it may be
 * imported / linked / executed only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py, and only as the checker.  The product path
 * (bullet_amd/) never calls into it and has no CPU fallback.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose semantics it restates.  The reference is Rust; there is no toolchain for
 * it here, so this restatement is pinned by the reference's own in-tree golden
 * vectors (tests/test_oracle_golden.py) and by the e2e fixture
 * examples/tests/batch.bf + printed losses (tests/test_e2e_reference_losses.py).
 */
#ifndef BULLET_ORACLE_H
#define BULLET_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* activation ids shared with include/bullet_amd.h (values must match) */
enum {
    ORC_ACT_IDENTITY = 0,
    ORC_ACT_RELU = 1,
    ORC_ACT_CRELU = 2,
    ORC_ACT_SCRELU = 3,
    ORC_ACT_SQRRELU = 4,
};

/* ------------------------------------------------------------------ ops */

/* SparseMatmul::evaluate, crates/compiler/src/tensor/operation/linear/sparse.rs:86-119.
 * brk=0: literal CPU semantics (skip out-of-range index, continue);
 * brk=1: device semantics (stop at first out-of-range index),
 *        crates/gpu/src/pointwise/write.rs:229-233. */
void orc_sparse_matmul_fwd(const float *w, const int32_t *idx, float *out, int batch, int rows, int cols, int stride,
                           int offset, int nnz, int brk);

/* SparseMatmulBwd::evaluate, sparse.rs:151-187 (zeroes dw[stride*cols] first). */
void orc_sparse_matmul_bwd(const float *g, const int32_t *idx, float *dw, int batch, int rows, int cols, int stride,
                           int offset, int nnz, int brk);

/* SparseMatmulBwdMulti::evaluate, sparse.rs:253-271: out = 0 + bwd_0 + bwd_1 + ... */
void orc_sparse_matmul_bwd_multi(int n, const float *const *g, const int32_t *const *idx, float *dw, int batch,
                                 const int *rows, int cols, int stride, const int *offset, int nnz, int brk);

/* Matmul::evaluate, crates/compiler/src/tensor/operation/linear/matmul.rs:62-93.
 * l is m x n, r is n x k, o is m x k (column-major), batched b times. */
void orc_matmul(const float *l, const float *r, float *o, int b, int m, int n, int k, int l_col_mjr, int r_col_mjr);

/* Select / SelectPad, crates/compiler/src/tensor/operation/index/select.rs:12-26,93-110. */
void orc_select(const float *in, const int32_t *indices, float *out, int batch, int inner, int divisor);
void orc_select_pad(const float *in, const int32_t *indices, float *out, int batch, int inner, int divisor);

/* ReduceAcrossDimension::Sum over the batch axis of a [batch][inner] tensor,
 * crates/compiler/src/tensor/operation/linear/reduce.rs:96-120 (sequential sum from 0). */
void orc_reduce_sum_batch(const float *in, float *out, int batch, int inner);

/* activations: forward lowering crates/compiler/src/model/operations/pointwise.rs:54-139,
 * derivative-from-output crates/compiler/src/tensor/operation/autograd/dfo.rs:47-119 */
void orc_act_fwd(int act, const float *x, float *y, size_t n);
/* gin = gout * f'(out) exactly as dfo.rs composes it */
void orc_act_bwd(int act, const float *out, const float *gout, float *gin, size_t n);

/* Sigmoid forward 1/(1+exp(-x)), pointwise.rs:156-164; backward o*(1-o), dfo.rs:114-118 */
float orc_sigmoid(float x);

/* AdamW element update, crates/trainer/src/optimiser/adam.rs:35-52,121-131 */
void orc_adamw(float *p, float *m, float *v, const float *g, size_t n, float adj, float lr, float decay, float beta1,
               float beta2, float wmin, float wmax);

/* RAdam (crates/trainer/src/optimiser/radam.rs:35-52 kernel, :283-302 host-side step size): `step` is the 1-based
 * update count. step_size / denom are computed in f32 exactly as the reference does on the host. */
void orc_radam_step(int step, float beta1, float beta2, float n_sma_threshold, float *step_size, int *denom);
void orc_radam(float *p, float *m, float *v, const float *g, size_t n, float adj, float lr, float step_size, int denom,
               float decay, float beta1, float beta2, float wmin, float wmax);
/* Ranger = RAdam + lookahead every k steps: w = s = alpha*w + (1-alpha)*s (optimiser/ranger.rs:23-44,78-96) */
void orc_lookahead(float *w, float *slow, size_t n, float alpha);

/* Loss functions a ValueTrainerBuilder::loss_fn closure builds. Returns sum_b ew_b * loss_b (ew may be NULL) and
 * writes grad_y [batch*width] = d(sum)/dy (may be NULL).
 *   SIGMOID_MSE   (sigmoid(y) - t)^2                    model/builder.rs:351-354
 *   SIGMOID_POWER |sigmoid(y) - t|^power                model/builder.rs:345-358, AbsPower pointwise.rs:248-269,
 *                 Power::backward binary.rs:167-172, Abs' = sgn (unary.rs:146; 0 at 0 as gpu write.rs:172-174)
 *   SOFTMAX_CE    -sum_k t_k log(softmax(y)_k)          autograd/softmax.rs:27-37, softmax tensor/builder.rs:159-170 */
enum { ORC_LOSS_SIGMOID_MSE = 0, ORC_LOSS_SIGMOID_POWER = 1, ORC_LOSS_SOFTMAX_CE = 2 };
float orc_loss(int kind, int batch, int width, const float *y, const float *targets, const float *entry_weights,
               float power, float *grad_y);

/* WeightClipping / WeightDecay kernels, optimiser/clip.rs:18-31, decay.rs:18-40 */
void orc_weight_clip(float *w, size_t n, float min_w, float max_w);
void orc_clip_pass_through(const float *x, float *y, size_t n, float min_v, float max_v);
void orc_faux_quantise(const float *x, float *y, size_t n, float value, int round);
void orc_weight_decay(float *w, size_t n, float decay, float lr);

/* ------------------------------------------------------------------ net */

#define ORC_MAX_DENSE 4

/* A value net of the family the examples/ programs build with ValueTrainerBuilder:
 * shared feature transformer over `perspectives` sparse inputs -> activation
 * (optionally pairwise product of the two CReLU'd halves) -> concat ->
 * n_dense affine layers, each followed by output-bucket select (if bucketed)
 * and an activation -> sigmoid -> squared error (-> * entry weight) -> batch sum. */
typedef struct {
    int num_inputs;   /* F: feature count                        */
    int nnz;          /* N: max active features per perspective  */
    int ft_out;       /* H: feature transformer rows             */
    int ft_act;       /* ORC_ACT_*                               */
    int pairwise;     /* 1: out = act(x[0:H/2]) * act(x[H/2:H])  */
    int perspectives; /* 1 or 2                                  */
    int num_buckets;  /* output buckets (1 = no select)          */
    int n_dense;
    int dense_out[ORC_MAX_DENSE];      /* outputs per bucket of layer k   */
    int dense_act[ORC_MAX_DENSE];      /* activation after layer k        */
    int dense_bucketed[ORC_MAX_DENSE]; /* 1: rows = out*num_buckets + select */
    int use_entry_weights;             /* value/builder.rs:137-140 */
    /* --- extensions used by examples/tests/test2.rs and examples/progression/3_input_buckets.rs --- */
    int ft_factoriser;                 /* NB > 0: W_eff = l0w + l0f.repeat(NB), l0f is [ft_out x num_inputs/NB]
                                          (3_input_buckets.rs:62-67; Broadcast(Cols), operations/linear.rs:37-55) */
    int dense_dual[ORC_MAX_DENSE];     /* 1: x -> concat(x, x*x) before the activation (test2.rs:58-59)        */
    int dense_skip[ORC_MAX_DENSE];     /* s: the last s outputs of layer k bypass the rest of the net and are added
                                          to the final output (slice_rows + `out + skip_neuron`, test2.rs:54-64) */
    int loss_kind;                     /* ORC_LOSS_*; softmax CE: dense_out[last] classes, targets [batch][classes] */
    float loss_power;                  /* exponent of ORC_LOSS_SIGMOID_POWER */
} orc_net_desc;

/* number of weight tensors: l0w,l0b,l1w,l1b,... and, last, l0f when ft_factoriser > 0 */
int orc_net_num_tensors(const orc_net_desc *d);
/* element count of tensor t (order l0w,l0b,l1w,l1b,...[,l0f]) */
size_t orc_net_tensor_size(const orc_net_desc *d, int t);
/* input width of dense layer k: the concat of the perspectives for k = 0, else
 * (dense_out[k-1] - dense_skip[k-1]) * (dense_dual[k-1] ? 2 : 1) */
int orc_net_dense_in(const orc_net_desc *d, int k);

/* One fwd+bwd pass (Appendix A of SURVEY.md). params/grads are arrays of
 * orc_net_num_tensors() pointers. Returns the summed loss L (run.rs:212-221 divides by B).
 * net_out (may be NULL) receives the pre-sigmoid output [batch]. */
float orc_net_forward_backward(const orc_net_desc *d, int batch, const float *const *params, float *const *grads,
                               const int32_t *stm, const int32_t *nstm, const int32_t *buckets, const float *targets,
                               const float *entry_weights, float *net_out);

/* forward only; returns loss sum */
float orc_net_forward(const orc_net_desc *d, int batch, const float *const *params, const int32_t *stm,
                      const int32_t *nstm, const int32_t *buckets, const float *targets, const float *entry_weights,
                      float *net_out);

/* ------------------------------------------------------------------ host input path */

/* bulletformat::ChessBoard, 32 bytes; layout pinned by examples/tests/test4.rs:8-18 */
typedef struct {
    uint64_t occ;
    uint8_t pcs[16];
    int16_t score;
    uint8_t result;
    uint8_t ksq;
    uint8_t opp_ksq;
    uint8_t extra[3];
} orc_chess_board;

/* bulletformat::AtaxxBoard, 32 bytes; layout pinned by examples/ataxx.rs:8-18 */
typedef struct {
    uint64_t bbs[3];
    int16_t score;
    uint8_t result;
    uint8_t origstm;
    uint16_t fullm;
    uint8_t halfm;
    uint8_t extra;
} orc_ataxx_board;

enum {
    ORC_INPUT_CHESS768 = 0,      /* chess768.rs:20-30 */
    ORC_INPUT_CHESS_BUCKETS = 1, /* chess_buckets.rs:28-33 */
    ORC_INPUT_CHESS_BUCKETS_MIRRORED = 2, /* chess_buckets.rs:56-67,82-88 */
    ORC_INPUT_ATAXX147 = 3,      /* ataxx147.rs:18-27 */
    ORC_INPUT_ATAXX98 = 4,       /* ataxx147.rs:51-56 */
};

typedef struct {
    int kind;
    int factorised;   /* factorised.rs:32-48 wrapper with Chess768 as factoriser (chess kinds only) */
    int buckets[64];  /* expanded 64-entry king-bucket table */
    int num_buckets;
} orc_input_desc;

/* ChessBucketsMirrored::new, chess_buckets.rs:56-67: expand 32-entry half-board layout */
void orc_input_mirrored_expand(const int *layout32, orc_input_desc *d);
/* ChessBuckets::new, chess_buckets.rs:11-15 */
void orc_input_buckets_set(const int *layout64, orc_input_desc *d);

int orc_input_num_inputs(const orc_input_desc *d);
int orc_input_max_active(const orc_input_desc *d);

/* SparseInputType::map_features for one 32-byte record; writes up to max_active
 * (stm, ntm) pairs; returns the count. */
int orc_map_features(const orc_input_desc *d, const void *record, int32_t *stm, int32_t *ntm);

/* MaterialCount<N>::bucket, crates/bullet_lib/src/game/outputs.rs:24-31 */
int orc_material_bucket(uint64_t occ, int n_buckets);

/* make_mapper closure, crates/bullet_lib/src/value.rs:77-117 for `count` records.
 * stm/ntm: [count][nnz] with -1 terminator (not pre-cleared, value.rs:86-89),
 * buckets [count], targets [count] (or [count*3] when wdl_output), weights [count].
 * n_out_buckets <= 1 -> bucket 0 (NoOutputBuckets, value/builder.rs:200-206). */
void orc_map_batch(const orc_input_desc *d, const void *records, int count, int n_out_buckets, float blend,
                   float eval_scale, int use_win_rate_model, int wdl_output, int32_t *stm, int32_t *ntm,
                   int32_t *buckets, float *targets, float *weights);

/* ------------------------------------------------------------------ weight init */

/* rand_xoshiro 0.8.1 Xoroshiro128Plus seeded with SeedableRng::seed_from_u64 (SplitMix64),
 * + rand_distr 0.6.0 StandardNormal (ziggurat, 256 layers) -> Normal<f32>.
 * These crates are NOT in /root/reference (crates.io deps, Cargo.lock pins); this is a
 * restatement of their published algorithms, pinned only through the e2e expected
 * losses of examples/tests/test{1,3,4}.rs.  Call site: crates/trainer/src/model/weights.rs:52-74,389-399. */
typedef struct {
    uint64_t s0, s1;
} orc_rng;
void orc_rng_seed(orc_rng *r, uint64_t seed);
uint64_t orc_rng_next_u64(orc_rng *r);
double orc_rng_standard_normal(orc_rng *r);
/* vec_f32(rng, n, mean, stdev, use_gaussian), weights.rs:389-399 */
void orc_rng_fill(orc_rng *r, float *out, size_t n, float mean, float stdev, int gaussian);

/* ModelWeights::new for an orc_net_desc net: consumes one RNG stream tensor by tensor in
 * creation order (l0w, l0b(zero), l1w, ...), Normal(0, sqrt(2/in)), biases zero
 * (crates/compiler/src/model/builder.rs:108-119). ft_effective_inputs > 0 overrides the l0
 * stdev as init_with_effective_input_size does (builder.rs:149-152). */
void orc_net_init(const orc_net_desc *d, uint64_t seed, int ft_effective_inputs, float *const *params);

/* ------------------------------------------------------------------ generic index / reduce ops
 * (index/broadcast.rs:66-80, index/pad.rs:76-102, index/slice.rs:74-91, linear/reduce.rs:96-120); 4-byte elements;
 * (outer, dimen, inner) of the dimension acted on; reduce kind 0 = Sum, 1 = Max, 2 = Min */
void orc_broadcast(const uint32_t *in, size_t outer, size_t inner, size_t repeats, uint32_t *out);
void orc_pad(const uint32_t *in, size_t outer, size_t dimen, size_t inner, size_t before, size_t after, uint32_t value,
             uint32_t *out);
void orc_slice(const uint32_t *in, size_t outer, size_t dimen, size_t inner, size_t start, size_t end, uint32_t *out);
void orc_reduce_f32(const float *in, size_t outer, size_t dimen, size_t inner, int kind, float *out);
void orc_reduce_i32(const int32_t *in, size_t outer, size_t dimen, size_t inner, int kind, int32_t *out);

#ifdef __cplusplus
}
#endif
/* bench.py's CPU baseline only: `steps` full training steps (forward + backward on `threads` contiguous shards of the
 * batch in parallel, shard gradients added in order, AdamW with the reference defaults); returns the last loss sum */
float orc_train_steps_mt(const orc_net_desc *d, int batch, float *const *params, float *const *mom, float *const *vel,
                         const int32_t *stm, const int32_t *nstm, const int32_t *buckets, const float *targets,
                         const float *weights, int threads, int steps, float lr);

#endif
