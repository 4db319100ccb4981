/*
 * bullet_amd.h -- C ABI of libbullet_amd.so: an MI355X (gfx950) native backend for the
 * training hot path of jw1912/bullet.  Plain C types only; every pointer named *_dev / w /
 * out / grad etc. is a DEVICE pointer unless the comment says host.  All entry points return
 * 0 on success and a non-zero code on failure; bt_last_error() then returns a thread-local
 * "<hipErrorName>: <string>" / "Mismatched sizes!"-style message, mirroring
 * crates/gpu/src/runtime/rocm.rs:295-311 and crates/gpu/src/function.rs:209-240.
 *
 * The reference's backend seam is the private trait `GpuBindings`
 * (crates/gpu/src/runtime/bindings.rs:66-174) -- that level receives kernel *source text*, so a
 * hand-written-kernel backend binds one level up, at `Function::execute`
 * (crates/gpu/src/function.rs:198-280) and `OptimiserState::update`
 * (crates/trainer/src/optimiser.rs:65-86).  Section 1 below replaces the GpuBindings memory /
 * stream primitives, section 2 the lowered operators that Function::execute replays, section 3
 * the whole lowered fwd+bwd+optimiser step for the value-net family of examples/ (one call per
 * batch, what run::train's loop body does at crates/trainer/src/run.rs:185-221), section 4 the
 * host batch preparation (crates/bullet_lib/src/value.rs:55-118).
 *
 * Citations are file:line relative to the reference checkout (jw1912/bullet @ 2026-06-30).
 */
#ifndef BULLET_AMD_H
#define BULLET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BT_OK 0
#define BT_ERR_INVALID 1   /* shape / argument check failed ("Mismatched sizes!") */
#define BT_ERR_HIP 2       /* a HIP runtime call failed                            */
#define BT_ERR_STATE 3     /* object used in the wrong state                       */
#define BT_ERR_IO 4        /* host file I/O                                        */

/* Fused activation ids (DiffableFromOutput family, crates/compiler/src/tensor/operation/autograd/dfo.rs:47-103) */
enum {
    BT_ACT_IDENTITY = 0,
    BT_ACT_RELU = 1,
    BT_ACT_CRELU = 2,
    BT_ACT_SCRELU = 3,
    BT_ACT_SQRRELU = 4,
};

int bt_version(void);
/* Thread-local message of the last failing call on this thread; "" if none. */
const char *bt_last_error(void);

/* ------------------------------------------------------------------------------------------
 * 1. Runtime primitives  (replaces GpuBindings, bindings.rs:76-125; ROCm impl rocm.rs:66-171)
 * ------------------------------------------------------------------------------------------ */

/* device_get / context_set, bindings.rs:78-86 (every reference wrapper call does hipSetDevice first,
 * crates/gpu/src/runtime.rs:84-86). */
int bt_device_count(int *count);
int bt_device_set(int ordinal);
/* device_props, bindings.rs:80: name + gcnArchName + wavefront size */
int bt_device_props(int ordinal, char *name, int name_len, char *arch, int arch_len, int *wave_size, int *num_cus);
/* context_sync, bindings.rs:88 */
int bt_device_sync(void);
/* context_malloc + context_memset(0) = Buffer::zeroed, bindings.rs:90-94, crates/gpu/src/buffer.rs:112-131 */
int bt_malloc(void **ptr, size_t bytes);
int bt_free(void *ptr);
/* pinned host staging (the reference uploads from pageable Vecs, run/dataloader.rs:72-116) */
int bt_host_alloc_pinned(void **ptr, size_t bytes);
int bt_host_free_pinned(void *ptr);
/* stream_create/destroy/sync, bindings.rs:100-104. A stream is a hipStream_t passed as void*;
 * NULL means the default stream, and any hipStream_t created elsewhere (e.g. by PyTorch) is accepted. */
int bt_stream_create(void **stream);
int bt_stream_destroy(void *stream);
int bt_stream_sync(void *stream);
/* Events order one stream behind another without blocking the host (no counterpart in GpuBindings: the reference
 * joins its copy and compute streams with stream_sync on the host, run.rs:194-221). An event is a hipEvent_t passed
 * as void* (timing disabled). */
int bt_event_create(void **event);
int bt_event_destroy(void *event);
int bt_event_record(void *event, void *stream);
int bt_stream_wait_event(void *stream, void *event);
int bt_event_sync(void *event); /* the host waits until the recorded work is done */
int bt_event_query(void *event, int *done); /* *done = 1 if it is (never blocks) */
/* stream_memset / stream_memcpy_h2d / stream_memcpy_d2h, bindings.rs:110-125 */
int bt_memset_async(void *ptr, int byte_value, size_t bytes, void *stream);
int bt_memcpy_h2d_async(void *dst, const void *src_host, size_t bytes, void *stream);
int bt_memcpy_d2h_async(void *dst_host, const void *src, size_t bytes, void *stream);
int bt_memcpy_d2d_async(void *dst, const void *src, size_t bytes, void *stream);

/* Per-kernel timing with HIP events recorded on the stream each kernel is launched on (the reference
 * only has wall-clock pos/sec, crates/trainer/src/run/logger.rs:38-77). bt_profile_start arms the
 * calling thread's profiler for up to max_launches kernel launches; bt_profile_stop waits for the
 * events and fills counts[id] / total_ms[id] for id < bt_profile_num_kernels(). */
int bt_profile_start(int max_launches);
/* restrict timing to the kernels whose id bit is set (ids < 64; after bt_profile_start); default all */
int bt_profile_set_filter(unsigned long long kernel_mask);
/* restrict timing to kernels whose name contains `name_part` (after bt_profile_start; calls add up). Kernels are
 * reported under their own names with template arguments, as rocprofv3 --kernel-trace prints them (e.g.
 * "ft_bwd_oct<false, true, false, true>"); an id is assigned at a kernel's first launch. */
int bt_profile_only(const char *name_part);
/* time only every n-th launch of each selected kernel (after bt_profile_start; default 1): a timed launch carries
 * profiling timestamps on its dispatch packet, which costs the stream ~10 us, so a caller that wants an average
 * over a run without slowing it samples */
int bt_profile_set_sampling(int every);
int bt_profile_stop(int *counts, double *total_ms, int n_ids);
int bt_profile_num_kernels(void);
const char *bt_profile_kernel_name(int id);

/* ------------------------------------------------------------------------------------------
 * 2. Operators  (what Function::execute replays, function.rs:198-280; kernel inventory SURVEY 2.1)
 * ------------------------------------------------------------------------------------------ */

/* Sparse feature transformer = SparseMatmul (+ bias broadcast + activation [+ pairwise product]),
 * crates/compiler/src/tensor/operation/linear/sparse.rs:7-15,86-119 with the device stop rule of
 * crates/gpu/src/pointwise/write.rs:229-233 (stop at the first index < 0 or >= cols), fused with
 * Affine::forward's bias add (crates/compiler/src/model/builder.rs:145-147) and the activation
 * peepholes (foldrules.rs:173-231).  Several sparse inputs ("perspectives", stm/nstm) may share
 * one weight matrix, as l0.forward(stm) / l0.forward(ntm) do in every example; their outputs are
 * written side by side (the `concat` of model/operations/index.rs:74-81 is never materialised). */
typedef struct {
    int batch;    /* B                                                             */
    int rows;     /* output rows of the op (H)                                     */
    int cols;     /* F: number of features; valid indices are 0 <= j < cols        */
    int stride;   /* column stride of W in floats (>= offset + rows)               */
    int offset;   /* row offset inside each column                                 */
    int nnz;      /* N: index slots per position                                   */
    int act;      /* BT_ACT_*                                                      */
    int pairwise; /* 1: out[r] = act(x[r]) * act(x[r + rows/2]), r < rows/2 (builder.rs:390-393) */
    int n_inputs; /* 1 or 2 sparse inputs sharing W                                */
    int out_ld;   /* floats per position in `out`; input k occupies columns [k*out_w, (k+1)*out_w),
                     out_w = pairwise ? rows/2 : rows                              */
    int segment;  /* 0, or the width of a column segment that holds ALL active features of any one
                     (position, input): 768 for the king-bucketed input sets, whose indices are
                     768*bucket[ksq] + piece-square (chess_buckets.rs:28-33,82-88). Tables too large for the
                     LDS-resident kernels are then processed segment by segment through them
                     (bt_ft_forward_prebuilt / bt_ft_backward*). It is a promise about the data: an index
                     outside its entry's segment sets a device flag (bt_ft_segment_violations) and turns that
                     entry's forward output into NaN instead of being summed. Must divide cols; the index lists
                     must then be 16-byte aligned with nnz a multiple of 4 and <= 32 (else: any-shape kernels). */
} bt_ft_desc;

/* w: F32[stride*cols] column-major (docs/4-saved-networks.md:21-29); bias: F32[rows] or NULL;
 * idx[k]: I32[batch*nnz]; out: F32[batch*out_ld];
 * saved: pairwise only, F32[n_inputs*batch*rows] receiving act(x) before the product (needed by
 * the backward); NULL otherwise (the backward differentiates from `out`). */
int bt_ft_forward(void *stream, const bt_ft_desc *desc, const float *w, const float *bias,
                  const int32_t *const *idx, float *out, float *saved);

/* Scratch size for bt_ft_backward (index lists, partial sums). */
size_t bt_ft_backward_workspace_bytes(const bt_ft_desc *desc);

/* SparseMatmulBwdMulti (sparse.rs:198-271; merge rule rewriterules.rs:267-301) fused with the
 * activation derivative from the output (dfo.rs:30-40) and the bias-gradient batch sum
 * (linear/reduce.rs:96-120).  Replaces the float-atomic SpMMT template (write.rs:294-307): each
 * dw element is a sorted-by-feature segmented sum in a fixed order, bit-reproducible run to run.
 * grad_out: F32[batch*out_ld] gradient w.r.t. `out`; out/saved as produced by bt_ft_forward.
 * dw: F32[stride*cols], fully overwritten (no pre-zeroing needed); dbias: F32[rows] or NULL. */
int bt_ft_backward(void *stream, const bt_ft_desc *desc, const int32_t *const *idx, const float *grad_out,
                   const float *out, const float *saved, float *dw, float *dbias, void *workspace,
                   size_t workspace_bytes);

/* The two halves of bt_ft_backward, for callers that overlap them: the sorted index depends only on
 * `idx`, so it can be built on another stream while the forward pass runs (the reference keeps two
 * streams too, run.rs:156-157). bt_ft_backward == bt_ft_build_index + bt_ft_backward_prebuilt. */
int bt_ft_build_index(void *stream, const bt_ft_desc *desc, const int32_t *const *idx, void *workspace,
                      size_t workspace_bytes);
int bt_ft_backward_prebuilt(void *stream, const bt_ft_desc *desc, const float *grad_out, const float *out,
                            const float *saved, float *dw, float *dbias, void *workspace, size_t workspace_bytes);
/* bt_ft_forward for descriptors with `segment` set: runs on the entry lists bt_ft_build_index left in the
 * workspace (same stream order: build first). Without `segment` it is bt_ft_forward. */
int bt_ft_forward_prebuilt(void *stream, const bt_ft_desc *desc, const float *w, const float *bias,
                           const int32_t *const *idx, float *out, float *saved, void *workspace, size_t workspace_bytes);
/* synchronises the stream and returns how many indices bt_ft_build_index / the segment kernels found outside
 * their entry's segment since the index was built (0 = the promise held) */
int bt_ft_segment_violations(void *stream, const bt_ft_desc *desc, void *workspace, size_t workspace_bytes, int *count);

/* Dense affine layer with optional output-bucket select and activation:
 * Matmul (linear/matmul.rs:62-93, lowered model/operations/linear.rs:128-149, device
 * function.rs:256-274 -> hipblasSgemm) + bias broadcast + Select (index/select.rs:12-26) + activation.
 * w: F32[(out*buckets) x in] column-major; bias F32[out*buckets]; x: F32[batch*in];
 * bucket: I32[batch] (ignored when buckets == 1); y: F32[batch*out] (activated). */
typedef struct {
    int batch;
    int in;      /* input width                     */
    int out;     /* outputs per bucket              */
    int buckets; /* 1 = no select                   */
    int act;     /* BT_ACT_* applied after select   */
} bt_dense_desc;

int bt_dense_forward(void *stream, const bt_dense_desc *desc, const float *w, const float *bias, const float *x,
                     const int32_t *bucket, float *y);
size_t bt_dense_backward_workspace_bytes(const bt_dense_desc *desc);
/* Matmul::backward (matmul.rs:99-126) + SelectPad (select.rs:93-110) + bias batch-sum + activation
 * derivative from output. grad_y: F32[batch*out] w.r.t. the activated y. Writes dw (full, rows of
 * unselected buckets get exact zeros), dbias, and grad_x F32[batch*in] (may be NULL). */
int bt_dense_backward(void *stream, const bt_dense_desc *desc, const float *w, const float *x,
                      const int32_t *bucket, const float *y, const float *grad_y, float *dw, float *dbias,
                      float *grad_x, void *workspace, size_t workspace_bytes);

/* bt_dense_backward without a workspace argument and without any allocation. It has NO counterpart in the reference
 * (whose Function::prealloc owns all intermediates, function.rs:180-196); it is for callers that cannot call
 * hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid allocation. The partial sums
 * live in a file-scope device array of BT_STATIC_SCRATCH_BYTES inside the library (each GPU holds its own copy until
 * the process ends); a shape whose bt_dense_backward_workspace_bytes() exceeds that is rejected with BT_ERR_INVALID.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams or threads) race. Results
 * are bit-identical to bt_dense_backward. */
#define BT_STATIC_SCRATCH_BYTES (16u << 20)
int bt_dense_backward_static(void *stream, const bt_dense_desc *desc, const float *w, const float *x,
                             const int32_t *bucket, const float *y, const float *grad_y, float *dw, float *dbias,
                             float *grad_x);

/* sigmoid + squared error (+ entry weight) + batch sum and its gradient in one pass:
 * model/operations/pointwise.rs:141-165, model/builder.rs:351-354, value/builder.rs:137-142,
 * derivative rules binary.rs:101-123 and dfo.rs:105-119.
 * y: F32[batch] raw net output; loss_sum: F32[1] (overwritten, un-normalised: run.rs:221 divides by B);
 * grad_y: F32[batch] d(loss_sum)/dy; entry_weights may be NULL. */
int bt_sigmoid_mse(void *stream, int batch, const float *y, const float *targets, const float *entry_weights,
                   float *loss_sum, float *grad_y);

/* The other loss functions a ValueTrainerBuilder::loss_fn closure can build, same contract as
 * bt_sigmoid_mse with `width` outputs / targets per position (y, targets, grad_y: F32[batch*width]):
 *   BT_LOSS_SIGMOID_MSE    sigmoid(y).squared_error(t)                         (width 1)
 *   BT_LOSS_SIGMOID_POWER  sigmoid(y).power_error(t, power) = |s - t|^power    (width 1; model/builder.rs:345-358,
 *                          AbsPower lowering pointwise.rs:248-269, Power::backward binary.rs:167-172, Abs' = sgn with
 *                          sgn(0) = 0 as the device code writes it, gpu/src/pointwise/write.rs:172-174)
 *   BT_LOSS_SOFTMAX_CE     y.softmax_crossentropy_loss(t) = -t*log(softmax(y)), summed over the `width` classes
 *                          (wdl_output: width 3; autograd/softmax.rs:27-37, softmax = tensor/builder.rs:159-170:
 *                          exp(y - max) * (1 / sum), gradient (softmax - t) * grad)
 * loss_sum[0] = sum over positions (and classes) of entry_weight * loss. */
enum { BT_LOSS_SIGMOID_MSE = 0, BT_LOSS_SIGMOID_POWER = 1, BT_LOSS_SOFTMAX_CE = 2 };
int bt_loss(void *stream, int kind, int batch, int width, const float *y, const float *targets,
            const float *entry_weights, float power, float *loss_sum, float *grad_y);

/* AdamW, crates/trainer/src/optimiser/adam.rs:29-62,98-235; one launch for all weight tensors
 * (the reference launches once per tensor, optimiser.rs:175-187). adj_dev / lr_dev are 1-element
 * device buffers exactly as in run.rs:130-133. */
typedef struct {
    const float *grad;
    float *param, *momentum, *velocity;
    size_t n;
    float decay, beta1, beta2, min_weight, max_weight; /* AdamWParams, adam.rs:20-33 */
} bt_adamw_tensor;
#define BT_ADAMW_MAX_TENSORS 16
int bt_adamw(void *stream, int n_tensors, const bt_adamw_tensor *tensors /* host */, const float *adj_dev,
             const float *lr_dev);
/* RAdam::update (crates/trainer/src/optimiser/radam.rs:35-52,264-313): same tensor table (decay, betas, clip; the
 * reference defaults are decay 0, no clip = +-FLT_MAX). `step` is the 1-based update count; the rectification
 * (step size, whether to divide by sqrt(v)) is computed on the host in f32 as the reference does. */
int bt_radam(void *stream, int n_tensors, const bt_adamw_tensor *tensors /* host */, const float *adj_dev,
             const float *lr_dev, int step, float n_sma_threshold);
/* the lookahead half of Ranger (optimiser/ranger.rs:23-44): param = slow = alpha*param + (1-alpha)*slow */
int bt_lookahead(void *stream, int n_tensors, float *const *params, float *const *slow, const size_t *counts, float alpha);
/* the kernels of the WeightClipping / WeightDecay optimiser wrappers (optimiser/clip.rs:18-31, decay.rs:18-40), for
 * callers that compose them around an update themselves (Placement::Before / After = call order):
 * clip: w = max(min(w, max), min); decay: w = w * (1 + -1 * (lr * decay)), lr read from its 1-element device buffer */
int bt_weight_clip(void *stream, float *w, size_t n, float min_w, float max_w);
int bt_weight_decay(void *stream, float *w, size_t n, float decay, const float *lr_dev);
/* ModelNode::clip_pass_through_grad(min, max) (compiler/src/model/builder.rs:365-368, model/operations/pointwise.rs:
 * 272-300): y = min(max(x, min), max). Its backward hands the incoming gradient on unchanged
 * (autograd/passthrough.rs:31-33): the caller uses grad_y as grad_x, there is no kernel. y may be x. */
int bt_clip_pass_through(void *stream, const float *x, float *y, size_t n, float min_v, float max_v);
/* ModelNode::faux_quantise(value, round) (builder.rs:370-373, pointwise.rs:196-221): y = round_or_trunc(value * x) *
 * (1 / value), roundf / truncf as in the generated code (gpu/src/pointwise/write.rs:196-197). Forward only, as in the
 * reference: the graph it lowers to has no gradient (Unary::Round / Truncate: `unimplemented!()`,
 * tensor/operation/pointwise/unary.rs:160-162), so it cannot sit on a trained path there either. y may be x. */
int bt_faux_quantise(void *stream, const float *x, float *y, size_t n, float value, int round);

/* ------------------------------------------------------------------------------------------
 * 3. The lowered training step for the ValueTrainerBuilder net family
 *    (crates/bullet_lib/src/value/builder.rs:118-181; step = run.rs:185-221)
 * ------------------------------------------------------------------------------------------ */
#define BT_MAX_DENSE 4
typedef struct {
    int num_inputs;   /* F */
    int nnz;          /* N */
    int ft_out;       /* H */
    int ft_act;       /* BT_ACT_* */
    int pairwise;     /* examples/progression/4_multi_layer.rs:79-81 */
    int perspectives; /* 1 = single_perspective, 2 = dual_perspective (value/builder.rs:217-245) */
    int num_buckets;  /* OutputBuckets::BUCKETS (game/outputs.rs:3-7); 1 = none */
    int n_dense;
    int dense_out[BT_MAX_DENSE];
    int dense_act[BT_MAX_DENSE];
    int dense_bucketed[BT_MAX_DENSE];
    int use_entry_weights;
    /* extensions (zero = off) used by examples/tests/test2.rs and examples/progression/3_input_buckets.rs */
    int ft_factoriser;            /* NB > 0: the FT runs on l0w + l0f.repeat(NB), l0f = [ft_out x num_inputs/NB] is one
                                     more weight tensor, stored LAST (3_input_buckets.rs:62-67; Broadcast(Cols),
                                     crates/compiler/src/model/operations/linear.rs:37-55) */
    int dense_dual[BT_MAX_DENSE]; /* 1: layer output x -> concat(x, x*x) before its activation (test2.rs:58) */
    int dense_skip[BT_MAX_DENSE]; /* s: the last s outputs of layer k are sliced off and added to the net output
                                     (slice_rows + `out + skip_neuron`, test2.rs:54-64); next layer sees
                                     (dense_out - s) * (dual ? 2 : 1) inputs */
    int ft_segment;               /* bt_ft_desc.segment for the feature transformer (768 for ChessBuckets /
                                     ChessBucketsMirrored inputs without the Factorised wrapper; 0 = none) */
    int loss_kind;                /* BT_LOSS_* (0 = sigmoid MSE). BT_LOSS_SOFTMAX_CE: the last layer has dense_out = number
                                     of classes (3 for wdl_output, value/builder.rs:133) and targets are
                                     F32[batch * classes] */
    float loss_power;             /* BT_LOSS_SIGMOID_POWER exponent */
} bt_net_desc;

typedef struct bt_net bt_net; /* opaque: weights, AdamW state, gradients, activations, scratch */

/* Allocates every device buffer for `max_batch` positions on the current device.
 * ext_params / ext_grads: optional caller-owned flat device buffers of bt_net_param_count()
 * floats (so a harness can all-reduce the flat gradient); NULL -> library-owned. */
size_t bt_net_param_count(const bt_net_desc *desc);
int bt_net_num_tensors(const bt_net_desc *desc);
size_t bt_net_tensor_size(const bt_net_desc *desc, int tensor);
/* offset (in floats) of a tensor inside the flat buffers; tensors are padded to 4 floats */
size_t bt_net_tensor_offset(const bt_net_desc *desc, int tensor);
int bt_net_create(const bt_net_desc *desc, int max_batch, float *ext_params, float *ext_grads, bt_net **net);
int bt_net_destroy(bt_net *net);
/* flat views (tensor order l0w,l0b,l1w,l1b,...: creation order, weights.rs:57) */
float *bt_net_params(bt_net *net);
float *bt_net_grads(bt_net *net);
float *bt_net_momentum(bt_net *net);
float *bt_net_velocity(bt_net *net);
/* floats in each flat buffer (= bt_net_param_count of its descriptor); the max_batch it was created for; whether
 * the gradient buffer is library-owned (it then has a spare float behind it, see bt_train_config.dp_loss_slot) */
size_t bt_net_flat_size(const bt_net *net);
int bt_net_max_batch(const bt_net *net);
int bt_net_grads_owned(const bt_net *net);
/* per-tensor AdamW overrides = Optimiser::set_params_for_weight, optimiser.rs:204-206 */
int bt_net_set_adamw(bt_net *net, int tensor, float decay, float beta1, float beta2, float min_w, float max_w);

/* Device batch tensors, layouts of crates/bullet_lib/src/value.rs:70-75. nstm may be NULL for
 * single-perspective nets, buckets NULL when num_buckets == 1, entry_weights NULL when unused. */
typedef struct {
    int batch;
    const int32_t *stm, *nstm, *buckets;
    const float *targets, *entry_weights;
} bt_batch;

/* fwd + bwd: fills the flat gradient (unscaled batch sums, run.rs:130) and loss_sum_dev[0]. */
int bt_net_forward_backward(bt_net *net, void *stream, const bt_batch *batch, float *loss_sum_dev);
/* Optional pipelining hint, given BEFORE bt_net_forward_backward(current): `next` is the batch of the following step
 * (same net, its tensors complete in `stream` order by now, or once `ready_event` -- a hipEvent_t, may be NULL -- has
 * fired). Multi-layer / king-bucketed nets then build next's index structures (entry lists, bucket order, sorted
 * lists: they depend on the indices only) on their side stream while the current step runs, in a second buffer
 * set, and the following bt_net_forward_backward(next) starts with them in place. The single-layer bucketed net does
 * the same (its forward lists are decoded there too) and orders the side stream behind the completion event that rode
 * on the previous step's last kernel, so no step records a fork event on `stream` (~5 us each). Results are
 * identical with or without the hint; a hint that does not match the following call is ignored. No counterpart in the
 * reference. */
int bt_net_hint_next_batch(bt_net *net, const bt_batch *next, void *ready_event);
/* Steady-state steps as one hipGraph launch each (no counterpart in the reference). With `on`, a
 * bt_net_train_step / bt_net_forward_backward whose batch was announced by bt_net_hint_next_batch during the step
 * before, and which itself has a next batch announced, is captured the first time its combination of batch tensors,
 * next-batch tensors, buffers and stream is seen (single-layer fused nets on a caller's stream; anything else, and any
 * capture failure, runs the ordinary way) and replayed afterwards: one driver call instead of about ten. Off by default,
 * and measured slower on ROCm 7.2 -- one hipGraphLaunch of this graph costs the host more than the ten calls it
 * replaces; kept for stacks where that changes. Results are bit-identical. The batch's ready event and `done_event` keep
 * their meaning. While the calling thread's profiler is armed (bt_profile_start) steps run the ordinary way: kernels
 * inside a graph carry no timestamps. bt_net_graph_launches: how many steps have run as graphs. */
int bt_net_use_graphs(bt_net *net, int on);
int bt_net_graph_launches(const bt_net *net);
/* AdamW on every tensor with g = adj * grad (adj = 1/global_batch), adam.rs:121-131 */
int bt_net_optimiser_step(bt_net *net, void *stream, const float *adj_dev, const float *lr_dev);
/* the same, and `done_event` (a hipEvent_t, may be NULL) fires when the update is complete. It travels on the update
 * kernel's own dispatch packet, which is cheaper for the stream than a bt_event_record behind the call. */
int bt_net_optimiser_step_event(bt_net *net, void *stream, const float *adj_dev, const float *lr_dev, void *done_event);
/* One training step of a single device = run.rs:185-192: bt_net_forward_backward + bt_net_optimiser_step_event in one
 * call, bit-identical to the two calls. For the single-layer nets whose backward pass ends in the feature transformer's
 * gradient merge (AdamW, no factoriser) the update runs inside that merge kernel, element by element as each gradient
 * sum completes -- one kernel and one pass over the parameters fewer. The flat gradient is still filled. */
int bt_net_train_step(bt_net *net, void *stream, const bt_batch *batch, float *loss_sum_dev, const float *adj_dev,
                      const float *lr_dev, void *done_event);
/* Which OptimiserState the step uses: AdamW (default, adam.rs), RAdam (radam.rs) or Ranger = RAdam + lookahead
 * (ranger.rs; defaults beta1 0.99, alpha 0.5, k 6; the slow weights start at ZERO as in the reference, ranger.rs:71).
 * Resets the step counter and the optimiser state (OptimiserState::reset). Per-tensor decay / betas / clip stay
 * with bt_net_set_adamw. */
enum { BT_OPT_ADAMW = 0, BT_OPT_RADAM = 1, BT_OPT_RANGER = 2 };
int bt_net_set_optimiser(bt_net *net, int kind, float n_sma_threshold, float alpha, int k);
/* forward only (ModelEvaluator, crates/trainer/src/model/evaluator.rs:50-81): raw outputs F32[batch] */
int bt_net_forward(bt_net *net, void *stream, const bt_batch *batch, float *out_dev, float *loss_sum_dev);

/* ------------------------------------------------------------------------------------------
 * 4. Host batch preparation (crates/bullet_lib/src/value.rs:55-118, game/inputs/, game/outputs.rs)
 * ------------------------------------------------------------------------------------------ */
enum {
    BT_INPUT_CHESS768 = 0,
    BT_INPUT_CHESS_BUCKETS = 1,
    BT_INPUT_CHESS_BUCKETS_MIRRORED = 2,
    BT_INPUT_ATAXX147 = 3,
    BT_INPUT_ATAXX98 = 4,
};
typedef struct {
    int kind;        /* BT_INPUT_* */
    int factorised;  /* game/inputs/factorised.rs (Chess768 factoriser) */
    int buckets[64]; /* expanded king-bucket table (chess_buckets.rs:11-15,56-67) */
    int num_buckets;
} bt_input_desc;

int bt_input_init(bt_input_desc *d, int kind, int factorised, const int *layout, int layout_len);
int bt_input_num_inputs(const bt_input_desc *d);
int bt_input_max_active(const bt_input_desc *d);

typedef struct {
    int out_buckets;        /* MaterialCount<N>; <=1 -> bucket 0 */
    float blend;            /* WDL lambda for this step (WdlScheduler::blend) */
    float eval_scale;       /* TrainingSchedule::eval_scale */
    int use_win_rate_model; /* value.rs:104-108 */
    int wdl_output;         /* value.rs:96-101: 3-way one-hot target */
    int threads;            /* ModelInputsMapper::map fan-out, model/inputs.rs:86-109 */
    /* Per-position closures of the builder (value/builder.rs:102-111), host mapper only, called on the worker threads:
     * blend_fn = wdl_adjust_function(pos, scheduled blend) -> blend used for this position (value.rs:113; the result
     * must be in [0, 1]); weight_fn = datapoint_weight_function(pos) -> entry weight (value.rs:94; the net must then
     * be created with use_entry_weights = 1, value/builder.rs:137-140). NULL = identity / 1.0. */
    float (*blend_fn)(void *user, const void *record32, float blend);
    float (*weight_fn)(void *user, const void *record32);
    void *fn_user;
} bt_map_opts;

/* records: host, count*32 bytes of ChessBoard / AtaxxBoard. Outputs are HOST buffers (pinned
 * recommended): stm,nstm I32[count*nnz]; buckets I32[count]; targets F32[count or 3*count];
 * weights F32[count]. Slots after the -1 terminator are left untouched (value.rs:86-89). Corrupt records are
 * rejected, never mapped (see bt_map_batch_device_errors below). */
int bt_map_batch(const bt_input_desc *d, const bt_map_opts *o, const void *records, int count, int32_t *stm,
                 int32_t *nstm, int32_t *buckets, float *targets, float *weights);
/* SfBinpackLoader (value/loader/sfbinpack.rs:50-225): Stockfish binpack files -> every entry that passes `filter` as
 * a bulletformat position (convert_to_bulletformat, :23-48) -> the same shuffle buffer as bt_reader_open_viribinpack.
 * The filter sees what the reference's closure reads off a TrainingDataEntry (examples/simple.rs:96-102): score and
 * result (side-to-move relative, result -1 / 0 / 1), ply, the move to be played, the piece type standing on its
 * destination (7 = none) and whether the side to move is in check; non-zero keeps the entry, NULL keeps all. The
 * wire format belongs to the un-vendored crate sfbinpack 0.6.4 and is restated (parity unpinned, see
 * csrc/sfbinpack.cpp). */
typedef struct {
    const void *chessboard; /* the position as the 32-byte ChessBoard it becomes */
    int16_t score;
    uint16_t ply;
    int16_t result;
    uint8_t move_from, move_to;
    uint8_t move_type;      /* 0 normal, 1 promotion, 2 castling (move_to = the rook's square), 3 en passant */
    uint8_t promotion;      /* N B R Q = 0..3 when move_type = 1 */
    uint8_t captured_piece; /* type on the destination square: P N B R Q K = 0..5, 7 = none */
    uint8_t in_check;
} bt_sf_entry;
typedef int (*bt_sf_filter)(void *user, const bt_sf_entry *entry);

/* The same mapping on the device: records_dev (count*32 bytes, 8-byte aligned) and all outputs are DEVICE buffers;
 * one wave per record, enqueued on `stream` (o->threads is ignored). Indices, terminators and buckets are
 * bit-identical to bt_map_batch; score targets may differ by 1 ulp where the two exp implementations round
 * differently. Only 32 B per position cross PCIe instead of 268 B, and no host thread touches the features. */
/* `<fen> | <score> | <result>` -> one 32-byte ChessBoard record (host): `ChessBoard::from_str` of bulletformat 1.8.0,
 * used by InMemoryTextLoader (value/loader/text.rs:28) and ValueTrainer::eval ("{fen} | 0 | 0.0", value.rs:217).
 * score: centipawns (i16), result 1.0 / 0.5 / 0.0 (also "[1.0]", "1", "1/2", "0" ...), both white-relative in the
 * text and side-to-move-relative in the record. */
int bt_chessboard_from_str(const char *line, void *record32);
int bt_map_batch_device(void *stream, const bt_input_desc *d, const bt_map_opts *o, const void *records_dev, int count,
                        int32_t *stm, int32_t *nstm, int32_t *buckets, float *targets, float *weights);
/* A record the reference's mapper would panic on (more than 32 pieces, piece types 6 / 7, king squares > 63, result
 * > 2 with WDL targets, fewer than two pieces with output buckets: value.rs:80,91,101, chess_buckets.rs:30,
 * outputs.rs:29) makes bt_map_batch fail with BT_ERR_INVALID ("bt_map_batch: record <i>: <reason>"). One deliberate,
 * stricter divergence: a result > 2 is rejected with scalar targets as well, where the reference would go on
 * training with the target blend * result / 2 > 1 (value.rs:112); it only shows on corrupt data. The device mapper cannot fail a call
 * it has only enqueued: it maps such a record to an empty list with zero weight and counts it; this call waits for
 * `stream`, returns the count since the last call on the current device and clears it. bt_train_run checks it after
 * every superbatch and fails with BT_ERR_INVALID when it is non-zero. */
int bt_map_batch_device_errors(void *stream, int *count);

/* ------------------------------------------------------------------------------------------
 * 5. Weight init, schedules, data reader and the superbatch training loop
 *    (crates/trainer/src/model/weights.rs:52-74, crates/bullet_lib/src/trainer/schedule/{lr,wdl}.rs,
 *     crates/trainer/src/reader/fixed_size.rs, crates/trainer/src/reader.rs, crates/trainer/src/run.rs:75-265)
 * ------------------------------------------------------------------------------------------ */

/* ModelWeights::new(defn, seed) for a bt_net_desc net into a HOST flat buffer of bt_net_param_count()
 * floats: one Xoroshiro128+ stream (seed_from_u64) consumed tensor by tensor in creation order, weights
 * Normal(0, sqrt(2/inputs)), biases zero (crates/compiler/src/model/builder.rs:108-119).
 * ft_effective_inputs > 0 = Affine::init_with_effective_input_size for l0 (builder.rs:149-152). */
int bt_net_init_weights_host(const bt_net_desc *desc, uint64_t seed, int ft_effective_inputs, float *flat_host);

/* LrScheduler (schedule/lr.rs): value of the learning rate for (batch, superbatch) */
enum { BT_LR_CONSTANT = 0, BT_LR_DROP = 1, BT_LR_STEP = 2, BT_LR_LINEAR = 3, BT_LR_COSINE = 4, BT_LR_EXP = 5 };
typedef struct {
    int kind;           /* BT_LR_*                                                                    */
    float a;            /* ConstantLR.value | DropLR/StepLR.start | Linear/Cosine/Exp .initial_lr     */
    float b;            /* DropLR/StepLR.gamma | Linear/Cosine/Exp .final_lr                           */
    int n;              /* DropLR.drop | StepLR.step | Linear/Cosine/Exp .final_superbatch             */
    int warmup_batches; /* > 0 wraps the scheduler in Warmup{inner, warmup_batches} (lr.rs:168-191)    */
} bt_lr_sched;
float bt_lr_at(const bt_lr_sched *s, int batch, int superbatch);

/* WdlScheduler (schedule/wdl.rs) */
enum { BT_WDL_CONSTANT = 0, BT_WDL_LINEAR = 1 };
typedef struct {
    int kind;
    float start, end;   /* ConstantWDL.value = start; LinearWDL.{start,end} */
    int warmup_batches;
} bt_wdl_sched;
float bt_wdl_at(const bt_wdl_sched *s, int batch, int superbatch, int final_superbatch);

/* FixedSizeDataReader / DirectSequentialDataLoader over 32-byte records: files are read in order and looped
 * forever; start_position (records) is taken modulo the epoch size and may start in a later file
 * (fixed_size.rs:42-107). bt_reader_next always delivers exactly `count` records (ReadMapLoader's
 * re-chunking with carry-over, reader.rs:31-81). */
typedef struct bt_reader bt_reader;
/* A single path of the form "text:<file>" selects InMemoryTextLoader instead (value/loader/text.rs:21-35): one
 * `<fen> | <score> | <result>` position per line, all parsed up front with bt_chessboard_from_str and then served
 * round and round; start_position is ignored there, as in the reference. */
int bt_reader_open(const char *const *paths, int n_paths, uint64_t start_position, bt_reader **reader);
/* dst_records == NULL skips `count` records (a seek for plain files): how a data-parallel rank passes over the other
 * ranks' shards of a global batch */
int bt_reader_next(bt_reader *reader, void *dst_records, int count);
/* ViriBinpackLoader (value/loader/viribinpack.rs:37-152): games in viriformat are replayed into bulletformat positions
 * on `threads` workers and pass through a shuffle buffer of buffer_size_mb * 2^20 / 32 / 2 positions (Fisher-Yates
 * over the xorshift of value/loader/rng.rs); files are read in order, round and round. filter = ViriFilter::Custom
 * (viribinpack.rs:188-206): called for every position before its move with the position as a 32-byte ChessBoard,
 * the move (viriformat encoding), its eval (white-relative centipawns), the game result for white (1.0 / 0.5 / 0.0)
 * and the ply inside the game; non-zero keeps the position; NULL keeps all. It runs on the worker threads.
 * shuffle_seed 0 = seeded from the clock like SimpleRand::with_seed. The wire format itself belongs to the
 * un-vendored crate viriformat 2.0.1 and is restated (parity unpinned, see csrc/viribinpack.cpp).
 * Use the reader through bt_train_config.reader. */
typedef int (*bt_viri_filter)(void *user, const void *chessboard32, uint16_t mv, int16_t eval, float wdl, int ply);
int bt_reader_open_viribinpack(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                               bt_viri_filter filter, void *user, uint64_t shuffle_seed, bt_reader **reader);
/* ViriFilter::Builtin (viribinpack.rs:17-27,178-187; `Filter::default()` in examples/simple.rs:84): the crate's own
 * `Filter`, restated from its published fields (parity unpinned like the format; its WDL-model filter, off by default,
 * is not restated). A position is DROPPED when: ply < min_ply (ply as in the FEN: 2 * (fullmove - 1) + black to move) |
 * |eval| >= max_eval | fewer than min_pieces pieces | filter_tactical and its move captures or promotes |
 * filter_check and the side to move is in check | filter_castling and its move castles | max_eval_incorrectness !=
 * UINT32_MAX and the eval points away from the game's result by more than it (drawn game: |eval|) |
 * random_fen_skipping and a uniform draw < random_fen_skip_probability. */
typedef struct {
    uint32_t min_ply, min_pieces, max_eval;
    int filter_tactical, filter_check, filter_castling;
    uint32_t max_eval_incorrectness;
    int random_fen_skipping;
    double random_fen_skip_probability;
} bt_viri_builtin_filter;
void bt_viri_filter_default(bt_viri_builtin_filter *filter);      /* Filter::default(): 16 / 4 / 20000, tactical + check on */
void bt_viri_filter_unrestricted(bt_viri_builtin_filter *filter); /* Filter::UNRESTRICTED: drops nothing */
int bt_reader_open_viribinpack_builtin(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                                       const bt_viri_builtin_filter *filter, uint64_t shuffle_seed, bt_reader **reader);
int bt_reader_open_sfbinpack(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                             bt_sf_filter filter, void *user, uint64_t shuffle_seed, bt_reader **reader);
/* MontyBinpackLoader (value/loader/montybinpack.rs:17-186): games in montyformat (`MontyValueFormat`) replayed into
 * bulletformat positions, the same pipeline and shuffle buffer as the viriformat loader. The filter is the reference's
 * `Fn(&Position, Move, i16, f32) -> bool` (montybinpack.rs:18,169): called for every position before its move with the
 * position as a 32-byte ChessBoard, the move (montyformat encoding: from << 10 | to << 4 | flag), its score
 * (white-relative) and the game result for white (1.0 / 0.5 / 0.0); non-zero keeps the position, NULL keeps all. The wire
 * format and the move maker belong to the un-vendored crate montyformat 0.9.2 and are restated (parity unpinned, see
 * csrc/montybinpack.cpp). */
typedef int (*bt_monty_filter)(void *user, const void *chessboard32, uint16_t mv, int16_t score, float result);
int bt_reader_open_montybinpack(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                                bt_monty_filter filter, void *user, uint64_t shuffle_seed, bt_reader **reader);
int bt_reader_close(bt_reader *reader);

/* run::train (run.rs:75-265) for one device: loader thread (reader -> bt_map_batch into a pinned ring) ->
 * double-buffered async H2D on a copy stream || fwd+bwd+AdamW on a compute stream; the loss of each batch is
 * read back asynchronously and delivered in order (the reference syncs every batch, run.rs:209-221). */
typedef struct {
    int batch_size, batches_per_superbatch, start_superbatch, end_superbatch; /* TrainingSteps, run/schedule.rs:5-11 */
    bt_lr_sched lr;
    bt_wdl_sched wdl;
    float eval_scale;        /* TrainingSchedule::eval_scale */
    int out_buckets;         /* MaterialCount<N> */
    int use_win_rate_model;
    int threads;             /* LocalSettings::threads */
    int log_rate;            /* progress line every log_rate batches (reference: 128); 0 = silent */
    int wdl_output;          /* ValueTrainerBuilder::wdl_output: 3-way one-hot targets (value.rs:96-101) for a net
                                with loss_kind = BT_LOSS_SOFTMAX_CE and 3 outputs */
    int device_mapper;       /* 1: the loader thread only reads records; they are uploaded raw and mapped by
                                bt_map_batch_device on the copy stream (`threads` is then unused) */
    /* Data-parallel training (no counterpart in the reference, which is single-device: SURVEY.md 8e). One process
     * per GPU calls bt_train_run with the same files and schedule; dp_world <= 1 = off. `batch_size` stays the GLOBAL
     * batch: every rank reads the same record stream, keeps positions [lo, hi) of each global batch (contiguous
     * shards, sizes differ by at most one, rank order), runs fwd+bwd on them, then calls `grad_sync` to SUM the
     * unscaled gradient and the loss over ranks, and steps the optimiser with adj = 1/batch_size, so the replicas
     * stay bit-identical. `net` needs max_batch >= the shard size. */
    int dp_rank, dp_world;
    /* Sums `n_floats` floats at `buf_dev` over all ranks, in place, ordered after the work already enqueued on
     * `stream` and before anything enqueued on it afterwards (e.g. an RCCL all-reduce). Returns 0 on success.
     * dp_loss_slot = 1 promises that the gradient buffer of `net` has room for one more float behind its
     * bt_net_param_count() floats (true for a library-owned buffer): the loss sum is then kept there and a step is ONE
     * call over n + 1 floats; otherwise the loss is summed by a second call over 1 float. */
    int (*grad_sync)(void *user, float *buf_dev, size_t n_floats, void *stream);
    void *grad_sync_user;
    int dp_loss_slot;
    /* > 0: leave the loop after this superbatch (start_superbatch <= stop_superbatch <= end_superbatch) while the
     * schedules keep seeing end_superbatch as the end of training -- how a caller checkpoints between superbatches
     * (value.rs:197-208): run to the checkpoint, save, call again with start_superbatch = stop_superbatch + 1. */
    int stop_superbatch;
    /* non-NULL: a bt_reader the caller opened (bt_reader_open_viribinpack ...) is used instead of `paths`; it is not
     * repositioned for start_superbatch (the binpack loaders ignore the start, viribinpack.rs:58) and stays open. */
    void *reader;
    /* non-NULL: used instead of `lr` / `wdl` -- any LrScheduler / WdlScheduler (the reference's are traits,
     * schedule/lr.rs:8-18, wdl.rs:6-14: Sequence, user-defined ...). wdl_fn gets the final superbatch as `max`. */
    float (*lr_fn)(void *user, int batch, int superbatch);
    float (*wdl_fn)(void *user, int batch, int superbatch, int max);
    void *sched_user;
    /* bt_map_opts.blend_fn / weight_fn / fn_user for the loop's mapper (not with device_mapper) */
    float (*blend_fn)(void *user, const void *record32, float blend);
    float (*weight_fn)(void *user, const void *record32);
    void *fn_user;
    /* non-NULL (a bt_comm of section 7, world = dp_world, rank = dp_rank): the loop exchanges gradients through it
     * with bt_net_dp_step -- RCCL issued from C, nothing of the host language on the step path -- and `grad_sync` is
     * not used. The net needs a library-owned gradient buffer or dp_loss_slot = 1. */
    void *comm;
    size_t dp_shard_min_floats; /* bt_net_dp_plan's threshold; 0 = BT_DP_SHARD_MIN_FLOATS */
} bt_train_config;
/* run::measure_max_cpu_throughput (run.rs:47-73): reader + host mapper alone over the schedule of `cfg` (no device
 * work); reports per superbatch when cfg->log_rate > 0 and returns the overall positions per second. */
int bt_measure_loader_throughput(const bt_input_desc *input, const bt_train_config *cfg, const char *const *paths,
                                 int n_paths, double *positions_per_sec);
typedef void (*bt_batch_callback)(void *user, int superbatch, int batch, float loss);
typedef void (*bt_superbatch_callback)(void *user, int superbatch, float running_loss, float positions_per_sec);
int bt_train_run(bt_net *net, const bt_input_desc *input, const bt_train_config *cfg, const char *const *paths,
                 int n_paths, bt_batch_callback on_batch, bt_superbatch_callback on_superbatch, void *user);

/* ------------------------------------------------------------------------------------------
 * 6. Checkpoints and saved network files (host code)
 *    (crates/trainer/src/model/weights.rs:95-177,183-334,404-467, crates/trainer/src/optimiser.rs:218-236,
 *     crates/trainer/src/optimiser/adam.rs:300-324, crates/bullet_lib/src/value/save.rs:24-85)
 * ------------------------------------------------------------------------------------------ */

/* Weight ids of the net family in flat-buffer order: "l0w","l0b","l1w","l1b",...[,"l0f"] (builder names,
 * crates/compiler/src/model/builder.rs:120-131) and their [rows x cols] shapes (column-major). */
const char *bt_net_tensor_name(const bt_net_desc *desc, int tensor);
int bt_net_tensor_shape(const bt_net_desc *desc, int tensor, int *rows, int *cols);

/* weights.bin / momentum.bin / velocity.bin: a stream of records `id '\n' | u64 LE count | f32 LE values`
 * (weights.rs:404-431), one per weight, in the reference's BTreeMap order = ascending id ("l0b","l0f","l0w",...).
 * flat_host is a bt_net_param_count() buffer. Reading accepts any record order; an unknown id or a size
 * mismatch is an error ("No weight with ID found!", "Invalid buffer size!", weights.rs:119-127). */
int bt_checkpoint_write(const bt_net_desc *desc, const float *flat_host, const char *path);
int bt_checkpoint_read(const bt_net_desc *desc, float *flat_host, const char *path);

/* SavedFormat (weights.rs:183-275). One entry of the network file: either custom bytes, or the weight `id`,
 * transformed, then quantised. Transforms run in this order: merge_factoriser (adds l0f.repeat(NB): the closure of
 * examples/progression/3_input_buckets.rs:47-51), the optional callback, transpose (weights.rs:262-275). */
enum { BT_QUANT_FLOAT = 0, BT_QUANT_I8 = 1, BT_QUANT_I16 = 2, BT_QUANT_I32 = 3 };
typedef int (*bt_save_transform)(void *user, const char *id, float *values, size_t count); /* in place; 0 = ok */
typedef struct {
    const char *id;      /* weight id; NULL for a custom entry */
    const void *custom;  /* SavedFormat::custom bytes (header, padding ...) */
    size_t custom_len;
    int quant;           /* BT_QUANT_* (QuantTarget, weights.rs:277-286) */
    int multiplier;      /* q in round_or_trunc(q * x) computed in f64 (weights.rs:288-331) */
    int round;           /* SavedFormat::round: round half away from zero instead of truncating */
    int transpose;
    int merge_factoriser;
    bt_save_transform transform;
    void *user;
} bt_saved_format;

/* quantised.bin: entries back to back, padded to a multiple of 64 bytes with the characters of "bullet"
 * (weights.rs:160-177). A value that does not fit the target type fails with
 * "Failed quantisation from f32 to i16!" and writes nothing. raw.bin: the untransformed f32 weights of every
 * entry that has an id (save.rs:46-68). */
int bt_save_quantised(const bt_net_desc *desc, const float *flat_host, const bt_saved_format *fmt, int n_fmt,
                      const char *path);
int bt_save_raw(const bt_net_desc *desc, const float *flat_host, const bt_saved_format *fmt, int n_fmt, const char *path);

/* save_to_checkpoint (save.rs:24-43): dir/optimiser_state/{weights,momentum,velocity}.bin, dir/raw.bin,
 * dir/quantised.bin from the device state of `net` (synchronises `stream`). load = Optimiser::load_from_checkpoint
 * (optimiser.rs:232-236): weights, momentum and velocity back onto the device. */
int bt_net_save_checkpoint(bt_net *net, void *stream, const bt_saved_format *fmt, int n_fmt, const char *dir);
int bt_net_load_checkpoint(bt_net *net, void *stream, const char *dir);

/* ------------------------------------------------------------------------------------------
 * 7. Data-parallel exchange (no counterpart in the reference: it drives ONE device, run.rs:84, and holds no
 *    collective library -- SURVEY.md 8e). One process per GPU; RCCL over xGMI, bound at run time (dlopen of
 *    librccl), so the library still loads on a host without it.
 * ------------------------------------------------------------------------------------------ */
typedef struct bt_comm bt_comm;
/* Rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means (a file, a socket,
 * torch.distributed's store ...); every rank then joins with its own GPU current (bt_device_set). */
#define BT_COMM_ID_BYTES 128
int bt_comm_unique_id(void *id128);
int bt_comm_init_rank(const void *id128, int rank, int world, bt_comm **comm);
/* A communicator whose collectives are the caller's functions (another collective library; gloo staged through the
 * host in the tests). Each sums / gathers floats in device memory, ordered after the work already enqueued on
 * `stream` and before what is enqueued on it afterwards; returns 0 on success. reduce_scatter: rank r receives the
 * sum of everyone's send[r*n .. (r+1)*n) (recv may be send + rank*n); all_gather: recv[r*n .. (r+1)*n) = rank r's
 * send (send may be recv + rank*n). reduce_scatter / all_gather may be NULL: the table is then never sharded. */
typedef struct {
    int (*all_reduce)(void *user, float *buf_dev, size_t n, void *stream);
    int (*reduce_scatter)(void *user, const float *send_dev, float *recv_dev, size_t n_per_rank, void *stream);
    int (*all_gather)(void *user, const float *send_dev, float *recv_dev, size_t n_per_rank, void *stream);
    void *user;
} bt_comm_hooks;
int bt_comm_from_hooks(const bt_comm_hooks *hooks, int rank, int world, bt_comm **comm);
/* 1 if librccl and every symbol the native communicator binds can be loaded in this process, else 0 (the reason is in
 * bt_last_error()). Launchers ask every rank BEFORE bt_comm_init_rank, which blocks until all ranks have joined: a rank
 * that cannot load the library would otherwise leave the others waiting inside ncclCommInitRank. */
int bt_comm_available(void);
/* Gives a hook communicator a collective stream of its own, as a native RCCL communicator has from the start:
 * bt_net_dp_step then calls the hooks with THAT stream and fences it against the step's stream with events (the
 * table's reduce-scatter runs beside the optimiser step of the other tensors). For collective libraries that want
 * their own stream; no counterpart in the reference (it has no multi-device path). */
int bt_comm_use_own_stream(bt_comm *comm);
int bt_comm_destroy(bt_comm *comm);
int bt_comm_rank(const bt_comm *comm);
int bt_comm_world(const bt_comm *comm);
/* In-place sum over ranks / reduce-scatter / all-gather of f32 in device memory, in `stream` order. */
int bt_allreduce_sum_f32(bt_comm *comm, void *stream, float *buf_dev, size_t n);
int bt_reduce_scatter_sum_f32(bt_comm *comm, void *stream, const float *send_dev, float *recv_dev, size_t n_per_rank);
int bt_allgather_f32(bt_comm *comm, void *stream, const float *send_dev, float *recv_dev, size_t n_per_rank);

/* How one step's exchange is split (host arithmetic only). A feature-transformer table (tensor 0, "l0w") of at
 * least `shard_min_floats` floats whose size divides by `world` is SHARDED: its gradient is reduce-scattered, rank r
 * runs the optimiser on floats [slice_lo, slice_lo + per_rank) of it only (and keeps momentum / velocity current for
 * that slice only), and the updated slices are all-gathered; the other tensors (and the loss slot behind the flat
 * gradient) are all-reduced as one message [tail_off, tail_off + tail_n) and stepped on every rank. Otherwise
 * sharded = 0 and the whole flat gradient + loss slot is one all-reduce. Every optimiser here is element-wise, so the
 * sharded step writes the same bits as the replicated one. shard_min_floats = 0 picks BT_DP_SHARD_MIN_FLOATS. */
#define BT_DP_SHARD_MIN_FLOATS ((size_t)1 << 22)
typedef struct {
    int sharded;
    size_t table_n, per_rank, slice_lo; /* sharded only */
    size_t tail_off, tail_n;            /* the all-reduced range of the flat gradient, INCLUDING the loss slot */
} bt_dp_plan;
int bt_net_dp_plan(const bt_net_desc *desc, int world, int rank, size_t shard_min_floats, bt_dp_plan *plan);
/* One data-parallel optimiser step after bt_net_forward_backward(net, stream, shard, loss): exchanges the gradient as
 * bt_net_dp_plan says and applies bt_net_optimiser_step's update (adj = 1 / GLOBAL batch). The loss sum must sit in
 * the slot behind the flat gradient (bt_net_grads(net) + bt_net_flat_size(net)) and is summed with it. With an RCCL
 * communicator the collectives run on the communicator's own stream, fenced by events against `stream`, so the
 * table's reduce-scatter overlaps the optimiser step of the other tensors. comm == NULL: a plain step. */
int bt_net_dp_step(bt_net *net, bt_comm *comm, void *stream, const float *adj_dev, const float *lr_dev,
                   size_t shard_min_floats);
/* After sharded steps momentum / velocity are current on their owner only: all-gather them (before a checkpoint).
 * bt_train_run does this before it returns. No-op when the plan is not sharded. */
int bt_net_dp_gather_state(bt_net *net, bt_comm *comm, void *stream, size_t shard_min_floats);

#ifdef __cplusplus
}
#endif
#endif
