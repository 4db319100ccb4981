// The lowered training step of the ValueTrainerBuilder net family as one object:
// what `Function::execute(backwards)` + `Optimiser::update` do per batch in
// crates/trainer/src/run.rs:185-192, with the op schedule of SURVEY.md section 3.2 hard-wired
// (FT fwd x perspectives -> dense stack with bucket select -> sigmoid-MSE -> dense bwd -> FT bwd -> AdamW).
// All buffers are allocated once in bt_net_create (Function::prealloc, function.rs:180-196);
// a step only enqueues kernels on the caller's stream -- no host sync, no allocation.
#include <sys/stat.h>

#include <algorithm>
#include <string>
#include <vector>

#include "common.h"

struct bt_net {
    bt_net_desc d;
    int max_batch;
    int n_tensors;
    size_t n_params;
    std::vector<size_t> t_off, t_size;
    float *params = nullptr, *grads = nullptr, *mom = nullptr, *vel = nullptr;
    bool own_params = false, own_grads = false;
    bt_adamw_tensor adam[BT_ADAMW_MAX_TENSORS];
    // optimiser selection (bt_net_set_optimiser): RAdam / Ranger keep an update counter; Ranger also slow weights
    int opt_kind = BT_OPT_ADAMW, opt_step = 0, opt_k = 6;
    float opt_thr = 5.0f, opt_alpha = 0.5f;
    float *slow = nullptr;
    // activations
    float *hid = nullptr, *saved = nullptr, *g_hid = nullptr, *grad_y = nullptr;
    float *gz0 = nullptr;            // [B] head pre-activation gradient (fused FT backward)
    float *loss_partials = nullptr;  // per-workgroup loss partials of the fused head forward
    int n_loss_partials = 0;
    float *zpart = nullptr;          // [n_inputs * row tiles][B] partial head outputs of the fused FT forward
    float *gz_parts = nullptr;       // [loss partials][buckets] sums of gz0 by bucket (head bias gradient partials)
    bool fuse_loss = false, fuse_ft = false;
    float *a[BT_MAX_DENSE] = {nullptr}, *ga[BT_MAX_DENSE] = {nullptr};
    // layers with a dual activation / skip neurons: a[k] holds the raw outputs z, ap[k] what the next layer sees,
    // gzp[k] the gradient w.r.t. z; yfin = last layer + skip neurons; weff = l0w + l0f.repeat(NB)
    float *ap[BT_MAX_DENSE] = {nullptr}, *gzp[BT_MAX_DENSE] = {nullptr};
    float *yfin = nullptr, *weff = nullptr;
    void *perm = nullptr;  // positions in output-bucket order + bucket-pure items for the bucketed layers (dense.hip)
    void *ft_ws = nullptr, *dense_ws = nullptr;
    // side stream: the FT sorted index (depends only on the batch indices) is built while the forward runs
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_index = nullptr, ev_fwd = nullptr, ev_head = nullptr;
    hipEvent_t ev_dfork = nullptr, ev_sums = nullptr;  // multi-layer nets: optimiser-only dense kernels on the side stream
    size_t ft_ws_bytes = 0, dense_ws_bytes = 0;
    // bt_net_hint_next_batch: the index structures of the NEXT batch (segment entry lists, bucket permutation, sorted
    // lists) are built on the side stream at the end of the current step, into a second set of buffers
    void *ft_ws_alt = nullptr, *perm_alt = nullptr;
    bt_batch next = {}, prep_key = {};
    bool has_next = false, prepared = false;
    // single-layer fused path: the completion event that rode on the previous step's last kernel (ev_step, or the
    // caller's done event). The side stream waits for it before it builds the next batch's index during this step, so
    // this step needs no fork packet on `s`. Null after anything else was queued that reads the index workspace.
    hipEvent_t ev_step = nullptr, prev_done = nullptr;
    hipEvent_t next_ready = nullptr, ev_next = nullptr, ev_next_groups = nullptr;
    // bt_net_use_graphs: steady-state steps of the single-layer fused path (batch prepared ahead, next batch hinted) are
    // captured once per (batch tensors, next batch tensors, buffer set, stream, ...) and replayed as one hipGraph launch
    struct StepGraph {
        bt_batch cur, next;
        const void *loss, *adj, *lr, *ws;
        hipStream_t stream;
        bool with_opt, stepped;
        hipGraphExec_t exec;
    };
    bool use_graphs = false, capturing = false, cap_ok = false, last_graph = false;
    hipEvent_t ev_cap_join = nullptr;
    std::vector<StepGraph> graphs;
    int graph_launches = 0;
};

namespace {

int net_validate(const bt_net_desc *d) {
    BT_REQUIRE(d, "bt_net: null descriptor");
    BT_REQUIRE(d->num_inputs > 0 && d->nnz > 0 && d->ft_out > 0, "bt_net: Mismatched sizes! (FT dims)");
    BT_REQUIRE(d->perspectives == 1 || d->perspectives == 2, "bt_net: perspectives must be 1 or 2");
    BT_REQUIRE(d->n_dense >= 1 && d->n_dense <= BT_MAX_DENSE, "bt_net: n_dense out of range");
    BT_REQUIRE(d->num_buckets >= 1, "bt_net: num_buckets must be >= 1");
    BT_REQUIRE(!d->pairwise || d->ft_out % 2 == 0, "bt_net: pairwise needs even ft_out");
    BT_REQUIRE(d->ft_segment >= 0 && (d->ft_segment == 0 || d->num_inputs % d->ft_segment == 0),
               "bt_net: ft_segment must divide num_inputs");
    BT_REQUIRE(d->loss_kind >= BT_LOSS_SIGMOID_MSE && d->loss_kind <= BT_LOSS_SOFTMAX_CE, "bt_net: unknown loss kind");
    if (d->loss_kind == BT_LOSS_SOFTMAX_CE)
        BT_REQUIRE(d->dense_out[d->n_dense - 1] >= 2, "bt_net: softmax cross-entropy needs >= 2 outputs per bucket");
    else
        BT_REQUIRE(d->dense_out[d->n_dense - 1] == 1, "bt_net: last layer must have one output per bucket");
    BT_REQUIRE(2 + 2 * d->n_dense + (d->ft_factoriser > 0) <= BT_ADAMW_MAX_TENSORS, "bt_net: too many tensors");
    BT_REQUIRE(d->ft_factoriser >= 0 && (d->ft_factoriser == 0 || d->num_inputs % d->ft_factoriser == 0),
               "bt_net: ft_factoriser must divide num_inputs");
    int skips = 0;
    for (int k = 0; k < d->n_dense; k++) {
        BT_REQUIRE(d->dense_skip[k] >= 0 && d->dense_skip[k] < d->dense_out[k], "bt_net: bad dense_skip[%d]", k);
        BT_REQUIRE(k < d->n_dense - 1 || (d->dense_skip[k] == 0 && d->dense_dual[k] == 0),
                   "bt_net: the last layer cannot have skip neurons or a dual activation");
        skips += d->dense_skip[k];
    }
    BT_REQUIRE(skips == 0 || skips == d->dense_out[d->n_dense - 1], "bt_net: skip neurons must match the net output");
    for (int k = 0; k < d->n_dense; k++)
        BT_REQUIRE(d->dense_skip[k] == 0 || d->dense_skip[k] == skips, "bt_net: one layer may own the skip neurons");
    return BT_OK;
}

bool has_post(const bt_net_desc *d, int k) { return d->dense_dual[k] || d->dense_skip[k]; }
bool has_skip(const bt_net_desc *d) {
    for (int k = 0; k < d->n_dense; k++)
        if (d->dense_skip[k]) return true;
    return false;
}
// width of what the next layer sees of layer k
int dense_width(const bt_net_desc *d, int k) { return (d->dense_out[k] - d->dense_skip[k]) * (d->dense_dual[k] ? 2 : 1); }

int dense_in(const bt_net_desc *d, int k) {
    if (k == 0) return d->perspectives * (d->pairwise ? d->ft_out / 2 : d->ft_out);
    return dense_width(d, k - 1);
}
int dense_rows(const bt_net_desc *d, int k) { return d->dense_out[k] * (d->dense_bucketed[k] ? d->num_buckets : 1); }

bt_ft_desc ft_desc(const bt_net *n, int batch) {
    const bt_net_desc &d = n->d;
    bt_ft_desc f;
    f.batch = batch, f.rows = d.ft_out, f.cols = d.num_inputs, f.stride = d.ft_out, f.offset = 0, f.nnz = d.nnz;
    f.act = d.ft_act, f.pairwise = d.pairwise, f.n_inputs = d.perspectives;
    f.out_ld = dense_in(&d, 0);
    f.segment = d.ft_segment;
    return f;
}
bt_dense_desc dn_desc(const bt_net *n, int k, int batch) {
    const bt_net_desc &d = n->d;
    bt_dense_desc q;
    q.batch = batch, q.in = dense_in(&d, k), q.out = d.dense_out[k];
    q.buckets = d.dense_bucketed[k] ? d.num_buckets : 1;
    q.act = has_post(&d, k) ? BT_ACT_IDENTITY : d.dense_act[k];  // post-op layers are activated by head_post_fwd
    return q;
}

// ---- small elementwise kernels of the extended head (dual activation, skip neurons) and the factoriser
__global__ void factor_expand_kernel(const float4 *__restrict__ w, const float4 *__restrict__ wf, float4 *__restrict__ weff,
                                     size_t n4, size_t fc4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 a = w[i], b = wf[i % fc4];  // repeat(NB): bucketed column c uses factoriser column c mod (F/NB)
    weff[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

__global__ void factor_reduce_kernel(const float4 *__restrict__ dw, float4 *__restrict__ dwf, size_t fc4, int reps) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fc4) return;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // Broadcast(Cols) backward = sum over the repeats, ascending
    for (int r = 0; r < reps; r++) {
        const float4 v = dw[(size_t)r * fc4 + i];
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    dwf[i] = s;
}

// z [B][out] -> ap [B][wd]: first m = out - skip rows, optionally widened to concat(x, x*x), then activated
__global__ void head_post_fwd_kernel(const float *__restrict__ z, float *__restrict__ ap, int batch, int out, int m,
                                     int dual, int act) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * m) return;
    const int b = t / m, i = t - b * m, wd = dual ? 2 * m : m;
    const float v = z[(size_t)b * out + i];
    ap[(size_t)b * wd + i] = bt_act_fwd(act, v);
    if (dual) ap[(size_t)b * wd + m + i] = bt_act_fwd(act, v * v);
}

// gradient w.r.t. z from the gradient w.r.t. ap (activation backward from the output, Concat/Slice backward, Mul
// backward with both operands x: contributions in reverse creation order) and, for the skip rows, w.r.t. the output
__global__ void head_post_bwd_kernel(const float *__restrict__ gap, const float *__restrict__ ap,
                                     const float *__restrict__ z, const float *__restrict__ gy, float *__restrict__ gz,
                                     int batch, int out, int m, int dual, int act, int n_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * out) return;
    const int b = t / out, i = t - b * out, wd = dual ? 2 * m : m;
    float g;
    if (i < m) {
        g = gap[(size_t)b * wd + i] * bt_act_deriv(act, ap[(size_t)b * wd + i]);
        if (dual) {
            const float g2 = gap[(size_t)b * wd + m + i] * bt_act_deriv(act, ap[(size_t)b * wd + m + i]);
            const float xv = z[(size_t)b * out + i];
            g = (g + g2 * xv) + g2 * xv;
        }
    } else {
        g = gy[(size_t)b * n_out + (i - m)];
    }
    gz[t] = g;
}

// y[b][j] += z[b][out - skip + j]
__global__ void head_skip_add_kernel(float *__restrict__ y, const float *__restrict__ z, int batch, int out, int skip) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * skip) return;
    const int b = t / skip, j = t - b * skip;
    y[t] = y[t] + z[(size_t)b * out + out - skip + j];
}

// the table the FT runs on: l0w, or l0w + l0f.repeat(NB) rebuilt for this step
int ft_weights(bt_net *n, hipStream_t s, const float **w0) {
    *w0 = n->params + n->t_off[0];
    if (n->d.ft_factoriser <= 0) return BT_OK;
    const size_t n4 = n->t_size[0] / 4, fc4 = n4 / n->d.ft_factoriser;
    factor_expand_kernel<<<dim3((unsigned)bt_div_up((long long)n4, 256)), dim3(256), 0, s>>>(
        reinterpret_cast<const float4 *>(n->params + n->t_off[0]),
        reinterpret_cast<const float4 *>(n->params + n->t_off[n->n_tensors - 1]), reinterpret_cast<float4 *>(n->weff), n4,
        fc4);
    BT_LAUNCH_CHECK();
    *w0 = n->weff;
    return BT_OK;
}

int dmalloc(float **p, size_t floats) {
    void *v = nullptr;
    if (int rc = bt_malloc(&v, floats * sizeof(float))) return rc;
    *p = static_cast<float *>(v);
    return BT_OK;
}

int check_batch(const bt_net *net, const bt_batch *b, const char *who) {
    BT_REQUIRE(net && b, "%s: null argument", who);
    BT_REQUIRE(b->batch > 0 && b->batch <= net->max_batch, "%s: Mismatched sizes! (batch %d, max %d)", who, b->batch,
               net->max_batch);
    BT_REQUIRE(b->stm && b->targets, "%s: null batch tensor", who);
    BT_REQUIRE(net->d.perspectives == 1 || b->nstm, "%s: dual-perspective net needs nstm", who);
    BT_REQUIRE(net->d.num_buckets == 1 || b->buckets, "%s: bucketed net needs bucket indices", who);
    BT_REQUIRE(!net->d.use_entry_weights || b->entry_weights, "%s: net needs entry_weights", who);
    return BT_OK;
}

// builds the bucket-order permutation of this batch (if the net has a layer that uses one) and installs it for the
// calling thread until the scope ends
struct PermScope {
    int rc = BT_OK;
    bool built = false;
    // build = false: n->perm already holds this batch's permutation (bt_net_hint_next_batch), install it only
    PermScope(bt_net *n, hipStream_t s, const bt_batch *b, bool build = true) {
        if (!perm_usable(n, b)) return;
        if (build) rc = bt_internal_bucket_perm(s, b->buckets, b->batch, n->d.num_buckets, n->perm);
        if (rc == BT_OK) bt_internal_dense_set_perm(n->perm, b->batch, n->d.num_buckets), built = true;
    }
    static bool perm_usable(const bt_net *n, const bt_batch *b) { return n->perm && b->buckets && b->batch <= 64 * 1024; }
    ~PermScope() { bt_internal_dense_set_perm(nullptr, 0, 0); }
};

// groups_ready: the segment entry lists of this batch are already in ft_ws (bt_net_forward_backward builds them
// before it forks the side stream); otherwise they are built here, in stream order
// perm_on_side: the bucket-order permutation is being built on the side stream, in front of the sorted index: the
// side stream is joined before the layers, once for both (ev_index; the index build ends before the FT forward does,
// and a second wait packet before the FT backward would cost another bubble on `s`)
// n_layers: dense layers to run (the fused tail of bt_net_forward_backward takes over after the first)
// fwd_event: record ev_fwd when the forward pass is complete (the side-stream head backward waits for it)
int forward_impl(bt_net *n, hipStream_t s, const bt_batch *b, bool fuse_loss, bool want_gz0 = false,
                 bool head_in_ft = false, bool groups_ready = false, bool perm_on_side = false, int n_layers = -1,
                 bool fwd_event = true, hipEvent_t join_layers = nullptr, bool joined = false) {
    bt_ft_desc f = ft_desc(n, b->batch);
    const int32_t *idx[2] = {b->stm, b->nstm};
    const float *w0 = nullptr;
    if (int rc = ft_weights(n, s, &w0)) return rc;
    if (!groups_ready)
        if (int rc = bt_internal_ft_build_groups(s, &f, idx, n->ft_ws, n->ft_ws_bytes)) return rc;
    if (head_in_ft) {
        // single one-output layer: its dot products are formed inside the FT forward, tile by tile, and a small
        // finishing kernel adds the partials, the bias, the loss and its gradient
        bt_dense_desc q = dn_desc(n, 0, b->batch);
        const int32_t *bk = n->d.dense_bucketed[0] ? b->buckets : nullptr;
        if (int rc = bt_internal_ft_forward_head(s, &f, w0, n->params + n->t_off[1], idx, n->hid,
                                                 n->saved, n->params + n->t_off[2], bk, q.buckets, n->zpart, n->ft_ws,
                                                 n->ft_ws_bytes))
            return rc;
        // Cross-stream packets cost a few us of bubble each on this stream. The index join goes here, between the FT
        // forward and the finishing kernel (it has long been satisfied), and "forward done" (ev_fwd) rides on the
        // finishing kernel's own dispatch packet instead of a separate hipEventRecord -- when somebody waits for it:
        // a completion signal on a packet delays the next kernel of the stream by 4-5 us too.
        if (!joined) BT_HIP(hipStreamWaitEvent(s, join_layers ? join_layers : n->ev_index, 0));
        if (fwd_event) bt_attach_stop_event(BTK_DENSE_FWD_SMALL, n->ev_fwd);
        const int rc = bt_internal_head_finish(s, &q, n->zpart, bt_internal_ft_head_parts(&f), n->params + n->t_off[3], bk,
                                               b->targets, n->d.use_entry_weights ? b->entry_weights : nullptr, n->a[0],
                                               n->grad_y, want_gz0 ? n->gz0 : nullptr, n->loss_partials,
                                               &n->n_loss_partials, want_gz0 ? n->gz_parts : nullptr);
        if (rc && fwd_event) (void)bt_take_stop_event(BTK_DENSE_FWD_SMALL);  // not launched: do not leave the event armed
        return rc;
    }
    if (int rc = bt_ft_forward_prebuilt(s, &f, w0, n->params + n->t_off[1], idx, n->hid, n->saved, n->ft_ws,
                                        n->ft_ws_bytes))
        return rc;
    const float *x = n->hid;
    if (perm_on_side) BT_HIP(hipStreamWaitEvent(s, n->ev_index, 0));
    if (join_layers) BT_HIP(hipStreamWaitEvent(s, join_layers, 0));  // a batch prepared ahead: its bucket order and sorted index
    if (n_layers < 0) n_layers = n->d.n_dense;
    for (int k = 0; k < n_layers; k++) {
        bt_dense_desc q = dn_desc(n, k, b->batch);
        if (fuse_loss && k == n->d.n_dense - 1) {
            // final layer + sigmoid-MSE + its gradient in one kernel; the partials are summed by the head backward
            if (int rc = bt_internal_dense_forward_loss(
                    s, &q, n->params + n->t_off[2 + 2 * k], n->params + n->t_off[3 + 2 * k], x, b->buckets, b->targets,
                    n->d.use_entry_weights ? b->entry_weights : nullptr, n->a[k], n->grad_y,
                    (want_gz0 && k == 0) ? n->gz0 : nullptr, n->loss_partials, &n->n_loss_partials))
                return rc;
        } else if (int rc = bt_dense_forward(s, &q, n->params + n->t_off[2 + 2 * k], n->params + n->t_off[3 + 2 * k], x,
                                             b->buckets, n->a[k]))
            return rc;
        x = n->a[k];
        if (has_post(&n->d, k)) {
            const int m = q.out - n->d.dense_skip[k];
            head_post_fwd_kernel<<<dim3((unsigned)bt_div_up((long long)b->batch * m, 256)), dim3(256), 0, s>>>(
                n->a[k], n->ap[k], b->batch, q.out, m, n->d.dense_dual[k], n->d.dense_act[k]);
            BT_LAUNCH_CHECK();
            x = n->ap[k];
        }
    }
    if (has_skip(&n->d) && n_layers == n->d.n_dense) {  // net output = last layer + skip neurons
        const int L = n->d.n_dense, n_out = n->d.dense_out[L - 1];
        BT_HIP(hipMemcpyAsync(n->yfin, n->a[L - 1], sizeof(float) * b->batch * n_out, hipMemcpyDeviceToDevice, s));
        for (int k = 0; k < L - 1; k++)
            if (n->d.dense_skip[k]) {
                head_skip_add_kernel<<<dim3((unsigned)bt_div_up((long long)b->batch * n_out, 256)), dim3(256), 0, s>>>(
                    n->yfin, n->a[k], b->batch, n->d.dense_out[k], n->d.dense_skip[k]);
                BT_LAUNCH_CHECK();
            }
    }
    return BT_OK;
}

// end of a step: build the hinted batch's index structures on the side stream into the second buffer set. The side
// stream is ordered behind ev_fork, i.e. behind everything queued on the compute stream before this step, which
// includes the last readers of that buffer set (the step before this one) -- or it is behind a kernel of this step
// already (side_behind: the fused tail's fork), and no ev_fork was recorded.
int prepare_next(bt_net *n, bool side_behind) {
    if (!n->has_next) return BT_OK;
    n->has_next = false;
    const bt_batch *b = &n->next;
    bt_ft_desc f = ft_desc(n, b->batch);
    const int32_t *idx[2] = {b->stm, b->nstm};
    if (!side_behind) BT_HIP(hipStreamWaitEvent(n->side, n->ev_fork, 0));
    // (the batch's ready event: usually long complete, and then not worth a wait packet)
    if (n->next_ready && !n->capturing && hipEventQuery(n->next_ready) != hipSuccess) {
        (void)hipGetLastError();  // ("not ready" is an answer, not an error to be found by the next launch check)
        BT_HIP(hipStreamWaitEvent(n->side, n->next_ready, 0));
    }
    // The next step's FT forward needs the segment entry lists only; the bucket order and the sorted index are joined
    // later (before the layers): ev_next_groups rides on the lists' last kernel (a table that is not grouped has no lists
    // and no such event)
    const bool grouped = bt_internal_ft_grouped(&f);
    if (int rc = bt_internal_ft_build_groups(n->side, &f, idx, n->ft_ws_alt, n->ft_ws_bytes, grouped ? n->ev_next_groups : nullptr))
        return rc;
    if (n->perm_alt && PermScope::perm_usable(n, b))
        if (int rc = bt_internal_bucket_perm(n->side, b->buckets, b->batch, n->d.num_buckets, n->perm_alt)) return rc;
    // (ev_next rides on the index kernel's packet)
    if (int rc = bt_internal_ft_build_sorted(n->side, &f, idx, n->ft_ws_alt, n->ft_ws_bytes, n->ev_next)) return rc;
    n->prepared = true, n->prep_key = *b;
    return BT_OK;
}

const float *net_output(const bt_net *n) { return has_skip(&n->d) ? n->yfin : n->a[n->d.n_dense - 1]; }

}  // namespace

// captured steps hold kernel arguments by value (optimiser settings among them): whoever changes those drops the graphs
void bt_internal_net_drop_graphs(bt_net *n) {
    for (auto &g : n->graphs) (void)hipGraphExecDestroy(g.exec);
    n->graphs.clear();
    n->last_graph = false;
}

int bt_internal_net_enable_hints(bt_net *n) {
    if (n->ft_ws_alt) return BT_OK;
    if (int rc = bt_malloc(&n->ft_ws_alt, n->ft_ws_bytes)) return rc;
    if (n->perm)
        if (int rc = bt_malloc(&n->perm_alt, bt_internal_bucket_perm_bytes(n->max_batch, n->d.num_buckets))) return rc;
    return BT_OK;
}

extern "C" {

int bt_net_num_tensors(const bt_net_desc *d) { return d ? 2 + 2 * d->n_dense + (d->ft_factoriser > 0 ? 1 : 0) : 0; }

size_t bt_net_tensor_size(const bt_net_desc *d, int t) {
    if (!d || t < 0 || t >= bt_net_num_tensors(d)) return 0;
    if (t == 2 + 2 * d->n_dense) return (size_t)d->ft_out * (d->num_inputs / d->ft_factoriser);  // l0f
    if (t == 0) return (size_t)d->ft_out * d->num_inputs;
    if (t == 1) return (size_t)d->ft_out;
    const int k = (t - 2) / 2;
    if ((t & 1) == 0) return (size_t)dense_rows(d, k) * dense_in(d, k);
    return (size_t)dense_rows(d, k);
}

size_t bt_net_tensor_offset(const bt_net_desc *d, int t) {
    size_t off = 0;
    for (int i = 0; i < t && i < bt_net_num_tensors(d); i++) off += (bt_net_tensor_size(d, i) + 3) / 4 * 4;
    return off;
}

size_t bt_net_param_count(const bt_net_desc *d) {
    size_t n = 0;
    for (int t = 0; t < bt_net_num_tensors(d); t++) n += (bt_net_tensor_size(d, t) + 3) / 4 * 4;  // 16-B aligned tensors
    return n;
}

int bt_net_create(const bt_net_desc *desc, int max_batch, float *ext_params, float *ext_grads, bt_net **out) {
    if (int rc = net_validate(desc)) return rc;
    BT_REQUIRE(max_batch > 0 && out, "bt_net_create: bad arguments");
    bt_net *n = new bt_net();
    n->d = *desc;
    n->max_batch = max_batch;
    n->n_tensors = bt_net_num_tensors(desc);
    size_t off = 0;
    for (int t = 0; t < n->n_tensors; t++) {
        n->t_off.push_back(off);
        n->t_size.push_back(bt_net_tensor_size(desc, t));
        off += (n->t_size[t] + 3) / 4 * 4;
    }
    n->n_params = off;
    int rc = BT_OK;
    auto fail = [&](int code) {
        bt_net_destroy(n);
        return code;
    };
    if (ext_params) n->params = ext_params;
    else {
        if ((rc = dmalloc(&n->params, off))) return fail(rc);
        n->own_params = true;
    }
    if (ext_grads) n->grads = ext_grads;
    else {
        if ((rc = dmalloc(&n->grads, off + 4))) return fail(rc);  // + the loss slot of data-parallel steps
        n->own_grads = true;
    }
    if ((rc = dmalloc(&n->mom, off))) return fail(rc);
    if ((rc = dmalloc(&n->vel, off))) return fail(rc);
    for (int t = 0; t < n->n_tensors; t++) {
        bt_adamw_tensor &a = n->adam[t];
        a.grad = n->grads + n->t_off[t], a.param = n->params + n->t_off[t];
        a.momentum = n->mom + n->t_off[t], a.velocity = n->vel + n->t_off[t];
        a.n = n->t_size[t];
        a.decay = 0.01f, a.beta1 = 0.9f, a.beta2 = 0.999f, a.min_weight = -1.98f, a.max_weight = 1.98f;  // adam.rs:29-33
    }
    const size_t B = (size_t)max_batch;
    const int K0 = dense_in(desc, 0);
    if ((rc = dmalloc(&n->hid, B * K0))) return fail(rc);
    if ((rc = dmalloc(&n->g_hid, B * K0))) return fail(rc);
    if (desc->pairwise && (rc = dmalloc(&n->saved, B * desc->perspectives * desc->ft_out))) return fail(rc);
    if ((rc = dmalloc(&n->grad_y, B * desc->dense_out[desc->n_dense - 1]))) return fail(rc);
    if ((rc = dmalloc(&n->gz0, B))) return fail(rc);
    if ((rc = dmalloc(&n->loss_partials, 1024))) return fail(rc);
    if ((rc = dmalloc(&n->gz_parts, (size_t)1024 * std::max(1, desc->num_buckets)))) return fail(rc);
    {
        bt_ft_desc fmax = ft_desc(n, B);
        if ((rc = dmalloc(&n->zpart, (size_t)(bt_internal_ft_head_parts(&fmax) + 1) * B))) return fail(rc);
    }
    if (desc->ft_factoriser > 0 && (rc = dmalloc(&n->weff, n->t_size[0]))) return fail(rc);
    if (has_skip(desc) && (rc = dmalloc(&n->yfin, B * desc->dense_out[desc->n_dense - 1]))) return fail(rc);
    for (int k = 0; k < desc->n_dense; k++) {
        const size_t wide = (size_t)std::max(desc->dense_out[k], dense_width(desc, k));
        if ((rc = dmalloc(&n->a[k], B * desc->dense_out[k]))) return fail(rc);
        if ((rc = dmalloc(&n->ga[k], B * wide))) return fail(rc);
        if (has_post(desc, k)) {
            if ((rc = dmalloc(&n->ap[k], B * dense_width(desc, k)))) return fail(rc);
            if ((rc = dmalloc(&n->gzp[k], B * desc->dense_out[k]))) return fail(rc);
        }
        bt_dense_desc q = dn_desc(n, k, max_batch);
        size_t wsb = bt_dense_backward_workspace_bytes(&q);
        if (wsb == 0) return fail(BT_ERR_INVALID);
        if (k >= 1 && k + 1 < desc->n_dense) {  // (fused tail partials, if that path is taken)
            const bt_dense_desc q2 = dn_desc(n, k + 1, max_batch);
            if (bt_internal_dense_tail_ok(&q, &q2)) wsb = std::max(wsb, bt_internal_dense_tail_workspace_bytes(&q));
        }
        if (wsb > n->dense_ws_bytes) n->dense_ws_bytes = wsb;
    }
    for (int k = 0; k < desc->n_dense && !n->perm; k++) {
        bt_dense_desc q = dn_desc(n, k, max_batch);
        if (bt_internal_dense_wants_perm(&q))
            if ((rc = bt_malloc(&n->perm, bt_internal_bucket_perm_bytes(max_batch, desc->num_buckets)))) return fail(rc);
    }
    bt_ft_desc f = ft_desc(n, max_batch);
    n->ft_ws_bytes = bt_ft_backward_workspace_bytes(&f);
    if (n->ft_ws_bytes == 0) return fail(BT_ERR_INVALID);
    if ((rc = bt_malloc(&n->ft_ws, n->ft_ws_bytes))) return fail(rc);
    if ((rc = bt_malloc(&n->dense_ws, n->dense_ws_bytes))) return fail(rc);
    if (hipStreamCreateWithFlags(&n->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_index, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_fwd, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_head, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_dfork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_sums, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_step, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_next, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_next_groups, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_cap_join, hipEventDisableTiming) != hipSuccess) {
        bt_set_error("bt_net_create: could not create the side stream / events");
        return fail(BT_ERR_HIP);
    }
    *out = n;
    return BT_OK;
}

int bt_net_destroy(bt_net *n) {
    if (!n) return BT_OK;
    if (n->own_params) bt_free(n->params);
    if (n->own_grads) bt_free(n->grads);
    bt_free(n->mom), bt_free(n->vel), bt_free(n->hid), bt_free(n->g_hid), bt_free(n->saved), bt_free(n->grad_y);
    bt_free(n->gz0), bt_free(n->loss_partials), bt_free(n->zpart), bt_free(n->gz_parts);
    for (int k = 0; k < BT_MAX_DENSE; k++) bt_free(n->a[k]), bt_free(n->ga[k]), bt_free(n->ap[k]), bt_free(n->gzp[k]);
    bt_free(n->yfin), bt_free(n->weff), bt_free(n->slow), bt_free(n->perm);
    bt_free(n->ft_ws), bt_free(n->dense_ws);
    if (n->ev_fork) (void)hipEventDestroy(n->ev_fork);
    if (n->ev_index) (void)hipEventDestroy(n->ev_index);
    if (n->ev_fwd) (void)hipEventDestroy(n->ev_fwd);
    if (n->ev_head) (void)hipEventDestroy(n->ev_head);
    if (n->ev_next) (void)hipEventDestroy(n->ev_next);
    if (n->ev_next_groups) (void)hipEventDestroy(n->ev_next_groups);
    if (n->ev_dfork) (void)hipEventDestroy(n->ev_dfork);
    if (n->ev_sums) (void)hipEventDestroy(n->ev_sums);
    if (n->ev_step) (void)hipEventDestroy(n->ev_step);
    if (n->ev_cap_join) (void)hipEventDestroy(n->ev_cap_join);
    for (auto &g : n->graphs) (void)hipGraphExecDestroy(g.exec);
    bt_free(n->ft_ws_alt), bt_free(n->perm_alt);
    if (n->side) (void)hipStreamDestroy(n->side);
    delete n;
    return BT_OK;
}

float *bt_net_params(bt_net *n) { return n ? n->params : nullptr; }
float *bt_net_grads(bt_net *n) { return n ? n->grads : nullptr; }
size_t bt_net_flat_size(const bt_net *n) { return n ? n->n_params : 0; }
int bt_net_max_batch(const bt_net *n) { return n ? n->max_batch : 0; }
int bt_net_grads_owned(const bt_net *n) { return n && n->own_grads; }
float *bt_net_momentum(bt_net *n) { return n ? n->mom : nullptr; }
float *bt_net_velocity(bt_net *n) { return n ? n->vel : nullptr; }

int bt_net_set_adamw(bt_net *n, int t, float decay, float beta1, float beta2, float min_w, float max_w) {
    BT_REQUIRE(n && t >= 0 && t < n->n_tensors, "bt_net_set_adamw: bad tensor index");
    BT_REQUIRE(max_w >= min_w, "Invalid clipping: %g >= %g", min_w, max_w);  // adam.rs:259-263
    bt_adamw_tensor &a = n->adam[t];
    a.decay = decay, a.beta1 = beta1, a.beta2 = beta2, a.min_weight = min_w, a.max_weight = max_w;
    bt_internal_net_drop_graphs(n);
    return BT_OK;
}

int bt_net_forward(bt_net *n, void *stream, const bt_batch *b, float *out_dev, float *loss_sum_dev) {
    if (int rc = check_batch(n, b, "bt_net_forward")) return rc;
    hipStream_t s = bt_stream(stream);
    n->prev_done = nullptr;  // (this pass uses the index workspace after the last step's completion event)
    PermScope perm(n, s, b);
    if (perm.rc) return perm.rc;
    if (int rc = forward_impl(n, s, b, false)) return rc;
    const float *y = net_output(n);
    const int n_out = n->d.dense_out[n->d.n_dense - 1];
    if (out_dev) BT_HIP(hipMemcpyAsync(out_dev, y, sizeof(float) * b->batch * n_out, hipMemcpyDeviceToDevice, s));
    if (loss_sum_dev)
        return bt_loss(s, n->d.loss_kind, b->batch, n_out, y, b->targets,
                       n->d.use_entry_weights ? b->entry_weights : nullptr, n->d.loss_power, loss_sum_dev, n->grad_y);
    return BT_OK;
}

namespace {
struct StepOpt {  // bt_net_train_step: the optimiser step that follows, for the paths that can fuse it
    const float *adj, *lr;
    hipEvent_t done;
    bool stepped = false;
};
int forward_backward_impl(bt_net *n, void *stream, const bt_batch *b, float *loss_sum_dev, StepOpt *opt);
}  // namespace

int bt_net_forward_backward(bt_net *n, void *stream, const bt_batch *b, float *loss_sum_dev) {
    return forward_backward_impl(n, stream, b, loss_sum_dev, nullptr);
}

// run.rs:185-192 for one device: forward, backward and the optimiser update of one batch. Where the gradient merge of
// the feature transformer is the last kernel of the backward pass (single-layer head, AdamW, no factoriser), AdamW runs
// inside it, element by element as each gradient sum completes; otherwise this is bt_net_forward_backward followed by
// bt_net_optimiser_step_event. The gradients are written either way, and the results are bit-identical.
int bt_net_train_step(bt_net *n, void *stream, const bt_batch *b, float *loss_sum_dev, const float *adj_dev,
                      const float *lr_dev, void *done_event) {
    BT_REQUIRE(adj_dev && lr_dev, "bt_net_train_step: null adj / lr");
    StepOpt opt{adj_dev, lr_dev, static_cast<hipEvent_t>(done_event)};
    if (int rc = forward_backward_impl(n, stream, b, loss_sum_dev, &opt)) return rc;
    if (opt.stepped) return BT_OK;
    return bt_net_optimiser_step_event(n, stream, adj_dev, lr_dev, done_event);
}

namespace {
bool same_batch(const bt_batch &a, const bt_batch &b) {
    return a.batch == b.batch && a.stm == b.stm && a.nstm == b.nstm && a.buckets == b.buckets && a.targets == b.targets &&
           a.entry_weights == b.entry_weights;
}

// A steady-state step as one hipGraph launch (bt_net_use_graphs). Eligible: this batch's index was built ahead, the next
// batch is hinted, a stream of the caller's (capture needs one), and the step takes the single-layer fused path -- which
// the first attempt finds out by capturing the ordinary enqueue code (cap_ok). Anything else, and any failure while
// capturing, runs the step the ordinary way; a failure switches graphs off for this net.
// Events: the two that come from outside (the hinted batch's ready event, the side stream's build left by an eager step)
// are joined on the stream before the launch, the caller's completion event is recorded behind it; inside the graph the
// side stream's work for the next batch is a branch (fork / join by events of the net, i.e. edges).
int forward_backward_eager(bt_net *n, void *stream, const bt_batch *b, float *loss_sum_dev, StepOpt *opt);
int forward_backward_impl(bt_net *n, void *stream, const bt_batch *b, float *loss_sum_dev, StepOpt *opt) {
    const bool ready = n->prepared && same_batch(n->prep_key, *b) && n->ft_ws_alt;
    const bool eligible = n->use_graphs && stream && ready && n->has_next && loss_sum_dev && (!opt || (opt->adj && opt->lr)) &&
                          !bt_capturing() && !bt_prof_active();  // (a graph's kernels carry no per-kernel timestamps)
    if (!eligible) {
        n->last_graph = false;
        return forward_backward_eager(n, stream, b, loss_sum_dev, opt);
    }
    if (int rc = check_batch(n, b, "bt_net_forward_backward")) return rc;
    hipStream_t s = bt_stream(stream);
    // (the graph of this step uses the buffer set that is about to be swapped in: ft_ws_alt)
    bt_net::StepGraph *g = nullptr;
    for (auto &e : n->graphs)
        if (same_batch(e.cur, *b) && same_batch(e.next, n->next) && e.loss == loss_sum_dev && e.ws == n->ft_ws_alt &&
            e.stream == s && e.with_opt == (opt != nullptr) && (!opt || (e.adj == opt->adj && e.lr == opt->lr)))
            g = &e;
    // joins with what lies outside the graph
    if (!n->last_graph) BT_HIP(hipStreamWaitEvent(s, n->ev_next, 0));
    if (n->next_ready && hipEventQuery(n->next_ready) != hipSuccess) {
        (void)hipGetLastError();
        BT_HIP(hipStreamWaitEvent(s, n->next_ready, 0));
    }
    hipEvent_t done = opt ? opt->done : nullptr;
    if (!g) {
        if (n->graphs.size() >= 16) {  // (a caller that rotates through more buffers than that: not worth graphs)
            n->use_graphs = false, n->last_graph = false;
            return forward_backward_eager(n, stream, b, loss_sum_dev, opt);
        }
        bt_net::StepGraph e{*b, n->next, loss_sum_dev, opt ? opt->adj : nullptr, opt ? opt->lr : nullptr, n->ft_ws_alt, s,
                            opt != nullptr, false, nullptr};
        n->capturing = true, n->cap_ok = false;
        bt_set_capturing(true);
        hipGraph_t graph = nullptr;
        int rc = BT_OK;
        hipError_t he = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        if (he == hipSuccess) {
            StepOpt o2 = opt ? *opt : StepOpt{nullptr, nullptr, nullptr};
            rc = forward_backward_eager(n, stream, b, loss_sum_dev, opt ? &o2 : nullptr);
            e.stepped = o2.stepped;
            if (opt && !o2.stepped) n->cap_ok = false;  // (an optimiser step of its own keeps host-side counters: not captured)
            // the side stream's branch back into the compute stream
            if (hipEventRecord(n->ev_cap_join, n->side) != hipSuccess || hipStreamWaitEvent(s, n->ev_cap_join, 0) != hipSuccess)
                rc = rc ? rc : BT_ERR_HIP;
            he = hipStreamEndCapture(s, &graph);
        }
        bt_set_capturing(false);
        n->capturing = false;
        const bool ok = he == hipSuccess && rc == BT_OK && n->cap_ok && graph &&
                        hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (!ok) {
            // The capture ran the step's host-side bookkeeping (buffer sets swapped, the next batch marked as prepared) but
            // enqueued nothing: undo the bookkeeping and run this step, and every later one, the ordinary way.
            bt_set_error("bt_net: the step could not be captured into a graph; running without graphs");
            std::swap(n->ft_ws, n->ft_ws_alt), std::swap(n->perm, n->perm_alt);
            n->prepared = true, n->prep_key = *b, n->has_next = true, n->next = e.next;
            n->prev_done = nullptr, n->use_graphs = false, n->last_graph = false;
            BT_HIP(hipStreamSynchronize(s));  // (what the side stream built for this batch has been joined above)
            BT_HIP(hipStreamSynchronize(n->side));
            return forward_backward_eager(n, stream, b, loss_sum_dev, opt);
        }
        n->graphs.push_back(e);
        g = &n->graphs.back();
    } else {
        // the bookkeeping the ordinary path does for such a step
        std::swap(n->ft_ws, n->ft_ws_alt), std::swap(n->perm, n->perm_alt);
        n->prepared = true, n->prep_key = n->next, n->has_next = false;
    }
    n->prev_done = nullptr;
    BT_HIP(hipGraphLaunch(g->exec, s));
    if (opt) opt->stepped = g->stepped;
    if (done) BT_HIP(hipEventRecord(done, s));
    n->last_graph = true, n->graph_launches++;
    return BT_OK;
}

int forward_backward_eager(bt_net *n, void *stream, const bt_batch *b, float *loss_sum_dev, StepOpt *opt) {
    if (int rc = check_batch(n, b, "bt_net_forward_backward")) return rc;
    BT_REQUIRE(loss_sum_dev, "bt_net_forward_backward: null loss buffer");
    hipStream_t s = bt_stream(stream);
    bt_ft_desc f = ft_desc(n, b->batch);
    const int32_t *idx[2] = {b->stm, b->nstm};
    // Were this batch's index structures built ahead (bt_net_hint_next_batch during the previous step)? Then they are
    // in the second buffer set: swap the sets and join the side stream once instead of building here.
    const bool ready = n->prepared && n->prep_key.batch == b->batch && n->prep_key.stm == b->stm &&
                       n->prep_key.nstm == b->nstm && n->prep_key.buckets == b->buckets;
    n->prepared = false;
    bool joined = false;  // ready: the side stream's build of this batch has been joined in full already
    if (ready) {
        std::swap(n->ft_ws, n->ft_ws_alt), std::swap(n->perm, n->perm_alt);
        // (the entry lists now; the bucket order and the sorted index before the layers, forward_impl's join_layers: the
        // side stream may still be building them when this step starts. ev_fork for prepare_next: below, unless the
        // tail forks anyway)
        // A table that is not grouped: one join, here (its forward may read lists decoded on the side stream, and the
        // whole build ended during the previous step anyway).
        if (bt_internal_ft_grouped(&f)) BT_HIP(hipStreamWaitEvent(s, n->ev_next_groups, 0));
        else {
            // (under capture the join is made before the graph is launched, graph_step)
            if (!n->capturing) BT_HIP(hipStreamWaitEvent(s, n->ev_next, 0));
            joined = true;
        }
    } else {
        // segment-grouped tables: the entry lists are needed by the forward pass too, so they are built on `s` first
        // fork: sorted index build on the side stream, ordered after everything already queued on `s` (grouped tables:
        // the fork event rides on the entry lists' last kernel instead of a packet of its own)
        const bool ride = bt_internal_ft_grouped(&f);
        if (int rc = bt_internal_ft_build_groups(s, &f, idx, n->ft_ws, n->ft_ws_bytes, ride ? n->ev_fork : nullptr))
            return rc;
        if (!ride) BT_HIP(hipEventRecord(n->ev_fork, s));
        BT_HIP(hipStreamWaitEvent(n->side, n->ev_fork, 0));
    }
    hipEvent_t prev_done = n->prev_done;
    n->prev_done = nullptr;
    PermScope perm(n, n->side, b, !ready);  // (multi-layer bucketed nets) needed after the FT forward: side stream first
    if (perm.rc) return perm.rc;
    if (!ready) {
        if (int rc = bt_internal_ft_build_sorted(n->side, &f, idx, n->ft_ws, n->ft_ws_bytes)) return rc;
        BT_HIP(hipEventRecord(n->ev_index, n->side));
    }
    const int L = n->d.n_dense;
    // fusion decisions (shape-dependent, see dense.hip / ft.hip): loss into the final head's forward, the head's dX
    // into the FT backward staging
    bt_dense_desc q_last = dn_desc(n, L - 1, b->batch), q0 = dn_desc(n, 0, b->batch);
    const bool fuse_loss = n->d.loss_kind == BT_LOSS_SIGMOID_MSE && bt_internal_dense_forward_loss_ok(&q_last) &&
                           bt_internal_dense_head_ok(&q_last) && !has_skip(&n->d);  // with skip neurons the loss is taken from yfin, not the last layer
    const bool fuse_ft = bt_internal_dense_head_ok(&q0) && bt_internal_ft_fused_ok(&f, q0.buckets) && !has_post(&n->d, 0);
    // Single-layer head with both fusions: the forward already produced gz (all the FT backward needs), so the
    // head's own backward (dW1, db1, final loss sum) runs on the side stream NEXT TO the FT backward -- one is
    // HBM-bound in a small-footprint kernel shape, the other LDS-bound, and they share the CUs.
    const bool side_head = fuse_loss && fuse_ft && L == 1;
    const bool head_in_ft = side_head && bt_internal_ft_forward_head_ok(&f, q0.buckets) && b->batch <= 1024 * 64;
    // three-layer nets whose two last layers are tiny and bucketed (4_multi_layer.rs): those layers, the loss and
    // their backward pass run as one kernel per bucket-pure item
    bt_dense_desc q1 = dn_desc(n, L >= 2 ? L - 2 : 0, b->batch);
    const bool tail = L == 3 && perm.built && n->d.loss_kind == BT_LOSS_SIGMOID_MSE && !has_skip(&n->d) &&
                      !has_post(&n->d, 0) && !has_post(&n->d, 1) && n->d.dense_bucketed[1] && n->d.dense_bucketed[2] &&
                      bt_internal_dense_tail_ok(&q1, &q_last);
    // the tail can hand layer 0 the gradient w.r.t. its pre-activation directly (one multiply more in the tail kernel,
    // one elementwise kernel less on the critical path)
    const bool tail_gz = tail && !bt_internal_dense_head_ok(&q0) && bt_internal_dense_takes_gz(&q0);
    // Every packet on `s` costs a bubble of ~5 us: with the fused tail the side stream is put behind the tail kernel of
    // this step (SideSums below), which orders prepare_next behind the previous step as well.
    // (the head's gradients come out of the FT backward and its merge: nothing runs beside them, nobody needs ev_fwd)
    const bool head_grads_in_ft = side_head && head_in_ft && bt_internal_ft_fused_head_grad_ok(&f, q0.buckets);
    // The side stream's work for the NEXT batch (prepare_next) has to stay behind the last readers of the second buffer
    // set, the step before this one. Single-layer fused path: behind that step's completion event, which rode on its
    // merge kernel -- no packet on `s` at all. Otherwise behind a fork event recorded on `s` here.
    const bool fork_by_done = ready && head_grads_in_ft && (prev_done || n->capturing);
    if (n->capturing) {  // inside a graph the fork is an edge, not a packet: the side stream joins the capture
        BT_HIP(hipEventRecord(n->ev_fork, s));
        BT_HIP(hipStreamWaitEvent(n->side, n->ev_fork, 0));
    } else if (fork_by_done) {
        BT_HIP(hipStreamWaitEvent(n->side, prev_done, 0));
    } else if (ready && !tail) {
        BT_HIP(hipEventRecord(n->ev_fork, s));
    }
    if (int rc = forward_impl(n, s, b, fuse_loss && !tail, side_head, head_in_ft, true, perm.built && !ready,
                              tail ? 1 : -1, !head_grads_in_ft, ready && !joined ? n->ev_next : nullptr, joined))
        return rc;
    // The fused tail's gradient sums and layer 0's weight gradient are needed by the optimiser only: with the tail they
    // run on the side stream (behind the index build, if one is in flight there), beside dX and the FT backward,
    // and the compute stream joins them right before the gradient merge (or at the end of this function).
    struct SideSums {
        bool on;
        explicit SideSums(bt_net *n, bool use) : on(use) {
            if (on) bt_internal_dense_set_side(n->side, n->ev_dfork);
        }
        ~SideSums() {
            if (on) bt_internal_dense_set_side(nullptr, nullptr);
        }
    } side_sums(n, tail);
    if (tail) {
        if (int rc = bt_internal_dense_tail(s, &q1, &q_last, n->params + n->t_off[4], n->params + n->t_off[5],
                                            n->params + n->t_off[6], n->params + n->t_off[7], n->a[0], b->targets,
                                            n->d.use_entry_weights ? b->entry_weights : nullptr, n->a[2], n->ga[0],
                                            n->grads + n->t_off[4], n->grads + n->t_off[5], n->grads + n->t_off[6],
                                            n->grads + n->t_off[7], loss_sum_dev, n->dense_ws, n->dense_ws_bytes,
                                            tail_gz ? n->d.dense_act[0] : -1))
            return rc;
    } else if (!fuse_loss)
        if (int rc = bt_loss(s, n->d.loss_kind, b->batch, n->d.dense_out[L - 1], net_output(n), b->targets,
                             n->d.use_entry_weights ? b->entry_weights : nullptr, n->d.loss_power, loss_sum_dev,
                             n->grad_y))
            return rc;
    // gradient of the factoriser: sum of the table gradient over the repeats (after the FT backward wrote it)
    auto factor_reduce = [&]() -> int {
        if (n->d.ft_factoriser <= 0) return BT_OK;
        const size_t fc4 = n->t_size[0] / 4 / n->d.ft_factoriser;
        factor_reduce_kernel<<<dim3((unsigned)bt_div_up((long long)fc4, 256)), dim3(256), 0, s>>>(
            reinterpret_cast<const float4 *>(n->grads + n->t_off[0]),
            reinterpret_cast<float4 *>(n->grads + n->t_off[n->n_tensors - 1]), fc4, n->d.ft_factoriser);
        BT_LAUNCH_CHECK();
        return BT_OK;
    };
    if (head_grads_in_ft) {
        // The FT backward forms the head's weight gradient from the activation tile it stages anyway (matrix cores,
        // beside its LDS-bound reduction), and its merge kernel adds up the head's bias gradient and the loss from
        // the finishing kernel's partials: no second stream, no second pass over the [B][2H] activations.
        const bool hinted = n->has_next;
        bt_ft_head_grads hg{n->grads + n->t_off[2], n->grads + n->t_off[3], n->gz_parts, n->loss_partials,
                            n->n_loss_partials, loss_sum_dev};
        bt_ft_fused_adamw fa;
        if (opt && n->opt_kind == BT_OPT_ADAMW && n->d.ft_factoriser <= 0 && n->n_tensors == 4) {
            for (int t = 0; t < 4; t++) fa.t[t] = n->adam[t];
            fa.adj = opt->adj, fa.lr = opt->lr;
            hg.step = &fa, hg.done = opt->done, opt->stepped = true;
        }
        // A hinted next batch: its index is built on the side stream during this step, into the second buffer set
        // (prepare_next). The batch after it will be built during the next step, behind this step's completion event,
        // which rides on the merge kernel's packet.
        if (hinted && !hg.done) hg.done = n->ev_step;
        if (int rc = bt_internal_ft_backward_fused(s, &f, n->gz0, n->params + n->t_off[2],
                                                   n->d.dense_bucketed[0] ? b->buckets : nullptr, q0.buckets, n->hid,
                                                   n->saved, n->grads + n->t_off[0], n->grads + n->t_off[1], n->ft_ws,
                                                   n->ft_ws_bytes, &hg))
            return rc;
        if (int rc = factor_reduce()) return rc;
        if (hinted && n->d.ft_factoriser <= 0) n->prev_done = hg.done;  // (the merge is this step's last kernel)
        if (n->capturing) n->cap_ok = ready && hinted && fork_by_done && n->d.ft_factoriser <= 0;
        return prepare_next(n, fork_by_done);
    }
    if (side_head) {
        // (a next-batch hint is not used here: this path's index build already hides under the forward pass, and
        // building the next one beside the FT backward instead was measured slower, 0.30 vs 0.27 ms)
        n->has_next = false;
        if (!head_in_ft) BT_HIP(hipEventRecord(n->ev_fwd, s));  // (head_in_ft: recorded by head_finish itself)
        BT_HIP(hipStreamWaitEvent(n->side, n->ev_fwd, 0));  // side: after the index build (stream order) and the forward
        if (int rc = bt_internal_dense_backward_head(n->side, &q0, n->params + n->t_off[2], n->hid, b->buckets, n->a[0],
                                                     n->grad_y, n->grads + n->t_off[2], n->grads + n->t_off[3], nullptr,
                                                     nullptr, n->loss_partials, n->n_loss_partials, loss_sum_dev,
                                                     n->dense_ws, n->dense_ws_bytes, true))
            return rc;
        BT_HIP(hipEventRecord(n->ev_head, n->side));
        if (!head_in_ft && !ready) BT_HIP(hipStreamWaitEvent(s, n->ev_index, 0));  // (else: joined inside forward_impl)
        if (int rc = bt_internal_ft_backward_fused(s, &f, n->gz0, n->params + n->t_off[2],
                                                   n->d.dense_bucketed[0] ? b->buckets : nullptr, q0.buckets, n->hid,
                                                   n->saved, n->grads + n->t_off[0], n->grads + n->t_off[1], n->ft_ws,
                                                   n->ft_ws_bytes))
            return rc;
        if (int rc = factor_reduce()) return rc;
        BT_HIP(hipStreamWaitEvent(s, n->ev_head, 0));  // the step's gradients and loss are complete on `s`
        return BT_OK;
    }
    for (int k = tail ? 0 : L - 1; k >= 0; k--) {
        bt_dense_desc q = dn_desc(n, k, b->batch);
        const float *x = (k == 0) ? n->hid : (has_post(&n->d, k - 1) ? n->ap[k - 1] : n->a[k - 1]);
        const float *gy = (k == L - 1) ? n->grad_y : n->ga[k];
        if (k < L - 1 && has_post(&n->d, k)) {
            // ga[k] is the gradient w.r.t. what layer k+1 saw; turn it into the gradient w.r.t. layer k's raw outputs
            const int m = q.out - n->d.dense_skip[k];
            head_post_bwd_kernel<<<dim3((unsigned)bt_div_up((long long)b->batch * q.out, 256)), dim3(256), 0, s>>>(
                n->ga[k], n->ap[k], n->a[k], n->grad_y, n->gzp[k], b->batch, q.out, m, n->d.dense_dual[k],
                n->d.dense_act[k], n->d.dense_out[L - 1]);
            BT_LAUNCH_CHECK();
            gy = n->gzp[k];
        }
        float *gx = (k == 0) ? n->g_hid : n->ga[k - 1];
        float *dwk = n->grads + n->t_off[2 + 2 * k], *dbk = n->grads + n->t_off[3 + 2 * k];
        const float *wk = n->params + n->t_off[2 + 2 * k];
        if (bt_internal_dense_head_ok(&q)) {
            const bool ft_here = (k == 0) && fuse_ft;
            const bool loss_here = (k == L - 1) && fuse_loss;
            if (int rc = bt_internal_dense_backward_head(s, &q, wk, x, b->buckets, n->a[k], gy, dwk, dbk,
                                                         ft_here ? nullptr : gx, ft_here ? n->gz0 : nullptr,
                                                         loss_here ? n->loss_partials : nullptr, n->n_loss_partials,
                                                         loss_sum_dev, n->dense_ws, n->dense_ws_bytes, false))
                return rc;
        } else {
            bt_internal_dense_set_gz_ready(k == 0 && tail_gz);
            const int rc = bt_dense_backward(s, &q, wk, x, b->buckets, n->a[k], gy, dwk, dbk, gx, n->dense_ws,
                                             n->dense_ws_bytes);
            bt_internal_dense_set_gz_ready(false);
            if (rc) return rc;
        }
    }
    if (side_sums.on) {
        bt_internal_dense_set_side(nullptr, nullptr);
        BT_HIP(hipEventRecord(n->ev_sums, n->side));
    }
    if (!ready && !perm.built) BT_HIP(hipStreamWaitEvent(s, n->ev_index, 0));  // join (perm.built: joined before the layers)
    // bt_net_train_step: AdamW inside the FT gradient merge (table + bias) and, by extra blocks of that launch, for the
    // other tensors, whose gradients are complete by now
    bt_ft_step_all step_all;
    struct Disarm {  // whatever path leaves this function: no armed step (it points into this frame) survives it
        ~Disarm() { bt_internal_ft_arm_step(nullptr); }
    } disarm;
    if (opt && n->opt_kind == BT_OPT_ADAMW && n->d.ft_factoriser <= 0 && n->n_tensors - 2 <= BT_ADAMW_MAX_TENSORS &&
        bt_internal_ft_step_ok(&f)) {
        step_all.base.t[0] = n->adam[0], step_all.base.t[1] = n->adam[1];
        step_all.base.adj = opt->adj, step_all.base.lr = opt->lr;
        step_all.n_extra = n->n_tensors - 2;
        for (int t = 2; t < n->n_tensors; t++) step_all.extra[t - 2] = n->adam[t];
        step_all.done = opt->done;
        step_all.wait = side_sums.on ? n->ev_sums : nullptr;
        bt_internal_ft_arm_step(&step_all);
        opt->stepped = true;
    }
    const bool join_late = side_sums.on && !(opt && opt->stepped);  // (armed: the merge launch waits for ev_sums itself)
    if (fuse_ft) {
        if (int rc = bt_internal_ft_backward_fused(s, &f, n->gz0, n->params + n->t_off[2],
                                                   n->d.dense_bucketed[0] ? b->buckets : nullptr, q0.buckets, n->hid,
                                                   n->saved, n->grads + n->t_off[0], n->grads + n->t_off[1], n->ft_ws,
                                                   n->ft_ws_bytes))
            return rc;
    } else if (int rc = bt_ft_backward_prebuilt(s, &f, n->g_hid, n->hid, n->saved, n->grads + n->t_off[0],
                                                n->grads + n->t_off[1], n->ft_ws, n->ft_ws_bytes))
        return rc;
    if (int rc = factor_reduce()) return rc;
    if (join_late) BT_HIP(hipStreamWaitEvent(s, n->ev_sums, 0));  // every gradient and the loss sum are complete on `s`
    return prepare_next(n, ready && tail);
}
}  // namespace

int bt_net_use_graphs(bt_net *n, int on) {
    BT_REQUIRE(n, "bt_net_use_graphs: null net");
    n->use_graphs = on != 0;
    if (!on) bt_internal_net_drop_graphs(n);
    return BT_OK;
}
int bt_net_graph_launches(const bt_net *n) { return n ? n->graph_launches : 0; }

int bt_net_hint_next_batch(bt_net *n, const bt_batch *next, void *ready_event) {
    BT_REQUIRE(n, "bt_net_hint_next_batch: null net");
    n->has_next = false;
    if (!next) return BT_OK;
    if (int rc = check_batch(n, next, "bt_net_hint_next_batch")) return rc;
    // the second buffer set: on the first hint (allocates and waits for the zero fill). bt_train_run asks for it before
    // its loop starts (bt_internal_net_enable_hints), so that no step of a training run allocates.
    if (int rc = bt_internal_net_enable_hints(n)) return rc;
    n->next = *next, n->next_ready = reinterpret_cast<hipEvent_t>(ready_event), n->has_next = true;
    return BT_OK;
}

// save_to_checkpoint, crates/bullet_lib/src/value/save.rs:24-43 (directories are created if missing). As in the
// reference (save.rs:34-42) a failing raw.bin / quantised.bin writer -- typically "Failed quantisation from f32 to
// i16!" on early or unclipped weights -- is printed and does NOT fail the checkpoint: training goes on and
// optimiser_state/ is complete. bt_last_error() keeps the writer's message. Failures of optimiser_state/ are fatal.
int bt_net_save_checkpoint(bt_net *n, void *stream, const bt_saved_format *fmt, int n_fmt, const char *dir) {
    BT_REQUIRE(n && dir, "bt_net_save_checkpoint: null argument");
    hipStream_t s = bt_stream(stream);
    std::vector<float> host(n->n_params);
    const std::string base(dir), opt = base + "/optimiser_state";
    (void)mkdir(base.c_str(), 0777);
    (void)mkdir(opt.c_str(), 0777);
    const struct {
        const float *dev;
        const char *file;
    } parts[3] = {{n->params, "/weights.bin"}, {n->mom, "/momentum.bin"}, {n->vel, "/velocity.bin"}};
    for (int i = 2; i >= 0; i--) {  // weights last: `host` then holds them for raw.bin / quantised.bin
        BT_HIP(hipMemcpyAsync(host.data(), parts[i].dev, n->n_params * sizeof(float), hipMemcpyDeviceToHost, s));
        BT_HIP(hipStreamSynchronize(s));
        if (int rc = bt_checkpoint_write(&n->d, host.data(), (opt + parts[i].file).c_str())) return rc;
    }
    if (n->opt_kind == BT_OPT_RANGER) {  // RangerLookahead::write_to_checkpoint, ranger.rs:124-129
        std::vector<float> slow(n->n_params);
        BT_HIP(hipMemcpyAsync(slow.data(), n->slow, n->n_params * sizeof(float), hipMemcpyDeviceToHost, s));
        BT_HIP(hipStreamSynchronize(s));
        if (int rc = bt_checkpoint_write(&n->d, slow.data(), (opt + "/slow.bin").c_str())) return rc;
    }
    if (bt_save_raw(&n->d, host.data(), fmt, n_fmt, (base + "/raw.bin").c_str()))
        std::printf("Failed to write raw network weights:\n%s\n", bt_last_error());  // save.rs:34-36
    if (bt_save_quantised(&n->d, host.data(), fmt, n_fmt, (base + "/quantised.bin").c_str()))
        std::printf("Failed to write quantised network weights:\n%s\n", bt_last_error());  // save.rs:38-42
    return BT_OK;
}

// Optimiser::load_from_checkpoint, crates/trainer/src/optimiser.rs:232-236 + adam.rs:309-324
int bt_net_load_checkpoint(bt_net *n, void *stream, const char *dir) {
    BT_REQUIRE(n && dir, "bt_net_load_checkpoint: null argument");
    hipStream_t s = bt_stream(stream);
    std::vector<float> host(n->n_params);
    const std::string opt = std::string(dir) + "/optimiser_state";
    const struct {
        float *dev;
        const char *file;
    } parts[3] = {{n->params, "/weights.bin"}, {n->mom, "/momentum.bin"}, {n->vel, "/velocity.bin"}};
    for (int i = 0; i < 3; i++) {
        std::fill(host.begin(), host.end(), 0.0f);
        if (int rc = bt_checkpoint_read(&n->d, host.data(), (opt + parts[i].file).c_str())) return rc;
        BT_HIP(hipMemcpyAsync(parts[i].dev, host.data(), n->n_params * sizeof(float), hipMemcpyHostToDevice, s));
        BT_HIP(hipStreamSynchronize(s));
    }
    if (n->opt_kind == BT_OPT_RANGER) {  // ranger.rs:112-122
        std::fill(host.begin(), host.end(), 0.0f);
        if (int rc = bt_checkpoint_read(&n->d, host.data(), (opt + "/slow.bin").c_str())) return rc;
        BT_HIP(hipMemcpyAsync(n->slow, host.data(), n->n_params * sizeof(float), hipMemcpyHostToDevice, s));
        BT_HIP(hipStreamSynchronize(s));
    }
    return BT_OK;
}

// the element-wise update of `nt` tensors (AdamW, or RAdam at the current step count)
static int opt_apply(bt_net *n, void *stream, int nt, const bt_adamw_tensor *tab, const float *adj_dev, const float *lr_dev) {
    if (n->opt_kind == BT_OPT_ADAMW) return bt_adamw(stream, nt, tab, adj_dev, lr_dev);
    return bt_radam(stream, nt, tab, adj_dev, lr_dev, n->opt_step, n->opt_thr);
}
static int opt_lookahead(bt_net *n, void *stream) {
    if (n->opt_kind != BT_OPT_RANGER || n->opt_step % n->opt_k != 0) return BT_OK;
    float *pp[1] = {n->params}, *ss[1] = {n->slow};
    const size_t cnt[1] = {n->n_params};  // the flat buffers: padding floats are zero on both sides
    return bt_lookahead(stream, 1, pp, ss, cnt, n->opt_alpha);
}

int bt_net_optimiser_step(bt_net *n, void *stream, const float *adj_dev, const float *lr_dev) {
    return bt_net_optimiser_step_event(n, stream, adj_dev, lr_dev, nullptr);
}

int bt_net_optimiser_step_event(bt_net *n, void *stream, const float *adj_dev, const float *lr_dev, void *done_event) {
    BT_REQUIRE(n, "bt_net_optimiser_step: null net");
    if (n->opt_kind != BT_OPT_ADAMW) n->opt_step++;  // RAdam::update / RangerLookahead::update count their calls (radam.rs:277, ranger.rs:88)
    hipEvent_t done = static_cast<hipEvent_t>(done_event);
    const bool lookahead = n->opt_kind == BT_OPT_RANGER && n->opt_step % n->opt_k == 0;
    // the event rides on the update kernel's own dispatch packet: a separate record costs the stream a few microseconds
    if (done && !lookahead) bt_attach_stop_event(BTK_ADAMW, done);
    if (int rc = opt_apply(n, stream, n->n_tensors, n->adam, adj_dev, lr_dev)) {
        if (done && !lookahead) (void)bt_take_stop_event(BTK_ADAMW);  // not launched: do not leave the event armed
        return rc;
    }
    if (int rc = opt_lookahead(n, stream)) return rc;
    if (done && lookahead) BT_HIP(hipEventRecord(done, bt_stream(stream)));
    return BT_OK;
}

// ---- data-parallel step (section 7 of the header; no counterpart in the reference)
int bt_net_dp_plan(const bt_net_desc *desc, int world, int rank, size_t shard_min_floats, bt_dp_plan *plan) {
    BT_REQUIRE(desc && plan && world >= 1 && rank >= 0 && rank < world, "bt_net_dp_plan: bad arguments");
    const size_t flat = bt_net_param_count(desc);
    BT_REQUIRE(flat > 0, "bt_net_dp_plan: invalid net description");
    const size_t table = bt_net_tensor_size(desc, 0), min_floats = shard_min_floats ? shard_min_floats : BT_DP_SHARD_MIN_FLOATS;
    *plan = bt_dp_plan{};
    plan->tail_off = 0, plan->tail_n = flat + 1;
    if (bt_net_num_tensors(desc) > 1 && table >= min_floats && table % (size_t)world == 0) {  // (world 1: the same path, a test aid)
        plan->sharded = 1;
        plan->table_n = table, plan->per_rank = table / (size_t)world, plan->slice_lo = plan->per_rank * (size_t)rank;
        plan->tail_off = bt_net_tensor_offset(desc, 1);  // tensor 0 sits at offset 0 of the flat buffers
        plan->tail_n = flat + 1 - plan->tail_off;
    }
    return BT_OK;
}

int bt_net_dp_step(bt_net *n, bt_comm *c, void *stream, const float *adj_dev, const float *lr_dev, size_t shard_min_floats) {
    BT_REQUIRE(n && adj_dev && lr_dev, "bt_net_dp_step: null argument");
    if (!c) return bt_net_optimiser_step(n, stream, adj_dev, lr_dev);
    bt_dp_plan pl;
    if (int rc = bt_net_dp_plan(&n->d, c->world, c->rank, shard_min_floats, &pl)) return rc;
    if (pl.sharded && c->hooked && !c->hooks.reduce_scatter) {  // hooks without the pair: replicated step
        pl.sharded = 0, pl.tail_off = 0, pl.tail_n = n->n_params + 1;
    }
    hipStream_t s = bt_stream(stream);
    // collectives on the communicator's stream when it has one (RCCL), fenced by events; else in `stream` order
    hipStream_t cs = c->stream ? c->stream : s;
    const bool fenced = cs != s;
    auto fence = [&](hipStream_t from, hipStream_t to, hipEvent_t ev) -> int {
        if (!fenced) return BT_OK;
        BT_HIP(hipEventRecord(ev, from));
        BT_HIP(hipStreamWaitEvent(to, ev, 0));
        return BT_OK;
    };
    if (!pl.sharded) {
        if (int rc = bt_allreduce_sum_f32(c, stream, n->grads, pl.tail_n)) return rc;
        return bt_net_optimiser_step(n, stream, adj_dev, lr_dev);
    }
    if (n->opt_kind != BT_OPT_ADAMW) n->opt_step++;
    if (int rc = fence(s, cs, c->ev[0])) return rc;  // the backward is done
    if (int rc = bt_allreduce_sum_f32(c, cs, n->grads + pl.tail_off, pl.tail_n)) return rc;
    if (int rc = fence(cs, s, c->ev[1])) return rc;
    // (in place: rank r's sum lands in its own slice of the gradient)
    if (int rc = bt_reduce_scatter_sum_f32(c, cs, n->grads, n->grads + pl.slice_lo, pl.per_rank)) return rc;
    // every tensor but the table, on every rank, while the table's gradient is still on the wire
    if (int rc = opt_apply(n, stream, n->n_tensors - 1, n->adam + 1, adj_dev, lr_dev)) return rc;
    if (int rc = fence(cs, s, c->ev[2])) return rc;
    bt_adamw_tensor own = n->adam[0];  // this rank's slice of the table
    own.grad += pl.slice_lo, own.param += pl.slice_lo, own.momentum += pl.slice_lo, own.velocity += pl.slice_lo;
    own.n = pl.per_rank;
    if (int rc = opt_apply(n, stream, 1, &own, adj_dev, lr_dev)) return rc;
    if (int rc = fence(s, cs, c->ev[3])) return rc;
    if (int rc = bt_allgather_f32(c, cs, n->params + pl.slice_lo, n->params, pl.per_rank)) return rc;
    if (int rc = fence(cs, s, c->ev[0])) return rc;
    return opt_lookahead(n, stream);
}

int bt_net_dp_gather_state(bt_net *n, bt_comm *c, void *stream, size_t shard_min_floats) {
    BT_REQUIRE(n, "bt_net_dp_gather_state: null net");
    if (!c) return BT_OK;
    bt_dp_plan pl;
    if (int rc = bt_net_dp_plan(&n->d, c->world, c->rank, shard_min_floats, &pl)) return rc;
    if (!pl.sharded || (c->hooked && !c->hooks.all_gather)) return BT_OK;
    if (int rc = bt_allgather_f32(c, stream, n->mom + pl.slice_lo, n->mom, pl.per_rank)) return rc;
    return bt_allgather_f32(c, stream, n->vel + pl.slice_lo, n->vel, pl.per_rank);
}

int bt_net_set_optimiser(bt_net *n, int kind, float n_sma_threshold, float alpha, int k) {
    BT_REQUIRE(n, "bt_net_set_optimiser: null net");
    BT_REQUIRE(kind >= BT_OPT_ADAMW && kind <= BT_OPT_RANGER, "bt_net_set_optimiser: unknown optimiser %d", kind);
    BT_REQUIRE(kind != BT_OPT_RANGER || k >= 1, "bt_net_set_optimiser: k must be >= 1");
    bt_internal_net_drop_graphs(n);
    n->opt_kind = kind, n->opt_step = 0, n->opt_thr = n_sma_threshold, n->opt_alpha = alpha, n->opt_k = k > 0 ? k : 6;
    BT_HIP(hipMemset(n->mom, 0, n->n_params * sizeof(float)));  // OptimiserState::reset
    BT_HIP(hipMemset(n->vel, 0, n->n_params * sizeof(float)));
    if (kind == BT_OPT_RANGER) {
        if (!n->slow)
            if (int rc = dmalloc(&n->slow, n->n_params)) return rc;
        BT_HIP(hipMemset(n->slow, 0, n->n_params * sizeof(float)));  // ranger.rs:71: slow weights start at zero
    }
    BT_HIP(hipStreamSynchronize(nullptr));  // the fills ran on the NULL stream: done before any (non-blocking) stream steps
    return BT_OK;
}

}  // extern "C"
