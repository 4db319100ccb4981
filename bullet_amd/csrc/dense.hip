// Small dense layers with output-bucket select, sigmoid-MSE loss and multi-tensor AdamW.
//
// Reference: Matmul (crates/compiler/src/tensor/operation/linear/matmul.rs:62-126) lowered to
// hipblasSgemm (crates/gpu/src/function.rs:256-274), Select/SelectPad (index/select.rs:12-26,93-110),
// bias broadcast/reduce (index/broadcast.rs, linear/reduce.rs:96-120), activations (dfo.rs), loss
// (model/operations/pointwise.rs:141-165, model/builder.rs:351-354), AdamW (trainer optimiser/adam.rs:35-62).
//
// The NNUE head layers have <= 32 outputs per bucket and are bandwidth-bound on the activation
// stream, so they are written as fused dot-product kernels: only the selected bucket's rows are ever
// computed (the reference computes all buckets with a GEMM, then discards), and the SelectPad zeros of
// the backward are never materialised.  Summation orders are fixed => bit-reproducible.
#include "common.h"

namespace {

constexpr int DENSE_OT = 8;      // outputs per pass in the forward kernel
constexpr int DW_ITILE = 64;     // input columns per workgroup in the dW kernel
constexpr int DW_RANGE = 512;    // positions per workgroup in the dW kernel

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per position: lanes stride over the inputs, DENSE_OT outputs at a time.
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float *__restrict__ w, const float *__restrict__ bias,
                                                        const float *__restrict__ x,
                                                        const int32_t *__restrict__ bucket, float *__restrict__ y,
                                                        bt_dense_desc d) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= d.batch) return;
    const int rows = d.out * d.buckets;
    const int base = (d.buckets > 1) ? bucket[b] * d.out : 0;
    const float *xb = x + (size_t)b * d.in;
    for (int o0 = 0; o0 < d.out; o0 += DENSE_OT) {
        float acc[DENSE_OT];
#pragma unroll
        for (int q = 0; q < DENSE_OT; q++) acc[q] = 0.0f;
        const int no = min(DENSE_OT, d.out - o0);
        for (int i = lane; i < d.in; i += 64) {
            const float xv = xb[i];
            const float *wc = w + (size_t)rows * i + base + o0;
#pragma unroll
            for (int q = 0; q < DENSE_OT; q++)
                if (q < no) acc[q] += wc[q] * xv;
        }
#pragma unroll
        for (int q = 0; q < DENSE_OT; q++) {
            float s = wave_sum(acc[q]);
            if (q < no && lane == 0) y[(size_t)b * d.out + o0 + q] = bt_act_fwd(d.act, s + bias[base + o0 + q]);
        }
    }
}

// gz = grad_y * f'(y); grad_x[b][i] = sum_o W[base+o][i] * gz[b][o]   (thread per (b, i))
__global__ __launch_bounds__(256) void dense_bwd_dx_kernel(const float *__restrict__ w,
                                                           const int32_t *__restrict__ bucket,
                                                           const float *__restrict__ y,
                                                           const float *__restrict__ grad_y,
                                                           float *__restrict__ grad_x, bt_dense_desc d) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)d.batch * d.in) return;
    const int b = (int)(t / d.in), i = (int)(t % d.in);
    const int rows = d.out * d.buckets;
    const int base = (d.buckets > 1) ? bucket[b] * d.out : 0;
    const float *wc = w + (size_t)rows * i + base;
    float s = 0.0f;
    for (int o = 0; o < d.out; o++) {
        const size_t yo = (size_t)b * d.out + o;
        s += wc[o] * (grad_y[yo] * bt_act_deriv(d.act, y[yo]));
    }
    grad_x[t] = s;
}

// dW partials: workgroup = (input-column tile, position range); thread (il, og) owns LDS accumulators
// acc[row][il] for rows == og (mod 4): exclusive ownership => no races, positions visited in order.
__global__ __launch_bounds__(256) void dense_bwd_dw_kernel(const float *__restrict__ x,
                                                           const int32_t *__restrict__ bucket,
                                                           const float *__restrict__ y,
                                                           const float *__restrict__ grad_y,
                                                           float *__restrict__ part_w, float *__restrict__ part_b,
                                                           bt_dense_desc d) {
    extern __shared__ float acc[];  // [rows][DW_ITILE] (+ [rows] bias accumulators)
    const int rows = d.out * d.buckets;
    float *accb = acc + (size_t)rows * DW_ITILE;
    const int il = threadIdx.x & (DW_ITILE - 1);
    const int og = threadIdx.x / DW_ITILE;  // 0..3
    const int i = blockIdx.x * DW_ITILE + il;
    const int b0 = blockIdx.y * DW_RANGE, b1 = min(d.batch, b0 + DW_RANGE);
    for (int q = threadIdx.x; q < rows * DW_ITILE + rows; q += blockDim.x) acc[q] = 0.0f;
    __syncthreads();
    const bool do_bias = (blockIdx.x == 0) && (il == 0);
    for (int b = b0; b < b1; b++) {
        const int base = (d.buckets > 1) ? bucket[b] * d.out : 0;
        const float xv = (i < d.in) ? x[(size_t)b * d.in + i] : 0.0f;
        for (int o = og; o < d.out; o += 4) {
            const size_t yo = (size_t)b * d.out + o;
            const float gz = grad_y[yo] * bt_act_deriv(d.act, y[yo]);
            acc[(size_t)(base + o) * DW_ITILE + il] += gz * xv;
            if (do_bias) accb[base + o] += gz;
        }
    }
    __syncthreads();
    // partial layout: [range][rows][in] in the same column-major order as W: (row + rows*i)
    float *pw = part_w + (size_t)blockIdx.y * rows * d.in;
    if (i < d.in)
        for (int r = og; r < rows; r += 4) pw[(size_t)rows * i + r] = acc[(size_t)r * DW_ITILE + il];
    if (blockIdx.x == 0)
        for (int r = threadIdx.x; r < rows; r += blockDim.x) part_b[(size_t)blockIdx.y * rows + r] = accb[r];
}

__global__ void sum_partials_kernel(const float *__restrict__ part, int nparts, size_t n, float *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float s = 0.0f;
    for (int p = 0; p < nparts; p++) s += part[(size_t)p * n + t];
    out[t] = s;
}

// ------------------------------------------------------------------ loss
// single workgroup: fixed strided partial sums + fixed tree => deterministic loss
__global__ __launch_bounds__(1024) void sigmoid_mse_kernel(int batch, const float *__restrict__ y,
                                                           const float *__restrict__ targets,
                                                           const float *__restrict__ ew, float *__restrict__ loss_sum,
                                                           float *__restrict__ grad_y) {
    __shared__ float red[1024];
    float part = 0.0f;
    for (int b = threadIdx.x; b < batch; b += blockDim.x) {
        const float s = 1.0f / (1.0f + expf(-1.0f * y[b]));  // Reciprocal(1 + Exp(-1*y)), pointwise.rs:156-164
        const float diff = s + (-1.0f * targets[b]);
        float l = diff * diff;
        const float g = ew ? ew[b] : 1.0f;
        if (ew) l = g * l;
        part += l;
        const float dd = g * diff + g * diff;                    // Mul backward, binary.rs:101-123
        grad_y[b] = dd * (s * (1.0f + (-1.0f * s)));              // sigmoid' from output, dfo.rs:114-118
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_sum[0] = red[0];
}

// ------------------------------------------------------------------ AdamW (multi-tensor, one launch)
struct AdamTable {
    int n_tensors;
    int block_start[BT_ADAMW_MAX_TENSORS + 1];  // prefix sum of blocks per tensor
    bt_adamw_tensor t[BT_ADAMW_MAX_TENSORS];
};
constexpr int ADAM_ELEMS_PER_BLOCK = 1024;  // 256 threads x float4

__device__ __forceinline__ void adam_op(float grad, float rate, float &p, float &m, float &v,
                                        const bt_adamw_tensor &t) {
    // adam.rs:43-51, in this exact order
    p *= 1.0f - t.decay * rate;
    m = t.beta1 * m + (1.0f - t.beta1) * grad;
    v = t.beta2 * v + (1.0f - t.beta2) * grad * grad;
    const float val = m / (__fsqrt_rn(v) + 0.00000001f);
    p -= rate * val;
    p = fminf(fmaxf(p, t.min_weight), t.max_weight);
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamTable tab, const float *__restrict__ adj_ptr,
                                                    const float *__restrict__ rate_ptr) {
    int ti = 0;
    while (ti + 1 < tab.n_tensors && (int)blockIdx.x >= tab.block_start[ti + 1]) ti++;
    const bt_adamw_tensor &t = tab.t[ti];
    const float adj = adj_ptr[0], rate = rate_ptr[0];
    const size_t e0 = ((size_t)(blockIdx.x - tab.block_start[ti]) * 256 + threadIdx.x) * 4;
    if (e0 >= t.n) return;
    if (e0 + 4 <= t.n && (t.n % 4 == 0)) {
        float4 p = *reinterpret_cast<float4 *>(t.param + e0);
        float4 m = *reinterpret_cast<float4 *>(t.momentum + e0);
        float4 v = *reinterpret_cast<float4 *>(t.velocity + e0);
        const float4 g = *reinterpret_cast<const float4 *>(t.grad + e0);
        adam_op(adj * g.x, rate, p.x, m.x, v.x, t);
        adam_op(adj * g.y, rate, p.y, m.y, v.y, t);
        adam_op(adj * g.z, rate, p.z, m.z, v.z, t);
        adam_op(adj * g.w, rate, p.w, m.w, v.w, t);
        *reinterpret_cast<float4 *>(t.param + e0) = p;
        *reinterpret_cast<float4 *>(t.momentum + e0) = m;
        *reinterpret_cast<float4 *>(t.velocity + e0) = v;
    } else {
        for (size_t e = e0; e < t.n && e < e0 + 4; e++) {
            float p = t.param[e], m = t.momentum[e], v = t.velocity[e];
            adam_op(adj * t.grad[e], rate, p, m, v, t);
            t.param[e] = p, t.momentum[e] = m, t.velocity[e] = v;
        }
    }
}

int dense_validate(const bt_dense_desc *d, const char *who) {
    BT_REQUIRE(d, "%s: null descriptor", who);
    BT_REQUIRE(d->batch > 0 && d->in > 0 && d->out > 0 && d->buckets > 0, "%s: Mismatched sizes! (non-positive dim)", who);
    BT_REQUIRE(d->act >= BT_ACT_IDENTITY && d->act <= BT_ACT_SQRRELU, "%s: unknown activation", who);
    BT_REQUIRE((size_t)d->out * d->buckets * (DW_ITILE + 1) * sizeof(float) <= 150 * 1024,
               "%s: out*buckets=%d too large for the fused head-layer kernels", who, d->out * d->buckets);
    return BT_OK;
}

}  // namespace

extern "C" {

int bt_dense_forward(void *stream, const bt_dense_desc *d, const float *w, const float *bias, const float *x,
                     const int32_t *bucket, float *y) {
    if (int rc = dense_validate(d, "bt_dense_forward")) return rc;
    BT_REQUIRE(w && bias && x && y, "bt_dense_forward: null pointer");
    BT_REQUIRE(d->buckets == 1 || bucket, "bt_dense_forward: bucketed layer needs bucket indices");
    BT_LAUNCH(BTK_DENSE_FWD, bt_stream(stream), dense_fwd_kernel, dim3(bt_div_up(d->batch, 4)), dim3(256), 0, w, bias, x,
                       bucket, y, *d);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

size_t bt_dense_backward_workspace_bytes(const bt_dense_desc *d) {
    if (dense_validate(d, "bt_dense_backward_workspace_bytes")) return 0;
    const int rows = d->out * d->buckets;
    const int nr = bt_div_up(d->batch, DW_RANGE);
    return bt_align((size_t)nr * rows * d->in * sizeof(float)) + bt_align((size_t)nr * rows * sizeof(float));
}

int bt_dense_backward(void *stream, const bt_dense_desc *d, const float *w, const float *x, const int32_t *bucket,
                      const float *y, const float *grad_y, float *dw, float *dbias, float *grad_x, void *workspace,
                      size_t workspace_bytes) {
    if (int rc = dense_validate(d, "bt_dense_backward")) return rc;
    BT_REQUIRE(w && x && y && grad_y && dw && dbias && workspace, "bt_dense_backward: null pointer");
    BT_REQUIRE(d->buckets == 1 || bucket, "bt_dense_backward: bucketed layer needs bucket indices");
    BT_REQUIRE(workspace_bytes >= bt_dense_backward_workspace_bytes(d), "bt_dense_backward: workspace too small");
    hipStream_t s = bt_stream(stream);
    const int rows = d->out * d->buckets;
    const int nr = bt_div_up(d->batch, DW_RANGE);
    float *part_w = static_cast<float *>(workspace);
    float *part_b = reinterpret_cast<float *>(static_cast<char *>(workspace) +
                                              bt_align((size_t)nr * rows * d->in * sizeof(float)));
    if (grad_x) {
        BT_LAUNCH(BTK_DENSE_BWD_DX, s, dense_bwd_dx_kernel, dim3(bt_div_up((long long)d->batch * d->in, 256)), dim3(256), 0, w,
                           bucket, y, grad_y, grad_x, *d);
        BT_LAUNCH_CHECK();
    }
    const size_t lds = ((size_t)rows * DW_ITILE + rows) * sizeof(float);
    BT_LAUNCH(BTK_DENSE_BWD_DW, s, dense_bwd_dw_kernel, dim3(bt_div_up(d->in, DW_ITILE), nr), dim3(256), lds, x, bucket, y,
                       grad_y, part_w, part_b, *d);
    BT_LAUNCH_CHECK();
    const size_t nw = (size_t)rows * d->in;
    BT_LAUNCH(BTK_SUM_PARTIALS, s, sum_partials_kernel, dim3(bt_div_up(nw, 256)), dim3(256), 0, part_w, nr, nw, dw);
    BT_LAUNCH_CHECK();
    BT_LAUNCH(BTK_SUM_PARTIALS, s, sum_partials_kernel, dim3(bt_div_up(rows, 256)), dim3(256), 0, part_b, nr, (size_t)rows,
                       dbias);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

int bt_sigmoid_mse(void *stream, int batch, const float *y, const float *targets, const float *entry_weights,
                   float *loss_sum, float *grad_y) {
    BT_REQUIRE(batch > 0, "bt_sigmoid_mse: Mismatched sizes! (batch <= 0)");
    BT_REQUIRE(y && targets && loss_sum && grad_y, "bt_sigmoid_mse: null pointer");
    BT_LAUNCH(BTK_SIGMOID_MSE, bt_stream(stream), sigmoid_mse_kernel, dim3(1), dim3(1024), 0, batch, y, targets,
                       entry_weights, loss_sum, grad_y);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

int bt_adamw(void *stream, int n_tensors, const bt_adamw_tensor *tensors, const float *adj_dev, const float *lr_dev) {
    BT_REQUIRE(n_tensors > 0 && n_tensors <= BT_ADAMW_MAX_TENSORS, "bt_adamw: n_tensors out of range");
    BT_REQUIRE(tensors && adj_dev && lr_dev, "bt_adamw: null pointer");
    AdamTable tab;
    tab.n_tensors = n_tensors;
    int blocks = 0;
    for (int i = 0; i < n_tensors; i++) {
        const bt_adamw_tensor &t = tensors[i];
        BT_REQUIRE(t.grad && t.param && t.momentum && t.velocity && t.n > 0, "bt_adamw: tensor %d has null/empty buffers", i);
        // OptimiserState::new, adam.rs:259-263
        BT_REQUIRE(t.max_weight >= t.min_weight, "Invalid clipping: %g >= %g", t.min_weight, t.max_weight);
        tab.block_start[i] = blocks;
        blocks += bt_div_up((long long)t.n, ADAM_ELEMS_PER_BLOCK);
        tab.t[i] = t;
    }
    tab.block_start[n_tensors] = blocks;
    BT_LAUNCH(BTK_ADAMW, bt_stream(stream), adamw_kernel, dim3(blocks), dim3(256), 0, tab, adj_dev, lr_dev);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

}  // extern "C"
