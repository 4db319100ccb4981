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
#include <mutex>

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

// out[t] = sum_p part[p][t] in a fixed order: 32 elements x 8 partial-slices per workgroup; each slice
// sums its contiguous share of the partials sequentially, the 8 slice sums are then added in order.
__global__ __launch_bounds__(256) void sum_partials_kernel(const float *__restrict__ part, int nparts, size_t n,
                                                           float *__restrict__ out) {
    __shared__ float red[8][32];
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const size_t t = (size_t)blockIdx.x * 32 + el;
    const int per = (nparts + 7) / 8;
    const int p0 = sl * per, p1 = min(nparts, p0 + per);
    float s = 0.0f;
    if (t < n)
        for (int p = p0; p < p1; p++) s += part[(size_t)p * n + t];
    red[sl][el] = s;
    __syncthreads();
    if (sl == 0 && t < n) {
        float r = red[0][el];
#pragma unroll
        for (int i = 1; i < 8; i++) r += red[i][el];
        out[t] = r;
    }
}

// same sums in the same order, four adjacent elements per thread (16-B loads; n % 4 == 0, 16-B aligned buffers)
__global__ __launch_bounds__(256) void sum_partials4_kernel(const float *__restrict__ part, int nparts, size_t n4,
                                                            float *__restrict__ out) {
    __shared__ float4 red[8][32];
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const size_t t = (size_t)blockIdx.x * 32 + el;
    const int per = (nparts + 7) / 8;
    const int p0 = sl * per, p1 = min(nparts, p0 + per);
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (t < n4) {
        const float4 *src = reinterpret_cast<const float4 *>(part) + t;
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(size_t)(p + u) * n4];
#pragma unroll
            for (int u = 0; u < 8; u++) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
        }
        for (; p < p1; p++) {
            const float4 v = src[(size_t)p * n4];
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
    }
    red[sl][el] = s;
    __syncthreads();
    if (sl == 0 && t < n4) {
        float4 r = red[0][el];
#pragma unroll
        for (int i = 1; i < 8; i++) r.x += red[i][el].x, r.y += red[i][el].y, r.z += red[i][el].z, r.w += red[i][el].w;
        reinterpret_cast<float4 *>(out)[t] = r;
    }
}

// LDS-free shape for launches that must fit next to a resident FT-backward workgroup: one thread sums one float4
// column over all partials, in order
__global__ __launch_bounds__(64) void sum_partials_seq4_kernel(const float *__restrict__ part, int nparts, size_t n4,
                                                               float *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= n4) return;
    const float4 *src = reinterpret_cast<const float4 *>(part) + t;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int p = 0;
    for (; p + 8 <= nparts; p += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[(size_t)(p + u) * n4];
#pragma unroll
        for (int u = 0; u < 8; u++) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
    }
    for (; p < nparts; p++) {
        const float4 v = src[(size_t)p * n4];
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    reinterpret_cast<float4 *>(out)[t] = s;
}

int sum_partials_launch(hipStream_t s, const float *part, int nparts, size_t n, float *out, bool light = false) {
    if (light && n % 4 == 0 && ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        BT_LAUNCH(BTK_SUM_PARTIALS, s, sum_partials_seq4_kernel, dim3(bt_div_up(n / 4, 64)), dim3(64), 0, part, nparts, n / 4,
                  out);
        BT_LAUNCH_CHECK();
        return BT_OK;
    }
    const bool v4 = n % 4 == 0 && ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (v4)
        BT_LAUNCH(BTK_SUM_PARTIALS, s, sum_partials4_kernel, dim3(bt_div_up(n / 4, 32)), dim3(256), 0, part, nparts, n / 4, out);
    else
        BT_LAUNCH(BTK_SUM_PARTIALS, s, sum_partials_kernel, dim3(bt_div_up(n, 32)), dim3(256), 0, part, nparts, n, out);
    BT_LAUNCH_CHECK();
    return BT_OK;
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

// the other loss_fn closures (bt_loss): same single-workgroup deterministic reduction
__global__ __launch_bounds__(1024) void loss_general_kernel(int kind, int batch, int width, const float *__restrict__ y,
                                                            const float *__restrict__ targets,
                                                            const float *__restrict__ ew, float power,
                                                            float *__restrict__ loss_sum, float *__restrict__ grad_y) {
    __shared__ float red[1024];
    float part = 0.0f;
    for (int b = threadIdx.x; b < batch; b += blockDim.x) {
        const float g = ew ? ew[b] : 1.0f;  // gradient arriving at the per-position loss (reduce_sum_batch, then `ew *`)
        if (kind == BT_LOSS_SIGMOID_POWER) {
            const float s = 1.0f / (1.0f + expf(-1.0f * y[b]));
            const float diff = s + (-1.0f * targets[b]);
            const float a = fabsf(diff);
            float l = powf(a, power);  // AbsPower, pointwise.rs:265-268
            if (ew) l = g * l;
            part += l;
            // Power::backward (binary.rs:167-172): grad * (p * a^(p + -1)); Abs' = sgn (0 at 0, write.rs:172-174)
            const float sg = diff == 0.0f ? 0.0f : (diff > 0.0f ? 1.0f : -1.0f);
            const float da = g * (power * powf(a, power + -1.0f));
            grad_y[b] = (da * sg) * (s * (1.0f + (-1.0f * s)));
        } else {  // BT_LOSS_SOFTMAX_CE: tensor/builder.rs:159-170, autograd/softmax.rs:27-37
            const float *yb = y + (size_t)b * width, *tb = targets + (size_t)b * width;
            float mx = yb[0];
            for (int k = 1; k < width; k++) mx = fmaxf(mx, yb[k]);
            float den = 0.0f;
            for (int k = 0; k < width; k++) den += expf(yb[k] + (-1.0f * mx));
            const float inv = 1.0f / den;
            float l = 0.0f;
            for (int k = 0; k < width; k++) {
                const float p = expf(yb[k] + (-1.0f * mx)) * inv;
                float lk = -1.0f * (tb[k] * logf(p));
                if (ew) lk = g * lk;
                l += lk;
                grad_y[(size_t)b * width + k] = (p + (-1.0f * tb[k])) * g;
            }
            part += l;
        }
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_sum[0] = red[0];
}

__global__ __launch_bounds__(256) void weight_clip_kernel(float *__restrict__ w, size_t n, float lo, float hi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = fmaxf(fminf(w[i], hi), lo);  // clip.rs:25-28: min with max first, then max with min
}

// ModelNode::clip_pass_through_grad (compiler/src/model/operations/pointwise.rs:272-300): x.max(min).min(max); its
// gradient is the incoming one, untouched (tensor/operation/autograd/passthrough.rs:31-33)
__global__ __launch_bounds__(256) void clip_pass_through_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                size_t n, float lo, float hi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fminf(fmaxf(x[i], lo), hi);
}

// ModelNode::faux_quantise (pointwise.rs:196-221): (value * x).round() or .trunc(), times the reciprocal of value;
// roundf / truncf as the reference's generated code calls them (gpu/src/pointwise/write.rs:196-197)
__global__ __launch_bounds__(256) void faux_quantise_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n,
                                                            float value, int round) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float denom = 1.0f / value, scaled = value * x[i];
        y[i] = (round ? roundf(scaled) : truncf(scaled)) * denom;
    }
}

__global__ __launch_bounds__(256) void weight_decay_kernel(float *__restrict__ w, size_t n, float decay,
                                                           const float *__restrict__ lr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = w[i] * (1.0f + (-1.0f * (lr[0] * decay)));  // decay.rs:32-35
}

// ------------------------------------------------------------------ AdamW (multi-tensor, one launch)
struct AdamTable {
    int n_tensors;
    int radam;        // 0: AdamW; 1: RAdam (rate scaled by step_size, division by sqrt(v) only when `denom`)
    float step_size;  // RAdam: rectification factor / bias correction, computed on the host (radam.rs:283-302)
    int denom;
    int block_start[BT_ADAMW_MAX_TENSORS + 1];  // prefix sum of blocks per tensor
    bt_adamw_tensor t[BT_ADAMW_MAX_TENSORS];
};
constexpr int ADAM_ELEMS_PER_BLOCK = 1024;  // 256 threads x float4

// radam.rs:35-52, in this exact order
__device__ __forceinline__ void radam_op(float grad, float rate, int denom, float &p, float &m, float &v,
                                         const bt_adamw_tensor &t) {
    p *= 1.0f - t.decay * rate;
    m = t.beta1 * m + (1.0f - t.beta1) * grad;
    v = t.beta2 * v + (1.0f - t.beta2) * grad * grad;
    float val = m;
    if (denom) val /= __fsqrt_rn(v) + 0.00000001f;
    p -= rate * val;
    p = fminf(fmaxf(p, t.min_weight), t.max_weight);
}

// Ranger lookahead: w = s = alpha*w + (1-alpha)*s (ranger.rs:23-44)
__global__ __launch_bounds__(256) void lookahead_kernel(float *__restrict__ w, float *__restrict__ slow, size_t n,
                                                        float alpha) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float nw = alpha * w[i] + (1.0f - alpha) * slow[i];
    w[i] = nw, slow[i] = nw;
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamTable tab, const float *__restrict__ adj_ptr,
                                                    const float *__restrict__ rate_ptr) {
    int ti = 0;
    while (ti + 1 < tab.n_tensors && (int)blockIdx.x >= tab.block_start[ti + 1]) ti++;
    const bt_adamw_tensor &t = tab.t[ti];
    const float adj = adj_ptr[0], rate = tab.radam ? rate_ptr[0] * tab.step_size : rate_ptr[0];
    if (tab.radam) {
        const size_t r0 = ((size_t)(blockIdx.x - tab.block_start[ti]) * 256 + threadIdx.x) * 4;
        for (size_t e = r0; e < t.n && e < r0 + 4; e++) {
            float p = t.param[e], m = t.momentum[e], v = t.velocity[e];
            radam_op(adj * t.grad[e], rate, tab.denom, p, m, v, t);
            t.param[e] = p, t.momentum[e] = m, t.velocity[e] = v;
        }
        return;
    }
    const size_t e0 = ((size_t)(blockIdx.x - tab.block_start[ti]) * 256 + threadIdx.x) * 4;
    if (e0 >= t.n) return;
    if (e0 + 4 <= t.n && (t.n % 4 == 0)) {
        float4 p = *reinterpret_cast<float4 *>(t.param + e0);
        float4 m = *reinterpret_cast<float4 *>(t.momentum + e0);
        float4 v = *reinterpret_cast<float4 *>(t.velocity + e0);
        const float4 g = *reinterpret_cast<const float4 *>(t.grad + e0);
        bt_adam_op(__fmul_rn(adj, g.x), rate, p.x, m.x, v.x, t);
        bt_adam_op(__fmul_rn(adj, g.y), rate, p.y, m.y, v.y, t);
        bt_adam_op(__fmul_rn(adj, g.z), rate, p.z, m.z, v.z, t);
        bt_adam_op(__fmul_rn(adj, g.w), rate, p.w, m.w, v.w, t);
        *reinterpret_cast<float4 *>(t.param + e0) = p;
        *reinterpret_cast<float4 *>(t.momentum + e0) = m;
        *reinterpret_cast<float4 *>(t.velocity + e0) = v;
    } else {
        for (size_t e = e0; e < t.n && e < e0 + 4; e++) {
            float p = t.param[e], m = t.momentum[e], v = t.velocity[e];
            bt_adam_op(__fmul_rn(adj, t.grad[e]), rate, p, m, v, t);
            t.param[e] = p, t.momentum[e] = m, t.velocity[e] = v;
        }
    }
}


// ------------------------------------------------------------------ small-head fast paths
// Forward for layers whose whole weight matrix fits in LDS (config 2: 8 x 2048 x 4 B = 64 KiB):
// W is transposed into LDS once per workgroup, then one wave per position streams x with float4
// loads (HBM-bound on x) and reads the selected bucket's rows from LDS.
constexpr int DF_THREADS = 1024;

// LOSS (final layer, out == 1): sigmoid + squared error + gradient are applied to y right where it is produced and
// each workgroup emits one loss partial (positions in ascending order per wave, waves combined in order).
struct LossArgs {
    const float *targets, *entry_weights;
    float *grad_y, *loss_partials;
    float *gz_out;  // optional: gradient w.r.t. this layer's pre-activation (what the FT backward consumes)
    float *gz_parts = nullptr;  // optional (head_finish): [workgroups][buckets] sums of gz_out by bucket = dbias partials
};

template <int NX, bool LOSS>
__global__ __launch_bounds__(DF_THREADS) void dense_fwd_small(const float *__restrict__ w,
                                                              const float *__restrict__ bias,
                                                              const float *__restrict__ x,
                                                              const int32_t *__restrict__ bucket,
                                                              float *__restrict__ y, bt_dense_desc d, int pos_per_wg,
                                                              LossArgs la) {
    extern __shared__ __align__(16) float wt[];  // [rows][in + 4] (16-B aligned: ds_read_b128)
    __shared__ float wave_loss[DF_THREADS / 64];
    float loss_acc = 0.0f;
    const int rows = d.out * d.buckets;
    const int ldw = d.in + 4;
    for (int q = threadIdx.x; q < rows * d.in; q += DF_THREADS) {
        const int i = q / rows, r = q - i * rows;
        wt[r * ldw + i] = w[q];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b_end = min(d.batch, (int)(blockIdx.x + 1) * pos_per_wg);
    for (int b = blockIdx.x * pos_per_wg + wave; b < b_end; b += DF_THREADS / 64) {
        const int base = (d.buckets > 1) ? bucket[b] * d.out : 0;
        const float4 *xb = reinterpret_cast<const float4 *>(x + (size_t)b * d.in);
        float4 xv[NX];
#pragma unroll
        for (int j = 0; j < NX; j++) xv[j] = xb[j * 64 + lane];
        for (int o = 0; o < d.out; o++) {
            const float4 *wr = reinterpret_cast<const float4 *>(wt + (base + o) * ldw);
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < NX; j++) {
                const float4 w4 = wr[j * 64 + lane];
                acc += xv[j].x * w4.x;
                acc += xv[j].y * w4.y;
                acc += xv[j].z * w4.z;
                acc += xv[j].w * w4.w;
            }
            const float sum = wave_sum(acc);
            const float yv = bt_act_fwd(d.act, sum + bias[base + o]);
            if (lane == 0) y[(size_t)b * d.out + o] = yv;
            if constexpr (LOSS) {  // same arithmetic as sigmoid_mse_kernel
                const float sg = 1.0f / (1.0f + expf(-1.0f * yv));
                const float diff = sg + (-1.0f * la.targets[b]);
                const float g = la.entry_weights ? la.entry_weights[b] : 1.0f;
                float l = diff * diff;
                if (la.entry_weights) l = g * l;
                loss_acc += l;
                const float dd = g * diff + g * diff;
                const float gy = dd * (sg * (1.0f + (-1.0f * sg)));
                if (lane == 0) {
                    la.grad_y[b] = gy;
                    if (la.gz_out) la.gz_out[b] = gy * bt_act_deriv(d.act, yv);  // same product as dense_bwd_small
                }
            }
        }
    }
    if constexpr (LOSS) {
        if (lane == 0) wave_loss[wave] = loss_acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.0f;
            for (int i = 0; i < DF_THREADS / 64; i++) t += wave_loss[i];
            la.loss_partials[blockIdx.x] = t;
        }
    }
}

// Finishing pass of the head forward that ft_fwd_lds<HEAD> started: z[b] = sum of the per-tile partial dot products
// (ascending part order) + bias[bucket]; then exactly the epilogue of dense_fwd_small<LOSS>.
constexpr int HF_THREADS = 64;  // one wave per workgroup: many small workgroups, the partial loads are the latency
__global__ __launch_bounds__(HF_THREADS) void head_finish_kernel(const float *__restrict__ zpart, int nparts,
                                                                 const float *__restrict__ bias,
                                                                 const int32_t *__restrict__ bucket,
                                                                 float *__restrict__ y, bt_dense_desc d, LossArgs la) {
    __shared__ float red[HF_THREADS];
    const int b = blockIdx.x * HF_THREADS + threadIdx.x;
    float l = 0.0f, gzv = 0.0f;
    int base_b = -1;
    if (b < d.batch) {
        float z = 0.0f;
        int p = 0;
        for (; p + 32 <= nparts; p += 32) {  // 32 loads in flight (the kernel is their latency), added in ascending order
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; u++) v[u] = zpart[(size_t)(p + u) * d.batch + b];
#pragma unroll
            for (int u = 0; u < 32; u++) z += v[u];
        }
        for (; p + 16 <= nparts; p += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = zpart[(size_t)(p + u) * d.batch + b];
#pragma unroll
            for (int u = 0; u < 16; u++) z += v[u];
        }
        for (; p < nparts; p++) z += zpart[(size_t)p * d.batch + b];
        const int base = (d.buckets > 1) ? bucket[b] : 0;
        const float yv = bt_act_fwd(d.act, z + bias[base]);
        y[b] = yv;
        const float sg = 1.0f / (1.0f + expf(-1.0f * yv));
        const float diff = sg + (-1.0f * la.targets[b]);
        const float g = la.entry_weights ? la.entry_weights[b] : 1.0f;
        l = diff * diff;
        if (la.entry_weights) l = g * l;
        const float dd = g * diff + g * diff;
        const float gy = dd * (sg * (1.0f + (-1.0f * sg)));
        la.grad_y[b] = gy;
        gzv = gy * bt_act_deriv(d.act, yv), base_b = base;
        if (la.gz_out) la.gz_out[b] = gzv;
    }
    red[threadIdx.x] = l;
    __syncthreads();
    for (int o = HF_THREADS >> 1; o > 0; o >>= 1) {  // fixed tree => deterministic partial
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) la.loss_partials[blockIdx.x] = red[0];
    if (la.gz_parts) {  // the bias gradient of the head, by bucket (Select backward + batch sum): one wave, fixed tree
        for (int m = 0; m < d.buckets; m++) {
            const float s = wave_sum(base_b == m ? gzv : 0.0f);
            if (threadIdx.x == 0) la.gz_parts[(size_t)blockIdx.x * d.buckets + m] = s;
        }
    }
}

constexpr int DBS_RANGE = 128;  // positions per workgroup in dense_bwd_small

// Backward for out == 1 heads (one selected row per position): every thread owns 4 input columns and keeps
// the dW accumulators of ALL bucket rows in registers. The bucket of a position is the same for every thread, so
// it is read into an SGPR and the select (SelectPad backward: zeros outside the selected row) becomes a scalar
// branch to the one row that changes: 4 FMAs per position. x is streamed with two batches of U loads in
// flight (HBM-bound). Positions are visited in ascending order inside a workgroup and workgroup partials are
// merged in order => deterministic. GX: also emit dX = gz * W[bucket] (skipped when the FT backward forms it itself).
constexpr int DBS_CT = 256;  // column threads (float4 columns) per workgroup
constexpr int DBS_SUB = 2;  // position sub-ranges of a workgroup (one per 256 threads), merged in order at the end
// DBS_SUB sub-ranges / U loads per batch: the stand-alone shape. <1, 4>: the "light" shape (4 waves, <= 96 VGPRs,
// < 1 KiB of LDS) that fits on a CU next to a resident ft_bwd_lds workgroup, see bt_net_forward_backward.
template <int ROWS, bool GX, int SUBS, int U>
__device__ __forceinline__ void dense_bwd_small_body(const float *__restrict__ w, const float *__restrict__ x,
                                                     const int32_t *__restrict__ bucket, const float *__restrict__ y,
                                                     const float *__restrict__ grad_y, float *__restrict__ part_w,
                                                     float *__restrict__ part_b, float *__restrict__ grad_x,
                                                     const bt_dense_desc &d, int range, float *__restrict__ gz_out,
                                                     const float *__restrict__ loss_partials, int n_loss_partials,
                                                     float *__restrict__ loss_sum) {
    const int rows = d.buckets;  // out == 1
    if (loss_partials && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        float t = 0.0f;  // fixed-order final sum of the forward kernel's loss partials
        for (int i = 0; i < n_loss_partials; i++) t += loss_partials[i];
        loss_sum[0] = t;
    }
    const int ct = threadIdx.x % DBS_CT, sub = (SUBS > 1) ? threadIdx.x / DBS_CT : 0;
    const int in4 = d.in >> 2;
    const int c4 = min((int)(blockIdx.x * DBS_CT + ct), in4 - 1);  // clamped: loads stay unconditional
    const bool active = (int)(blockIdx.x * DBS_CT + ct) < in4;
    float4 wr[GX ? ROWS : 1], acc[ROWS];
    float accb[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        acc[r] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        accb[r] = 0.0f;
        if constexpr (GX) {
            wr[r] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (r < rows) {
                const float *wc = w + (size_t)rows * (c4 * 4) + r;
                wr[r] = make_float4(wc[0], wc[rows], wc[2 * rows], wc[3 * rows]);
            }
        }
    }
    const int b0 = blockIdx.y * range, b1 = min(d.batch, b0 + range);
    const int nb = b1 - b0;
    // per-position scalars (gz, bucket) staged once in LDS: the position loop then has no dependent global loads
    __shared__ float s_gz[DBS_RANGE];
    __shared__ uint8_t s_bk[DBS_RANGE];  // dense_bwd_small_ok: buckets <= 16
    constexpr int RCH = 2048 / DBS_CT;  // rows merged per round (32 KiB of LDS)
    __shared__ float4 red[SUBS > 1 ? RCH : 1][SUBS > 1 ? DBS_CT : 1];
    __shared__ float redb[SUBS > 1 ? SUBS : 1][16];
    for (int t = threadIdx.x; t < DBS_RANGE; t += blockDim.x) {
        float gz = 0.0f;  // positions past the end contribute nothing
        int bk = 0;
        if (t < nb) {
            const int b = b0 + t;
            gz = grad_y[b] * bt_act_deriv(d.act, y[b]);
            bk = (d.buckets > 1) ? bucket[b] : 0;
            if (gz_out && blockIdx.x == 0) gz_out[b] = gz;
        }
        s_gz[t] = gz;
        s_bk[t] = (uint8_t)bk;
    }
    __syncthreads();
    // this sub-range: positions [p0, p1) of the workgroup's range
    constexpr int PER = DBS_RANGE / SUBS;
    const int p0 = sub * PER, p1 = min(nb, p0 + PER);
    const float4 *x4 = reinterpret_cast<const float4 *>(x) + (size_t)b0 * in4 + c4;
    float4 *gx4 = (GX && grad_x) ? reinterpret_cast<float4 *>(grad_x) + (size_t)b0 * in4 + c4 : nullptr;
    auto load = [&](int t0, float4(&xv)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) xv[u] = x4[(size_t)min(t0 + u, max(nb - 1, 0)) * in4];
    };
    auto consume = [&](int t0, const float4(&xv)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int bk = __builtin_amdgcn_readfirstlane((int)s_bk[min(t0 + u, DBS_RANGE - 1)]);
            const float gz = (t0 + u < p1) ? s_gz[min(t0 + u, DBS_RANGE - 1)] : 0.0f;
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                if (r == bk) {  // scalar compare: one row executes
                    acc[r].x += gz * xv[u].x, acc[r].y += gz * xv[u].y, acc[r].z += gz * xv[u].z, acc[r].w += gz * xv[u].w;
                    accb[r] += gz;
                    if constexpr (GX) {
                        // 0 + gz*w, as the masked sum over rows gave
                        if (gx4 && active && t0 + u < p1)
                            gx4[(size_t)(t0 + u) * in4] = make_float4(0.0f + gz * wr[r].x, 0.0f + gz * wr[r].y,
                                                                      0.0f + gz * wr[r].z, 0.0f + gz * wr[r].w);
                    }
                }
            }
        }
    };
    if (p0 < p1) {
        float4 xa[U], xb[U];
        load(p0, xa);
        for (int t0 = p0; t0 < p1; t0 += 2 * U) {
            load(t0 + U, xb);
            consume(t0, xa);
            load(t0 + 2 * U, xa);
            consume(t0 + U, xb);
        }
    }
    // merge the sub-ranges in ascending order (sub 0 ends up with the workgroup's sums)
    for (int rc = 0; SUBS > 1 && rc < ROWS; rc += RCH) {
        for (int sdone = 1; sdone < SUBS; sdone++) {
            __syncthreads();
            if (sub == sdone) {
#pragma unroll
                for (int r = 0; r < RCH; r++)
                    if (rc + r < ROWS) red[r][ct] = acc[rc + r];
            }
            __syncthreads();
            if (sub == 0) {
#pragma unroll
                for (int r = 0; r < RCH; r++)
                    if (rc + r < ROWS) {
                        const float4 v = red[r][ct];
                        acc[rc + r].x += v.x, acc[rc + r].y += v.y, acc[rc + r].z += v.z, acc[rc + r].w += v.w;
                    }
            }
        }
    }
    if constexpr (SUBS > 1) {
        if (ct == 0) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) redb[sub][r] = accb[r];
        }
        __syncthreads();
    }
    if (sub == 0 && active) {
        float *pw = part_w + (size_t)blockIdx.y * rows * d.in + (size_t)rows * (c4 * 4);
#pragma unroll
        for (int r = 0; r < ROWS; r++)
            if (r < rows) pw[r] = acc[r].x, pw[rows + r] = acc[r].y, pw[2 * rows + r] = acc[r].z, pw[3 * rows + r] = acc[r].w;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int r = 0; r < ROWS; r++)
            if (r < rows) {
                float t = accb[r];
                if constexpr (SUBS > 1) {
                    t = redb[0][r];
                    for (int s2 = 1; s2 < SUBS; s2++) t += redb[s2][r];
                }
                part_b[(size_t)blockIdx.y * rows + r] = t;
            }
    }
}

template <int ROWS, bool GX>
__global__ __launch_bounds__(DBS_CT * DBS_SUB) void dense_bwd_small(const float *__restrict__ w, const float *__restrict__ x,
                                                                    const int32_t *__restrict__ bucket,
                                                                    const float *__restrict__ y,
                                                                    const float *__restrict__ grad_y,
                                                                    float *__restrict__ part_w, float *__restrict__ part_b,
                                                                    float *__restrict__ grad_x, bt_dense_desc d, int range,
                                                                    float *__restrict__ gz_out,
                                                                    const float *__restrict__ loss_partials,
                                                                    int n_loss_partials, float *__restrict__ loss_sum) {
    dense_bwd_small_body<ROWS, GX, DBS_SUB, 8>(w, x, bucket, y, grad_y, part_w, part_b, grad_x, d, range, gz_out,
                                               loss_partials, n_loss_partials, loss_sum);
}

template <int ROWS>
__global__ __launch_bounds__(DBS_CT) __attribute__((amdgpu_waves_per_eu(5, 5))) void dense_bwd_small_light(
    const float *__restrict__ w, const float *__restrict__ x, const int32_t *__restrict__ bucket,
    const float *__restrict__ y, const float *__restrict__ grad_y, float *__restrict__ part_w,
    float *__restrict__ part_b, bt_dense_desc d, int range, const float *__restrict__ loss_partials,
    int n_loss_partials, float *__restrict__ loss_sum) {
    dense_bwd_small_body<ROWS, false, 1, 4>(w, x, bucket, y, grad_y, part_w, part_b, nullptr, d, range, nullptr,
                                            loss_partials, n_loss_partials, loss_sum);
}


// ------------------------------------------------------------------ mid-size bucketed layers
// Layers with a few outputs per bucket and a wide input (HalfKA nets: 1024 -> 16 x 8 buckets). The reference runs
// a GEMM over ALL bucket rows and discards 7/8 of it (matmul.rs:62-93 + select.rs:12-26); here only the selected
// bucket is computed. One bucket's rows (out x in, 64 KiB) fit in LDS, all buckets do not, so a workgroup takes a
// chunk of positions and goes through the buckets present in it: load that bucket's rows transposed, then one wave
// per matching position. Positions keep their ascending order inside a bucket => fixed summation orders.
constexpr int DM_THREADS = 1024;
constexpr int DM_CHUNK = 64;    // positions per workgroup, forward / dX
constexpr int DM_RANGE = 128;   // positions per workgroup, dW
constexpr int DM_OH = 8;        // output rows per workgroup, dW

// Positions sorted by output bucket (stable) for the kernels below: with real data every chunk of consecutive
// positions holds all the buckets, and a chunk re-stages the bucket's 64 KiB of weights once per bucket present;
// walking the batch in bucket order makes that once per chunk. net.hip builds the permutation once per step
// (bucket_count_kernel + bucket_place_kernel) and installs it for the calling thread; the public bt_dense_* entry points run without.
// The same kernel cuts the sorted order into bucket-pure items of <= DX_CH positions {bucket, start, count}; the
// 16-output layers then run as small f32 GEMMs on the matrix cores, one item per workgroup (dense_mfma_*).
struct DensePermTab {
    const int32_t *perm = nullptr;  // [batch]
    const int4 *items = nullptr;    // [max_items]
    const int *bucket_items = nullptr;  // [nb + 1] item range of each bucket
    const int *counts = nullptr;    // [0] number of items
    int max_items = 0;
    int *blk_cnt = nullptr;         // scratch of the two-kernel build: [ceil(batch / BPC_BLOCK)][nb] block histograms
};
thread_local DensePermTab g_dense_tab;
// set by net.hip around a bt_dense_backward whose grad_y already is the gradient w.r.t. the layer's pre-activation
// (the fused tail formed it): the mid-size paths then skip dense_gz_kernel
thread_local bool g_dense_gy_is_gz = false;
thread_local hipStream_t g_dense_side = nullptr;  // bt_internal_dense_set_side
thread_local hipEvent_t g_dense_fork = nullptr;
thread_local bool g_dense_side_behind = false;  // the side stream already waits for the kernel that produced grad_y
// Work that only the optimiser waits for goes to the side stream, ordered behind the kernel launched LAST on `s`: the
// fork event rides on that kernel's dispatch packet (dense_fork_on, before its launch) -- a record of its own on `s`
// would put a barrier packet between two kernels of the critical path (12 us measured) -- and the side stream waits.
void dense_fork_on(int kernel_id) {
    if (g_dense_side) bt_attach_stop_event(kernel_id, g_dense_fork);
}
int dense_side_join_fork(hipStream_t s, hipStream_t *out, bool attached) {
    *out = s;
    if (!g_dense_side) return BT_OK;
    if (!attached) BT_HIP(hipEventRecord(g_dense_fork, s));
    BT_HIP(hipStreamWaitEvent(g_dense_side, g_dense_fork, 0));
    *out = g_dense_side;
    return BT_OK;
}
constexpr int DX_CH = 64;  // positions per item = 4 waves x one 16-row MFMA tile

// Stable counting sort of the positions by output bucket in two multi-workgroup kernels (the segment grouping of
// ft.hip has the same shape): per-block histograms, then every block sums the histograms of the blocks before it for
// itself -- nblk * nb integers that sit in L2 -- ranks its positions with one ballot per bucket and places them;
// block 0 also cuts the order into bucket-pure items of DX_CH positions. Same order as a serial stable sort.
constexpr int BPC_BLOCK = 256;
__global__ __launch_bounds__(BPC_BLOCK) void bucket_count_kernel(const int32_t *__restrict__ bucket, int batch, int nb,
                                                                 int *__restrict__ blk_cnt) {
    __shared__ int hist[16];
    if ((int)threadIdx.x < nb) hist[threadIdx.x] = 0;
    __syncthreads();
    const int p = blockIdx.x * BPC_BLOCK + threadIdx.x;
    if (p < batch) atomicAdd(&hist[min(max(bucket[p], 0), nb - 1)], 1);
    __syncthreads();
    if ((int)threadIdx.x < nb) blk_cnt[(size_t)blockIdx.x * nb + threadIdx.x] = hist[threadIdx.x];
}

__global__ __launch_bounds__(BPC_BLOCK) void bucket_place_kernel(const int32_t *__restrict__ bucket, int batch, int nb,
                                                                 const int *__restrict__ blk_cnt, int nblk,
                                                                 int32_t *__restrict__ perm, int4 *__restrict__ items,
                                                                 int *__restrict__ bucket_items, int *__restrict__ counts) {
    __shared__ int tot[16], before[16], s_start[17], s_istart[17], wcnt[BPC_BLOCK / 64][16];
    if ((int)threadIdx.x < nb) tot[threadIdx.x] = 0, before[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nblk * nb; i += BPC_BLOCK) {  // integer sums: order does not matter
        const int v = blk_cnt[i];
        if (v) {
            const int blk = i / nb, k = i - blk * nb;
            atomicAdd(&tot[k], v);
            if (blk < (int)blockIdx.x) atomicAdd(&before[k], v);
        }
    }
    const int p = blockIdx.x * BPC_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bk = p < batch ? min(max(bucket[p], 0), nb - 1) : -1;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int rank = 0;
    for (int k = 0; k < nb; k++) {
        const unsigned long long m = __ballot(bk == k);
        if (bk == k) rank = __popcll(m & lt_mask);
        if (lane == 0) wcnt[wave][k] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, ni = 0;
        for (int k = 0; k < nb; k++) {
            s_start[k] = a, s_istart[k] = ni;
            a += tot[k], ni += (tot[k] + DX_CH - 1) / DX_CH;
        }
        s_start[nb] = a, s_istart[nb] = ni;
    }
    __syncthreads();
    if (p < batch) {
        int pos = s_start[bk] + before[bk] + rank;
        for (int w2 = 0; w2 < wave; w2++) pos += wcnt[w2][bk];
        perm[pos] = p;
    }
    if (blockIdx.x != 0) return;
    if ((int)threadIdx.x <= nb) bucket_items[threadIdx.x] = s_istart[threadIdx.x];
    if (threadIdx.x == 0) counts[0] = s_istart[nb];
    for (int i = threadIdx.x; i < s_istart[nb]; i += BPC_BLOCK) {
        int k = 0;
        while (k + 1 < nb && s_istart[k + 1] <= i) k++;
        const int o = s_start[k] + (i - s_istart[k]) * DX_CH;
        items[i] = make_int4(k, o, min(DX_CH, s_start[k + 1] - o), 0);
    }
}

// ---- 16 outputs per bucket (the first hidden layer of the multi-layer nets, (16 x buckets) x 1024): f32 MFMA
// (v_mfma_f32_16x16x4_f32: exact f32, a k-ordered fma chain) on bucket-pure items. The reference computes all buckets
// as one GEMM and discards all but one (matmul.rs:62-93 + select.rs:12-26); here an item multiplies its 64 positions
// with its own bucket's 16 rows only, so the passes are bound by streaming x / grad_x (64 MiB at B = 16384) once.
// Lane l of a wave: tile row/column l & 15, k = l >> 4; D: column l & 15, rows 4 (l >> 4) + r.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

bool dense_mfma_ok(const bt_dense_desc *d) {
    return d->buckets > 1 && d->buckets <= 16 && d->out == 16 && d->in % 64 == 0 && d->in >= 64 && d->in <= 2048;
}

// y[pos][o] = f(W[bucket][o] . x[pos] + bias). 16 waves: wave (m, kq) multiplies the 16-position tile m with the kq-th
// quarter of the inputs (W tile from LDS), so four waves per SIMD keep HBM requests in flight; the four partial tiles
// of a position tile are added in the order kq = 0, 1, 2, 3 through LDS.
constexpr int MF_THREADS = 1024;
__global__ __launch_bounds__(MF_THREADS) void dense_mfma_fwd(const float *__restrict__ w, const float *__restrict__ bias,
                                                             const float *__restrict__ x, float *__restrict__ y,
                                                             bt_dense_desc d, DensePermTab tab) {
    extern __shared__ __align__(16) float wt[];  // [16][in + 4]; afterwards the partial tiles [4 m][3 kq][64][4]
    const int item = blockIdx.x;
    if (item >= tab.counts[0]) return;
    const int4 it = tab.items[item];
    const int bk = it.x, start = it.y, cnt = it.z;
    const int rows = d.out * d.buckets, ldw = d.in + 4;
    for (int q = threadIdx.x; q < 4 * d.in; q += MF_THREADS) {  // (rows and bk * 16 are multiples of 4: 16-B aligned loads)
        const int i = q >> 2, r4 = (q & 3) * 4;
        const float4 v = *reinterpret_cast<const float4 *>(w + (size_t)rows * i + bk * 16 + r4);
        wt[(r4 + 0) * ldw + i] = v.x, wt[(r4 + 1) * ldw + i] = v.y, wt[(r4 + 2) * ldw + i] = v.z, wt[(r4 + 3) * ldw + i] = v.w;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, qd = lane >> 4;
    const int m = wave & 3, kq = wave >> 2;
    const int pos = tab.perm[start + min(16 * m + n, cnt - 1)];
    const int nk = d.in / 64;  // k-blocks of 16 per quarter (in % 64 == 0)
    const float4 *xr = reinterpret_cast<const float4 *>(x + (size_t)pos * d.in) + qd + 4 * kq * nk;
    const float4 *wr = reinterpret_cast<const float4 *>(wt + n * ldw) + qd + 4 * kq * nk;
    f32x4_t acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    // k = 16 t + 4 qd + j on both operands (any consistent k order gives the same set of products)
    for (int t = 0; t < nk; t++) {
        const float4 a = xr[4 * t], b = wr[4 * t];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc1, 0, 0, 0);
    }
    f32x4_t acc = acc0 + acc1;
    __syncthreads();  // every wave is done with the W tile
    f32x4_t *red = reinterpret_cast<f32x4_t *>(wt);
    if (kq > 0) red[(m * 3 + kq - 1) * 64 + lane] = acc;
    __syncthreads();
    if (kq > 0) return;
    for (int k2 = 0; k2 < 3; k2++) acc += red[(m * 3 + k2) * 64 + lane];
    const float bias_n = bias[bk * 16 + n];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int pl = 16 * m + 4 * qd + r;
        if (pl < cnt) y[(size_t)tab.perm[start + pl] * 16 + n] = bt_act_fwd(d.act, acc[r] + bias_n);
    }
}

// gx[pos][i] = sum_o W[bucket][o][i] * gz[pos][o]: K = the 16 outputs (one MFMA quad per 16 x 16 tile); a wave keeps
// the gz fragments of the item's four position tiles and walks its quarter of the input columns
__global__ __launch_bounds__(256) void dense_mfma_dx(const float *__restrict__ w, const float *__restrict__ gz,
                                                     float *__restrict__ gx, bt_dense_desc d, DensePermTab tab) {
    __shared__ int s_pos[DX_CH];
    const int item = blockIdx.x;
    if (item >= tab.counts[0]) return;
    const int4 it = tab.items[item];
    const int bk = it.x, start = it.y, cnt = it.z;
    const int rows = d.out * d.buckets;
    if ((int)threadIdx.x < DX_CH) s_pos[threadIdx.x] = tab.perm[start + min((int)threadIdx.x, cnt - 1)];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, qd = lane >> 4;
    float4 a[4];  // A[position 16 m + n][o = 4 qd + j]
#pragma unroll
    for (int m = 0; m < 4; m++) {
        a[m] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (16 * m + n < cnt) a[m] = *reinterpret_cast<const float4 *>(gz + (size_t)s_pos[16 * m + n] * 16 + 4 * qd);
    }
    const int ntile = d.in / 16, per = (ntile + 3) / 4;
    for (int nt = wave * per; nt < min(ntile, (wave + 1) * per); nt++) {
        // B[o = 4 qd + j][i = 16 nt + n]: four consecutive rows of W's column i
        const float4 b = *reinterpret_cast<const float4 *>(w + (size_t)rows * (16 * nt + n) + bk * 16 + 4 * qd);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            if (16 * m >= cnt) break;
            f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].w, b.w, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pl = 16 * m + 4 * qd + r;
                if (pl < cnt) gx[(size_t)s_pos[pl] * d.in + 16 * nt + n] = acc[r];
            }
        }
    }
}

// part_w[item][o][i] = sum over the item's positions of gz[pos][o] * x[pos][i]; part_b[item][o] = sum of gz[pos][o].
// K = positions, 4 per MFMA; a lane reads x as float4 and feeds register j to the tile of columns 64 T + 4 n + j,
// so one 256-B row segment serves four MFMAs. A wave owns a quarter of the 64-column blocks T.
template <int TW>
__global__ __launch_bounds__(256) void dense_mfma_dw(const float *__restrict__ x, const float *__restrict__ gz,
                                                     float *__restrict__ part_w, float *__restrict__ part_b,
                                                     bt_dense_desc d, DensePermTab tab) {
    __shared__ int s_pos[DX_CH];
    const int item = blockIdx.x;
    if (item >= tab.counts[0]) return;
    const int4 it = tab.items[item];
    const int start = it.y, cnt = it.z;
    if ((int)threadIdx.x < DX_CH) s_pos[threadIdx.x] = tab.perm[start + min((int)threadIdx.x, cnt - 1)];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, qd = lane >> 4;
    const int nT = d.in / 64, T0 = wave * TW;
    f32x4_t acc[TW][4], accb = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < TW; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[t][j] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    for (int p0 = 0; p0 < cnt; p0 += 4) {
        const int p = p0 + qd;
        const bool live = p < cnt;
        const int pos = s_pos[min(p, cnt - 1)];
        const float av = live ? gz[(size_t)pos * 16 + n] : 0.0f;  // A[o = n][k = position p0 + qd]
        const float *xr = x + (size_t)pos * d.in + 4 * n;
        float4 b[TW];
#pragma unroll
        for (int t = 0; t < TW; t++)
            b[t] = (T0 + t < nT) ? *reinterpret_cast<const float4 *>(xr + 64 * (T0 + t)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int t = 0; t < TW; t++) {
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[t].x, acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[t].y, acc[t][1], 0, 0, 0);
            acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[t].z, acc[t][2], 0, 0, 0);
            acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[t].w, acc[t][3], 0, 0, 0);
        }
        if (wave == 0) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(av, 1.0f, accb, 0, 0, 0);  // every column = sum of gz
    }
    float *pw = part_w + (size_t)item * 16 * d.in;
#pragma unroll
    for (int t = 0; t < TW; t++) {
        if (T0 + t >= nT) break;
#pragma unroll
        for (int r = 0; r < 4; r++)  // D row o = 4 qd + r; columns 64 T + 4 n + j, j = 0..3
            *reinterpret_cast<float4 *>(pw + (size_t)(4 * qd + r) * d.in + 64 * (T0 + t) + 4 * n) =
                make_float4(acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]);
    }
    if (wave == 0 && n == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) part_b[(size_t)item * 16 + 4 * qd + r] = accb[r];
    }
}

// Sum of element e over the items [lo, hi) of one bucket, for a 256-thread block laid out as 32 elements x 8 slices:
// slice s adds items lo + s, lo + s + 8, ... in ascending order, the eight slice sums are then added in slice order.
// The order depends only on the item list, so results are bit-identical run to run; with every position in one
// bucket (hundreds of items) the chain per thread is 8 times shorter than a plain loop.
__device__ __forceinline__ float sliced_item_sum(const float *__restrict__ part, size_t item_stride, size_t e, int lo,
                                                 int hi, bool live, float *red /* [8][32] */) {
    const int el = threadIdx.x & 31, slice = threadIdx.x >> 5;
    float s = 0.0f;
    if (live)
        for (int g = lo + slice; g < hi; g += 8) s += part[(size_t)g * item_stride + e];
    red[slice * 32 + el] = s;
    __syncthreads();
    float tot = 0.0f;
    if (slice == 0)
        for (int k = 0; k < 8; k++) tot += red[k * 32 + el];
    return tot;  // valid on slice 0
}

// dw[bucket*16 + o][i] = sum of the bucket's item partials (zeros for a bucket without positions); grid (ceil((16 in + 16) / 32), nb)
__global__ __launch_bounds__(256) void dense_mfma_sum(const float *__restrict__ part_w, const float *__restrict__ part_b,
                                                      float *__restrict__ dw, float *__restrict__ dbias, bt_dense_desc d,
                                                      DensePermTab tab) {
    __shared__ float red[256];
    const int rows = d.out * d.buckets, bk = blockIdx.y;
    const int e = blockIdx.x * 32 + (threadIdx.x & 31), nw = 16 * d.in;
    const int lo = tab.bucket_items[bk], hi = tab.bucket_items[bk + 1];
    const bool isw = e < nw, isb = !isw && e < nw + 16;
    const float tot = sliced_item_sum(isw ? part_w : part_b, isw ? (size_t)16 * d.in : 16, isw ? e : e - nw, lo, hi,
                                      isw || isb, red);
    if ((threadIdx.x >> 5) != 0) return;
    if (isw) {
        const int o = e / d.in, i = e - o * d.in;
        dw[(size_t)rows * i + bk * 16 + o] = tot;
    } else if (isb) {
        dbias[bk * 16 + (e - nw)] = tot;
    }
}

// ---- fused tail of the multi-layer nets (examples/progression/4_multi_layer.rs: ... -> 16 -> 32 -> 1, all
// bucketed): the two last layers, the sigmoid-MSE loss and their whole backward pass for one bucket-pure item of 64
// positions in one workgroup. Everything between the first hidden layer's output a1 and the gradient w.r.t. a1 stays
// in LDS; thirteen small launches of the layer-by-layer path become two (this kernel + dense_tail_sum).
// Per-position arithmetic is the same as the layer kernels' (dot products in ascending input order from 0.0f, bias
// added last, derivatives from outputs); weight gradients are summed position by position inside an item and item by
// item (list order) per bucket.
constexpr int TAIL_THREADS = 256;
struct TailArgs {
    const float *w2, *b2, *w3, *b3;     // layer A: [(n2*nb) x n1] column-major, layer B: [nb x n2]
    const float *a1;                    // [B][n1] activated output of the layer before
    const float *targets, *ew;          // ew may be null
    float *y, *ga1;                     // net output [B], gradient w.r.t. a1 [B][n1]
    float *part;                        // [max_items][tail_part_floats]
    int n1, n2, nb, act2, act3;
    int act1;                           // >= 0: ga1 receives gradient * f'(a1) of that activation, i.e. the gradient
                                        // w.r.t. the pre-activation of the layer before (its backward then skips
                                        // dense_gz_kernel); -1: the gradient w.r.t. a1 itself
};
__host__ __device__ inline int tail_part_floats(int n1, int n2) { return n2 * n1 + 2 * n2 + 2; }  // dW2 db2 dW3 db3 loss

__global__ __launch_bounds__(TAIL_THREADS) void dense_tail_kernel(TailArgs A, DensePermTab tab) {
    extern __shared__ __align__(16) float tl[];
    const int item = blockIdx.x;
    if (item >= tab.counts[0]) return;
    const int4 it = tab.items[item];
    const int bk = it.x, start = it.y, cnt = it.z;
    const int n1 = A.n1, n2 = A.n2, l1 = n1 + 1, l2 = n2 + 1;
    float *W2 = tl;                      // [n2][n1]
    float *B2 = W2 + n2 * n1;            // [n2]
    float *W3 = B2 + n2;                 // [n2]
    float *a1 = W3 + n2;                 // [64][n1 + 1]
    float *a2 = a1 + DX_CH * l1;         // [64][n2 + 1]
    float *g2 = a2 + DX_CH * l2;         // [64][n2 + 1]  gradient w.r.t. layer A's pre-activation
    float *g3 = g2 + DX_CH * l2;         // [64]          gradient w.r.t. layer B's pre-activation
    float *ls = g3 + DX_CH;              // [64]          loss terms
    __shared__ int s_pos[DX_CH];
    const int t = threadIdx.x, rows2 = n2 * A.nb;
    if (t < DX_CH) s_pos[t] = tab.perm[start + min(t, cnt - 1)];
    for (int q = t; q < n2 * n1; q += TAIL_THREADS) {
        const int o = q % n2, i = q / n2;
        W2[o * n1 + i] = A.w2[(size_t)rows2 * i + bk * n2 + o];
    }
    for (int q = t; q < n2; q += TAIL_THREADS) B2[q] = A.b2[bk * n2 + q], W3[q] = A.w3[(size_t)A.nb * q + bk];
    __syncthreads();
    for (int q = t; q < DX_CH * n1; q += TAIL_THREADS) {
        const int p = q / n1, i = q - p * n1;
        a1[p * l1 + i] = A.a1[(size_t)s_pos[p] * n1 + i];
    }
    __syncthreads();
    // layer A forward
    for (int q = t; q < DX_CH * n2; q += TAIL_THREADS) {
        const int p = q & (DX_CH - 1), o = q / DX_CH;
        float z = 0.0f;
        for (int i = 0; i < n1; i++) z += W2[o * n1 + i] * a1[p * l1 + i];
        a2[p * l2 + o] = bt_act_fwd(A.act2, z + B2[o]);
    }
    __syncthreads();
    // layer B forward, loss, gradient of the loss (same arithmetic as sigmoid_mse_kernel)
    if (t < DX_CH) {
        float l = 0.0f, gz = 0.0f;
        if (t < cnt) {
            float z = 0.0f;
            for (int i = 0; i < n2; i++) z += W3[i] * a2[t * l2 + i];
            const float yv = bt_act_fwd(A.act3, z + A.b3[bk]);
            const int pos = s_pos[t];
            A.y[pos] = yv;
            const float sg = 1.0f / (1.0f + expf(-1.0f * yv));
            const float diff = sg + (-1.0f * A.targets[pos]);
            const float g = A.ew ? A.ew[pos] : 1.0f;
            l = diff * diff;
            if (A.ew) l = g * l;
            const float dd = g * diff + g * diff;
            gz = (dd * (sg * (1.0f + (-1.0f * sg)))) * bt_act_deriv(A.act3, yv);
        }
        g3[t] = gz, ls[t] = l;
    }
    __syncthreads();
    // layer B backward (dX = W3 * gz, one output) and layer A's activation derivative
    for (int q = t; q < DX_CH * n2; q += TAIL_THREADS) {
        const int p = q & (DX_CH - 1), o = q / DX_CH;
        g2[p * l2 + o] = (W3[o] * g3[p]) * bt_act_deriv(A.act2, a2[p * l2 + o]);
    }
    __syncthreads();
    // layer A dX -> gradient w.r.t. a1
    for (int q = t; q < DX_CH * n1; q += TAIL_THREADS) {
        const int p = q / n1, i = q - p * n1;
        if (p < cnt) {
            float sacc = 0.0f;
            for (int o = 0; o < n2; o++) sacc += W2[o * n1 + i] * g2[p * l2 + o];
            if (A.act1 >= 0) sacc = sacc * bt_act_deriv(A.act1, a1[p * l1 + i]);  // = dense_gz_kernel's one multiply
            A.ga1[(size_t)s_pos[p] * n1 + i] = sacc;
        }
    }
    // weight-gradient partials of the item (positions beyond cnt hold zero gradients)
    float *part = A.part + (size_t)item * tail_part_floats(n1, n2);
    for (int q = t; q < n2 * n1; q += TAIL_THREADS) {
        const int o = q / n1, i = q - o * n1;
        float sacc = 0.0f;
        for (int p = 0; p < DX_CH; p++) sacc += g2[p * l2 + o] * a1[p * l1 + i];
        part[q] = sacc;
    }
    for (int q = t; q < n2; q += TAIL_THREADS) {
        float sb = 0.0f, sw = 0.0f;
        for (int p = 0; p < DX_CH; p++) sb += g2[p * l2 + q], sw += g3[p] * a2[p * l2 + q];
        part[n2 * n1 + q] = sb, part[n2 * n1 + n2 + q] = sw;
    }
    if (t == TAIL_THREADS - 1) {
        float sb = 0.0f, sl = 0.0f;
        for (int p = 0; p < DX_CH; p++) sb += g3[p], sl += ls[p];
        part[n2 * n1 + 2 * n2] = sb, part[n2 * n1 + 2 * n2 + 1] = sl;
    }
}

// gradients of both tail layers = per-bucket sums of the item partials; grid (ceil(per-bucket elements / 32), nb + 1):
// the extra row of blocks adds up the loss terms of all items
__global__ __launch_bounds__(256) void dense_tail_sum(const float *__restrict__ part, int n1, int n2, int nb,
                                                      float *__restrict__ dw2, float *__restrict__ db2,
                                                      float *__restrict__ dw3, float *__restrict__ db3,
                                                      float *__restrict__ loss_sum, DensePermTab tab) {
    __shared__ float red[256];
    const int P = tail_part_floats(n1, n2), per_b = P - 1;
    const int bk = blockIdx.y;
    if (bk == nb) {  // loss: element P - 1 of every item
        if (blockIdx.x != 0) return;
        const int ni = tab.counts[0];
        float s = 0.0f;
        for (int g = threadIdx.x; g < ni; g += 256) s += part[(size_t)g * P + P - 1];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss_sum[0] = red[0];
        return;
    }
    const int e = blockIdx.x * 32 + (threadIdx.x & 31);
    const float s = sliced_item_sum(part, (size_t)P, (size_t)e, tab.bucket_items[bk], tab.bucket_items[bk + 1], e < per_b, red);
    if ((threadIdx.x >> 5) != 0 || e >= per_b) return;
    const int rows2 = n2 * nb;
    if (e < n2 * n1) {
        const int o = e / n1, i = e - o * n1;
        dw2[(size_t)rows2 * i + bk * n2 + o] = s;
    } else if (e < n2 * n1 + n2) {
        db2[bk * n2 + (e - n2 * n1)] = s;
    } else if (e < n2 * n1 + 2 * n2) {
        dw3[(size_t)nb * (e - n2 * n1 - n2) + bk] = s;
    } else {
        db3[bk] = s;
    }
}

bool dense_mid_ok(const bt_dense_desc *d) {
    return d->out > 1 && d->out <= 32 && d->buckets <= 32 && d->in % 256 == 0 && d->in / 256 <= 8 &&
           (size_t)d->out * (d->in + 4) * sizeof(float) <= 144 * 1024;
}

// positions of the chunk whose bucket is bk, ascending, into s_list; returns their number (all threads)
__device__ __forceinline__ int dm_bucket_list(const uint8_t *s_bk, int npos, int bk, int *s_list, int *s_wcnt) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned long long m = 0;
    bool flag = false;
    if (t < DM_RANGE) {  // whole waves: DM_RANGE is a multiple of 64
        flag = t < npos && s_bk[t] == bk;
        m = __ballot(flag);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
    }
    __syncthreads();
    if (flag) {
        int base = 0;
        for (int w2 = 0; w2 < wave; w2++) base += s_wcnt[w2];
        s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = t;
    }
    int cnt = 0;
    for (int w2 = 0; w2 < DM_RANGE / 64; w2++) cnt += s_wcnt[w2];
    __syncthreads();
    return cnt;
}

__global__ __launch_bounds__(256) void dense_gz_kernel(const float *__restrict__ y, const float *__restrict__ grad_y,
                                                       float *__restrict__ gz, size_t n, int act) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) gz[t] = grad_y[t] * bt_act_deriv(act, y[t]);
}

// DX = false: y[b][o] = f(W[bucket][o] . x[b] + bias);  DX = true: gx[b][i] = sum_o W[bucket][o][i] * gz[b][o]
template <int NX, bool DX>
__global__ __launch_bounds__(DM_THREADS) void dense_mid_kernel(const float *__restrict__ w, const float *__restrict__ bias,
                                                               const float *__restrict__ src,
                                                               const int32_t *__restrict__ bucket,
                                                               float *__restrict__ dst, bt_dense_desc d,
                                                               const int32_t *__restrict__ perm) {
    extern __shared__ __align__(16) float wt[];  // [out][in + 4]: the current bucket's rows, transposed (16-B aligned)
    __shared__ uint8_t s_bk[DM_RANGE];
    __shared__ int s_list[DM_RANGE], s_wcnt[DM_RANGE / 64], s_pos[DM_RANGE];
    __shared__ unsigned s_mask;
    const int rows = d.out * d.buckets, ldw = d.in + 4;
    const int b0 = blockIdx.x * DM_CHUNK, npos = min(DM_CHUNK, d.batch - b0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_mask = 0;
    __syncthreads();
    if ((int)threadIdx.x < npos) {
        const int pos = perm ? perm[b0 + threadIdx.x] : b0 + (int)threadIdx.x;
        const int bk = (d.buckets > 1) ? bucket[pos] : 0;
        s_pos[threadIdx.x] = pos;
        s_bk[threadIdx.x] = (uint8_t)bk;
        atomicOr(&s_mask, 1u << bk);
    }
    __syncthreads();
    const unsigned mask = s_mask;
    for (int bk = 0; bk < d.buckets; bk++) {
        if (!((mask >> bk) & 1u)) continue;
        const int cnt = dm_bucket_list(s_bk, npos, bk, s_list, s_wcnt);  // (barriers inside: wt readers are done)
        for (int q = threadIdx.x; q < d.out * d.in; q += DM_THREADS) {
            const int i = q / d.out, r = q - i * d.out;
            wt[r * ldw + i] = w[(size_t)rows * i + bk * d.out + r];
        }
        __syncthreads();
        // per-output scalars live in lanes 0..out-1 (one load per round / position) and are read with v_readlane, so
        // the output loop has no dependent global load
        const float bias_l = (!DX && lane < d.out) ? bias[bk * d.out + lane] : 0.0f;
        for (int j = wave; j < cnt; j += DM_THREADS / 64) {
            const int b = __builtin_amdgcn_readfirstlane(s_pos[s_list[j]]);
            if constexpr (!DX) {
                const float4 *xb = reinterpret_cast<const float4 *>(src + (size_t)b * d.in);
                float4 xv[NX];
#pragma unroll
                for (int c = 0; c < NX; c++) xv[c] = xb[c * 64 + lane];
                float res = 0.0f;  // lane o ends up with output o
                for (int o = 0; o < d.out; o++) {
                    const float4 *wr = reinterpret_cast<const float4 *>(wt + o * ldw);
                    float acc = 0.0f;
#pragma unroll
                    for (int c = 0; c < NX; c++) {
                        const float4 w4 = wr[c * 64 + lane];
                        acc += xv[c].x * w4.x, acc += xv[c].y * w4.y, acc += xv[c].z * w4.z, acc += xv[c].w * w4.w;
                    }
                    const float sum = wave_sum(acc);
                    if (lane == o) res = bt_act_fwd(d.act, sum + bias_l);
                }
                if (lane < d.out) dst[(size_t)b * d.out + lane] = res;
            } else {
                const float g_l = (lane < d.out) ? src[(size_t)b * d.out + lane] : 0.0f;
                float4 acc[NX];
#pragma unroll
                for (int c = 0; c < NX; c++) acc[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                for (int o = 0; o < d.out; o++) {
                    const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_l), o));
                    const float4 *wr = reinterpret_cast<const float4 *>(wt + o * ldw);
#pragma unroll
                    for (int c = 0; c < NX; c++) {
                        const float4 w4 = wr[c * 64 + lane];
                        acc[c].x += w4.x * g, acc[c].y += w4.y * g, acc[c].z += w4.z * g, acc[c].w += w4.w * g;
                    }
                }
                float4 *gx = reinterpret_cast<float4 *>(dst + (size_t)b * d.in);
#pragma unroll
                for (int c = 0; c < NX; c++) gx[c * 64 + lane] = acc[c];
            }
        }
    }
}

// dW / dbias partials of one position range, DM_OH output rows per workgroup: thread i owns input column i (and
// i + 1024) for those rows in registers; gz values are wave-uniform. Partial layout [range][row][in].
__global__ __launch_bounds__(DM_THREADS) void dense_mid_dw_kernel(const float *__restrict__ x,
                                                                  const int32_t *__restrict__ bucket,
                                                                  const float *__restrict__ gz, float *__restrict__ part_w,
                                                                  float *__restrict__ part_b, bt_dense_desc d,
                                                                  const int32_t *__restrict__ perm) {
    __shared__ uint8_t s_bk[DM_RANGE];
    __shared__ int s_list[DM_RANGE], s_wcnt[DM_RANGE / 64], s_pos[DM_RANGE];
    const int rows = d.out * d.buckets;
    const int b0 = blockIdx.x * DM_RANGE, npos = min(DM_RANGE, d.batch - b0);
    const int o0 = blockIdx.y * DM_OH, noh = min(DM_OH, d.out - o0);
    const int i0 = threadIdx.x, i1 = threadIdx.x + DM_THREADS;
    const bool c0 = i0 < d.in, c1 = i1 < d.in;
    if ((int)threadIdx.x < npos) {
        const int pos = perm ? perm[b0 + threadIdx.x] : b0 + (int)threadIdx.x;
        s_pos[threadIdx.x] = pos;
        s_bk[threadIdx.x] = (uint8_t)((d.buckets > 1) ? bucket[pos] : 0);
    }
    __syncthreads();
    float *pw = part_w + (size_t)blockIdx.x * rows * d.in;
    for (int bk = 0; bk < d.buckets; bk++) {
        const int cnt = dm_bucket_list(s_bk, npos, bk, s_list, s_wcnt);
        float a0[DM_OH], a1[DM_OH], ab[DM_OH];
#pragma unroll
        for (int o = 0; o < DM_OH; o++) a0[o] = a1[o] = ab[o] = 0.0f;
        constexpr int U = 8;
        // lanes 0..DM_OH-1 of every wave hold the position's gz values (read back with v_readlane); two batches of
        // U positions are in flight
        auto load = [&](int j0, float(&x0)[U], float(&x1)[U], float(&gl)[U]) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int b = __builtin_amdgcn_readfirstlane(s_pos[s_list[min(j0 + u, max(cnt - 1, 0))]]);
                x0[u] = c0 ? x[(size_t)b * d.in + i0] : 0.0f;
                x1[u] = c1 ? x[(size_t)b * d.in + i1] : 0.0f;
                const int lo = threadIdx.x & 63;
                gl[u] = (lo < noh) ? gz[(size_t)b * d.out + o0 + lo] : 0.0f;
            }
        };
        auto consume = [&](int j0, const float(&x0)[U], const float(&x1)[U], const float(&gl)[U]) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (j0 + u < cnt) {
#pragma unroll
                    for (int o = 0; o < DM_OH; o++) {
                        const float g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gl[u]), o));
                        a0[o] += g * x0[u], a1[o] += g * x1[u], ab[o] += g;
                    }
                }
            }
        };
        if (cnt > 0) {
            float xa0[U], xa1[U], ga[U], xb0[U], xb1[U], gb[U];
            load(0, xa0, xa1, ga);
            for (int j0 = 0; j0 < cnt; j0 += 2 * U) {
                load(j0 + U, xb0, xb1, gb);
                consume(j0, xa0, xa1, ga);
                load(j0 + 2 * U, xa0, xa1, ga);
                consume(j0 + U, xb0, xb1, gb);
            }
        }
#pragma unroll
        for (int o = 0; o < DM_OH; o++) {
            if (o < noh) {
                float *pr = pw + (size_t)(bk * d.out + o0 + o) * d.in;  // zeros for buckets absent from the range
                if (c0) pr[i0] = a0[o];
                if (c1) pr[i1] = a1[o];
                if (threadIdx.x == 0) part_b[(size_t)blockIdx.x * rows + bk * d.out + o0 + o] = ab[o];
            }
        }
    }
}

// dw[row + rows*i] = sum over ranges of part[range][row][i] (ascending), i.e. back to W's column-major layout
__global__ __launch_bounds__(256) void sum_partials_t_kernel(const float *__restrict__ part, int nparts, int rows, int in,
                                                             float *__restrict__ dw) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)rows * in) return;
    const int r = (int)(t / in), i = (int)(t - (size_t)r * in);
    const size_t stride = (size_t)rows * in;
    float s = 0.0f;
    int p = 0;
    for (; p + 8 <= nparts; p += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = part[(size_t)(p + u) * stride + t];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    for (; p < nparts; p++) s += part[(size_t)p * stride + t];
    dw[(size_t)rows * i + r] = s;
}

// ------------------------------------------------------------------ tiny layers (HalfKA nets: 16 -> 32 x 8, 32 -> 1 x 8)
// A few hundred weights per bucket: one thread per output element, the weights come from L1.
bool dense_tiny_ok(const bt_dense_desc *d) {
    return d->in <= 64 && d->out <= 64 && d->out * d->in <= 1024 && d->buckets <= 16;
}

__global__ __launch_bounds__(256) void dense_tiny_fwd_kernel(const float *__restrict__ w, const float *__restrict__ bias,
                                                             const float *__restrict__ x,
                                                             const int32_t *__restrict__ bucket, float *__restrict__ y,
                                                             bt_dense_desc d) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)d.batch * d.out) return;
    const int b = (int)(t / d.out), o = (int)(t - (size_t)b * d.out);
    const int rows = d.out * d.buckets, row = ((d.buckets > 1) ? bucket[b] * d.out : 0) + o;
    const float *xb = x + (size_t)b * d.in;
    float z = 0.0f;
    for (int i = 0; i < d.in; i++) z += w[(size_t)rows * i + row] * xb[i];  // ascending i, like the oracle
    y[t] = bt_act_fwd(d.act, z + bias[row]);
}

__global__ __launch_bounds__(256) void dense_tiny_dx_kernel(const float *__restrict__ w, const float *__restrict__ gz,
                                                            const int32_t *__restrict__ bucket,
                                                            float *__restrict__ grad_x, bt_dense_desc d) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)d.batch * d.in) return;
    const int b = (int)(t / d.in), i = (int)(t - (size_t)b * d.in);
    const int rows = d.out * d.buckets, base = (d.buckets > 1) ? bucket[b] * d.out : 0;
    const float *wc = w + (size_t)rows * i + base, *g = gz + (size_t)b * d.out;
    float s = 0.0f;
    for (int o = 0; o < d.out; o++) s += wc[o] * g[o];
    grad_x[t] = s;
}

// thread (o, i) owns dW[bucket*out + o][i] for every bucket in registers; the position's bucket is wave-uniform, so
// the select is a scalar branch. Partial layout [range][row][in], dbias partial from the i == 0 threads.
template <int NB>
__global__ __launch_bounds__(1024) void dense_tiny_dw_kernel(const float *__restrict__ x,
                                                             const int32_t *__restrict__ bucket,
                                                             const float *__restrict__ gz, float *__restrict__ part_w,
                                                             float *__restrict__ part_b, bt_dense_desc d) {
    __shared__ int s_bk[DM_RANGE];
    const int rows = d.out * d.buckets;
    const int b0 = blockIdx.x * DM_RANGE, npos = min(DM_RANGE, d.batch - b0);
    const int o = threadIdx.x / d.in, i = threadIdx.x - o * d.in;  // blockDim.x == out * in
    for (int p = threadIdx.x; p < npos; p += blockDim.x) s_bk[p] = (d.buckets > 1) ? bucket[b0 + p] : 0;  // blockDim may be < DM_RANGE
    __syncthreads();
    float acc[NB], ab[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) acc[k] = ab[k] = 0.0f;
    constexpr int U = 8;
    for (int p0 = 0; p0 < npos; p0 += U) {
        float xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int b = b0 + min(p0 + u, npos - 1);
            xv[u] = x[(size_t)b * d.in + i], gv[u] = gz[(size_t)b * d.out + o];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (p0 + u < npos) {
                const int bk = __builtin_amdgcn_readfirstlane(s_bk[p0 + u]);
#pragma unroll
                for (int k = 0; k < NB; k++)
                    if (k == bk) acc[k] += gv[u] * xv[u], ab[k] += gv[u];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB; k++) {
        if (k < d.buckets) {
            part_w[(size_t)blockIdx.x * rows * d.in + (size_t)(k * d.out + o) * d.in + i] = acc[k];
            if (i == 0) part_b[(size_t)blockIdx.x * rows + k * d.out + o] = ab[k];
        }
    }
}

bool dense_fwd_small_ok(const bt_dense_desc *d) {
    const int rows = d->out * d->buckets;
    return d->in % 256 == 0 && d->in / 256 <= 8 && (size_t)rows * (d->in + 4) * sizeof(float) <= 96 * 1024;
}
bool dense_bwd_small_ok(const bt_dense_desc *d) { return d->out == 1 && d->buckets <= 16 && d->in % 4 == 0; }

template <typename K>
int dense_set_lds(K kernel, size_t bytes) {
    return bt_set_max_lds(reinterpret_cast<const void *>(kernel), bytes);
}

int dense_fwd_small_launch(hipStream_t s, const bt_dense_desc *d, const float *w, const float *bias, const float *x,
                           const int32_t *bucket, float *y, const LossArgs *la, int *n_partials) {
    const int rows = d->out * d->buckets;
    const size_t lds = (size_t)rows * (d->in + 4) * sizeof(float);
    int nwg = bt_div_up(d->batch, 64);
    if (nwg > 256) nwg = 256;
    const int ppw = bt_div_up(d->batch, nwg);
    nwg = bt_div_up(d->batch, ppw);
    if (n_partials) *n_partials = nwg;
    const LossArgs none{nullptr, nullptr, nullptr, nullptr, nullptr};
#define DFS(NX)                                                                                                   \
    case NX:                                                                                                      \
        if (la) {                                                                                                 \
            if (int rc = dense_set_lds(dense_fwd_small<NX, true>, lds)) return rc;                                \
            BT_LAUNCH(BTK_DENSE_FWD_SMALL, s, (dense_fwd_small<NX, true>), dim3(nwg), dim3(DF_THREADS), lds, w, bias, x, \
                      bucket, y, *d, ppw, *la);                                                                   \
        } else {                                                                                                  \
            if (int rc = dense_set_lds(dense_fwd_small<NX, false>, lds)) return rc;                               \
            BT_LAUNCH(BTK_DENSE_FWD_SMALL, s, (dense_fwd_small<NX, false>), dim3(nwg), dim3(DF_THREADS), lds, w, bias, x, \
                      bucket, y, *d, ppw, none);                                                                  \
        }                                                                                                         \
        break;
    switch (d->in / 256) {
        DFS(1) DFS(2) DFS(3) DFS(4) DFS(5) DFS(6) DFS(7) DFS(8)
    }
#undef DFS
    BT_LAUNCH_CHECK();
    return BT_OK;
}

int dense_bwd_small_launch(hipStream_t s, const bt_dense_desc *d, const float *w, const float *x, const int32_t *bucket,
                           const float *y, const float *grad_y, float *dw, float *dbias, float *grad_x, float *gz_out,
                           const float *loss_partials, int n_loss_partials, float *loss_sum, float *part_w,
                           float *part_b, int nr, bool light = false) {
    const int rows = d->out * d->buckets;
    dim3 grid(bt_div_up(d->in / 4, DBS_CT), nr), block(DBS_CT * DBS_SUB);
    if (light && !grad_x && !gz_out && rows <= 8) {
#define DBL(R)                                                                                                    \
    BT_LAUNCH(BTK_DENSE_BWD_SMALL, s, (dense_bwd_small_light<R>), grid, dim3(DBS_CT), 0, w, x, bucket, y, grad_y, part_w, \
              part_b, *d, DBS_RANGE, loss_partials, n_loss_partials, loss_sum)
        if (rows <= 1) DBL(1);
        else if (rows <= 2) DBL(2);
        else if (rows <= 4) DBL(4);
        else DBL(8);
#undef DBL
        BT_LAUNCH_CHECK();
        const size_t nwl = (size_t)rows * d->in;
        if (int rc = sum_partials_launch(s, part_w, nr, nwl, dw, true)) return rc;
        return sum_partials_launch(s, part_b, nr, (size_t)rows, dbias, true);
    }
#define DBS(R)                                                                                                   \
    do {                                                                                                         \
        if (grad_x)                                                                                              \
            BT_LAUNCH(BTK_DENSE_BWD_SMALL, s, (dense_bwd_small<R, true>), grid, block, 0, w, x, bucket, y, grad_y, \
                      part_w, part_b, grad_x, *d, DBS_RANGE, gz_out, loss_partials, n_loss_partials, loss_sum);  \
        else                                                                                                     \
            BT_LAUNCH(BTK_DENSE_BWD_SMALL, s, (dense_bwd_small<R, false>), grid, block, 0, w, x, bucket, y, grad_y, \
                      part_w, part_b, grad_x, *d, DBS_RANGE, gz_out, loss_partials, n_loss_partials, loss_sum);  \
    } while (0)
    if (rows <= 1) DBS(1);
    else if (rows <= 2) DBS(2);
    else if (rows <= 4) DBS(4);
    else if (rows <= 8) DBS(8);
    else DBS(16);
#undef DBS
    BT_LAUNCH_CHECK();
    const size_t nw = (size_t)rows * d->in;
    if (int rc = sum_partials_launch(s, part_w, nr, nw, dw)) return rc;
    if (int rc = sum_partials_launch(s, part_b, nr, (size_t)rows, dbias)) return rc;
    return BT_OK;
}

// forward (dx = false: src = x, dst = y) or dX (dx = true: src = gz, dst = grad_x) of a mid-size bucketed layer
int dense_mid_launch(hipStream_t s, const bt_dense_desc *d, const float *w, const float *bias, const float *src,
                     const int32_t *bucket, float *dst, bool dx) {
    const size_t lds = (size_t)d->out * (d->in + 4) * sizeof(float);
    dim3 grid(bt_div_up(d->batch, DM_CHUNK)), block(DM_THREADS);
    const int32_t *perm = d->buckets > 1 ? g_dense_tab.perm : nullptr;
#define DML(NX)                                                                                                      \
    case NX:                                                                                                         \
        if (dx) {                                                                                                    \
            if (int rc = dense_set_lds(dense_mid_kernel<NX, true>, lds)) return rc;                                  \
            BT_LAUNCH(BTK_DENSE_BWD_DX, s, (dense_mid_kernel<NX, true>), grid, block, lds, w, bias, src, bucket, dst, *d, \
                      perm);                                                                                         \
        } else {                                                                                                     \
            if (int rc = dense_set_lds(dense_mid_kernel<NX, false>, lds)) return rc;                                 \
            BT_LAUNCH(BTK_DENSE_FWD, s, (dense_mid_kernel<NX, false>), grid, block, lds, w, bias, src, bucket, dst, *d,   \
                      perm);                                                                                         \
        }                                                                                                            \
        break;
    switch (d->in / 256) {
        DML(1) DML(2) DML(3) DML(4) DML(5) DML(6) DML(7) DML(8)
    }
#undef DML
    BT_LAUNCH_CHECK();
    return BT_OK;
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

// Scratch of bt_dense_backward_static (include/bullet_amd.h): BT_STATIC_SCRATCH_BYTES of device memory that every GPU
// the library runs on holds until the process ends. Host code never passes `scratch` itself to a kernel or a memset:
// only the per-device address obtained with hipGetSymbolAddress.
static __device__ __align__(256) unsigned char scratch[BT_STATIC_SCRATCH_BYTES];
static constexpr int kMaxScratchDevices = 64;
static void *g_scratch_ptr[kMaxScratchDevices] = {nullptr};
static std::mutex g_scratch_mutex;

// ---- internal entry points used by net.hip (declared in common.h)
// bucket-sorted position order for the mid-size bucketed layers (see g_dense_tab.perm)
bool bt_internal_dense_wants_perm(const bt_dense_desc *d) {
    return dense_validate(d, "dense_wants_perm") == BT_OK && d->buckets > 1 && d->buckets <= 16 && !dense_fwd_small_ok(d) &&
           dense_mid_ok(d);
}
// layout of the permutation buffer: perm [batch] | items int4[batch / DX_CH + nb] | bucket_items [nb + 1] | counts [4]
// | block histograms [ceil(batch / BPC_BLOCK)][nb]
static DensePermTab dense_perm_tab(void *buf, int batch, int nb) {
    DensePermTab t;
    char *p = static_cast<char *>(buf);
    t.max_items = bt_div_up(batch, DX_CH) + nb;
    t.perm = reinterpret_cast<const int32_t *>(p);
    p += bt_align((size_t)batch * sizeof(int32_t));
    t.items = reinterpret_cast<const int4 *>(p);
    p += bt_align((size_t)t.max_items * sizeof(int4));
    t.bucket_items = reinterpret_cast<const int *>(p);
    p += bt_align((size_t)(nb + 1) * sizeof(int));
    t.counts = reinterpret_cast<const int *>(p);
    p += bt_align(4 * sizeof(int));
    t.blk_cnt = reinterpret_cast<int *>(p);
    return t;
}
size_t bt_internal_bucket_perm_bytes(int batch, int nb) {
    return bt_align((size_t)batch * sizeof(int32_t)) + bt_align((size_t)(bt_div_up(batch, DX_CH) + nb) * sizeof(int4)) +
           bt_align((size_t)(nb + 1) * sizeof(int)) + bt_align(4 * sizeof(int)) +
           bt_align((size_t)bt_div_up(batch, BPC_BLOCK) * nb * sizeof(int));
}
int bt_internal_bucket_perm(void *stream, const int32_t *bucket, int batch, int nb, void *buf) {
    BT_REQUIRE(bucket && buf && batch > 0 && batch <= 64 * 1024 && nb > 1 && nb <= 16, "bucket_perm: bad arguments");
    const DensePermTab t = dense_perm_tab(buf, batch, nb);
    const int nblk = bt_div_up(batch, BPC_BLOCK);
    BT_LAUNCH(BTK_DENSE_FWD, bt_stream(stream), bucket_count_kernel, dim3(nblk), dim3(BPC_BLOCK), 0, bucket, batch, nb,
              t.blk_cnt);
    BT_LAUNCH_CHECK();
    BT_LAUNCH(BTK_DENSE_FWD, bt_stream(stream), bucket_place_kernel, dim3(nblk), dim3(BPC_BLOCK), 0, bucket, batch, nb,
              t.blk_cnt, nblk, const_cast<int32_t *>(t.perm), const_cast<int4 *>(t.items),
              const_cast<int *>(t.bucket_items), const_cast<int *>(t.counts));
    BT_LAUNCH_CHECK();
    return BT_OK;
}
// fused tail (see dense_tail_kernel): q1 = the layer before the last, q2 = the last layer
bool bt_internal_dense_tail_ok(const bt_dense_desc *q1, const bt_dense_desc *q2) {
    return q1 && q2 && q1->buckets > 1 && q1->buckets <= 16 && q2->buckets == q1->buckets && q2->out == 1 &&
           q2->in == q1->out && q1->in >= 1 && q1->in <= 64 && q1->out >= 1 && q1->out <= 64 && q1->batch == q2->batch;
}
size_t bt_internal_dense_tail_workspace_bytes(const bt_dense_desc *q1) {
    return (size_t)(bt_div_up(q1->batch, DX_CH) + q1->buckets) * tail_part_floats(q1->in, q1->out) * sizeof(float);
}
int bt_internal_dense_tail(void *stream, const bt_dense_desc *q1, const bt_dense_desc *q2, const float *w2, const float *b2,
                           const float *w3, const float *b3, const float *a1, const float *targets, const float *ew,
                           float *y, float *ga1, float *dw2, float *db2, float *dw3, float *db3, float *loss_sum,
                           void *workspace, size_t workspace_bytes, int act_before) {
    BT_REQUIRE(bt_internal_dense_tail_ok(q1, q2), "dense_tail: shape not supported");
    BT_REQUIRE(g_dense_tab.items, "dense_tail: no bucket permutation installed");
    BT_REQUIRE(w2 && b2 && w3 && b3 && a1 && targets && y && ga1 && dw2 && db2 && dw3 && db3 && loss_sum && workspace,
               "dense_tail: null pointer");
    BT_REQUIRE(workspace_bytes >= bt_internal_dense_tail_workspace_bytes(q1), "dense_tail: workspace too small");
    const DensePermTab tab = g_dense_tab;
    const int n1 = q1->in, n2 = q1->out, nb = q1->buckets;
    TailArgs A{w2, b2, w3, b3, a1, targets, ew, y, ga1, static_cast<float *>(workspace), n1, n2, nb, q1->act, q2->act,
               act_before};
    const size_t lds = ((size_t)n2 * n1 + 2 * n2 + DX_CH * (n1 + 1) + 2 * DX_CH * (n2 + 1) + 2 * DX_CH) * sizeof(float);
    hipStream_t s = bt_stream(stream);
    if (int rc = dense_set_lds(dense_tail_kernel, lds)) return rc;
    dense_fork_on(BTK_DENSE_BWD_SMALL);
    BT_LAUNCH(BTK_DENSE_BWD_SMALL, s, dense_tail_kernel, dim3(tab.max_items), dim3(TAIL_THREADS), lds, A, tab);
    BT_LAUNCH_CHECK();
    hipStream_t s2;
    if (int rc = dense_side_join_fork(s, &s2, true)) return rc;
    g_dense_side_behind = g_dense_side != nullptr;  // ga1 (the next layer's grad_y) is this kernel's
    BT_LAUNCH(BTK_SUM_PARTIALS, s2, dense_tail_sum, dim3(bt_div_up(tail_part_floats(n1, n2) - 1, 32), nb + 1), dim3(256), 0,
              A.part, n1, n2, nb, dw2, db2, dw3, db3, loss_sum, tab);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

void bt_internal_dense_set_gz_ready(bool on) { g_dense_gy_is_gz = on; }
void bt_internal_dense_set_side(void *side_stream, hipEvent_t fork) {
    g_dense_side = (side_stream && fork) ? bt_stream(side_stream) : nullptr;
    g_dense_fork = g_dense_side ? fork : nullptr;
    g_dense_side_behind = false;
}
bool bt_internal_dense_takes_gz(const bt_dense_desc *d) {
    return dense_validate(d, "dense_takes_gz") == BT_OK && !dense_bwd_small_ok(d) && (dense_mid_ok(d) || dense_tiny_ok(d));
}

void bt_internal_dense_set_perm(void *buf, int batch, int nb) {
    g_dense_tab = buf ? dense_perm_tab(buf, batch, nb) : DensePermTab();
}

bool bt_internal_dense_head_ok(const bt_dense_desc *d) { return dense_validate(d, "dense_head_ok") == BT_OK && dense_bwd_small_ok(d); }

int bt_internal_dense_backward_head(void *stream, const bt_dense_desc *d, const float *w, const float *x,
                                    const int32_t *bucket, const float *y, const float *grad_y, float *dw, float *dbias,
                                    float *grad_x, float *gz_out, const float *loss_partials, int n_loss_partials,
                                    float *loss_sum, void *workspace, size_t workspace_bytes, bool light) {
    if (int rc = dense_validate(d, "bt_dense_backward")) return rc;
    BT_REQUIRE(dense_bwd_small_ok(d), "bt_dense_backward(head): shape not supported");
    BT_REQUIRE(w && x && y && grad_y && dw && dbias && workspace, "bt_dense_backward: null pointer");
    BT_REQUIRE(workspace_bytes >= bt_dense_backward_workspace_bytes(d), "bt_dense_backward: workspace too small");
    const int rows = d->out * d->buckets;
    const int nr = bt_div_up(d->batch, DBS_RANGE);
    float *part_w = static_cast<float *>(workspace);
    float *part_b = reinterpret_cast<float *>(static_cast<char *>(workspace) +
                                              bt_align((size_t)nr * rows * d->in * sizeof(float)));
    return dense_bwd_small_launch(bt_stream(stream), d, w, x, bucket, y, grad_y, dw, dbias, grad_x, gz_out, loss_partials,
                                  n_loss_partials, loss_sum, part_w, part_b, nr, light);
}

bool bt_internal_dense_forward_loss_ok(const bt_dense_desc *d) {
    return dense_validate(d, "dense_forward_loss_ok") == BT_OK && d->out == 1 && dense_fwd_small_ok(d);
}

int bt_internal_dense_forward_loss(void *stream, const bt_dense_desc *d, const float *w, const float *bias,
                                   const float *x, const int32_t *bucket, const float *targets,
                                   const float *entry_weights, float *y, float *grad_y, float *gz_out,
                                   float *loss_partials, int *n_loss_partials) {
    if (int rc = dense_validate(d, "bt_dense_forward")) return rc;
    BT_REQUIRE(bt_internal_dense_forward_loss_ok(d), "bt_dense_forward(loss): shape not supported");
    BT_REQUIRE(w && bias && x && y && targets && grad_y && loss_partials, "bt_dense_forward(loss): null pointer");
    const LossArgs la{targets, entry_weights, grad_y, loss_partials, gz_out};
    return dense_fwd_small_launch(bt_stream(stream), d, w, bias, x, bucket, y, &la, n_loss_partials);
}

int bt_internal_head_finish(void *stream, const bt_dense_desc *d, const float *zpart, int nparts, const float *bias,
                            const int32_t *bucket, const float *targets, const float *entry_weights, float *y,
                            float *grad_y, float *gz_out, float *loss_partials, int *n_loss_partials, float *gz_parts) {
    if (int rc = dense_validate(d, "bt_dense_forward")) return rc;
    BT_REQUIRE(d->out == 1 && zpart && bias && targets && y && grad_y && loss_partials && n_loss_partials,
               "bt_dense_forward(head finish): bad arguments");
    BT_REQUIRE(d->buckets == 1 || bucket, "bt_dense_forward: bucketed layer needs bucket indices");
    const int nwg = bt_div_up(d->batch, HF_THREADS);
    BT_REQUIRE(nwg <= 1024, "bt_dense_forward(head finish): batch too large for the loss-partial buffer");
    const LossArgs la{targets, entry_weights, grad_y, loss_partials, gz_out, gz_parts};
    BT_LAUNCH(BTK_DENSE_FWD_SMALL, bt_stream(stream), head_finish_kernel, dim3(nwg), dim3(HF_THREADS), 0, zpart, nparts,
              bias, bucket, y, *d, la);
    BT_LAUNCH_CHECK();
    *n_loss_partials = nwg;
    return BT_OK;
}

extern "C" {

int bt_dense_forward(void *stream, const bt_dense_desc *d, const float *w, const float *bias, const float *x,
                     const int32_t *bucket, float *y) {
    if (int rc = dense_validate(d, "bt_dense_forward")) return rc;
    BT_REQUIRE(w && bias && x && y, "bt_dense_forward: null pointer");
    BT_REQUIRE(d->buckets == 1 || bucket, "bt_dense_forward: bucketed layer needs bucket indices");
    if (dense_fwd_small_ok(d)) return dense_fwd_small_launch(bt_stream(stream), d, w, bias, x, bucket, y, nullptr, nullptr);
    if (dense_mfma_ok(d) && g_dense_tab.items && !dense_fwd_small_ok(d)) {
        const size_t lds = (size_t)16 * (d->in + 4) * sizeof(float);
        if (int rc = dense_set_lds(dense_mfma_fwd, lds)) return rc;
        BT_LAUNCH(BTK_DENSE_FWD, bt_stream(stream), dense_mfma_fwd, dim3(g_dense_tab.max_items), dim3(MF_THREADS), lds, w, bias, x, y,
                  *d, g_dense_tab);
        BT_LAUNCH_CHECK();
        return BT_OK;
    }
    if (dense_mid_ok(d)) return dense_mid_launch(bt_stream(stream), d, w, bias, x, bucket, y, false);
    if (dense_tiny_ok(d)) {
        BT_LAUNCH(BTK_DENSE_FWD, bt_stream(stream), dense_tiny_fwd_kernel, dim3(bt_div_up((long long)d->batch * d->out, 256)),
                  dim3(256), 0, w, bias, x, bucket, y, *d);
        BT_LAUNCH_CHECK();
        return BT_OK;
    }
    BT_LAUNCH(BTK_DENSE_FWD, bt_stream(stream), dense_fwd_kernel, dim3(bt_div_up(d->batch, 4)), dim3(256), 0, w, bias, x,
                       bucket, y, *d);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

size_t bt_dense_backward_workspace_bytes(const bt_dense_desc *d) {
    if (dense_validate(d, "bt_dense_backward_workspace_bytes")) return 0;
    const int rows = d->out * d->buckets;
    const bool mid = !dense_bwd_small_ok(d) && (dense_mid_ok(d) || dense_tiny_ok(d));
    const int nr = dense_bwd_small_ok(d) ? bt_div_up(d->batch, DBS_RANGE) : bt_div_up(d->batch, mid ? DM_RANGE : DW_RANGE);
    return bt_align((size_t)nr * rows * d->in * sizeof(float)) + bt_align((size_t)nr * rows * sizeof(float)) +
           (mid ? bt_align((size_t)d->batch * d->out * sizeof(float)) : 0);  // mid path: + gz [B][out]
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
    const bool small = dense_bwd_small_ok(d);
    const int nr = small ? bt_div_up(d->batch, DBS_RANGE) : bt_div_up(d->batch, DW_RANGE);
    float *part_w = static_cast<float *>(workspace);
    float *part_b = reinterpret_cast<float *>(static_cast<char *>(workspace) +
                                              bt_align((size_t)nr * rows * d->in * sizeof(float)));
    if (small)
        return dense_bwd_small_launch(s, d, w, x, bucket, y, grad_y, dw, dbias, grad_x, nullptr, nullptr, 0, nullptr, part_w,
                                      part_b, nr);
    if (dense_mid_ok(d) || dense_tiny_ok(d)) {
        const bool tiny = !dense_mid_ok(d);
        const int nrm = bt_div_up(d->batch, DM_RANGE);
        char *ws = static_cast<char *>(workspace);
        float *pw = reinterpret_cast<float *>(ws);
        float *pb = reinterpret_cast<float *>(ws + bt_align((size_t)nrm * rows * d->in * sizeof(float)));
        float *gz = reinterpret_cast<float *>(ws + bt_align((size_t)nrm * rows * d->in * sizeof(float)) +
                                              bt_align((size_t)nrm * rows * sizeof(float)));
        const size_t ngz = (size_t)d->batch * d->out;
        const bool mfma_path = !tiny && dense_mfma_ok(d) && g_dense_tab.items &&
                               (size_t)g_dense_tab.max_items * 16 * d->in <= (size_t)nrm * rows * d->in;
        const bool behind = g_dense_side_behind;
        g_dense_side_behind = false;
        if (g_dense_gy_is_gz) {
            gz = const_cast<float *>(grad_y);  // (read only below)
        } else {
            if (mfma_path) dense_fork_on(BTK_DENSE_BWD_DX);
            BT_LAUNCH(BTK_DENSE_BWD_DX, s, dense_gz_kernel, dim3(bt_div_up((long long)ngz, 256)), dim3(256), 0, y, grad_y, gz,
                      ngz, d->act);
            BT_LAUNCH_CHECK();
        }
        if (mfma_path) {
            const DensePermTab tab = g_dense_tab;
            hipStream_t s2 = s;  // dW and its sum: beside dX (and whatever follows it) when a side stream is set
            if (g_dense_gy_is_gz && behind) s2 = g_dense_side;  // (the side stream waits for grad_y's kernel already)
            else if (int rc = dense_side_join_fork(s, &s2, !g_dense_gy_is_gz)) return rc;
            if (grad_x) {
                BT_LAUNCH(BTK_DENSE_BWD_DX, s, dense_mfma_dx, dim3(tab.max_items), dim3(256), 0, w, gz, grad_x, *d, tab);
                BT_LAUNCH_CHECK();
            }
            const int tw = bt_div_up(d->in / 64, 4);  // 64-column blocks per wave: up to 8 (in <= 2048)
#define MDW(TW)                                                                                                      \
    case TW:                                                                                                         \
        BT_LAUNCH(BTK_DENSE_BWD_DW, s2, dense_mfma_dw<TW>, dim3(tab.max_items), dim3(256), 0, x, gz, pw, pb, *d, tab); \
        break;
            switch (tw) { MDW(1) MDW(2) MDW(3) MDW(4) MDW(5) MDW(6) MDW(7) MDW(8) }
#undef MDW
            BT_LAUNCH_CHECK();
            BT_LAUNCH(BTK_SUM_PARTIALS, s2, dense_mfma_sum, dim3(bt_div_up(16 * d->in + 16, 32), d->buckets), dim3(256), 0,
                      pw, pb, dw, dbias, *d, tab);
            BT_LAUNCH_CHECK();
            return BT_OK;
        }
        if (tiny) {
            if (grad_x) {
                BT_LAUNCH(BTK_DENSE_BWD_DX, s, dense_tiny_dx_kernel, dim3(bt_div_up((long long)d->batch * d->in, 256)),
                          dim3(256), 0, w, gz, bucket, grad_x, *d);
                BT_LAUNCH_CHECK();
            }
#define DTW(NB) BT_LAUNCH(BTK_DENSE_BWD_DW, s, dense_tiny_dw_kernel<NB>, dim3(nrm), dim3(d->out * d->in), 0, x, bucket, gz, pw, pb, *d)
            if (d->buckets <= 1) DTW(1);
            else if (d->buckets <= 2) DTW(2);
            else if (d->buckets <= 4) DTW(4);
            else if (d->buckets <= 8) DTW(8);
            else DTW(16);
#undef DTW
            BT_LAUNCH_CHECK();
        } else {
            if (grad_x)
                if (int rc = dense_mid_launch(s, d, w, nullptr, gz, bucket, grad_x, true)) return rc;
            BT_LAUNCH(BTK_DENSE_BWD_DW, s, dense_mid_dw_kernel, dim3(nrm, bt_div_up(d->out, DM_OH)), dim3(DM_THREADS), 0, x,
                      bucket, gz, pw, pb, *d, d->buckets > 1 ? g_dense_tab.perm : nullptr);
            BT_LAUNCH_CHECK();
        }
        BT_LAUNCH(BTK_SUM_PARTIALS, s, sum_partials_t_kernel, dim3(bt_div_up((long long)rows * d->in, 256)), dim3(256), 0, pw,
                  nrm, rows, d->in, dw);
        BT_LAUNCH_CHECK();
        return sum_partials_launch(s, pb, nrm, (size_t)rows, dbias);
    }
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
    if (int rc = sum_partials_launch(s, part_w, nr, nw, dw)) return rc;
    return sum_partials_launch(s, part_b, nr, (size_t)rows, dbias);
}

// bt_dense_backward for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid
// hipMalloc): the partial-sum workspace is the file-scope device array `scratch` instead of a caller buffer. The
// array's name is only a host-side stand-in; its device address is looked up once per device (each GPU holds its
// own copy at its own address) and that pointer is what the kernels get. No hipMalloc / hipFree on this path.
int bt_dense_backward_static(void *stream, const bt_dense_desc *d, const float *w, const float *x, const int32_t *bucket,
                             const float *y, const float *grad_y, float *dw, float *dbias, float *grad_x) {
    if (int rc = dense_validate(d, "bt_dense_backward_static")) return rc;
    const size_t need = bt_dense_backward_workspace_bytes(d);
    BT_REQUIRE(need > 0 && need <= (size_t)BT_STATIC_SCRATCH_BYTES,
               "bt_dense_backward_static: this shape needs %zu bytes of scratch, the static buffer holds %zu", need,
               (size_t)BT_STATIC_SCRATCH_BYTES);
    int dev = -1;
    BT_HIP(hipGetDevice(&dev));
    BT_REQUIRE(dev >= 0 && dev < kMaxScratchDevices, "bt_dense_backward_static: device ordinal %d out of range", dev);
    void *ptr = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[dev]) {
            void *p = nullptr;
            BT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            BT_REQUIRE(p, "bt_dense_backward_static: hipGetSymbolAddress returned a null address");
            g_scratch_ptr[dev] = p;
        }
        ptr = g_scratch_ptr[dev];
    }
    // every partial the reduction kernels read is written by the kernels of this same call, so nothing has to be
    // cleared between calls; calls that overlap on one GPU (two streams, two threads) would race on the array
    return bt_dense_backward(stream, d, w, x, bucket, y, grad_y, dw, dbias, grad_x, ptr, BT_STATIC_SCRATCH_BYTES);
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

int bt_loss(void *stream, int kind, int batch, int width, const float *y, const float *targets,
            const float *entry_weights, float power, float *loss_sum, float *grad_y) {
    if (kind == BT_LOSS_SIGMOID_MSE) {
        BT_REQUIRE(width == 1, "bt_loss: sigmoid losses take one output per position");
        return bt_sigmoid_mse(stream, batch, y, targets, entry_weights, loss_sum, grad_y);
    }
    BT_REQUIRE(kind == BT_LOSS_SIGMOID_POWER || kind == BT_LOSS_SOFTMAX_CE, "bt_loss: unknown loss kind %d", kind);
    BT_REQUIRE(batch > 0 && width > 0, "bt_loss: Mismatched sizes! (batch or width <= 0)");
    BT_REQUIRE(kind != BT_LOSS_SIGMOID_POWER || width == 1, "bt_loss: sigmoid losses take one output per position");
    BT_REQUIRE(y && targets && loss_sum && grad_y, "bt_loss: null pointer");
    BT_LAUNCH(BTK_SIGMOID_MSE, bt_stream(stream), loss_general_kernel, dim3(1), dim3(1024), 0, kind, batch, width, y,
              targets, entry_weights, power, loss_sum, grad_y);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

int bt_weight_clip(void *stream, float *w, size_t n, float min_w, float max_w) {
    BT_REQUIRE(w && n > 0, "bt_weight_clip: null/empty buffer");
    BT_LAUNCH(BTK_ADAMW, bt_stream(stream), weight_clip_kernel, dim3(bt_div_up((long long)n, 256)), dim3(256), 0, w, n,
              min_w, max_w);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

int bt_clip_pass_through(void *stream, const float *x, float *y, size_t n, float min_v, float max_v) {
    BT_REQUIRE(x && y && n > 0, "bt_clip_pass_through: null/empty buffer");
    BT_LAUNCH(BTK_ADAMW, bt_stream(stream), clip_pass_through_kernel, dim3(bt_div_up((long long)n, 256)), dim3(256), 0, x,
              y, n, min_v, max_v);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

int bt_faux_quantise(void *stream, const float *x, float *y, size_t n, float value, int round) {
    BT_REQUIRE(x && y && n > 0, "bt_faux_quantise: null/empty buffer");
    BT_REQUIRE(value != 0.0f, "bt_faux_quantise: value must not be zero");
    BT_LAUNCH(BTK_ADAMW, bt_stream(stream), faux_quantise_kernel, dim3(bt_div_up((long long)n, 256)), dim3(256), 0, x, y, n,
              value, round);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

int bt_weight_decay(void *stream, float *w, size_t n, float decay, const float *lr_dev) {
    BT_REQUIRE(w && n > 0 && lr_dev, "bt_weight_decay: null/empty buffer");
    BT_LAUNCH(BTK_ADAMW, bt_stream(stream), weight_decay_kernel, dim3(bt_div_up((long long)n, 256)), dim3(256), 0, w, n,
              decay, lr_dev);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

static int adam_launch(void *stream, int n_tensors, const bt_adamw_tensor *tensors, const float *adj_dev,
                       const float *lr_dev, int radam, float step_size, int denom);

int bt_adamw(void *stream, int n_tensors, const bt_adamw_tensor *tensors, const float *adj_dev, const float *lr_dev) {
    return adam_launch(stream, n_tensors, tensors, adj_dev, lr_dev, 0, 1.0f, 1);
}

int bt_radam(void *stream, int n_tensors, const bt_adamw_tensor *tensors, const float *adj_dev, const float *lr_dev,
             int istep, float thr) {
    BT_REQUIRE(istep >= 1, "bt_radam: step must be >= 1");
    BT_REQUIRE(n_tensors > 0 && tensors, "bt_radam: null pointer");
    // RAdam::update, radam.rs:283-302 (f32 arithmetic; betas of the first tensor, as all tensors share RAdamParams)
    const float beta1 = tensors[0].beta1, beta2 = tensors[0].beta2, step = (float)istep;
    const float beta2_t = powf(beta2, step);
    const float n_sma_max = 2.0f / (1.0f - beta2) - 1.0f;
    const float n_sma = n_sma_max - 2.0f * step * beta2_t / (1.0f - beta2_t);
    const float dn = 1.0f - powf(beta1, step);
    float step_size;
    if (n_sma > thr) {
        const float p1 = (n_sma - 4.0f) / (n_sma_max - 4.0f), p2 = (n_sma - 2.0f) / n_sma, p3 = n_sma_max / (n_sma_max - 2.0f);
        step_size = sqrtf((1.0f - beta2_t) * p1 * p2 * p3) / dn;
    } else {
        step_size = 1.0f / dn;
    }
    return adam_launch(stream, n_tensors, tensors, adj_dev, lr_dev, 1, step_size, n_sma > thr ? 1 : 0);
}

int bt_lookahead(void *stream, int n_tensors, float *const *params, float *const *slow, const size_t *counts, float alpha) {
    BT_REQUIRE(n_tensors > 0 && params && slow && counts, "bt_lookahead: null pointer");
    for (int i = 0; i < n_tensors; i++) {
        BT_REQUIRE(params[i] && slow[i] && counts[i] > 0, "bt_lookahead: tensor %d has null/empty buffers", i);
        BT_LAUNCH(BTK_ADAMW, bt_stream(stream), lookahead_kernel, dim3(bt_div_up((long long)counts[i], 256)), dim3(256), 0,
                  params[i], slow[i], counts[i], alpha);
        BT_LAUNCH_CHECK();
    }
    return BT_OK;
}

static int adam_launch(void *stream, int n_tensors, const bt_adamw_tensor *tensors, const float *adj_dev,
                       const float *lr_dev, int radam, float step_size, int denom) {
    BT_REQUIRE(n_tensors > 0 && n_tensors <= BT_ADAMW_MAX_TENSORS, "bt_adamw: n_tensors out of range");
    BT_REQUIRE(tensors && adj_dev && lr_dev, "bt_adamw: null pointer");
    AdamTable tab;
    tab.n_tensors = n_tensors;
    tab.radam = radam, tab.step_size = step_size, tab.denom = denom;
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
