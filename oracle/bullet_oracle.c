/*
 * bullet_oracle.c -- CPU restatement of the jw1912/bullet NNUE training hot path.
 * TEST INFRASTRUCTURE ONLY (see bullet_oracle.h).  Plain C, scalar, single thread,
 * fp32 arithmetic in the reference's operation order.  Build: oracle/Makefile
 * (-O2 -ffp-contract=off so no FMA contraction changes the rounding).
 *
 * Citations are file:line relative to /root/reference.
 */
#include "bullet_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ====================================================================== ops */

/* sparse.rs:86-119 (evaluate) + write.rs:229-233 (device break rule) */
void orc_sparse_matmul_fwd(const float *w, const int32_t *idx, float *out, int batch, int rows, int cols, int stride,
                           int offset, int nnz, int brk) {
    for (int bi = 0; bi < batch; bi++) {
        float *o = out + (size_t)rows * bi;
        for (int ri = 0; ri < rows; ri++) o[ri] = 0.0f;
        /* per output element the additions happen in index order ni = 0..nnz-1
         * starting from 0.0f, exactly as the (bi, ri, ni) loop nest of the reference;
         * interchanging ri and ni does not alter any element's summation order. */
        for (int ni = 0; ni < nnz; ni++) {
            int32_t j = idx[(size_t)nnz * bi + ni];
            if (j < 0 || j >= cols) {
                if (brk) break;
                continue;
            }
            const float *col = w + (size_t)stride * j + offset;
            for (int ri = 0; ri < rows; ri++) o[ri] = o[ri] + col[ri];
        }
    }
}

/* sparse.rs:151-187 */
void orc_sparse_matmul_bwd(const float *g, const int32_t *idx, float *dw, int batch, int rows, int cols, int stride,
                           int offset, int nnz, int brk) {
    for (size_t i = 0; i < (size_t)stride * cols; i++) dw[i] = 0.0f;
    /* reference order for one dw element: (bi, ni) ascending; the ri loop is independent */
    for (int bi = 0; bi < batch; bi++) {
        const float *gb = g + (size_t)rows * bi;
        for (int ni = 0; ni < nnz; ni++) {
            int32_t j = idx[(size_t)nnz * bi + ni];
            if (j < 0 || j >= cols) {
                if (brk) break;
                continue;
            }
            float *col = dw + (size_t)stride * j + offset;
            for (int ri = 0; ri < rows; ri++) col[ri] = gb[ri] + col[ri];
        }
    }
}

/* sparse.rs:253-271 */
void orc_sparse_matmul_bwd_multi(int n, const float *const *g, const int32_t *const *idx, float *dw, int batch,
                                 const int *rows, int cols, int stride, const int *offset, int nnz, int brk) {
    size_t sz = (size_t)stride * cols;
    float *tmp = (float *)malloc(sz * sizeof(float));
    for (size_t i = 0; i < sz; i++) dw[i] = 0.0f;
    for (int k = 0; k < n; k++) {
        orc_sparse_matmul_bwd(g[k], idx[k], tmp, batch, rows[k], cols, stride, offset[k], nnz, brk);
        for (size_t i = 0; i < sz; i++) dw[i] = dw[i] + tmp[i];
    }
    free(tmp);
}

/* matmul.rs:62-93 */
void orc_matmul(const float *l, const float *r, float *o, int b, int m, int n, int k, int lcm, int rcm) {
    for (int bi = 0; bi < b; bi++)
        for (int ki = 0; ki < k; ki++)
            for (int mi = 0; mi < m; mi++) {
                float sum = 0.0f;
                for (int ni = 0; ni < n; ni++) {
                    size_t a = (size_t)m * n * bi + (lcm ? (size_t)m * ni + mi : (size_t)n * mi + ni);
                    size_t c = (size_t)n * k * bi + (rcm ? (size_t)n * ki + ni : (size_t)k * ni + ki);
                    float prod = l[a] * r[c];
                    sum = sum + prod;
                }
                o[(size_t)m * k * bi + (size_t)m * ki + mi] = sum;
            }
}

/* select.rs:12-26 */
void orc_select(const float *in, const int32_t *indices, float *out, int batch, int inner, int divisor) {
    int sub = inner / divisor;
    for (int o = 0; o < batch; o++)
        memcpy(out + (size_t)sub * o, in + (size_t)inner * o + (size_t)sub * indices[o], sizeof(float) * sub);
}

/* select.rs:93-110 */
void orc_select_pad(const float *in, const int32_t *indices, float *out, int batch, int inner, int divisor) {
    int sub = inner / divisor;
    for (int o = 0; o < batch; o++) {
        for (int i = 0; i < inner; i++) out[(size_t)inner * o + i] = 0.0f;
        memcpy(out + (size_t)inner * o + (size_t)sub * indices[o], in + (size_t)sub * o, sizeof(float) * sub);
    }
}

/* reduce.rs:96-120 */
void orc_reduce_sum_batch(const float *in, float *out, int batch, int inner) {
    for (int i = 0; i < inner; i++) out[i] = 0.0f;
    for (int b = 0; b < batch; b++)
        for (int i = 0; i < inner; i++) out[i] = out[i] + in[(size_t)inner * b + i];
}

static inline float maxf_(float a, float b) { return a > b ? a : b; }
static inline float minf_(float a, float b) { return a < b ? a : b; }
/* Unary::IsPositive, crates/compiler/src/tensor/operation/pointwise/unary.rs:37-79 */
static inline float ispos_(float x) { return x > 0.0f ? 1.0f : 0.0f; }

/* pointwise.rs:54-139 */
void orc_act_fwd(int act, const float *x, float *y, size_t n) {
    for (size_t i = 0; i < n; i++) {
        float v = x[i];
        switch (act) {
        case ORC_ACT_IDENTITY: y[i] = v; break;
        case ORC_ACT_RELU: y[i] = maxf_(v, 0.0f); break;
        case ORC_ACT_CRELU: y[i] = minf_(maxf_(v, 0.0f), 1.0f); break;
        case ORC_ACT_SCRELU: {
            float c = minf_(maxf_(v, 0.0f), 1.0f);
            y[i] = c * c;
        } break;
        case ORC_ACT_SQRRELU: {
            float c = maxf_(v, 0.0f);
            y[i] = c * c;
        } break;
        default: y[i] = v;
        }
    }
}

/* dfo.rs:30-40 (grad * f'(out)) with f' per dfo.rs:47-103 */
static inline float act_deriv_from_out(int act, float o) {
    switch (act) {
    case ORC_ACT_IDENTITY: return 1.0f;
    case ORC_ACT_RELU: return ispos_(o);
    case ORC_ACT_CRELU: return ispos_(o) * ispos_(1.0f - o);
    case ORC_ACT_SCRELU: {
        float half = (ispos_(o) * ispos_(1.0f - o)) * sqrtf(o);
        return half + half;
    }
    case ORC_ACT_SQRRELU: {
        float half = ispos_(o) * sqrtf(o);
        return half + half;
    }
    default: return 1.0f;
    }
}

void orc_act_bwd(int act, const float *out, const float *gout, float *gin, size_t n) {
    for (size_t i = 0; i < n; i++) gin[i] = gout[i] * act_deriv_from_out(act, out[i]);
}

/* pointwise.rs:156-164: Reciprocal(1 + Exp(-1 * x)) */
float orc_sigmoid(float x) { return 1.0f / (1.0f + expf(-1.0f * x)); }

/* adam.rs:35-52, call adamOp(adj * g, rate, ...) adam.rs:128 */
void orc_adamw(float *p, float *m, float *v, const float *g, size_t n, float adj, float lr, float decay, float beta1,
               float beta2, float wmin, float wmax) {
    for (size_t i = 0; i < n; i++) {
        float grad = adj * g[i];
        float pi = p[i];
        pi *= 1.0f - decay * lr;
        m[i] = beta1 * m[i] + (1.0f - beta1) * grad;
        v[i] = beta2 * v[i] + (1.0f - beta2) * grad * grad;
        float val = m[i] / (sqrtf(v[i]) + 0.00000001f);
        pi -= lr * val;
        p[i] = minf_(maxf_(pi, wmin), wmax);
    }
}

void orc_radam_step(int istep, float beta1, float beta2, float thr, float *step_size, int *denom) {
    const float step = (float)istep;
    const float beta2_t = powf(beta2, step);
    const float n_sma_max = 2.0f / (1.0f - beta2) - 1.0f;
    const float n_sma = n_sma_max - 2.0f * step * beta2_t / (1.0f - beta2_t);
    const float dn = 1.0f - powf(beta1, step);
    if (n_sma > thr) {
        const float p1 = (n_sma - 4.0f) / (n_sma_max - 4.0f);
        const float p2 = (n_sma - 2.0f) / n_sma;
        const float p3 = n_sma_max / (n_sma_max - 2.0f);
        *step_size = sqrtf((1.0f - beta2_t) * p1 * p2 * p3) / dn;
    } else {
        *step_size = 1.0f / dn;
    }
    *denom = n_sma > thr;
}

void orc_radam(float *p, float *m, float *v, const float *g, size_t n, float adj, float lr, float step_size, int denom,
               float decay, float beta1, float beta2, float wmin, float wmax) {
    const float rate = lr * step_size; /* rate_ptr[0] * step_size_ptr[0] */
    for (size_t i = 0; i < n; i++) {
        const float grad = adj * g[i];
        float pi = p[i];
        pi *= 1.0f - decay * rate;
        m[i] = beta1 * m[i] + (1.0f - beta1) * grad;
        v[i] = beta2 * v[i] + (1.0f - beta2) * grad * grad;
        float val = m[i];
        if (denom) val /= sqrtf(v[i]) + 0.00000001f;
        pi -= rate * val;
        p[i] = minf_(maxf_(pi, wmin), wmax);
    }
}

float orc_loss(int kind, int batch, int width, const float *y, const float *targets, const float *ew, float power,
               float *grad_y) {
    float loss = 0.0f;
    for (int b = 0; b < batch; b++) {
        const float g = ew ? ew[b] : 1.0f;
        if (kind == ORC_LOSS_SOFTMAX_CE) {
            const float *yb = y + (size_t)b * width, *tb = targets + (size_t)b * width;
            float mx = yb[0];
            for (int k = 1; k < width; k++) mx = yb[k] > mx ? yb[k] : mx; /* reduce_max */
            float den = 0.0f;
            for (int k = 0; k < width; k++) den = den + expf(yb[k] + (-1.0f * mx)); /* reduce_sum of exp(x - max) */
            const float inv = 1.0f / den;                                           /* exps / denom = exps * (1/denom) */
            float l = 0.0f;
            for (int k = 0; k < width; k++) {
                const float p = expf(yb[k] + (-1.0f * mx)) * inv;
                float lk = -1.0f * (tb[k] * logf(p));
                if (ew) lk = g * lk;
                l = l + lk;
                if (grad_y) grad_y[(size_t)b * width + k] = (p + (-1.0f * tb[k])) * g;
            }
            loss = loss + l;
            continue;
        }
        const float s = orc_sigmoid(y[b]);
        const float diff = s + (-1.0f * targets[b]);
        float l, dd;
        if (kind == ORC_LOSS_SIGMOID_POWER) {
            const float a = fabsf(diff);
            l = powf(a, power);
            const float sg = diff == 0.0f ? 0.0f : (diff > 0.0f ? 1.0f : -1.0f);
            dd = (g * (power * powf(a, power + -1.0f))) * sg;
        } else {
            l = diff * diff;
            dd = g * diff + g * diff;
        }
        if (ew) l = g * l;
        loss = loss + l;
        if (grad_y) grad_y[b] = dd * (s * (1.0f + (-1.0f * s)));
    }
    return loss;
}

void orc_weight_clip(float *w, size_t n, float min_w, float max_w) {
    for (size_t i = 0; i < n; i++) {
        float high = w[i] < max_w ? w[i] : max_w;
        w[i] = high > min_w ? high : min_w;
    }
}

/* ModelNode::clip_pass_through_grad: x.max(min).min(max) (compiler/src/model/operations/pointwise.rs:289-298);
 * backward = the incoming gradient (tensor/operation/autograd/passthrough.rs:31-33) */
void orc_clip_pass_through(const float *x, float *y, size_t n, float min_v, float max_v) {
    for (size_t i = 0; i < n; i++) {
        float low = x[i] > min_v ? x[i] : min_v;
        y[i] = low < max_v ? low : max_v;
    }
}

/* ModelNode::faux_quantise: scalar * input -> Round | Truncate -> * Reciprocal(scalar)
 * (compiler/src/model/operations/pointwise.rs:212-220; f32::round = half away from zero, unary.rs:51-52) */
void orc_faux_quantise(const float *x, float *y, size_t n, float value, int round) {
    const float denom = 1.0f / value;
    for (size_t i = 0; i < n; i++) {
        const float scaled = value * x[i];
        y[i] = (round ? roundf(scaled) : truncf(scaled)) * denom;
    }
}

void orc_weight_decay(float *w, size_t n, float decay, float lr) {
    for (size_t i = 0; i < n; i++) w[i] = w[i] * (1.0f + (-1.0f * (lr * decay)));
}

void orc_lookahead(float *w, float *slow, size_t n, float alpha) {
    for (size_t i = 0; i < n; i++) {
        const float nw = alpha * w[i] + (1.0f - alpha) * slow[i];
        w[i] = nw;
        slow[i] = nw;
    }
}

/* ====================================================================== net */

int orc_net_num_tensors(const orc_net_desc *d) { return 2 + 2 * d->n_dense + (d->ft_factoriser > 0 ? 1 : 0); }

/* width of the activated output of dense layer k (what the next layer sees) */
static int dense_width(const orc_net_desc *d, int k) {
    return (d->dense_out[k] - d->dense_skip[k]) * (d->dense_dual[k] ? 2 : 1);
}

int orc_net_dense_in(const orc_net_desc *d, int k) {
    if (k == 0) return d->perspectives * (d->pairwise ? d->ft_out / 2 : d->ft_out);
    return dense_width(d, k - 1);
}

static int dense_rows_full(const orc_net_desc *d, int k) {
    return d->dense_out[k] * (d->dense_bucketed[k] ? d->num_buckets : 1);
}

size_t orc_net_tensor_size(const orc_net_desc *d, int t) {
    if (d->ft_factoriser > 0 && t == 2 + 2 * d->n_dense) return (size_t)d->ft_out * (d->num_inputs / d->ft_factoriser);
    if (t == 0) return (size_t)d->ft_out * d->num_inputs;
    if (t == 1) return (size_t)d->ft_out;
    int k = (t - 2) / 2;
    if ((t & 1) == 0) return (size_t)dense_rows_full(d, k) * orc_net_dense_in(d, k);
    return (size_t)dense_rows_full(d, k);
}

typedef struct {
    /* per perspective */
    float *c[2];   /* activated FT rows [B][H] (pairwise: CReLU'd halves; else act output) */
    float *hid;    /* concat hidden [B][K0] */
    float *a[ORC_MAX_DENSE];  /* activated output of dense k: [B][width_k] */
    float *z[ORC_MAX_DENSE];  /* pre-activation output of dense k: [B][out_k] (kept for dual / skip layers) */
    float *y;      /* net output [B][out_last] = a[last] + skip neurons */
    float *weff;   /* l0w + repeat(l0f) when the net has a factoriser */
    float *sig;    /* sigmoid(y) [B] */
} orc_acts;

/* W_eff = l0w + l0f.repeat(NB): column c of the bucketed table gets factoriser column c mod (F/NB) */
static float *factorised_weights(const orc_net_desc *d, const float *const *params) {
    const int H = d->ft_out, F = d->num_inputs, Fc = F / d->ft_factoriser;
    const float *w0 = params[0], *wf = params[2 + 2 * d->n_dense];
    float *we = (float *)malloc(sizeof(float) * (size_t)H * F);
    for (int c = 0; c < F; c++)
        for (int r = 0; r < H; r++) we[(size_t)c * H + r] = w0[(size_t)c * H + r] + wf[(size_t)(c % Fc) * H + r];
    return we;
}

static float net_forward_impl(const orc_net_desc *d, int B, const float *const *params, const int32_t *stm,
                              const int32_t *nstm, const int32_t *buckets, const float *targets,
                              const float *entry_weights, float *net_out, orc_acts *A) {
    const int H = d->ft_out, P = d->perspectives, F = d->num_inputs, N = d->nnz;
    const int Hout = d->pairwise ? H / 2 : H;
    const int K0 = P * Hout;
    const float *w0 = params[0], *b0 = params[1];
    const int32_t *idxs[2] = {stm, nstm};
    if (d->ft_factoriser > 0) w0 = A->weff = factorised_weights(d, params);

    A->hid = (float *)malloc(sizeof(float) * (size_t)B * K0);
    for (int p = 0; p < P; p++) {
        float *acc = (float *)malloc(sizeof(float) * (size_t)B * H);
        /* a7: SparseMatmul with the device break rule; pairwise nets issue two sliced ops
         * (rows H/2, offsets 0 and H/2: foldrules.rs:255-262) which is element-for-element this. */
        orc_sparse_matmul_fwd(w0, idxs[p], acc, B, H, F, H, 0, N, 1);
        /* a8: Affine::forward adds the bias after the sum, builder.rs:145-147 */
        for (int b = 0; b < B; b++)
            for (int r = 0; r < H; r++) acc[(size_t)b * H + r] = acc[(size_t)b * H + r] + b0[r];
        orc_act_fwd(d->ft_act, acc, acc, (size_t)B * H);
        A->c[p] = acc;
        for (int b = 0; b < B; b++) {
            float *h = A->hid + (size_t)b * K0 + (size_t)p * Hout; /* concat: stm first, index.rs:74-81 */
            const float *cb = acc + (size_t)b * H;
            if (d->pairwise)
                for (int r = 0; r < Hout; r++) h[r] = cb[r] * cb[r + Hout];
            else
                for (int r = 0; r < Hout; r++) h[r] = cb[r];
        }
    }

    const float *x = A->hid;
    int in = K0;
    for (int k = 0; k < d->n_dense; k++) {
        const int out = d->dense_out[k];
        const int rows = dense_rows_full(d, k);
        const float *w = params[2 + 2 * k], *bias = params[3 + 2 * k];
        float *a = (float *)malloc(sizeof(float) * (size_t)B * out);
        for (int b = 0; b < B; b++) {
            /* a10: select keeps rows [bucket*out, (bucket+1)*out); the reference computes every
             * bucket row and discards the rest -- the kept rows are the same dot products. */
            int base = d->dense_bucketed[k] ? buckets[b] * out : 0;
            const float *xb = x + (size_t)b * in;
            for (int o = 0; o < out; o++) {
                int row = base + o;
                float z;
                if (k == 0 && P == 2) {
                    /* ConcatMatmulToAddSlicedMatmuls, rewriterules.rs:211-265: W[:, :Hout] a + W[:, Hout:] b */
                    float s0 = 0.0f, s1 = 0.0f;
                    for (int i = 0; i < Hout; i++) s0 = s0 + w[(size_t)rows * i + row] * xb[i];
                    for (int i = Hout; i < in; i++) s1 = s1 + w[(size_t)rows * i + row] * xb[i];
                    z = s0 + s1;
                } else {
                    z = 0.0f;
                    for (int i = 0; i < in; i++) z = z + w[(size_t)rows * i + row] * xb[i];
                }
                z = z + bias[row];
                a[(size_t)b * out + o] = z;
            }
        }
        if (d->dense_dual[k] || d->dense_skip[k]) {
            /* skip neurons are sliced off first (slice_rows, test2.rs:54-56); the rest may be widened to
             * concat(x, x*x) (test2.rs:58) and is then activated */
            const int m = out - d->dense_skip[k], wd = dense_width(d, k);
            float *act = (float *)malloc(sizeof(float) * (size_t)B * wd);
            for (int b = 0; b < B; b++)
                for (int i = 0; i < m; i++) {
                    float v = a[(size_t)b * out + i];
                    act[(size_t)b * wd + i] = v;
                    if (d->dense_dual[k]) act[(size_t)b * wd + m + i] = v * v;
                }
            orc_act_fwd(d->dense_act[k], act, act, (size_t)B * wd);
            A->z[k] = a;
            A->a[k] = act;
            x = act;
            in = wd;
            continue;
        }
        orc_act_fwd(d->dense_act[k], a, a, (size_t)B * out);
        A->a[k] = a;
        x = a;
        in = out;
    }

    /* net output: last layer + every skip neuron (`out + skip_neuron`, test2.rs:64) */
    const int L = d->n_dense, out_last = dense_width(d, L - 1);
    A->y = (float *)malloc(sizeof(float) * (size_t)B * out_last);
    for (size_t i = 0; i < (size_t)B * out_last; i++) A->y[i] = A->a[L - 1][i];
    for (int k = 0; k < L - 1; k++)
        for (int b = 0; b < B && d->dense_skip[k]; b++)
            for (int j = 0; j < d->dense_skip[k]; j++)
                A->y[(size_t)b * out_last + j] =
                    A->y[(size_t)b * out_last + j] + A->z[k][(size_t)b * d->dense_out[k] + d->dense_out[k] - d->dense_skip[k] + j];

    /* a11: loss = sum_b w_b (sigmoid(y_b) - t_b)^2, value/builder.rs:137-142 */
    const float *y = A->y;
    if (d->loss_kind != ORC_LOSS_SIGMOID_MSE) {
        if (net_out) memcpy(net_out, y, sizeof(float) * (size_t)B * out_last);
        return orc_loss(d->loss_kind, B, out_last, y, targets, d->use_entry_weights ? entry_weights : NULL,
                        d->loss_power, NULL);
    }
    A->sig = (float *)malloc(sizeof(float) * (size_t)B);
    float loss = 0.0f;
    for (int b = 0; b < B; b++) {
        float s = orc_sigmoid(y[b]);
        A->sig[b] = s;
        float diff = s + (-1.0f * targets[b]);
        float l = diff * diff;
        if (d->use_entry_weights) l = entry_weights[b] * l;
        loss = loss + l;
        if (net_out) net_out[b] = y[b];
    }
    return loss;
}

static void acts_free(const orc_net_desc *d, orc_acts *A) {
    for (int p = 0; p < d->perspectives; p++) free(A->c[p]);
    free(A->hid);
    for (int k = 0; k < d->n_dense; k++) free(A->a[k]), free(A->z[k]);
    free(A->y);
    free(A->weff);
    free(A->sig);
}

float orc_net_forward(const orc_net_desc *d, int batch, const float *const *params, const int32_t *stm,
                      const int32_t *nstm, const int32_t *buckets, const float *targets, const float *entry_weights,
                      float *net_out) {
    orc_acts A;
    memset(&A, 0, sizeof(A));
    float l = net_forward_impl(d, batch, params, stm, nstm, buckets, targets, entry_weights, net_out, &A);
    acts_free(d, &A);
    return l;
}

float orc_net_forward_backward(const orc_net_desc *d, int B, const float *const *params, float *const *grads,
                               const int32_t *stm, const int32_t *nstm, const int32_t *buckets, const float *targets,
                               const float *entry_weights, float *net_out) {
    const int H = d->ft_out, P = d->perspectives, F = d->num_inputs, N = d->nnz;
    const int Hout = d->pairwise ? H / 2 : H;
    const int K0 = P * Hout;
    const int32_t *idxs[2] = {stm, nstm};
    orc_acts A;
    memset(&A, 0, sizeof(A));
    float loss = net_forward_impl(d, B, params, stm, nstm, buckets, targets, entry_weights, net_out, &A);

    /* seed dL = 1 (definition.rs:101); d(sum) = broadcast; squared_error diff*diff -> g*diff + g*diff
     * (binary.rs:101-123); sigmoid' = o * (1 - o) (dfo.rs:114-118) */
    int out_last = dense_width(d, d->n_dense - 1);
    float *gz = (float *)malloc(sizeof(float) * (size_t)B * out_last); /* grad wrt activated output of last layer */
    float *gy = (float *)malloc(sizeof(float) * (size_t)B * out_last); /* grad wrt the net output (skip neurons get it too) */
    if (d->loss_kind != ORC_LOSS_SIGMOID_MSE)
        (void)orc_loss(d->loss_kind, B, out_last, A.y, targets, d->use_entry_weights ? entry_weights : NULL,
                       d->loss_power, gz);
    for (int b = 0; b < B && d->loss_kind == ORC_LOSS_SIGMOID_MSE; b++) {
        float s = A.sig[b];
        float g = d->use_entry_weights ? entry_weights[b] : 1.0f;
        float diff = s + (-1.0f * targets[b]);
        float dd = g * diff + g * diff;
        gz[b] = dd * (s * (1.0f + (-1.0f * s)));
    }
    memcpy(gy, gz, sizeof(float) * (size_t)B * out_last);

    for (int k = d->n_dense - 1; k >= 0; k--) {
        const int out = d->dense_out[k];
        const int rows = dense_rows_full(d, k);
        const int in = orc_net_dense_in(d, k);
        const float *w = params[2 + 2 * k];
        float *dw = grads[2 + 2 * k], *db = grads[3 + 2 * k];
        const float *x = (k == 0) ? A.hid : A.a[k - 1];
        if (d->dense_dual[k] || d->dense_skip[k]) {
            /* activation backward on the widened vector, then Concat/Slice backward (pad.rs, slice.rs) and Mul backward
             * with both operands = x: contributions arrive in reverse creation order: concat slice, then g*x twice */
            const int m = out - d->dense_skip[k], wd = dense_width(d, k);
            orc_act_bwd(d->dense_act[k], A.a[k], gz, gz, (size_t)B * wd);
            float *gzz = (float *)malloc(sizeof(float) * (size_t)B * out);
            for (int b = 0; b < B; b++) {
                for (int i = 0; i < m; i++) {
                    float g1 = gz[(size_t)b * wd + i];
                    if (d->dense_dual[k]) {
                        float g2 = gz[(size_t)b * wd + m + i], xv = A.z[k][(size_t)b * out + i];
                        g1 = (g1 + g2 * xv) + g2 * xv;
                    }
                    gzz[(size_t)b * out + i] = g1;
                }
                for (int j = 0; j < d->dense_skip[k]; j++) gzz[(size_t)b * out + m + j] = gy[(size_t)b * out_last + j];
            }
            free(gz);
            gz = gzz;
        } else {
            /* activation backward from output */
            orc_act_bwd(d->dense_act[k], A.a[k], gz, gz, (size_t)B * out);
        }
        /* SelectPad puts zeros in all other bucket rows (select.rs:93-110); adding exact zeros
         * to dW/db/dX changes nothing, so only the selected rows are visited. Batch order is
         * ascending, as in Matmul::evaluate's ni loop over the batch-contracted operand. */
        for (size_t i = 0; i < (size_t)rows * in; i++) dw[i] = 0.0f;
        for (int i = 0; i < rows; i++) db[i] = 0.0f;
        float *gx = (float *)malloc(sizeof(float) * (size_t)B * in);
        for (int b = 0; b < B; b++) {
            int base = d->dense_bucketed[k] ? buckets[b] * out : 0;
            const float *xb = x + (size_t)b * in;
            const float *gb = gz + (size_t)b * out;
            for (int o = 0; o < out; o++) db[base + o] = db[base + o] + gb[o];
            for (int i = 0; i < in; i++) {
                float s = 0.0f;
                for (int o = 0; o < out; o++) {
                    size_t wi = (size_t)rows * i + base + o;
                    dw[wi] = dw[wi] + gb[o] * xb[i]; /* dW = dY X^T, matmul.rs:109-115 */
                    s = s + w[wi] * gb[o];           /* dX = W^T dY, matmul.rs:117-123 */
                }
                gx[(size_t)b * in + i] = s;
            }
        }
        free(gz);
        gz = gx;
    }
    free(gy);

    /* gz is now d(hidden) [B][K0]; FT backward */
    float *gp[2] = {NULL, NULL};
    float *db0 = grads[1];
    for (int r = 0; r < H; r++) db0[r] = 0.0f;
    for (int p = 0; p < P; p++) {
        float *g = (float *)malloc(sizeof(float) * (size_t)B * H);
        for (int b = 0; b < B; b++) {
            const float *gh = gz + (size_t)b * K0 + (size_t)p * Hout;
            const float *cb = A.c[p] + (size_t)b * H;
            float *gb = g + (size_t)b * H;
            if (d->pairwise) {
                /* Mul backward: other operand (binary.rs:101-123), then CReLU' from each half's output */
                for (int r = 0; r < Hout; r++) {
                    gb[r] = (gh[r] * cb[r + Hout]) * act_deriv_from_out(d->ft_act, cb[r]);
                    gb[r + Hout] = (gh[r] * cb[r]) * act_deriv_from_out(d->ft_act, cb[r + Hout]);
                }
            } else {
                for (int r = 0; r < H; r++) gb[r] = gh[r] * act_deriv_from_out(d->ft_act, cb[r]);
            }
        }
        /* a14: bias grad = batch sum per perspective, accumulated over the uses of l0b */
        float *tmp = (float *)malloc(sizeof(float) * (size_t)H);
        orc_reduce_sum_batch(g, tmp, B, H);
        for (int r = 0; r < H; r++) db0[r] = db0[r] + tmp[r];
        free(tmp);
        gp[p] = g;
    }
    {
        int rows[2] = {H, H}, offs[2] = {0, 0};
        const float *gc[2] = {gp[0], gp[1]};
        orc_sparse_matmul_bwd_multi(P, gc, idxs, grads[0], B, rows, F, H, offs, N, 1);
    }
    if (d->ft_factoriser > 0) {
        /* l0w + repeat(l0f): Add passes the gradient to both; Broadcast(Cols) backward is the sum over the
         * repeats (broadcast.rs:121-124, reduce.rs:96-120), taken here in ascending repeat order */
        const int Fc = F / d->ft_factoriser;
        float *dwf = grads[2 + 2 * d->n_dense];
        for (int j = 0; j < Fc; j++)
            for (int r = 0; r < H; r++) {
                float sacc = 0.0f;
                for (int rep = 0; rep < d->ft_factoriser; rep++) sacc = sacc + grads[0][((size_t)rep * Fc + j) * H + r];
                dwf[(size_t)j * H + r] = sacc;
            }
    }
    for (int p = 0; p < P; p++) free(gp[p]);
    free(gz);
    acts_free(d, &A);
    return loss;
}

/* ====================================================================== host input path */

void orc_input_mirrored_expand(const int *layout32, orc_input_desc *d) {
    static const int mir[8] = {0, 1, 2, 3, 3, 2, 1, 0};
    int mx = 0;
    for (int i = 0; i < 32; i++) mx = layout32[i] > mx ? layout32[i] : mx;
    for (int idx = 0; idx < 64; idx++) d->buckets[idx] = layout32[(idx / 8) * 4 + mir[idx % 8]];
    d->num_buckets = mx + 1; /* get_num_buckets: max + 1, crates/bullet_lib/src/game/inputs.rs */
    d->kind = ORC_INPUT_CHESS_BUCKETS_MIRRORED;
}

void orc_input_buckets_set(const int *layout64, orc_input_desc *d) {
    int mx = 0;
    for (int i = 0; i < 64; i++) {
        d->buckets[i] = layout64[i];
        mx = layout64[i] > mx ? layout64[i] : mx;
    }
    d->num_buckets = mx + 1;
    d->kind = ORC_INPUT_CHESS_BUCKETS;
}

static int base_num_inputs(const orc_input_desc *d) {
    switch (d->kind) {
    case ORC_INPUT_CHESS768: return 768;
    case ORC_INPUT_CHESS_BUCKETS:
    case ORC_INPUT_CHESS_BUCKETS_MIRRORED: return 768 * d->num_buckets;
    case ORC_INPUT_ATAXX147: return 147;
    case ORC_INPUT_ATAXX98: return 98;
    }
    return 0;
}

int orc_input_num_inputs(const orc_input_desc *d) {
    /* factorised.rs:24-26 */
    return base_num_inputs(d) + (d->factorised ? 768 : 0);
}

int orc_input_max_active(const orc_input_desc *d) {
    int m = (d->kind == ORC_INPUT_ATAXX147 || d->kind == ORC_INPUT_ATAXX98) ? 49 : 32;
    return d->factorised ? 2 * m : m; /* factorised.rs:28-30 */
}

int orc_map_features(const orc_input_desc *d, const void *record, int32_t *stm, int32_t *ntm) {
    int cnt = 0;
    if (d->kind == ORC_INPUT_ATAXX147 || d->kind == ORC_INPUT_ATAXX98) {
        orc_ataxx_board bd;
        memcpy(&bd, record, sizeof(bd));
        /* AtaxxBoard iterator: boards 0,1,2 in bit-scan order (examples/ataxx.rs:20-31) */
        for (int pc = 0; pc < 3; pc++) {
            uint64_t bb = bd.bbs[pc];
            while (bb) {
                int sq = __builtin_ctzll(bb);
                bb &= bb - 1;
                int s = 49 * pc + sq;
                int n = pc == 2 ? s : 49 * (1 - pc) + sq; /* ataxx147.rs:23-24 */
                if (d->kind == ORC_INPUT_ATAXX98 && s >= 98) continue; /* ataxx147.rs:52-56 */
                stm[cnt] = s;
                ntm[cnt] = n;
                cnt++;
            }
        }
        return cnt;
    }
    orc_chess_board bd;
    memcpy(&bd, record, sizeof(bd));
    int sflip = 0, nflip = 0, sb = 0, nb = 0;
    if (d->kind == ORC_INPUT_CHESS_BUCKETS) {
        sb = 768 * d->buckets[bd.ksq]; /* chess_buckets.rs:29-30 */
        nb = 768 * d->buckets[bd.opp_ksq];
    } else if (d->kind == ORC_INPUT_CHESS_BUCKETS_MIRRORED) {
        sflip = (bd.ksq % 8 > 3) ? 7 : 0; /* chess_buckets.rs:83 */
        nflip = (bd.opp_ksq % 8 > 3) ? 7 : 0;
        sb = 768 * d->buckets[bd.ksq];
        nb = 768 * d->buckets[bd.opp_ksq];
    }
    const int off = d->factorised ? 768 : 0; /* factorised.rs:15-17: offset = factoriser.num_inputs() */
    uint64_t occ = bd.occ;
    int i = 0;
    while (occ) { /* examples/tests/test4.rs:23-36 */
        int sq = __builtin_ctzll(occ);
        int piece = (bd.pcs[i / 2] >> (4 * (i % 2))) & 0xF;
        occ &= occ - 1;
        i++;
        int c = (piece & 8) > 0;
        int pc = 64 * (piece & 7);
        int s = (c ? 384 : 0) + pc + sq;          /* chess768.rs:26 */
        int n = (c ? 0 : 384) + pc + (sq ^ 56);   /* chess768.rs:27 */
        int s2 = sb + (s ^ sflip), n2 = nb + (n ^ nflip);
        stm[cnt] = off + s2;
        ntm[cnt] = off + n2;
        cnt++;
        if (d->factorised) {
            /* factorised.rs:36-43 with Chess768 as factoriser: derive_feature = feat % 768 */
            stm[cnt] = s2 % 768;
            ntm[cnt] = n2 % 768;
            cnt++;
        }
    }
    return cnt;
}

int orc_material_bucket(uint64_t occ, int n) {
    int divisor = (32 + n - 1) / n; /* outputs.rs:28 */
    return (uint8_t)((uint8_t)__builtin_popcountll(occ) - 2) / (uint8_t)divisor;
}

static float sigmoid_host(float x) { return 1.0f / (1.0f + expf(-x)); } /* value.rs:66-68 */

void orc_map_batch(const orc_input_desc *d, const void *records, int count, int n_out_buckets, float blend,
                   float eval_scale, int use_win_rate_model, int wdl_output, int32_t *stm, int32_t *ntm,
                   int32_t *buckets, float *targets, float *weights) {
    const int nnz = orc_input_max_active(d);
    const int is_ataxx = (d->kind == ORC_INPUT_ATAXX147 || d->kind == ORC_INPUT_ATAXX98);
    const float rscale = 1.0f / eval_scale; /* value.rs:64 */
    const uint8_t *rec = (const uint8_t *)records;
    for (int k = 0; k < count; k++, rec += 32) {
        int32_t *s = stm + (size_t)k * nnz, *n = ntm + (size_t)k * nnz;
        int cnt = orc_map_features(d, rec, s, n);
        if (cnt < nnz) { /* value.rs:86-89 */
            s[cnt] = -1;
            n[cnt] = -1;
        }
        int16_t score_i;
        uint8_t result;
        uint64_t occ = 0;
        if (is_ataxx) {
            orc_ataxx_board bd;
            memcpy(&bd, rec, 32);
            score_i = bd.score;
            result = bd.result;
        } else {
            orc_chess_board bd;
            memcpy(&bd, rec, 32);
            score_i = bd.score;
            result = bd.result;
            occ = bd.occ;
        }
        buckets[k] = (n_out_buckets > 1 && !is_ataxx) ? orc_material_bucket(occ, n_out_buckets) : 0;
        weights[k] = 1.0f; /* value.rs:94, no weight getter */
        if (wdl_output) {  /* value.rs:96-101 */
            targets[3 * k] = targets[3 * k + 1] = targets[3 * k + 2] = 0.0f;
            targets[3 * k + result] = 1.0f;
        } else { /* value.rs:103-115 */
            float score = (float)score_i;
            if (use_win_rate_model) {
                float p = (score - 270.0f) / 380.0f;
                float pm = (-score - 270.0f) / 380.0f;
                score = 0.5f * (1.0f + sigmoid_host(p) - sigmoid_host(pm));
            } else {
                score = sigmoid_host(rscale * score);
            }
            float res = (float)result / 2.0f;
            targets[k] = blend * res + (1.0f - blend) * score;
        }
    }
}

/* ====================================================================== weight-init RNG */

static uint64_t splitmix64(uint64_t *x) {
    *x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = *x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* rand_xoshiro: Xoroshiro128Plus::seed_from_u64 fills the 16-byte seed from SplitMix64 */
void orc_rng_seed(orc_rng *r, uint64_t seed) {
    uint64_t x = seed;
    r->s0 = splitmix64(&x);
    r->s1 = splitmix64(&x);
}

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

/* xoroshiro128+ (Blackman & Vigna 2018 constants 24, 16, 37) */
uint64_t orc_rng_next_u64(orc_rng *r) {
    uint64_t s0 = r->s0, s1 = r->s1;
    uint64_t res = s0 + s1;
    s1 ^= s0;
    r->s0 = rotl64(s0, 24) ^ s1 ^ (s1 << 16);
    r->s1 = rotl64(s1, 37);
    return res;
}

/* Ziggurat tables for N(0,1), 256 layers, as generated by rand_distr's
 * ziggurat_tables.py: x[0] = v/f(r), x[1] = r, x[i] = f_inv(v/x[i-1] + f(x[i-1])), x[256] = 0 */
#define ZIG_NORM_R 3.654152885361008796
static double zig_x[257], zig_f[257];
static int zig_ready = 0;
static void zig_init(void) {
    if (zig_ready) return;
    const double r = ZIG_NORM_R, v = 0.00492867323399;
    zig_x[0] = v / exp(-r * r / 2.0);
    zig_x[1] = r;
    for (int i = 2; i < 256; i++) {
        double last = zig_x[i - 1];
        zig_x[i] = sqrt(-2.0 * log(v / last + exp(-last * last / 2.0)));
    }
    zig_x[256] = 0.0;
    for (int i = 0; i < 257; i++) zig_f[i] = exp(-zig_x[i] * zig_x[i] / 2.0);
    zig_ready = 1;
}

static inline double u64_bits_to_f64(uint64_t b) {
    double d;
    memcpy(&d, &b, 8);
    return d;
}
/* into_float_with_exponent(e): mantissa = top 52 bits already shifted, exponent e */
static inline double float_with_exp(uint64_t mant52, int e) {
    return u64_bits_to_f64(((uint64_t)(1023 + e) << 52) | mant52);
}
/* Open01: (next_u64 >> 12) as mantissa with exponent 0, minus (1 - EPSILON/2) */
static inline double open01(orc_rng *r) {
    return float_with_exp(orc_rng_next_u64(r) >> 12, 0) - (1.0 - 2.220446049250313e-16 / 2.0);
}
/* StandardUniform f64: (next_u64 >> 11) * 2^-53 */
static inline double uniform01(orc_rng *r) { return (double)(orc_rng_next_u64(r) >> 11) * (1.0 / 9007199254740992.0); }

double orc_rng_standard_normal(orc_rng *r) {
    zig_init();
    for (;;) {
        uint64_t bits = orc_rng_next_u64(r);
        int i = (int)(bits & 0xff);
        double u = float_with_exp(bits >> 12, 1) - (3.0 - 2.220446049250313e-16 / 2.0);
        double x = u * zig_x[i];
        double test_x = fabs(x);
        if (test_x < zig_x[i + 1]) return x;
        if (i == 0) {
            double xx = 1.0, yy = 0.0;
            while (-2.0 * yy < xx * xx) {
                double x_ = open01(r);
                double y_ = open01(r);
                xx = log(x_) / ZIG_NORM_R;
                yy = log(y_);
            }
            return u < 0.0 ? xx - ZIG_NORM_R : ZIG_NORM_R - xx;
        }
        if (zig_f[i + 1] + (zig_f[i] - zig_f[i + 1]) * uniform01(r) < exp(-x * x / 2.0)) return x;
    }
}

void orc_rng_fill(orc_rng *r, float *out, size_t n, float mean, float stdev, int gaussian) {
    for (size_t i = 0; i < n; i++) {
        if (gaussian) {
            float z = (float)orc_rng_standard_normal(r); /* StandardNormal f32 = f64 sample cast */
            out[i] = mean + stdev * z;                   /* Normal::from_zscore */
        } else {
            /* Uniform::new(mean - stdev, mean + stdev) for f32: low + scale * u01(23 bits) */
            float low = mean - stdev, high = mean + stdev;
            float scale = high - low;
            uint32_t bits = (uint32_t)(orc_rng_next_u64(r) >> 32); /* next_u32 = high half of next_u64 */
            float u;
            uint32_t fb = 0x3F800000u | (bits >> 9);
            memcpy(&u, &fb, 4);
            u -= 1.0f;
            out[i] = u * scale + low;
        }
    }
}

void orc_net_init(const orc_net_desc *d, uint64_t seed, int ft_eff, float *const *params) {
    orc_rng r;
    orc_rng_seed(&r, seed);
    int nt = orc_net_num_tensors(d);
    for (int t = 0; t < nt; t++) {
        size_t n = orc_net_tensor_size(d, t);
        if ((t & 1) || (d->ft_factoriser > 0 && t == nt - 1)) { /* biases and l0f: InitSettings::Zeroed consumes no RNG, weights.rs:64 */
            memset(params[t], 0, n * sizeof(float));
            continue;
        }
        int in = (t == 0) ? (ft_eff > 0 ? ft_eff : d->num_inputs) : orc_net_dense_in(d, (t - 2) / 2);
        float stdev = sqrtf(2.0f / ((float)in * 1.0f)); /* builder.rs:115 with bias_cols = 1 */
        orc_rng_fill(&r, params[t], n, 0.0f, stdev, 1);
    }
}

/* ------------------------------------------------------------------ generic index / reduce ops
 * The ops the lowered graph is made of besides the ones above: concat = pad + pad + add
 * (model/operations/index.rs:74-81), the pairwise halves and the skip neuron = slices, bias = broadcast, bias
 * gradients and the loss = reduce-sum, softmax = reduce-max. Shapes are given as (outer, dimen, inner) of the
 * dimension acted on: outer = product of the dimensions before it, inner = product of those after it
 * (broadcast: inner includes the dimension itself). Elements are 4 bytes (f32 or i32) for the copying ops. */

/* BroadcastAcrossDimension::apply, index/broadcast.rs:66-80 */
void orc_broadcast(const uint32_t *in, size_t outer, size_t inner, size_t repeats, uint32_t *out) {
    for (size_t i = 0; i < outer; i++)
        for (size_t j = 0; j < inner; j++) {
            const size_t ld = inner * i;
            for (size_t r = 0; r < repeats; r++) out[ld * repeats + inner * r + j] = in[ld + j];
        }
}

/* PadAcrossDimension::apply, index/pad.rs:76-102 */
void orc_pad(const uint32_t *in, size_t outer, size_t dimen, size_t inner, size_t before, size_t after, uint32_t value,
             uint32_t *out) {
    const size_t middle = before + dimen + after;
    for (size_t o = 0; o < outer; o++)
        for (size_t i = 0; i < inner; i++) {
            const size_t base = middle * inner * o;
            for (size_t b = 0; b < before; b++) out[base + inner * b + i] = value;
            for (size_t d = 0; d < dimen; d++) out[base + inner * (before + d) + i] = in[inner * (dimen * o + d) + i];
            for (size_t a = 0; a < after; a++) out[base + inner * (before + dimen + a) + i] = value;
        }
}

/* SliceAcrossDimension::apply, index/slice.rs:74-91 */
void orc_slice(const uint32_t *in, size_t outer, size_t dimen, size_t inner, size_t start, size_t end, uint32_t *out) {
    const size_t od = end - start;
    for (size_t o = 0; o < outer; o++)
        for (size_t d = 0; d < od; d++)
            memcpy(out + inner * od * o + inner * d, in + inner * dimen * o + inner * (d + start), inner * 4);
}

/* ReduceAcrossDimension::apply, linear/reduce.rs:96-120; kind 0 = Sum, 1 = Max, 2 = Min; acc starts at element 0 */
void orc_reduce_f32(const float *in, size_t outer, size_t dimen, size_t inner, int kind, float *out) {
    for (size_t i = 0; i < outer; i++)
        for (size_t j = 0; j < inner; j++) {
            float acc = in[i * dimen * inner + j];
            for (size_t k = 1; k < dimen; k++) {
                const float x = in[i * dimen * inner + k * inner + j];
                acc = kind == 0 ? acc + x : kind == 1 ? (x > acc ? x : acc) : (x < acc ? x : acc);
            }
            out[i * inner + j] = acc;
        }
}
void orc_reduce_i32(const int32_t *in, size_t outer, size_t dimen, size_t inner, int kind, int32_t *out) {
    for (size_t i = 0; i < outer; i++)
        for (size_t j = 0; j < inner; j++) {
            int32_t acc = in[i * dimen * inner + j];
            for (size_t k = 1; k < dimen; k++) {
                const int32_t x = in[i * dimen * inner + k * inner + j];
                acc = kind == 0 ? acc + x : kind == 1 ? (x > acc ? x : acc) : (x < acc ? x : acc);
            }
            out[i * inner + j] = acc;
        }
}

/* ---- batch-parallel training steps for the CPU baseline of bench.py (the restatement run over all host
 * cores with std::thread-style workers inside C; not used by any parity test). Each worker runs the scalar
 * forward + backward on a contiguous shard of the batch into its own gradient buffers; the shard gradients are added
 * in shard order and AdamW (adam.rs:43-51) steps every tensor. Returns the loss sum of the last step. */
#include <pthread.h>

typedef struct {
    const orc_net_desc *d;
    int lo, hi, nnz;
    const float *const *params;
    float **grads;
    const int32_t *stm, *nstm, *buckets;
    const float *targets, *weights;
    float loss;
} orc_mt_job;

static void *orc_mt_worker(void *arg) {
    orc_mt_job *j = (orc_mt_job *)arg;
    const int n = j->hi - j->lo;
    const int tw = j->d->loss_kind == ORC_LOSS_SOFTMAX_CE ? j->d->dense_out[j->d->n_dense - 1] : 1;
    j->loss = n > 0 ? orc_net_forward_backward(j->d, n, j->params, j->grads, j->stm + (size_t)j->lo * j->nnz,
                                               j->nstm ? j->nstm + (size_t)j->lo * j->nnz : NULL,
                                               j->buckets ? j->buckets + j->lo : NULL, j->targets + (size_t)j->lo * tw,
                                               j->weights ? j->weights + j->lo : NULL, NULL)
                    : 0.0f;
    return NULL;
}

float orc_train_steps_mt(const orc_net_desc *d, int batch, float *const *params, float *const *mom, float *const *vel,
                         const int32_t *stm, const int32_t *nstm, const int32_t *buckets, const float *targets,
                         const float *weights, int threads, int steps, float lr) {
    const int nt = orc_net_num_tensors(d);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    orc_mt_job *jobs = (orc_mt_job *)calloc((size_t)threads, sizeof(orc_mt_job));
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        jobs[t].grads = (float **)calloc((size_t)nt, sizeof(float *));
        for (int i = 0; i < nt; i++) jobs[t].grads[i] = (float *)calloc(orc_net_tensor_size(d, i), sizeof(float));
    }
    float loss = 0.0f;
    for (int s = 0; s < steps; s++) {
        for (int t = 0; t < threads; t++) {
            orc_mt_job *j = &jobs[t];
            j->d = d, j->nnz = d->nnz, j->params = (const float *const *)params;
            j->lo = (int)((long long)batch * t / threads), j->hi = (int)((long long)batch * (t + 1) / threads);
            j->stm = stm, j->nstm = nstm, j->buckets = buckets, j->targets = targets, j->weights = weights;
            for (int i = 0; i < nt; i++) memset(j->grads[i], 0, orc_net_tensor_size(d, i) * sizeof(float));
            if (threads > 1) pthread_create(&tid[t], NULL, orc_mt_worker, j);
            else orc_mt_worker(j);
        }
        loss = 0.0f;
        for (int t = 0; t < threads; t++) {
            if (threads > 1) pthread_join(tid[t], NULL);
            loss += jobs[t].loss;
        }
        for (int i = 0; i < nt; i++) {
            const size_t n = orc_net_tensor_size(d, i);
            float *g = jobs[0].grads[i];
            for (int t = 1; t < threads; t++)
                for (size_t q = 0; q < n; q++) g[q] += jobs[t].grads[i][q];
            orc_adamw(params[i], mom[i], vel[i], g, n, 1.0f / (float)batch, lr, 0.01f, 0.9f, 0.999f, -1.98f, 1.98f);
        }
    }
    for (int t = 0; t < threads; t++) {
        for (int i = 0; i < nt; i++) free(jobs[t].grads[i]);
        free(jobs[t].grads);
    }
    free(jobs), free(tid);
    return loss;
}
