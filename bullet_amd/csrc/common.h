// Shared helpers for libbullet_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include <functional>
#include <vector>

#include "../../include/bullet_amd.h"

int bt_internal_sf_source(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads, bt_sf_filter filter,
                          void *user, uint64_t shuffle_seed, std::function<int(std::vector<uint8_t> &)> *out_refill);
// viribinpack.cpp: a source of shuffled 32-byte records for bt_reader (each call delivers one full shuffle buffer)
int bt_internal_viri_source(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                            bt_viri_filter filter, void *user, const bt_viri_builtin_filter *builtin,
                            uint64_t shuffle_seed, std::function<int(std::vector<uint8_t> &)> *out_refill);

// montybinpack.cpp: the same kind of source over montyformat games
int bt_internal_monty_source(const char *const *paths, int n_paths, size_t buffer_size_mb, int threads,
                             bt_monty_filter filter, void *user, uint64_t shuffle_seed,
                             std::function<int(std::vector<uint8_t> &)> *out_refill);

void bt_set_error(const char *fmt, ...);

// comm.cpp: a communicator is either RCCL (rccl = the ncclComm_t, with a stream of its own for the collectives and
// events to fence it against the caller's stream) or the caller's hooks (collectives then go on the caller's stream)
struct bt_comm {
    int rank = 0, world = 1;
    void *rccl = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool hooked = false;
    bt_comm_hooks hooks = {};
};

#define BT_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            bt_set_error("%s: %s", hipGetErrorName(e_), hipGetErrorString(e_));   \
            return BT_ERR_HIP;                                                    \
        }                                                                         \
    } while (0)

#define BT_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            bt_set_error(__VA_ARGS__); \
            return BT_ERR_INVALID;     \
        }                              \
    } while (0)

#define BT_LAUNCH_CHECK() BT_HIP(hipGetLastError())

static inline hipStream_t bt_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline int bt_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t bt_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- per-kernel HIP-event profiler (bt_profile_* in the C ABI) ----
// Every kernel launch goes through BT_LAUNCH(class, stream, kernel, ...): when profiling is on, a pair of events
// is recorded on the kernel's own stream around the launch; bt_profile_stop() sums the elapsed times per KERNEL,
// under the kernel's own name with its template arguments, as rocprofv3 prints it (the launch site registers the
// spelling of its `kernel` argument once: bt_prof_register). The class ids below only pair a completion event with
// the launch it is meant to ride on (bt_attach_stop_event).
enum BtKernelId {
    BTK_FT_FWD_GENERIC = 0,
    BTK_FT_FWD_LDS,
    BTK_FT_INDEX,
    BTK_FT_BWD_MAKE_G,
    BTK_FT_BWD_GATHER,
    BTK_FT_BWD_LDS,
    BTK_FT_BWD_MERGE,
    BTK_COLSUM_PARTIAL,
    BTK_COLSUM_FINAL,
    BTK_DENSE_FWD,
    BTK_DENSE_FWD_SMALL,
    BTK_DENSE_BWD_SMALL,
    BTK_DENSE_BWD_DX,
    BTK_DENSE_BWD_DW,
    BTK_SUM_PARTIALS,
    BTK_SIGMOID_MSE,
    BTK_ADAMW,
    BTK_COUNT
};
int bt_prof_register(const char *kernel_expr);  // -> profiler id of the kernel spelled `kernel_expr`
bool bt_prof_enabled(int id);
bool bt_prof_active();  // the calling thread's profiler is armed (bt_profile_start .. bt_profile_stop)
void bt_prof_record(int id, hipStream_t s, bool begin);
struct BtProfScope {
    int id;
    hipStream_t s;
    bool on;
    BtProfScope(int id_, hipStream_t s_, bool on_) : id(id_), s(s_), on(on_) {  // on_: bt_prof_enabled(id), asked ONCE per launch
        if (on) bt_prof_record(id, s, true);
    }
    ~BtProfScope() {
        if (on) bt_prof_record(id, s, false);
    }
};
// A completion event can ride on a kernel's own dispatch packet (hipExtLaunchKernelGGL stop event) instead of a
// separate hipEventRecord packet, which costs ~6 us of bubble in front of the next kernel of the stream:
// bt_attach_stop_event(id, ev) arms it for the calling thread's next launch of kernel `id`.
void bt_attach_stop_event(int kernel_id, hipEvent_t ev);
hipEvent_t bt_take_stop_event(int kernel_id);
// Profiled launches carry their two timing events on the kernel's own dispatch packet as well (start / stop of
// hipExtLaunchKernelGGL): the elapsed time is the kernel's execution time, and no extra packet sits in front of the
// next kernel (bracketing with hipEventRecord measured the waits in front of a kernel as well). A launch that also
// carries a completion event records that one separately while it is being timed.
bool bt_prof_take_pair(int id, hipEvent_t *a, hipEvent_t *b);
// While the calling thread captures a step into a hipGraph (net.hip), launches are plain: no timing, and a completion
// event that would ride on the packet is dropped (the capturing code records it after the graph's launch instead).
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel) and size reached so far, not once per launch
int bt_set_max_lds(const void *kernel, size_t bytes);
bool bt_capturing();
void bt_set_capturing(bool on);
#define BT_LAUNCH(id, stream, kernel, grid, block, lds, ...)                                        \
    do {                                                                                            \
        static const int pk_ = bt_prof_register(#kernel);                                           \
        const bool cap_ = bt_capturing();                                                           \
        hipEvent_t stop_ev_ = bt_take_stop_event(id), pa_ = nullptr, pb_ = nullptr;                 \
        if (cap_) stop_ev_ = nullptr;                                                               \
        const bool prof_ = !cap_ && bt_prof_enabled(pk_);                                           \
        if (prof_ && bt_prof_take_pair(pk_, &pa_, &pb_)) {                                          \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, pa_, pb_, 0, __VA_ARGS__);      \
            if (stop_ev_) (void)hipEventRecord(stop_ev_, stream); /* timed: the completion event gets its own packet */ \
        } else {                                                                                    \
            BtProfScope prof_scope_(pk_, stream, prof_);                                            \
            if (stop_ev_)                                                                           \
                hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, stop_ev_, 0, __VA_ARGS__); \
            else                                                                                    \
                hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                  \
        }                                                                                           \
    } while (0)

// ---- internal cross-file entry points (not in the public header)
bool bt_internal_ft_fused_ok(const bt_ft_desc *d, int head_rows);
// FT forward that applies a one-output dense head to each tile while it is in registers: zpart receives
// bt_internal_ft_head_parts(d) partial dot products per position ([part][batch]), summed by bt_internal_head_finish
bool bt_internal_ft_forward_head_ok(const bt_ft_desc *d, int head_rows);
int bt_internal_ft_head_parts(const bt_ft_desc *d);
int bt_internal_ft_forward_head(void *stream, const bt_ft_desc *d, const float *w, const float *bias,
                                const int32_t *const *idx, float *out, float *saved, const float *w1,
                                const int32_t *bucket, int head_rows, float *zpart, void *workspace,
                                size_t workspace_bytes);
// the two halves of bt_ft_build_index: segment entry lists (needed by the forward of a segment-grouped table; no-op
// otherwise) and the per-chunk sorted lists (backward only); bt_internal_ft_grouped: does the plan use the lists?
// (done: optional completion event that rides on the last kernel's dispatch packet; grouped tables only)
int bt_internal_ft_build_groups(void *stream, const bt_ft_desc *d, const int32_t *const *idx, void *workspace,
                                size_t workspace_bytes, hipEvent_t done = nullptr);
int bt_internal_ft_build_sorted(void *stream, const bt_ft_desc *d, const int32_t *const *idx, void *workspace,
                                size_t workspace_bytes, hipEvent_t done = nullptr);
bool bt_internal_ft_grouped(const bt_ft_desc *d);
// y = f(sum of the partials in order + bias[bucket]); sigmoid-MSE, its gradient, gz and loss partials like
// bt_internal_dense_forward_loss
int bt_internal_head_finish(void *stream, const bt_dense_desc *d, const float *zpart, int nparts, const float *bias,
                            const int32_t *bucket, const float *targets, const float *entry_weights, float *y,
                            float *grad_y, float *gz_out, float *loss_partials, int *n_loss_partials,
                            float *gz_parts = nullptr);
// head: optional. When given (and bt_internal_ft_fused_head_grad_ok), the FT backward also produces the head's own
// gradients from the activation tile it stages -- dw1, db1 (from gz_parts [n_loss_partials][head_rows], the
// per-workgroup sums bt_internal_head_finish leaves) and the final loss sum -- and no other kernel has to.
// `step`: AdamW applied to every gradient element right where the merge kernel has just formed it (the lowered
// training step of a single device, bt_net_train_step): tensors l0w, l0b, l1w, l1b whose .grad pointers are dw, dbias,
// dw1, db1; adj / lr as in bt_adamw. The gradients are still written.
struct bt_ft_fused_adamw {
    bt_adamw_tensor t[4];
    const float *adj, *lr;
};
struct bt_ft_head_grads {
    float *dw1, *db1;
    const float *gz_parts, *loss_partials;
    int n_loss_partials;
    float *loss_sum;
    const bt_ft_fused_adamw *step = nullptr;
    hipEvent_t done = nullptr;  // rides on the merge kernel's dispatch packet
};
// The same for nets whose backward pass ends in ft_bwd_merge / ft_bwd_merge_grp (multi-layer nets): the table and its
// bias are updated where their gradients are merged, the other tensors (their gradients are complete before the FT
// backward starts) by extra blocks of the same launch. Armed for the calling thread's next LDS-path FT backward.
struct bt_ft_step_all {
    bt_ft_fused_adamw base;  // t[0] = table, t[1] = FT bias
    int n_extra;
    bt_adamw_tensor extra[BT_ADAMW_MAX_TENSORS];
    hipEvent_t done = nullptr;  // rides on the merge kernel's dispatch packet (may be null)
    hipEvent_t wait = nullptr;  // the stream waits for it right before the merge launch: gradients of the extra
                                // tensors that were formed on another stream (bt_internal_dense_set_side)
};
bool bt_internal_ft_step_ok(const bt_ft_desc *d);
void bt_internal_ft_arm_step(const bt_ft_step_all *step);
// AdamW on one element: adam.rs:43-51, in this exact order (shared by adamw_kernel and the fused merge)
// No fma contraction: the op is inlined into two kernels and must round the same way in both (and as the CPU oracle,
// built with -ffp-contract=off, does).
__device__ __forceinline__ void bt_adam_op(float grad, float rate, float &p, float &m, float &v, const bt_adamw_tensor &t) {
#pragma clang fp contract(off)
    p *= 1.0f - t.decay * rate;
    m = t.beta1 * m + (1.0f - t.beta1) * grad;
    v = t.beta2 * v + (1.0f - t.beta2) * grad * grad;
    const float val = m / (__fsqrt_rn(v) + 0.00000001f);
    p -= rate * val;
    p = fminf(fmaxf(p, t.min_weight), t.max_weight);
}
bool bt_internal_ft_fused_head_grad_ok(const bt_ft_desc *d, int head_rows);
int bt_internal_ft_backward_fused(void *stream, const bt_ft_desc *d, const float *gz, const float *w1,
                                  const int32_t *bucket, int head_rows, const float *out, const float *saved, float *dw,
                                  float *dbias, void *workspace, size_t workspace_bytes,
                                  const bt_ft_head_grads *head = nullptr);
// dense head (out == 1) backward with optional extras: gz_out [B] receives grad_y * f'(y); when loss_partials is
// given, block (0,0) also sums them in order into loss_sum[0]. light: the small-footprint kernel shape that can share
// a CU with a resident FT-backward workgroup (only without grad_x / gz_out)
int bt_internal_dense_backward_head(void *stream, const bt_dense_desc *d, const float *w, const float *x,
                                    const int32_t *bucket, const float *y, const float *grad_y, float *dw, float *dbias,
                                    float *grad_x, float *gz_out, const float *loss_partials, int n_loss_partials,
                                    float *loss_sum, void *workspace, size_t workspace_bytes, bool light);
bool bt_internal_dense_head_ok(const bt_dense_desc *d);
// mid-size bucketed layers walk the batch in bucket order when a (stable) permutation is installed for the calling
// thread: wants_perm says whether a layer would use one, bucket_perm builds it, set_perm(nullptr) removes it
bool bt_internal_dense_wants_perm(const bt_dense_desc *d);
size_t bt_internal_bucket_perm_bytes(int batch, int nb);
int bt_internal_bucket_perm(void *stream, const int32_t *bucket, int batch, int nb, void *buf);
void bt_internal_dense_set_perm(void *buf, int batch, int nb);
// the two last (tiny, bucketed) layers + sigmoid-MSE + their backward pass in one kernel over the installed
// permutation's items: y [B], ga1 = gradient w.r.t. the activated input a1 [B][q1->in], all four gradients, loss sum
bool bt_internal_dense_tail_ok(const bt_dense_desc *q1, const bt_dense_desc *q2);
size_t bt_internal_dense_tail_workspace_bytes(const bt_dense_desc *q1);
int bt_internal_dense_tail(void *stream, const bt_dense_desc *q1, const bt_dense_desc *q2, const float *w2, const float *b2,
                           const float *w3, const float *b3, const float *a1, const float *targets, const float *ew,
                           float *y, float *ga1, float *dw2, float *db2, float *dw3, float *db3, float *loss_sum,
                           void *workspace, size_t workspace_bytes, int act_before = -1);
// act_before >= 0: ga1 receives the gradient w.r.t. the PRE-activation of the layer before the tail (its activation's
// derivative applied here); that layer's bt_dense_backward must then run between bt_internal_dense_set_gz_ready(true)
// and (false), and only if bt_internal_dense_takes_gz(desc) says its path forms gz in a kernel of its own
bool bt_internal_dense_takes_gz(const bt_dense_desc *d);
void bt_internal_dense_set_gz_ready(bool on);
// While a side stream is set (calling thread), the kernels of the fused tail and of the sorted MFMA layer whose results
// only the optimiser needs -- dense_tail_sum, dense_mfma_dw, dense_mfma_sum -- are issued on it, ordered behind the
// caller's stream through `fork` at the point where their inputs are complete. The caller joins the side stream before
// it uses the weight / bias gradients or the loss sum. (nullptr, nullptr) switches it off.
void bt_internal_dense_set_side(void *side_stream, hipEvent_t fork);
// dense head forward fused with sigmoid-MSE: writes y, grad_y, per-workgroup loss partials and (optionally) gz_out =
// grad_y * f'(y), the gradient w.r.t. the layer's pre-activation
int bt_internal_dense_forward_loss(void *stream, const bt_dense_desc *d, const float *w, const float *bias,
                                   const float *x, const int32_t *bucket, const float *targets,
                                   const float *entry_weights, float *y, float *grad_y, float *gz_out,
                                   float *loss_partials, int *n_loss_partials);
bool bt_internal_dense_forward_loss_ok(const bt_dense_desc *d);

// allocates the second index-buffer set of bt_net_hint_next_batch now (instead of on the first hint)
int bt_internal_net_enable_hints(bt_net *net);
void bt_internal_net_drop_graphs(bt_net *net);

// ---- device-side activation helpers (semantics: dfo.rs:47-103, pointwise.rs:54-139) ----
__device__ __forceinline__ float bt_act_fwd(int act, float v) {
    switch (act) {
    case BT_ACT_RELU: return fmaxf(v, 0.0f);
    case BT_ACT_CRELU: return fminf(fmaxf(v, 0.0f), 1.0f);
    case BT_ACT_SCRELU: {
        float c = fminf(fmaxf(v, 0.0f), 1.0f);
        return c * c;
    }
    case BT_ACT_SQRRELU: {
        float c = fmaxf(v, 0.0f);
        return c * c;
    }
    default: return v;
    }
}

// 4 derivatives behind one switch; raw v_sqrt_f32 (same result as sqrtf for normal inputs; denormal outputs,
// |o| < 1.2e-38, give 0 instead of ~1e-19 -- far below anything the gradient keeps)
__device__ __forceinline__ float4 bt_act_deriv4(int act, float4 o) {
    auto pos = [](float v) { return v > 0.0f ? 1.0f : 0.0f; };
    switch (act) {
    case BT_ACT_RELU: return make_float4(pos(o.x), pos(o.y), pos(o.z), pos(o.w));
    case BT_ACT_CRELU:
        return make_float4(pos(o.x) * pos(1.0f - o.x), pos(o.y) * pos(1.0f - o.y), pos(o.z) * pos(1.0f - o.z),
                           pos(o.w) * pos(1.0f - o.w));
    // (one compare per element: for every finite o the products of dfo.rs with their 0/1 factors come out as below,
    // signed zeros and the NaN of a negative o included)
    case BT_ACT_SCRELU: {
        auto h = [](float v) {
            const float s = __builtin_amdgcn_sqrtf(v);
            return !(v >= 1.0f) ? s + s : 0.0f;
        };
        return make_float4(h(o.x), h(o.y), h(o.z), h(o.w));
    }
    case BT_ACT_SQRRELU: {
        const float sx = __builtin_amdgcn_sqrtf(o.x), sy = __builtin_amdgcn_sqrtf(o.y);
        const float sz = __builtin_amdgcn_sqrtf(o.z), sw = __builtin_amdgcn_sqrtf(o.w);
        return make_float4(sx + sx, sy + sy, sz + sz, sw + sw);
    }
    default: return make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    }
}

// derivative computed from the activation OUTPUT, composed exactly as dfo.rs does:
// CReLU' = [o>0]*[1-o>0]; SCReLU' = half+half, half = CReLU'(o)*sqrt(o); SqrReLU' likewise with ReLU'.
__device__ __forceinline__ float bt_act_deriv(int act, float o) {
    switch (act) {
    case BT_ACT_RELU: return o > 0.0f ? 1.0f : 0.0f;
    case BT_ACT_CRELU: return (o > 0.0f ? 1.0f : 0.0f) * ((1.0f - o) > 0.0f ? 1.0f : 0.0f);
    case BT_ACT_SCRELU: {
        float half = ((o > 0.0f ? 1.0f : 0.0f) * ((1.0f - o) > 0.0f ? 1.0f : 0.0f)) * __fsqrt_rn(o);
        return half + half;
    }
    case BT_ACT_SQRRELU: {
        float half = (o > 0.0f ? 1.0f : 0.0f) * __fsqrt_rn(o);
        return half + half;
    }
    default: return 1.0f;
    }
}
