// Batch preparation on the device: 32-byte position records (already in HBM) -> the five batch tensors.
// Same arithmetic as host_inputs.cpp (SparseInputType::map_features, crates/bullet_lib/src/game/inputs/
// {chess768,chess_buckets,ataxx147,factorised}.rs; OutputBuckets::bucket, game/outputs.rs:24-31; the make_mapper
// closure, crates/bullet_lib/src/value.rs:77-117) with one wave per record: lane `sq` owns square `sq`, its slot in
// the index list is the number of occupied squares below it -- the ascending-square order of the reference's
// bit-scan iterator (examples/tests/test4.rs:23-36). Indices, terminators and output buckets are bit-exact with the
// host mapper; the sigmoid of the score target goes through a double-precision exp rounded to float, which
// equals glibc's expf except for results within ~0.002 ulp of a rounding tie (<= 1 ulp apart then).
// The host then only reads the file: 32 B per position cross PCIe instead of 268 B, and the training loop is no
// longer bounded by the host mapper threads.
#include "common.h"

namespace {

constexpr int MAP_WAVES = 4;  // records per workgroup

// Records the host mapper would reject (bt_map_batch: too many pieces, piece types 6 / 7, king squares > 63, results
// > 2, fewer than two pieces with output buckets) are counted here and mapped to an EMPTY list with zero weight and
// target, so a corrupt record can neither write outside its slots nor push an out-of-range bucket into the kernels
// behind it. bt_map_batch_device_errors reads and clears the counter (bt_train_run does after every superbatch).
__device__ int g_map_errors;

__device__ __forceinline__ float sigmoid_f32(float x) {  // value.rs:66-68: 1 / (1 + exp(-x)) in f32
    const float e = (float)exp((double)(-x));
    return 1.0f / (1.0f + e);
}

__global__ __launch_bounds__(MAP_WAVES * 64) void map_batch_kernel(bt_input_desc d, bt_map_opts o,
                                                                   const uint8_t *__restrict__ records, int count,
                                                                   int nnz, int32_t *__restrict__ stm,
                                                                   int32_t *__restrict__ ntm,
                                                                   int32_t *__restrict__ buckets,
                                                                   float *__restrict__ targets,
                                                                   float *__restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * MAP_WAVES + (threadIdx.x >> 6);
    if (k >= count) return;
    const uint64_t *r64 = reinterpret_cast<const uint64_t *>(records + (size_t)k * 32);  // records are 8-B aligned
    const uint64_t w0 = r64[0], w1 = r64[1], w2 = r64[2], w3 = r64[3];
    const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int32_t *s = stm + (size_t)k * nnz, *n = ntm + (size_t)k * nnz;
    const bool ataxx = d.kind == BT_INPUT_ATAXX147 || d.kind == BT_INPUT_ATAXX98;
    int cnt;
    int16_t score_i;
    int result;
    uint64_t occ = 0;
    bool bad = false;
    if (ataxx) {
        // AtaxxBoard: bbs[3] | score i16 | result u8 | ... (examples/ataxx.rs:8-18); features 49*pc + sq, the
        // other side swaps pieces 0 <-> 1 and keeps 2 (ataxx147.rs:18-27); Ataxx98 drops stm >= 98 (ataxx147.rs:51-56)
        const uint64_t bbs[3] = {w0, w1, w2};
        score_i = (int16_t)(w3 & 0xffffu), result = (int)((w3 >> 16) & 0xffu);
        const int npc = d.kind == BT_INPUT_ATAXX98 ? 2 : 3;
        // squares 49..63 do not exist on the 7x7 board: such a record leaves the input range (value.rs:80)
        bad = ((bbs[0] | bbs[1] | bbs[2]) >> 49) != 0 || __popcll(bbs[0]) + __popcll(bbs[1]) + (npc == 3 ? __popcll(bbs[2]) : 0) > nnz;
        int base = 0;
        for (int pc = 0; pc < npc && !bad; pc++) {
            if ((bbs[pc] >> lane) & 1ull) {
                const int slot = base + __popcll(bbs[pc] & below);
                s[slot] = 49 * pc + lane;
                n[slot] = (pc == 2) ? 49 * pc + lane : 49 * (1 - pc) + lane;
            }
            base += __popcll(bbs[pc]);
        }
        cnt = base;
    } else {
        // ChessBoard: occ u64 | pcs [u8;16] | score i16 | result u8 | ksq u8 | opp_ksq u8 | extra (test4.rs:8-18)
        occ = w0;
        score_i = (int16_t)(w3 & 0xffffu), result = (int)((w3 >> 16) & 0xffu);
        const int ksq = (int)((w3 >> 24) & 0xffu), opp_ksq = (int)((w3 >> 32) & 0xffu);
        int s_flip = 0, n_flip = 0, s_base = 0, n_base = 0;
        cnt = __popcll(occ);
        bad = cnt > 32 || (o.out_buckets > 1 && cnt < 2);
        if (d.kind != BT_INPUT_CHESS768) {  // chess_buckets.rs:28-33,82-88
            bad |= ksq > 63 || opp_ksq > 63;
            s_base = 768 * d.buckets[ksq & 63];
            n_base = 768 * d.buckets[opp_ksq & 63];
            if (d.kind == BT_INPUT_CHESS_BUCKETS_MIRRORED) {
                s_flip = (ksq & 7) > 3 ? 7 : 0;
                n_flip = (opp_ksq & 7) > 3 ? 7 : 0;
            }
        }
        const int shift = d.factorised ? 768 : 0;
        const bool mine = (occ >> lane) & 1ull;
        const int piece_no = __popcll(occ & below);  // low nibble first (test4.rs:23-36)
        const uint64_t word = piece_no < 16 ? w1 : w2;
        const int nib = (int)((word >> ((piece_no & 15) * 4)) & 15ull);
        bad |= __any(mine && piece_no < 32 && (nib & 7) > 5);  // piece types 6, 7: the index would leave its 768 block
        if (mine && !bad) {
            const int side = nib >> 3, type_off = (nib & 7) << 6;
            const int fs = s_base + ((side * 384 + type_off + lane) ^ s_flip);  // chess768.rs:20-30
            const int fn = n_base + (((1 - side) * 384 + type_off + (lane ^ 56)) ^ n_flip);
            if (d.factorised) {  // factorised.rs:32-48: the bucketed index, then the same piece-square unbucketed
                s[2 * piece_no] = shift + fs, n[2 * piece_no] = shift + fn;
                s[2 * piece_no + 1] = fs % 768, n[2 * piece_no + 1] = fn % 768;
            } else {
                s[piece_no] = fs, n[piece_no] = fn;
            }
        }
        if (d.factorised) cnt *= 2;
    }
    bad |= result > 2;  // value.rs:101 (WDL: a panic); scalar targets too, on purpose (stricter than value.rs:112, see bullet_amd.h)
    if (lane != 0) return;
    if (bad) {
        atomicAdd(&g_map_errors, 1);
        s[0] = n[0] = -1;
        buckets[k] = 0, weights[k] = 0.0f;
        const int tw = o.wdl_output ? 3 : 1;
        for (int j = 0; j < tw; j++) targets[tw * (size_t)k + j] = 0.0f;
        return;
    }
    if (cnt < nnz) s[cnt] = n[cnt] = -1;  // value.rs:86-89; slots behind the terminator stay as they were
    int bucket = 0;
    if (o.out_buckets > 1 && !ataxx) {  // MaterialCount<N>, outputs.rs:24-31 (u8 arithmetic)
        const int divisor = (32 + o.out_buckets - 1) / o.out_buckets;
        bucket = (int)((uint8_t)((uint8_t)__popcll(occ) - 2) / (uint8_t)divisor);
    }
    buckets[k] = bucket;
    weights[k] = 1.0f;
    if (o.wdl_output) {  // value.rs:96-101
        float *t = targets + 3 * (size_t)k;
        t[0] = t[1] = t[2] = 0.0f;
        t[result] = 1.0f;
    } else {  // value.rs:103-115
        float score = (float)score_i;
        if (o.use_win_rate_model) {
            const float p = (score - 270.0f) / 380.0f, pm = (-score - 270.0f) / 380.0f;
            score = 0.5f * (1.0f + sigmoid_f32(p) - sigmoid_f32(pm));
        } else {
            score = sigmoid_f32((1.0f / o.eval_scale) * score);
        }
        const float res = (float)result / 2.0f;
        targets[k] = o.blend * res + (1.0f - o.blend) * score;
    }
}

}  // namespace

extern "C" int bt_map_batch_device(void *stream, const bt_input_desc *d, const bt_map_opts *o, const void *records_dev,
                                   int count, int32_t *stm, int32_t *nstm, int32_t *buckets, float *targets,
                                   float *weights) {
    BT_REQUIRE(d && o && records_dev && stm && nstm && buckets && targets && weights, "bt_map_batch_device: null argument");
    BT_REQUIRE(count >= 0, "bt_map_batch_device: negative count");
    BT_REQUIRE(o->wdl_output || (o->blend >= 0.0f && o->blend <= 1.0f), "WDL proportion must be in [0, 1]");  // value.rs:114
    BT_REQUIRE((reinterpret_cast<uintptr_t>(records_dev) & 7) == 0, "bt_map_batch_device: records must be 8-byte aligned");
    if (count == 0) return BT_OK;
    BT_LAUNCH(BTK_FT_INDEX, bt_stream(stream), map_batch_kernel, dim3(bt_div_up(count, MAP_WAVES)), dim3(MAP_WAVES * 64), 0,
              *d, *o, static_cast<const uint8_t *>(records_dev), count, bt_input_max_active(d), stm, nstm, buckets, targets,
              weights);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

extern "C" int bt_map_batch_device_errors(void *stream, int *count) {
    BT_REQUIRE(count, "bt_map_batch_device_errors: null argument");
    hipStream_t st = bt_stream(stream);
    int zero = 0;
    *count = 0;
    BT_HIP(hipMemcpyFromSymbolAsync(count, HIP_SYMBOL(g_map_errors), sizeof(int), 0, hipMemcpyDeviceToHost, st));
    BT_HIP(hipStreamSynchronize(st));
    if (*count) {
        BT_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_map_errors), &zero, sizeof(int), 0, hipMemcpyHostToDevice, st));
        BT_HIP(hipStreamSynchronize(st));
    }
    return BT_OK;
}
