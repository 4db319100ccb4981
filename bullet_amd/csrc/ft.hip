// Sparse feature transformer: forward (gather-sum + bias + activation [+ pairwise]) and
// backward (sorted-by-feature segmented reduction, no float atomics).
//
// Reference semantics: SparseMatmul / SparseMatmulBwdMulti,
// crates/compiler/src/tensor/operation/linear/sparse.rs:86-119,151-187,253-271, with the device
// stop-at-first-invalid-index rule of crates/gpu/src/pointwise/write.rs:229-233,300-304.
//
// Data layout (all fp32 / int32):
//   W    [cols][stride]   column-major weights, feature j = contiguous column W[j*stride + offset ...]
//   idx  [B][nnz]         per input ("perspective"), valid indices first, first invalid index ends the list
//   out  [B][out_ld]      input k occupies columns [k*out_w, (k+1)*out_w)
//   saved[n_inputs][B][rows]  (pairwise only) activated halves before the product
//
// Entry e = k*B + b enumerates (input, position) pairs.  The backward splits the entries of each input
// into chunks of FT_CHUNK positions; `ft_index_kernel` builds, per chunk, the list of local entry ids
// sorted by feature (stable, so ascending entry order inside a feature) plus per-feature offsets.
// Every dW element is then summed chunk by chunk, entry by entry, in that fixed order: the result is
// bit-identical run to run (the reference's atomicAdd order is arbitrary).
#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "common.h"

namespace {

constexpr int FT_CHUNK = 256;  // positions per index chunk (local ids fit u16)

// entries of one chunk's sorted list: every pair plus <= 3 padding ids per feature (runs are 4-aligned)
__host__ __device__ inline int ft_sorted_cap(int nnz, int cols) { return (FT_CHUNK * nnz + 3 * cols + 7) & ~7; }

// Test switches, read from the environment when the library is loaded and again only when a test asks for it
// (bt_internal_reload_env): BT_FT_FORCE_GENERIC=1 runs the any-shape kernels on every shape, BT_FT_FORCE_GROUPED=1
// the segment-grouped ones wherever a segment is given, BT_FT_FWD_ITEM / BT_FT_BWD_CHUNKS override the item sizes,
// BT_FT_NO_PREDEC=1 makes the fused-head forward decode its lists itself (as the op-level forward does),
// BT_FT_DECODE_PARITY=0 keeps the decoded lists in list order.
struct FtEnv {
    bool force_generic = false, force_grouped = false, no_predec = false, decode_parity = true;
    int fwd_item = 0, bwd_chunks = 0;
    FtEnv() { reload(); }
    void reload() {
        auto num = [](const char *name) {
            const char *e = getenv(name);
            return e && e[0] ? atoi(e) : 0;
        };
        force_generic = num("BT_FT_FORCE_GENERIC") == 1, force_grouped = num("BT_FT_FORCE_GROUPED") != 0;
        fwd_item = num("BT_FT_FWD_ITEM"), bwd_chunks = num("BT_FT_BWD_CHUNKS");
        no_predec = num("BT_FT_NO_PREDEC") == 1;
        decode_parity = !(getenv("BT_FT_DECODE_PARITY") && getenv("BT_FT_DECODE_PARITY")[0] == '0');
    }
};
FtEnv g_ft_env;
bool force_generic() { return g_ft_env.force_generic; }

struct float4_t {
    float x, y, z, w;
};

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
    float v[1];
};
template <>
struct Vec<4> {
    float v[4];
};

template <int VEC>
__device__ __forceinline__ Vec<VEC> vload(const float *p) {
    Vec<VEC> r;
    if constexpr (VEC == 4) {
        float4 t = *reinterpret_cast<const float4 *>(p);
        r.v[0] = t.x, r.v[1] = t.y, r.v[2] = t.z, r.v[3] = t.w;
    } else {
        r.v[0] = *p;
    }
    return r;
}

template <int VEC>
__device__ __forceinline__ void vstore(float *p, const Vec<VEC> &r) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    } else {
        *p = r.v[0];
    }
}

// ------------------------------------------------------------------ generic forward
// One thread per (entry, VEC rows) -- any shape.  Used when the LDS-resident path does not apply.
template <int VEC, bool PAIR>
__global__ __launch_bounds__(256) void ft_fwd_generic(const float *__restrict__ w, const float *__restrict__ bias,
                                                      const int32_t *__restrict__ idx0,
                                                      const int32_t *__restrict__ idx1, float *__restrict__ out,
                                                      float *__restrict__ saved, bt_ft_desc d) {
    const int out_w = PAIR ? d.rows / 2 : d.rows;
    const int groups = out_w / VEC;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)d.n_inputs * d.batch * groups;
    if (t >= total) return;
    const int g = (int)(t % groups);
    const long long e = t / groups;
    const int k = (int)(e / d.batch);
    const int b = (int)(e % d.batch);
    const int32_t *idx = (k == 0 ? idx0 : idx1) + (size_t)b * d.nnz;
    const int r0 = g * VEC;

    Vec<VEC> a0, a1;
#pragma unroll
    for (int q = 0; q < VEC; q++) a0.v[q] = 0.0f, a1.v[q] = 0.0f;
    for (int i = 0; i < d.nnz; i++) {
        const int j = idx[i];
        if (j < 0 || j >= d.cols) break;  // write.rs:231
        const float *col = w + (size_t)j * d.stride + d.offset + r0;
        Vec<VEC> x = vload<VEC>(col);
#pragma unroll
        for (int q = 0; q < VEC; q++) a0.v[q] += x.v[q];
        if constexpr (PAIR) {
            Vec<VEC> y = vload<VEC>(col + out_w);
#pragma unroll
            for (int q = 0; q < VEC; q++) a1.v[q] += y.v[q];
        }
    }
    Vec<VEC> o;
    if constexpr (PAIR) {
        Vec<VEC> c0, c1;
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            float x0 = a0.v[q] + (bias ? bias[r0 + q] : 0.0f);
            float x1 = a1.v[q] + (bias ? bias[r0 + out_w + q] : 0.0f);
            c0.v[q] = bt_act_fwd(d.act, x0);
            c1.v[q] = bt_act_fwd(d.act, x1);
            o.v[q] = c0.v[q] * c1.v[q];
        }
        float *sv = saved + ((size_t)k * d.batch + b) * d.rows + r0;
        vstore<VEC>(sv, c0);
        vstore<VEC>(sv + out_w, c1);
    } else {
#pragma unroll
        for (int q = 0; q < VEC; q++) o.v[q] = bt_act_fwd(d.act, a0.v[q] + (bias ? bias[r0 + q] : 0.0f));
    }
    vstore<VEC>(out + (size_t)b * d.out_ld + (size_t)k * out_w + r0, o);
}

// ------------------------------------------------------------------ segment groups (bt_ft_desc.segment)
// King-bucketed input sets draw all features of an entry e = (input, position) from one segment of `segment`
// columns. Entries are counting-sorted by segment (stable: ascending entry id inside a segment) into `elist`;
// the LDS-resident kernels then run per (row tile, segment) on slices of that list:
//   chunks : <= FT_CHUNK consecutive elist entries of one segment  (unit of the sorted index / backward staging)
//   fitems : <= fwd_item consecutive elist entries of one segment   (one forward workgroup per row tile)
//   bitems : <= g_ch consecutive chunks of one segment              (one backward workgroup per row tile)
// All tables live in the workspace and are built on the device (no host sync); grids are sized for the worst
// case and workgroups beyond the counts exit at once.
struct FtGrp {
    const int *elist;     // [E]
    const int4 *items;    // forward: fitems {seg, start, count}; backward: bitems {seg, first chunk, n chunks}
    const int4 *chunks;   // {seg, start, count}
    const int *counts;    // [0] n fitems, [1] n bitems, [2] n chunks
    int *viol;            // indices found outside their entry's segment
    int tile_cols;        // = segment
};

constexpr int GRP_BLOCK = 256;

// pass 1: segment of every entry (from its first index; an empty list counts as segment 0) + per-block histogram
__global__ __launch_bounds__(GRP_BLOCK) void ft_group_count(const int32_t *__restrict__ idx0,
                                                            const int32_t *__restrict__ idx1, int batch, int nnz,
                                                            int cols, int seg_cols, int nseg, int E,
                                                            uint8_t *__restrict__ seg_of, int *__restrict__ blk_cnt) {
    __shared__ int hist[256];
    if ((int)threadIdx.x < nseg) hist[threadIdx.x] = 0;
    __syncthreads();
    const int e = blockIdx.x * GRP_BLOCK + threadIdx.x;
    if (e < E) {
        const int k = e >= batch ? 1 : 0, b = e - k * batch;
        const int j = (k ? idx1 : idx0)[(size_t)b * nnz];
        const int sg = (j >= 0 && j < cols) ? j / seg_cols : 0;
        seg_of[e] = (uint8_t)sg;
        atomicAdd(&hist[sg], 1);
    }
    __syncthreads();
    if ((int)threadIdx.x < nseg) blk_cnt[(size_t)blockIdx.x * nseg + threadIdx.x] = hist[threadIdx.x];
}

// pass 2: stable placement (ascending entry id inside a segment). Every block sums the per-block histograms for
// itself -- totals and the counts of the blocks before it, nblk * nseg integers that sit in L2 -- instead of waiting
// for a one-workgroup scan kernel in between (that kernel cost 20-35 us of pure latency on the step's critical
// path); block 0 also writes the chunk / item tables.
__global__ __launch_bounds__(GRP_BLOCK) void ft_group_place(const uint8_t *__restrict__ seg_of,
                                                            const int *__restrict__ blk_cnt, int nblk, int nseg, int E,
                                                            int fwd_item, int g_ch, int *__restrict__ elist,
                                                            int4 *__restrict__ chunks, int4 *__restrict__ fitems,
                                                            int4 *__restrict__ bitems, int *__restrict__ seg_items,
                                                            int *__restrict__ counts) {
    __shared__ int wcnt[GRP_BLOCK / 64][256];
    __shared__ int tot[256], before[256], seg_start[257];
    __shared__ int ch_start[257], fi_start[257], bi_start[257], fi_per[256], bi_per[256];  // block 0 only
    for (int i = threadIdx.x; i < nseg; i += GRP_BLOCK) tot[i] = 0, before[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nblk * nseg; i += GRP_BLOCK) {  // integer sums: order does not matter
        const int v = blk_cnt[i];
        if (v) {
            const int blk = i / nseg, sg2 = i - blk * nseg;
            atomicAdd(&tot[sg2], v);
            if (blk < (int)blockIdx.x) atomicAdd(&before[sg2], v);
        }
    }
    const int e = blockIdx.x * GRP_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sg = e < E ? (int)seg_of[e] : -1;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int rank = 0;
    for (int s2 = 0; s2 < nseg; s2++) {
        const unsigned long long m = __ballot(sg == s2);
        if (sg == s2) rank = __popcll(m & lt_mask);
        if (lane == 0) wcnt[wave][s2] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0;
        for (int s2 = 0; s2 < nseg; s2++) seg_start[s2] = a, a += tot[s2];
        seg_start[nseg] = a;
    }
    __syncthreads();
    if (e < E) {
        int pos = seg_start[sg] + before[sg] + rank;
        for (int w2 = 0; w2 < wave; w2++) pos += wcnt[w2][sg];
        elist[pos] = e;
    }
    if (blockIdx.x != 0) return;
    for (int s2 = threadIdx.x; s2 < nseg; s2 += GRP_BLOCK) {  // (the divisions in parallel, one segment per thread)
        const int n = tot[s2], nc = (n + FT_CHUNK - 1) / FT_CHUNK;
        const int nf = (n + fwd_item - 1) / fwd_item, nb = (nc + g_ch - 1) / g_ch;
        // equal shares instead of full items + a remainder: (n / nf) rounded up to whole passes; nf never grows
        fi_per[s2] = nf ? ((n + nf - 1) / nf + 127) / 128 * 128 : fwd_item;
        bi_per[s2] = nb ? (nc + nb - 1) / nb : g_ch;
        ch_start[s2] = nc, fi_start[s2] = nf ? (n + fi_per[s2] - 1) / fi_per[s2] : 0;
        bi_start[s2] = nb ? (nc + bi_per[s2] - 1) / bi_per[s2] : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // counts -> exclusive prefixes
        int c = 0, f = 0, bi = 0;
        for (int s2 = 0; s2 < nseg; s2++) {
            const int nc = ch_start[s2], nf = fi_start[s2], nb = bi_start[s2];
            ch_start[s2] = c, fi_start[s2] = f, bi_start[s2] = bi;
            c += nc, f += nf, bi += nb;
        }
        ch_start[nseg] = c, fi_start[nseg] = f, bi_start[nseg] = bi;
        counts[0] = f, counts[1] = bi, counts[2] = c, counts[3] = 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= nseg; i += GRP_BLOCK) seg_items[i] = bi_start[i];
    auto seg_of_slot = [&](const int *start, int i) {
        int s2 = 0;
        while (s2 + 1 < nseg && start[s2 + 1] <= i) s2++;
        return s2;
    };
    for (int i = threadIdx.x; i < ch_start[nseg]; i += GRP_BLOCK) {
        const int s2 = seg_of_slot(ch_start, i), j = i - ch_start[s2];
        chunks[i] = make_int4(s2, seg_start[s2] + j * FT_CHUNK, min(FT_CHUNK, tot[s2] - j * FT_CHUNK), 0);
    }
    for (int i = threadIdx.x; i < fi_start[nseg]; i += GRP_BLOCK) {
        const int s2 = seg_of_slot(fi_start, i), j = i - fi_start[s2];
        fitems[i] = make_int4(s2, seg_start[s2] + j * fi_per[s2], min(fi_per[s2], tot[s2] - j * fi_per[s2]), 0);
    }
    for (int i = threadIdx.x; i < bi_start[nseg]; i += GRP_BLOCK) {
        const int s2 = seg_of_slot(bi_start, i), j = i - bi_start[s2];
        const int nc = ch_start[s2 + 1] - ch_start[s2];
        bitems[i] = make_int4(s2, ch_start[s2] + j * bi_per[s2], min(bi_per[s2], nc - j * bi_per[s2]), 0);
    }
}

// ------------------------------------------------------------------ index build of the any-shape backward
// One workgroup per chunk.  W waves; wave w owns the contiguous entry sub-range w of the chunk and a
// private per-feature counter row, so the fill pass is race-free and stable without cross-wave ordering.
//   pass 0: n_valid per entry (first invalid index ends the list)
//   pass 1: per-wave histograms (LDS integer atomics: order-independent)
//   scan  : exclusive scan in (feature major, wave minor) order -> start[w][f]; offs[chunk][f] to global
//   pass 2: each wave walks its entries in order, 64/nnz_pad entries per step; lanes with equal feature
//           rank themselves by ballot (one ballot per feature-id bit) and take consecutive slots.
// (tables that fit the LDS kernels, segment-grouped or not, use ft_index_oct_kernel instead)
__global__ void ft_index_kernel(const int32_t *__restrict__ idx0, const int32_t *__restrict__ idx1, int batch,
                                int nnz, int cols, int chunks_per_input, int fbits, uint16_t *__restrict__ sorted,
                                uint32_t *__restrict__ offs) {
    extern __shared__ uint32_t lds[];
    const int nwaves = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    uint32_t *cnt = lds;                       // [nwaves][cols]
    uint32_t *tsum = lds + (size_t)nwaves * cols;  // [blockDim.x] scan scratch

    const int c = blockIdx.x;
    const int k = c / chunks_per_input;
    const int cb = c % chunks_per_input;
    const int b0 = cb * FT_CHUNK;
    const int n_e = min(FT_CHUNK, batch - b0);
    const int32_t *idx = (k == 0 ? idx0 : idx1) + (size_t)b0 * nnz;
    uint16_t *my_sorted = sorted + (size_t)c * ft_sorted_cap(nnz, cols);
    uint32_t *my_offs = offs + (size_t)c * (cols + 1);

    for (int i = threadIdx.x; i < nwaves * cols; i += blockDim.x) cnt[i] = 0;
    __syncthreads();

    // entries per wave-step: 64 / (nnz rounded up to a power of two <= 64)
    int slot = 1;
    while (slot < nnz) slot <<= 1;           // host guarantees nnz <= 64
    const int eps = 64 / slot;                // entries per step
    const int sub = lane / slot;              // which entry of the step
    const int li = lane % slot;               // index slot inside the entry
    const int per_wave = (n_e + nwaves - 1) / nwaves;
    const int e_lo = wave * per_wave;
    const int e_hi = min(n_e, e_lo + per_wave);

    // The wave's index lists are read once, all loads in flight together (IDX_REG steps live in registers; longer
    // sub-ranges -- few waves or very short lists -- reload the rest from L2 in pass 2).
    constexpr int IDX_REG = 16;
    const int nsteps = (per_wave + eps - 1) / eps;
    // index slot `li` of chunk-local entry e (callers guarantee e < e_hi, li < nnz)
    auto load_j = [&](int e) -> int { return idx[(size_t)e * nnz + li]; };
    int jr[IDX_REG];
#pragma unroll
    for (int st = 0; st < IDX_REG; st++) {
        const int e = e_lo + st * eps + sub;
        jr[st] = (st < nsteps && e < e_hi && li < nnz) ? load_j(e) : -1;
    }
    const unsigned long long seg = (slot == 64) ? ~0ull : (((1ull << slot) - 1) << (sub * slot));
    // list length of this lane's entry: the first invalid index ends it (write.rs:231)
    auto n_valid = [&](int e, int j) {
        const bool inval = (e < e_hi) && (li >= nnz || j < 0 || j >= cols);
        const unsigned long long m = __ballot(inval) & seg;
        return m ? (__ffsll((long long)m) - 1 - sub * slot) : slot;
    };

    // pass 1
    auto count_step = [&](int e, int j) {
        const int nvalid = n_valid(e, j);
        if (e < e_hi && li < nvalid) atomicAdd(&cnt[(size_t)wave * cols + j], 1u);
    };
#pragma unroll
    for (int st = 0; st < IDX_REG; st++)
        if (st < nsteps) count_step(e_lo + st * eps + sub, jr[st]);
    for (int st = IDX_REG; st < nsteps; st++) {
        const int e = e_lo + st * eps + sub;
        count_step(e, (e < e_hi && li < nnz) ? load_j(e) : -1);
    }
    __syncthreads();

    // scan: thread t owns features [t*per, (t+1)*per); block-wide exclusive scan of the padded totals
    {
        const int per = (cols + blockDim.x - 1) / blockDim.x;
        const int f_lo = min(cols, (int)threadIdx.x * per), f_hi = min(cols, f_lo + per);
        uint32_t s = 0;
        for (int f = f_lo; f < f_hi; f++) {
            uint32_t t = 0;
            for (int w2 = 0; w2 < nwaves; w2++) t += cnt[(size_t)w2 * cols + f];
            s += (t + 3) & ~3u;  // every feature's run is padded to a multiple of 4 entries
        }
        uint32_t incl = s;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) tsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - s;
        for (int w2 = 0; w2 < wave; w2++) run += tsum[w2];
        for (int f = f_lo; f < f_hi; f++) {
            my_offs[f] = run;
            for (int w2 = 0; w2 < nwaves; w2++) {
                uint32_t v = cnt[(size_t)w2 * cols + f];
                cnt[(size_t)w2 * cols + f] = run;
                run += v;
            }
            while (run & 3) my_sorted[run++] = (uint16_t)(FT_CHUNK << 7);  // pad the run; id FT_CHUNK = "add nothing"
        }
        if (f_hi == cols && f_lo < cols) my_offs[cols] = run;
    }
    __syncthreads();

    // pass 2
    uint32_t *start = cnt + (size_t)wave * cols;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    auto place_step = [&](int e, int j) {
        const int nvalid = n_valid(e, j);
        const bool active = (e < e_hi) && (li < nvalid);
        // lanes holding the same feature
        unsigned long long peers = __ballot(active);
        for (int bit = 0; bit < fbits; bit++) {
            unsigned long long bb = __ballot(active && ((j >> bit) & 1));
            peers &= ((j >> bit) & 1) ? bb : ~bb;
        }
        if (active) {
            const int rank = __popcll(peers & lt_mask);
            const int count = __popcll(peers);
            const uint32_t base = start[j];
            my_sorted[base + rank] = (uint16_t)(e << 7);  // stored as the byte offset of the entry's 128-B tile row
            if (rank == count - 1) start[j] = base + count;
        }
    };
#pragma unroll
    for (int st = 0; st < IDX_REG; st++)
        if (st < nsteps) place_step(e_lo + st * eps + sub, jr[st]);
    for (int st = IDX_REG; st < nsteps; st++) {
        const int e = e_lo + st * eps + sub;
        place_step(e, (e < e_hi && li < nnz) ? load_j(e) : -1);
    }
}

// ------------------------------------------------------------------ index build, "octet" layout (LDS backward)
// The LDS-resident backward sums every feature's run of a chunk with one 8-lane group; the eight groups of a wave
// work in lockstep, so the index hands each wave OCTETS: eight runs of (nearly) equal length, sorted by length and
// dealt to the 16 waves of the backward workgroup in serpentine order (balanced to within one run). A run longer
// than L = 2^ls blocks of 4 pairs is cut into sub-runs of <= L blocks ("parts") that go to different groups; L is
// the smallest power of two >= 8 for which a chunk has <= OCT_SLOTS parts. Each wave's work is one contiguous
// stream of 64-byte rows in the workspace, one uint2 per group:
//   four u16 byte offsets (entry * 128) of rows of the staged gradient tile, padding = the zero row. The low 7 bits
//   of an offset are free; in the LAST row of an octet those of the two low halves say where the sum the group has
//   built goes: x bit 0 = 1, x bits 1-2 = kind (1 = dW tile row `id`: a whole run, 2 = part slot `id`: a sub-run,
//   0 = nowhere: an empty group of the octet), id = x bits 3-6 | y bits 0-6 << 4
// plus per chunk the waves' row ranges and the list of cut features {feature, first slot, parts}, which the backward
// adds up in part order after the chunk. Every dW row has exactly one writer per phase and every sum a fixed order
// (pairs of a run in ascending entry order, parts in order), so the result does not depend on where a run lands.
constexpr int OCT_SLOTS = 64;                    // part slots per chunk (LDS: 64 x 128 B)
constexpr int OCT_FIX = OCT_SLOTS / 2;           // cut features per chunk (each has >= 2 parts)
constexpr int OCT_WAVES = 12;                    // waves of the backward workgroup that reduce (one stream of rows each)
constexpr int OCT_STAGERS = 4;                   // its other waves form the staged gradient tile (one per SIMD)
constexpr int OCT_IDX_THREADS = 512;
constexpr int OCT_META_BYTES = 64 + OCT_FIX * 8; // u16 wave_start[17], n_fix | uint2 fix[OCT_FIX]
constexpr uint32_t OCT_PAD2 = ((uint32_t)(FT_CHUNK << 7) << 16) | (uint32_t)(FT_CHUNK << 7);
__host__ __device__ inline int oct_item_cap(int tc) { return (tc + OCT_SLOTS + 7) & ~7; }
// rows of a chunk's streams: every octet is at most as long as the sum of its runs
__host__ __device__ inline int oct_rows_cap(int nnz, int tc) {
    return (FT_CHUNK * nnz + 3 * (tc + OCT_SLOTS)) / 4 + 4 * OCT_WAVES + 8;
}
__host__ __device__ inline size_t oct_index_lds_bytes(int tc) {
    const int IC = oct_item_cap(tc);
    return ((size_t)(OCT_IDX_THREADS / 64 + 1) * tc + 2 * IC + IC / 8 + 16 + 16 + 64 + 24) * 4 +
           ((size_t)((tc + 1) & ~1) + 3 * IC + IC / 8 + (size_t)(IC / 64 + 1) * 64) * 2;
}
__device__ __forceinline__ int oct_wave_of(int o) {  // serpentine deal of octet o (sorted by length) to the waves
    const int r = o / OCT_WAVES, i = o % OCT_WAVES;
    return (r & 1) ? OCT_WAVES - 1 - i : i;
}

// (min. 6 waves per SIMD = at most 80 VGPRs: a workgroup of this kernel then fits beside a resident FT-forward
// workgroup -- 4 x 88 registers per SIMD lane -- and the index build hides under the forward pass)
template <bool GRP>
__global__ __launch_bounds__(OCT_IDX_THREADS, 6) void ft_index_oct_kernel(const int32_t *__restrict__ idx0,
                                                                       const int32_t *__restrict__ idx1, int batch, int nnz,
                                                                       int cols, int chunks_per_input, int fbits,
                                                                       uint2 *__restrict__ stream,
                                                                       uint8_t *__restrict__ meta, FtGrp gp,
                                                                       int cols_total) {
    extern __shared__ __align__(16) uint32_t lds[];
    constexpr int T = OCT_IDX_THREADS, NW = T / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int IC = oct_item_cap(cols);
    // Pairs are counted by the parity of their position (the low / high half of a word: even / odd rows of the staged
    // tile lie in different halves of the LDS banks, and the two groups that share a ds_read_b128 pass collide when
    // their rows have equal parity). A run lists the positions of one parity first, then the other -- which one first
    // depends on the lane group the run lands in (even groups: even rows first), so that the two groups of a pass read
    // opposite halves for most of a run. Inside a parity the positions stay in ascending order: the order of every sum
    // is still a function of the batch alone.
    uint32_t *cnt = lds;                     // [NW][cols] pairs per (wave, feature), even | odd << 16 -> starts of the wave
    uint32_t *tpar = cnt + NW * cols;        // [cols] pairs of the feature, even | odd << 16
    uint32_t *idest = tpar + cols;           // [IC] dest word of an item (a run or a part of one)
    uint32_t *itembase = idest + IC;         // [IC] u16 index of the item's first offset in the chunk's stream
    uint32_t *omax = itembase + IC;          // [IC/8] length of an octet = its longest item
    uint32_t *tsum = omax + IC / 8;          // [16] scan scratch
    uint32_t *psum = tsum + 16;              // [16] parts a chunk would have for L = 8 << k
    uint32_t *keyv = psum + 16;              // [64] items per sort key -> first position of the key
    uint32_t *wstart = keyv + 64;            // [24] first row of each wave (17), [20] items, [22] cut features
    uint16_t *item0 = reinterpret_cast<uint16_t *>(wstart + 24);  // [cols] pairs of the feature, then its first item
    uint16_t *ilen = item0 + ((cols + 1) & ~1);  // [IC] blocks of an item
    uint16_t *ipos = ilen + IC;              // [IC] item -> position in length order
    uint16_t *pitem = ipos + IC;             // [IC] position -> item
    uint16_t *octrow = pitem + IC;           // [IC/8] first row of an octet inside its wave's stream
    uint16_t *bcnt = octrow + IC / 8;        // [IC/64 + 1][64] items per (64-item block, key)

    const int c = blockIdx.x;
    int n_e, col0 = 0, start0 = 0;
    const int32_t *idx = nullptr;
    if constexpr (GRP) {
        if (c >= gp.counts[2]) return;
        const int4 ch = gp.chunks[c];
        col0 = ch.x * cols, start0 = ch.y, n_e = ch.z;
    } else {
        const int k = c / chunks_per_input;
        const int b0 = (c % chunks_per_input) * FT_CHUNK;
        n_e = min(FT_CHUNK, batch - b0);
        idx = (k == 0 ? idx0 : idx1) + (size_t)b0 * nnz;
    }
    uint2 *my_stream = stream + (size_t)c * oct_rows_cap(nnz, cols) * 8;
    uint16_t *my_hdr = reinterpret_cast<uint16_t *>(meta + (size_t)c * OCT_META_BYTES);
    uint2 *my_fix = reinterpret_cast<uint2 *>(meta + (size_t)c * OCT_META_BYTES + 64);

    for (int i = threadIdx.x; i < NW * cols; i += T) cnt[i] = 0;
    for (int i = threadIdx.x; i < IC / 8; i += T) omax[i] = 0;
    for (int i = threadIdx.x; i < (IC / 64 + 1) * 64; i += T) bcnt[i] = 0;
    if (threadIdx.x < 16) psum[threadIdx.x] = 0;
    __syncthreads();

    // ---- the index lists of this wave's entries, read once (as ft_index_kernel)
    int slot = 1;
    while (slot < nnz) slot <<= 1;
    const int eps = 64 / slot, sub = lane / slot, li = lane % slot;
    const int per_wave = (n_e + NW - 1) / NW;
    const int e_lo = wave * per_wave, e_hi = min(n_e, e_lo + per_wave);
    constexpr int IDX_REG = 16;  // 512 threads: 32 entries per wave, 2 entries per step at nnz = 32 -> all in registers
    const int nsteps = (per_wave + eps - 1) / eps;
    auto load_j = [&](int e) -> int {
        if constexpr (GRP) {
            const int eid = gp.elist[start0 + e];
            const int k2 = eid >= batch ? 1 : 0;
            return (k2 ? idx1 : idx0)[(size_t)(eid - k2 * batch) * nnz + li];
        } else {
            return idx[(size_t)e * nnz + li];
        }
    };
    int jr[IDX_REG];
#pragma unroll
    for (int st = 0; st < IDX_REG; st++) {
        const int e = e_lo + st * eps + sub;
        jr[st] = (st < nsteps && e < e_hi && li < nnz) ? load_j(e) : -1;
    }
    const unsigned long long seg = (slot == 64) ? ~0ull : (((1ull << slot) - 1) << (sub * slot));
    auto n_valid = [&](int e, int j) {  // the first invalid index ends the list (write.rs:231)
        const bool inval = (e < e_hi) && (li >= nnz || j < 0 || j >= cols_total);
        const unsigned long long m = __ballot(inval) & seg;
        return m ? (__ffsll((long long)m) - 1 - sub * slot) : slot;
    };
    auto in_seg = [&](int jl) { return !GRP || (jl >= 0 && jl < cols); };
    auto count_step = [&](int e, int j) {
        const int nvalid = n_valid(e, j);
        if (e < e_hi && li < nvalid) {
            if (in_seg(j - col0)) atomicAdd(&cnt[(size_t)wave * cols + (j - col0)], (e & 1) ? 0x10000u : 1u);
            else if constexpr (GRP) atomicAdd(gp.viol, 1);
        }
    };
#pragma unroll
    for (int st = 0; st < IDX_REG; st++)
        if (st < nsteps) count_step(e_lo + st * eps + sub, jr[st]);
    for (int st = IDX_REG; st < nsteps; st++) {
        const int e = e_lo + st * eps + sub;
        count_step(e, (e < e_hi && li < nnz) ? load_j(e) : -1);
    }
    __syncthreads();

    // ---- A: pairs per feature; how many parts would a chunk have for L = 8, 16, ... 4096 blocks?
    const int per = (cols + T - 1) / T;
    const int f_lo = min(cols, (int)threadIdx.x * per), f_hi = min(cols, f_lo + per);
    {
        uint32_t pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int f = f_lo; f < f_hi; f++) {
            uint32_t t = 0;
            for (int w2 = 0; w2 < NW; w2++) {
                const uint32_t v = cnt[(size_t)w2 * cols + f];
                cnt[(size_t)w2 * cols + f] = t;  // pairs of f in the waves before w2 (per parity): ascending entry order
                t += v;
            }
            tpar[f] = t;
            t = (t & 0xffffu) + (t >> 16);
            item0[f] = (uint16_t)t;
            const uint32_t blocks = (t + 3) >> 2;
#pragma unroll
            for (int k = 0; k < 10; k++)
                if (blocks > (8u << k)) pc[k] += (blocks + (8u << k) - 1) >> (3 + k);
        }
#pragma unroll
        for (int k = 0; k < 10; k++)
            if (pc[k]) atomicAdd(&psum[k], pc[k]);
    }
    __syncthreads();
    int ls = 3;
    while (ls < 12 && psum[ls - 3] > (uint32_t)OCT_SLOTS) ls++;
    const uint32_t L = 1u << ls;

    // ---- B: items (whole runs and parts) in feature order; packed scan: items | part slots << 12 | cut features << 20
    {
        uint32_t s = 0;
        for (int f = f_lo; f < f_hi; f++) {
            const uint32_t blocks = ((uint32_t)item0[f] + 3) >> 2;
            const uint32_t nsub = blocks ? (blocks + L - 1) >> ls : 0;
            s += nsub | (nsub > 1 ? (nsub << 12) | (1u << 20) : 0u);
        }
        uint32_t incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) tsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - s;
        for (int w2 = 0; w2 < wave; w2++) run += tsum[w2];
        for (int f = f_lo; f < f_hi; f++) {
            const uint32_t blocks = ((uint32_t)item0[f] + 3) >> 2;
            const uint32_t nsub = blocks ? (blocks + L - 1) >> ls : 0;
            const uint32_t it0 = run & 0xfffu, ps0 = (run >> 12) & 0xffu, fx = run >> 20;
            item0[f] = (uint16_t)it0;
            for (uint32_t q = 0; q < nsub; q++) {
                ilen[it0 + q] = (uint16_t)(q + 1 < nsub ? L : blocks - (nsub - 1) * L);
                idest[it0 + q] = nsub > 1 ? (2u | ((ps0 + q) << 16)) : (1u | ((uint32_t)f << 16));  // kind | id << 16
            }
            if (nsub > 1) my_fix[fx] = make_uint2((uint32_t)f | (ps0 << 16), nsub);
            run += nsub | (nsub > 1 ? (nsub << 12) | (1u << 20) : 0u);
        }
        if (threadIdx.x == T - 1) wstart[20] = run & 0xfffu, wstart[22] = run >> 20;
    }
    __syncthreads();

    // ---- C: stable sort of the items by length, longest first (key = min(length, 63))
    const int n_items = (int)wstart[20];
    const int nb64 = (n_items + 63) >> 6;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int r_rank[2], r_key[2], r_len[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int blk = wave + j * NW, item = blk * 64 + lane;
        const bool valid = blk < nb64 && item < n_items;
        const int len = valid ? (int)ilen[item] : 0, key = min(len, 63);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 6; bit++) {
            const unsigned long long bb = __ballot(valid && ((key >> bit) & 1));
            peers &= ((key >> bit) & 1) ? bb : ~bb;
        }
        r_rank[j] = __popcll(peers & lt_mask), r_key[j] = key, r_len[j] = len;
        if (valid && r_rank[j] == 0) bcnt[blk * 64 + key] = (uint16_t)__popcll(peers);
    }
    __syncthreads();
    if (wave == 0) {  // per key: items in the blocks before each block, then the first position of the key (descending)
        uint32_t tot = 0;
        for (int blk = 0; blk < nb64; blk++) {
            const uint32_t v = bcnt[blk * 64 + lane];
            bcnt[blk * 64 + lane] = (uint16_t)tot;
            tot += v;
        }
        uint32_t incl = tot;  // sum over keys >= lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t dn = __shfl_down(incl, o);
            if (lane + o < 64) incl += dn;
        }
        keyv[lane] = incl - tot;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int blk = wave + j * NW, item = blk * 64 + lane;
        if (blk < nb64 && item < n_items) {
            const int pos = (int)keyv[r_key[j]] + (int)bcnt[blk * 64 + r_key[j]] + r_rank[j];
            ipos[item] = (uint16_t)pos, pitem[pos] = (uint16_t)item;
            atomicMax(&omax[pos >> 3], (uint32_t)r_len[j]);
        }
    }
    __syncthreads();

    // ---- D: rows of each wave's stream (an octet = as many rows as its longest item), waves rounded up to 4 rows
    const int n_oct = (n_items + 7) >> 3;
    if (wave == 0) {
        uint32_t rows = 0;
        if (lane < OCT_WAVES)
            for (int r = 0; r * OCT_WAVES < n_oct; r++) {
                const int o = r * OCT_WAVES + ((r & 1) ? OCT_WAVES - 1 - lane : lane);
                if (o < n_oct) octrow[o] = (uint16_t)rows, rows += omax[o];
            }
        rows = (rows + 3) & ~3u;
        uint32_t incl = rows;
#pragma unroll
        for (int o = 1; o < OCT_WAVES; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane < OCT_WAVES) wstart[lane + 1] = incl, my_hdr[lane + 1] = (uint16_t)incl;
        if (lane == 0) wstart[0] = 0, my_hdr[0] = 0, my_hdr[17] = (uint16_t)wstart[22];
    }
    __syncthreads();

    // ---- E: every row starts as padding (offsets of the zero row); where each item's offsets begin
    {
        const int R = (int)wstart[OCT_WAVES];
        uint4 *s4 = reinterpret_cast<uint4 *>(my_stream);
        for (int i = threadIdx.x; i < R * 4; i += T) s4[i] = make_uint4(OCT_PAD2, OCT_PAD2, OCT_PAD2, OCT_PAD2);
        for (int item = threadIdx.x; item < n_items; item += T) {
            const int pos = ipos[item], o = pos >> 3;
            itembase[item] = ((wstart[oct_wave_of(o)] + octrow[o]) * 8 + (pos & 7)) * 4;
        }
    }
    __syncthreads();  // (orders the padding stores before the stores below: same workgroup, vmcnt drained at the barrier)

    // ---- F: the pairs, each wave its own entries in order (stable inside a run)
    uint32_t *start = cnt + (size_t)wave * cols;
    uint16_t *s16 = reinterpret_cast<uint16_t *>(my_stream);
    const uint32_t in_mask = (4u << ls) - 1;
    // The entries of a step take their slots one entry after the other (a wave's LDS operations execute in order), so
    // a run holds its pairs in ascending entry order whatever the lanes do inside one atomic; equal features inside
    // ONE list share the entry, so their order among themselves does not matter. `start` is this wave's own cursor row.
    auto place_step = [&](int e, int jraw) {
        const int nvalid = n_valid(e, jraw);
        const int j = jraw - col0;
        const bool active = (e < e_hi) && (li < nvalid) && in_seg(j);
        uint32_t i = 0;
        for (int q = 0; q < eps; q++) {
            if (active && sub == q) i = atomicAdd(&start[j], (e & 1) ? 0x10000u : 1u);
            __builtin_amdgcn_wave_barrier();  // one entry after the other: the iterations are not to be merged
        }
        if (active) {
            // rank inside the run: the group's preferred parity first (the group = the sorted position of the run's first item)
            const uint32_t tp = tpar[j], odd = (uint32_t)e & 1u, pref = (uint32_t)ipos[item0[j]] & 1u;
            const uint32_t mine = odd ? i >> 16 : i & 0xffffu, before = pref ? tp >> 16 : tp & 0xffffu;
            i = odd == pref ? mine : before + mine;
            const uint32_t within = i & in_mask;
            s16[itembase[item0[j] + (i >> (ls + 2))] + (within >> 2) * 32 + (within & 3)] = (uint16_t)(e << 7);
        }
    };
#pragma unroll
    for (int st = 0; st < IDX_REG; st++)
        if (st < nsteps) place_step(e_lo + st * eps + sub, jr[st]);
    for (int st = IDX_REG; st < nsteps; st++) {
        const int e = e_lo + st * eps + sub;
        place_step(e, (e < e_hi && li < nnz) ? load_j(e) : -1);
    }
    __syncthreads();  // the offsets are in place (stores drained): now the control bits of every octet's last row
    for (int q = threadIdx.x; q < n_oct * 8; q += T) {
        const int o = q >> 3;
        const uint32_t row = wstart[oct_wave_of(o)] + octrow[o] + omax[o] - 1;
        const uint32_t dw = q < n_items ? idest[pitem[q]] : 0u, kind = dw & 3u, id = dw >> 16;
        unsigned *wp = reinterpret_cast<unsigned *>(my_stream + row * 8 + (q & 7));
        atomicOr(wp, 1u | (kind << 1) | ((id & 15u) << 3));  // (atomics act on L2, where this workgroup's stores have landed)
        if (id >> 4) atomicOr(wp + 1, id >> 4);
    }
}

// ------------------------------------------------------------------ generic backward
// g[k][b][r] = grad_out * f'(out)  (pairwise: product rule first), dfo.rs:30-40, binary.rs:101-123
template <int VEC, bool PAIR>
__global__ __launch_bounds__(256) void ft_bwd_make_g(const float *__restrict__ grad_out,
                                                     const float *__restrict__ out,
                                                     const float *__restrict__ saved, float *__restrict__ g,
                                                     bt_ft_desc d) {
    const int out_w = PAIR ? d.rows / 2 : d.rows;
    const int groups = out_w / VEC;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)d.n_inputs * d.batch * groups;
    if (t >= total) return;
    const int gi = (int)(t % groups);
    const long long e = t / groups;
    const int k = (int)(e / d.batch);
    const int b = (int)(e % d.batch);
    const int r0 = gi * VEC;
    const size_t oo = (size_t)b * d.out_ld + (size_t)k * out_w + r0;
    Vec<VEC> go = vload<VEC>(grad_out + oo);
    float *gp = g + (size_t)e * d.rows + r0;
    if constexpr (PAIR) {
        const float *sv = saved + (size_t)e * d.rows + r0;
        Vec<VEC> c0 = vload<VEC>(sv), c1 = vload<VEC>(sv + out_w), g0, g1;
#pragma unroll
        for (int q = 0; q < VEC; q++) {
            g0.v[q] = (go.v[q] * c1.v[q]) * bt_act_deriv(d.act, c0.v[q]);
            g1.v[q] = (go.v[q] * c0.v[q]) * bt_act_deriv(d.act, c1.v[q]);
        }
        vstore<VEC>(gp, g0);
        vstore<VEC>(gp + out_w, g1);
    } else {
        Vec<VEC> o = vload<VEC>(out + oo), r;
#pragma unroll
        for (int q = 0; q < VEC; q++) r.v[q] = go.v[q] * bt_act_deriv(d.act, o.v[q]);
        vstore<VEC>(gp, r);
    }
}

// One workgroup per feature; threads cover the rows; chunks and entries visited in ascending order.
template <int VEC>
__global__ void ft_bwd_gather(const float *__restrict__ g, const uint16_t *__restrict__ sorted,
                              const uint32_t *__restrict__ offs, float *__restrict__ dw, bt_ft_desc d,
                              int chunks_per_input) {
    const int f = blockIdx.x;
    const int nchunks = d.n_inputs * chunks_per_input;
    for (int r0 = threadIdx.x * VEC; r0 < d.rows; r0 += blockDim.x * VEC) {
        Vec<VEC> acc;
#pragma unroll
        for (int q = 0; q < VEC; q++) acc.v[q] = 0.0f;
        for (int c = 0; c < nchunks; c++) {
            const uint32_t s = offs[(size_t)c * (d.cols + 1) + f];
            const uint32_t e = offs[(size_t)c * (d.cols + 1) + f + 1];
            if (s == e) continue;
            const int k = c / chunks_per_input, cb = c % chunks_per_input;
            const size_t ebase = (size_t)k * d.batch + (size_t)cb * FT_CHUNK;
            const uint16_t *lst = sorted + (size_t)c * ft_sorted_cap(d.nnz, d.cols);
            for (uint32_t q2 = s; q2 < e; q2++) {
                const int el = lst[q2] >> 7;
                if (el >= FT_CHUNK) break;  // run padding
                Vec<VEC> x = vload<VEC>(g + (ebase + el) * d.rows + r0);
#pragma unroll
                for (int q = 0; q < VEC; q++) acc.v[q] += x.v[q];
            }
        }
        vstore<VEC>(dw + (size_t)f * d.stride + d.offset + r0, acc);
    }
}

// deterministic column sums of an [M][C] matrix: pass 1 = sequential sums over row blocks,
// pass 2 = sequential sum of the partials
__global__ void colsum_partial(const float *__restrict__ in, int M, int C, int rows_per_block,
                               float *__restrict__ partial) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    float s = 0.0f;
    for (int m = m0; m < m1; m++) s += in[(size_t)m * C + c];
    partial[(size_t)blockIdx.y * C + c] = s;
}

__global__ void colsum_final(const float *__restrict__ partial, int nblk, int C, float *__restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.0f;
    for (int i = 0; i < nblk; i++) s += partial[(size_t)i * C + c];
    out[c] = s;
}


// ================================================================== LDS-resident fast paths
// When (cols+1) * 32 rows * 4 B fits in LDS (Chess768: 96 KiB, Ataxx: 18 KiB) each workgroup owns a
// 32-row tile of the table for its whole life:
//   forward : the W tile is loaded once, then every (position, feature) gather is a 128-B LDS read
//             (ds_read_b64 x 16 lanes) instead of an L2 read -- ~8x the gather bandwidth of the L2 path;
//   backward: the dW tile is accumulated in LDS; per chunk of 256 positions the activated gradient tile is
//             staged once (one HBM read of grad_out/out per step) and every feature's sorted entry list
//             is summed from LDS in a fixed order.
// Row tile of a pairwise op = 16 rows of the first half + the matching 16 rows of the second half.
constexpr int LDS_RT = 32;
constexpr int LDS_THREADS = 1024;

__device__ __forceinline__ float row_ror8(float v) {  // the value of lane ^ 8 (rotation by 8 inside each 16-lane row)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));
}
template <int N>
__device__ __forceinline__ int row_share(int v) {  // broadcast lane N of each 16-lane row (DPP, no LDS traffic)
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + N, 0xF, 0xF, true);
}

// 16 bytes at an LDS byte address (an integer, not a pointer derived from the dynamic LDS symbol: the compiler leaves an
// add of the symbol's base, 0, on every address formed from the pointer)
typedef float lds_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_load4(uint32_t byte_addr) {
    const lds_f32x4 t = *(const __attribute__((address_space(3))) lds_f32x4 *)(uintptr_t)byte_addr;
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}

__device__ __forceinline__ int tile_row(bool pair, int rt, int half_rows, int local4) {
    // local4: float offset (multiple of 2 or 4) inside the 32-float tile -> global row
    if (!pair) return rt * LDS_RT + local4;
    return (local4 < 16) ? rt * 16 + local4 : half_rows + rt * 16 + (local4 - 16);
}

typedef unsigned lds_u4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void add4(float4 &a, const float4 &v) { a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w; }

// Forward: 8 lanes x float4 cover the 32-row tile of one position-perspective ("entry"), so a wave sums 8 entries
// at a time. The wave's index lists are validated once (stop rule, write.rs:231), turned into byte offsets of
// tile rows (masked slots -> the zero row) and parked in a per-wave LDS strip; the gather loop then costs, per 4
// features, 1 ds_read_b128 of offsets + 4 address adds + 4 ds_read_b128 gathers + 8 packed adds: 0.75 VALU per
// summed element. The sum runs over the features in list order, like the reference's loop (sparse.rs:100-113) -- or, with
// lists decoded ahead (PRE), in the order ft_fwd_decode_kernel<PARITY> gave them: the same rows, sorted by parity.
constexpr int FWD_EPW = 8;  // entries per wave per pass

// ds_read_b128 is served in four 16-lane groups, {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32; only
// lanes of one group can bank-conflict. Giving each group two whole 128-B rows (instead of four half rows)
// makes a conflict a coin flip on the two rows' parity: 1.5 LDS cycles per group on average instead of 1.75.
__device__ __forceinline__ void fwd_gather_slot(int lane, int &entry, int &slot) {
    const int m = lane & 31, h = lane >> 5;
    int q, sl;
    if (m < 4) q = 0, sl = m;
    else if (m < 12) q = 2, sl = m - 4;
    else if (m < 16) q = 0, sl = m - 8;
    else if (m < 20) q = 3, sl = m - 16;
    else if (m < 28) q = 1, sl = m - 20;
    else q = 3, sl = m - 24;
    entry = h * 4 + q, slot = sl;
}
__host__ __device__ constexpr int fwd_strip_words(int nround) { return nround * 32 + 4; }  // +4: bank skew

// HEAD: a one-output dense head (Matmul + Select of one row per position, matmul.rs:62-93, select.rs:12-26) is
// applied to the tile while it is still in registers: every (perspective, row tile) leaves one partial dot
// product per position in zpart[(k*ntiles + rt)][b]; a small finishing kernel adds them up in order. The
// [B][2H] activations are still written (both backward passes read them) but not read back by a head forward.
struct FtHeadSink {
    const float *w1;        // head weights, column-major [w1_rows][n_inputs*out_w]
    const int32_t *bucket;  // [B] or null
    int w1_rows;            // = number of output buckets (one output each)
    float *zpart;           // [n_inputs * ntiles][batch]
};

// The lane holding part gs ^ 4 of the same entry (fwd_gather_slot's map keeps an entry's two quads inside one 16-lane
// row, 4 or 12 lanes apart): two DPP row rotations and a select instead of a ds_bpermute through the LDS pipe.
__device__ __forceinline__ float fwd_partner(float v, int lane) {
    const float r4 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true));   // from lane - 4
    const float r12 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x12C, 0xF, 0xF, true));  // from lane + 4
    return (lane & 4) ? r12 : r4;
}

__device__ __forceinline__ float quad_sum(float v) {  // sum over the 4 lanes of an aligned quad, on every lane
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    return v;
}

// GRP (bt_ft_desc.segment): the workgroup is (row tile, forward item); the tile holds the item's segment of
// `tc` columns, entries come from the grouped list, and one more LDS row (all NaN) receives the slots whose
// index lies outside the segment, so a broken promise poisons the entry's output instead of passing silently.
// The forward's list decode, once per batch instead of once per row tile: 8 threads per entry, 4 slots each, as the
// decode map of ft_fwd_lds. offs[e][32]: byte offset of the tile row of every slot, masked slots (behind the first
// invalid index, write.rs:231) -> the zero row `cols`; nblk[e / 8]: the 4-slot blocks that hold a live slot for at
// least one of the group's eight entries (live blocks form a prefix in every entry).
// PARITY: an entry's live rows are listed by the parity of the row number -- even rows first for even entries, odd rows
// first for odd ones, each kind in list order, the masked slots behind them. One 16-lane service group of the forward's
// gathers reads a whole 128-B row for each of two neighbouring entries (fwd_gather_slot), and two rows of equal parity
// collide in the LDS banks: a coin flip in list order, rare in this order. The sum of an entry then runs over the same
// rows in another (still fixed) order.
template <bool PARITY>
__global__ __launch_bounds__(256) void ft_fwd_decode_kernel(const int32_t *__restrict__ idx0,
                                                            const int32_t *__restrict__ idx1, int batch, int nnz, int cols,
                                                            int E, lds_u4_t *__restrict__ offs, uint8_t *__restrict__ nblk) {
    const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const int e = t >> 3, sub = t & 7, eslot = lane >> 3;
    const int ec = min(e, E - 1), k = ec >= batch ? 1 : 0;
    const int32_t *ip = (k ? idx1 : idx0) + (size_t)(ec - k * batch) * nnz;
    bool inv[4];
    int r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = sub * 4 + j;
        r[j] = ip[min(p, nnz - 1)];
        inv[j] = p >= nnz || r[j] < 0 || r[j] >= cols;
    }
    const bool any = inv[0] | inv[1] | inv[2] | inv[3];
    const unsigned m = (unsigned)(__ballot(any) >> (eslot * 8)) & 0xFFu;
    bool kill = e >= E || (m & ((1u << sub) - 1u));
    const unsigned long long open = __ballot(!(kill || inv[0]));
    if constexpr (PARITY) {
        // positions: packed (even | odd << 16) counts of live rows, scanned over the entry's 8 lanes by DPP row shifts
        unsigned r4[4];
        bool lv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            kill |= inv[j];
            lv[j] = !kill;
            r4[j] = (unsigned)(kill ? cols : r[j]);
        }
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) c += lv[j] ? ((r4[j] & 1u) ? 0x10000u : 1u) : 0u;
        int sc = (int)c;
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x111, 0xF, 0xF, true);  // row_shr:1 ... 8: inclusive scan of a 16-lane row
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x112, 0xF, 0xF, true);
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x114, 0xF, 0xF, true);
        sc += __builtin_amdgcn_update_dpp(0, sc, 0x118, 0xF, 0xF, true);
        const unsigned t7 = (unsigned)row_share<7>(sc), t15 = (unsigned)row_share<15>(sc);
        const bool hi8 = (lane & 8) != 0;  // the row's second entry
        const unsigned tot = hi8 ? t15 - t7 : t7;
        const unsigned before = (hi8 ? (unsigned)sc - t7 : (unsigned)sc) - c;
        const bool odd_first = (e & 1) != 0;
        unsigned pos_e = (odd_first ? tot >> 16 : 0u) + (before & 0xffffu);
        unsigned pos_o = (odd_first ? 0u : tot & 0xffffu) + (before >> 16);
        unsigned *mine = reinterpret_cast<unsigned *>(offs + (size_t)min(e, E - 1) * 8);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool od = (r4[j] & 1u) != 0;
            const unsigned at = lv[j] ? (od ? pos_o : pos_e) : (unsigned)(sub * 4 + j);
            pos_o += (lv[j] && od) ? 1u : 0u, pos_e += (lv[j] && !od) ? 1u : 0u;
            if (e < E) mine[at] = r4[j] * (LDS_RT * 4);
        }
    } else {
        lds_u4_t o;
        kill |= inv[0], o.x = (unsigned)(kill ? cols : r[0]) * (LDS_RT * 4);
        kill |= inv[1], o.y = (unsigned)(kill ? cols : r[1]) * (LDS_RT * 4);
        kill |= inv[2], o.z = (unsigned)(kill ? cols : r[2]) * (LDS_RT * 4);
        kill |= inv[3], o.w = (unsigned)(kill ? cols : r[3]) * (LDS_RT * 4);
        if (e < E) offs[(size_t)e * 8 + sub] = o;
    }
    unsigned fold = (unsigned)open | (unsigned)(open >> 32);
    fold |= fold >> 16;
    fold |= fold >> 8;
    if (lane == 0 && e < E) nblk[e >> 3] = (uint8_t)__popc(fold & 0xFFu);
}

// (Tried: the entries put in an order in which the 128 entries of a workgroup's pass have equal list lengths -- a pass
// walks as many 4-slot blocks as its longest list -- by a counting sort on the side stream. At the fixture's piece counts
// the forward went from 70 to 62.5 us, at full occupancy from 69.5 to 72.5 (the indirection, scattered partial writes),
// and the sort needs more than one workgroup to stay off the critical path: not kept.)

// PRE: the lists were validated and decoded once per batch (ft_fwd_decode_kernel) instead of once per row tile here:
// pre.offs holds each entry's 32 strip words as the decode below would form them, pre.nblk the live 4-slot blocks of
// every aligned group of 8 entries. The same rows (in list order without PARITY: then the same bits).
struct FtPre {
    const lds_u4_t *offs;  // [E][8]
    const uint8_t *nblk;   // [ceil(E / 8)]
};
template <bool PAIR, bool ALIGNED4, int NROUND, bool HEAD, bool GRP, bool PRE = false>
__global__ __launch_bounds__(LDS_THREADS) void ft_fwd_lds(const float *__restrict__ w, const float *__restrict__ bias,
                                                          const int32_t *__restrict__ idx0,
                                                          const int32_t *__restrict__ idx1, float *__restrict__ out,
                                                          float *__restrict__ saved, bt_ft_desc d, int ntiles,
                                                          int entries_per_wg, FtHeadSink hs, FtGrp gp, FtPre pre) {
    static_assert(!PRE || (NROUND == 1 && !GRP), "pre-decoded lists: one round, not segment-grouped");
    extern __shared__ __align__(16) float wl[];  // [(tc + 1)][32]; row `tc` is all zero (target of masked slots)
    constexpr int SS = fwd_strip_words(NROUND);
    constexpr int TW = PAIR ? 16 : 32;  // output columns per perspective covered by this row tile
    constexpr int XR = GRP ? 2 : 1;     // extra rows behind the tile: zeros [, NaNs]
    const int rt = blockIdx.x % ntiles, grp = blockIdx.x / ntiles;
    const int half_rows = d.rows / 2;
    const int tc = GRP ? gp.tile_cols : d.cols;
    int col0 = 0, g_begin = 0, g_end = 0;
    if constexpr (GRP) {
        if (grp >= gp.counts[0]) return;
        const int4 it = gp.items[grp];
        col0 = it.x * tc, g_begin = it.y, g_end = it.y + it.z;
    }
    for (int q = threadIdx.x; q < tc * 8; q += LDS_THREADS) {
        const int f = q >> 3, s8 = q & 7;
        const float4 v = *reinterpret_cast<const float4 *>(w + (size_t)(col0 + f) * d.stride + d.offset +
                                                            tile_row(PAIR, rt, half_rows, s8 * 4));
        *reinterpret_cast<float4 *>(wl + f * LDS_RT + s8 * 4) = v;
    }
    if (threadIdx.x < LDS_RT) wl[tc * LDS_RT + threadIdx.x] = 0.0f;
    if constexpr (GRP)
        if (threadIdx.x < LDS_RT) wl[(tc + 1) * LDS_RT + threadIdx.x] = __int_as_float(0x7fc00000);
    float *w1t = wl + (tc + XR) * LDS_RT + (LDS_THREADS / 64) * FWD_EPW * SS;  // HEAD: [n_inputs][w1_rows][TW]
    if constexpr (HEAD) {
        const int ow = PAIR ? half_rows : d.rows;
        for (int q = threadIdx.x; q < d.n_inputs * hs.w1_rows * TW; q += LDS_THREADS) {
            const int c = q % TW, bk = (q / TW) % hs.w1_rows, k = q / (TW * hs.w1_rows);
            w1t[q] = hs.w1[(size_t)(k * ow + rt * TW + c) * hs.w1_rows + bk];
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // two lane -> (entry, part) maps: contiguous 8-lane groups while the index lists are decoded (ballots),
    // the bank-friendly map of fwd_gather_slot while rows are gathered, summed and stored
    const int sub = lane & 7, eslot = lane >> 3;
    int ge, gs;
    fwd_gather_slot(lane, ge, gs);
    uint32_t *wstrip = reinterpret_cast<uint32_t *>(wl + (tc + XR) * LDS_RT) + wave * FWD_EPW * SS;
    uint32_t *dstrip = wstrip + eslot * SS;        // written by the decode map
    const uint32_t *strip = wstrip + ge * SS;      // read by the gather map
    const int row = tile_row(PAIR, rt, half_rows, gs * 4);
    float4 bv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (bias) bv = *reinterpret_cast<const float4 *>(bias + row);
    // entries e = k*batch + b as 32-bit ints (host checks n_inputs*batch < 2^30); n_inputs <= 2, so the
    // (k, b) split is a compare + subtract -- no integer division in the hot loop
    const int E = d.n_inputs * d.batch;
    const int e_begin = GRP ? g_begin : grp * entries_per_wg;
    const int e_end = GRP ? g_end : min(E, e_begin + entries_per_wg);
    const int out_w = PAIR ? half_rows : d.rows;
    constexpr int PASS = (LDS_THREADS / 64) * FWD_EPW;
    // entry id at list position `pos` (GRP: through the grouped list, clamped to the item)
    auto ent = [&](int pos) -> int {
        if constexpr (GRP) return gp.elist[min(pos, e_end - 1)];
        else return min(pos, E - 1);
    };

    // raw indices of the NEXT pass are loaded while the current one is being summed. Unconditional, clamped
    // loads (validity is applied at decode time): with no branch around them the compiler keeps them in
    // flight behind a counted s_waitcnt instead of draining with vmcnt(0).
    auto load_idx = [&](int e2, int(&r)[NROUND][4]) {
        const int k2 = (e2 >= d.batch) ? 1 : 0;
        const int b2 = e2 - k2 * d.batch;
        const int32_t *ip = (k2 == 0 ? idx0 : idx1) + (size_t)b2 * d.nnz;
#pragma unroll
        for (int q = 0; q < NROUND; q++) {
            if constexpr (ALIGNED4) {
                const int4 v = *reinterpret_cast<const int4 *>(ip + min(q * 32 + sub * 4, d.nnz - 4));
                r[q][0] = v.x, r[q][1] = v.y, r[q][2] = v.z, r[q][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) r[q][j] = ip[min(q * 32 + sub * 4 + j, d.nnz - 1)];
            }
        }
    };
    // HEAD: the bucket of this lane's gather-map entry, fetched one pass ahead like the indices
    auto load_bucket = [&](int e2) {
        if (!HEAD || !hs.bucket) return 0;
        return (int)hs.bucket[e2 - ((e2 >= d.batch) ? d.batch : 0)];
    };
    int nxt[NROUND][4];
    const int eb0 = e_begin + wave * FWD_EPW;
    // PRE: this lane's four strip words of entry `pos` and the block count of the pass that starts at entry `pos0`
    auto load_pre = [&](int pos, int(&r)[NROUND][4]) {
        const lds_u4_t v = pre.offs[(size_t)min(pos, E - 1) * 8 + sub];
        r[0][0] = (int)v.x, r[0][1] = (int)v.y, r[0][2] = (int)v.z, r[0][3] = (int)v.w;
    };
    auto load_nblk = [&](int pos0) -> int { return PRE ? (int)pre.nblk[min(pos0, E - 1) >> 3] : 0; };
    int nxt_nblk = load_nblk(eb0);
    // GRP: entry ids are fetched two passes ahead (decode map d1/d2, gather map g0/g1), so the index and bucket
    // loads that depend on them are still issued one full pass before use
    int d1 = 0, d2 = 0, g0 = 0, g1 = 0;
    if constexpr (GRP) {
        d1 = ent(eb0 + PASS + eslot), d2 = ent(eb0 + 2 * PASS + eslot);
        g0 = ent(eb0 + ge), g1 = ent(eb0 + PASS + ge);
    }
    if constexpr (PRE) load_pre(eb0 + eslot, nxt);
    else load_idx(ent(eb0 + eslot), nxt);
    int nxt_bk = load_bucket(GRP ? g0 : ent(eb0 + ge));
    for (int eb = eb0; eb < e_end; eb += PASS) {
        // rotate the issue priority among the four waves of a SIMD every four passes: they would otherwise issue
        // oldest-first, the low wave ids would finish their share early and the CU would drain half empty (-1.5 %)
        switch ((((eb - eb0) / PASS) / 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 8)) & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
        const int bk = nxt_bk;
        const int e_cur = GRP ? g0 : eb + ge;  // gather-map entry of this pass
        if constexpr (GRP) {
            nxt_bk = load_bucket(g1);
            g0 = g1, g1 = ent(eb + 2 * PASS + ge);
        } else {
            nxt_bk = load_bucket(ent(eb + PASS + ge));
        }
        int reg[NROUND][4];
#pragma unroll
        for (int q = 0; q < NROUND; q++)
#pragma unroll
            for (int j = 0; j < 4; j++) reg[q][j] = nxt[q][j];
        if constexpr (GRP) {
            load_idx(d1, nxt);
            d1 = d2, d2 = ent(eb + 3 * PASS + eslot);
        } else if constexpr (PRE) {
            load_pre(eb + PASS + eslot, nxt);
        } else {
            load_idx(ent(eb + PASS + eslot), nxt);
        }
        // decode: slot p is masked iff any slot <= p of the same entry is invalid
        bool dead = eb + eslot >= e_end;
        int nblk = 0;  // wave-uniform: 4-slot blocks that hold a live slot for at least one entry
        if constexpr (PRE) {
            lds_u4_t o;
            o.x = (unsigned)reg[0][0], o.y = (unsigned)reg[0][1], o.z = (unsigned)reg[0][2], o.w = (unsigned)reg[0][3];
            *reinterpret_cast<lds_u4_t *>(dstrip + sub * 4) = o;
            nblk = __builtin_amdgcn_readfirstlane(nxt_nblk);
            nxt_nblk = load_nblk(eb + PASS);
        }
#pragma unroll
        for (int q = 0; q < (PRE ? 0 : NROUND); q++) {
            bool inv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // ALIGNED4 loads are clamped per block of 4, so a clamped block repeats data of slots < nnz
                const int p = q * 32 + sub * 4 + j;
                inv[j] = (p >= d.nnz) || reg[q][j] < 0 || reg[q][j] >= d.cols;
            }
            const bool any = inv[0] | inv[1] | inv[2] | inv[3];
            const unsigned m = (unsigned)(__ballot(any) >> (eslot * 8)) & 0xFFu;
            bool kill = dead || (m & ((1u << sub) - 1u));
            const unsigned long long open = __ballot(!(kill || inv[0]));  // blocks whose first slot is live
            lds_u4_t o;
            // live slot -> its tile row; GRP: outside the segment -> the NaN row (and the violation counter)
            auto slot_row = [&](int j) -> unsigned {
                if constexpr (GRP) {
                    const int loc = j - col0;
                    if (loc < 0 || loc >= tc) {
                        atomicAdd(gp.viol, 1);
                        return (unsigned)(tc + 1);
                    }
                    return (unsigned)loc;
                } else {
                    return (unsigned)j;
                }
            };
            kill |= inv[0], o.x = (kill ? (unsigned)tc : slot_row(reg[q][0])) * (LDS_RT * 4);
            kill |= inv[1], o.y = (kill ? (unsigned)tc : slot_row(reg[q][1])) * (LDS_RT * 4);
            kill |= inv[2], o.z = (kill ? (unsigned)tc : slot_row(reg[q][2])) * (LDS_RT * 4);
            kill |= inv[3], o.w = (kill ? (unsigned)tc : slot_row(reg[q][3])) * (LDS_RT * 4);
            *reinterpret_cast<lds_u4_t *>(dstrip + q * 32 + sub * 4) = o;
            dead |= (m != 0);
            unsigned fold = (unsigned)open | (unsigned)(open >> 32);
            fold |= fold >> 16;
            fold |= fold >> 8;
            nblk += __popc(fold & 0xFFu);  // live blocks form a prefix in every entry, so the union is a prefix
        }
        __builtin_amdgcn_wave_barrier();  // the strips are private to this wave; LDS executes a wave's ops in order

        // 8 features per trip (an odd tail block reads masked slots = the zero row)
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const char *wlb = reinterpret_cast<const char *>(wl) + gs * 16;
        const int ntrip = (nblk + 1) >> 1;
        if (ntrip > 0) {
            lds_u4_t ia = *reinterpret_cast<const lds_u4_t *>(strip), ib = *reinterpret_cast<const lds_u4_t *>(strip + 4);
            for (int t = 0; t < ntrip; t++) {
                // volatile + LDS address space: the next trip's offsets are fetched in THIS iteration, ahead of
                // the gathers, so a trip pays one LDS latency instead of two
                typedef const volatile __attribute__((address_space(3))) lds_u4_t *vstrip_t;
                const int tn = min(t + 1, ntrip - 1) * 8;
                const lds_u4_t na = *(vstrip_t)(strip + tn), nb = *(vstrip_t)(strip + tn + 4);
                const float4 v0 = *reinterpret_cast<const float4 *>(wlb + ia.x);
                const float4 v1 = *reinterpret_cast<const float4 *>(wlb + ia.y);
                const float4 v2 = *reinterpret_cast<const float4 *>(wlb + ia.z);
                const float4 v3 = *reinterpret_cast<const float4 *>(wlb + ia.w);
                const float4 v4 = *reinterpret_cast<const float4 *>(wlb + ib.x);
                const float4 v5 = *reinterpret_cast<const float4 *>(wlb + ib.y);
                const float4 v6 = *reinterpret_cast<const float4 *>(wlb + ib.z);
                const float4 v7 = *reinterpret_cast<const float4 *>(wlb + ib.w);
                add4(acc, v0), add4(acc, v1), add4(acc, v2), add4(acc, v3);
                add4(acc, v4), add4(acc, v5), add4(acc, v6), add4(acc, v7);
                ia = na, ib = nb;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int e = e_cur;
        const bool live = eb + ge < e_end;
        const int k = (live && e >= d.batch) ? 1 : 0;
        const int b = live ? e - k * d.batch : 0;
        float4 c = make_float4(bt_act_fwd(d.act, acc.x + bv.x), bt_act_fwd(d.act, acc.y + bv.y),
                               bt_act_fwd(d.act, acc.z + bv.z), bt_act_fwd(d.act, acc.w + bv.w));
        if constexpr (GRP) {  // a sum that met the NaN row stays NaN through the clamps of the activation
            if (acc.x != acc.x) c = make_float4(acc.x, acc.x, acc.x, acc.x);
        }
        if constexpr (PAIR) {
            // the lane holding part gs^4 of the same entry has the other factors of the same output columns
            const float4 o = make_float4(fwd_partner(c.x, lane), fwd_partner(c.y, lane), fwd_partner(c.z, lane),
                                         fwd_partner(c.w, lane));
            const float4 pr = make_float4(c.x * o.x, c.y * o.y, c.z * o.z, c.w * o.w);
            if (live) {
                *reinterpret_cast<float4 *>(saved + (size_t)e * d.rows + row) = c;
                if (gs < 4) *reinterpret_cast<float4 *>(out + (size_t)b * d.out_ld + (size_t)k * out_w + rt * 16 + gs * 4) = pr;
            }
            if constexpr (HEAD) {  // the 16 products live on the quad gs < 4
                const float4 wv = *reinterpret_cast<const float4 *>(w1t + (k * hs.w1_rows + bk) * TW + (gs & 3) * 4);
                float p = 0.0f;
                p += pr.x * wv.x, p += pr.y * wv.y, p += pr.z * wv.z, p += pr.w * wv.w;
                p = quad_sum(p);
                if (live && gs == 0) hs.zpart[(size_t)(k * ntiles + rt) * d.batch + b] = p;
            }
        } else {
            if (live) *reinterpret_cast<float4 *>(out + (size_t)b * d.out_ld + (size_t)k * out_w + row) = c;
            if constexpr (HEAD) {
                const float4 wv = *reinterpret_cast<const float4 *>(w1t + (k * hs.w1_rows + bk) * TW + gs * 4);
                float p = 0.0f;
                p += c.x * wv.x, p += c.y * wv.y, p += c.z * wv.z, p += c.w * wv.w;
                p = quad_sum(p);  // columns 0-15 on the quad gs < 4, 16-31 on the other
                const float hi = fwd_partner(p, lane);
                if (live && gs == 0) hs.zpart[(size_t)(k * ntiles + rt) * d.batch + b] = p + hi;
            }
        }
    }
}

// backward: workgroup (row tile rt, chunk group cg) -> partial_w[cg][cols][stride], partial_b[cg][rows]
// FUSED: the incoming gradient is not read from memory but formed on the fly from a one-output dense head,
// grad_out[b][k*out_w + c] = gz[b] * W1[bucket[b]][k*out_w + c]  (Matmul::backward dX = W^T dY with a single
// selected row, matmul.rs:117-123 + select.rs:93-110), so the [B][2H] gradient buffer never exists.
struct FtFusedSrc {
    const float *gz;        // [B] gradient w.r.t. the head's pre-activation
    const float *w1;        // head weights, column-major [w1_rows][n_inputs*out_w]
    const int32_t *bucket;  // [B] or null
    int w1_rows;            // = number of output buckets (one output each)
};

typedef float oct_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Developer build only (-DBT_STAMPS, tools/stamps_bwd.sh): shader-clock stamps at the phase boundaries of the octet
// backward's chunk loop, one per wave of the first BT_STAMP_WGS workgroups, into a buffer nothing else reads.
#ifdef BT_STAMPS
constexpr int BT_STAMP_WGS = 8, BT_STAMP_CHUNKS = 20, BT_STAMP_N = 8;
__device__ unsigned long long bt_stamps[BT_STAMP_WGS][16][BT_STAMP_CHUNKS][BT_STAMP_N];
#define BT_STAMPC(i, ch)                                                                                  \
    do {                                                                                                  \
        if (blockIdx.x < BT_STAMP_WGS && (threadIdx.x & 63) == 0 && (ch) - c_lo < BT_STAMP_CHUNKS)        \
            bt_stamps[blockIdx.x][wv][(ch) - c_lo][i] = __builtin_amdgcn_s_memtime();                     \
    } while (0)
#define BT_STAMP(i) BT_STAMPC(i, c)
#ifndef BT_STAMP7_POS
#define BT_STAMP7_POS 3
#endif
#define BT_STAMP7(pos, ch)      \
    do {                        \
        if (BT_STAMP7_POS == pos) BT_STAMPC(7, ch); \
    } while (0)
#else
#define BT_STAMP(i)
#define BT_STAMPC(i, ch)
#define BT_STAMP7(pos, ch)
#endif
struct FtHeadGradSink {
    float *pw1;         // [G][n_inputs * out_w * w1_rows] per-chunk-group partials of dW1 (column-major like w1)
    size_t pw1_stride;  // floats per chunk group
};

// The workgroup's 16 waves have two roles. OCT_WAVES of them REDUCE: each walks its stream of rows of the chunk's octet
// index, gathers rows of the staged gradient tile and adds what it has summed to the dW tile; the LDS array bounds them.
// The other OCT_STAGERS (waves 0-3, one per SIMD, the oldest there and so first in the issue order) STAGE: they load the
// next chunk's activations, form g = dout * f'(out), the bias sums and the head's dW on the matrix cores, keep the
// result in registers and write it to the tile between the two barriers that separate chunks. So the global loads, the
// activation derivative and the MFMAs of chunk c + 1 run beside the gathers of chunk c, on waves that never gather, the
// tile needs no second buffer, and a reducing wave issues no vector-memory instruction but its own row fetches.
// (Tried: a stream of a third of a reducer's share for each staging wave, reduced between forming one chunk and requesting
// the next. The reduction's registers on top of the staging registers spill, and the kernel went from 99 to 117 us.)
template <bool PAIR, bool FUSED, bool GRP, bool HDW>
__device__ __forceinline__ void ft_bwd_oct_body(const float *__restrict__ grad_out, const FtFusedSrc &fs,
                                                const float *__restrict__ out, const float *__restrict__ saved,
                                                const uint2 *__restrict__ stream, const uint8_t *__restrict__ meta,
                                                float *__restrict__ partial_w, float *__restrict__ partial_b,
                                                const bt_ft_desc &d, int ntiles, int chunks_per_input, int chunks_per_wg,
                                                const FtGrp &gp, int rows_cap, const FtHeadGradSink &hg) {
    static_assert(!HDW || (FUSED && !PAIR && !GRP), "HDW: fused single-output head over an ungrouped, unpaired table");
    static_assert((OCT_WAVES + OCT_STAGERS) * 64 == LDS_THREADS && FT_CHUNK % (OCT_STAGERS * 8) == 0, "wave roles");
    extern __shared__ __align__(16) float lds_f[];
    const int tc = GRP ? gp.tile_cols : d.cols;
    // (the gradient tile sits at LDS offset 0: a gather's address is then (row offset) | (lane's 16 bytes), one v_and_or)
    float *gt = lds_f;                                  // [FT_CHUNK + 1][32], last row = zeros (padding target)
    float *acc = gt + (FT_CHUNK + 1) * LDS_RT;          // [tc][32]
    float *parts = acc + (size_t)tc * LDS_RT;           // [OCT_SLOTS][32] sums of the parts of cut runs
    float *w1t = parts + OCT_SLOTS * LDS_RT;            // FUSED: [n_inputs][w1_rows][TW]
    const int rt = blockIdx.x % ntiles, cg = blockIdx.x / ntiles;
    const int half_rows = d.rows / 2;
    const int nchunks = d.n_inputs * chunks_per_input;
    int g_clo = 0, g_chi = 0;
    if constexpr (GRP) {
        if (cg >= gp.counts[1]) return;
        const int4 it = gp.items[cg];
        g_clo = it.y, g_chi = it.y + it.z;
    }
    const int out_w = PAIR ? half_rows : d.rows;
    constexpr int TW = PAIR ? 16 : 32;

    if constexpr (FUSED) {
        for (int q = threadIdx.x; q < d.n_inputs * fs.w1_rows * TW; q += LDS_THREADS) {
            const int c = q % TW, bk = (q / TW) % fs.w1_rows, k = q / (TW * fs.w1_rows);
            w1t[q] = fs.w1[(size_t)(k * out_w + rt * TW + c) * fs.w1_rows + bk];
        }
    }
    for (int q = threadIdx.x; q < tc * LDS_RT; q += LDS_THREADS) acc[q] = 0.0f;
    if (threadIdx.x < LDS_RT) gt[FT_CHUNK * LDS_RT + threadIdx.x] = 0.0f;
    if (lds_addr_of(gt) & 127u) __builtin_trap();  // (folds away: this kernel has no static LDS in front of the tile)

    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool stager = wv < OCT_STAGERS;
    const int c_lo = GRP ? g_clo : cg * chunks_per_wg, c_hi = GRP ? g_chi : min(nchunks, c_lo + chunks_per_wg);
    // what the roles hand to the common epilogue
    float4 bsum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // stagers: bias sums of this thread's tile columns
    oct_f32x4 hd[2];                                     // stagers, HDW: dW1 accumulators [bucket rows 4 * (lane / 16) ..][column slot lane % 16]
#pragma unroll
    for (int i = 0; i < 2; i++) hd[i] = oct_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    uint2 last_fix = make_uint2(0u, 0u);                 // reducers: the cut features of the last chunk
    uint32_t last_nfix = 0;
    int rg_e, rg_s;                                      // reduce role: group rg_e of the wave, float4 rg_s of the row
    fwd_gather_slot(threadIdx.x & 63, rg_e, rg_s);
    const int grp = (wv - OCT_STAGERS) * 8 + rg_e;       // (reducers)
    // parts of the cut runs of the chunk just reduced -> dW tile, in part order (one group per cut feature)
    auto fixup = [&](const uint2 fx, uint32_t nfix) {
        if ((uint32_t)grp < nfix) {
            const uint32_t f = fx.x & 0xffffu, s0 = fx.x >> 16, np = fx.y;
            float4 s = *reinterpret_cast<const float4 *>(parts + s0 * LDS_RT + rg_s * 4);
            for (uint32_t j = 1; j < np; j++) add4(s, *reinterpret_cast<const float4 *>(parts + (s0 + j) * LDS_RT + rg_s * 4));
            float4 *dst = reinterpret_cast<float4 *>(acc + f * LDS_RT + rg_s * 4);
            float4 cur = *dst;
            add4(cur, s);
            *dst = cur;
        }
    };

    // ---- staging, for any number of passes NP over entries tg, tg + stride, ...: 8 lanes x float4 per entry
    const int s8 = threadIdx.x & 7;
    const uint32_t t_col = PAIR ? rt * 16 + (s8 & 3) * 4 : rt * LDS_RT + s8 * 4;
    // PAIR: the lane holds its own half's activations; the other half's come from lane ^ 4 when the product is formed
    const uint32_t sv_col = rt * 16 + (s8 & 3) * 4 + ((s8 & 4) ? half_rows : 0);
    auto load_eids = [&](auto np, int stride, int tg, int *r_eid, int c) {
        if constexpr (GRP) {
            const int4 ch = gp.chunks[min(c, c_hi - 1)];
#pragma unroll
            for (int pass = 0; pass < decltype(np)::value; pass++)
                r_eid[pass] = gp.elist[ch.y + min(tg + pass * stride, ch.z - 1)];
        }
    };
    // All loads of a chunk are unconditional (rows past its end re-read its last entry and count as zero when the
    // gradient is formed): their number is then known to the compiler, and the counted waits stay exact.
    auto request = [&](auto np, int stride, int tg, float4 *r_go, float4 *r_a, int *r_eid, int cc) {
        constexpr int NP = decltype(np)::value;
        if constexpr (GRP) {
#pragma unroll
            for (int pass = 0; pass < NP; pass++) {
                const int e = r_eid[pass];
                const int k = e >= d.batch ? 1 : 0, b = e - k * d.batch;
                const size_t o_out = (size_t)b * d.out_ld + (size_t)k * out_w + t_col;
                r_go[pass] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if constexpr (FUSED) {
                    r_go[pass].x = fs.gz[b];
                    r_go[pass].y = __int_as_float(fs.bucket ? fs.bucket[b] : 0);
                    r_go[pass].z = __int_as_float(k);
                } else {
                    r_go[pass] = *reinterpret_cast<const float4 *>(grad_out + o_out);
                }
                if constexpr (PAIR) r_a[pass] = *reinterpret_cast<const float4 *>(saved + (size_t)e * d.rows + sv_col);
                else r_a[pass] = *reinterpret_cast<const float4 *>(out + o_out);
            }
            load_eids(np, stride, tg, r_eid, cc + 1);  // the entry ids of the chunk after this one, for the next request
        } else {
            // wave-uniform chunk base (SGPRs) + 32-bit per-lane element offsets
            const int kc = cc / chunks_per_input, b0 = (cc % chunks_per_input) * FT_CHUNK;
            const int n_e = min(FT_CHUNK, d.batch - b0);
            const float *go_base = grad_out + (size_t)b0 * d.out_ld + (size_t)kc * out_w;
            const float *out_base = out + (size_t)b0 * d.out_ld + (size_t)kc * out_w;
            const float *sav_base = saved + ((size_t)kc * d.batch + b0) * d.rows;
#pragma unroll
            for (int pass = 0; pass < NP; pass++) {
                const uint32_t el = (uint32_t)min(tg + pass * stride, n_e - 1);
                const uint32_t o_out = el * (uint32_t)d.out_ld + t_col;
                r_go[pass] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if constexpr (FUSED) {
                    r_go[pass].x = fs.gz[b0 + el];
                    r_go[pass].y = __int_as_float(fs.bucket ? fs.bucket[b0 + el] : 0);
                } else {
                    r_go[pass] = *reinterpret_cast<const float4 *>(go_base + o_out);
                }
                if constexpr (PAIR) r_a[pass] = *reinterpret_cast<const float4 *>(sav_base + el * (uint32_t)d.rows + sv_col);
                else r_a[pass] = *reinterpret_cast<const float4 *>(out_base + o_out);
            }
        }
    };
    auto xor4 = [](float v) {  // the value of lane ^ 4 (two DPP row rotations, each writing its own banks)
        int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x12C, 0xF, 0x5, false);  // lanes 0-3, 8-11: from lane + 4
        t = __builtin_amdgcn_update_dpp(t, __float_as_int(v), 0x124, 0xF, 0xA, false);      // lanes 4-7, 12-15: from lane - 4
        return __int_as_float(t);
    };
    // the activations of chunk cc (requested earlier) -> its staged gradient g_out[pass], the bias sums, the head's dW
    auto form = [&](auto np, int stride, int tg, const float4 *r_go, const float4 *r_a, float4 *g_out, int cc) {
        constexpr int NP = decltype(np)::value;
        int n_e, kc = 0;
        if constexpr (GRP) n_e = gp.chunks[cc].z;
        else kc = cc / chunks_per_input, n_e = min(FT_CHUNK, d.batch - (cc % chunks_per_input) * FT_CHUNK);
#pragma unroll
        for (int pass = 0; pass < NP; pass++) {
            const int el = tg + pass * stride;
            if constexpr (HDW) {
                // (whole wave: MFMA runs on all lanes; entries past the chunk's end count with gz = 0)
                // A 16-lane row holds the entries 2k (lanes 0-7) and 2k + 1 (lanes 8-15), 8 lanes x float4 each. The
                // B operand wants one entry per row (k = lane / 16), so the two halves trade half their float4
                // (one DPP rotation by 8 lanes per value): afterwards the row holds entry 2k as 16 x float2 in
                // (e0, e1) and entry 2k + 1 in (o0, o1), column 4 * (n & 7) + 2 * (n >> 3) + component for lane n.
                const float gzv = el < n_e ? r_go[pass].x : 0.0f;
                const int bk = __float_as_int(r_go[pass].y);
                const float4 o = r_a[pass];
                const float4 pv = make_float4(gzv * o.x, gzv * o.y, gzv * o.z, gzv * o.w);
                const bool hi = (threadIdx.x >> 3) & 1;
                const float x0 = row_ror8(hi ? pv.x : pv.z), x1 = row_ror8(hi ? pv.y : pv.w);  // from lane ^ 8
                const float e0 = hi ? x0 : pv.x, e1 = hi ? x1 : pv.y;
                const float o0 = hi ? pv.z : x0, o1 = hi ? pv.w : x1;
                const int m = threadIdx.x & 15;
                const float a_e = row_share<0>(bk) == m ? 1.0f : 0.0f, a_o = row_share<8>(bk) == m ? 1.0f : 0.0f;
                hd[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_e, e0, hd[0], 0, 0, 0);
                hd[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_e, e1, hd[1], 0, 0, 0);
                hd[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_o, o0, hd[0], 0, 0, 0);
                hd[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_o, o1, hd[1], 0, 0, 0);
            }
            float4 oth = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if constexpr (PAIR)  // (whole wave: DPP reads the partner lane's registers)
                oth = make_float4(xor4(r_a[pass].x), xor4(r_a[pass].y), xor4(r_a[pass].z), xor4(r_a[pass].w));
            float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (el < n_e) {
                float4 go = r_go[pass];
                if constexpr (FUSED) {
                    const float gzv = go.x;
                    const int bk = __float_as_int(go.y);
                    const int ke = GRP ? __float_as_int(go.z) : kc;
                    const float4 wv4 = *reinterpret_cast<const float4 *>(
                        w1t + (ke * fs.w1_rows + bk) * TW + (PAIR ? (s8 & 3) : s8) * 4);
                    go = make_float4(gzv * wv4.x, gzv * wv4.y, gzv * wv4.z, gzv * wv4.w);
                }
                const float4 dv = bt_act_deriv4(d.act, r_a[pass]);
                if constexpr (PAIR)
                    g = make_float4((go.x * oth.x) * dv.x, (go.y * oth.y) * dv.y, (go.z * oth.z) * dv.z,
                                    (go.w * oth.w) * dv.w);
                else
                    g = make_float4(go.x * dv.x, go.y * dv.y, go.z * dv.z, go.w * dv.w);
                bsum.x += g.x, bsum.y += g.y, bsum.z += g.z, bsum.w += g.w;
            }
            g_out[pass] = g;
        }
    };
    // The first chunk is staged by all 16 waves, straight into the tile (nobody reads it yet): a workgroup of a
    // segment-grouped table has only a handful of chunks, and four waves would take a chunk's time over it.
    if (c_lo < c_hi) {
        constexpr int FP = FT_CHUNK / (LDS_THREADS / 8);
        const std::integral_constant<int, FP> fp;
        const int tg = threadIdx.x >> 3;
        float4 f_go[FP], f_a[FP], f_g[FP];
        int f_eid[FP];
        load_eids(fp, LDS_THREADS / 8, tg, f_eid, c_lo);
        request(fp, LDS_THREADS / 8, tg, f_go, f_a, f_eid, c_lo);
        __syncthreads();  // (the head rows in LDS are complete; c_lo and c_hi are the same for the whole workgroup)
        form(fp, LDS_THREADS / 8, tg, f_go, f_a, f_g, c_lo);
#pragma unroll
        for (int pass = 0; pass < FP; pass++)
            *reinterpret_cast<float4 *>(gt + (tg + pass * (LDS_THREADS / 8)) * LDS_RT + s8 * 4) = f_g[pass];
    }

    if (stager) {
        // ---------------------------------------------------------------- stage: 256 threads, 32 entries per pass
        // A chunk's activations are requested a whole chunk interval before they are formed (the loads of chunk c + 2 go
        // out when chunk c + 1 has been formed), so a staging wave never waits for HBM between the two barriers.
        constexpr int SP = FT_CHUNK / (OCT_STAGERS * 8);
        constexpr int SS = OCT_STAGERS * 8;
        const std::integral_constant<int, SP> sp_c;
        const int el0 = threadIdx.x >> 3;
        float4 g_reg[SP], r_go[SP], r_a[SP];
        int r_eid[SP];
        if (c_lo < c_hi) {
            load_eids(sp_c, SS, el0, r_eid, c_lo + 1);
            request(sp_c, SS, el0, r_go, r_a, r_eid, min(c_lo + 1, c_hi - 1));
        }
        for (int c = c_lo; c < c_hi; c++) {
            BT_STAMP(0);
            lds_barrier();  // the previous chunk's readers are done with the tile
            BT_STAMP(1);
            if (c > c_lo) {
#pragma unroll
                for (int pass = 0; pass < SP; pass++)
                    *reinterpret_cast<float4 *>(gt + (el0 + pass * SS) * LDS_RT + s8 * 4) = g_reg[pass];
            }
            BT_STAMP(2);
            lds_barrier();
            BT_STAMP(3);
            if (c + 1 < c_hi) form(sp_c, SS, el0, r_go, r_a, g_reg, c + 1);
            BT_STAMP(4);
            // (the last chunks ask for the last one again: unconditional on purpose, see above)
            request(sp_c, SS, el0, r_go, r_a, r_eid, min(c + 2, c_hi - 1));
            BT_STAMP(5);
        }
    } else {
        // ---------------------------------------------------------------- reduce
        const int rw = wv - OCT_STAGERS;
        uint32_t m_r0 = 0, m_r1 = 0, m_nfix = 0;  // meta of the next chunk: this wave's row range, the cut features
        uint2 m_fix = make_uint2(0u, 0u);
        auto load_meta = [&](int c) {
            const uint16_t *hdr = reinterpret_cast<const uint16_t *>(meta + (size_t)c * OCT_META_BYTES);
            m_r0 = hdr[rw], m_r1 = hdr[rw + 1], m_nfix = hdr[17];
            m_fix = reinterpret_cast<const uint2 *>(meta + (size_t)c * OCT_META_BYTES + 64)[min(grp, OCT_FIX - 1)];
        };
        // this wave's stream of a chunk: rows [r0, r0 + n), n a multiple of 4; the first two 4-row sets
        uint2 qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;
        qa0 = qa1 = qa2 = qa3 = qb0 = qb1 = qb2 = qb3 = make_uint2(0u, 0u);
        auto first_rows = [&](int c, int r0, int n) {
            const uint2 *sp = stream + ((size_t)c * rows_cap + r0) * 8 + rg_e;
            const int t1 = min(4, max(n - 4, 0));
            qa0 = sp[0], qa1 = sp[8], qa2 = sp[16], qa3 = sp[24];
            qb0 = sp[(t1 + 0) * 8], qb1 = sp[(t1 + 1) * 8], qb2 = sp[(t1 + 2) * 8], qb3 = sp[(t1 + 3) * 8];
        };
        int n_r0 = 0, n_n = 0;  // row range of the chunk about to be reduced
        if (c_lo < c_hi) {
            load_meta(c_lo);
            n_r0 = __builtin_amdgcn_readfirstlane((int)m_r0), n_n = __builtin_amdgcn_readfirstlane((int)m_r1) - n_r0;
            first_rows(c_lo, n_r0, n_n);
        }
        for (int c = c_lo; c < c_hi; c++) {
            const int r0 = n_r0, n = n_n;
            const uint2 cur_fix = m_fix;
            const uint32_t cur_nfix = __builtin_amdgcn_readfirstlane((int)m_nfix);
            BT_STAMP(0);
            lds_barrier();  // every wave is done with the previous chunk: its parts are complete
            BT_STAMP(1);
            fixup(last_fix, last_nfix);
            last_fix = cur_fix, last_nfix = cur_nfix;
            load_meta(min(c + 1, c_hi - 1));
            BT_STAMP(2);
            lds_barrier();  // the tile holds this chunk's gradient
            BT_STAMP(3);

            const uint2 *sp = stream + ((size_t)c * rows_cap + r0) * 8 + rg_e;
            float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const uint32_t lane16 = lds_addr_of(gt) + (uint32_t)rg_s * 16;  // (the tile's LDS address is 0, checked above)
            // Two rows in flight: the gathers of row r + 1 are issued before the sums of row r are taken, so a wave keeps
            // eight ds_read_b128 outstanding and its adds and address arithmetic run under the LDS latency. (Reading the dW
            // row of a flush in front of the flagged row's own gathers, so that the flush never drains the wave's LDS
            // queue, was measured and is slower: +8 us; LDS float atomics for the flush: five times slower.)
            float4 va0, va1, va2, va3, vb0, vb1, vb2, vb3;
#define ISSUE(V, Q)                                          \
    V##0 = lds_load4(((Q).x & 0xff80u) | lane16),            \
    V##1 = lds_load4(((Q).x >> 16) + lane16),                \
    V##2 = lds_load4(((Q).y & 0xff80u) | lane16),            \
    V##3 = lds_load4(((Q).y >> 16) + lane16)
            auto flush = [&](const uint2 q) {  // last row of an octet: the whole wave ends its runs here
                const uint32_t kind = (q.x >> 1) & 3u, id = ((q.x >> 3) & 15u) | ((q.y & 127u) << 4);
                if (kind == 1u) {
                    float4 *dst = reinterpret_cast<float4 *>(acc + id * LDS_RT + rg_s * 4);
                    float4 cur = *dst;
                    add4(cur, a);
                    *dst = cur;
                } else if (kind == 2u) {
                    *reinterpret_cast<float4 *>(parts + id * LDS_RT + rg_s * 4) = a;
                }
                a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            };
#define CONSUME(V, Q)                                                            \
    do {                                                                         \
        add4(a, V##0), add4(a, V##1), add4(a, V##2), add4(a, V##3);              \
        if (__builtin_amdgcn_readfirstlane((int)(Q).x) & 1) flush(Q);            \
    } while (0)
            // a set of four rows whose first row's gathers are in flight (in va); NEXT: the first row of the set behind it
#define SET4(Q0, Q1, Q2, Q3, HAS_NEXT, NEXT) \
    do {                                     \
        ISSUE(vb, Q1);                       \
        CONSUME(va, Q0);                     \
        ISSUE(va, Q2);                       \
        CONSUME(vb, Q1);                     \
        ISSUE(vb, Q3);                       \
        CONSUME(va, Q2);                     \
        if (HAS_NEXT) ISSUE(va, NEXT);       \
        CONSUME(vb, Q3);                     \
    } while (0)
            // rows t .. t+3 of the stream (clamped to its last set: a short stream re-reads rows it never uses)
            auto fetch = [&](int t, uint2 &x0, uint2 &x1, uint2 &x2, uint2 &x3) {
                const int tt = min(t, max(n - 4, 0));
                x0 = sp[(tt + 0) * 8], x1 = sp[(tt + 1) * 8], x2 = sp[(tt + 2) * 8], x3 = sp[(tt + 3) * 8];
            };
            // sets 0 and 1 (qa, qb) were fetched at the end of the chunk before; every later set one set ahead
#ifndef BT_OCT_PRIO
#define BT_OCT_PRIO 1
#endif
#if BT_OCT_PRIO > 0
            // The waves of a SIMD issue oldest-first: with equal work the older reducers run ahead and the youngest
            // finishes the chunk alone. Rotating the issue priority among the three every BT_OCT_PRIO sets keeps them level.
            const int wq3 = rw >> 2;  // position among the three reducing waves of this SIMD
            auto prio = [&](int k) {
                switch ((k / BT_OCT_PRIO + wq3) % 3) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(2); break;
                }
            };
#else
            auto prio = [&](int) {};
#endif
            if (0 < n) {
                ISSUE(va, qa0);
                for (int t = 0;; t += 8) {
                    const bool has_b = t + 4 < n;
                    prio(t >> 2);
                    SET4(qa0, qa1, qa2, qa3, has_b, qb0);
                    fetch(t + 8, qa0, qa1, qa2, qa3);
                    if (!has_b) break;
                    const bool has_a = t + 8 < n;
                    prio((t >> 2) + 1);
                    SET4(qb0, qb1, qb2, qb3, has_a, qa0);
                    fetch(t + 12, qb0, qb1, qb2, qb3);
                    if (!has_a) break;
                }
            }
#undef SET4
#undef CONSUME
#undef ISSUE
            BT_STAMP(5);
            // the next chunk's first rows (the last chunk fetches its own again: every load is unconditional)
            n_r0 = __builtin_amdgcn_readfirstlane((int)m_r0), n_n = __builtin_amdgcn_readfirstlane((int)m_r1) - n_r0;
            first_rows(min(c + 1, c_hi - 1), n_r0, n_n);
            BT_STAMP(6);
        }
    }
    __syncthreads();
    if (!stager) fixup(last_fix, last_nfix);
    __syncthreads();
    float *pw = partial_w + (size_t)cg * tc * d.stride;
    for (int q = threadIdx.x; q < tc * 8; q += LDS_THREADS) {
        const int f = q >> 3, t8 = q & 7;
        *reinterpret_cast<float4 *>(pw + (size_t)f * d.stride + d.offset + tile_row(PAIR, rt, half_rows, t8 * 4)) =
            *reinterpret_cast<const float4 *>(acc + f * LDS_RT + t8 * 4);
    }
    // bias partial: fixed-order sum over the 128 thread groups (8 lanes x float4 each; the reducers' hold the first
    // chunk's share only), in two levels (8 x 16)
    *reinterpret_cast<float4 *>(gt + (threadIdx.x >> 3) * LDS_RT + s8 * 4) = bsum;
    __syncthreads();
    float bpart = 0.0f;
    if (threadIdx.x < 8 * LDS_RT) {
        const int col = threadIdx.x & (LDS_RT - 1), part = threadIdx.x / LDS_RT;
        for (int i = 0; i < 16; i++) bpart += gt[(part * 16 + i) * LDS_RT + col];
    }
    __syncthreads();
    if (threadIdx.x < 8 * LDS_RT) gt[threadIdx.x] = bpart;
    __syncthreads();
    if (threadIdx.x < LDS_RT) {
        float sb = 0.0f;
        for (int i = 0; i < 8; i++) sb += gt[i * LDS_RT + threadIdx.x];
        partial_b[(size_t)cg * d.rows + tile_row(PAIR, rt, half_rows, threadIdx.x)] = sb;
    }
    if constexpr (HDW) {
        // the waves' dW1 accumulators -> one partial per (bucket, column) of this tile, the waves in order.
        // The dW tile has been written out (barriers above), so its LDS holds the 16 x w1_rows x 32 floats.
        float *tmp = acc;
        const int lane = threadIdx.x & 63, qd = lane >> 4, j = lane & 15;
        __syncthreads();  // (every thread has read its part of the dW tile)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = 4 * qd + r;
            if (m < fs.w1_rows) {
                float *t2 = tmp + ((size_t)wv * fs.w1_rows + m) * 32 + 4 * (j & 7) + 2 * (j >> 3);
                t2[0] = hd[0][r], t2[1] = hd[1][r];
            }
        }
        __syncthreads();
        const int kk = c_lo / chunks_per_input;  // the one input all chunks of this workgroup belong to (host-checked)
        for (int q = threadIdx.x; q < fs.w1_rows * 32; q += LDS_THREADS) {
            const int m = q >> 5, col = q & 31;
            float sh = 0.0f;
            for (int w2 = 0; w2 < LDS_THREADS / 64; w2++) sh += tmp[((size_t)w2 * fs.w1_rows + m) * 32 + col];
            hg.pw1[(size_t)cg * hg.pw1_stride + (size_t)(kk * out_w + rt * 32 + col) * fs.w1_rows + m] = sh;
        }
    }
}

template <bool PAIR, bool FUSED, bool GRP, bool HDW>
__global__ __launch_bounds__(LDS_THREADS) void ft_bwd_oct(const float *__restrict__ grad_out, FtFusedSrc fs,
                                                          const float *__restrict__ out, const float *__restrict__ saved,
                                                          const uint2 *__restrict__ stream,
                                                          const uint8_t *__restrict__ meta, float *__restrict__ partial_w,
                                                          float *__restrict__ partial_b, bt_ft_desc d, int ntiles,
                                                          int chunks_per_input, int chunks_per_wg, FtGrp gp, int rows_cap,
                                                          FtHeadGradSink hg) {
    ft_bwd_oct_body<PAIR, FUSED, GRP, HDW>(grad_out, fs, out, saved, stream, meta, partial_w, partial_b, d, ntiles,
                                           chunks_per_input, chunks_per_wg, gp, rows_cap, hg);
}

// Merge of the fused-head backward: the dW tile and bias as ft_bwd_merge, plus the head's gradients that used to
// come from a kernel of their own: dW1 = sum of the chunk groups' partials (the groups of input k, in order), the
// head bias gradient db1[m] = sum over positions with bucket m of gz and the loss sum, both from the per-workgroup
// partials the head's finishing kernel left, in order.
struct FtHeadMerge {
    const float *pw1;
    size_t pw1_stride;
    float *dw1, *db1;
    const float *gz_parts;  // [n_loss_partials][w1_rows] sums of gz by bucket, one row per workgroup of the head finish
    const float *loss_partials;
    int n_loss_partials;
    float *loss_sum;
    int w1_rows, groups_per_input, n1;  // n1 = n_inputs * out_w * w1_rows floats of dW1
};
// STEP: AdamW on each element as soon as its gradient is complete (bt_ft_fused_adamw: tensor i's buffers are
// indexed like its gradient: dw, dbias, dw1, db1) -- the same bt_adam_op on the same values as adamw_kernel.
template <bool STEP>
__global__ __launch_bounds__(256) void ft_bwd_merge_head(const float *__restrict__ partial_w,
                                                         const float *__restrict__ partial_b, float *__restrict__ dw,
                                                         float *__restrict__ dbias, bt_ft_desc d, int G, FtHeadMerge hm,
                                                         int blocks_w, bt_ft_fused_adamw st) {
    const int rows4 = d.rows / 4;
    float adj = 0.0f, rate = 0.0f;
    if constexpr (STEP) adj = st.adj[0], rate = st.lr[0];
    auto update = [&](const bt_adamw_tensor &t, size_t i, float g) {
        float p = t.param[i], m = t.momentum[i], v = t.velocity[i];
        bt_adam_op(__fmul_rn(adj, g), rate, p, m, v, t);
        t.param[i] = p, t.momentum[i] = m, t.velocity[i] = v;
    };
    if ((int)blockIdx.x < blocks_w) {
        const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const long long nw = (long long)d.cols * rows4;
        if (t < nw) {
            const int f = (int)(t / rows4), r = (int)(t % rows4) * 4;
            const size_t o = (size_t)f * d.stride + d.offset + r;
            float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int g = 0; g < G; g++) {
                const float4 v = *reinterpret_cast<const float4 *>(partial_w + (size_t)g * d.cols * d.stride + o);
                s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
            }
            *reinterpret_cast<float4 *>(dw + o) = s;
            if constexpr (STEP) {
                const bt_adamw_tensor &t0 = st.t[0];
                float4 p = *reinterpret_cast<const float4 *>(t0.param + o);
                float4 m = *reinterpret_cast<const float4 *>(t0.momentum + o);
                float4 v = *reinterpret_cast<const float4 *>(t0.velocity + o);
                bt_adam_op(__fmul_rn(adj, s.x), rate, p.x, m.x, v.x, t0);
                bt_adam_op(__fmul_rn(adj, s.y), rate, p.y, m.y, v.y, t0);
                bt_adam_op(__fmul_rn(adj, s.z), rate, p.z, m.z, v.z, t0);
                bt_adam_op(__fmul_rn(adj, s.w), rate, p.w, m.w, v.w, t0);
                *reinterpret_cast<float4 *>(t0.param + o) = p;
                *reinterpret_cast<float4 *>(t0.momentum + o) = m;
                *reinterpret_cast<float4 *>(t0.velocity + o) = v;
            }
        } else if (dbias && t < nw + d.rows) {
            const int r = (int)(t - nw);
            float s = 0.0f;
            for (int g = 0; g < G; g++) s += partial_b[(size_t)g * d.rows + r];
            dbias[r] = s;
            if constexpr (STEP) update(st.t[1], (size_t)r, s);
        }
        return;
    }
    const int hb = blockIdx.x - blocks_w;  // head blocks: dW1 elements, then one block each for the loss and db1[m]
    const int blocks_1 = (hm.n1 + 255) / 256;
    if (hb < blocks_1) {
        const int e = hb * 256 + threadIdx.x;
        if (e < hm.n1) {
            const int k = e / (hm.n1 / d.n_inputs);
            float s = 0.0f;
            for (int g = k * hm.groups_per_input; g < (k + 1) * hm.groups_per_input; g++) s += hm.pw1[(size_t)g * hm.pw1_stride + e];
            hm.dw1[e] = s;
            if constexpr (STEP) update(st.t[2], (size_t)e, s);
        }
        return;
    }
    // last 1 + w1_rows blocks, one sum each: the loss (m = -1) and db1[m], from the finishing kernel's per-workgroup
    // partials -- thread t takes partials t, t + 256, ... in order, then a fixed tree over the threads
    __shared__ float red[256];
    const int m = hb - blocks_1 - 1;
    float s = 0.0f;
    for (int i = threadIdx.x; i < hm.n_loss_partials; i += 256)
        s += m < 0 ? hm.loss_partials[i] : hm.gz_parts[(size_t)i * hm.w1_rows + m];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        (m < 0 ? hm.loss_sum : hm.db1 + m)[0] = red[0];
        if constexpr (STEP)
            if (m >= 0) update(st.t[3], (size_t)m, red[0]);
    }
}

// segment-grouped merge: dw[seg*tc + f] = sum over the segment's backward items of their slabs (item order = list
// order, fixed); a segment without entries gets zeros; dbias = sum over all items
__global__ __launch_bounds__(256) void ft_bwd_merge_grp(const float *__restrict__ partial_w,
                                                        const float *__restrict__ partial_b, float *__restrict__ dw,
                                                        float *__restrict__ dbias, bt_ft_desc d, int tc,
                                                        const int *__restrict__ seg_items,
                                                        const int *__restrict__ counts) {
    const int rows4 = d.rows / 4;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nw = (long long)d.cols * rows4;
    if (t < nw) {
        const int f = (int)(t / rows4), r = (int)(t % rows4) * 4;
        const int sg = f / tc, fl = f - sg * tc;
        const size_t o = (size_t)fl * d.stride + d.offset + r;
        float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int g = seg_items[sg]; g < seg_items[sg + 1]; g++) {
            const float4 v = *reinterpret_cast<const float4 *>(partial_w + (size_t)g * tc * d.stride + o);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        *reinterpret_cast<float4 *>(dw + (size_t)f * d.stride + d.offset + r) = s;
    } else if (dbias && t < nw + d.rows) {
        const int r = (int)(t - nw), G = counts[1];
        float s = 0.0f;
        for (int g = 0; g < G; g++) s += partial_b[(size_t)g * d.rows + r];
        dbias[r] = s;
    }
}

// ft_bwd_merge / ft_bwd_merge_grp with the optimiser step inside (bt_ft_step_all): blocks [0, blocks_w) merge the table
// and bias gradients and apply AdamW to those elements at once; the blocks behind them are adamw_kernel's, for the
// other tensors of the net. Same bt_adam_op on the same values as the separate launch: bit-identical.
struct FtStepTab {
    bt_adamw_tensor t0, t1;
    const float *adj, *lr;
    int n_extra;
    int block_start[BT_ADAMW_MAX_TENSORS + 1];  // prefix sum of 1024-element blocks per extra tensor
    bt_adamw_tensor extra[BT_ADAMW_MAX_TENSORS];
};
template <bool GRP>
__global__ __launch_bounds__(256) void ft_bwd_merge_step(const float *__restrict__ partial_w,
                                                         const float *__restrict__ partial_b, float *__restrict__ dw,
                                                         float *__restrict__ dbias, bt_ft_desc d, int tc_or_g,
                                                         const int *__restrict__ seg_items,
                                                         const int *__restrict__ counts, int blocks_w, FtStepTab st) {
    const float adj = st.adj[0], rate = st.lr[0];
    if ((int)blockIdx.x >= blocks_w) {  // the other tensors: adamw_kernel's body
        const int xb = blockIdx.x - blocks_w;
        int ti = 0;
        while (ti + 1 < st.n_extra && xb >= st.block_start[ti + 1]) ti++;
        const bt_adamw_tensor &t = st.extra[ti];
        const size_t e0 = ((size_t)(xb - st.block_start[ti]) * 256 + threadIdx.x) * 4;
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
        return;
    }
    const int rows4 = d.rows / 4;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nw = (long long)d.cols * rows4;
    if (t < nw) {
        const int f = (int)(t / rows4), r = (int)(t % rows4) * 4;
        const size_t o = (size_t)f * d.stride + d.offset + r;  // (stride == rows, offset == 0: host-checked)
        float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (GRP) {
            const int tc = tc_or_g, sg = f / tc, fl = f - sg * tc;
            const size_t ol = (size_t)fl * d.stride + d.offset + r;
            for (int g = seg_items[sg]; g < seg_items[sg + 1]; g++) {
                const float4 v = *reinterpret_cast<const float4 *>(partial_w + (size_t)g * tc * d.stride + ol);
                s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
            }
        } else {
            for (int g = 0; g < tc_or_g; g++) {
                const float4 v = *reinterpret_cast<const float4 *>(partial_w + (size_t)g * d.cols * d.stride + o);
                s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
            }
        }
        *reinterpret_cast<float4 *>(dw + o) = s;
        const bt_adamw_tensor &t0 = st.t0;
        float4 p = *reinterpret_cast<const float4 *>(t0.param + o);
        float4 m = *reinterpret_cast<const float4 *>(t0.momentum + o);
        float4 v = *reinterpret_cast<const float4 *>(t0.velocity + o);
        bt_adam_op(__fmul_rn(adj, s.x), rate, p.x, m.x, v.x, t0);
        bt_adam_op(__fmul_rn(adj, s.y), rate, p.y, m.y, v.y, t0);
        bt_adam_op(__fmul_rn(adj, s.z), rate, p.z, m.z, v.z, t0);
        bt_adam_op(__fmul_rn(adj, s.w), rate, p.w, m.w, v.w, t0);
        *reinterpret_cast<float4 *>(t0.param + o) = p;
        *reinterpret_cast<float4 *>(t0.momentum + o) = m;
        *reinterpret_cast<float4 *>(t0.velocity + o) = v;
    } else if (t < nw + d.rows) {
        const int r = (int)(t - nw), G = GRP ? counts[1] : tc_or_g;
        float s = 0.0f;
        for (int g = 0; g < G; g++) s += partial_b[(size_t)g * d.rows + r];
        dbias[r] = s;
        const bt_adamw_tensor &t1 = st.t1;
        float p = t1.param[r], m = t1.momentum[r], v = t1.velocity[r];
        bt_adam_op(__fmul_rn(adj, s), rate, p, m, v, t1);
        t1.param[r] = p, t1.momentum[r] = m, t1.velocity[r] = v;
    }
}

// dw[f][offset + r] = sum_g partial_w[g][f][offset + r] (fixed order); dbias likewise
__global__ __launch_bounds__(256) void ft_bwd_merge(const float *__restrict__ partial_w,
                                                    const float *__restrict__ partial_b, float *__restrict__ dw,
                                                    float *__restrict__ dbias, bt_ft_desc d, int G) {
    const int rows4 = d.rows / 4;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nw = (long long)d.cols * rows4;
    if (t < nw) {
        const int f = (int)(t / rows4), r = (int)(t % rows4) * 4;
        const size_t o = (size_t)f * d.stride + d.offset + r;
        float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int g = 0; g < G; g++) {
            const float4 v = *reinterpret_cast<const float4 *>(partial_w + (size_t)g * d.cols * d.stride + o);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        *reinterpret_cast<float4 *>(dw + o) = s;
    } else if (dbias && t < nw + d.rows) {
        const int r = (int)(t - nw);
        float s = 0.0f;
        for (int g = 0; g < G; g++) s += partial_b[(size_t)g * d.rows + r];
        dbias[r] = s;
    }
}

struct FtPlan {
    bool lds_path;
    int ntiles, G, chunks_per_wg;
    size_t lds_fwd_bytes, off_pw, off_pb;
    static constexpr int kFusedMaxRows = 16;  // head rows (buckets) the fused backward keeps in LDS
    int vec;
    int chunks_per_input, nchunks;
    int index_waves;
    int fbits;
    size_t off_sorted, off_offs, off_g, off_bpart, total;
    int bias_blocks;
    // segment-grouped mode (bt_ft_desc.segment): tc = tile columns (= segment, else cols)
    bool grouped;
    int tc, nseg, grp_blocks, max_chunks, max_fitems, max_bitems, fwd_item, g_ch;
    size_t off_segof, off_blkcnt, off_elist, off_chunks, off_fitems, off_bitems, off_counts, off_segitems;
    // octet index of the LDS backward (ft_index_oct_kernel / ft_bwd_oct)
    bool oct;
    int rows_cap;
    size_t off_stream, off_meta, lds_oct_bytes;
    // head weight gradient formed inside the fused backward (ft_bwd_oct<HDW>): per-chunk-group partials of dW1
    bool hdw;
    size_t off_pw1, pw1_stride;
    // forward lists decoded once per batch (ft_fwd_decode_kernel, read by ft_fwd_lds<PRE>)
    bool predec;
    size_t off_foffs, off_fblk;
    int fwd_G, fwd_epw;  // grid of the ungrouped LDS forward: workgroups per row tile, entries per workgroup
};

int ft_validate(const bt_ft_desc *d, const char *who) {
    BT_REQUIRE(d, "%s: null descriptor", who);
    BT_REQUIRE(d->batch > 0 && d->rows > 0 && d->cols > 0 && d->nnz > 0, "%s: Mismatched sizes! (non-positive dim)", who);
    BT_REQUIRE(d->rows + d->offset <= d->stride, "%s: %d + %d > %d", who, d->rows, d->offset, d->stride);  // sparse.rs:30-32
    BT_REQUIRE(d->n_inputs >= 1 && d->n_inputs <= 2, "%s: n_inputs must be 1 or 2", who);
    BT_REQUIRE(d->nnz <= 64, "%s: nnz > 64 unsupported", who);
    BT_REQUIRE((long long)d->n_inputs * d->batch < (1ll << 30), "%s: batch too large", who);
    BT_REQUIRE(d->act >= BT_ACT_IDENTITY && d->act <= BT_ACT_SQRRELU, "%s: unknown activation", who);
    BT_REQUIRE(!d->pairwise || d->rows % 2 == 0, "%s: pairwise needs even rows", who);
    const int out_w = d->pairwise ? d->rows / 2 : d->rows;
    BT_REQUIRE(d->out_ld >= d->n_inputs * out_w, "%s: Mismatched sizes! (out_ld < n_inputs*out_w)", who);
    BT_REQUIRE(d->segment >= 0 && (d->segment == 0 || d->cols % d->segment == 0), "%s: segment must divide cols", who);
    return BT_OK;
}

int ft_vec(const bt_ft_desc *d) {
    const int out_w = d->pairwise ? d->rows / 2 : d->rows;
    return (out_w % 4 == 0 && d->stride % 4 == 0 && d->offset % 4 == 0 && d->out_ld % 4 == 0 && d->rows % 4 == 0) ? 4 : 1;
}

size_t ft_lds_fwd_bytes(const bt_ft_desc *d, int tc, bool grouped) {
    return (size_t)(tc + (grouped ? 2 : 1)) * LDS_RT * sizeof(float) +
           (size_t)(LDS_THREADS / 64) * FWD_EPW * fwd_strip_words((d->nnz + 31) / 32) * sizeof(uint32_t);
}
size_t ft_lds_bwd_bytes(int tc) {  // ft_bwd_oct: gradient tile + zero row, dW tile, part slots (+ the head rows of a fused head)
    return ((size_t)(FT_CHUNK + 1) * LDS_RT + (size_t)tc * LDS_RT + OCT_SLOTS * LDS_RT) * sizeof(float);
}
bool ft_lds_fits(const bt_ft_desc *d, int tc, bool grouped) {
    return ft_vec(d) == 4 && d->rows % LDS_RT == 0 && ft_lds_bwd_bytes(tc) <= 160 * 1024 &&
           ft_lds_fwd_bytes(d, tc, grouped) <= 160 * 1024 && oct_item_cap(tc) <= 1024 &&
           oct_rows_cap(d->nnz, tc) < 65536 && oct_index_lds_bytes(tc) <= 150 * 1024;
}

int ft_plan(const bt_ft_desc *d, FtPlan *p) {
    p->vec = ft_vec(d);
    // Segment-grouped mode: only when the whole table does not fit the LDS kernels but one segment does (and the
    // index lists allow the aligned single-round forward); otherwise `segment` is ignored.
    const long long E = (long long)d->n_inputs * d->batch;
    p->nseg = d->segment > 0 ? d->cols / d->segment : 1;
    p->grouped = d->segment > 0 && p->nseg > 1 && p->nseg <= 256 && !force_generic() &&
                 (!ft_lds_fits(d, d->cols, false) || g_ft_env.force_grouped) &&
                 ft_lds_fits(d, d->segment, true) && d->nnz % 4 == 0 && d->nnz <= 32 &&
                 (size_t)bt_div_up(E, GRP_BLOCK) * p->nseg * sizeof(int) <= 128 * 1024;
    p->tc = p->grouped ? d->segment : d->cols;
    p->chunks_per_input = bt_div_up(d->batch, FT_CHUNK);
    p->nchunks = p->grouped ? bt_div_up(E, FT_CHUNK) + p->nseg : p->chunks_per_input * d->n_inputs;
    p->max_chunks = p->nchunks;
    // LDS budget for the index kernel: nwaves*cols + 64*nwaves words
    int wv = 8;
    while (wv > 1 && ((size_t)wv * p->tc + 64 * wv) * 4 > 150 * 1024) wv >>= 1;
    BT_REQUIRE(((size_t)wv * p->tc + 64 * wv) * 4 <= 150 * 1024, "bt_ft_backward: cols=%d too large", d->cols);
    p->index_waves = wv;
    p->fbits = 1;
    while ((1 << p->fbits) < p->tc) p->fbits++;
    size_t o = 0;
    p->off_sorted = o;
    o += bt_align((size_t)p->nchunks * ft_sorted_cap(d->nnz, p->tc) * sizeof(uint16_t));
    p->off_offs = o;
    o += bt_align((size_t)p->nchunks * (p->tc + 1) * sizeof(uint32_t));
    p->off_g = o;
    o += bt_align((size_t)d->n_inputs * d->batch * d->rows * sizeof(float));
    p->bias_blocks = bt_div_up((long long)d->n_inputs * d->batch, 256);
    p->off_bpart = o;
    o += bt_align((size_t)p->bias_blocks * d->rows * sizeof(float));
    // LDS-resident path: 32-row tiles, table tile + staging must fit the 160 KiB LDS
    p->lds_fwd_bytes = ft_lds_fwd_bytes(d, p->tc, p->grouped);
    p->lds_path = ft_lds_fits(d, p->tc, p->grouped);
    p->ntiles = d->rows / LDS_RT;
    p->G = 1, p->chunks_per_wg = p->nchunks;
    if (p->grouped) {
        // item sizes: ~512 / ntiles items per launch (two workgroups per CU when the segments are balanced; every
        // workgroup pays a few us of tile load / slab write-back, so smaller items cost more than they balance)
        const int tgt_f = std::max(1, 512 / p->ntiles), tgt_b = std::max(1, 512 / p->ntiles);
        int fi = (int)std::min<long long>(4096, std::max<long long>(128, (E / tgt_f + 127) / 128 * 128));
        int gc = std::max(1, std::min(16, p->max_chunks / tgt_b));
        p->fwd_item = g_ft_env.fwd_item > 0 ? g_ft_env.fwd_item : fi, p->g_ch = g_ft_env.bwd_chunks > 0 ? g_ft_env.bwd_chunks : gc;
        p->fwd_item = std::max(128, p->fwd_item / 128 * 128), p->g_ch = std::max(1, p->g_ch);
        p->grp_blocks = bt_div_up(E, GRP_BLOCK);
        p->max_fitems = bt_div_up(E, p->fwd_item) + p->nseg;
        p->max_bitems = bt_div_up(p->max_chunks, p->g_ch) + p->nseg;
        p->G = p->max_bitems;
        p->off_pw = o;
        o += bt_align((size_t)p->max_bitems * p->tc * d->stride * sizeof(float));
        p->off_pb = o;
        o += bt_align((size_t)p->max_bitems * d->rows * sizeof(float));
        p->off_segof = o, o += bt_align((size_t)E);
        p->off_blkcnt = o, o += bt_align((size_t)p->grp_blocks * p->nseg * sizeof(int));
        p->off_elist = o, o += bt_align((size_t)E * sizeof(int));
        p->off_chunks = o, o += bt_align((size_t)p->max_chunks * sizeof(int4));
        p->off_fitems = o, o += bt_align((size_t)p->max_fitems * sizeof(int4));
        p->off_bitems = o, o += bt_align((size_t)p->max_bitems * sizeof(int4));
        p->off_counts = o, o += bt_align(8 * sizeof(int));
        p->off_segitems = o, o += bt_align((size_t)(p->nseg + 1) * sizeof(int));
    } else if (p->lds_path) {
        int G = 256 / p->ntiles;
        if (G < 1) G = 1;
        if (G > p->nchunks) G = p->nchunks;
        p->chunks_per_wg = bt_div_up(p->nchunks, G);
        p->G = bt_div_up(p->nchunks, p->chunks_per_wg);
        p->off_pw = o;
        o += bt_align((size_t)p->G * d->cols * d->stride * sizeof(float));
        p->off_pb = o;
        o += bt_align((size_t)p->G * d->rows * sizeof(float));
    }
    p->rows_cap = oct_rows_cap(d->nnz, p->tc);
    p->lds_oct_bytes = ft_lds_bwd_bytes(p->tc);
    p->oct = p->lds_path;  // (the octet index and ft_bwd_oct serve every shape of the LDS path)
    p->off_stream = p->off_meta = p->off_pw1 = 0;
    if (p->oct) {
        p->off_stream = o, o += bt_align((size_t)p->nchunks * p->rows_cap * 8 * sizeof(uint2));
        p->off_meta = o, o += bt_align((size_t)p->nchunks * OCT_META_BYTES);
    }
    // (every chunk of a workgroup must belong to one input; the waves' accumulators are merged through the dW tile's LDS)
    p->pw1_stride = (size_t)d->n_inputs * d->rows * FtPlan::kFusedMaxRows;
    p->hdw = p->oct && !p->grouped && !d->pairwise && p->chunks_per_input % p->chunks_per_wg == 0 &&
             (size_t)p->tc * LDS_RT >= (size_t)(LDS_THREADS / 64) * FtPlan::kFusedMaxRows * 64;
    if (p->hdw) p->off_pw1 = o, o += bt_align((size_t)p->G * p->pw1_stride * sizeof(float));
    // (not for segment-grouped tables: tried, by list position, with and without the parity order -- cfg3 0.449 -> 0.446 ms,
    // all of it from the parity order, which makes the grouped forward differ from the ungrouped one in bits and moves a
    // few dozen CReLU boundary cases of 19 M gradient elements past the bar between the two paths: not kept)
    p->predec = p->lds_path && !p->grouped && d->nnz <= 32 && !g_ft_env.no_predec;
    p->off_foffs = p->off_fblk = 0;
    if (p->predec) {
        p->off_foffs = o, o += bt_align((size_t)E * 32 * sizeof(uint32_t));
        p->off_fblk = o, o += bt_align((size_t)bt_div_up(E, 8));
    }
    {
        int G = std::max(1, 256 / std::max(1, p->ntiles));
        long long epw = (E + G - 1) / G;
        epw = (epw + 127) / 128 * 128;  // whole passes of 16 waves x 8 entries
        p->fwd_G = (int)((E + epw - 1) / epw), p->fwd_epw = (int)epw;
    }
    p->total = o;
    return BT_OK;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    return bt_set_max_lds(reinterpret_cast<const void *>(kernel), bytes);
}

// the group tables inside a workspace; `fwd` selects which item list the kernels walk
FtGrp ft_grp(const FtPlan &p, char *ws, bool fwd) {
    if (!p.grouped) return FtGrp{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    int *counts = reinterpret_cast<int *>(ws + p.off_counts);
    return FtGrp{reinterpret_cast<const int *>(ws + p.off_elist),
                 reinterpret_cast<const int4 *>(ws + (fwd ? p.off_fitems : p.off_bitems)),
                 reinterpret_cast<const int4 *>(ws + p.off_chunks), counts, counts + 3, p.tc};
}

thread_local const bt_ft_step_all *tl_step = nullptr;  // bt_internal_ft_arm_step

int ft_backward_lds_launch(hipStream_t s, const bt_ft_desc *d, const FtPlan &p, const float *grad_out, FtFusedSrc fs,
                           const float *out, const float *saved, const uint16_t *sorted, const uint32_t *offs, char *ws,
                           float *dw, float *dbias, const bt_ft_head_grads *head = nullptr) {
    float *pw = reinterpret_cast<float *>(ws + p.off_pw), *pb = reinterpret_cast<float *>(ws + p.off_pb);
    dim3 grid(p.ntiles * p.G), block(LDS_THREADS);
    const bool fused = fs.gz != nullptr;
    const size_t lds = p.lds_oct_bytes + (fused ? (size_t)d->n_inputs * fs.w1_rows * 32 * sizeof(float) : 0);
    const FtGrp gp = ft_grp(p, ws, false);
    const int cpi = p.grouped ? 1 : p.chunks_per_input;
    {
        const uint2 *stream = reinterpret_cast<const uint2 *>(ws + p.off_stream);
        const uint8_t *meta = reinterpret_cast<const uint8_t *>(ws + p.off_meta);
        const FtHeadGradSink hg{head ? reinterpret_cast<float *>(ws + p.off_pw1) : nullptr, p.pw1_stride};
#define BWD_OCT(PAIR, FUSED, GRP)                                                                                   \
    do {                                                                                                            \
        if (int rc = set_lds(ft_bwd_oct<PAIR, FUSED, GRP, false>, lds)) return rc;                                  \
        BT_LAUNCH(BTK_FT_BWD_LDS, s, (ft_bwd_oct<PAIR, FUSED, GRP, false>), grid, block, lds, grad_out, fs, out, saved, \
                  stream, meta, pw, pb, *d, p.ntiles, cpi, p.chunks_per_wg, gp, p.rows_cap, hg);                    \
    } while (0)
#define BWD_OCT2(PAIR, FUSED)                  \
    do {                                       \
        if (p.grouped) BWD_OCT(PAIR, FUSED, true); \
        else BWD_OCT(PAIR, FUSED, false);      \
    } while (0)
        if (head) {
            if (int rc = set_lds(ft_bwd_oct<false, true, false, true>, lds)) return rc;
            BT_LAUNCH(BTK_FT_BWD_LDS, s, (ft_bwd_oct<false, true, false, true>), grid, block, lds, grad_out, fs, out, saved,
                      stream, meta, pw, pb, *d, p.ntiles, cpi, p.chunks_per_wg, gp, p.rows_cap, hg);
        } else if (d->pairwise) {
            if (fused) BWD_OCT2(true, true);
            else BWD_OCT2(true, false);
        } else {
            if (fused) BWD_OCT2(false, true);
            else BWD_OCT2(false, false);
        }
#undef BWD_OCT2
#undef BWD_OCT
    }
    BT_LAUNCH_CHECK();
    if (d->rows != d->stride) BT_HIP(hipMemsetAsync(dw, 0, (size_t)d->stride * d->cols * sizeof(float), s));
    const long long n = (long long)d->cols * (d->rows / 4) + d->rows;
    if (head) {
        const int n1 = d->n_inputs * d->rows * fs.w1_rows, blocks_w = bt_div_up(n, 256);
        const FtHeadMerge hm{reinterpret_cast<const float *>(ws + p.off_pw1), p.pw1_stride, head->dw1, head->db1,
                             head->gz_parts, head->loss_partials, head->n_loss_partials, head->loss_sum, fs.w1_rows,
                             p.chunks_per_input / p.chunks_per_wg, n1};
        const dim3 mgrid(blocks_w + bt_div_up(n1, 256) + 1 + fs.w1_rows);
        BT_REQUIRE(!head->step || (d->rows == d->stride && d->offset == 0 && dbias && head->step->t[0].grad == dw &&
                                   head->step->t[1].grad == dbias && head->step->t[2].grad == head->dw1 &&
                                   head->step->t[3].grad == head->db1),
                   "bt_ft_backward(fused step): the optimiser tensors do not match the gradient buffers");
        if (head->done) bt_attach_stop_event(BTK_FT_BWD_MERGE, head->done);
        if (head->step) {
            BT_LAUNCH(BTK_FT_BWD_MERGE, s, ft_bwd_merge_head<true>, mgrid, dim3(256), 0, pw, pb, dw, dbias, *d, p.G, hm,
                      blocks_w, *head->step);
        } else {
            BT_LAUNCH(BTK_FT_BWD_MERGE, s, ft_bwd_merge_head<false>, mgrid, dim3(256), 0, pw, pb, dw, dbias, *d, p.G, hm,
                      blocks_w, bt_ft_fused_adamw{});
        }
    } else if (const bt_ft_step_all *step = tl_step) {
        tl_step = nullptr;
        BT_REQUIRE(d->rows == d->stride && d->offset == 0 && dbias && step->base.t[0].grad == dw &&
                       step->base.t[1].grad == dbias && step->n_extra >= 0 && step->n_extra <= BT_ADAMW_MAX_TENSORS,
                   "bt_ft_backward(fused step): the optimiser tensors do not match the gradient buffers");
        FtStepTab st;
        st.t0 = step->base.t[0], st.t1 = step->base.t[1], st.adj = step->base.adj, st.lr = step->base.lr;
        st.n_extra = step->n_extra;
        int xb = 0;
        for (int i = 0; i < step->n_extra; i++) {
            st.block_start[i] = xb, st.extra[i] = step->extra[i];
            xb += bt_div_up((long long)step->extra[i].n, 1024);
        }
        st.block_start[step->n_extra] = xb;
        const int blocks_w = bt_div_up(n, 256);
        if (step->wait) BT_HIP(hipStreamWaitEvent(s, step->wait, 0));
        if (step->done) bt_attach_stop_event(BTK_FT_BWD_MERGE, step->done);
        if (p.grouped)
            BT_LAUNCH(BTK_FT_BWD_MERGE, s, ft_bwd_merge_step<true>, dim3(blocks_w + xb), dim3(256), 0, pw, pb, dw, dbias, *d,
                      p.tc, reinterpret_cast<const int *>(ws + p.off_segitems),
                      reinterpret_cast<const int *>(ws + p.off_counts), blocks_w, st);
        else
            BT_LAUNCH(BTK_FT_BWD_MERGE, s, ft_bwd_merge_step<false>, dim3(blocks_w + xb), dim3(256), 0, pw, pb, dw, dbias, *d,
                      p.G, nullptr, nullptr, blocks_w, st);
    } else if (p.grouped)
        BT_LAUNCH(BTK_FT_BWD_MERGE, s, ft_bwd_merge_grp, dim3(bt_div_up(n, 256)), dim3(256), 0, pw, pb, dw, dbias, *d, p.tc,
                  reinterpret_cast<const int *>(ws + p.off_segitems), reinterpret_cast<const int *>(ws + p.off_counts));
    else
        BT_LAUNCH(BTK_FT_BWD_MERGE, s, ft_bwd_merge, dim3(bt_div_up(n, 256)), dim3(256), 0, pw, pb, dw, dbias, *d, p.G);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

}  // namespace

void bt_internal_ft_arm_step(const bt_ft_step_all *step) { tl_step = step; }
bool bt_internal_ft_step_ok(const bt_ft_desc *d) {
    FtPlan p;
    if (ft_validate(d, "ft_step_ok") || ft_plan(d, &p)) return false;
    return p.lds_path && !force_generic() && d->rows == d->stride && d->offset == 0;
}

// ---- internal (not part of the public header): FT backward fed by a one-output dense head, used by net.hip
bool bt_internal_ft_fused_ok(const bt_ft_desc *d, int head_rows) {
    FtPlan p;
    if (ft_validate(d, "ft_fused_ok") || ft_plan(d, &p)) return false;
    return p.lds_path && !force_generic() && head_rows >= 1 && head_rows <= FtPlan::kFusedMaxRows &&
           p.lds_oct_bytes + (size_t)d->n_inputs * head_rows * 32 * sizeof(float) <= 160 * 1024;
}

bool bt_internal_ft_fused_head_grad_ok(const bt_ft_desc *d, int head_rows) {
    FtPlan p;
    return bt_internal_ft_fused_ok(d, head_rows) && !ft_plan(d, &p) && p.hdw;
}

int bt_internal_ft_backward_fused(void *stream, const bt_ft_desc *d, const float *gz, const float *w1,
                                  const int32_t *bucket, int head_rows, const float *out, const float *saved, float *dw,
                                  float *dbias, void *workspace, size_t workspace_bytes, const bt_ft_head_grads *head) {
    if (int rc = ft_validate(d, "bt_ft_backward")) return rc;
    BT_REQUIRE(gz && w1 && dw && workspace, "bt_ft_backward(fused): null pointer");
    BT_REQUIRE(bt_internal_ft_fused_ok(d, head_rows), "bt_ft_backward(fused): shape not supported");
    FtPlan p;
    if (int rc = ft_plan(d, &p)) return rc;
    BT_REQUIRE(workspace_bytes >= p.total, "bt_ft_backward: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    if (head) {
        BT_REQUIRE(p.hdw, "bt_ft_backward(fused): head gradients not supported for this shape");
        BT_REQUIRE(head->dw1 && head->db1 && head->gz_parts && head->loss_partials && head->loss_sum && out,
                   "bt_ft_backward(fused): null head gradient pointer");
    }
    char *ws = static_cast<char *>(workspace);
    return ft_backward_lds_launch(bt_stream(stream), d, p, nullptr, FtFusedSrc{gz, w1, bucket, head_rows}, out, saved,
                                  reinterpret_cast<const uint16_t *>(ws + p.off_sorted),
                                  reinterpret_cast<const uint32_t *>(ws + p.off_offs), ws, dw, dbias, head);
}

namespace {

size_t ft_head_lds_bytes(const bt_ft_desc *d, int head_rows) {
    return (size_t)d->n_inputs * head_rows * (d->pairwise ? 16 : 32) * sizeof(float);
}

int ft_forward_lds_launch(hipStream_t s, const bt_ft_desc *d, const FtPlan &p, const float *w, const float *bias,
                          const int32_t *i0, const int32_t *i1, float *out, float *saved, const FtHeadSink *head,
                          char *ws = nullptr, bool pre_decoded = false) {
    pre_decoded = pre_decoded && p.predec && head && ws;
    const int G = p.fwd_G;
    const long long epw = p.fwd_epw;
    const int nround = (d->nnz + 31) / 32;
    const bool al4 = d->nnz % 4 == 0 && (reinterpret_cast<uintptr_t>(i0) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(i1) & 15) == 0;
    dim3 lgrid(p.ntiles * G), lblock(LDS_THREADS);
    const FtHeadSink hs = head ? *head : FtHeadSink{nullptr, nullptr, 0, nullptr};
    const size_t lds = p.lds_fwd_bytes + (head ? ft_head_lds_bytes(d, hs.w1_rows) : 0);
    if (p.grouped) {
        BT_REQUIRE(ws, "bt_ft_forward: segment-grouped forward needs the workspace of bt_ft_build_index");
        BT_REQUIRE(al4, "bt_ft_forward: segment-grouped forward needs 16-byte aligned index lists");
        const FtGrp gp = ft_grp(p, ws, true);
        dim3 ggrid(p.ntiles * p.max_fitems);
#define FWD_GRP(PAIR, HEAD)                                                                                         \
    do {                                                                                                            \
        if (int rc = set_lds(ft_fwd_lds<PAIR, true, 1, HEAD, true>, lds)) return rc;                                \
        BT_LAUNCH(BTK_FT_FWD_LDS, s, (ft_fwd_lds<PAIR, true, 1, HEAD, true>), ggrid, lblock, lds, w, bias, i0, i1, out, \
                  saved, *d, p.ntiles, 0, hs, gp, FtPre{});                                                         \
    } while (0)
        if (d->pairwise) {
            if (head) FWD_GRP(true, true);
            else FWD_GRP(true, false);
        } else {
            if (head) FWD_GRP(false, true);
            else FWD_GRP(false, false);
        }
#undef FWD_GRP
        BT_LAUNCH_CHECK();
        return BT_OK;
    }
    if (pre_decoded) {  // (the caller ran bt_internal_ft_build_groups on this batch: net.hip's head path)
        const FtPre pre{reinterpret_cast<const lds_u4_t *>(ws + p.off_foffs), reinterpret_cast<const uint8_t *>(ws + p.off_fblk)};
#define FWD_PRE(PAIR)                                                                                              \
    do {                                                                                                           \
        if (int rc = set_lds(ft_fwd_lds<PAIR, true, 1, true, false, true>, lds)) return rc;                        \
        BT_LAUNCH(BTK_FT_FWD_LDS, s, (ft_fwd_lds<PAIR, true, 1, true, false, true>), lgrid, lblock, lds, w, bias, i0, i1, \
                  out, saved, *d, p.ntiles, (int)epw, hs, FtGrp{}, pre);                                           \
    } while (0)
        if (d->pairwise) FWD_PRE(true);
        else FWD_PRE(false);
#undef FWD_PRE
        BT_LAUNCH_CHECK();
        return BT_OK;
    }
#define FWD_LDS(PAIR, AL4, NROUND, HEAD)                                                                           \
    do {                                                                                                           \
        if (int rc = set_lds(ft_fwd_lds<PAIR, AL4, NROUND, HEAD, false>, lds)) return rc;                          \
        BT_LAUNCH(BTK_FT_FWD_LDS, s, (ft_fwd_lds<PAIR, AL4, NROUND, HEAD, false>), lgrid, lblock, lds, w, bias, i0, i1, \
                  out, saved, *d, p.ntiles, (int)epw, hs, FtGrp{}, FtPre{});                                       \
    } while (0)
#define FWD_LDS3(PAIR, AL4, NROUND)                                                                                \
    do {                                                                                                           \
        if (head) FWD_LDS(PAIR, AL4, NROUND, true);                                                                \
        else FWD_LDS(PAIR, AL4, NROUND, false);                                                                    \
    } while (0)
#define FWD_LDS2(PAIR, AL4)                                                                                        \
    do {                                                                                                           \
        if (nround == 1) FWD_LDS3(PAIR, AL4, 1);                                                                   \
        else FWD_LDS3(PAIR, AL4, 2);                                                                               \
    } while (0)
    if (d->pairwise) {
        if (al4) FWD_LDS2(true, true);
        else FWD_LDS2(true, false);
    } else {
        if (al4) FWD_LDS2(false, true);
        else FWD_LDS2(false, false);
    }
#undef FWD_LDS2
#undef FWD_LDS3
#undef FWD_LDS
    BT_LAUNCH_CHECK();
    return BT_OK;
}

}  // namespace

// ---- internal: FT forward that also leaves the partial dot products of a one-output dense head (net.hip)
bool bt_internal_ft_forward_head_ok(const bt_ft_desc *d, int head_rows) {
    FtPlan p;
    if (ft_validate(d, "ft_forward_head_ok") || ft_plan(d, &p)) return false;
    return p.lds_path && !force_generic() && head_rows >= 1 && head_rows <= FtPlan::kFusedMaxRows &&
           p.lds_fwd_bytes + ft_head_lds_bytes(d, head_rows) <= 160 * 1024;
}

int bt_internal_ft_head_parts(const bt_ft_desc *d) { return d->n_inputs * (d->rows / LDS_RT); }

int bt_internal_ft_forward_head(void *stream, const bt_ft_desc *d, const float *w, const float *bias,
                                const int32_t *const *idx, float *out, float *saved, const float *w1,
                                const int32_t *bucket, int head_rows, float *zpart, void *workspace,
                                size_t workspace_bytes) {
    if (int rc = ft_validate(d, "bt_ft_forward")) return rc;
    BT_REQUIRE(w && idx && idx[0] && out && w1 && zpart, "bt_ft_forward(head): null pointer");
    BT_REQUIRE(d->n_inputs == 1 || idx[1], "bt_ft_forward: null index pointer for input 1");
    BT_REQUIRE(!d->pairwise || saved, "bt_ft_forward: pairwise needs a `saved` buffer");
    BT_REQUIRE(bt_internal_ft_forward_head_ok(d, head_rows), "bt_ft_forward(head): shape not supported");
    FtPlan p;
    if (int rc = ft_plan(d, &p)) return rc;
    BT_REQUIRE(!p.grouped || (workspace && workspace_bytes >= p.total), "bt_ft_forward(head): workspace too small");
    const FtHeadSink hs{w1, bucket, head_rows, zpart};
    BT_REQUIRE(!p.predec || !workspace || workspace_bytes >= p.total, "bt_ft_forward(head): workspace too small");
    return ft_forward_lds_launch(bt_stream(stream), d, p, w, bias, idx[0], d->n_inputs > 1 ? idx[1] : idx[0], out, saved,
                                 &hs, static_cast<char *>(workspace), workspace != nullptr);
}

// ---- internal: the two halves of bt_ft_build_index. The entry lists (segment-grouped tables only) are needed by
// the forward pass, the per-chunk sorted lists only by the backward pass, so net.hip puts them on different streams.
int bt_internal_ft_build_groups(void *stream, const bt_ft_desc *d, const int32_t *const *idx, void *workspace,
                                size_t workspace_bytes, hipEvent_t done) {
    if (int rc = ft_validate(d, "bt_ft_build_index")) return rc;
    BT_REQUIRE(idx && idx[0] && workspace, "bt_ft_build_index: null pointer");
    BT_REQUIRE(d->n_inputs == 1 || idx[1], "bt_ft_build_index: null index pointer for input 1");
    FtPlan p;
    if (int rc = ft_plan(d, &p)) return rc;
    BT_REQUIRE(workspace_bytes >= p.total, "bt_ft_build_index: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    hipStream_t s = bt_stream(stream);
    char *ws = static_cast<char *>(workspace);
    const int32_t *i0 = idx[0], *i1 = d->n_inputs > 1 ? idx[1] : idx[0];
    const int E = d->n_inputs * d->batch;
    if (p.predec) {
        lds_u4_t *foffs = reinterpret_cast<lds_u4_t *>(ws + p.off_foffs);
        uint8_t *fblk = reinterpret_cast<uint8_t *>(ws + p.off_fblk);
        const dim3 dgrid(bt_div_up((long long)E * 8, 256));
        if (done) bt_attach_stop_event(BTK_FT_INDEX, done);
        if (g_ft_env.decode_parity)
            BT_LAUNCH(BTK_FT_INDEX, s, ft_fwd_decode_kernel<true>, dgrid, dim3(256), 0, i0, i1, d->batch, d->nnz, d->cols, E, foffs,
                      fblk);
        else
            BT_LAUNCH(BTK_FT_INDEX, s, ft_fwd_decode_kernel<false>, dgrid, dim3(256), 0, i0, i1, d->batch, d->nnz, d->cols, E, foffs,
                      fblk);
        BT_LAUNCH_CHECK();
        return BT_OK;
    }
    if (!p.grouped) return BT_OK;
    uint8_t *seg_of = reinterpret_cast<uint8_t *>(ws + p.off_segof);
    int *blk = reinterpret_cast<int *>(ws + p.off_blkcnt);
    BT_LAUNCH(BTK_FT_INDEX, s, ft_group_count, dim3(p.grp_blocks), dim3(GRP_BLOCK), 0, i0, i1, d->batch, d->nnz, d->cols,
              p.tc, p.nseg, E, seg_of, blk);
    BT_LAUNCH_CHECK();
    if (done) bt_attach_stop_event(BTK_FT_INDEX, done);
    BT_LAUNCH(BTK_FT_INDEX, s, ft_group_place, dim3(p.grp_blocks), dim3(GRP_BLOCK), 0, seg_of, blk, p.grp_blocks, p.nseg, E,
              p.fwd_item, p.g_ch, reinterpret_cast<int *>(ws + p.off_elist), reinterpret_cast<int4 *>(ws + p.off_chunks),
              reinterpret_cast<int4 *>(ws + p.off_fitems), reinterpret_cast<int4 *>(ws + p.off_bitems),
              reinterpret_cast<int *>(ws + p.off_segitems), reinterpret_cast<int *>(ws + p.off_counts));
    BT_LAUNCH_CHECK();
    return BT_OK;
}

int bt_internal_ft_build_sorted(void *stream, const bt_ft_desc *d, const int32_t *const *idx, void *workspace,
                                size_t workspace_bytes, hipEvent_t done) {
    if (int rc = ft_validate(d, "bt_ft_build_index")) return rc;
    BT_REQUIRE(idx && idx[0] && workspace, "bt_ft_build_index: null pointer");
    BT_REQUIRE(d->n_inputs == 1 || idx[1], "bt_ft_build_index: null index pointer for input 1");
    FtPlan p;
    if (int rc = ft_plan(d, &p)) return rc;
    BT_REQUIRE(workspace_bytes >= p.total, "bt_ft_build_index: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    hipStream_t s = bt_stream(stream);
    char *ws = static_cast<char *>(workspace);
    uint16_t *sorted = reinterpret_cast<uint16_t *>(ws + p.off_sorted);
    uint32_t *offs = reinterpret_cast<uint32_t *>(ws + p.off_offs);
    const int32_t *i0 = idx[0], *i1 = d->n_inputs > 1 ? idx[1] : idx[0];
    const int threads = p.index_waves * 64;
    const size_t lds = ((size_t)p.index_waves * p.tc + threads) * sizeof(uint32_t);
    const FtGrp gp = ft_grp(p, ws, false);
    if (p.oct && !force_generic()) {
        uint2 *stream = reinterpret_cast<uint2 *>(ws + p.off_stream);
        uint8_t *meta = reinterpret_cast<uint8_t *>(ws + p.off_meta);
        const size_t olds = oct_index_lds_bytes(p.tc);
        if (done) bt_attach_stop_event(BTK_FT_INDEX, done);  // (one launch follows: the event rides on it)
        if (p.grouped) {
            if (int rc = set_lds(ft_index_oct_kernel<true>, olds)) return rc;
            BT_LAUNCH(BTK_FT_INDEX, s, ft_index_oct_kernel<true>, dim3(p.nchunks), dim3(OCT_IDX_THREADS), olds, i0, i1,
                      d->batch, d->nnz, p.tc, 1, p.fbits, stream, meta, gp, d->cols);
        } else {
            if (int rc = set_lds(ft_index_oct_kernel<false>, olds)) return rc;
            BT_LAUNCH(BTK_FT_INDEX, s, ft_index_oct_kernel<false>, dim3(p.nchunks), dim3(OCT_IDX_THREADS), olds, i0, i1,
                      d->batch, d->nnz, d->cols, p.chunks_per_input, p.fbits, stream, meta, gp, d->cols);
        }
        BT_LAUNCH_CHECK();
        return BT_OK;
    }
    // (a segment-grouped table fits the LDS kernels by definition, ft_plan: it took the branch above)
    if (done) bt_attach_stop_event(BTK_FT_INDEX, done);
    BT_LAUNCH(BTK_FT_INDEX, s, ft_index_kernel, dim3(p.nchunks), dim3(threads), lds, i0, i1, d->batch, d->nnz, d->cols,
              p.chunks_per_input, p.fbits, sorted, offs);
    BT_LAUNCH_CHECK();
    return BT_OK;
}

// tests change the switches between cases of one process: they call this after touching the environment
extern "C" void bt_internal_reload_env(void) { g_ft_env.reload(); }
#ifdef BT_STAMPS
extern "C" int bt_internal_read_stamps(void *dst, size_t bytes) {
    BT_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(bt_stamps), std::min(bytes, sizeof(bt_stamps))));
    return BT_OK;
}
#endif

bool bt_internal_ft_grouped(const bt_ft_desc *d) {
    FtPlan p;
    return !ft_validate(d, "ft_grouped") && !ft_plan(d, &p) && p.grouped;
}

extern "C" {

int bt_ft_forward(void *stream, const bt_ft_desc *d, const float *w, const float *bias, const int32_t *const *idx,
                  float *out, float *saved) {
    if (int rc = ft_validate(d, "bt_ft_forward")) return rc;
    BT_REQUIRE(w && idx && idx[0] && out, "bt_ft_forward: null pointer");
    BT_REQUIRE(d->n_inputs == 1 || idx[1], "bt_ft_forward: null index pointer for input 1");
    BT_REQUIRE(!d->pairwise || saved, "bt_ft_forward: pairwise needs a `saved` buffer");
    const int vec = ft_vec(d);
    const int out_w = d->pairwise ? d->rows / 2 : d->rows;
    const long long total = (long long)d->n_inputs * d->batch * (out_w / vec);
    dim3 grid(bt_div_up(total, 256)), block(256);
    hipStream_t s = bt_stream(stream);
    const int32_t *i0 = idx[0], *i1 = d->n_inputs > 1 ? idx[1] : idx[0];
    FtPlan p;
    if (int rc = ft_plan(d, &p)) return rc;
    // (segment-grouped tables need the entry lists: bt_ft_forward_prebuilt; here they take the any-shape kernel)
    if (p.lds_path && !p.grouped && !force_generic())
        return ft_forward_lds_launch(s, d, p, w, bias, i0, i1, out, saved, nullptr);
    if (vec == 4) {
        if (d->pairwise)
            BT_LAUNCH(BTK_FT_FWD_GENERIC, s, (ft_fwd_generic<4, true>), grid, block, 0, w, bias, i0, i1, out, saved, *d);
        else
            BT_LAUNCH(BTK_FT_FWD_GENERIC, s, (ft_fwd_generic<4, false>), grid, block, 0, w, bias, i0, i1, out, saved, *d);
    } else {
        if (d->pairwise)
            BT_LAUNCH(BTK_FT_FWD_GENERIC, s, (ft_fwd_generic<1, true>), grid, block, 0, w, bias, i0, i1, out, saved, *d);
        else
            BT_LAUNCH(BTK_FT_FWD_GENERIC, s, (ft_fwd_generic<1, false>), grid, block, 0, w, bias, i0, i1, out, saved, *d);
    }
    BT_LAUNCH_CHECK();
    return BT_OK;
}

size_t bt_ft_backward_workspace_bytes(const bt_ft_desc *d) {
    if (ft_validate(d, "bt_ft_backward_workspace_bytes")) return 0;
    FtPlan p;
    if (ft_plan(d, &p)) return 0;
    return p.total;
}

int bt_ft_build_index(void *stream, const bt_ft_desc *d, const int32_t *const *idx, void *workspace,
                      size_t workspace_bytes) {
    if (int rc = ft_validate(d, "bt_ft_build_index")) return rc;
    BT_REQUIRE(idx && idx[0] && workspace, "bt_ft_build_index: null pointer");
    BT_REQUIRE(d->n_inputs == 1 || idx[1], "bt_ft_build_index: null index pointer for input 1");
    FtPlan p;
    if (int rc = ft_plan(d, &p)) return rc;
    BT_REQUIRE(workspace_bytes >= p.total, "bt_ft_build_index: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    if (int rc = bt_internal_ft_build_groups(stream, d, idx, workspace, workspace_bytes)) return rc;
    return bt_internal_ft_build_sorted(stream, d, idx, workspace, workspace_bytes);
}

int bt_ft_forward_prebuilt(void *stream, const bt_ft_desc *d, const float *w, const float *bias,
                           const int32_t *const *idx, float *out, float *saved, void *workspace, size_t workspace_bytes) {
    if (int rc = ft_validate(d, "bt_ft_forward")) return rc;
    FtPlan p;
    if (int rc = ft_plan(d, &p)) return rc;
    if (!p.grouped) return bt_ft_forward(stream, d, w, bias, idx, out, saved);
    BT_REQUIRE(w && idx && idx[0] && out && workspace, "bt_ft_forward: null pointer");
    BT_REQUIRE(d->n_inputs == 1 || idx[1], "bt_ft_forward: null index pointer for input 1");
    BT_REQUIRE(!d->pairwise || saved, "bt_ft_forward: pairwise needs a `saved` buffer");
    BT_REQUIRE(workspace_bytes >= p.total, "bt_ft_forward: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    return ft_forward_lds_launch(bt_stream(stream), d, p, w, bias, idx[0], d->n_inputs > 1 ? idx[1] : idx[0], out, saved,
                                 nullptr, static_cast<char *>(workspace));
}

int bt_ft_segment_violations(void *stream, const bt_ft_desc *d, void *workspace, size_t workspace_bytes, int *count) {
    if (int rc = ft_validate(d, "bt_ft_segment_violations")) return rc;
    BT_REQUIRE(workspace && count, "bt_ft_segment_violations: null pointer");
    FtPlan p;
    if (int rc = ft_plan(d, &p)) return rc;
    *count = 0;
    if (!p.grouped) return BT_OK;
    BT_REQUIRE(workspace_bytes >= p.total, "bt_ft_segment_violations: workspace too small");
    BT_HIP(hipMemcpyAsync(count, static_cast<char *>(workspace) + p.off_counts + 3 * sizeof(int), sizeof(int),
                          hipMemcpyDeviceToHost, bt_stream(stream)));
    BT_HIP(hipStreamSynchronize(bt_stream(stream)));
    return BT_OK;
}

int bt_ft_backward(void *stream, const bt_ft_desc *d, const int32_t *const *idx, const float *grad_out,
                   const float *out, const float *saved, float *dw, float *dbias, void *workspace,
                   size_t workspace_bytes) {
    if (int rc = bt_ft_build_index(stream, d, idx, workspace, workspace_bytes)) return rc;
    return bt_ft_backward_prebuilt(stream, d, grad_out, out, saved, dw, dbias, workspace, workspace_bytes);
}

int bt_ft_backward_prebuilt(void *stream, const bt_ft_desc *d, const float *grad_out, const float *out,
                            const float *saved, float *dw, float *dbias, void *workspace, size_t workspace_bytes) {
    if (int rc = ft_validate(d, "bt_ft_backward")) return rc;
    BT_REQUIRE(grad_out && dw && workspace, "bt_ft_backward: null pointer");
    BT_REQUIRE(d->pairwise ? (saved != nullptr) : (out != nullptr), "bt_ft_backward: missing forward activations");
    FtPlan p;
    if (int rc = ft_plan(d, &p)) return rc;
    BT_REQUIRE(workspace_bytes >= p.total, "bt_ft_backward: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    hipStream_t s = bt_stream(stream);
    char *ws = static_cast<char *>(workspace);
    uint16_t *sorted = reinterpret_cast<uint16_t *>(ws + p.off_sorted);
    uint32_t *offs = reinterpret_cast<uint32_t *>(ws + p.off_offs);
    float *g = reinterpret_cast<float *>(ws + p.off_g);
    float *bpart = reinterpret_cast<float *>(ws + p.off_bpart);

    if (p.lds_path && !force_generic())
        return ft_backward_lds_launch(s, d, p, grad_out, FtFusedSrc{nullptr, nullptr, nullptr, 0}, out, saved, sorted, offs,
                                      ws, dw, dbias);
    {
        const int out_w = d->pairwise ? d->rows / 2 : d->rows;
        const long long total = (long long)d->n_inputs * d->batch * (out_w / p.vec);
        dim3 grid(bt_div_up(total, 256)), block(256);
        if (p.vec == 4) {
            if (d->pairwise)
                BT_LAUNCH(BTK_FT_BWD_MAKE_G, s, (ft_bwd_make_g<4, true>), grid, block, 0, grad_out, out, saved, g, *d);
            else
                BT_LAUNCH(BTK_FT_BWD_MAKE_G, s, (ft_bwd_make_g<4, false>), grid, block, 0, grad_out, out, saved, g, *d);
        } else {
            if (d->pairwise)
                BT_LAUNCH(BTK_FT_BWD_MAKE_G, s, (ft_bwd_make_g<1, true>), grid, block, 0, grad_out, out, saved, g, *d);
            else
                BT_LAUNCH(BTK_FT_BWD_MAKE_G, s, (ft_bwd_make_g<1, false>), grid, block, 0, grad_out, out, saved, g, *d);
        }
        BT_LAUNCH_CHECK();
    }
    if (d->rows != d->stride)  // rows outside [offset, offset+rows) of each column stay zero (sparse.rs:168-170)
        BT_HIP(hipMemsetAsync(dw, 0, (size_t)d->stride * d->cols * sizeof(float), s));
    {
        int threads = bt_div_up(d->rows, p.vec);
        threads = ((threads + 63) / 64) * 64;
        if (threads > 256) threads = 256;
        if (p.vec == 4)
            BT_LAUNCH(BTK_FT_BWD_GATHER, s, (ft_bwd_gather<4>), dim3(d->cols), dim3(threads), 0, g, sorted, offs, dw, *d,
                               p.chunks_per_input);
        else
            BT_LAUNCH(BTK_FT_BWD_GATHER, s, (ft_bwd_gather<1>), dim3(d->cols), dim3(threads), 0, g, sorted, offs, dw, *d,
                               p.chunks_per_input);
        BT_LAUNCH_CHECK();
    }
    if (dbias) {
        const int M = d->n_inputs * d->batch;
        dim3 grid(bt_div_up(d->rows, 64), p.bias_blocks), block(64);
        BT_LAUNCH(BTK_COLSUM_PARTIAL, s, colsum_partial, grid, block, 0, g, M, d->rows, 256, bpart);
        BT_LAUNCH_CHECK();
        BT_LAUNCH(BTK_COLSUM_FINAL, s, colsum_final, dim3(bt_div_up(d->rows, 64)), dim3(64), 0, bpart, p.bias_blocks, d->rows,
                           dbias);
        BT_LAUNCH_CHECK();
    }
    return BT_OK;
}

}  // extern "C"
