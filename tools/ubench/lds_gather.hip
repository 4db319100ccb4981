// Developer microbenchmark: what the LDS gives the octet reduce loop. Every 8-lane group sums random 128-B rows of a
// 32 KiB tile (4 rows per "stream row", as ft_bwd_oct), with DEPTH stream rows in flight per wave and W waves per CU.
// Prints shader cycles per stream row per wave and the implied ds_read_b128 cost.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_gather.hip -o tools/ubench/lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void add4(float4 &a, const float4 &v) { a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w; }

__device__ __forceinline__ void slot(int lane, int &entry, int &sl) {  // fwd_gather_slot of ft.hip
    const int m = lane & 31, h = lane >> 5;
    int q;
    if (m < 4) q = 0, sl = m;
    else if (m < 12) q = 2, sl = m - 4;
    else if (m < 16) q = 0, sl = m - 8;
    else if (m < 20) q = 3, sl = m - 16;
    else if (m < 28) q = 1, sl = m - 20;
    else q = 3, sl = m - 24;
    entry = h * 4 + q;
}

template <int DEPTH, bool CONFLICT_FREE>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, float *sink, int rows) {
    extern __shared__ __align__(16) float gt[];  // [257][32]
    for (int i = threadIdx.x; i < 257 * 32; i += blockDim.x) gt[i] = (float)(i & 1023);
    __syncthreads();
    int ge, gs;
    slot(threadIdx.x & 63, ge, gs);
    const char *gb = reinterpret_cast<const char *>(gt) + gs * 16;
    uint32_t x = (threadIdx.x >> 6) * 977u + ge * 131u + 12345u;  // same stream for the 8 lanes of a group
    auto next = [&]() {
        x = x * 1664525u + 1013904223u;
        uint32_t o = (x >> 9) & 0x7f80u;
        if (CONFLICT_FREE) o = (o & ~0x80u) | ((ge & 1) << 7);  // the two groups of a 16-lane row on opposite bank halves
        return o;
    };
    float4 a = make_float4(0, 0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rows; r += DEPTH) {
        float4 v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int j = 0; j < 4; j++) v[d][j] = *reinterpret_cast<const float4 *>(gb + next());
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
            for (int j = 0; j < 4; j++) add4(a, v[d][j]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    sink[(blockIdx.x * 1024 + threadIdx.x) % 4096] = a.x + a.y + a.z + a.w;
}

template <int DEPTH, bool CF>
void run(int waves) {
    unsigned long long *out;
    float *sink;
    hipMalloc(&out, 256 * 16 * 8);
    hipMalloc(&sink, 4096 * 4);
    const int rows = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<DEPTH, CF>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (int rep = 0; rep < 3; rep++) k<DEPTH, CF><<<256, waves * 64, 140 * 1024>>>(out, sink, rows);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    int n = 0;
    for (int b = 0; b < 256; b++)
        for (int w = 0; w < waves; w++) s += (double)h[b * 16 + w], n++;
    s /= n;
    printf("depth %d %s waves/CU %2d: %.0f cycles per stream row per wave -> %.2f LDS cycles per ds_read_b128 (CU level)\n", DEPTH,
           CF ? "conflict-free" : "random       ", waves, s / rows, s / rows / 4.0 / waves);
    hipFree(out);
    hipFree(sink);
}

int main() {
    for (int w : {16, 12, 8, 4}) run<1, false>(w);
    for (int w : {16, 12, 8, 4}) run<2, false>(w);
    for (int w : {16, 8}) run<4, false>(w);
    for (int w : {16, 8, 4}) run<1, true>(w);
    for (int w : {16, 8}) run<2, true>(w);
    return 0;
}
