// Developer microbenchmark: issue cost of the vector instructions the FT kernels are made of, at the kernels' own
// occupancy (16 waves per CU = 4 per SIMD, 256 workgroups). Prints shader cycles per wave-instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates && tools/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, float *sink, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) lds[i] = (float)i;
    __syncthreads();
    float a0 = threadIdx.x, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float b = sink[0];
    const f2 pb = {b, b};
    int iv = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if constexpr (MODE == 0) {  // 8 v_add_f32
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (MODE == 1) {  // 4 v_pk_add_f32 (the same 8 adds)
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
            } else if constexpr (MODE == 2) {  // 8 v_and_b32
                asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                             "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (MODE == 3) {  // 8 v_mov_b32 dpp row_share
                asm volatile("v_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0x9\n v_mov_b32_dpp %1, %1 row_newbcast:3 row_mask:0xf bank_mask:0x9\n"
                             "v_mov_b32_dpp %2, %2 row_newbcast:3 row_mask:0xf bank_mask:0x9\n v_mov_b32_dpp %3, %3 row_newbcast:3 row_mask:0xf bank_mask:0x9\n"
                             "v_mov_b32_dpp %4, %4 row_newbcast:3 row_mask:0xf bank_mask:0x9\n v_mov_b32_dpp %5, %5 row_newbcast:3 row_mask:0xf bank_mask:0x9\n"
                             "v_mov_b32_dpp %6, %6 row_newbcast:3 row_mask:0xf bank_mask:0x9\n v_mov_b32_dpp %7, %7 row_newbcast:3 row_mask:0xf bank_mask:0x9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (MODE == 4) {  // 8 v_sqrt_f32
                asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                             "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (MODE == 5) {  // 4 v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
            } else if constexpr (MODE == 6) {  // 8 v_add_u32 sdwa
                asm volatile("v_add_u32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_add_u32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                             : "+v"(iv), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (MODE == 7) {  // 8 v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                             "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    sink[1 + (blockIdx.x * 1024 + threadIdx.x) % 1024] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + iv + lds[threadIdx.x];
}

template <int MODE>
void run(const char *name, int per_iter) {
    unsigned long long *out;
    float *sink;
    hipMalloc(&out, 256 * 16 * 8);
    hipMalloc(&sink, 4096 * 4);
    hipMemset(sink, 0, 4096 * 4);
    const int iters = 2000;
    for (int rep = 0; rep < 3; rep++) k<MODE><<<256, 1024>>>(out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    s /= h.size();
    // 4 waves per SIMD each issue iters*4*per_iter instructions; SIMD time = wave time (they run concurrently)
    printf("%-22s wave cycles %.0f -> %.2f cycles per wave-instruction per SIMD (4 waves/SIMD)\n", name, s,
           s / (4.0 * iters * 4 * per_iter));
    hipFree(out);
    hipFree(sink);
}

int main() {
    run<0>("v_add_f32", 8);
    run<1>("v_pk_add_f32", 4);
    run<5>("v_pk_mul_f32", 4);
    run<7>("v_fma_f32", 8);
    run<2>("v_and_b32", 8);
    run<6>("v_add_u32_sdwa", 8);
    run<3>("v_mov_b32_dpp row_share", 8);
    run<4>("v_sqrt_f32", 8);
    return 0;
}
