// Micro-benchmark: how fast can a CU run the inner pattern of the 64x64 conv kernel?
//   per iteration and wave: NREAD ds_read_b128 pairs + 4*NREAD... (see MODE below), no global memory.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_probe mfma_lds_probe.hip ; run: ./mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 2 reads -> 4 dependent 32x32x2 (the 64x64 kernel)      MODE 1: same, MFMAs only (no LDS reads)
// MODE 2: 2 reads -> 4 MFMAs alternating two accumulators        MODE 3: 4 reads -> 16 MFMAs 16x16x4 on 4 accumulators
// MODE 4: MODE 0 + a workgroup barrier per 4 read groups (one "K-tile")
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t; i < 8192; i += 256) smem[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int frag_row = lane & 31, half = lane >> 5, rswz = (frag_row >> 1) & 7;
    const float* rd_a = smem + ((wv >> 1) * 32 + frag_row) * 32;
    const float* rd_b = smem + 64 * 32 + ((wv & 1) * 32 + frag_row) * 32;
    f32x16 acc = {0}, acc2 = {0};
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int pc = ((kk * 2 + half) ^ rswz) * 4;
            float4 a, b;
            if (MODE == 1) { a = make_float4(1.f, 2.f, 3.f, 4.f); b = a; asm volatile("" : "+v"(a.x), "+v"(b.x)); }
            else { a = *reinterpret_cast<const float4*>(rd_a + pc); b = *reinterpret_cast<const float4*>(rd_b + pc); }
            if (MODE == 0 || MODE == 1 || MODE == 4) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            } else if (MODE == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc2, 0, 0, 0);
            } else {   // 16x16x4: 8 MFMAs of 32 clk = same MFMA time as 4 of 64
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.x, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.y, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c3, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.z, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.w, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c3, 0, 0, 0);
            }
        }
        if (MODE == 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    s += c0[0] + c1[1] + c2[2] + c3[3];
    if (s == 12345.678f) out[t] = s;
}

template <int MODE>
static void run(const char* name, int blocks_per_cu, float* out) {
    const int iters = 2000;
    const size_t lds = blocks_per_cu >= 5 ? 32768 : blocks_per_cu == 4 ? 36 * 1024 : blocks_per_cu == 3 ? 50 * 1024
                       : blocks_per_cu == 2 ? 72 * 1024 : 128 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), lds, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // MFMA time per wave-iteration = 16 * 64 clk = 1024 clk of its SIMD; per CU and iteration-round: blocks_per_cu * 1024 clk
    const double us_per_blockiter = ms * 1e3 / iters / blocks_per_cu;
    const double flop = (double)grid * 4 * iters * 16 * 4096.0;
    printf("%-34s blocks/CU %d: %7.3f us per block-iteration (ideal 0.465 @2.2GHz), %6.1f TFLOP/s\n", name, blocks_per_cu,
           us_per_blockiter, flop / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 4096);
    for (int b : {1, 2, 4, 5}) {
        run<1>("MFMA only (dependent chain)", b, out);
        run<0>("2 ds_read_b128 -> 4 dep MFMA", b, out);
        run<2>("2 ds_read_b128 -> 2x2 alternating", b, out);
        run<3>("2 ds_read_b128 -> 8x 16x16x4 (4 acc)", b, out);
        run<4>("as the 64x64 kernel + barrier/tile", b, out);
    }
    return 0;
}
