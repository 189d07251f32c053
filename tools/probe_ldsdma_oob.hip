// Probe: does `buffer_load_dwordx4 ... lds` write zeros for out-of-range lanes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, int n_valid_floats) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = -7.f;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n_valid_floats * 4, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // lanes with odd index ask for an out-of-range offset
    unsigned voff = (lane & 1) ? 0x7ffffff0u : (unsigned)(threadIdx.x * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) y[i] = lds[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i + 1;
    float *x, *y;
    hipMalloc(&x, 4096 * 4); hipMalloc(&y, 1024 * 4);
    hipMemcpy(x, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, x, y, 4096);
    std::vector<float> o(1024);
    hipMemcpy(o.data(), y, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t)
        for (int j = 0; j < 4; ++j) {
            float want = (t & 1) ? 0.f : (float)(t * 4 + j) + 1;
            if (o[t * 4 + j] != want) { if (bad < 8) printf("t=%d j=%d got %g want %g\n", t, j, o[t*4+j], want); ++bad; }
        }
    printf("lds-dma oob probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK zeros written", bad);
    return 0;
}
