// Shared helpers for the FGN HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FGN_OK 0
#define FGN_ERR_SHAPE (-1)      // operand shape not supported by the kernel
#define FGN_ERR_ARG (-2)        // null pointer / bad argument

#define FGN_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised on the
// function object of the CURRENT device (a process may drive several GPUs): set once per (call site, device);
// `done_mask` is a static of the call site, bit d = device d done.
static inline hipError_t fgn_allow_full_lds(const void* fn, unsigned long long* done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && ((*done_mask >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && dev < 64) *done_mask |= 1ull << dev;
    return e;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
