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

// Kernel-exact timing for a profiler (bench.py roofline): fgn_profile_next_launch(start, stop) arms the calling
// thread; the next launch that goes through FGN_LAUNCH_TIMED is dispatched with hipExtLaunchKernelGGL, which stamps
// the two events with the start / end of that kernel alone (what a rocprofv3 kernel trace reports), instead of
// bracketing it with hipEventRecord, which also measures the command processor's gaps around it.
#include <hip/hip_ext.h>
extern thread_local hipEvent_t fgn_prof_start;
extern thread_local hipEvent_t fgn_prof_stop;
#define FGN_LAUNCH_TIMED(kernel, grid, block, lds, stream, ...)                                            \
    do {                                                                                                   \
        if (fgn_prof_start) {                                                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, fgn_prof_start, fgn_prof_stop, 0, __VA_ARGS__); \
            fgn_prof_start = nullptr;                                                                      \
            fgn_prof_stop = nullptr;                                                                       \
        } else {                                                                                           \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                             \
        }                                                                                                  \
    } while (0)

// Launch records of the dominant kernel that work INSIDE a replayed hipGraph (fgn_profile_stamps): while the calling
// thread is armed, every launch of conv_pw_persist_kernel is handed the next record (FGN_STAMP_WORDS uint64) of the
// caller's device buffer; the kernel folds each of its executions into that record (first workgroup start -> last
// workgroup end on the 100 MHz s_memrealtime clock: count, sum, min, max).  nullptr: no stamp instruction executes.
constexpr int FGN_STAMP_WORDS = 72;      // uint64 per record: one 64-byte header line + eight 64-byte shard lines
extern thread_local unsigned long long* fgn_stamp_base;
extern thread_local int fgn_stamp_next, fgn_stamp_cap;
static inline unsigned long long* fgn_next_stamp_record() {
    if (!fgn_stamp_base || fgn_stamp_next >= fgn_stamp_cap) return nullptr;
    return fgn_stamp_base + FGN_STAMP_WORDS * (size_t)fgn_stamp_next++;
}

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
