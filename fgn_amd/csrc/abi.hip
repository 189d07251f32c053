// ABI version of libfgn_hip.so (bumped whenever include/fgn_hip.h changes incompatibly) and the profiler hook.
#include "common.h"
extern "C" int fgn_abi_version(void) { return 29; }

thread_local hipEvent_t fgn_prof_start = nullptr;
thread_local hipEvent_t fgn_prof_stop = nullptr;

// Arm the calling thread: the next convolution-family kernel launched from it stamps `start_event` / `stop_event`
// (created by the caller, e.g. torch.cuda.Event(enable_timing=True) after a first record()) with its own start
// and end.  Passing NULL disarms.
extern "C" int fgn_profile_next_launch(void* start_event, void* stop_event) {
    fgn_prof_start = reinterpret_cast<hipEvent_t>(start_event);
    fgn_prof_stop = reinterpret_cast<hipEvent_t>(stop_event);
    return FGN_OK;
}

thread_local unsigned long long* fgn_stamp_base = nullptr;
thread_local int fgn_stamp_next = 0, fgn_stamp_cap = 0;
// Arm (records != NULL) / disarm the calling thread for launch records (common.h).  `records`: device memory, `capacity`
// records of fgn_profile_stamp_words() x uint64, all zero except word 4 = ~0: {start of the execution in flight, sum of
// spans, shards arrived, executions, shortest, longest, -, -, then eight shard counters on lines of their own} in 10 ns
// ticks.  Returns the number of records handed out since the last arming (the launches recorded, in launch order).
extern "C" int fgn_profile_stamp_words(void) { return FGN_STAMP_WORDS; }
extern "C" int fgn_profile_stamps(void* records, int capacity) {
    const int used = fgn_stamp_next;
    fgn_stamp_base = reinterpret_cast<unsigned long long*>(records);
    fgn_stamp_cap = records ? capacity : 0;
    fgn_stamp_next = 0;
    return used;
}

// ------------------------------------------------------------------------------------------------
// Phase marks between streams that replay captured graphs (a captured graph cannot record an event another stream can
// wait for: external events are not available on this stack).  A one-thread kernel inside the episode bumps a counter
// in device memory; the other stream, before ITS next episode, runs a one-wave kernel that sleeps until the counter has
// reached `target` - or until `timeout_us` have passed (it then gives up: a missing signal costs the phase lock, never
// the process).  The waiting wave holds one wave slot of one CU and issues `s_sleep` between polls.
// ------------------------------------------------------------------------------------------------
__global__ void phase_signal_kernel(int32_t* counter) {
    // relaxed on purpose: the mark is a timing gate, nobody consumes data through it - a release at agent scope would
    // write back the XCD's L2 in the middle of the episode
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void phase_wait_kernel(const int32_t* counter, int32_t target, unsigned timeout_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(64);
    }
}
extern "C" int fgn_phase_signal(int32_t* counter, hipStream_t stream) {
    if (!counter) return FGN_ERR_ARG;
    hipLaunchKernelGGL(phase_signal_kernel, dim3(1), dim3(1), 0, stream, counter);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
extern "C" int fgn_phase_wait(const int32_t* counter, int32_t target, int timeout_us, hipStream_t stream) {
    if (!counter || timeout_us < 0 || timeout_us > 1000000) return FGN_ERR_ARG;
    hipLaunchKernelGGL(phase_wait_kernel, dim3(1), dim3(64), 0, stream, counter, target, (unsigned)timeout_us * 100u);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
