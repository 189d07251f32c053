// ABI version of libfgn_hip.so (bumped whenever include/fgn_hip.h changes incompatibly) and the profiler hook.
#include "common.h"
extern "C" int fgn_abi_version(void) { return 24; }

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
