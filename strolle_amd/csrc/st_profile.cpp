// st_profile.cpp — host engine of libstrolle_hip.so: per-kernel event timing (st_profile_enable / st_profile_read). See st_engine.h.
#include "st_engine.h"

namespace st {

// ---- profiling
hipEvent_t Engine::take_event() {
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}

void Engine::profile_begin(int slot, hipStream_t s, double bytes) {
    if (!profiling) return;
    if (open_scope.slot == slot && open_scope.stream == s) { open_scope.bytes += bytes; open_scope.launches += 1; return; }
    const bool chained = open_scope.slot >= 0 && open_scope.stream == s;
    hipEvent_t boundary = profile_close();
    open_scope.slot = slot; open_scope.stream = s; open_scope.bytes = bytes; open_scope.launches = 1;
    if (chained) { open_scope.start = boundary; open_scope.owns_start = false; }
    else { open_scope.start = take_event(); open_scope.owns_start = true; (void)hipEventRecord(open_scope.start, s); }
}

hipEvent_t Engine::profile_close() {
    if (open_scope.slot < 0) return nullptr;
    hipEvent_t stop = take_event();
    (void)hipEventRecord(stop, open_scope.stream);
    profile_records.push_back({open_scope.slot, open_scope.start, stop, open_scope.bytes, open_scope.launches, open_scope.owns_start});
    open_scope.slot = -1;
    return stop;
}

int Engine::drain_profile() {
    for (auto& r : profile_records) {
        ST_HIP(hipEventSynchronize(r.stop));
        float ms = 0.0f;
        ST_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
        profile_totals[r.slot].launches += r.launches; profile_totals[r.slot].total_ms += ms; profile_totals[r.slot].algorithmic_bytes += r.bytes;
        if (r.owns_start) event_pool.push_back(r.start);
        event_pool.push_back(r.stop);
    }
    profile_records.clear();
    return ST_OK;
}

}  // namespace st
