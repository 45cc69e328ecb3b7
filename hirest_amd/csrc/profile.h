// Per-launch event timing hooks (see hirest_profile_* in include/hirest_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

bool hirest_prof_on();
// returns a slot id (or -1 when profiling is off) after recording the start event on `s`
int hirest_prof_begin(int kind, int tag, int64_t d0, int64_t d1, int64_t d2, hipStream_t s);
void hirest_prof_end(int slot, hipStream_t s);

struct HirestProfScope {
    int slot; hipStream_t s;
    HirestProfScope(int kind, int tag, int64_t d0, int64_t d1, int64_t d2, hipStream_t st)
        : slot(hirest_prof_on() ? hirest_prof_begin(kind, tag, d0, d1, d2, st) : -1), s(st) {}
    ~HirestProfScope() { if (slot >= 0) hirest_prof_end(slot, s); }
};
