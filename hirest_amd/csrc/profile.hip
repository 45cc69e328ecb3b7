// Event-pair timing of individual kernel launches, used by bench.py for the live roofline figure.
#include "common.h"
#include "profile.h"
#include <vector>

namespace {
struct Slot { hipEvent_t a, b; hirest_prof_record r; };
bool g_on = false;
std::vector<Slot> g_slots;
size_t g_used = 0;
}

bool hirest_prof_on() { return g_on; }

int hirest_prof_begin(int kind, int tag, int64_t d0, int64_t d1, int64_t d2, hipStream_t s) {
    if (!g_on) return -1;
    if (g_used == g_slots.size()) {
        Slot sl;
        if (hipEventCreate(&sl.a) != hipSuccess || hipEventCreate(&sl.b) != hipSuccess) return -1;
        g_slots.push_back(sl);
    }
    Slot& sl = g_slots[g_used];
    sl.r.kind = kind; sl.r.tag = tag; sl.r.d0 = d0; sl.r.d1 = d1; sl.r.d2 = d2; sl.r.ms = 0.f;
    (void)hipEventRecord(sl.a, s);
    return (int)g_used++;
}

void hirest_prof_end(int slot, hipStream_t s) {
    if (slot >= 0 && (size_t)slot < g_slots.size()) (void)hipEventRecord(g_slots[slot].b, s);
}

extern "C" int hirest_profile_enable(int32_t on) {
    g_on = on != 0;
    g_used = 0;
    return 0;
}

extern "C" int hirest_profile_collect(hirest_prof_record* out, int32_t max_records) {
    if (!out || max_records < 0) return HIREST_E_BADARG;
    int n = 0;
    for (size_t i = 0; i < g_used && n < max_records; ++i) {
        Slot& sl = g_slots[i];
        if (hipEventSynchronize(sl.b) != hipSuccess) return -100;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sl.a, sl.b) != hipSuccess) return -101;
        sl.r.ms = ms;
        out[n++] = sl.r;
    }
    g_used = 0;
    return n;
}
