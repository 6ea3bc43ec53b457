// the engine's launch tracing (dev.h: GPE_LAUNCH) lives in engine.hip, which the stand-alone tools do not link
bool gpe_trace_on() { return false; }
hipEvent_t gpe_trace_event() { return nullptr; }
void gpe_trace_add(const char*, hipStream_t, hipEvent_t, hipEvent_t, dim3, dim3) {}
