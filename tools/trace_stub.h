// the engine's launch tracing (dev.h: GPE_LAUNCH) lives in engine.hip, which the stand-alone tools do not link
bool gpe_trace_on() { return false; }
hipEvent_t gpe_trace_event() { return nullptr; }
void gpe_trace_add(const char*, hipStream_t, hipEvent_t, hipEvent_t, dim3, dim3) {}
// ... and so does the gate that orders data-flow launches of several handles (dev.h: FlowGate): one handle here, nothing to order
void flow_gate_enter(hipStream_t) {}
void flow_gate_leave(hipStream_t) {}
void flow_gate_forget(hipStream_t) {}
