// C-ABI plumbing: version, error state, hipGraph capture helpers, HIP-event timing.
#include <stdarg.h>
#include <string.h>

#include "common.hpp"
#include "../../include/mvd_hip.h"

static thread_local char g_err[512] = "";

void mvd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mvd_version(void) { return MVD_VERSION; }
extern "C" const char* mvd_last_error(void) { return g_err; }

#define MVD_HIP(call, name)                                              \
  do {                                                                   \
    hipError_t _e = (call);                                              \
    if (_e != hipSuccess) {                                              \
      mvd_set_error("%s: %s", name, hipGetErrorString(_e));              \
      return -3;                                                         \
    }                                                                    \
  } while (0)

extern "C" int mvd_graph_begin(mvd_stream_t stream) {
  MVD_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
  return 0;
}

extern "C" int mvd_graph_end(mvd_stream_t stream, void** graph_exec) {
  MVD_CHECK_ARG(graph_exec != nullptr, "mvd_graph_end: null out pointer");
  hipGraph_t graph = nullptr;
  MVD_HIP(hipStreamEndCapture((hipStream_t)stream, &graph), "hipStreamEndCapture");
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    mvd_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
    return -3;
  }
  *graph_exec = (void*)exec;
  return 0;
}

extern "C" int mvd_graph_launch(void* graph_exec, mvd_stream_t stream) {
  MVD_CHECK_ARG(graph_exec != nullptr, "mvd_graph_launch: null graph");
  MVD_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream), "hipGraphLaunch");
  return 0;
}

extern "C" int mvd_graph_destroy(void* graph_exec) {
  if (graph_exec) MVD_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec), "hipGraphExecDestroy");
  return 0;
}

extern "C" int mvd_event_create(void** ev) {
  MVD_CHECK_ARG(ev != nullptr, "mvd_event_create: null out pointer");
  hipEvent_t e;
  MVD_HIP(hipEventCreate(&e), "hipEventCreate");
  *ev = (void*)e;
  return 0;
}
extern "C" int mvd_event_record(void* ev, mvd_stream_t stream) {
  MVD_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream), "hipEventRecord");
  return 0;
}
extern "C" int mvd_event_elapsed_ms(void* start, void* stop, float* ms) {
  MVD_CHECK_ARG(ms != nullptr, "mvd_event_elapsed_ms: null out pointer");
  MVD_HIP(hipEventSynchronize((hipEvent_t)stop), "hipEventSynchronize");
  MVD_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop), "hipEventElapsedTime");
  return 0;
}
extern "C" int mvd_event_destroy(void* ev) {
  if (ev) MVD_HIP(hipEventDestroy((hipEvent_t)ev), "hipEventDestroy");
  return 0;
}
