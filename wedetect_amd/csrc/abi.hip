// abi.hip — version / error-string entry points of libwedetect_hip.so.
#include "common.h"

extern "C" int wd_abi_version(void) { return 14; }

extern "C" int wd_sizeof_conv_gemm(void) { return (int)sizeof(WdConvGemm); }

extern "C" const char* wd_strerror(int code) {
  switch (code) {
    case WD_OK: return "ok";
    case WD_ERR_BAD_ARG: return "bad argument (shape, alignment or null pointer)";
    case WD_ERR_LAUNCH: return "HIP launch or runtime call failed";
    case WD_ERR_WORKSPACE: return "workspace too small";
    case WD_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error code";
  }
}

thread_local WdLaunchTiming wd_launch_timing = {nullptr, nullptr};

extern "C" int wd_time_next_gemm(void* start_event, void* stop_event) {
  wd_launch_timing = WdLaunchTiming{static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event)};
  return WD_OK;
}
