// Version / error strings of librsx.so.
#include "rsx.h"

extern "C" int rsx_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* rsx_strerror(int status) {
  switch (status) {
    case RSX_OK: return "ok";
    case RSX_EINVAL: return "invalid argument";
    case RSX_ELAUNCH: return "kernel launch failed (hipGetLastError)";
    case RSX_EUNSUPPORTED: return "request outside the implemented envelope";
    case RSX_EDATA: return "corrupt input data";
    case RSX_ECOMM: return "collective library call failed (see rsx_comm_last_error_h)";
    default: return "unknown rsx status";
  }
}

#include "rsx_launch_count.h"
extern "C" unsigned long long rsx_dbg_launch_count(void) { return rsx_launches_g.load(std::memory_order_relaxed); }
