// The library's kernel-launch counter (host only; see rsx_common.h RSX_LAUNCH and rsx_dbg_launch_count in api.cpp).
#pragma once
#include <atomic>
inline std::atomic<unsigned long long> rsx_launches_g{0};
