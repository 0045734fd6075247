// pm_tune.hpp -- the A/B knobs of the tuning build.  The shipped library reads NO environment variable: every
// schedule and LDS-budget choice is fixed by the code (results never depended on them).  `make tuning` builds
// lib/libvehicle_pm_gpu_tuning.so with -DPM_TUNING, in which the knobs named at the call sites (PM_STREAM_PRIO,
// PM_PAIR_LANES, PM_RUNBLK_*, PM_G16_*, ...) are read once per process; tools/ select that build through PM_LIB.
#pragma once

#include <cstdlib>

namespace pm {

inline const char* tune_env(const char* name) {
#ifdef PM_TUNING
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

}  // namespace pm
