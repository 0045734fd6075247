/* Test hooks of libvehicle_pm_gpu.so -- NOT part of the supported C ABI (include/pm/patchmatch.h): entry points that
 * exist only so that tests/ can drive the library into states a well-behaved caller never produces.  They have no
 * counterpart in the reference (src/vehicle/patchmatch_gpu/patchmatch_gpu.h:77-124) and may change or vanish between
 * ABI versions; the C++ wrapper (host/patchmatch_gpu.hpp) does not use them. */
#ifndef PM_TESTING_H_
#define PM_TESTING_H_
#include "pm/patchmatch.h"
#ifdef __cplusplus
extern "C" {
#endif
/* forks an empty dependency onto an internal stream of an open capture and leaves it unjoined, so that the guard in
 * pm_capture_end (PM_ERR_STATE instead of a fault inside the runtime) can be exercised */
int pm_debug_capture_fork(pm_handle* h);
#ifdef __cplusplus
}
#endif
#endif /* PM_TESTING_H_ */
