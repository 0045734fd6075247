// pm_internal.hpp -- what the other translation units of libvehicle_pm_gpu.so may use of a pm_handle
// (defined in pm_engine.hip).  Not part of the public ABI.
#pragma once

#include <hip/hip_runtime.h>

#include "pm/patchmatch.h"

namespace pm {
struct BgrSource;
}

namespace pm_internal __attribute__((visibility("hidden"))) {
int device(const pm_handle* h);
hipStream_t stream(pm_handle* h);
const pm_params& params(const pm_handle* h);
void plan_size(const pm_handle* h, int* max_rows, int* max_cols);
void set_error(pm_handle* h, const char* fmt, ...);
void** imaging_slot(pm_handle* h);     // storage for pm_imaging.hip's state
// pm_match_bgr_device: while non-null, pm_match_device's prep stage reads this source (pm_kernels.hpp::k_prep_bgr)
void set_bgr_source(pm_handle* h, const pm::BgrSource* src);
void release_imaging(pm_handle* h);    // defined in pm_imaging.hip, called by pm_destroy
}  // namespace pm_internal
