// pm_sweeps.hpp -- the one entry point of the directional-sweep kernels (pm_sweeps.hip: serial anchor, wave engine,
// run engines of pm_run2.hpp / pm_run3.hpp).  They live in their own translation unit: most of the library's compile
// time is their template instantiations.
#pragma once

#include "pm_sweep_defs.hpp"

namespace pm {

// One directional sweep of every chain of `slots` slots, in place, on `stream`.  engine = pm_params.engine
// (PM_ENGINE_*); amp = the noise amplitude of the iteration (tuning only: it selects the lanes per chain segment).
void launch_sweep(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int engine, float amp,
                  hipStream_t stream);

}  // namespace pm
