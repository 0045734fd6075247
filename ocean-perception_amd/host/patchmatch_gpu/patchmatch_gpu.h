// The reference's include line -- #include "patchmatch_gpu/patchmatch_gpu.h"
// (test/stereo_matching/patchmatch_gpu_test.cpp:13; the header it names is src/vehicle/patchmatch_gpu/patchmatch_gpu.h) --
// resolves here when ocean-perception_amd/host is on the include path (the CMake target vehicle_pm_gpu puts it there).
#pragma once
#include "../patchmatch_gpu.hpp"
