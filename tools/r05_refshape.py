"""The reference_test_shape leg of bench.py alone (376x240, patchmatch_gpu_test.cpp:68-88), with and without the
small-image graph; prints the leg's JSON."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import bench
import pm_ctypes as pm
class A: no_cpu_baseline = True
print(json.dumps(bench.reference_test_shape_leg(pm, A(), 0)))
