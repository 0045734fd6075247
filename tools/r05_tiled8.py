"""4096x2160 in 8 bands on device 0 through the C-ABI driver (bench.py's eight_bands_on_this_device leg alone)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ocean-perception_amd", "python"))
import tiled
a = argparse.Namespace(rows=720, cols=1280, iters=8, patch=11, semantics=0, engine=0, self_seed=False, backend="nccl",
                       dry_run=False)
bands = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = tiled.bench_single_process(a, [0] * bands, steps=3)
print(json.dumps({k: r[k] for k in ("ms_per_frame", "bands", "boundary_rows_moved_per_match", "foreground_within_1px")}))
