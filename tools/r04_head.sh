#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest --tb=short tests/test_gpu_parity.py tests/test_cpp_wrapper.py tests/test_seed.py -x -q -m gpu > gpurun_out/r04/head_tests.log 2>&1 || { tail -40 gpurun_out/r04/head_tests.log; exit 1; }
tail -2 gpurun_out/r04/head_tests.log
timeout -k 10 600 python tools/fuzz_api.py --cases 80 --seed 77 > gpurun_out/r04/head_fuzz.log 2>&1 || { tail -5 gpurun_out/r04/head_fuzz.log; exit 1; }
tail -1 gpurun_out/r04/head_fuzz.log
timeout -k 10 600 python tools/fuzz_selfseed.py --cases 40 --seed 78 > gpurun_out/r04/head_fuzz2.log 2>&1 || { tail -5 gpurun_out/r04/head_fuzz2.log; exit 1; }
tail -1 gpurun_out/r04/head_fuzz2.log
bash tools/r04_bench.sh
