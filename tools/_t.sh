#!/bin/bash
timeout -k 10 600 python3 -m pytest tests/test_gpu_sparse.py tests/test_gpu_real_clouds.py tests/test_gpu_randomized.py tests/test_gpu_edge_cases.py tests/test_gpu_harness.py tests/test_gpu_shim.py tests/test_gpu_next_rows.py -x -q > gpurun_out/r11_test.log 2>&1 || { tail -30 gpurun_out/r11_test.log; exit 1; }
tail -1 gpurun_out/r11_test.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r11_prof -- python3 tools/conv_layers.py 64 15000 0.02 > /dev/null 2>&1
python3 tools/kernel_stats.py gpurun_out/r11_prof 60 2>/dev/null | grep -i "segmax"
rm -rf gpurun_out/r11_prof
