python -m pytest tests/test_gpu_sparse.py tests/test_gpu_shim.py -x -q -m gpu > gpurun_out/r3j_tests.log 2>&1 || { tail -30 gpurun_out/r3j_tests.log; exit 1; }
tail -2 gpurun_out/r3j_tests.log
python tools/conv_layers.py 64 15000 0.02 > gpurun_out/r3j_layers_new_64.txt 2>&1; tail -2 gpurun_out/r3j_layers_new_64.txt
python tools/conv_layers.py 32 10000 0.03 > gpurun_out/r3j_layers_new_32.txt 2>&1; tail -2 gpurun_out/r3j_layers_new_32.txt
CS_CONV_TRACE=1 python tools/conv_layers.py 64 15000 0.02 2>&1 | grep "conv trace" | tail -22 | sed 's/.*cfg/cfg/' | sort | uniq -c > gpurun_out/r3j_trace.txt
