python -m pytest tests/test_gpu_post.py tests/test_gpu_full_size.py tests/test_gpu_randomized.py -x -q -m gpu 2>&1 | tail -2
python tools/knn_bench.py 2>&1 | tail -4
python tools/topk_full.py --dims 256 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin)['d256']; print('topk 1Mx1M', d['seconds'], d['executed_f16_mfma_tflops'], d['spot_check_64_queries_equal_exact_path'], d['recomputed_by_f64_path'])"
