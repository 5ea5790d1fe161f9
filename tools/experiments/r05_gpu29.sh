#!/bin/bash
# the GPU suite and the default bench line on the round's final HEAD (the collection ran three commits earlier: the one-launch evaluation of device-resident verification came after it)
O=gpurun_out/r05; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/gpu_test_log_final_head.txt 2>&1
echo "pytest rc=$?" >> $O/gpu_test_log_final_head.txt
tail -3 $O/gpu_test_log_final_head.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_final_head.json python bench.py > $O/bench_line_final_head.json 2> $O/bench_final_head_err.txt
tail -c 300 $O/bench_line_final_head.json
