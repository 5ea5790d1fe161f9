mkdir -p gpurun_out/r04h
timeout 1500 python -m pytest tests/test_gpu_lagrange.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04h/lagrange_tests.txt
tail -25 gpurun_out/r04h/lagrange_tests.txt
