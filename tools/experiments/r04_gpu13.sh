mkdir -p gpurun_out/r04m
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_setups_unstructured.py tests/test_gpu_proof_parity.py tests/test_gpu_setups.py tests/test_gpu_lagrange.py -m gpu -x -v -k "bucket or engine or adversarial or repairs" > gpurun_out/r04m/log.txt 2>&1
grep -n "PASSED\|FAILED\|Fatal\|fault\|Memory access\|test_" gpurun_out/r04m/log.txt | tail -25
