set -x
mkdir -p gpurun_out/r04b
python -m pytest tests/test_gpu_multi.py tests/test_gpu_host_api_extras.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04b/new_tests.txt
tail -5 gpurun_out/r04b/new_tests.txt
LWKZG_DIRECT=0 python tools/single_blob_timing.py > gpurun_out/r04b/single_default.txt 2>&1
LWKZG_DIRECT=16 python tools/single_blob_timing.py > gpurun_out/r04b/single_16.txt 2>&1
cat gpurun_out/r04b/single_default.txt gpurun_out/r04b/single_16.txt
