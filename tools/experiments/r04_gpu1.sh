set -x
mkdir -p gpurun_out/r04a
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04a/gpu_tests.txt
# the pre-fix library must FAIL the single-collision tests (it is the proof that they test what they say)
LWKZG_LIBRARY=$PWD/lambdaworks_kzg_amd/lib_prefix/liblambdaworks_kzg.so python -m pytest tests/test_gpu_parity.py -m gpu -q -k "colliding" 2>&1 | tail -25 > gpurun_out/r04a/prefix_collision.txt
python bench.py > gpurun_out/r04a/bench_line.json 2> gpurun_out/r04a/bench_err.txt
wc -c gpurun_out/r04a/bench_line.json
cp bench_detail.json gpurun_out/r04a/bench_detail.json
tail -3 gpurun_out/r04a/gpu_tests.txt; tail -5 gpurun_out/r04a/prefix_collision.txt
