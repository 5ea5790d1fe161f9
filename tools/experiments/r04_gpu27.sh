mkdir -p gpurun_out/r04r
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r04r/gpu_tests.txt
cat gpurun_out/r04r/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1
(time python bench.py > gpurun_out/r04r/line.json 2> gpurun_out/r04r/err.txt) 2>&1 | grep real
wc -c gpurun_out/r04r/line.json
python -c "
import json; l=json.load(open('gpurun_out/r04r/line.json')); print(l['value'], l['roofline']['frac'], l['roofline']['int_mad']['frac_of_theoretical'], l['cpu_baseline']['value'], l['cpu_baseline']['gpu_outputs_match_oracle'], l['bucket_engine']['kernel'], l['bucket_engine']['kernel_ms'])"
