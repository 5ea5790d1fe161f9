mkdir -p gpurun_out/r04s
python tools/leak_check_ckzg.py 2>&1 | grep -v amdgpu > gpurun_out/r04s/leak.txt; tail -12 gpurun_out/r04s/leak.txt
rm -f gpurun_out/stress.log
RUNS=12 HOLD=90 bash tools/stress_mirror.sh 2>&1 | tail -3 > gpurun_out/r04s/stress.txt; cat gpurun_out/r04s/stress.txt
