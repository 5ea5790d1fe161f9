set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_mid
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x --durations=10 > $O/gpu_test_log.txt 2>&1
echo "pytest rc=$?" >> $O/gpu_test_log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_full -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_full_line.json 2> $O/kt_full_err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_full -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch_full_line.json 2> $O/fetch_full_err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_full -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write_full_line.json 2> $O/write_full_err.txt
rm -f $O/kt_full/*kernel_trace.csv
du -sh $O
tail -5 $O/gpu_test_log.txt
