set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_test_log.txt 2>&1
echo "pytest rc=$?" >> $O/gpu_test_log.txt
tail -4 $O/gpu_test_log.txt
python bench.py > $O/bench_line.json 2> $O/bench_err.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -1 $O/smoke.txt
