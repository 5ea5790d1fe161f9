# round 2, pass 2: new dist / shard tests, the self-sufficient bench line, tail experiments on the direct kernel
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_dist.py -x -q --durations=10 > $O/pytest_dist.log 2>&1
echo "pytest rc=$?" >> $O/pytest_dist.log
tail -15 $O/pytest_dist.log
python bench.py > $O/bench_line.json 2> $O/bench_err.txt
# tail experiment: two / four workgroups per blob at 1024 blobs (more, shorter workgroups; + a fold launch)
LWKZG_DIRECT_FILL=2048 python bench.py --no-cpu-baseline --no-extra-legs > $O/bench_fill2048.json 2>> $O/bench_err.txt
LWKZG_DIRECT_FILL=4096 python bench.py --no-cpu-baseline --no-extra-legs > $O/bench_fill4096.json 2>> $O/bench_err.txt
python bench.py --no-cpu-baseline --no-extra-legs > $O/bench_fill512.json 2>> $O/bench_err.txt
python bench.py --op verify_batch --batch 4096 --steps 5 --no-cpu-baseline > $O/bench_verify4096.json 2>> $O/bench_err.txt
python bench.py --op blob_proof --batch 1024 --no-cpu-baseline > $O/bench_proof_b1024.json 2>> $O/bench_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/kt_line.json 2> $O/kt_err.txt
du -sh $O
