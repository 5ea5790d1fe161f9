# round 2, pass 3: whole GPU suite with the coalescing front; how much of the direct kernel's time is gather latency
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 -s > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "coalesced single-blob|passed|failed|rc=" $O/pytest_gpu.log | tail
B="python bench.py --no-cpu-baseline --no-extra-legs"
$B > $O/commit_31.json 2> $O/err.txt
$B --scalars full > $O/commit_full.json 2>> $O/err.txt
LWKZG_LIBRARY=$R/lambdaworks_kzg_amd/lib_mask/liblambdaworks_kzg.so $B > $O/mask_31.json 2>> $O/err.txt
LWKZG_LIBRARY=$R/lambdaworks_kzg_amd/lib_mask/liblambdaworks_kzg.so $B --scalars full > $O/mask_full.json 2>> $O/err.txt
$B --direct-bits 13 --scalars full > $O/d13_full.json 2>> $O/err.txt
$B --direct-bits 0 --scalars full > $O/bucket_full.json 2>> $O/err.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_full -o sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --scalars full > $O/pmc_full_line.json 2> $O/pmc_full_err.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_full2 -o sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --scalars full > $O/pmc_full2_line.json 2> $O/pmc_full2_err.txt
du -sh $O
