# round 2, pass 1: the GPU test suite on the refactored engine, the default bench line, and SQ/GRBM counter passes
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench_err.txt
python bench.py --direct-bits 0 --no-cpu-baseline > $O/bench_line_bucket.json 2>> $O/bench_err.txt
python bench.py --direct-bits 13 --no-cpu-baseline > $O/bench_line_d13.json 2>> $O/bench_err.txt
python bench.py --op blob_proof --batch 256 --no-cpu-baseline > $O/bench_proof_b256.json 2>> $O/bench_err.txt
python bench.py --op blob_proof --batch 4096 --steps 5 --no-cpu-baseline > $O/bench_proof_b4096.json 2>> $O/bench_err.txt
# counters: separate passes, --pmc only (no trace flags)
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_sq1 -o sq1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq1_line.json 2> $O/pmc_sq1_err.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq2 -o sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq2_line.json 2> $O/pmc_sq2_err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/pmc_grbm -o grbm -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_grbm_line.json 2> $O/pmc_grbm_err.txt
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $O/pmc_sq3 -o sq3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq3_line.json 2> $O/pmc_sq3_err.txt
rocprofv3 -L > $O/counters_list.txt 2>&1
find $O -name "*.csv" | head -30
du -sh $O
