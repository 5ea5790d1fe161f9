#!/bin/bash
# the ramp of a burst of 256-blob proof calls; the c-kzg proof files of the collection again on the coalesced evaluation-form kernel; the GPU suite through the host-ASan + UBSan build
python tools/experiments/r05_proof_ramp.py 16 2>/dev/null | tail -1
python tools/experiments/r05_proof_ramp.py 13 2>/dev/null | tail -1
O=gpurun_out/final; mkdir -p $O
LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b1024.json python bench.py --op blob_proof --batch 1024 --no-cpu-baseline > $O/bench_line_blob_proof_b1024.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b1024_ckzg.json python bench.py --op blob_proof --batch 1024 --mode ckzg --no-cpu-baseline > $O/bench_line_blob_proof_b1024_ckzg.json 2>> $O/bench_err.txt
LWKZG_CKZG_EVAL_PROOFS=0 LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b1024_ckzg_coefficient_arm.json python bench.py --op blob_proof --batch 1024 --mode ckzg --no-cpu-baseline > $O/bench_line_blob_proof_b1024_ckzg_coefficient_arm.json 2>> $O/bench_err.txt
export LWKZG_BENCH_DETAIL=$O/bench_detail_profiled_runs.json
PK="python3 bench.py --op blob_proof --mode ckzg --batch 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs"
rm -rf $O/kt_proof_ckzg $O/pmc_ckzg_sq1 $O/pmc_ckzg_sq2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_proof_ckzg -o kt -- $PK > $O/kt_proof_ckzg_line.json 2> $O/kt_proof_ckzg_err.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_ckzg_sq1 -o sq -- $PK > $O/pmc_ckzg_sq1_line.json 2> $O/pmc_ckzg_sq1_err.txt
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_ckzg_sq2 -o sq -- $PK > $O/pmc_ckzg_sq2_line.json 2> $O/pmc_ckzg_sq2_err.txt
unset LWKZG_BENCH_DETAIL
for f in bench_line_blob_proof_b1024 bench_line_blob_proof_b1024_ckzg bench_line_blob_proof_b1024_ckzg_coefficient_arm; do tail -1 $O/$f.json | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$f', l['value'], l['ms_per_step'], l['box'])"; done
bash tools/host_asan_gpu.sh
