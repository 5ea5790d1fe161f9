set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01b
mkdir -p $O
cd $R
python bench.py > $O/bench_line.json 2> $O/bench_err.txt
python bench.py --direct-bits 0 --no-cpu-baseline > $O/bench_line_bucket_path.json 2>> $O/bench_err.txt
python bench.py --mode ckzg --no-cpu-baseline > $O/bench_line_ckzg_mode.json 2>> $O/bench_err.txt
python bench.py --op blob_proof --batch 256 --no-cpu-baseline > $O/bench_line_blob_proof_b256.json 2>> $O/bench_err.txt
python bench.py --op blob_proof --batch 1024 --no-cpu-baseline > $O/bench_line_blob_proof_b1024.json 2>> $O/bench_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/kt_line.json 2> $O/kt_err.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch_line.json 2> $O/fetch_err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write_line.json 2> $O/write_err.txt
python tools/host_api_timing.py > $O/host_api_timing.txt 2>&1
python tools/config_sweep.py --direct-bits 16 > $O/config_sweep_direct16.json 2> $O/sweep_err.txt
python tools/config_sweep.py --direct-bits 0 --max-verify 1024 > $O/config_sweep_bucket.json 2>> $O/sweep_err.txt
python tools/verify_timing.py > $O/verify_timing.txt 2>&1
LWKZG_DIRECT=16 python tools/host_api_timing.py > $O/host_api_timing_direct.txt 2>&1
find $O -name "*.csv" | head -30
du -sh $O
