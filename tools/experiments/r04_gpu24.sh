set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
LWKZG_BENCH_DETAIL=$O/bench_detail.json python bench.py > $O/bench_line.json 2> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b256.json python bench.py --op blob_proof --batch 256 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_line_blob_proof_b256.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_blob_proof_b256_two_streams.json python bench.py --op blob_proof --batch 256 --caller-streams 2 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_line_blob_proof_b256_two_streams.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_commit_prove_b256.json python bench.py --op commit_prove --batch 256 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_line_commit_prove_b256.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_bucket_compiler_arm.json LWKZG_BUCKET_ASM=0 python bench.py --direct-bits 0 --no-cpu-baseline --no-config-legs > $O/bench_line_bucket_compiler_arm.json 2>> $O/bench_err.txt
LWKZG_BENCH_DETAIL=$O/bench_detail_bucket.json python bench.py --direct-bits 0 --no-cpu-baseline --no-config-legs > $O/bench_line_bucket.json 2>> $O/bench_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/kt_line.json 2> $O/kt_err.txt
for f in bench_line bench_line_blob_proof_b256 bench_line_blob_proof_b256_two_streams bench_line_commit_prove_b256 bench_line_bucket bench_line_bucket_compiler_arm; do python3 -c "
import json; l=json.load(open('$O/$f.json')); print('$f', l['value'], l['ms_per_step'], l['roofline']['kernel'], l['roofline']['avg_launch_ms'])"; done
