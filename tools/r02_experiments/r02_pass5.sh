set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_proof_parity.py -x -q -k "caller_streams or sixteen or longer" -s > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-extra-legs"
$B --op blob_proof --batch 256 > $O/p256_s1.json 2> $O/err.txt
$B --op blob_proof --batch 256 --caller-streams 2 > $O/p256_s2.json 2>> $O/err.txt
$B --op blob_proof --batch 1024 > $O/p1024_s1.json 2>> $O/err.txt
$B --op blob_proof --batch 1024 --caller-streams 2 > $O/p1024_s2.json 2>> $O/err.txt
$B --op blob_proof --batch 1024 --caller-streams 3 > $O/p1024_s3.json 2>> $O/err.txt
$B --caller-streams 2 > $O/c1024_s2.json 2>> $O/err.txt
$B > $O/c1024_s1.json 2>> $O/err.txt
$B --op blob_proof --batch 256 --caller-streams 2 --direct-bits default > $O/p256_s2_default.json 2>> $O/err.txt
$B --op blob_proof --batch 256 --direct-bits default > $O/p256_s1_default.json 2>> $O/err.txt
rocprofv3 --kernel-trace --output-format csv -d $O/kt_s2 -o kt -- python3 bench.py --op blob_proof --batch 256 --caller-streams 2 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/kt_s2_line.json 2> $O/kt_s2_err.txt
