set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_proof_parity.py tests/test_gpu_zz_env.py -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-extra-legs --op blob_proof"
$B --batch 256 > $O/p256.json 2> $O/err.txt
$B --batch 1024 > $O/p1024.json 2>> $O/err.txt
$B --batch 256 --caller-streams 2 > $O/p256_s2.json 2>> $O/err.txt
$B --batch 1024 --caller-streams 2 > $O/p1024_s2.json 2>> $O/err.txt
