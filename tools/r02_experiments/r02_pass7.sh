set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-extra-legs --op blob_proof"
$B --batch 256 --caller-streams 2 > $O/p256_s2.json 2> $O/err.txt
$B --batch 1024 --caller-streams 2 > $O/p1024_s2.json 2>> $O/err.txt
$B --batch 256 > $O/p256_s1.json 2>> $O/err.txt
$B --batch 256 --caller-streams 2 --direct-bits default > $O/p256_s2_default.json 2>> $O/err.txt
$B --batch 1024 --caller-streams 2 --direct-bits default > $O/p1024_s2_default.json 2>> $O/err.txt
python bench.py --no-cpu-baseline --no-extra-legs --caller-streams 2 > $O/c1024_s2.json 2>> $O/err.txt
