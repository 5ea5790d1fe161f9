set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02s
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-extra-legs --op blob_proof"
$B --batch 1024 > $O/p1024.json 2> $O/err.txt
for cfg in 256,500 256,1000 256,2000 512,2000 64,4000; do
LWKZG_MEMWARM=$cfg $B --batch 1024 > $O/p1024_mw_${cfg/,/_}.json 2>> $O/err.txt
done
$B --batch 256 > $O/p256.json 2>> $O/err.txt
LWKZG_MEMWARM=256,1000 $B --batch 256 > $O/p256_mw_256_1000.json 2>> $O/err.txt
