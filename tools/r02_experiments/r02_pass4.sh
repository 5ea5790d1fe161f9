set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
B="python bench.py --no-cpu-baseline --no-extra-legs"
$B --idle-ms 0 > $O/c_idle0.json 2> $O/err.txt
$B --idle-ms 1 > $O/c_idle1.json 2>> $O/err.txt
$B --idle-ms 4 > $O/c_idle4.json 2>> $O/err.txt
$B --idle-ms 20 > $O/c_idle20.json 2>> $O/err.txt
LWKZG_KEEPWARM=256,20000 $B --op blob_proof --batch 1024 > $O/p1024_kw20k.json 2>> $O/err.txt
LWKZG_KEEPWARM=256,40000 $B --op blob_proof --batch 1024 > $O/p1024_kw40k.json 2>> $O/err.txt
LWKZG_KEEPWARM=256,55000 $B --op blob_proof --batch 1024 > $O/p1024_kw55k.json 2>> $O/err.txt
