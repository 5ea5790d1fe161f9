set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "direct_commitments or batch_1024 or hand_built or adversarial" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-extra-legs"
for t in 0 8 4 16 2; do
LWKZG_DIRECT_TAIL=$t $B > $O/c_tail$t.json 2>> $O/err.txt
done
LWKZG_DIRECT_TAIL=0 $B --direct-bits default > $O/d13_tail0.json 2>> $O/err.txt
LWKZG_DIRECT_TAIL=8 $B --direct-bits default > $O/d13_tail8.json 2>> $O/err.txt
LWKZG_DIRECT_TAIL=0 $B > $O/c_tail0_again.json 2>> $O/err.txt
LWKZG_DIRECT_TAIL=8 $B > $O/c_tail8_again.json 2>> $O/err.txt
