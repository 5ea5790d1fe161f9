set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02stress
mkdir -p $O
cd $R
LWKZG_TEST_STRESS_SECONDS=120 timeout 900 python -m pytest tests/test_gpu_proof_parity.py -x -q -s -k "everything" > $O/stress.log 2>&1
echo "rc=$?" >> $O/stress.log
grep -E "mixed concurrency|passed|failed|rc=" $O/stress.log
