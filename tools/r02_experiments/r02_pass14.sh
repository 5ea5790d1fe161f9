set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02t
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -s --durations=8 > $O/gpu_test_log.txt 2>&1
echo "pytest rc=$?" >> $O/gpu_test_log.txt
grep -E "coalesced|mixed concurrency|passed|failed|rc=" $O/gpu_test_log.txt
LWKZG_TEST_STRESS_SECONDS=60 timeout 900 python -m pytest tests/test_gpu_proof_parity.py -x -q -s -k "everything" > $O/stress.log 2>&1
echo "rc=$?" >> $O/stress.log
grep -E "mixed concurrency|passed|failed|rc=" $O/stress.log
python tools/host_api_timing.py > $O/host_api_timing.txt 2>&1
