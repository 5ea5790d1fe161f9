set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02soak
mkdir -p $O
cd $R
timeout 2400 python tools/soak.py --batches 2000 --direct-bits 16 > $O/soak.json 2> $O/soak_err.txt
echo "{\"soak_rc\": $?}" >> $O/soak.json
LWKZG_DIRECT=16 timeout 500 python tools/soak_verify.py 300 2> $O/soak_verify_err.txt | tail -1 > $O/soak_verify_direct16.json
timeout 500 python tools/soak_verify.py 300 2>> $O/soak_verify_err.txt | tail -1 > $O/soak_verify_default.json
LWKZG_DIRECT_BITS=0 timeout 500 python tools/soak_verify.py 200 2>> $O/soak_verify_err.txt | tail -1 > $O/soak_verify_bucket.json
cat $O/*.json
LWKZG_TEST_STRESS_SECONDS=90 timeout 900 python -m pytest tests/test_gpu_proof_parity.py -x -q -s -k "everything" > $O/stress90.log 2>&1
echo "rc=$?" >> $O/stress90.log
grep -E "mixed concurrency|passed|failed|rc=" $O/stress90.log
