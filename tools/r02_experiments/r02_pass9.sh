set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_proof_parity.py -x -q -s -k "everything or sixteen" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
grep -E "mixed concurrency|coalesced|passed|failed|rc=" $O/pytest.log
for s0 in 0 128 256; do
LWKZG_SLICE0=$s0 python bench.py --no-cpu-baseline > $O/slice$s0.json 2>> $O/err.txt
done
timeout 1500 python tools/soak.py --batches 60 --direct-bits 16 > $O/soak.json 2> $O/soak_err.txt
echo "soak rc=$?" >> $O/soak.json
LWKZG_DIRECT=16 timeout 600 python tools/soak_verify.py 150 > $O/soak_verify.jsonl 2> $O/soak_verify_err.txt
echo "soak_verify rc=$?" >> $O/soak_verify.jsonl
tail -2 $O/soak_verify.jsonl
timeout 600 python tools/soak_verify.py 100 > $O/soak_verify_default.jsonl 2>> $O/soak_verify_err.txt
tail -1 $O/soak_verify_default.jsonl
bash tools/stress_mirror.sh > $O/stress_mirror.log 2>&1
tail -3 $O/stress_mirror.log
