export LWKZG_EXPERIMENTAL=1   # the A/B arms below are experiment knobs (csrc/knobs.h, r06)
cd $GRAFT_REPO_ROOT
L=$PWD/lambdaworks_kzg_amd/lib
gcc -std=c11 -O1 -I include tests/lib_test_mirror.c -o /tmp/mirror -L $L -llambdaworks_kzg -Wl,-rpath,$L -Wl,-rpath,/opt/rocm/lib || exit 1
# a second process holding a context with a workspace, like the pytest parent
python3 -c "
import sys,time; sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
import lambdaworks_kzg_amd as K, blobs as B
ts=K.TrustedSetup.from_file('tests/golden/trusted_setup.txt'); ts.reserve(1024)
K.blob_to_kzg_commitment_batch(B.synthetic_batch(0,4), ts)
time.sleep(int(__import__('os').environ.get('HOLD','60')))" &
PID=$!
sleep 8
fail=0; slow=0
for i in $(seq 1 ${RUNS:-12}); do
  s=$(date +%s.%N)
  MIRROR_TRACE=1 LWKZG_VERBOSE=1 timeout 20 /tmp/mirror tests/golden/trusted_setup.txt > /tmp/m.out 2>&1; rc=$?
  e=$(date +%s.%N)
  dt=$(python3 -c "print(round($e-$s,2))")
  echo "run $i rc=$rc dt=$dt" | tee -a gpurun_out/stress.log; if [ $rc -ne 0 ]; then fail=$((fail+1)); tail -3 /tmp/m.out | tee -a gpurun_out/stress.log; fi
  if python3 -c "import sys; sys.exit(0 if $dt > 5 else 1)"; then slow=$((slow+1)); echo "run $i slow dt=$dt"; fi
done
echo "fails=$fail slow=$slow"
kill $PID 2>/dev/null
wait $PID 2>/dev/null
