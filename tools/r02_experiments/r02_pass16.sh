#!/bin/bash
# single-call latency: per-kernel clock and a kernel-trace timeline of lone blob_to_kzg_commitment calls
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p16; export TMPDIR=/tmp
for b in 13 16 0; do
  LWKZG_DIRECT=$b timeout 300 python tools/single_blob_timing.py > gpurun_out/p16/single_$b.txt 2>&1
done
cat > /tmp/one.py <<'PY'
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
blob = B.synthetic_blob(1)
for _ in range(30):
    K.blob_to_kzg_commitment(blob, ts)
PY
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/p16/kt -o kt --output-format csv -- python3 /tmp/one.py > $GRAFT_REPO_ROOT/gpurun_out/p16/kt.log 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/p16/kt -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/timeline.py {} 24 > gpurun_out/p16/timeline.txt 2>&1
find gpurun_out/p16/kt -name "*memory_copy_trace.csv" | head -1 | xargs -I{} tail -12 {} > gpurun_out/p16/copies.txt 2>&1
rm -rf gpurun_out/p16/kt
cat gpurun_out/p16/single_13.txt gpurun_out/p16/single_16.txt gpurun_out/p16/single_0.txt; cat gpurun_out/p16/timeline.txt
