#!/bin/bash
# first contact of the cooperative kernel: smoke, small-batch parity tests, single-blob timing (A/B against LWKZG_COOP=0)
mkdir -p gpurun_out/r05
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.txt 2>&1; echo "smoke rc=$?" 
tail -3 gpurun_out/r05/smoke.txt
for rpq in 2 4; do
  LWKZG_COOP_RPQ=$rpq python tools/single_blob_timing.py > gpurun_out/r05/single_blob_rpq$rpq.txt 2>&1; cat gpurun_out/r05/single_blob_rpq$rpq.txt
  LWKZG_DIRECT=13 LWKZG_COOP_RPQ=$rpq python tools/single_blob_timing.py > gpurun_out/r05/single_blob_13_rpq$rpq.txt 2>&1; cat gpurun_out/r05/single_blob_13_rpq$rpq.txt
done
LWKZG_COOP=0 python tools/single_blob_timing.py > gpurun_out/r05/single_blob_nocoop.txt 2>&1; cat gpurun_out/r05/single_blob_nocoop.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not 1024 and not 4096" > gpurun_out/r05/parity_small.txt 2>&1; tail -5 gpurun_out/r05/parity_small.txt
