#!/bin/bash
# r06 call 17: ONE compute_blob_kzg_proof with the commitment validated on a thread of its own, the parse beside the hashing, digest / sum / redo flag in pinned memory
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_proof_parity.py tests/test_gpu_coop.py tests/test_gpu_fuzz_seeds.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lagrange.py -x -q -m gpu -k "proof or lib_test or mirror or vectors or threads" 2>&1 | tail -3
LWKZG_DIRECT=0 python tools/single_blob_timing.py 2>&1 | tail -4 | tee $O/g17_single_blob_timing.txt
LWKZG_EXPERIMENTAL=1 LWKZG_ZERO_COPY=0 LWKZG_DIRECT=0 python tools/single_blob_timing.py 2>&1 | tail -4 | tee $O/g17_single_blob_timing_r05_arm.txt
