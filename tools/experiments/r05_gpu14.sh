#!/bin/bash
# where the small (all-host) proof path should hand over to the pipelined mid path now that the GPU validates in 1.1 ms; c-kzg proofs against reference mode
bp() { python bench.py --op blob_proof --batch $1 --steps 40 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$3 n=$1:', l['value'], l['ms_per_step'])"; }
for n in 8 16 32 64 128; do
  LWKZG_SMALL_PROOF_HOST=128 bp $n "" "small-host-path"
  LWKZG_SMALL_PROOF_HOST=4 LWKZG_MID_PROOF_PIPE_MIN=100000 bp $n "" "mid-unpiped"
  LWKZG_SMALL_PROOF_HOST=4 LWKZG_MID_PROOF_PIPE_MIN=8 bp $n "" "mid-piped"
done
for parts in 2 4; do LWKZG_MID_PROOF_PARTS=$parts bp 384 "" "parts$parts"; done
for n in 256 1024; do
  bp $n "--mode reference" "reference16"
  bp $n "--mode ckzg" "ckzg16"
  bp $n "--mode reference --direct-bits 13" "reference13"
  bp $n "--mode ckzg --direct-bits 13" "ckzg13"
done
