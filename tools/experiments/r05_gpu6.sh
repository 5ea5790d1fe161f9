#!/bin/bash
# in-place table rebuild on a mode switch (16 bits), device verification without per-chunk syncs
python - <<'PY'
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
t0 = time.perf_counter(); ts.enable_direct_table(16); print("enable 16 bits: %.2f s" % (time.perf_counter() - t0), ts.timing_report().get("last_table_build"))
be = B.synthetic_blob(5); le = B.synthetic_blob(5, big_endian=False)
c_ref = K.blob_to_kzg_commitment(be, ts)
for rep in range(2):
    t0 = time.perf_counter(); ts.set_mode(K.MODE_CKZG); t1 = time.perf_counter() - t0
    print("set_mode ckzg: %.2f s forms=%d" % (t1, ts.direct_table_forms()), ts.timing_report().get("last_table_build"))
    c_le = K.blob_to_kzg_commitment(le, ts)
    t0 = time.perf_counter(); ts.set_mode(K.MODE_REFERENCE); t1 = time.perf_counter() - t0
    print("set_mode reference: %.2f s forms=%d" % (t1, ts.direct_table_forms()), ts.timing_report().get("last_table_build"))
    assert K.blob_to_kzg_commitment(be, ts) == c_ref
ts.free()
PY
timeout 900 python -m pytest tests/test_gpu_lagrange.py tests/test_gpu_verify_device.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print('value',l['value'])
for k in ('verify_batch_b4096','verify_batch_b4096_device','ckzg_commit_b1024_lagrange','blob_proof_b256'):
    print(k, l['configs'][k])
"
python -c "
import json
d=json.load(open('bench_detail.json'))
print('set_mode_s', d['configs']['ckzg_commit_b1024_lagrange'].get('settings_set_mode_s'))
"
