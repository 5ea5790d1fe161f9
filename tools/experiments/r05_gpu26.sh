#!/bin/bash
# what is the one call of ~10 ms among a process's first proof calls? HIP API + kernel + copy trace of tools/experiments/r05_proof_cold.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/trace_cold; rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $O -o t -- python3 tools/experiments/r05_proof_cold.py > $O/out.txt 2> $O/err.txt
tail -1 $O/out.txt
ls $O
python3 - <<'PY'
import csv, glob, os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r05/trace_cold'
f=glob.glob(O+'/*hip_api_trace.csv')
if f:
    rows=list(csv.DictReader(open(f[0])))
    rows=[r for r in rows if r.get('Function')]
    for r in rows: r['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    t0=min(int(r['Start_Timestamp']) for r in rows)
    long_=[r for r in rows if r['dur']>1.0]
    print('HIP API calls > 1 ms:', len(long_))
    for r in long_[-60:]: print(round((int(r['Start_Timestamp'])-t0)/1e6,1), r['Function'], round(r['dur'],2), r.get('Thread_Id'))
PY
