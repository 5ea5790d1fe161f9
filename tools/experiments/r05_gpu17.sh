#!/bin/bash
# what k_eval_quotient_evalform spends its 0.43 ms on: issue counters of a c-kzg proof run (own rocprofv3 runs, --pmc only)
O=gpurun_out/r05/pmc_evf; mkdir -p $O
P="python3 bench.py --op blob_proof --mode ckzg --batch 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sq1 -o sq -- $P > $O/l1.json 2> $O/e1.txt
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $O/sq3 -o sq -- $P > $O/l3.json 2> $O/e3.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq2 -o sq -- $P > $O/l2.json 2> $O/e2.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -o grbm -- $P > $O/l4.json 2> $O/e4.txt
python3 - <<'PY'
import csv, glob, collections
for d in ('sq1','sq3','sq2','grbm'):
    for f in glob.glob('gpurun_out/r05/pmc_evf/%s/*counter_collection.csv' % d):
        rows=list(csv.DictReader(open(f)))
        acc=collections.defaultdict(lambda: [0,0.0])
        for r in rows:
            k=r['Kernel_Name']
            if 'eval_quotient' in k or 'copy_le' in k:
                a=acc[(k[:60], r['Counter_Name'])]; a[0]+=1; a[1]+=float(r['Counter_Value'])
        for (k,c),(n,v) in sorted(acc.items()): print(k, c, 'launches*dims', n, 'avg', v/n)
PY
