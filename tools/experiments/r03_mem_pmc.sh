# memory-side counters of the headline kernel (one gpurun call; separate --pmc passes, no trace flags): how long do the row gathers take?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mem_pmc
rm -rf $O; mkdir -p $O
cd $R
P="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs"
i=0
for set in "VmemLatency" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES" "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_LATENCY_FIFO_FULL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  LWKZG_LIBRARY=$LIBV rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- $P > $O/p$i.json 2> $O/p$i.err
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/mem_pmc/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_direct_accumulate_asm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-50s %16.4g  (launches %d)" % (k, sum(v) / len(v), len(v)))
PY
