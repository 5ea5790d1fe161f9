#!/bin/bash
# Is k_direct_accumulate limited by instruction issue or by the board's power limit? Two builds, same box:
#   chain   = products as chained multiply-adds (5 % fewer instructions), nochain = -DLWK_NO_MAD_CHAIN (the compiler's form)
# for each: bench line, SQ_INSTS_VALU and GRBM_GUI_ACTIVE passes (clock held), and power / clock samples from rocm-smi.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/p20; mkdir -p $O
cd $R
for v in chain nochain chain nochain; do
  if [ $v = nochain ]; then export LWKZG_LIBRARY=$R/lambdaworks_kzg_amd/lib_nochain/liblambdaworks_kzg.so; else unset LWKZG_LIBRARY; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import json,sys; j=json.load(sys.stdin); print('$v', round(j['value']), 'ops/s', j['roofline']['avg_launch_ms'], 'ms')" >> $O/ab.txt
done
for v in chain nochain; do
  if [ $v = nochain ]; then export LWKZG_LIBRARY=$R/lambdaworks_kzg_amd/lib_nochain/liblambdaworks_kzg.so; else unset LWKZG_LIBRARY; fi
  for ctr in SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
    (cd /tmp && rocprofv3 --pmc $ctr -d $O/pmc_${v}_$ctr -o p --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/pmc_${v}_$ctr.log 2>&1)
  done
  python3 tools/pmc_issue_summary.py k_direct_accumulate $O/issue_$v.json $(find $O/pmc_${v}_SQ_INSTS_VALU $O/pmc_${v}_GRBM_GUI_ACTIVE -name "*counter_collection.csv") > /dev/null 2>&1
  # power and clock while the kernel runs back to back for a few seconds
  (timeout 300 python bench.py --steps 400 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/long_$v.json 2>/dev/null &) 
  sleep 14
  for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.4; done > $O/smi_$v.txt 2>&1
  wait; sleep 8
done
rm -rf $O/pmc_*_SQ_INSTS_VALU $O/pmc_*_GRBM_GUI_ACTIVE
cat $O/ab.txt; for v in chain nochain; do python3 -c "import json; j=json.load(open('$O/issue_$v.json')); print('$v', {k: j.get(k) for k in ('clock_GHz','cycles_per_VALU_instruction_per_SIMD','avg_launch_ms_under_the_profiler')}, j['counters_per_launch'])"; cat $O/smi_$v.txt | head -4; python3 -c "import json; j=json.load(open('$O/long_$v.json')); print('long run $v', round(j['value']))"; done
