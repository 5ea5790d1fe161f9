# What the board does while the headline kernel runs back to back: socket power, power cap, shader clock and temperature sampled from
# sysfs / rocm-smi every ~100 ms beside `bench.py --steps 800` (about 8 s of accumulation). One gpurun call; writes gpurun_out/power/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/power
rm -rf $O; mkdir -p $O
cd $R
rocm-smi --showpower --showmaxpower --showclocks --showtemp --showperflevel --showvoltage > $O/idle_rocm_smi.txt 2>&1
amd-smi metric --power --clock --temperature > $O/idle_amd_smi.txt 2>&1
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > $O/hwmon_ls.txt 2>&1
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
D=$(dirname $(dirname $H))
echo "hwmon=$H dev=$D" > $O/paths.txt
cat $H/power1_cap $H/power1_cap_max $H/power1_cap_default >> $O/paths.txt 2>&1
cat $D/power_dpm_force_performance_level >> $O/paths.txt 2>&1
cat $D/pp_dpm_sclk >> $O/paths.txt 2>&1
( python bench.py --steps 800 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/line.json 2> $O/err.txt; touch $O/done ) &
BP=$!
: > $O/samples.txt
while [ ! -e $O/done ]; do
  T=$(date +%s.%N)
  P=$(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null)
  F=$(cat $H/freq1_input 2>/dev/null)
  M=$(cat $H/freq2_input 2>/dev/null)
  C=$(cat $H/temp1_input 2>/dev/null)
  J=$(cat $H/temp2_input 2>/dev/null)
  echo "$T power_uW=$P sclk_Hz=$F mclk_Hz=$M temp_edge=$C temp_junction=$J" >> $O/samples.txt
  sleep 0.1
done
wait $BP
rocm-smi --showpower --showclocks > $O/after_rocm_smi.txt 2>&1
tail -40 $O/samples.txt
cat $O/paths.txt
python3 -c "
import json;d=json.load(open('$O/line.json'));print(d['value'],d['ms_per_step'],d['kernels'])"
