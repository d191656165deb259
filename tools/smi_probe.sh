#!/bin/bash
# sample power / clocks from sysfs while a kernel loop runs
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
D=$(dirname $(dirname $H))
echo "hwmon=$H"; ls $H | tr '\n' ' '; echo
cat $H/power1_cap $H/power1_cap_max 2>/dev/null
ls $D | grep -i "pp_\|gpu_busy\|power" | tr '\n' ' '; echo
cat $D/pp_dpm_sclk 2>/dev/null | head -5
./flash_attention_from_scratch_amd/lib/tune64 only=$1 reps=$2 > /tmp/t.log 2>&1 &
PID=$!
for i in $(seq 1 60); do
  p=$(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null); f=$(cat $H/freq1_input 2>/dev/null); t=$(cat $H/temp1_input 2>/dev/null)
  echo "$i power_uW=$p sclk_Hz=$f temp=$t busy=$(cat $D/gpu_busy_percent 2>/dev/null)"
  sleep 0.1
  kill -0 $PID 2>/dev/null || break
done
wait $PID
grep "S= 4096" /tmp/t.log
