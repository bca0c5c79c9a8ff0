#!/bin/bash
# Development aid (GPU box): shader clock / power / temperature reported by rocm-smi while one kernel runs in a loop.
#   scripts/power_probe.sh n_fft hop what [seconds]
cd "$(dirname "$0")/.."
nf=$1; hop=$2; what=$3; secs=${4:-8}
PROBE_PREWARM_S=$secs timeout 300 python scripts/size_probe.py $nf $hop 30 $what > /tmp/pp_$$.log 2>&1 &
pid=$!
sleep 4   # (import + set-up)
for i in 1 2 3 4; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 1
done
wait $pid
grep n_fft /tmp/pp_$$.log
rm -f /tmp/pp_$$.log
