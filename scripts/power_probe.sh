#!/bin/bash
# Is a tile kernel power-limited?  Samples rocm-smi (power, clocks; sysfs reads, no HIP) every ~0.15 s in a shell loop while
# scripts/power_load.py sweeps one workload back to back for a few seconds.  usage: power_probe.sh MODEL [N] [P] [SECONDS]
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
model=${1:-egno}; n=${2:-4096}; P=${3:-1}; secs=${4:-4}
log=gpurun_out/r06_power_${model//:/_}_${n}_x${P}.txt
: > "$log"
rocm-smi --showmaxpower --showpower -d 0 >> "$log" 2>&1
( for i in $(seq 1 400); do echo "t=$(date +%s.%N)" >> "$log"; rocm-smi -d 0 --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" >> "$log"; sleep 0.1; [ -f gpurun_out/.power_done ] && break; done ) &
sampler=$!
rm -f gpurun_out/.power_done
python scripts/power_load.py "$model" "$n" "$P" "$secs" >> "$log" 2>&1
rc=$?
touch gpurun_out/.power_done
wait $sampler
rm -f gpurun_out/.power_done
exit $rc
