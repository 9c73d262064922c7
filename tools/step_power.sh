#!/bin/bash
# Board power / shader clock while the C2 step loops (is the step as a whole at the 1 400 W cap, or only its matrix kernels?):
#   tools/step_power.sh [bench args]   -> 20 rocm-smi samples 0.5 s apart, taken while `bench.py --steps 900` runs
python bench.py --steps ${STEPS:-900} --warmup 5 --no-cpu-baseline --no-alone --no-secondary "$@" > /tmp/step_power_bench.json 2>/dev/null &
PID=$!
sleep ${SECS:-12}
for i in $(seq 20); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "package power\|sclk\|mclk" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $PID
python -c "
import json; d=json.loads(open('/tmp/step_power_bench.json').read().strip().splitlines()[-1]); print('bench:', d['value'], 'crops/s', d['ms_per_step'], 'ms per step over', d['steps'], 'steps')"
