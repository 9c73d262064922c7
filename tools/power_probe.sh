#!/bin/bash
# Sample GPU power / clocks while a kernel loop runs:  tools/power_probe.sh <args for conv_probe.py>
python tools/conv_probe.py --rounds 25000 "$@" > /tmp/power_probe_run.log 2>&1 &
PID=$!
sleep 22
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "package power\|sclk\|junction" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  sleep 1
done
wait $PID 2>/dev/null
tail -2 /tmp/power_probe_run.log
