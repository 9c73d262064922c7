#!/bin/bash
# Sample GPU power / clocks while a kernel loop runs:  tools/power_probe.sh <args for conv_probe.py>   (SECS = seconds of loop before sampling, default 8)
python tools/conv_probe.py --rounds ${ROUNDS:-4000} "$@" > /tmp/power_probe_run.log 2>&1 &
PID=$!
sleep ${SECS:-12}
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "package power\|sclk\|junction" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  sleep 1
done
kill $PID 2>/dev/null
wait $PID 2>/dev/null
tail -2 /tmp/power_probe_run.log
