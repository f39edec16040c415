#!/bin/bash
# Sample clocks / power with rocm-smi while a command runs: tools/power_trace.sh <outfile> <cmd...>
out=$1; shift
"$@" > "$out.cmd.log" 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | head -c 4000 >> "$out"
  echo >> "$out"
  sleep 0.5
done
wait $pid
tail -1 "$out.cmd.log"
