#!/bin/bash
# Clock and socket power while the D = 128 extend kernel runs back to back, random vs all-zero operands (same instruction
# stream): rocm-smi sampled twice a second next to tools/ext32_ab.py.
for Z in "" 1; do
  echo "=== ZERO=$Z"
  ZERO=$Z ROUNDS=60 REPS=8 VARIANTS=0 python tools/ext32_ab.py > /tmp/ab_$Z.log 2>&1 &
  PID=$!
  sleep 14
  for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
    sleep 0.5
  done
  wait $PID
  tail -1 /tmp/ab_$Z.log
done
