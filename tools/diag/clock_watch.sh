#!/usr/bin/env bash
# sclk / mclk / socket power as rocm-smi reports them WHILE a command runs (one sample every ~0.25 s):  bash tools/diag/clock_watch.sh <out.log> <command...>
OUT="$1"; shift
"$@" > "${OUT%.log}.cmd.out" 2> "${OUT%.log}.cmd.err" &
PID=$!
: > "$OUT"
while kill -0 $PID 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' '|' >> "$OUT"
  echo >> "$OUT"
  sleep 0.25
done
wait $PID
