#!/bin/bash
# Sample sclk / power with rocm-smi while the bench (or any command given) runs: is the kernel clock- or power-limited?
OUT=${OUT:-gpurun_out/clock_watch.txt}
mkdir -p "$(dirname "$OUT")"
"$@" > "$OUT.cmd" 2>&1 &
pid=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr -s ' ' | tr '\n' ';' >> "$OUT"
  echo >> "$OUT"
  sleep 1
done
wait $pid
cat "$OUT"
tail -c 600 "$OUT.cmd"
