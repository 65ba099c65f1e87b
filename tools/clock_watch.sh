#!/bin/bash
# Engine clock and socket power while a workload runs: rocm-smi sampled every 0.3 s beside the command.
# usage (GPU box, repo root): tools/clock_watch.sh <tag> <command ...>
tag=$1; shift
out=gpurun_out/clock_$tag.txt; : > $out
"$@" > gpurun_out/clock_$tag.run.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use" | tr '\n' ' ' | sed 's/  */ /g' >> $out
  echo >> $out
  sleep 0.3
done
wait $pid
echo "== $tag: $(tail -1 gpurun_out/clock_$tag.run.log | cut -c1-160)"
python3 - "$out" <<'PY'
import re, sys
sclk, pw = [], []
for l in open(sys.argv[1]):
    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)
    u = re.search(r"GPU use \(%\): (\d+)", l)
    p = re.search(r"Power \(W\): ([\d.]+)", l)
    if m and u and int(u.group(1)) > 50:
        sclk.append(int(m.group(1)))
        if p: pw.append(float(p.group(1)))
if sclk:
    sclk.sort(); pw.sort()
    print(f"   {len(sclk)} samples under load: sclk min {sclk[0]} median {sclk[len(sclk)//2]} max {sclk[-1]} MHz" + (f"; power median {pw[len(pw)//2]:.0f} W max {pw[-1]:.0f} W" if pw else ""))
else:
    print("   no samples under load"); print(open(sys.argv[1]).read()[:600])
PY
