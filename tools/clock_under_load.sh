#!/bin/bash
# Shader clock and package power while the LM loop runs (rocm-smi sampled every 0.2 s beside bench.py).
# usage: bash tools/clock_under_load.sh [workload] [extra bench flags...]
R=${GRAFT_REPO_ROOT:-/root/repo}
w=${1:-c3}; shift
mkdir -p $R/gpurun_out/clk
python3 $R/bench.py --no-cpu-baseline --no-api --workload $w --min-seconds 6 "$@" > $R/gpurun_out/clk/bench_$w.json 2> $R/gpurun_out/clk/bench_$w.err &
pid=$!
: > $R/gpurun_out/clk/smi_$w.log
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> $R/gpurun_out/clk/smi_$w.log
  echo "--" >> $R/gpurun_out/clk/smi_$w.log
  sleep 0.2
done
wait $pid
python3 - $R/gpurun_out/clk/smi_$w.log <<'PY'
import re, sys
s, p = [], []
for ln in open(sys.argv[1]):
    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", ln)
    if m: s.append(int(m.group(1)))
    m = re.search(r"Power \(W\): ([0-9.]+)", ln)
    if m: p.append(float(m.group(1)))
s.sort(); p.sort()
if s: print("sclk MHz: min", s[0], "median", s[len(s)//2], "max", s[-1], "samples", len(s))
if p: print("power W: min", p[0], "median", p[len(p)//2], "max", p[-1])
PY
tail -3 $R/gpurun_out/clk/smi_$w.log
