#!/bin/bash
# end-to-end A/B of option sets on ONE box: tools/exp_e2e.sh <out> "<env assignments>" ...   (each run: headline, 20 steps)
out=$1; shift
mkdir -p $out
for rep in 1 2 3; do
  for cfg in "$@"; do
    line=$(env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline --also "" --traffic none --configs "" --no-probe 2>/dev/null | grep "^{")
    echo "$cfg $(echo $line | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")" | tee -a $out/e2e.txt
  done
done
