#!/bin/bash
# round 6, r9c: small-problem Winograd transforms (csrc/winograd.hip *_small_kernel): parity + batch-1 A/B.  From the repo root.
out=${1:-gpurun_out/r9c}
mkdir -p $out
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "winograd or wino" > $out/pytest_wino.txt 2>&1
tail -4 $out/pytest_wino.txt
for v in 0 1024 0 1024; do
  echo "wino_small_maxwg=$v pred720: $(PEANUT_WINO_SMALL_MAXWG=$v python bench.py --batch 1 --size 720 --steps 50 --warmup 5 --no-cpu-baseline --traffic none --no-probe --configs '' | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')" | tee -a $out/ab.txt
  echo "wino_small_maxwg=$v detector b1: $(PEANUT_WINO_SMALL_MAXWG=$v python tools/bench_rcnn.py 1 2>/dev/null | grep '^{' | python -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d[k] for k in d if "ms" in k or k in ("batch",)})' | head -3 | tr '\n' ' ')" | tee -a $out/ab.txt
done
