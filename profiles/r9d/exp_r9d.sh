#!/bin/bash
# round 6, r9d: conv1's split-K partial tiles summed by conv2's Winograd input transform (option defer_splitk): parity + batch-1 A/B.
out=${1:-gpurun_out/r9d}
mkdir -p $out
timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_pred_gpu.py tests/test_rcnn_gpu.py -x -q -m gpu -k "winograd or wino or split_k or deployed_720 or golden_vectors or front_end or inference_end_to_end or agent_frame" > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
for v in 0 1 0 1 0 1; do
  echo "defer_splitk=$v pred720: $(PEANUT_DEFER_SPLITK=$v python bench.py --batch 1 --size 720 --steps 50 --warmup 5 --no-cpu-baseline --traffic none --no-probe --configs '' | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')" | tee -a $out/ab.txt
  echo "defer_splitk=$v detector b1: $(PEANUT_DEFER_SPLITK=$v python tools/bench_rcnn.py 1 2>/dev/null | grep '^{' | python -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d[k] for k in d if "ms" in k or k in ("batch",)})' | head -1 | tr '\n' ' ')" | tee -a $out/ab.txt
done
