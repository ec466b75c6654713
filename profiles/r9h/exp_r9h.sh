#!/bin/bash
# round 6, r9h: PSP head at batch 1 (pooling loads unserialised; deferred split-K loads with the part loop outermost)
out=${1:-gpurun_out/r9h}
mkdir -p $out
timeout 1200 python -m pytest tests/test_pred_gpu.py tests/test_rcnn_gpu.py -x -q -m gpu -k "split_k or golden or pool or pyramid or ppm or two_stream or deployed_720 or b4_480" > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
for sz in 720 240 480; do
echo "pred$sz b1: $(python bench.py --batch 1 --size $sz --steps 50 --warmup 5 --no-cpu-baseline --traffic none --no-probe --configs '' | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')" | tee -a $out/ab.txt
done
echo "headline: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic none --no-probe --configs '' | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" | tee -a $out/ab.txt
bash tools/pred_timeline.sh $out 1 720 | grep -E "ppm_|forwards" | head
bash tools/pred_timeline.sh $out 1 240 | grep -E "ppm_|forwards|wino4_input" | head
