#!/bin/bash
# round 6, r9i: skinny grouped GEMM for the PSP pyramid's per-scale convs and Q tables at batch 1 (csrc/gemm_skinny.hip, option pw_skinny)
out=${1:-gpurun_out/r9i}
mkdir -p $out
timeout 1200 python -m pytest tests/test_pred_gpu.py -x -q -m gpu -k "golden or deployed_720 or two_stream or pyramid or ppm or get_prediction or variant or fold" > $out/pytest.txt 2>&1
tail -4 $out/pytest.txt
for rep in 1 2; do for v in 0 1; do for sz in 240 480 720; do
echo "pw_skinny=$v pred$sz b1: $(PEANUT_PW_SKINNY=$v python bench.py --batch 1 --size $sz --steps 50 --warmup 5 --no-cpu-baseline --traffic none --no-probe --configs '' | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')" | tee -a $out/ab.txt
done; done; done
bash tools/pred_timeline.sh $out 1 720 | tail -16
bash tools/pred_timeline.sh $out 1 240 | tail -14
