for f in 0 1; do
for cfg in "--batch 32 --size 480 --steps 10" "--batch 1 --size 240 --steps 50" "--batch 1 --size 720 --steps 50" "--batch 32 --size 480 --steps 10 --precision bf16x6"; do
PEANUT_SPLIT_MODEL=$f python bench.py $cfg --warmup 3 --no-cpu-baseline --also "" --traffic none --no-probe 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('model=$f', '$cfg', d['value'], d['ms_per_step'])
"
done
PEANUT_SPLIT_MODEL=$f python tools/bench_rcnn.py 1 2>&1 | grep -o '"precision.*"front_end_ms": [0-9.]*' 
PEANUT_SPLIT_MODEL=$f python tools/bench_rcnn.py 16 2>&1 | grep -o '"precision.*"front_end_ms": [0-9.]*' 
done
