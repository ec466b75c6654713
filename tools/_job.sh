mkdir -p gpurun_out/r3p
out=gpurun_out/r3p/solo_sweep.txt; : > $out
for solo in 1.0 1.1 1.2 1.3 1.5; do
  for sz in 720 240; do
    for p in fp32 bf16x6; do
      ms=$(PEANUT_SPLIT_SOLO=$solo python bench.py --batch 1 --size $sz --steps 100 --warmup 10 --precision $p --also "" --no-cpu-baseline --traffic none --no-probe 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "solo=$solo size=$sz $p ms=$ms" >> $out
    done
  done
  PEANUT_SPLIT_SOLO=$solo PRECS=fp32,bf16x6 python tools/bench_rcnn.py 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('solo=$solo detector', d['precision'], d['ms_per_batch'], 'front', d['front_end_ms'])" >> $out
done
cat $out
