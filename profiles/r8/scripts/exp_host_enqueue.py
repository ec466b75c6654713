import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
from bench import synth_maps
from peanut_amd.prediction import PEANUT_Prediction_Model
from peanut_amd.weights import PredCfg, make_seeded_state_dict
dev = torch.device("cuda", 0)
cfg = PredCfg()
m = PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=0), state_dict=make_seeded_state_dict(cfg, 0), cfg=cfg)
x = synth_maps(1, cfg.in_channels, 720, dev, seed0=5)
for _ in range(3):
    m.get_prediction_batch(x)
torch.cuda.synchronize()
host, total = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    y = m.get_prediction_batch(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t) * 1e3); total.append((t2 - t) * 1e3)
print(json.dumps({"host_enqueue_ms": round(sum(host) / len(host), 3), "total_ms": round(sum(total) / len(total), 3), "min_host": round(min(host), 3)}))
# the raw forward
f = m.model
host, total = [], []
out = torch.empty((1, cfg.num_classes, 720, 720), device=dev)
for _ in range(20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    y = f.forward_logits(x, apply_sigmoid=True, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t) * 1e3); total.append((t2 - t) * 1e3)
print(json.dumps({"raw_forward_host_ms": round(sum(host) / len(host), 3), "total_ms": round(sum(total) / len(total), 3)}))
