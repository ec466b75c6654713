import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peanut_amd import _lib
from peanut_amd.rcnn import MaskRCNN
from peanut_amd.rcnn_weights import RcnnCfg, make_seeded_rcnn_state_dict
key = sys.argv[1]
on_value = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
rcfg = RcnnCfg(score_thresh_test=0.5)
sd = make_seeded_rcnn_state_dict(rcfg, 0)
nets = {}
for v in (1, 0):
    with _lib.default_options(**{key: (on_value if v else 0)}):
        nets[v] = MaskRCNN(rcfg, sd, device=dev)
g = torch.Generator().manual_seed(3)
img = torch.randint(0, 256, (1, 480, 640, 3), generator=g, dtype=torch.uint8).to(dev)
res = {1: [], 0: []}
for rep in range(6):
    for v in (1, 0):
        n = nets[v]
        for _ in range(5):
            n.semantic(img, rcfg.num_classes, 0.5, 0.5, None)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(40):
            n.semantic(img, rcfg.num_classes, 0.5, 0.5, None)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t) / 40 * 1e3)
print(json.dumps({key: {str(v): [round(x, 3) for x in r] for v, r in res.items()}, "median_on": sorted(res[1])[3], "median_off": sorted(res[0])[3]}))
