import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peanut_amd.rcnn import MaskRCNN
from peanut_amd.rcnn_weights import RcnnCfg, make_seeded_rcnn_state_dict
from peanut_amd.segmentation import accumulate_instances
cfg = RcnnCfg(score_thresh_test=0.5)
sd = make_seeded_rcnn_state_dict(cfg, 0)
m = MaskRCNN(cfg, sd)
img = torch.randint(0, 256, (1, 480, 640, 3), dtype=torch.uint8, device="cuda")
for _ in range(3):
    res = m.inference(img)
    sem = [accumulate_instances(r["pred_masks"], r["pred_classes"], r["scores"], cfg.num_classes, 0.5, 0.5, None) for r in res]
torch.cuda.synchronize()
print(len(res[0]["scores"]))
