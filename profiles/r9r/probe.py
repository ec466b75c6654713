"""One 720 x 720 map (and one 240 x 240) through the emulated modes N times: hashes of the logits.  The skinny kernel's inner-product form
comes from PEANUT_SKINNY_FORM (experiment build of gemm_skinny.hip, profiles/r9r/skinny_forms.patch)."""
import collections, hashlib, os, sys, torch
sys.path.insert(0, os.getcwd())
from types import SimpleNamespace
from peanut_amd.prediction import PEANUT_Prediction_Model
from peanut_amd.weights import PredCfg, make_seeded_state_dict
cfg = PredCfg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for precision in ("fp16x3", "bf16x3", "fp32"):
    m = PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=0), state_dict=make_seeded_state_dict(cfg, 0), cfg=cfg, precision=precision)
    for S in (720, 240):
        g = torch.Generator().manual_seed(1000 + S)
        x = (torch.rand((1, cfg.in_channels, S, S), generator=g) > 0.7).float().cuda()
        hs = collections.Counter()
        for _ in range(n):
            y = m.get_prediction_batch(x)
            hs[hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12]] += 1
        print(f"form={os.environ.get('PEANUT_SKINNY_FORM', '0')} {precision} S={S}: {dict(hs)}")
