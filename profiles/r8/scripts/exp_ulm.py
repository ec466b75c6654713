import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_pipeline as bp
from peanut_amd.agent_helper import preprocess_obs
from peanut_amd.agent_state import Agent_State, default_args
from peanut_amd.weights import PredCfg, make_seeded_state_dict
dev = torch.device("cuda", 0)
args = default_args(only_explore=0, sem_gpu_id=0, pred_precision="fp32", select_goal=True)
st = Agent_State(args, state_dict=make_seeded_state_dict(PredCfg(), 0))
ep = bp.synth_episode(1000, 40, dev)
acc = {}
def T(name, fn, sync=True):
    if sync: torch.cuda.synchronize()
    t = time.perf_counter(); r = fn()
    if sync: torch.cuda.synchronize()
    acc.setdefault(name, []).append((time.perf_counter() - t) * 1e6); return r
st.reset()
sem = torch.zeros((480, 640, args.num_sem_categories), device=dev)
for i, fr in enumerate(ep):
    obs = preprocess_obs(fr["rgb"], fr["depth"], fr.get("sem", sem) if "sem" in fr else sem, args)
    infos = {"sensor_pose": fr["sensor_pose"], "goal_cat_id": 3}
    if i == 0:
        st.init_with_obs(obs, infos)
    self = st
    self.poses = T("pose_h2d", lambda: torch.from_numpy(np.asarray(infos['sensor_pose'])).float().to(self.device))
    T("map_step", lambda: self._map_step(obs))
    locs = T("pose_d2h", lambda: self.local_pose.cpu().numpy())
    self.planner_pose_inputs[:3] = locs + self.origins
    T("fill_ch2", lambda: self.local_map[2, :, :].fill_(0.))
    r, c = locs[1], locs[0]
    loc_r = int(r * 100.0 / args.map_resolution); loc_c = int(c * 100.0 / args.map_resolution)
    def traj():
        self.local_map[2:4, loc_r - 2:loc_r + 3, loc_c - 2:loc_c + 3] = 1.
    T("traj", traj)
    off = int(args.col_rad + 1)
    def sel():
        self.local_map[1][self._selem_r - off + loc_r, self._selem_c - off + loc_c] = 1.
    T("selem", sel)
    # whole thing unsynchronised
    def whole():
        self.poses = torch.from_numpy(np.asarray(infos['sensor_pose'])).float().to(self.device)
        self.update_local_map(obs)
    T("whole_update_local_map", whole)
    t = time.perf_counter(); 
    self.poses = torch.from_numpy(np.asarray(infos['sensor_pose'])).float().to(self.device)
    self.update_local_map(obs)
    acc.setdefault("whole_host_only_nosync", []).append((time.perf_counter() - t) * 1e6)
print(json.dumps({k: round(sum(v[5:]) / len(v[5:]), 1) for k, v in acc.items()}))
