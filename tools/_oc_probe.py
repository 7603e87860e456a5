import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
o = bench.other_configs(dev)
for k in ("train_step_mdtv_B1024", "train_step_c3_mdtv_B1024"):
    print(k, o[k].get("ms_per_step"), o[k].get("sustained_mhz"), o[k].get("error"))
