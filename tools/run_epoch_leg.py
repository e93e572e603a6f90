import json, sys, torch
sys.path.insert(0, ".")
import bench
r = bench.config3_epoch_leg(torch, torch.device("cuda"))
print(json.dumps(r))
