import sys, torch, json
sys.path.insert(0, '.')
from oracle import torch_standin as TS
for layout in ("chw", "hwc"):
    r = TS.time_train_steps("cuda:0", 4096, steps=10, warmup=3, plane_layout=layout)
    print(layout, json.dumps({k: r[k] for k in ("seconds_per_step", "rays_per_s", "loss")}))
    torch.cuda.empty_cache()
