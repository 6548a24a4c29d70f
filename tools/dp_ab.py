import os, sys, time, json
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "cilrs-autonomous-driving-carla_amd"))
import torch, torch.distributed as dist
mode = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
if "async" in mode: os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"] = "1"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
kw = {"device_id": dev} if "devid" in mode else {}
dist.init_process_group("nccl", rank=0, world_size=1, **kw)
from cilrs_mi355 import CILRS, CONFIG_A, Trainer
from cilrs_mi355.parallel import broadcast_parameters
torch.manual_seed(0)
m = CILRS(4, 0.0).to(dev)
tr = Trainer(m, CONFIG_A, process_group=dist.group.WORLD)
broadcast_parameters(tr.eng, dist.group.WORLD)
B = 128
img = torch.randn(B, 3, 88, 200, device=dev); spd = torch.rand(B, device=dev)
cmd = torch.randint(0, 4, (B,), device=dev); tgt = torch.rand(B, 3, device=dev)
for _ in range(10): tr.train_step(img, spd, cmd, tgt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): tr.train_step(img, spd, cmd, tgt)
torch.cuda.synchronize()
print(mode, round((time.perf_counter() - t0) / 50 * 1e3, 3), "ms/step", flush=True)
dist.destroy_process_group()
