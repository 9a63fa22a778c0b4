import os, torch, torch.distributed as dist, sys
sys.path.insert(0, "/root/repo")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from huggingface_asr_amd import parallel as PL
import time
dt = PL.timed(lambda: time.sleep(0.01), 3, sync=torch.cuda.synchronize, device=dev)
print("timed", dt, "mean", float(PL.mean_over_ranks(torch.tensor(2.0, device=dev))))
from huggingface_asr_amd.train import GradSync
g = torch.ones(1000, device=dev)
s = GradSync(g); print("gradsync on", s.on, s.world)
h = dist.all_reduce(g[100:200], async_op=True); h.wait(); torch.cuda.synchronize(); print(float(g.sum()))
dist.barrier(); dist.destroy_process_group(); print("ok")
