"""Host-side cost of one exchange of the sharded path (sharding._Gather: device tensors, all_gather_into_tensor, the copy
back) under an RCCL process group of ONE rank on this GPU -- what every rank pays per gather besides the wire time.
usage: python tools/exchange_probe.py   (single process; sets up its own rendezvous on 127.0.0.1)"""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from gpyreg_amd import sharding as sh

print("exchange:", os.environ.get("GPYREG_AMD_EXCHANGE", "torch"), "(GPYREG_AMD_EXCHANGE=rccl: one direct ncclAllGather through ctypes)")
print(f"frame = 1 x {sh.FRAME} doubles per rank: the ONE gather of a sharded call whose rows fit in it (round 5)")
for rows, cols in ((1, sh.FRAME), (1, 3), (2, 15), (8, 15), (16, 2001)):
    a = np.random.default_rng(0).standard_normal((rows, cols))
    for _ in range(20):
        sh._Gather(a, rows).result()
    t0 = time.perf_counter()
    for _ in range(200):
        g = sh._Gather(a, rows)
    t_issue = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        sh._Gather(a, rows).result()
    t_all = (time.perf_counter() - t0) / 200
    print(f"block {rows} x {cols}: issue {t_issue*1e6:6.1f} us   issue + wait + copy back {t_all*1e6:6.1f} us", flush=True)
dist.barrier()
dist.destroy_process_group()
