"""Latency of the one collective of a sharded sweep (32-byte record all-gather + host copy),
through a real RCCL communicator of one rank (developer aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl")
from optbayesexpt_amd.dist import SettingsShard
sh = SettingsShard()
rec = torch.zeros(4, dtype=torch.float64, device="cuda")
for _ in range(20): sh.combine_records(rec, 1000)
for label, f in (("combine_records (all_gather_into_tensor + .cpu())", lambda: sh.combine_records(rec, 1000)),
                 ("plain .cpu() of the record", lambda: rec.cpu())):
    torch.cuda.synchronize(); ts = []
    for _ in range(200):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    print(f"{label}: median {1e6*np.median(ts):.1f} us, min {1e6*min(ts):.1f}, max {1e6*max(ts):.1f}")
dist.destroy_process_group()
