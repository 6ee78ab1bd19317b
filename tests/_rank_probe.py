"""Child script of tests/test_parallel_cpu.py::test_bench_spawn_path: what one rank of `bench.py --gpus N` does around the model —
read RANK / WORLD_SIZE / MASTER_* from the environment torch.distributed.run prepared, form the process group (gloo here, RCCL
in bench.py), average the real parameter set of SchNetNoSum through FlatGradients with the overlapped early bucket, and let
rank 0 print ONE JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from conan_fgw_amd.parallel import FlatGradients
from conan_fgw_amd.schnet import SchNetNoSum

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.manual_seed(5)
model = SchNetNoSum(torch.device("cpu"), hidden_channels=32, num_filters=32, num_interactions=2)
flat = FlatGradients(model.parameters())
gen = torch.Generator().manual_seed(100 + rank)
coef = [torch.randn(p.shape, generator=gen) for p in flat.params]


def backward():
    flat.zero()
    # a loss that reaches every parameter, in reverse registration order like a real backward (the model itself has no CPU path)
    loss = sum((p * c).sum() for p, c in zip(reversed(flat.params), reversed(coef)))
    loss.backward()


flat.enable_overlap(0.5)
backward()
early, total = flat.calibrate()
launches = []
for _ in range(2):
    backward()
    flat.all_reduce_mean()
    launches.append(flat.last_allreduce_launches)
ref = torch.cat([c.reshape(-1) for c in (coef[i] for i in flat._order_idx)])
allref = [torch.zeros_like(ref) for _ in range(world)]
dist.all_gather(allref, ref)
expect = sum(allref) / world
ok = bool(torch.allclose(flat.flat, expect, rtol=1e-6, atol=1e-7))
aliased = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(flat.params, flat._views))
if rank == 0:
    print(json.dumps({"n_ranks": dist.get_world_size(), "early": early, "total": total, "launches": launches, "ok": ok, "aliased": aliased,
                      "args": sys.argv[1:]}), flush=True)
dist.destroy_process_group()
