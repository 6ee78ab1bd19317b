"""Child script of tests/test_gpu_zz_rccl.py::test_overlapped_buckets_with_deferred_weight_gradients: one rank of a 2-rank job whose ranks
share cuda:0 (the boxes have one GPU; gloo moves CUDA tensors through the host, RCCL needs one device per rank).  Each rank runs the
real stage-2 model on its own shard with FlatGradients.backward() — deferred weight gradients — and the overlapped early bucket, and
compares the averaged gradients with the plain (single all-reduce, immediate weight gradients) result on the same shards."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
from conan_fgw_amd.parallel import FlatGradients
from conan_fgw_amd.synthetic import make_batch, make_bond_graph

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
dev = torch.device("cuda:0")
K = 3
b = make_batch("esol", 6, K, seed=300 + rank)
g = make_bond_graph(b, seed=400 + rank)
t = lambda a: torch.from_numpy(a).to(dev)
batch = types.SimpleNamespace(z=t(b.z), pos=t(b.pos), x=t(g.x), edge_index=t(g.edge_index), edge_attr=t(g.edge_attr), batch=t(b.batch))
y = t(b.y)[:, None]
torch.manual_seed(5)
model = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
cidx = model.create_aggregation_index(b.num_graphs, dev)
flat = FlatGradients(model.parameters())


def step(deferred):
    flat.zero()
    loss = torch.nn.functional.mse_loss(model(batch, cidx, batch.batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes), y)
    if deferred:
        flat.backward(loss)
    else:
        loss.backward()
    flat.all_reduce_mean()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in model.named_parameters()}


plain = step(False)                               # one all-reduce, immediate weight gradients
err, launches, early, total = None, [], 0, 0
try:
    for frac in (0.5, 0.25, 0.75):                # different byte cuts: the cut lands inside different multi-output autograd nodes
        flat.enable_overlap(frac)
        step(True)                                # records the production order
        early, total = flat.calibrate()
        for _ in range(2):
            got = step(True)
            launches.append(flat.last_allreduce_launches)
            for k in plain:
                if not torch.allclose(got[k], plain[k], rtol=2e-5, atol=1e-7):
                    raise AssertionError(f"gradient of {k} differs with the early bucket at {frac}")
except Exception as e:                            # noqa: BLE001 - reported to the parent
    err = f"{type(e).__name__}: {e}"[:400]
if rank == 0:
    print(json.dumps({"n_ranks": dist.get_world_size(), "early": early, "total": total, "launches": launches, "error": err}), flush=True)
dist.destroy_process_group()
