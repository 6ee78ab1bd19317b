#!/usr/bin/env python3
"""Benchmark of the ConAN hot path on MI355X:  molecules/s (K=5 conformers).

One "step" = one training step of the stage-2 hot path on one batch of synthetic ESOL-shaped conformers resident in HBM:
SchNetNoSum.forward_w_barycenter (radius graph, 3 interaction blocks, two heads, FGW barycenter over the K conformers)
+ covalent GAT branch on the 2-D bond graph + conformer-aggregation head (the reference's whole stage-2 model,
schnet_based_models.py:135-173) + MSE loss, backward through all of it, one flat gradient all-reduce (RCCL) and Adam.
Workload at every N: BASELINE.json configs[1] per GPU (ESOL + SchNet-128, K=5, batch=256) => weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|fwd] [--batch B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events around the CFConv gather/segment-sum kernel
(the HBM-bound kernel BASELINE.json's target is quoted on); `cpu_baseline` times the CPU oracle on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["train", "fwd"], default="train")
    ap.add_argument("--batch", type=int, default=256, help="molecules per GPU")
    ap.add_argument("--conformers", type=int, default=5)
    ap.add_argument("--shape", default="esol")
    ap.add_argument("--model", choices=["schnet", "visnet"], default="schnet", help="backbone (BASELINE.json configs[3] = visnet + bace)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def cfconv_algorithmic_bytes(E, P, n_atoms, F):
    """Compulsory HBM bytes of one CFConv message+aggregate launch, every tensor touched once:
    4F per UNIQUE filter row (the filter is shared by the two directions of a pair, W_ij = W_ji: P ~ E/2 rows; P = E
    without sharing) + n*4F [x read once: the E gathers of x_j are re-reads of those n rows] + n*4F [store out]
    + int32 indices 4*(2E + n+1) [col, pid, rowptr]."""
    return P * 4 * F + 2 * n_atoms * 4 * F + 4 * (2 * E + n_atoms + 1)


def cfconv_survey_bytes(E, P, n_atoms, F):
    """SURVEY.md 8(d) convention (W materialised, every gather of x_j counted as E*4F of traffic): an upper bound that
    ignores cache reuse of x; reported next to the compulsory figure, not used for `frac`."""
    return E * 4 * F + P * 4 * F + 4 * (2 * E + n_atoms + 1) + n_atoms * 4 * F


def cpu_baseline(args, mode, gpu_model=None):
    """CPU oracle ("port": oracle/schnet.py + the C FGW restatement) on the host cores, bounded sample."""
    from conan_fgw_amd.synthetic import make_batch, make_bond_graph
    from oracle.head import Stage2Oracle
    nb = 8
    b = make_batch(args.shape, nb, args.conformers, seed=4321)
    bg = make_bond_graph(b, seed=4322)
    torch.manual_seed(5)
    m = Stage2Oracle(args.conformers)
    z, pos, batch = torch.from_numpy(b.z), torch.from_numpy(b.pos), torch.from_numpy(b.batch)
    bx, bei, bea = torch.from_numpy(bg.x), torch.from_numpy(bg.edge_index), torch.from_numpy(bg.edge_attr)
    y = torch.from_numpy(b.y)[:, None]

    def step():
        pred = m(z, pos, batch, bx, bei, bea)
        if mode == "train":
            loss = torch.nn.functional.mse_loss(pred, y)
            loss.backward()
    step()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        step(); n += 1
    dt = time.perf_counter() - t0
    # the same oracle, forward only (end-to-end forward next to the training step), ~3 s
    with torch.no_grad():
        m(z, pos, batch, bx, bei, bea)
        tf0 = time.perf_counter(); nfw = 0
        while time.perf_counter() - tf0 < min(3.0, args.cpu_seconds):
            m(z, pos, batch, bx, bei, bea); nfw += 1
        fwd_rate = nb * nfw / (time.perf_counter() - tf0)
    out = {"value": round(nb * n / dt, 3), "unit": "molecules/s", "cores": torch.get_num_threads(), "kind": "port",
           "forward_only_molecules_per_s": round(fwd_rate, 3),
           "sample": f"{n} {mode} steps of {nb} {args.shape}-shaped molecules (K={args.conformers}), CPU oracle fp32 (SchNet + FGW + GAT + head), "
                     f"{torch.get_num_threads()} torch threads of {os.cpu_count()} host cores; FGW = scalar C restatement"}
    if gpu_model is not None and args.model == "schnet":
        # the same sample through the HIP path and through the fp64 oracle with the HIP model's current weights
        import types
        ref = Stage2Oracle(args.conformers)
        ref.load_state_dict({k: v.detach().cpu() for k, v in gpu_model.state_dict().items()})
        ref = ref.double()
        dev = next(gpu_model.parameters()).device
        ns = types.SimpleNamespace(z=z.to(dev), pos=pos.to(dev), x=bx.to(dev), edge_index=bei.to(dev), edge_attr=bea.to(dev), batch=batch.to(dev))
        with torch.no_grad():
            gy = gpu_model(ns, None, ns.batch)
            g3, gb = gpu_model.node_embeddings_model.forward_w_barycenter(ns.z, ns.pos, args.conformers, ns.batch)
            ry = ref(z, pos.double(), batch, bx, bei, bea)
            r3, rb = ref.node_embeddings_model.forward_w_barycenter(z, pos.double(), args.conformers, batch)
        rel = lambda a, r: float((a.detach().cpu().double() - r).norm() / r.norm())
        out["gpu_vs_oracle_fp64"] = {"y_pred_rel_err (energies)": float(f"{rel(gy, ry):.3e}"), "h_3d_rel_err": float(f"{rel(g3, r3):.3e}"),
                                     "h_bary_rel_err (FGW)": float(f"{rel(gb, rb):.3e}"), "tolerance": 1e-4,
                                     "sample": f"{nb} molecules of the CPU sample, current weights"}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")              # "nccl" is RCCL on ROCm: one rank per GPU over xGMI

    from conan_fgw_amd import ops
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from conan_fgw_amd.parallel import FlatGradients
    from conan_fgw_amd.synthetic import make_batch, make_bond_graph
    import types

    K = args.conformers
    b = make_batch(args.shape, args.batch, K, seed=1236 + 1000 * rank)            # cfg2 seed (1234 + 2) on rank 0
    bg = make_bond_graph(b, seed=2236 + 1000 * rank)                              # 2-D bond graph of the same molecules
    z, pos, batch = (torch.from_numpy(a).to(dev) for a in (b.z, b.pos, b.batch))
    data = types.SimpleNamespace(z=z, pos=pos, batch=batch, x=torch.from_numpy(bg.x).to(dev),
                                 edge_index=torch.from_numpy(bg.edge_index).to(dev), edge_attr=torch.from_numpy(bg.edge_attr).to(dev))
    y = torch.from_numpy(b.y).to(dev)[:, None]
    torch.manual_seed(5)                                                          # train_val.py:223
    # the reference's stage-2 model: backbone (common.py:524-529 / :542-546) + GAT branch + aggregation head
    model = EmbeddingsWithGATAggregationBaryCenter(K, dev, model_name=args.model).to(dev)
    cidx = model.create_aggregation_index(b.num_graphs, dev)
    params = list(model.parameters())
    flat = FlatGradients(params)
    opt = torch.optim.Adam(flat.params, lr=1e-4, fused=True)

    # live per-kernel timing of the CFConv forward kernel with HIP events on the launch stream
    ev = []
    ev_empty = []
    ev_other = {"conan_filter_fwd": [], "conan_fgw_barycenter_fwd": []}
    orig_call = ops.call

    def timed_call(name, *a):
        if timed_call.on and (name == "conan_cfconv_fwd" or name in ev_other):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); orig_call(name, *a); e.record()
            (ev if name == "conan_cfconv_fwd" else ev_other[name]).append((s, e))
            if name == "conan_cfconv_fwd":       # an EMPTY bracket right behind it: what two event packets cost on this queue by themselves
                s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record(); e0.record()
                ev_empty.append((s0, e0))
        else:
            orig_call(name, *a)
    timed_call.on = False
    ops.call = timed_call

    def step():
        if args.mode == "train":
            flat.zero()
            pred = model(data, cidx, batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)
            loss = torch.nn.functional.mse_loss(pred, y)
            loss.backward()
            flat.all_reduce_mean()
            opt.step()
        else:
            with torch.no_grad():
                model(data, cidx, batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timed_call.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    timed_call.on = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # forward-only throughput of the same workload (outside the timed region; this rank's shard)
    fwd_extra = None
    if args.mode == "train":
        def fwd_step():
            with torch.no_grad():
                model(data, cidx, batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)
        for _ in range(2):
            fwd_step()
        torch.cuda.synchronize()
        tf = time.perf_counter()
        nf = max(5, args.steps // 2)
        for _ in range(nf):
            fwd_step()
        torch.cuda.synchronize()
        tf = (time.perf_counter() - tf) / nf
        fwd_extra = {"molecules_per_s_per_gpu": round(args.batch / tf, 1), "ms_per_step": round(1e3 * tf, 4)}

    # edge statistics of this rank's batch (device graph of the last step is rebuilt here only for reporting)
    gp = ops.graph_ptr_from_batch(batch, b.num_graphs)
    _g = ops.RadiusGraph(pos, gp, b.num_graphs, 10.0, 32)
    E = _g.num_edges
    P = int(_g.pairs().num_pairs_dev.item()) if args.model == "schnet" else E
    n_atoms = int(z.shape[0])
    kdur_ms = float(np.mean([s.elapsed_time(e) for s, e in ev])) if ev else float("nan")
    empty_ms = float(np.mean([s.elapsed_time(e) for s, e in ev_empty])) if ev_empty else float("nan")
    # HBM traffic of the same kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, run separately and
    # committed under profiles/; gfx950 correction 2*FETCH_SIZE + WRITE_SIZE) -- only valid for the default workload
    traffic = None
    try:
        if args.shape == "esol" and args.batch == 256 and K == 5:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_hbm.json")))["kernels"]["k_cfconv_fwd<1>"]
            traffic = int(pm["traffic_bytes_corrected"])
    except Exception:
        traffic = None
    alg = cfconv_algorithmic_bytes(E, P, n_atoms, 128)
    achieved = alg / (kdur_ms * 1e-3) / 1e9

    other = []
    if ev_other["conan_filter_fwd"]:
        t_ms = float(np.mean([s.elapsed_time(e) for s, e in ev_other["conan_filter_fwd"]]))
        fl = P * 2.0 * (50 * 128 + 128 * 128)                       # (2*Gs*F + 2*F*F) per filter row; P rows (pairs)
        other.append({"kernel": "k_filter_fused (rbf -> filter MLP -> cosine cutoff)", "bound": "mfma", "achieved": round(fl / (t_ms * 1e-3) / 1e12, 2),
                      "peak": 157.3, "unit": "TFLOP/s", "frac": round(fl / (t_ms * 1e-3) / 1e12 / 157.3, 4), "avg_launch_ms": round(t_ms, 5),
                      "note": "fp32-equivalent algorithmic FLOP vs the fp32 MFMA peak; the second GEMM runs as an exact 3-way bf16 split"})
    if ev_other["conan_fgw_barycenter_fwd"]:
        t_ms = float(np.mean([s.elapsed_time(e) for s, e in ev_other["conan_fgw_barycenter_fwd"]]))
        other.append({"kernel": "FGW barycenter, whole batched solve (init + 5 x (coupling + update))", "bound": "fp64 issue / latency",
                      "avg_ms": round(t_ms, 4), "us_per_molecule": round(1e3 * t_ms / args.batch, 3),
                      "algorithmic_bytes_per_molecule": 4 * (2 * K * b.max_nodes ** 2 + K * b.max_nodes * 64 + b.max_nodes * 64 + b.max_nodes ** 2)})
    if rank == 0:
        mol = args.batch * world * args.steps
        out = {
            "metric": "molecules/s (K=5 conformers)", "value": round(mol / dt, 1), "unit": "molecules/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.shape.upper()}-shaped + " + ("ViSNet-128 (6 layers, 8 heads, 32 RBF, cutoff 5 A), " if args.model == "visnet" else "SchNet-128 (3 interactions, 50 gaussians, cutoff 10 A, cap 32), ")
                                   + f"K={K}, batch={args.batch} molecules per GPU, {args.mode} step "
                                   + ("(stage-2 model incl. GAT branch: fwd + bwd + flat-gradient all-reduce + Adam)" if args.mode == "train" else "(forward_w_barycenter + GAT branch + head)"),
                       "molecules_per_gpu": args.batch, "conformers": K, "atoms": n_atoms, "edges": E, "filter_pairs": P, "max_nodes": b.max_nodes,
                       "mode": args.mode, "parallelism": f"dp{world}", "fgw": "alpha=0.1 eps=0.1 max_iter=5 numItermax=5, fp64 core"},
            "roofline": {"kernel": "k_cfconv_fwd (CFConv gather * filter, CSR segment-sum)", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": "profiles/r1_pmc_hbm.json (separate rocprofv3 --pmc passes)" if traffic else None,
                         "algorithmic_bytes_per_launch": alg, "survey_convention_bytes_per_launch": cfconv_survey_bytes(E, P, n_atoms, 128),
                         "avg_launch_ms": round(kdur_ms, 5),
                         "empty_event_bracket_ms": round(empty_ms, 5),
                         "note": "avg_launch_ms is the raw HIP-event bracket around the launch (start event, kernel, end event on the launch "
                                 "stream); it contains the cost of the event packets themselves, measured live as empty_event_bracket_ms; "
                                 "rocprofv3's kernel-only duration (profiles/) is therefore shorter by about that amount. frac uses the raw bracket.",
                         "launches_timed": len(ev)},
        }
        out["roofline_other"] = other
        out["forward_only"] = fwd_extra
        if not args.no_cpu_baseline and world == 1:      # CPU oracle timed on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, args.mode, model)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
