#!/usr/bin/env python3
"""Benchmark of the ConAN hot path on MI355X:  molecules/s (K=5 conformers).

One "step" = one training step of the stage-2 hot path on one batch of synthetic ESOL-shaped conformers resident in HBM:
SchNetNoSum.forward_w_barycenter (radius graph, 3 interaction blocks, two heads, FGW barycenter over the K conformers)
+ covalent GAT branch on the 2-D bond graph + conformer-aggregation head (the reference's whole stage-2 model,
schnet_based_models.py:135-173) + MSE loss, backward through all of it, one flat gradient all-reduce (RCCL) and Adam.
Workload at every N: BASELINE.json configs[1] per GPU (ESOL + SchNet-128, K=5, batch=256) => weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|fwd] [--batch B] [--eager] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one process per GPU, before anything
touches the GPU) — both launch forms end in the same per-rank code.

Execution of the timed steps: the step is captured once into HIP graphs (forward + backward + gradient pack | all-reduce |
Adam) and the K timed steps replay them — the same kernels, no host work on the critical path; `--eager` times the plain
Python step instead, and the eager rate is always reported beside the graph rate.  Prints ONE JSON line on rank 0.
`roofline` is measured with HIP events around the CFConv gather/segment-sum kernel (the HBM-bound kernel BASELINE.json's
target is quoted on) in a separate short eager pass AFTER the timed region, so the instrumentation never sits inside the
number it annotates; `cpu_baseline` times the CPU oracle on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3      # dense fp32 matrix peak (MI355X_MICROARCH.md)
BF16_MFMA_PEAK_TF = 2500.0     # dense bf16 matrix peak (MI355X_MICROARCH.md; 2:1-sparsity figures are never used)


CONFIGS = {"cfg2": dict(shape="esol", batch=256, conformers=5, model="schnet"), "cfg3": dict(shape="lipo", batch=128, conformers=5, model="schnet"),
           "cfg4": dict(shape="bace", batch=64, conformers=5, model="visnet", head="classification"), "cfg5": dict(shape="freesolv", batch=64, conformers=20, model="schnet")}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["train", "fwd"], default="train")
    ap.add_argument("--batch", type=int, default=256, help="molecules per GPU")
    ap.add_argument("--conformers", type=int, default=5)
    ap.add_argument("--shape", default="esol")
    ap.add_argument("--model", choices=["schnet", "visnet"], default="schnet", help="backbone (BASELINE.json configs[3] = visnet + bace)")
    ap.add_argument("--head", choices=["regression", "classification"], default="regression",
                    help="classification = EmbeddingsWithGATAggregationClassificationBaryCenter (sigmoid head, BCE): SURVEY 8(d) cfg4 with --model visnet --shape bace")
    ap.add_argument("--blocks", type=int, default=3, help="timed blocks of --steps steps each (value = the median block; min / max are printed beside it)")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the world > 1 step at any world size: graph A -> all_reduce of the flat gradient buffer -> graph B (1-GPU test of the RCCL path)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the ranks: nccl (= RCCL over xGMI, one GPU per rank) or gloo (collectives through host memory; "
                         "ranks share the visible GPUs round-robin — exercises the whole world > 1 step on a ONE-GPU box, never a performance number)")
    ap.add_argument("--eager", action="store_true", help="time the eager Python step instead of the HIP-graph replay")
    ap.add_argument("--overlap", action="store_true", help="eager step: reduce the early gradient bucket while backward still runs (parallel.FlatGradients)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=48.0, help="wall-clock budget of the whole CPU-baseline leg")
    ap.add_argument("--cpu-full", action="store_true", help="SURVEY 8(d) protocol in full: 3 warm-up + 10 timed batches per leg")
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="BASELINE.json configs by name (per-GPU share): cfg2 = ESOL + SchNet, K=5, 256 molecules (the default workload); cfg3 = Lipophilicity + SchNet, "
                         "K=5, 128 per GPU (1024 over 8 GPUs); cfg4 = BACE classification + ViSNet, K=5, 64 (sigmoid head, BCE); cfg5 = FreeSolv + SchNet, K=20, 64.  Overrides --shape/--batch/--conformers/--model")
    ap.add_argument("--no-clip", action="store_true", help="leave out the global-norm gradient clipping (Trainer(gradient_clip_val=1.0), trainer.py:177) between all-reduce and Adam")
    ap.add_argument("--torch-adam", action="store_true", help="optimizer step on torch.optim.Adam(fused=True, capturable=True) instead of the one-launch FlatAdam")
    ap.add_argument("--no-pack8", action="store_true", help="skip the eight-concurrent-packer-processes leg of with_input_pipeline (it is skipped anyway under a profiler preload)")
    argv = list(sys.argv[1:] if argv is None else argv)
    a = ap.parse_args(argv)
    if a.config:
        # an explicit flag beside --config wins (e.g. --config cfg3 --batch 16 for a dry run) — told from the command line itself, so that a flag
        # that happens to equal the parser's default (--config cfg3 --batch 256) is still honoured
        given = {k for k in CONFIGS[a.config] if any(t == "--" + k or t.startswith("--" + k + "=") for t in argv)}
        for k, v in CONFIGS[a.config].items():
            if k not in given:
                setattr(a, k, v)
    return a


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int, script: str, script_args, env=None) -> int:
    """Start n ranks of `script` on this node through torch.distributed.run (one process per GPU, 127.0.0.1 rendezvous) and
    return its exit code.  Called before anything in this process has touched the GPU; the children are fresh interpreters."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(script_args)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this driver)
    e.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, n))))
    return subprocess.call(cmd, env=e)


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


# ---------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(args, gpu_model=None):
    """the CPU oracle ("port": oracle/schnet.py + the C restatement of the FGW solver) on the host cores, on a bounded
    sample of the GPU workload — the FIRST 32 molecules of rank 0's batch (same generator, same seed: identical atoms and coordinates;
    32 = the batch of BASELINE.json configs[0]): (i) FGW only, (ii) backbone only, (iii) end-to-end forward, plus the training step, each
    at 1 thread and at all cores, median over the timed batches; the training step is also timed on the configs[0] batch itself (seed
    1234 + 1).  Full protocol (--cpu-full): 3 warm-up + 10 timed batches per leg; default: the same legs inside a wall-clock budget
    (>= 1 warm-up, >= 2 timed)."""
    import numpy as np
    import torch
    from conan_fgw_amd.synthetic import make_batch, make_bond_graph
    from oracle import fgw as ofgw
    from oracle.head import Stage2Oracle
    from oracle.pyg_semantics import to_dense_adj, to_dense_batch
    from oracle.schnet import normalize_tensor

    nb, K = 32, args.conformers
    b = make_batch(args.shape, nb, K, seed=1236)                              # = the first 32 molecules of rank 0's GPU batch (make_batch draws molecule by molecule)
    bg = make_bond_graph(b, seed=2236)
    torch.manual_seed(5)
    m = Stage2Oracle(K, model_name=args.model)
    z, pos, batch = torch.from_numpy(b.z), torch.from_numpy(b.pos), torch.from_numpy(b.batch)
    bx, bei, bea = torch.from_numpy(bg.x), torch.from_numpy(bg.edge_index), torch.from_numpy(bg.edge_attr)
    y = torch.from_numpy(b.y)[:, None]
    bb = m.node_embeddings_model
    with torch.no_grad():                                                    # FGW-only inputs: what _compute_barycenter hands the solver
        _h, hb = bb.forward_3d_bary(z, pos, batch)
        ei, _ = bb.interaction_graph(pos, batch)
        dense, _ = to_dense_batch(hb, batch)
        adj = to_dense_adj(ei, batch).to(hb.dtype)
        fgw_in = []
        for i in range(nb):
            slab = dense[i * K:(i + 1) * K] + bb.FEATURE_SHIFT
            fgw_in.append((torch.stack([normalize_tensor(s, 0.1, 2.0) for s in slab]).numpy(), adj[i * K:(i + 1) * K].numpy()))

    def leg_fgw():
        for Ys, Cs in fgw_in:
            ofgw.fgw_barycenter(Ys, Cs, dtype=np.float32)

    def leg_backbone():
        with torch.no_grad():
            bb.forward_3d_bary(z, pos, batch)

    def leg_fwd():
        with torch.no_grad():
            m(z, pos, batch, bx, bei, bea)

    def leg_train():
        for p in m.parameters():
            p.grad = None
        torch.nn.functional.mse_loss(m(z, pos, batch, bx, bei, bea), y).backward()

    legs = [("fgw_only", leg_fgw), ("backbone_only", leg_backbone), ("end_to_end_forward", leg_fwd), ("train_step", leg_train)]
    try:
        all_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        all_cores = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    # "All cores": this path is op-dispatch bound (the reference's FGW loop does not scale at all) and torch's
    # intra-op pool collapses when oversubscribed (256 threads on the 256-CPU GPU host: 0.5 molecules/s against 55 at 1
    # thread, measured in round 2 with --cpu-full), so the default run times 1 thread and one 16-thread NUMA-local pool;
    # --cpu-full times 1 thread and literally all cores.  `value` is the best of the measured thread counts.
    many = all_cores if args.cpu_full else min(16, all_cores)
    configs = [1] + ([many] if many > 1 else [])
    budget = args.cpu_seconds / (len(legs) * len(configs))
    table = {}
    t_start = time.perf_counter()
    for nt in configs:
        torch.set_num_threads(nt)
        for name, fn in legs:
            timed = 10
            t0 = time.perf_counter()
            ts = []
            t1 = time.perf_counter()
            fn()
            cold = time.perf_counter() - t1
            # 3 warm-up + 10 timed.  A leg whose batch is short enough for the full protocol inside its share of the
            # budget gets it; a longer one keeps 1 warm-up and as many timed batches as fit (recorded per leg).
            full = args.cpu_full or cold * 13.5 <= budget
            warm = 3 if full else 1
            for w in range(warm - 1):
                t1 = time.perf_counter()
                fn()
                cold = time.perf_counter() - t1
            if not full and time.perf_counter() - t0 > budget:
                ts = [cold]                      # one batch already spends this leg's budget: it is the sample (noted as un-warmed)
            else:
                for _ in range(timed):
                    t1 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t1)
                    if not full and time.perf_counter() - t0 > budget:
                        break
            table[f"{name}@{nt}t"] = {"molecules_per_s": round(nb / float(np.median(ts)), 2), "timed_batches": len(ts), "warmups": warm,
                                      "warmed": not (len(ts) == 1 and ts[0] == cold)}
            print(f"[cpu_baseline] {name}@{nt}t: {table[f'{name}@{nt}t']}  ({time.perf_counter() - t_start:.1f} s)", file=sys.stderr, flush=True)
    leg = "train_step" if args.mode == "train" else "end_to_end_forward"
    best = max(configs, key=lambda nt: table[f"{leg}@{nt}t"]["molecules_per_s"])
    key = f"{leg}@{best}t"
    # the same leg on BASELINE.json configs[0] itself (its own seed): the two samples are draws of one distribution
    b1 = make_batch(args.shape, nb, K, seed=1235); bg1 = make_bond_graph(b1, seed=2235)
    z1, pos1, batch1 = torch.from_numpy(b1.z), torch.from_numpy(b1.pos), torch.from_numpy(b1.batch)
    x1, ei1, ea1, y1 = torch.from_numpy(bg1.x), torch.from_numpy(bg1.edge_index), torch.from_numpy(bg1.edge_attr), torch.from_numpy(b1.y)[:, None]

    def leg_cfg1():
        if args.mode == "train":
            for p in m.parameters():
                p.grad = None
            torch.nn.functional.mse_loss(m(z1, pos1, batch1, x1, ei1, ea1), y1).backward()
        else:
            with torch.no_grad():
                m(z1, pos1, batch1, x1, ei1, ea1)
    torch.set_num_threads(best)
    leg_cfg1()
    ts = []
    for _ in range(3):
        t1 = time.perf_counter(); leg_cfg1(); ts.append(time.perf_counter() - t1)
    table[f"{leg}_on_configs0_batch@{best}t"] = {"molecules_per_s": round(nb / float(np.median(ts)), 2), "timed_batches": len(ts), "warmups": 1, "warmed": True}
    torch.set_num_threads(default_threads)
    out = {"value": table[key]["molecules_per_s"], "unit": "molecules/s", "cores": best, "kind": "port",
           "sample": f"the first {nb} molecules of the GPU run's {args.shape.upper()}-shaped batch (same seed; {nb} = the batch of BASELINE configs[0]), K={K}: median of {table[key]['timed_batches']} "
                     f"{'training steps' if args.mode == 'train' else 'forwards'} of the CPU oracle in fp32 (SchNet trunk in torch, FGW = scalar C restatement, "
                     f"GAT + head), {best} torch thread(s) (the faster of {configs}) on {all_cores} host cores; legs = SURVEY 8(d) (i)-(iii) + training step",
           "protocol": "3 warm-up + 10 timed, median" if args.cpu_full else
                       f"per leg inside a {args.cpu_seconds:.0f} s budget: 3 warm-up + 10 timed where that fits the leg's share (legs[*].warmups == 3), "
                       "else 1 warm-up + as many timed batches as fit; median",
           "legs": table, "host_cores": all_cores, "wall_s": round(time.perf_counter() - t_start, 1)}
    if gpu_model is not None and args.model == "schnet":
        # the same sample through the HIP path and through the fp64 oracle with the HIP model's current weights
        import types
        ref = Stage2Oracle(K)
        ref.load_state_dict({k: v.detach().cpu() for k, v in gpu_model.state_dict().items()})
        ref = ref.double()
        dev = next(gpu_model.parameters()).device
        sel = b.graph_ptr[8 * K]                                              # first 8 molecules (the fp64 oracle is slow)
        zs, ps_, bs = z[:sel], pos[:sel], batch[:sel]
        eb = (bei[0] < sel) & (bei[1] < sel)
        ns = types.SimpleNamespace(z=zs.to(dev), pos=ps_.to(dev), x=bx[:sel].to(dev), edge_index=bei[:, eb].to(dev), edge_attr=bea[eb].to(dev), batch=bs.to(dev))
        with torch.no_grad():
            gy = gpu_model(ns, None, ns.batch)
            g3, gb = gpu_model.node_embeddings_model.forward_w_barycenter(ns.z, ns.pos, K, ns.batch)
            ry = ref(zs, ps_.double(), bs, bx[:sel], bei[:, eb], bea[eb])
            r3, rb = ref.node_embeddings_model.forward_w_barycenter(zs, ps_.double(), K, bs)
        rel = lambda a, r: float((a.detach().cpu().double() - r).norm() / r.norm())
        out["gpu_vs_oracle_fp64"] = {"y_pred_rel_err (energies)": float(f"{rel(gy, ry):.3e}"), "h_3d_rel_err": float(f"{rel(g3, r3):.3e}"),
                                     "h_bary_rel_err (FGW)": float(f"{rel(gb, rb):.3e}"), "tolerance": 1e-4,
                                     "sample": "first 8 molecules of the CPU sample, current weights"}
    return out


# ---------------------------------------------------------------------------------------------------------- one rank
def run_rank(args):
    # dmabuf IPC for RCCL across processes on this driver: set before the first GPU call of THIS process, whichever launcher started it
    # (spawn_ranks sets it for its children too; ranks that arrive through `python -m torch.distributed.run ... bench.py` only pass here)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.backend == "gloo":
        local = local % torch.cuda.device_count()                    # gloo ranks may share a GPU (one-GPU test of the world > 1 step)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)          # "nccl" is RCCL on ROCm: one rank per GPU over xGMI
        else:
            dist.init_process_group("gloo")

    from conan_fgw_amd import ops
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from conan_fgw_amd.parallel import FlatGradients, all_reduce_group_stream, barrier_group_stream
    from conan_fgw_amd.synthetic import make_batch, make_bond_graph
    import types

    K = args.conformers
    b = make_batch(args.shape, args.batch, K, seed=1236 + 1000 * rank)            # cfg2 seed (1234 + 2) on rank 0
    bg = make_bond_graph(b, seed=2236 + 1000 * rank)                              # 2-D bond graph of the same molecules
    # The batch reaches the device the way a training loop would deliver it: dataset items (one molecule, K conformers each)
    # through the collator — pinned pack, one H2D copy, device-side expansion (conan-fgw_amd/collate.py).  static=True: fixed
    # tensor addresses, which the captured HIP graphs below need.  The timed steps run on this resident batch.
    from conan_fgw_amd.collate import DeviceCollator, molecules_from_synthetic
    items = molecules_from_synthetic(b, bg)
    collator = DeviceCollator(dev, K, depth=4, static=True)
    data = collator(items).wait()
    z, pos, batch = data.z, data.pos, data.batch
    y = torch.from_numpy(b.y).to(dev)[:, None]
    torch.manual_seed(5)                                                          # train_val.py:223
    # the reference's stage-2 model: backbone (common.py:524-529 / :542-546) + GAT branch + aggregation head
    classify = args.head == "classification"
    if classify:                                                                  # schnet_based_models.py:308-369 (SURVEY 8(d) cfg4 with ViSNet-128)
        from conan_fgw_amd.head import EmbeddingsWithGATAggregationClassificationBaryCenter
        model = EmbeddingsWithGATAggregationClassificationBaryCenter(K, dev, model_name=args.model, feat_dim=128 if args.model == "visnet" else 512).to(dev)
        y = (y > 0).to(torch.float32)                                             # synthetic binary labels
    else:
        model = EmbeddingsWithGATAggregationBaryCenter(K, dev, model_name=args.model).to(dev)
    cidx = model.create_aggregation_index(b.num_graphs, dev)
    flat = FlatGradients(model.parameters())
    # Adam over the flat parameter / gradient / moment buffers in ONE launch (parallel.FlatAdam, checked against torch.optim.Adam step by step in
    # tests/test_gpu_stage2.py); --torch-adam keeps torch's multi-tensor kernels for the A/B
    if args.torch_adam:
        opt = torch.optim.Adam(flat.params, lr=1e-4, fused=True, capturable=True)
    else:
        from conan_fgw_amd.parallel import FlatAdam
        opt = FlatAdam(flat, lr=1e-4)
    loss_box = [torch.zeros((), device=dev)]      # the step's loss tensor itself (no copy kernel in the step): the eager step's, or the captured graph's fixed output
    train = args.mode == "train"
    clip = train and not args.no_clip
    inv_world = 1.0 / world
    collective = train and use_dist and (world > 1 or args.force_collective)      # the step contains the RCCL all-reduce

    seed = torch.full((), inv_world if collective else 1.0, device=dev)      # gradient seed of every backward pass (no scaling pass over the flat buffer afterwards)

    def fwd_bwd():
        flat.zero()
        pred = model(data, cidx, batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)
        loss = torch.nn.functional.binary_cross_entropy(pred, y) if classify else ops.mse_loss(pred, y)   # common.py: BCE / MSE (MSE and its gradient: one launch)
        flat.backward(loss, grad_scale=seed)       # = loss.backward(seed) with the slab sums of all weight gradients batched into one launch; seed = 1 / world: the SUM all-reduce yields the mean
        loss_box[0] = loss.detach()

    def eager_step():
        if train:
            fwd_bwd()
            flat.all_reduce_mean(force=args.force_collective, prescaled=collective)      # pack + (world > 1 or forced) RCCL all-reduce(s); the mean comes from the seed
            if clip:
                flat.clip_grad_norm_(1.0)                   # Lightning's gradient_clip_val=1.0 of the reference's Trainer (trainer.py:177): norm over the flat buffer, scaled in place
            opt.step()
        else:
            with torch.no_grad():
                model(data, cidx, batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)

    def barrier():
        # (collectives that precede the HIP-graph capture never run on this stream: parallel.all_reduce_group_stream)
        if use_dist and world > 1:
            barrier_group_stream(dev)
        torch.cuda.synchronize()

    def timed(step_fn, n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step_fn()
        barrier()
        dt = time.perf_counter() - t0
        local_dts.append(dt / max(1, n))
        if use_dist and world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            all_reduce_group_stream(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt
    local_dts = []                     # this rank's own seconds per step of every timed block (before the MAX over ranks)

    # Everything below runs on a side stream: HIP-graph capture needs a non-default stream, and autograd binds its
    # accumulation nodes to the stream of the first backward.
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    ev_empty = []
    ev_all = {}        # entry-point key -> [(start event, end event)] of the bracketed eager pass below (every C-ABI call of the step)
    with torch.cuda.stream(side):
        # ---- eager warm-up (also: gradient-order calibration for the overlapped all-reduce of the eager step)
        overlap = args.overlap and train and use_dist and world > 1      # opt-in: the default run keeps ONE collective per step on every path
        if overlap:
            flat.enable_overlap()
        eager_step()
        buckets = flat.calibrate() if overlap else (0, len(flat.params))
        for _ in range(max(1, args.warmup) - 1):
            eager_step()
        torch.cuda.synchronize()

        # ---- eager rate (always measured; it is `value` only with --eager)
        n_eager = args.steps if args.eager else max(3, min(args.steps, 10))
        # (without --eager this rate is informational: three blocks and their median as well — a single block right behind the warm-up carried the
        # process's cold-start hiccups, 4 to 15 ms per "step" on a fresh box for a step that takes 2.7)
        dt_eager_blocks = [timed(eager_step, n_eager) for _ in range(max(1, args.blocks) if args.eager else 3)]
        dt_eager = float(np.median(dt_eager_blocks))
        dt_blocks = dt_eager_blocks
        loss_eager = float(loss_box[0])

        # ---- HIP-graph capture of the same step: A = forward + backward + pack | all-reduce (eager RCCL call between the two
        # replays, world > 1 only) | B = mean + Adam.  Same kernels as the eager step, zero host work between them.
        dt_graph, graph_err, host_launch_ms = None, None, None
        allreduce_exposed_us, local_value_ms, loss_after_blocks = None, None, None
        if not args.eager:
            flat.suspend_overlap(True)
            # thread_local: only this thread's calls are policed during capture.  Other threads of the process make legal HIP calls
            # meanwhile (ProcessGroupNCCL's watchdog polls its work events, the collator's copy thread) which "global" would turn into
            # a failed capture.
            cmode = "thread_local"
            captured = False
            try:
                barrier()                                            # every rank enters capture with its queue and its process group idle
                gA = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gA, stream=side, capture_error_mode=cmode):
                    if train:
                        fwd_bwd()
                        flat.pack()
                    else:
                        with torch.no_grad():
                            model(data, cidx, batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)
                gB = None
                if train:
                    gB = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gB, stream=side, pool=gA.pool(), capture_error_mode=cmode):
                        if clip:
                            flat.clip_grad_norm_(1.0)
                        opt.step()                                   # (the gradients were seeded with 1 / world: the summed buffer is the mean)
                captured = True
            except Exception as e:                                   # capture is an optimisation of the launch path, never a requirement
                graph_err = f"{type(e).__name__}: {e}"[:300]
                import traceback
                print(f"[bench rank {rank}] HIP-graph capture failed, falling back to the eager step:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
                torch.cuda.synchronize()
            if use_dist and world > 1:
                # all ranks replay or none does: a rank stepping eagerly issues a different sequence of collectives
                ok = torch.tensor([1.0 if captured else 0.0], device=dev)
                all_reduce_group_stream(ok, op=dist.ReduceOp.MIN)
                if captured and ok.item() == 0.0:
                    captured, graph_err = False, "capture failed on another rank"

            def graph_step():
                gA.replay()
                if train:
                    if collective:
                        # synchronous form: the collective runs on THIS stream between the two replays (no event hop to the group's
                        # stream and back).  Safe here because nothing is captured on this stream any more (parallel.py).
                        dist.all_reduce(flat.flat, op=dist.ReduceOp.SUM)
                    gB.replay()
            host_launch_ms = None
            if captured:
                for _ in range(max(2, args.warmup)):
                    graph_step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    graph_step()
                host_launch_ms = 1e3 * (time.perf_counter() - t0) / 5      # host time to ENQUEUE one step's replays (the queue is empty: nothing blocks)
                torch.cuda.synchronize()
                n0 = len(local_dts)
                dt_blocks = [timed(graph_step, args.steps) for _ in range(max(1, args.blocks))]
                dt_graph = float(np.median(dt_blocks))
                local_value_ms = 1e3 * float(np.median(local_dts[n0:]))
                loss_after_blocks = float(loss_box[0])
            flat.suspend_overlap(False)
        loss_last = loss_after_blocks if loss_after_blocks is not None else float(loss_box[0])
        # every rank has applied the same averaged gradients: the parameters must agree across ranks bit for bit, the losses are the
        # ranks' own (different shards)
        per_rank = None
        if use_dist:
            mine = torch.stack([loss_box[0].double().reshape(()), torch.cat([p.detach().reshape(-1) for p in flat.params]).double().sum()])
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            torch.cuda.synchronize()
            lv = local_value_ms if local_value_ms is not None else 1e3 * dt_eager / n_eager
            mine_t = torch.tensor([lv], dtype=torch.float64, device=dev)
            allt = [torch.zeros_like(mine_t) for _ in range(world)]
            dist.all_gather(allt, mine_t)
            torch.cuda.synchronize()
            ms = [float(t[0]) for t in allt]
            per_rank = {"loss": [float(t[0]) for t in allr], "parameter_checksum": [float(t[1]) for t in allr],
                        "ms_per_step": [round(v, 4) for v in ms], "ms_per_step_min": round(min(ms), 4), "ms_per_step_max": round(max(ms), 4),
                        "ms_per_step_note": "each rank's own clock around its median timed block (value uses the MAX over ranks per block); a wide max / min spread "
                                            "points at one slow rank (host contention, a throttled GPU), a uniform rise over the 1-rank step at the collective"}

        # ---- what the collective costs the step where it sits (between the two replays, nothing overlapped): the same replays without it.  Timing
        # only, and only here — AFTER the per-rank checksums were taken: without the all-reduce every rank applies its own gradients, so from
        # this point on the ranks' parameters differ (nothing below depends on them being equal)
        if collective and dt_graph is not None:
            def graph_step_no_collective():
                gA.replay(); gB.replay()
            dt_nc = timed(graph_step_no_collective, args.steps)
            allreduce_exposed_us = 1e6 * (dt_graph - dt_nc) / args.steps

        # ---- the same step fed by the input pipeline: every step re-collates the batch on the host (C pack into a pinned
        # buffer), copies it (one H2D transfer on the copy stream, overlapping the previous step) and expands it on the device.
        # PCIe-inclusive rate: reported beside `value`, never as `value`.
        pipe = None
        try:
            from conan_fgw_amd.collate import CollatePipeline
            import itertools, threading
            step_fn = graph_step if dt_graph is not None else eager_step
            feed = CollatePipeline(collator, itertools.repeat(items), prefetch=2)      # host half on a worker thread, two batches ahead

            def pipe_step():
                next(feed).wait()
                step_fn()
            for _ in range(3):
                pipe_step()
            n_pipe = max(5, args.steps)
            dt_pipe = timed(pipe_step, n_pipe)
            feed.close()
            torch.cuda.synchronize()
            time.sleep(0.05)                                             # (the worker finishes the pack it was in)
            t0 = time.perf_counter()
            for _ in range(5):
                collator.pack(items)
            host_ms = 1e3 * (time.perf_counter() - t0) / 5
            host_cold_ms = None
            try:                                                         # the same pack with no cached item records (the cached figure is a 100 %-hit best case)
                import copy as _copy
                t0 = time.perf_counter()
                for _ in range(3):
                    cold_items = [_copy.copy(it) for it in items]
                    for it in cold_items:
                        it.__dict__.pop("_conan_record", None)
                    collator.pack(cold_items)
                host_cold_ms = round(1e3 * (time.perf_counter() - t0) / 3, 4)
            except Exception as e:
                print(f"[bench] cold-item pack leg skipped: {type(e).__name__}: {e}", file=sys.stderr)
            # eight packers at once on this host — what the eight ranks of one node do: eight fresh processes (no GPU), each packing the same
            # batch 20 times into host memory through the same C entry points; the slowest one's mean is reported
            pack8 = None
            profiled = any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
            if rank == 0 and world == 1 and not args.no_pack8 and not profiled:
                child_env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
                code = ("import sys, time, ctypes, numpy as np; sys.path.insert(0, %r); "
                        "from conan_fgw_amd.synthetic import make_batch, make_bond_graph; from conan_fgw_amd import collate as C; "
                        "b = make_batch(%r, %d, %d, seed=1236); items = C.molecules_from_synthetic(b, make_bond_graph(b, seed=2236)); "
                        "t = C.host_pack_benchmark(items, %d, 20); print('PACKMS', t)") % (ROOT, args.shape, args.batch, K, K)
                procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=child_env) for _ in range(8)]
                vals = []
                for pr in procs:
                    out_, _ = pr.communicate(timeout=120)
                    vals += [float(l.split()[1]) for l in out_.splitlines() if l.startswith("PACKMS")]
                if len(vals) != 8:
                    print(f"[bench] eight-packer leg: {len(vals)} of 8 children reported (exit codes {[pr.returncode for pr in procs]})", file=sys.stderr)
                pack8 = round(max(vals), 4) if len(vals) == 8 else None
            pipe = {"molecules_per_s": round(args.batch * world * n_pipe / dt_pipe, 1), "ms_per_step": round(1e3 * dt_pipe / n_pipe, 4), "steps": n_pipe,
                    "packed_bytes_per_batch": collator.last_packed_bytes, "host_pack_ms": round(host_ms, 4),
                    "host_pack_ms_8_concurrent_processes": pack8,
                    "host_pack_note": "host_pack_ms is for dataset items whose validated arrays are cached on the item (a dataset hands the same objects out every epoch); "
                                      "host_pack_ms_cold_items re-validates every array of every item (first epoch / reference-style items rebuilt per batch)",
                    "host_pack_ms_cold_items": host_cold_ms,
                    "what": "per step: collate on a worker thread (item records cached, C pack into a pinned ring, then the H2D copy and the expansion kernel enqueued by "
                            "that thread once the GPU has released the landing copy: no device-side wait is queued ahead) + one device-to-device transfer to the fixed "
                            "addresses the captured step reads + the step; copy and expansion overlap the previous step"}
        except Exception as e:
            pipe = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.synchronize()

        # ---- all-reduce cost by itself (world > 1): the flat buffer, 20 back-to-back calls
        ar_us = None
        if train and use_dist:
            probe = torch.zeros_like(flat.flat)
            for _ in range(3):
                dist.all_reduce(probe)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                dist.all_reduce(probe)
            torch.cuda.synchronize()
            ar_us = 1e6 * (time.perf_counter() - t0) / 20

        # ---- forward-only rate of the same workload (this rank's shard)
        fwd_extra = None
        if train:
            def fwd_step():
                with torch.no_grad():
                    model(data, cidx, batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)
            for _ in range(2):
                fwd_step()
            torch.cuda.synchronize()
            nf = max(5, args.steps // 2)
            t0 = time.perf_counter()
            for _ in range(nf):
                fwd_step()
            torch.cuda.synchronize()
            tf = (time.perf_counter() - t0) / nf
            fwd_extra = {"molecules_per_s_per_gpu": round(args.batch / tf, 1), "ms_per_step": round(1e3 * tf, 4), "execution": "eager"}

        # ---- per-entry-point HIP-event brackets: a separate short eager pass, outside every timed region.  EVERY C-ABI call of the step is
        # bracketed (set_call_trace sits inside _lib.call, so no module's own binding of `call` escapes it): the roofline entries below name
        # the entry points they need and the run FAILS when one of them never fired (round 4 silently lost two entries to renamed calls).
        from conan_fgw_amd import _lib as _cl
        n_trace = 10

        n_atoms_rows = int(z.shape[0])

        def edge_rows(m):
            return m > 3 * n_atoms_rows                                       # node-level calls see n or (ViSNet's vector channels) 3 n rows, edge-level ones one row per edge

        def trace_key(name, a):
            if name == "conan_linear_fwd":
                return name + (":edge" if edge_rows(a[4]) else ":node")
            if name == "conan_linear_multi_fwd":
                return name + (":edge" if edge_rows(a[3]) else ":node")
            if name == "conan_linear_sum_fwd":
                return name + (":edge" if edge_rows(a[7]) else ":node")
            return name

        def tracer(name, fn, a):
            key = trace_key(name, a)
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(); rc = fn(*a); e_.record()
            ev_all.setdefault(key, []).append((s_, e_))
            if key == PRIMARY or key == "conan_filter_cfconv_fwd":   # an EMPTY bracket right behind it: what two event packets cost on this queue by themselves
                s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record(); e0.record()
                ev_empty.append((s0, e0))
            return rc
        PRIMARY = "conan_cfconv_fwd" if args.model == "schnet" else "conan_visnet_attn_message"
        _cl.set_call_trace(tracer)
        try:
            for _ in range(n_trace):
                eager_step()
            torch.cuda.synchronize()
        finally:
            _cl.set_call_trace(None)

        # edge statistics of this rank's batch (reporting only)
        gp = ops.graph_ptr_from_batch(batch, b.num_graphs)
        cutoff = 10.0 if args.model == "schnet" else 5.0
        _g = ops.RadiusGraph(pos, gp, b.num_graphs, cutoff, 32, loop=args.model != "schnet")
        E = _g.num_edges
        P = int(_g.pairs().num_pairs_dev.item()) if args.model == "schnet" else E
        # the same kernel with NOTHING cached: a 1 GiB fill in front of every launch evicts the filter tensor from L2 and the 256 MiB
        # Infinity Cache (in the step it was written by the kernel just before and is partly served from there)
        cold_ms = None
        if args.model == "schnet":
            try:
                xc = torch.randn(int(z.shape[0]), 128, device=dev); Wc = torch.randn(_g.max_edges, 128, device=dev); oc = torch.empty_like(xc)
                big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
                ts = []
                for _ in range(12):
                    big.fill_(1.0)
                    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s_.record()
                    ops.call("conan_cfconv_fwd", ops.ptr(xc), ops.ptr(Wc), ops.ptr(_g.rowptr), ops.ptr(_g.col), ops.ptr(_g.pid), int(z.shape[0]), 128, ops.ptr(oc), None, ops.stream_ptr())
                    e_.record(); torch.cuda.synchronize()
                    ts.append(s_.elapsed_time(e_))
                cold_ms = float(np.median(ts[2:]))
                del big, xc, Wc, oc
            except Exception as e:
                print(f"[bench] cold-cache CFConv pass skipped: {type(e).__name__}: {e}", file=sys.stderr)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    use_graph = dt_graph is not None
    dt, steps = (dt_graph, args.steps) if use_graph else (dt_eager, n_eager)
    n_atoms = int(z.shape[0])
    mean_ms = lambda pairs: float(np.mean([s.elapsed_time(e) for s, e in pairs])) if pairs else float("nan")
    ev = ev_all.get(PRIMARY, [])
    kdur_ms, empty_ms = mean_ms(ev), mean_ms(ev_empty)
    if empty_ms != empty_ms:
        empty_ms = 0.0                                               # (no bracketed launch of the roofline kernel: reported as missing below)
    default_cfg2 = args.shape == "esol" and args.batch == 256 and K == 5 and args.model == "schnet"

    def committed_pmc(kernel_prefix, files):
        """HBM traffic of a kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs, profiles/): only
        quoted for the workload those passes ran (the default cfg2 batch)."""
        if not default_cfg2:
            return None, None
        for name in files:
            try:
                ks = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
                kk = [k for k in ks if k.startswith(kernel_prefix)][0]
                return int(ks[kk]["traffic_bytes_corrected"]), f"profiles/{name} (separate rocprofv3 --pmc passes)"
            except Exception:
                continue
        return None, None

    # every entry the line promises must have fired in the bracketed pass: fail loudly instead of dropping it
    F_ = 128
    fused_bwd = "conan_filter_bwd2" if "conan_filter_bwd2" in ev_all else "conan_filter_bwd"
    fgw_entry = "conan_fgw_barycenter_fwd_ragged" if "conan_fgw_barycenter_fwd_ragged" in ev_all else "conan_fgw_barycenter_fwd"
    # inference on batches whose filter tensor outgrows the Infinity Cache takes the fused generator + gather (schnet.py): one entry point then
    # stands where conan_filter_fwd and conan_cfconv_fwd stand otherwise
    fused_fwd = args.model == "schnet" and not train and "conan_filter_cfconv_fwd" in ev_all and "conan_cfconv_fwd" not in ev_all
    if fused_fwd:
        PRIMARY = "conan_filter_cfconv_fwd"
        ev = ev_all[PRIMARY]
        kdur_ms = mean_ms(ev)
    required = [PRIMARY, fgw_entry] + (([] if fused_fwd else ["conan_filter_fwd"]) + ([fused_bwd] if train else []) if args.model == "schnet" else ["conan_linear_multi_fwd:edge"])
    missing = [k for k in required if not ev_all.get(k)]
    if missing:
        raise RuntimeError(f"bench.py: the bracketed eager pass never saw {missing} (entry points seen: {sorted(ev_all)}): the roofline entries "
                           "would be silently incomplete — fix the hook names")

    roofline = None
    if fused_fwd:
        issued = E * 3 * 2.0 * (64 * 128 + 128 * 128)               # one filter row per DIRECTED edge, 3 fp16 partial products per fp32 product, Gs padded to 64
        alg = 12 * E + 2 * n_atoms * 4 * F_                         # distance + two indices per edge, x in and out once
        roofline = {"kernel": "k_filter_fused<128, true> (filter rows generated per directed edge and consumed by the CFConv gather in the same launch: no [E,F] tensor)",
                    "entry_point": PRIMARY, "bound": "mfma", "achieved": round(issued / (kdur_ms * 1e-3) / 1e12, 1), "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(issued / (kdur_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4), "traffic": None, "traffic_source": None,
                    "algorithmic_bytes_per_launch": alg, "hbm_gbs_of_algorithmic_bytes": round(alg / (kdur_ms * 1e-3) / 1e9, 1),
                    "avg_launch_ms": round(kdur_ms, 5), "empty_event_bracket_ms": round(empty_ms, 5), "launches_timed": len(ev),
                    "note": "inference path of batches whose pair-shared filter tensor would outgrow the Infinity Cache (DESIGN 3.1c): the CFConv gather is the epilogue "
                            "of the filter generator here, so the kernel is bound by the generator's dependent chain (matrix + vector issue), not by HBM; `achieved` "
                            "counts the issued fp16 FLOP against the dense 16-bit peak.  The training step and smaller batches run k_cfconv_fwd (see `cold` for that "
                            "kernel on this batch's graph)",
                    "cold": None if not cold_ms else {"kernel": "k_cfconv_fwd on the same graph, nothing cached", "avg_launch_ms": round(cold_ms, 5),
                                                       "achieved_gbs": round(cfconv_algorithmic_bytes(E, P, n_atoms, F_) / (cold_ms * 1e-3) / 1e9, 1),
                                                       "frac_of_hbm_peak": round(cfconv_algorithmic_bytes(E, P, n_atoms, F_) / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    elif args.model == "schnet":
        traffic, traffic_src = committed_pmc("k_cfconv_fwd", ("r6_pmc_hbm.json", "r5_pmc_hbm.json", "r4_pmc_hbm.json", "r3_pmc_hbm.json"))
        alg = cfconv_algorithmic_bytes(E, P, n_atoms, F_)
        achieved = alg / (kdur_ms * 1e-3) / 1e9
        roofline = {"kernel": "k_cfconv_fwd (CFConv gather * filter, CSR segment-sum)", "entry_point": PRIMARY, "bound": "hbm",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": alg, "survey_convention_bytes_per_launch": cfconv_survey_bytes(E, P, n_atoms, F_),
                    "avg_launch_ms": round(kdur_ms, 5), "empty_event_bracket_ms": round(empty_ms, 5),
                    "cold": None if not cold_ms else {"avg_launch_ms": round(cold_ms, 5), "achieved": round(alg / (cold_ms * 1e-3) / 1e9, 1),
                                                       "frac": round(alg / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                       "what": "same kernel, same graph, 1 GiB written in front of every launch (L2 and Infinity Cache evicted); median of 10 raw brackets"},
                    "note": "avg_launch_ms is the raw HIP-event bracket around the launch (start event, kernel, end event on the launch "
                            "stream), taken in a separate eager pass after the timed region; it contains the cost of the event packets "
                            "themselves (empty_event_bracket_ms), so rocprofv3's kernel-only duration (profiles/) is shorter by about that "
                            "amount. frac uses the raw bracket and the compulsory bytes.",
                    "launches_timed": len(ev)}
    else:
        # ViSNet: the attention message + scalar aggregation kernel (ViS_MP.message / aggregate, torch_geometric_visnet.py:632-645,671) — every
        # [E,H] tensor once: dk, dv in, vmsg out; the node rows q, k, v in and xagg out once (the E gathers of k_j / v_j are re-reads of n rows)
        alg = 4 * F_ * (3 * E + 4 * n_atoms) + 8 * E + 4 * (n_atoms + 1)
        achieved = alg / (kdur_ms * 1e-3) / 1e9
        roofline = {"kernel": "k_attn_msg (ViS_MP attention message + scalar aggregation, one wavefront per target)", "entry_point": PRIMARY, "bound": "hbm",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": None, "traffic_source": None, "algorithmic_bytes_per_launch": alg,
                    "avg_launch_ms": round(kdur_ms, 5), "empty_event_bracket_ms": round(empty_ms, 5), "launches_timed": len(ev),
                    "note": "raw HIP-event bracket around the launch in a separate eager pass (see empty_event_bracket_ms); compulsory bytes: "
                            "4H(3E + 4n) + 8E + 4(n+1)"}
    other = []
    if ev_all.get("conan_filter_fwd"):
        t_ms = mean_ms(ev_all["conan_filter_fwd"])
        fl = P * 2.0 * (50 * 128 + 128 * 128)                       # (2*Gs*F + 2*F*F) per filter row; P rows (pairs)
        issued = P * 3 * 2.0 * (64 * 128 + 128 * 128)              # what the matrix pipe executes: 3 fp16 partial products per fp32 product (two planes per operand), Gs padded to 64
        other.append({"kernel": "k_filter_fused (rbf -> filter MLP -> cosine cutoff)", "entry_point": "conan_filter_fwd", "bound": "mfma", "achieved": round(issued / (t_ms * 1e-3) / 1e12, 1),
                      "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(issued / (t_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4), "avg_launch_ms": round(t_ms, 5),
                      "algorithmic_fp32_tflops": round(fl / (t_ms * 1e-3) / 1e12, 2),
                      "note": "both GEMMs run on two fp16 planes per operand (3 x v_mfma_f32_32x32x16_f16 per 32x32x16 block, fp32-class result): `achieved` counts the "
                              "issued fp16 FLOP against the dense 16-bit peak; algorithmic_fp32_tflops is the fp32-equivalent rate (2*(50*F + F*F) per row) against the "
                              "fp32 MFMA peak 157.3 TF/s.  The kernel sits on its two 132 MB output streams (W and, in training, h1): hbm_gbs_of_output below"})
        other[-1]["hbm_gbs_of_output"] = round(P * 128 * 4 * (2 if train else 1) / (t_ms * 1e-3) / 1e9, 1)
    if ev_all.get(fused_bwd):
        t_ms = mean_ms(ev_all[fused_bwd])
        one_pass = fused_bwd == "conan_filter_bwd2"
        byts = P * (2 * 4 * 128 + 4)                                # compulsory: g and h1 rows in, distances in; the weight-gradient slabs are noise
        issued = P * 3 * 2.0 * (128 * 128 + 64 * 128 + (128 * 128 if one_pass else 0))      # dx GEMM + dh1^T rbf (Gs padded to 64) (+ dw2 = g^T h1 in the one-pass kernel), 3 fp16 partial products each
        other.append({"kernel": ("k_filter_bwd2 (filter-network backward in ONE pass over g and h1: dw2 = g^T h1, dh1 = (g w2) * ssp'(h1) in registers, dw1 = dh1^T rbf)" if one_pass else
                                 "k_filter_bwd (filter-network backward below the second Linear)"), "entry_point": fused_bwd, "bound": "hbm",
                      "achieved": round(byts / (t_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(byts / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "avg_launch_ms": round(t_ms, 5), "issued_fp16_tflops": round(issued / (t_ms * 1e-3) / 1e12, 1), "algorithmic_bytes_per_launch": byts,
                      "note": "event bracket around kernel + slab write (the slab reduction is batched elsewhere); two fp16 planes per operand with the gradient scaled "
                              "from its device-side maximum; eight wavefronts per 32-row tile (DESIGN 3.1b: bound by its staging instructions, 61 us stream floor at cfg2)"})
    if ev_all.get("conan_cfconv_bwd_xw_pairs"):
        t_ms = mean_ms(ev_all["conan_cfconv_bwd_xw_pairs"])
        byts = 2 * P * 4 * F_ + 3 * n_atoms * 4 * F_ + 4 * (3 * E + 3 * P + n_atoms + 1)      # W rows in, pair-gradient rows out, x / dout in and dx out once, indices
        other.append({"kernel": "k_cfconv_bwd_xw128 (CFConv backward: dx and the pair gradient from one walk of the by-source CSR)", "entry_point": "conan_cfconv_bwd_xw_pairs",
                      "bound": "hbm", "achieved": round(byts / (t_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(byts / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "avg_launch_ms": round(t_ms, 5), "algorithmic_bytes_per_launch": byts,
                      "note": "raw HIP-event bracket; the filter tensor it reads was written a forward pass earlier (cold), the pair gradient it writes is read back by k_filter_bwd2"})
    if ev_all.get("conan_linear_multi_fwd:edge") or ev_all.get("conan_linear_fwd:edge"):
        # edge-level Linear work of ViSNet (two fp16 planes per operand): the projections of one f in one launch (k_linear_fan16), their input gradients
        # summed in the accumulators (k_linear_sum16), s_proj and its 256-wide contraction (k_linear_t16 / k_linear_sum16<2> behind conan_linear_fwd)
        U = E * 128 * 4                                                     # one [E,128] fp32 tensor
        for key, kern, units in (("conan_linear_multi_fwd:edge", "k_linear_fan16<NL> (the 128 -> 128 layers of one input, one workgroup per tile and output half)",
                                  lambda n: 4 * (n - 1) + 3),               # per step: n - 1 launches of three layers (1 U in, 3 U out), the last layer's two (1 + 2)
                                 ("conan_linear_sum_fwd:edge", "k_linear_sum16<NCH> (sum of the input gradients of those layers + the handed-through gradient, in the accumulators)",
                                  lambda n: 5 * (n - 1) + 3),               # 3 U in + seed + 1 U out; last layer: 2 in, no seed, 1 out
                                 ("conan_linear_fwd:edge", "k_linear_t16<128,128> / k_linear_sum16<2> (s_proj forward 128 -> 256, its input gradient 256 -> 128, the rbf projections)", None)):
            if not ev_all.get(key):
                continue
            t_ms = mean_ms(ev_all[key])
            n_step = len(ev_all[key]) / n_trace
            ent = {"kernel": kern, "entry_point": key, "bound": "hbm", "avg_launch_ms": round(t_ms, 5), "launches_per_step": round(n_step, 1), "launches_timed": len(ev_all[key])}
            if units is not None and n_step >= 2:
                byts = units(int(round(n_step))) * U
                ent.update({"algorithmic_bytes_per_step": byts, "achieved": round(byts / (t_ms * n_step * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(byts / (t_ms * n_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
            else:
                ent["note"] = "mixed shapes under one entry point: avg_launch_ms is the mean over the step's edge-level calls; per-launch figures: profiles/r5_visnet_edge_linears.txt"
            other.append(ent)
    if ev_all.get(fgw_entry):
        t_ms = mean_ms(ev_all[fgw_entry])
        N_, d_ = b.max_nodes, 64
        fgw_bytes = 4 * (2 * K * N_ ** 2 + K * N_ * d_ + N_ * d_ + N_ ** 2)
        fgw_flop = 5 * K * 5 * (4 * N_ ** 3 + 5 * 12 * N_ ** 2)    # outer 5 x K x PGD 5 x (4N^3 + Sinkhorn 5 x ~12N^2), worst case
        fgw_pmc = None                                                  # counters of the coupling kernel from the committed PMC passes (profiles/)
        if default_cfg2:
            for rr in ("r6", "r5", "r4", "r3"):
                try:
                    sq = json.load(open(os.path.join(ROOT, "profiles", f"{rr}_fgw_pmc_sq.json")))["kernels"]
                    hb = json.load(open(os.path.join(ROOT, "profiles", f"{rr}_fgw_pmc_hbm.json")))["kernels"]
                    kk = [k for k in sq if k.startswith("k_fgw_coupling_fast")][0]
                    fgw_pmc = {"kernel": kk, "valu_issue_frac": sq[kk].get("valu_issue_frac"), "mfma_busy_frac": sq[kk].get("mfma_busy_frac"),
                               "avg_launch_us_under_pmc": sq[kk].get("avg_duration_us_under_pmc"),
                               "hbm_traffic_bytes_per_launch": hb.get(kk, {}).get("traffic_bytes_corrected"),
                               "source": f"profiles/{rr}_fgw_pmc_sq.json, profiles/{rr}_fgw_pmc_hbm.json (separate rocprofv3 --pmc passes over tools/fgw_pmc.py)"}
                    break
                except Exception:
                    fgw_pmc = None
        other.append({"kernel": "FGW barycenter, whole batched solve (init + 5 x (coupling + second pass + update)); N <= 64: k_fgw_coupling_fast, N > 64: k_fgw_coupling_big",
                      "entry_point": fgw_entry, "bound": "fp64 vector issue / latency",
                      "coupling_kernel_counters": fgw_pmc,
                      "avg_ms": round(t_ms, 4), "us_per_molecule": round(1e3 * t_ms / args.batch, 3),
                      "algorithmic_bytes_per_molecule": fgw_bytes, "worst_case_flop_per_molecule": fgw_flop,
                      "achieved_fp64_tflops_upper": round(args.batch * fgw_flop / (t_ms * 1e-3) / 1e12, 3), "fp64_peak_tflops": 78.6,
                      "achieved": round(args.batch * fgw_flop / (t_ms * 1e-3) / 1e12, 3), "peak": 78.6, "unit": "TFLOP/s (fp64, worst-case FLOP count: an upper bound)",
                      "frac": round(args.batch * fgw_flop / (t_ms * 1e-3) / 1e12 / 78.6, 4),
                      "hbm_gbs_of_algorithmic_bytes": round(args.batch * fgw_bytes / (t_ms * 1e-3) / 1e9, 1)})

    # where the step goes, by C-ABI entry point: the same bracketed eager pass, the cost of an empty bracket subtracted per call.  The brackets of
    # the covalent branch (a second stream) overlap the main stream's, so the shares are of the SUM of all brackets, not of the wall time.
    KERNELS = {"conan_fgw_barycenter_fwd_ragged": "k_fgw_small_vectors + 5 x (k_fgw_coupling_fast | k_fgw_coupling_big, second pass, k_fgw_update_parts)",
               "conan_filter_fwd": "k_filter_fused", "conan_filter_cfconv_fwd": "k_filter_fused<128, true> (generator + gather)", "conan_filter_bwd2": "k_filter_bwd2", "conan_cfconv_fwd": "k_cfconv_fwd", "conan_cfconv_bwd_x": "k_cfconv_bwd_x128",
               "conan_cfconv_bwd_w_pairs": "k_cfconv_bwd_wp128", "conan_cfconv_bwd_xw_pairs": "k_cfconv_bwd_xw128 (dx + pair gradient in one launch)", "conan_linear_wgrad_slabs_batch": "k_wgrad_lds_batch / k_wgrad_lds_shared<3>", "conan_wgrad_reduce_batch": "k_wgrad_reduce4_batch",
               "conan_mlp2_fwd": "k_mlp2", "conan_mlp2_bwd": "k_mlp2", "conan_linear_fwd:node": "k_linear_t16 (node level)", "conan_linear_fwd:edge": "k_linear_t16 (edge level)",
               "conan_linear_multi_fwd:edge": "k_linear_fan16 (edge level, layers of one input)", "conan_linear_sum_fwd:edge": "k_linear_sum16 (edge level, summed input gradients)",
               "conan_linear_sum_fwd:node": "k_linear_sum16 (node level)", "conan_linear_wgrad_slabs": "k_wgrad_lds / k_wgrad_lds_shared<2>", "conan_visnet_attn_message": "k_attn_msg",
               "conan_visnet_attn_message_bwd": "k_attn_bwd_target + k_attn_bwd_source", "conan_visnet_vec_aggregate": "k_vec_aggregate",
               "conan_visnet_vec_aggregate_bwd": "k_vec_aggregate_bwd_s + _v", "conan_visnet_edge_update": "k_edge_update", "conan_visnet_edge_update_bwd": "k_edge_update_bwd_t + _s",
               "conan_radius_graph_csr": "k_radius + k_exclusive_scan", "conan_linear_wgrad_scaled": "k_wgrad_lds_h16"}
    tot = {k: max(0.0, sum(s_.elapsed_time(e_) for s_, e_ in v) - (empty_ms if empty_ms == empty_ms else 0.0) * len(v)) / n_trace for k, v in ev_all.items()}
    tot_all = sum(tot.values()) or 1.0
    kernel_share = {"what": "top entry points of ONE eager step by bracketed GPU time (HIP events around every C-ABI call, empty-bracket cost subtracted; the covalent "
                            "branch's brackets run on a second stream and overlap the rest, so `share` is of the sum of all brackets)",
                    "sum_of_brackets_ms": round(tot_all, 4), "entry_points_seen": len(ev_all),
                    "top": [{"entry_point": k, "kernels": KERNELS.get(k), "calls_per_step": round(len(ev_all[k]) / n_trace, 1), "ms_per_step": round(t, 4), "share": round(t / tot_all, 4)}
                            for k, t in sorted(tot.items(), key=lambda kv: -kv[1])[:8]]}
    if rank == 0:
        mol = args.batch * world * steps
        exe = (("HIP-graph replay (fwd+bwd+pack | RCCL all-reduce | clip + Adam)" if clip else "HIP-graph replay (fwd+bwd+pack | RCCL all-reduce | Adam)") if train else "HIP-graph replay") if use_graph else "eager"
        out = {
            "metric": "molecules/s (K=5 conformers)", "value": round(mol / dt, 1), "unit": "molecules/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / steps, 4),
            "blocks": {"n": len(dt_blocks), "steps_each": steps, "ms_per_step": [round(1e3 * t / steps, 4) for t in dt_blocks],
                       "min": round(1e3 * min(dt_blocks) / steps, 4), "max": round(1e3 * max(dt_blocks) / steps, 4),
                       "note": "each block = exactly --steps steps between barrier + synchronize, max over ranks; value / ms_per_step = the median block"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.shape.upper()}-shaped + " + ("ViSNet-128 (6 layers, 8 heads, 32 RBF, cutoff 5 A), " if args.model == "visnet" else "SchNet-128 (3 interactions, 50 gaussians, cutoff 10 A, cap 32), ")
                                   + f"K={K}, batch={args.batch} molecules per GPU, {args.mode} step "
                                   + ("classification head (sigmoid, BCE) " if classify else "") + ("(stage-2 model incl. GAT branch: fwd + bwd + flat-gradient all-reduce" + (" + global-norm clip 1.0" if clip else "") + " + Adam)" if train else "(forward_w_barycenter + GAT branch + head)"),
                       "molecules_per_gpu": args.batch, "conformers": K, "atoms": n_atoms, "edges": E, "filter_pairs": P, "max_nodes": b.max_nodes,
                       "mode": args.mode, "parallelism": f"dp{world}", "execution": exe, "fgw": "alpha=0.1 eps=0.1 max_iter=5 numItermax=5, fp64 core; exact reformulations of the same iteration: the exchangeable padded nodes of a conformer graph / of the barycenter solved as one node, products against a complete graph's structure matrix as row sums (DESIGN.md 3.3 round 6; iteration counts and results = the full-size solve's)",
                       "optimizer": "torch.optim.Adam(fused, capturable)" if args.torch_adam else "Adam in one launch over flat buffers (parallel.FlatAdam: torch.optim.Adam's update)",
                       "grad_clip": "global L2 norm 1.0 on the flat buffer (conan_grad_clip_flat; Trainer(gradient_clip_val=1.0), trainer.py:177)" if clip else None},
            "rccl_ranks": dist.get_world_size() if use_dist and args.backend == "nccl" else 0,
            "dist": {"backend": ("rccl (torch.distributed 'nccl')" if args.backend == "nccl" else "gloo (host memory; ranks may share a GPU: a functional run of the world > 1 step, not a rate)") if use_dist else None,
                     "ranks": world, "distinct_gpus": min(world, torch.cuda.device_count()) if args.backend == "gloo" else world, "per_rank": per_rank},
            "allreduce_us": None if ar_us is None else round(ar_us, 1),
            "allreduce_exposed_us": None if allreduce_exposed_us is None else round(allreduce_exposed_us, 1),
            "allreduce": {"payload_bytes": int(flat.flat.numel()) * 4, "calls_per_step": (1 if collective else 0) if use_graph else flat.last_allreduce_launches,
                          "in_timed_step": bool(collective), "forced": bool(args.force_collective and world == 1),
                          "eager_overlap_buckets": list(buckets)},
            "eager": {"molecules_per_s": round(args.batch * world * n_eager / dt_eager, 1), "ms_per_step": round(1e3 * dt_eager / n_eager, 4), "steps": n_eager,
                      "blocks_ms_per_step": [round(1e3 * t / n_eager, 4) for t in dt_eager_blocks]},
            "with_input_pipeline": pipe,
            "graph_capture_error": graph_err,
            "graph_launch_host_ms": None if (args.eager or host_launch_ms is None) else round(host_launch_ms, 4),
            "loss": {"after_eager_phase": loss_eager, "last": loss_last},
            "roofline": roofline,
        }
        out["roofline_other"] = other
        out["kernel_share"] = kernel_share
        out["forward_only"] = fwd_extra
        if not args.no_cpu_baseline and world == 1:      # CPU oracle timed on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, model)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks here, before this process touches the GPU
        sys.exit(spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
