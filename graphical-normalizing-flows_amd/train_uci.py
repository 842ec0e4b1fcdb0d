"""Tabular density-estimation driver on the MI355X flow path: the run configurations, model construction, epoch
schedule and checkpoint files of the reference's UCIExperiments.py (:58-220, argument names :226-256, yml schema of
UCIExperimentsConfigurations.yml), re-hosted on one process per GPU (`torchrun --nproc-per-node N train_uci.py ...`,
RCCL) with the fused step of gnf_hip.dp.  Datasets are not bundled: pass `-data file.npz` (arrays trn / val / tst,
standardised as in the reference's loaders) or `-data synthetic` for N(0,1) data of the dataset's dimension.

Checkpoints: `model.pt` is `state_dict()` with the reference's keys, `ADAM.pt` is in torch.optim.Adam's format, so
either side can resume the other's run."""
import argparse
import math
import os
import re
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from models import (buildFCNormalizingFlow, DAGConditioner, CouplingConditioner, AutoregressiveConditioner,  # noqa: E402
                    AffineNormalizer, MonotonicNormalizer)
from gnf_hip import dp  # noqa: E402

COND = {"DAG": DAGConditioner, "Coupling": CouplingConditioner, "Autoregressive": AutoregressiveConditioner}
NORM = {"affine": AffineNormalizer, "monotonic": MonotonicNormalizer}
DIMS = {"power": 6, "gas": 8, "hepmass": 21, "miniboone": 43, "bsds300": 63, "digits": 64, "proteins": 11}


def load_split(spec, dataset, seed=0):
    if spec == "synthetic":
        g = torch.Generator().manual_seed(seed)
        d = DIMS[dataset]
        return [torch.randn(n, d, generator=g) for n in (20000, 2000, 2000)]
    z = np.load(spec)
    return [torch.from_numpy(np.asarray(z[k], dtype=np.float32)) for k in ("trn", "val", "tst")]


def batches(X, b, shuffle, gen):
    idx = torch.randperm(X.shape[0], generator=gen) if shuffle else torch.arange(X.shape[0])
    for i in range(0, X.shape[0], b):
        yield X[idx[i:i + b].to(X.device)]


def shard_batches(n, b, rank, world, gen):
    """Index tensors of this rank's share of every global minibatch of one epoch.  The permutation is drawn from `gen`,
    which every rank seeds identically, so all ranks cut the SAME global batches; each batch is trimmed to a multiple
    of `world` rows and dealt round-robin, so every rank runs the same number of steps with the same shard size (no rank
    is left waiting in the per-step all-reduce, and the mean of the rank means is the global batch mean)."""
    idx = torch.randperm(n, generator=gen)
    out = []
    for i in range(0, n, b):
        cur = idx[i:i + b]
        keep = cur.numel() // world * world
        if keep:
            out.append(cur[:keep][rank::world])
    return out


def build(args, dim):
    cond_t, norm_t = COND[args.conditioner], NORM[args.normalizer]
    cargs = {"in_size": dim, "hidden": args.emb_net[:-1], "out_size": args.emb_net[-1]}
    if cond_t is DAGConditioner:          # reference :83-88 (gumble_T is pinned to .5 there)
        cargs.update(l1=args.l1, gumble_T=.5, nb_epoch_update=args.nb_steps_dual, hot_encoding=True)
    nargs = {}
    if norm_t is MonotonicNormalizer:
        nargs = {"integrand_net": args.int_net, "cond_size": args.emb_net[-1], "nb_steps": args.nb_steps,
                 "solver": args.solver}
    return buildFCNormalizingFlow(args.nb_flow, cond_t, cargs, norm_t, nargs), cond_t, norm_t


@torch.no_grad()
def mean_ll(model, X, b, gen):
    tot, n = 0., 0
    for cur in batches(X, b, False, gen):
        z, jac = model(cur)
        tot += (model.z_log_density(z) + jac).mean().item()
        n += 1
    return tot / max(n, 1)


def train(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train_uci.py needs an MI355X (the flow kernels have no CPU fallback)")
    backend = os.environ.get("GNF_DIST_BACKEND", "nccl")     # "nccl" is RCCL; "gloo" only to exercise N>1 on a 1-GPU box
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local >= ndev:
        raise SystemExit("rank %d has no GPU (%d visible)" % (local, ndev))
    local %= max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    trn, val, tst = [t.to(dev) for t in load_split(args.data, args.dataset)]
    torch.manual_seed(0)
    model, cond_t, norm_t = build(args, trn.shape[1])
    os.makedirs(args.folder, exist_ok=True)
    tag = "_" + args.f_number if args.f_number is not None else ""
    if args.load:
        model.load_state_dict(torch.load(os.path.join(args.folder, "model%s.pt" % tag), map_location="cpu"))
    model.to(dev)
    dp.seed_gates(model, rank)                               # independent gate noise per rank and per flow step
    state = dp.FlatState(model)
    state.broadcast(0)
    adam_file = os.path.join(args.folder, "ADAM%s.pt" % tag)
    if args.load and os.path.isfile(adam_file):
        state.load_optimizer_state_dict(model, torch.load(adam_file, map_location=dev))
    gen = torch.Generator().manual_seed(1234)
    best = math.inf
    log = open(os.path.join(args.folder, "logs"), "a") if rank == 0 else None

    def say(msg):
        if rank == 0:
            print(msg, flush=True)
            log.write(msg + "\n")
            log.flush()

    say(str(vars(args)))
    for epoch in range(args.nb_epoch):
        t0 = time.perf_counter()
        if cond_t is DAGConditioner:
            with torch.no_grad():
                for c in model.getConditioners():
                    c.constrainA(zero_threshold=0.)
        ll_tot, n = torch.zeros((), device=dev), 0
        if not args.test:
            for rows in shard_batches(trn.shape[0], max(args.b_size, world), rank, world, gen):
                if norm_t is MonotonicNormalizer:            # node-count jitter of the reference (:131-133)
                    k = args.nb_steps + int(torch.randint(0, 10, [1], generator=gen))
                    for nrm in model.getNormalizers():
                        nrm.nb_steps = k
                loss = dp.train_step(model, state, trn[rows.to(dev)], lr=args.learning_rate,
                                     weight_decay=args.weight_decay)
                ll_tot += loss.detach()
                n += 1
            ll_tot /= max(n, 1)
            if world > 1:
                # model.step() decides the dual update / post-processing from this value (DAGConditioner.step): every
                # replica must take the same branch, so they all see the mean over ranks
                dp.all_reduce_sum(ll_tot)
                ll_tot /= world
            if not torch.isfinite(ll_tot):
                if rank == 0:
                    torch.save(model.state_dict(), os.path.join(args.folder, "NANmodel.pt"))
                raise SystemExit("non-finite loss")
            model.step(epoch, ll_tot)
            if not dp.replicas_identical(state, model):
                raise SystemExit("data-parallel replicas diverged in epoch %d" % epoch)
        if norm_t is MonotonicNormalizer:
            for nrm in model.getNormalizers():
                nrm.nb_steps = args.nb_steps + 20
        ll_val = mean_ll(model, val, args.b_size, gen)
        with torch.no_grad():
            dagness = float(max(model.DAGness()))
        say("epoch: %d - Train loss: %4f - Valid log-likelihood: %4f - <<DAGness>>: %4f - Elapsed time per epoch %4f "
            "(seconds)" % (epoch, float(ll_tot), ll_val, dagness, time.perf_counter() - t0))
        if rank == 0:
            if dagness < 1e-20 and -ll_val < best:
                best = -ll_val
                torch.save(model.state_dict(), os.path.join(args.folder, "best_model.pt"))
                say("epoch: %d - Test log-likelihood: %4f - <<DAGness>>: %4f"
                    % (epoch, mean_ll(model, tst, args.b_size, gen), dagness))
            torch.save(model.state_dict(), os.path.join(args.folder, "model.pt"))
            torch.save(state.optimizer_state_dict(model, args.learning_rate, args.weight_decay),
                       os.path.join(args.folder, "ADAM.pt"))
    if world > 1:
        dist.destroy_process_group()
    return model


_SCI = re.compile(r"[-+]?(\d[\d_]*\.?[\d_]*|\.[\d_]+)([eE][-+]?\d+)?")


def _yml_number(v):
    """PyYAML reads `1e-3` and `.5e1` as strings (YAML 1.1 wants a dot AND a signed exponent); the reference installs a
    YAML 1.2 float resolver so that its configuration file's `weight_decay: 1e-3` arrives as a float
    (UCIExperiments.py:258-269, UCIExperimentsConfigurations.yml:84).  Same outcome: plain-number strings -> float."""
    if isinstance(v, str) and _SCI.fullmatch(v.strip()):
        return float(v.replace("_", ""))
    if isinstance(v, list):
        return [_yml_number(e) for e in v]
    return v


def parse(argv=None):
    ap = argparse.ArgumentParser(description="UCI density estimation with graphical normalizing flows on MI355X")
    ap.add_argument("-load_config", default=None, type=str)
    ap.add_argument("-config_file", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "uci_configs.yml"))
    ap.add_argument("-dataset", default=None, choices=sorted(DIMS))
    ap.add_argument("-data", default="synthetic")
    ap.add_argument("-load", default=False, action="store_true")
    ap.add_argument("-folder", default="")
    ap.add_argument("-f_number", default=None, type=str)
    ap.add_argument("-test", default=False, action="store_true")
    ap.add_argument("-nb_flow", type=int, default=1)
    ap.add_argument("-weight_decay", default=1e-5, type=float)
    ap.add_argument("-learning_rate", default=1e-3, type=float)
    ap.add_argument("-nb_epoch", default=10000, type=int)
    ap.add_argument("-b_size", default=100, type=int)
    ap.add_argument("-conditioner", default="DAG", choices=sorted(COND))
    ap.add_argument("-emb_net", default=[100, 100, 100, 10], nargs="+", type=int)
    ap.add_argument("-nb_steps_dual", default=100, type=int)
    ap.add_argument("-l1", default=.2, type=float)
    ap.add_argument("-gumble_T", default=1., type=float)
    ap.add_argument("-normalizer", default="affine", choices=sorted(NORM))
    ap.add_argument("-int_net", default=[100, 100, 100, 100], nargs="+", type=int)
    ap.add_argument("-nb_steps", default=20, type=int)
    ap.add_argument("-solver", default="CC", type=str, choices=["CC", "CCParallel"])
    args = ap.parse_args(argv)
    if args.load_config is not None:                         # a named entry of the yml overrides the command line
        with open(args.config_file) as f:
            cfg = yaml.safe_load(f)[args.load_config]
        for k, v in cfg.items():
            setattr(args, k, _yml_number(v))
    if args.dataset is None:
        ap.error("-dataset (or a -load_config entry naming one) is required")
    if not args.folder:
        args.folder = os.path.join("runs", args.load_config or args.dataset)
    return args


if __name__ == "__main__":
    train(parse())
