"""Image density-estimation driver on the MI355X flow path -- the caller of the headline configuration (MNIST d = 784,
Monotonic normalizer + DAG conditioner with the MNISTCNN embedding).  Argument names and defaults, model construction,
data preparation (uniform dequantisation + logit), epoch schedule, validation / bits-per-pixel report, threshold sweep,
sampling and checkpoint files are those of the reference's ImageExperiments.py (:23-36 data transforms and bpp, :130-170
model, :184-253 epoch loop, :361-384 arguments), re-hosted on one process per GPU:

    python train_image.py -dataset MNIST -normalizer Monotonic -no_hot_encoding -prior_A_kernel 2 -b_size 100
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_image.py ...      (RCCL, batch sharded)

What differs, on purpose:
  * `nn.DataParallel` (:168) is replaced by gnf_hip.dp: every rank owns b_size rows of a global batch of
    world * b_size, one all-reduce of the flat gradient per optimiser step, the DAG constraint counted once.  The
    reference divides the loss by `batch_per_optim_step * n_gpu` (:205) although its mean already runs over the gathered
    batch; here the gradient is the plain global mean.  `-nb_gpus` is accepted and ignored (the world
    size comes from the launcher).
  * MNIST is read from the IDX files torchvision leaves under `<-data_root>/MNIST/raw` (no download: there is no
    network); `-data_root synthetic` draws sparse pseudo-digits of the same shape and dtype instead.  CIFAR10 is parsed
    and refused: its embedding front is outside the hot path.
  * The matplotlib movies / degree plots of the reference (:296-333) are not produced; the numbers they show (in / out
    degrees of the thresholded adjacency) go into the log.  Sample grids are written as binary PGM files.
  * `model.pt` / `best_model.pt` carry the reference's `module.`-prefixed keys (it saves the DataParallel wrapper), so
    either side can resume the other's run; `ADAM.pt` is in torch.optim.Adam's format."""
import argparse
import gzip
import math
import os
import struct
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from models import (buildFCNormalizingFlow, DAGConditioner, CouplingConditioner, AutoregressiveConditioner,  # noqa: E402
                    AffineNormalizer, MonotonicNormalizer)
from models.NormalizingFlowFactories import buildMNISTNormalizingFlow  # noqa: E402
from gnf_hip import dp  # noqa: E402
from train_uci import shard_batches  # noqa: E402

COND = {"DAG": DAGConditioner, "Coupling": CouplingConditioner, "Autoregressive": AutoregressiveConditioner}
THRESHOLDS = (.95, .5, .1, .01, .0001)
TEMPERATURES = (.1, .25, .5, .75, 1.)


# ----------------------------------------------------------------------------- data
def read_idx(path):
    """An IDX file (optionally gzipped) as a uint8 tensor: magic 0x0000 08 <ndim>, big-endian dimensions, raw bytes."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    zero, dtype, ndim = struct.unpack(">HBB", raw[:4])
    if zero != 0 or dtype != 0x08:
        raise ValueError("%s is not an unsigned-byte IDX file" % path)
    dims = struct.unpack(">" + "I" * ndim, raw[4:4 + 4 * ndim])
    data = np.frombuffer(raw, dtype=np.uint8, offset=4 + 4 * ndim)
    if data.size != int(np.prod(dims)):
        raise ValueError("%s: %d bytes of payload for shape %s" % (path, data.size, dims))
    return torch.from_numpy(data.reshape(dims).copy())


def _find(root, stem):
    for d in (os.path.join(root, "MNIST", "raw"), os.path.join(root, "raw"), root):
        for ext in ("", ".gz"):
            p = os.path.join(d, stem + ext)
            if os.path.isfile(p):
                return p
    raise FileNotFoundError("%s(.gz) not found under %s (torchvision layout MNIST/raw/); no download here -- pass "
                            "-data_root synthetic for pseudo-digits" % (stem, root))


def synthetic_digits(n, gen):
    """uint8 [n, 28, 28] images with MNIST's statistics in the large: ~19 % of the pixels lit, in a centred blob."""
    yy, xx = torch.meshgrid(torch.arange(28.), torch.arange(28.), indexing="ij")
    centre = torch.exp(-((yy - 13.5) ** 2 + (xx - 13.5) ** 2) / (2 * 6. ** 2))
    lit = torch.rand(n, 28, 28, generator=gen) < .45 * centre
    val = (torch.rand(n, 28, 28, generator=gen) * 255).to(torch.uint8)
    return torch.where(lit, val, torch.zeros((), dtype=torch.uint8))


def load_mnist(root, dataset, gen):
    """(train, valid, test) uint8 tensors [n, 784].  "MNIST": 50 000 / 10 000 random split of the training file
    (reference :46); "MNIST<digit>": that label only, 5 000 / rest (:66)."""
    if root == "synthetic":
        pix, lab = synthetic_digits(1200, gen), torch.randint(0, 10, (1200,), generator=gen)
        tpix, tlab = synthetic_digits(200, gen), torch.randint(0, 10, (200,), generator=gen)
        n_train = 1000
    else:
        pix, lab = read_idx(_find(root, "train-images-idx3-ubyte")), read_idx(_find(root, "train-labels-idx1-ubyte"))
        tpix, tlab = read_idx(_find(root, "t10k-images-idx3-ubyte")), read_idx(_find(root, "t10k-labels-idx1-ubyte"))
        n_train = 50000
    if len(dataset) == 6:
        digit = int(dataset[5])
        pix, tpix = pix[lab == digit], tpix[tlab == digit]
        n_train = min(5000, pix.shape[0] - 1) if root != "synthetic" else pix.shape[0] * 5 // 6
    perm = torch.randperm(pix.shape[0], generator=gen)
    flat = pix.reshape(pix.shape[0], -1)[perm]
    return flat[:n_train], flat[n_train:], tpix.reshape(tpix.shape[0], -1)


def dequantise(u8, alpha, gen):
    """reference lib/transform.py:5-21 (AddUniformNoise): x = logit(alpha + (1 - 2 alpha) (pixel + U[0,1)) / 256); the
    noise is drawn anew every time a batch is read."""
    y = (u8.to(torch.float32) + torch.rand(u8.shape, device=u8.device, generator=gen)) / 256.
    y = alpha + (1. - 2. * alpha) * y
    return torch.log(y) - torch.log(1. - y)                 # the reference's expression, term by term (fp32)


def logit_back(x, alpha):
    """lib/transform.py:9-11"""
    return (torch.sigmoid(x) - alpha) / (1. - 2. * alpha)


def compute_bpp(ll, x, alpha=1e-6):
    """bits per pixel of the ORIGINAL image from the log-likelihood of its logit-transformed dequantisation
    (ImageExperiments.py:33-37): -ll / (d ln 2) - log2(1 - 2 alpha) + 8 + mean_d [log2 s(x) + log2 (1 - s(x))]"""
    d = x.shape[1]
    s = torch.sigmoid(x)
    return (-ll / (d * math.log(2.)) - math.log2(1. - 2. * alpha) + 8.
            + (torch.log2(s) + torch.log2(1. - s)).sum(1) / d)


# ----------------------------------------------------------------------------- model, checkpoints
def build(args):
    if args.normalizer == "Affine":
        norm_t, nargs = AffineNormalizer, {}
    else:
        norm_t, nargs = MonotonicNormalizer, {"integrand_net": args.int_net, "nb_steps": 15, "solver": args.solver}
    cond_t = COND[args.conditioner]
    if cond_t is DAGConditioner:
        model = buildMNISTNormalizingFlow(args.nb_flow, norm_t, nargs, args.l1, nb_epoch_update=args.nb_steps_dual,
                                          hot_encoding=not args.no_hot_encoding, prior_kernel=args.prior_A_kernel)
        if model is None:
            raise SystemExit("-nb_flow takes 1 or 3 values with the DAG conditioner (one 28x28 scale or 28/14/7)")
    else:
        cargs = {"in_size": 784, "hidden": args.emb_net[:-1], "out_size": args.emb_net[-1]}
        if norm_t is MonotonicNormalizer:
            nargs["cond_size"] = args.emb_net[-1]
        model = buildFCNormalizingFlow(args.nb_flow[0], cond_t, cargs, norm_t, nargs)
    return model, cond_t, norm_t


def wrapped_keys(sd):
    """state_dict as the reference writes it: through nn.DataParallel, i.e. with a `module.` prefix (:247,337)"""
    return {"module." + k: v for k, v in sd.items()}


def plain_keys(sd):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def write_pgm(path, x01, nrow=4):
    """[n, 784] images in [0, 1] as one binary PGM grid (the reference saves a torchvision grid PNG, :329-330)"""
    n = x01.shape[0]
    rows = (n + nrow - 1) // nrow
    grid = torch.zeros(rows * 28, nrow * 28)
    for k in range(n):
        r, c = divmod(k, nrow)
        grid[28 * r:28 * r + 28, 28 * c:28 * c + 28] = x01[k].view(28, 28).cpu()
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (grid.shape[1], grid.shape[0]))
        f.write((grid.clamp(0, 1) * 255).round().to(torch.uint8).numpy().tobytes())


# ----------------------------------------------------------------------------- evaluation
@torch.no_grad()
def evaluate(model, u8, b, alpha, gen):
    """mean log-likelihood and bits per pixel over the full batches of a split (drop_last as the reference's loaders)"""
    ll_sum = bpp_sum = 0.
    n = 0
    for i in range(0, u8.shape[0] - b + 1, b):
        x = dequantise(u8[i:i + b], alpha, gen)
        z, jac = model(x)
        ll = model.z_log_density(z) + jac
        ll_sum += ll.mean().item()
        bpp_sum += compute_bpp(ll, x, alpha).mean().item()
        n += 1
    return ll_sum / max(n, 1), bpp_sum / max(n, 1)


def train(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dataset == "CIFAR10":
        raise SystemExit("CIFAR10: the CIFAR embedding front is outside the MI355X hot path (DESIGN.md section 8)")
    if not torch.cuda.is_available():
        raise SystemExit("train_image.py needs an MI355X (the flow kernels have no CPU fallback)")
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
    alpha = 1e-6
    host_gen = torch.Generator().manual_seed(1234)           # identical on every rank: same split, same global batches
    trn, val, tst = [t.to(dev) for t in load_mnist(args.data_root, args.dataset, host_gen)]
    noise = torch.Generator(device=dev).manual_seed(4321 + rank)
    torch.manual_seed(0)
    model, cond_t, norm_t = build(args)
    os.makedirs(args.folder, exist_ok=True)
    tag = "_" + args.f_number if args.f_number is not None else ""
    if args.load:
        model.load_state_dict(plain_keys(torch.load(os.path.join(args.folder, "model%s.pt" % tag), map_location="cpu")))
    model.to(dev)
    dp.seed_gates(model, rank)
    state = dp.FlatState(model)
    state.broadcast(0)
    adam_file = os.path.join(args.folder, "ADAM%s.pt" % tag)
    if args.load and os.path.isfile(adam_file):
        state.load_optimizer_state_dict(model, torch.load(adam_file, map_location=dev))
    if args.load:                                            # reference :186-188
        for c in model.getConditioners():
            if hasattr(c, "getAlpha"):
                c._set("alpha", c.getAlpha())     # in place: the buffer keeps its device address
    log = open(os.path.join(args.folder, "logs"), "a") if rank == 0 else None

    def say(msg):
        if rank == 0:
            print(msg, flush=True)
            log.write(msg + "\n")
            log.flush()

    say(str(vars(args)))
    say("Number of parameters: %d" % sum(p.numel() for p in model.parameters()))
    b, k_acc = args.b_size, max(args.batch_per_optim_step, 1)
    best = math.inf
    for epoch in range(args.nb_epoch):
        t0 = time.perf_counter()
        ll_tot = torch.zeros((), device=dev)
        if not args.test:
            n = 0
            # full global batches only (drop_last); every rank cuts the same ones and takes its b rows of each
            shards = [r for r in shard_batches(trn.shape[0], b * world, rank, world, host_gen) if r.numel() == b]
            if args.max_batches:
                shards = shards[:args.max_batches]
            if k_acc > 1:                                    # micro-batch gradients left over from an epoch whose batch
                state.drop_grads()                           # count is no multiple of k are dropped, as the reference's
                                                             # zero_grad() at batch_idx % k == 0 does (:210-211)
            for n, rows in enumerate(shards, 1):
                if norm_t is MonotonicNormalizer:            # node-count jitter (:201-203)
                    k = args.nb_steps + int(torch.randint(0, 10, [1], generator=host_gen))
                    for nrm in model.getNormalizers():
                        nrm.nb_steps = k
                x = dequantise(trn[rows.to(dev)], alpha, noise)
                if k_acc == 1:
                    loss = dp.train_step(model, state, x, lr=args.learning_rate, weight_decay=args.weight_decay)
                else:
                    loss = dp.accumulate(model, x, 1. / k_acc)
                    if n % k_acc == 0:
                        dp.apply_step(state, args.learning_rate, args.weight_decay)
                ll_tot += loss.detach()
            ll_tot /= max(n, 1)
            if world > 1:                                    # every replica must take the same branch in model.step()
                dp.all_reduce_sum(ll_tot)
                ll_tot /= world
            if not torch.isfinite(ll_tot):
                say("Error Nan in loss")
                with torch.no_grad():
                    say("Dagness: %s" % [float(v) for v in model.DAGness()])
                raise SystemExit(1)
            with torch.no_grad():
                say("Dagness: %s" % [float(v) for v in model.DAGness()])
            model.step(epoch, ll_tot)
            if not dp.replicas_identical(state, model):
                raise SystemExit("data-parallel replicas diverged in epoch %d" % epoch)
        # ---- validation (:222-236): 150 quadrature steps
        for nrm in model.getNormalizers():
            if type(nrm) is MonotonicNormalizer:
                nrm.nb_steps = 150
        ll_val, bpp_val = evaluate(model, val, b, alpha, noise)
        with torch.no_grad():
            dagness = float(max(model.DAGness()))
        say("epoch: %d - Train loss: %4f - Valid log-likelihood: %4f - Valid BPP %4f - <<DAGness>>: %4f - Elapsed time "
            "per epoch %4f (seconds)" % (epoch, float(ll_tot), ll_val, bpp_val, dagness, time.perf_counter() - t0))
        if model.isInvertible() and -ll_val < best:
            best = -ll_val
            say("------- New best validation loss --------")
            if rank == 0:
                torch.save(wrapped_keys(model.state_dict()), os.path.join(args.folder, "best_model.pt"))
            ll_tst, bpp_tst = evaluate(model, tst, b, alpha, noise)
            say("epoch: %d - Test log-likelihood: %4f - Test BPP %4f - <<DAGness>>: %4f" % (epoch, ll_tst, bpp_tst, dagness))
        if epoch % 10 == 0 and cond_t is DAGConditioner:
            threshold_sweep(model, val, b, alpha, noise, epoch, say)
        if model.isInvertible() and rank == 0:
            sample_grids(model, dev, alpha, epoch, args.folder, say)
        if rank == 0:
            if epoch % args.nb_steps_dual == 0:
                say("Saving model N°%d" % epoch)
                torch.save(wrapped_keys(model.state_dict()), os.path.join(args.folder, "model_%d.pt" % epoch))
                torch.save(state.optimizer_state_dict(model, args.learning_rate, args.weight_decay),
                           os.path.join(args.folder, "ADAM_%d.pt" % epoch))
            torch.save(wrapped_keys(model.state_dict()), os.path.join(args.folder, "model.pt"))
            torch.save(state.optimizer_state_dict(model, args.learning_rate, args.weight_decay),
                       os.path.join(args.folder, "ADAM.pt"))
    if world > 1:
        dist.destroy_process_group()
    return model


@torch.no_grad()
def threshold_sweep(model, val, b, alpha, noise, epoch, say):
    """every 10th epoch (:258-286): validation with the deterministic hard-thresholded adjacency at five thresholds, the
    gate settings restored afterwards; the in / out degrees the reference plots (:318-333) as numbers"""
    conds = model.getConditioners()
    saved = [(c.stoch_gate, c.noise_gate, c.s_thresh) for c in conds]
    for c in conds:
        c.stoch_gate, c.noise_gate, c.s_thresh = False, False, True
    for th in THRESHOLDS:
        for c in conds:
            c.h_thresh = th
        ll, bpp = evaluate(model, val, b, alpha, noise)
        say("epoch: %d - Threshold: %4f - Valid log-likelihood: %4f - Valid BPP %4f - <<DAGness>>: %4f"
            % (epoch, th, ll, bpp, float(max(model.DAGness()))))
    for c, (sg, ng, st) in zip(conds, saved):
        c.h_thresh, c.stoch_gate, c.noise_gate, c.s_thresh = 0., sg, ng, st
    A = conds[0].soft_thresholded_A() > 0.
    deg_out, deg_in = A.sum(0).float(), A.sum(1).float()
    say("epoch: %d - in-degree mean %.2f max %d - out-degree mean %.2f max %d"
        % (epoch, deg_in.mean().item(), int(deg_in.max()), deg_out.mean().item(), int(deg_out.max())))


@torch.no_grad()
def sample_grids(model, dev, alpha, epoch, folder, say, n_images=16):
    """once the graph is a DAG (:322-330): 16 samples at five temperatures, the round-trip error, a 4x4 grid each"""
    g = torch.Generator(device=dev).manual_seed(epoch)
    for T in TEMPERATURES:
        z = torch.randn(n_images, 784, device=dev, generator=g) * T
        x = model.invert(z)
        say("epoch: %d - T %.2f - |z - f(f^-1(z))| %.3e" % (epoch, T, (z - model(x)[0]).abs().mean().item()))
        write_pgm(os.path.join(folder, "images_%d_%f.pgm" % (epoch, T)), logit_back(x, alpha))


# (flag, default, type / None for a switch, nargs, choices): names and defaults of the reference's driver
_ARGS = (
    ("load", False, None, None, None), ("folder", "", str, None, None), ("nb_steps_dual", 100, int, None, None),
    ("l1", 10., float, None, None), ("nb_epoch", 10000, int, None, None), ("b_size", 1, int, None, None),
    ("int_net", [50, 50, 50], int, "+", None), ("nb_steps", 20, int, None, None), ("f_number", None, str, None, None),
    ("solver", "CC", str, None, ["CC", "CCParallel"]), ("nb_flow", [1], int, "+", None), ("test", False, None, None, None),
    ("weight_decay", 1e-5, float, None, None), ("learning_rate", 1e-3, float, None, None),
    ("batch_per_optim_step", 1, int, None, None), ("nb_gpus", 1, int, None, None),
    ("dataset", "MNIST", str, None, ["MNIST", "CIFAR10"] + ["MNIST%d" % k for k in range(10)]),
    ("normalizer", "Affine", str, None, ["Affine", "Monotonic"]), ("no_hot_encoding", False, None, None, None),
    ("prior_A_kernel", None, int, None, None), ("conditioner", "DAG", str, None, sorted(COND)),
    ("emb_net", [100, 100, 100, 10], int, "+", None),
    # additions of this driver
    ("max_batches", 0, int, None, None),      # stop an epoch after this many batches (smoke runs; 0: all)
    ("data_root", ".", str, None, None),      # directory holding MNIST/raw/*-ubyte(.gz), or `synthetic`
)


def parse(argv=None):
    ap = argparse.ArgumentParser(description="image density estimation with graphical normalizing flows on MI355X "
                                             "(b_size = rows per GPU; nb_gpus is ignored, the launcher sets the world size)")
    for name, default, typ, nargs, choices in _ARGS:
        if typ is None:
            ap.add_argument("-" + name, default=default, action="store_true")
        else:
            kw = {"nargs": nargs} if nargs else {}
            if choices:
                kw["choices"] = choices
            ap.add_argument("-" + name, default=default, type=typ, **kw)
    args = ap.parse_args(argv)
    if not args.folder:
        args.folder = os.path.join(args.dataset, time.strftime("%m_%d_%Y_%H_%M_%S"))
    return args


if __name__ == "__main__":
    train(parse())
