"""Headline benchmark: samples/sec of one full training step (fwd + log|det J| + NLL + bwd
+ gradient all-reduce + Adam) of the MNIST d=784 Monotonic+DAG flow (BASELINE.json
configs[3]) on synthetic logit-space pseudo-MNIST, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around the
dominant hand-written kernel on the stream it is launched on; `cpu_baseline` times the
CPU oracle (oracle/gnf_oracle.py, a PyTorch-CPU restatement of the reference path, parity
checked against reference-generated golden vectors) on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "graphical-normalizing-flows_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

D = 784
B_PER_GPU = 100
INT_NET = [50, 50, 50]
COND = 30
S_NODES = 20
PEAK_F32_TFLOPS = 157.3          # MI355X fp32 MFMA/vector peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.         # dense bf16 MFMA peak (same guide); the split-bf16 Monotonic forward is priced against it
NOMINAL_GHZ = 2.4                # the clock behind that peak: 256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz
PMC_INPUTS = "r06_bench_inputs.json"
DOMINANT_OP = "gnf_mnistcnn_conv_bwd"       # the entry point of the dominant kernel (cnn_bwd_wino_k): timed live in the region
# round 5: with x frozen the step goes through the plan variants of three entry points (include/gnf_hip.h, "structural zeros of
# the gate backward"); their times are reported under the names of the calls they replace
PLAN_VARIANT = {"gnf_mnistcnn_conv_bwd_cols": "gnf_mnistcnn_conv_bwd", "gnf_dag_gate_fwd_plan": "gnf_dag_gate_fwd",
                "gnf_dag_gate_bwd_cols": "gnf_dag_gate_bwd"}


def _collect(abi):
    """abi.profile_collect() with the plan variants filed under the entry points they stand in for"""
    prof = abi.profile_collect()
    for alias, name in PLAN_VARIANT.items():
        if alias in prof and name not in prof:               # a run in which BOTH forms executed (a multi-step flow whose later
            prof[name] = prof.pop(alias)                     # steps hand x a gradient) keeps both keys: nothing is overwritten
    return prof
OPS_STEPS = 5                               # untimed steps behind the region in which every entry point is timed (at least; = --steps)


def pseudo_mnist(gen, B, d):
    from gnf_hip import configs
    return configs.pseudo_mnist(gen, B, d)


def build_flow():
    """cfg4 of BASELINE.json: buildMNISTNormalizingFlow([1], MonotonicNormalizer [50,50,50], prior kernel 2,
    hot_encoding=False) -- gnf_hip.configs holds the builders of all five configurations"""
    from gnf_hip import configs
    return configs.build_cfg4_flow()


def train_step(flow, state, x):
    """one optimisation step on the local shard (gnf_hip.dp.train_step) at the headline node
    count S=20 (the drivers jitter nb_steps per batch, ImageExperiments.py:201-203)."""
    from gnf_hip import dp
    for nrm in flow.getNormalizers():
        nrm.nb_steps = S_NODES
    return dp.train_step(flow, state, x, lr=1e-3, weight_decay=1e-5)


CPU_B = 8                                   # rows of the CPU sample ("B = 8-16 ... acceptable if stated")


def cpu_baseline():
    """CPU oracle, same model/inputs, a B = 8 sample (B=100 needs >10 GB of conv activations and minutes per step on the
    host; round 4 took B = 2, where 6 272 single-channel images give torch's conv too little work per thread): fwd +
    log|det J| + NLL + bwd, thread counts swept up to the physical cores."""
    from oracle import gnf_oracle as O
    torch.manual_seed(0)
    flow = build_flow()
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    pre = "steps.0.conditioner."
    cnn = {k[len(pre + "embedding_net."):]: v.requires_grad_(True) for k, v in sd.items() if "embedding_net." in k}
    A = sd[pre + "A"].requires_grad_(True)
    layers, k = [], 0
    ipre = "steps.0.normalizer.integrand_net.net."
    while ipre + "%d.weight" % k in sd:
        layers.append((sd[ipre + "%d.weight" % k].requires_grad_(True), sd[ipre + "%d.bias" % k].requires_grad_(True)))
        k += 2
    Bc = CPU_B
    x = pseudo_mnist(torch.Generator().manual_seed(1234), Bc, D)
    ncpu = os.cpu_count() or 1
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass

    def step():
        u1, u2 = torch.rand(Bc, D, D), torch.rand(Bc, D, D)
        e = O.dag_masked_inputs(x, A, True, 0., True, False, 1., u1, u2, None, False)
        h = O.mnistcnn_forward(e, cnn).view(Bc, D, -1)
        z, jac = O.monotonic_forward(x, h, layers, S_NODES)
        closs = O.dag_loss(A, sd[pre + "alpha"], D % 50, sd[pre + "lambd"], sd[pre + "c"], sd[pre + "dag_const"],
                           sd[pre + "l1_weight"])
        O.flow_loss(z, torch.log(jac).sum(1), closs).backward()
    # the named baseline is the ALL-PHYSICAL-CORES figure (os.cpu_count() counts SMT siblings), with a
    # 1-thread figure; the short sweep over smaller thread counts is reported beside it (`best_of_sweep`: on the 128-core
    # hosts of this pool 16-32 threads are ~3x faster than 128 on this B = 8 sample -- printed, not substituted)
    phys = max(1, ncpu // 2) if ncpu >= 16 else ncpu
    sweep = {}
    for th in sorted({min(16, phys), min(32, phys), min(64, phys)} - {phys}):
        torch.set_num_threads(th)
        step()                               # warm-up at this setting
        t0 = time.perf_counter()
        step(); step()
        sweep[th] = Bc / ((time.perf_counter() - t0) / 2)
    torch.set_num_threads(phys)
    step()                                   # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 2 or (time.perf_counter() - t0 < 10. and n < 40):     # a bounded sample: ~10 s of host work
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    sweep[phys] = Bc / dt
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(1)                 # plus a 1-thread figure
    t0 = time.perf_counter()
    step()
    dt1 = time.perf_counter() - t0
    torch.set_num_threads(phys)
    return {"value": Bc / dt, "unit": "samples/s", "cores": phys, "kind": "port",
            "os_cpu_count": ncpu, "physical_cores_assumed": phys, "cpu_model": model, "threads": phys,
            "threads_sweep_samples_per_s": {str(k): round(v, 2) for k, v in sorted(sweep.items())},
            "best_of_sweep": {"threads": best, "value": round(sweep[best], 2)},
            "sample": "%d steps of B=%d (same d=784 model, S=20), fwd+logdet+NLL+bwd, torch %d threads = all physical cores"
                      % (n, Bc, phys),
            "value_1thread": Bc / dt1}


def measured_peaks(dev):
    """Measured device ceilings: sustained v_mfma_f32_16x16x4_f32 rate of a pure-MFMA kernel over
    ~0.2 s and STREAM-copy bandwidth of a 1 GiB buffer, both timed with HIP events on the launch stream."""
    from gnf_hip import abi
    lib = abi.load()
    st = abi.stream()
    out = torch.zeros(4, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    lib.gnf_probe_mfma_f32(abi.ptr(out), 2000, 2048, st)            # warm up clocks
    ev[0].record()
    flops = 0
    for _ in range(4):
        flops += lib.gnf_probe_mfma_f32(abi.ptr(out), 20000, 2048, st)
    ev[1].record()
    n = 1 << 28
    a = torch.empty(n, device=dev)
    b = torch.ones(n, device=dev)
    lib.gnf_probe_copy(abi.ptr(a), abi.ptr(b), n, st)
    ev[2].record()
    for _ in range(10):
        lib.gnf_probe_copy(abi.ptr(a), abi.ptr(b), n, st)
    ev[3].record()
    torch.cuda.synchronize()
    return {"mfma_f32_TFLOPs": round(flops / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12, 1),
            "stream_copy_GBps": round(10 * 2 * 4 * n / (ev[2].elapsed_time(ev[3]) * 1e-3) / 1e9, 1),
            "note": "pure-MFMA kernel (8 independent 16x16x4 f32 chains/wave, 2 waves/SIMD) and a 1 GiB float4 copy (contiguous chunk per workgroup, 8 non-temporal 16-B loads in flight per lane: best of tools/copy_probe.hip)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: the GLOBAL batch G is fixed and every rank takes G / N rows (default: weak "
                         "scaling, BASELINE's 100 rows per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (tests of the N>1 path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: this process becomes the launcher and never touches the GPU -- it starts one rank
        # per GPU as CHILD processes (torch.distributed.run), waits, and exits with their code.
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or run `python bench.py --gpus N`"
                         " and let it spawn them)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.global_batch and (args.global_batch <= 0 or args.global_batch % world):
        raise SystemExit("bench.py: --global-batch %d is not a positive multiple of the %d ranks" % (args.global_batch, world))
    b_rank = args.global_batch // world if args.global_batch else B_PER_GPU      # rows of this rank per step
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback of the product path)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("GNF_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; "gloo" only to exercise the
    if backend == "nccl" and local >= ndev:                  # N>1 path on a 1-GPU box (ranks share the device)
        raise SystemExit("rank %d has no GPU (%d visible)" % (local, ndev))
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force = os.environ.get("GNF_FORCE_DIST") == "1"          # world size 1 with the collective path on (dp.collective_on)
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    collective = world > 1 or force

    from gnf_hip import abi, ops
    abi.load()
    flow = build_flow().to(dev)
    from gnf_hip import dp
    dp.seed_gates(flow, rank)                                # per-rank, per-conditioner Philox keys (like DP replicas)
    state = dp.FlatState(flow)
    state.broadcast(0)
    _flush_c_stdio()                                         # RCCL has announced itself by now (first collective): every rank
    x = pseudo_mnist(torch.Generator().manual_seed(1234 + rank), b_rank, D).to(dev)

    def fence():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(flow, state, x)
    fence()
    # HIP events INSIDE the timed region around the dominant kernel's entry point only: an event record between two
    # dependent launches leaves the GPU idle for ~5.5 us (kernel trace, tools/trace_gaps.py: with all seven entry points
    # instrumented a step carried 12 such gaps = 65 us = 1 % of it).  The other entry points are timed the same way in
    # OPS_STEPS untimed steps right behind the region.
    abi.profile_enable((DOMINANT_OP, "gnf_mnistcnn_conv_bwd_cols"))
    dp.comm_profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(flow, state, x)
    fence()
    dt = time.perf_counter() - t0
    prof_live = _collect(abi)
    allreduce_ms = dp.comm_profile(False)                    # HIP events around the step's one collective
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite loss")
    ops_steps = max(OPS_STEPS, args.steps)
    for _ in range(2):                                       # lead-in: the fence above left the GPU idle
        train_step(flow, state, x)
    abi.profile_enable(("gnf_mnistcnn_conv_fwd", "gnf_mnistcnn_conv_bwd", "gnf_monotonic_fwd", "gnf_monotonic_bwd",
                        "gnf_dag_gate_fwd", "gnf_dag_gate_bwd", "gnf_gemm") + tuple(PLAN_VARIANT))
    for _ in range(ops_steps):
        train_step(flow, state, x)
    fence()
    prof = _collect(abi)
    dom_ms_untimed_pass = prof.get(DOMINANT_OP)
    prof.update(prof_live)                                   # the dominant kernel: the live figure of the timed region

    # secondary figures of SURVEY.md 8(d), outside the timed region of the headline: (i) fwd + log-det + NLL + bwd
    # without all-reduce / Adam, (ii) the full step with the training-realistic node count S ~ U{20..29}
    def timed(fn, n, warm=2):
        for i in range(warm):                # untimed: the first pass of a new loop shape grows the allocator pools (a 4 ms
            fn(i)                            # one-off that read as "fwd+bwd alone is slower than the full step" in round 5)
        fence()
        t = time.perf_counter()
        for i in range(n):
            fn(i)
        fence()
        return (time.perf_counter() - t) / n

    def fwd_bwd(_):
        state.drop_grads()                   # .grad = None AND the flat-buffer slots handed back: the backward kernels write the
        z, ld = flow(x)                      # gradients in place as in a training step (with p.grad = None alone every step
        flow.loss(z, ld).backward()          # allocated fresh gradient tensors and autograd added A's two contributions)

    def mixed(i):
        for nrm in flow.getNormalizers():
            nrm.nb_steps = 20 + (7 * i + 3) % 10
        from gnf_hip import dp as _dp
        _dp.train_step(flow, state, x, lr=1e-3, weight_decay=1e-5)

    secondary = not args.no_secondary
    t_fb = timed(fwd_bwd, 10) if secondary else None
    for p in flow.parameters():
        p.grad = None
    if secondary:
        for i in range(10):          # one untimed step per node count: quadrature rules and workspaces exist, as in training
            mixed(i)
    t_mix = timed(mixed, 10) if secondary else None

    # (ii') the evaluation path of the reference: likelihoods under no_grad at nb_steps = 150 (ImageExperiments.py:232-243:
    # z, jac = model(x); ll = z_log_density(z) + jac) -- parity-tested in tests/test_gpu_eval_path.py, timed here
    def eval_fwd(_):
        for nrm in flow.getNormalizers():
            nrm.nb_steps = 150
        with torch.no_grad():
            z, ld = flow(x)
            return flow.z_log_density(z) + ld
    if secondary:
        eval_fwd(0)
    t_eval = timed(eval_fwd, 10) if secondary else None
    for nrm in flow.getNormalizers():
        nrm.nb_steps = S_NODES

    # (iii) the same full step after the DAG phase: post_process() froze a binary A and the gate is deterministic, so
    # the embedding net runs on the sparse crop kernels.  Fresh flow: A leaves the optimiser state.
    from gnf_hip import dp as _dp
    t_det, det_error, t_evaldet, flow_det = None, None, None, None
    try:                                   # a secondary figure must not take the headline down with it
        if not secondary:
            raise RuntimeError("skipped (--no-secondary)")
        flow_det = build_flow().to(dev)
        with torch.no_grad():
            for c in flow_det.getConditioners():
                c.post_process(zero_threshold=.1)
        for nrm in flow_det.getNormalizers():
            nrm.nb_steps = 20
        state_det = _dp.FlatState(flow_det)
        state_det.broadcast(0)

        def frozen(_):
            _dp.train_step(flow_det, state_det, x, lr=1e-3, weight_decay=1e-5)
        timed(frozen, 3)
        t_det = timed(frozen, 10)
        if not all(c._sparse_checked[1] for c in flow_det.getConditioners()):
            raise RuntimeError("frozen-gate step did not run on the sparse embedding kernels")

    except Exception as exc:               # noqa: BLE001
        t_det, det_error, flow_det = None, repr(exc), None
    evaldet_error = None
    try:                                   # its own guard: a failing evaluation path must not discard the frozen-gate figure
        if flow_det is None:
            raise RuntimeError("no frozen-gate flow")

        # (iii') the evaluation path on that flow: no_grad, nb_steps = 150, sparse front in its evaluation form
        def eval_det(_):
            for nrm in flow_det.getNormalizers():
                nrm.nb_steps = 150
            with torch.no_grad():
                z, ld = flow_det(x)
                return flow_det.z_log_density(z) + ld
        eval_det(0)
        t_evaldet = timed(eval_det, 10)
    except Exception as exc:               # noqa: BLE001
        t_evaldet, evaldet_error = None, repr(exc)
    # max over ranks of every timing (a missing secondary figure travels as -1); replicas must have stayed identical:
    # compare an order-independent bit checksum of the flat parameter buffer across ranks
    tmax = torch.tensor([dt, t_fb or -1., t_mix or -1., t_det or -1., allreduce_ms or -1., t_eval or -1., t_evaldet or -1.],
                        dtype=torch.float64)
    replicas_identical = dp.replicas_identical(state, flow)
    per_rank_ms = [dt / args.steps * 1e3]
    if collective:
        cdev = dev if backend == "nccl" else torch.device("cpu")     # RCCL moves device buffers, gloo host buffers
        mine = torch.tensor([dt / args.steps * 1e3], dtype=torch.float64, device=cdev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)                      # per-rank step time: the spread the max hides
        per_rank_ms = [float(g.item()) for g in gathered]
        tmax = tmax.to(cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tmax = tmax.cpu()
    dt, t_fb, t_mix, t_det, allreduce_ms, t_eval, t_evaldet = [v if v > 0 else None for v in tmax.tolist()]
    if not replicas_identical:
        raise SystemExit("data-parallel replicas diverged (parameter checksums differ across ranks)")
    # which physical device every rank sat on (review of round 5, item 6): uuid + PCI address of the rank's device,
    # all-gathered, so that whoever reads the line can verify an N-rank run used N devices
    rank_devices = [_device_id(local)]
    if collective:
        raw = rank_devices[0].encode()[:96].ljust(96, b"\0")
        mine = torch.tensor(list(raw), dtype=torch.uint8, device=dev if backend == "nccl" else "cpu")
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        rank_devices = [bytes(g.cpu().tolist()).rstrip(b"\0").decode() for g in gathered]
        if backend == "nccl" and len(set(rank_devices)) != world:
            raise SystemExit("bench.py: %d ranks under the nccl (RCCL) backend share devices: %r" % (world, rank_devices))

    if rank == 0:
        n_elem = b_rank * D                                     # = masked images per step = Monotonic elements
        macs = (1 + COND) * INT_NET[0] + sum(a * b for a, b in zip(INT_NET[:-1], INT_NET[1:])) + INT_NET[-1]
        # algorithmic flop per launch (SURVEY.md 8d / DESIGN.md 4); padding and recompute are NOT counted as work
        CONV1, CONV2 = 97344, 1327104                           # MACs per 28x28 image (MLP.py:36-41)
        # round 5: x is frozen in a training step, so the cotangent of a masked copy is needed at the columns with dP/dA != 0
        # only (17 172 / 784 = 21.9 per image at the prior, 9 taps x 16 channels each) -- the algorithmic count of the conv
        # backward shrinks with it: dW2 + da1 (conv2-sized) + dW1 (conv1-sized) + de at those columns
        de_macs = 17172. / 784. * 144.
        work = {
            "gnf_mnistcnn_conv_bwd": ("cnn_bwd_wino_k (conv backward: dW2 and da1 as Winograd F(2x2,3x3) on MFMA, dW1, de at the "
                                      "columns with dP/dA != 0; conv1 recomputed)", 2. * (2 * CONV2 + CONV1 + de_macs) * n_elem),
            "gnf_mnistcnn_conv_fwd": ("cnn_fwd_wino_k (conv1+ReLU, conv2 as Winograd F(2x2,3x3) on MFMA, maxpool)", 2. * (CONV1 + CONV2) * n_elem),
            "gnf_monotonic_fwd": ("mono_fwd_x_k<3,2> (Clenshaw-Curtis quadrature; 3 tiles on MFMA, units 48-49 peeled onto the VALU)",
                                  2. * macs * (S_NODES + 2) * n_elem),
            "gnf_monotonic_bwd": ("mono_bwd_pair_x_k<3,3,2> (two nodes per pass, weight gradients in-kernel; since round 6 recompute and data "
                                  "gradient of the main blocks as 3 x bf16 splits) + unpack",
                                  4. * macs * (S_NODES + 2) * n_elem),
        }
        # PMC counters cannot be read from inside this process: HBM bytes per launch, MFMA instructions per image, other VALU
        # instructions per MFMA and the effective clock of the hand-written kernels are READ from
        # profiles/r06_bench_inputs.json, which tools/make_bench_inputs.py writes from rocprofv3 --pmc passes over the same
        # kernels at the same per-GPU size (null when the file is missing or the size differs)
        pmc, pmc_source = {}, None
        try:
            with open(os.path.join(ROOT, "profiles", PMC_INPUTS)) as f:
                pmc_file = json.load(f)
            if pmc_file.get("n_images") == n_elem:
                pmc = pmc_file["kernels"]
                pmc_source = "profiles/" + PMC_INPUTS + " (" + pmc_file["how"][:60] + "...)"
        except (OSError, ValueError, KeyError):
            pass
        # `frac` prices a kernel at the flop it must EXECUTE IN THE ALGORITHM IT IMPLEMENTS (review of round 5, item 1).  The
        # conv pair implements Winograd F(2x2,3x3): 16 multiplies per 2x2 output tile and channel pair where the direct form
        # has 36, and the backward skips the de / T products at the structural zeros -- dividing the DIRECT-convolution count
        # of SURVEY.md 8(d) by the time gave 1.01 / 1.15 "of peak" in round 5.  Executed flop = MFMA instructions per image
        # (counter pass, profiles/) x 2048; without the counter file: the Winograd-domain count of the same contractions
        # (conv2-sized parts x 4/9, da1 on the 13x13 tile grid of the full correlation, T = W1^T dpre1 and dW1 dense, no
        # recompute).  The direct-convolution rate stays beside it as `*_direct_conv_equivalent`.  The Monotonic kernels are
        # priced at the algorithmic 2 M (S+2) / 4 M (S+2) of SURVEY.md 8(d): padding, peeled units and the backward's
        # recompute are issued but NOT counted, so their `frac` is below their `mfma_issue_frac`.
        WINO = 4. / 9.
        executed_fallback = {"gnf_mnistcnn_conv_fwd": 2. * (CONV1 + CONV2 * WINO) * n_elem,
                             "gnf_mnistcnn_conv_bwd": 2. * (CONV2 * WINO * (1. + 169. / 144.) + 2. * CONV1) * n_elem}
        # round 6: the Monotonic forward of the peeled [50]^3 net runs its 48 x 48 main blocks as six bf16 MFMA terms per product
        # (mono_fwd_x_k<split>): priced at what it EXECUTES against the dense bf16 peak, the algorithmic fp32 rate beside it
        lib = abi.load()
        mono_fwd_split = bool(lib.gnf_gemm_split_enabled()) and lib.gnf_monotonic_fwd_kernel().decode() == "mono_fwd_x_k<split>"
        # per (element, node) and hidden->hidden layer: 3 out tiles x 6 terms x 2 instructions (the 32- and the 16-wide part of the
        # 48-unit contraction, both on v_mfma_f32_16x16x32_bf16: 16 384 flop) per 16 elements
        mono_split_exec = (len(INT_NET) - 1) * 36 * 16384. / 16. * (S_NODES + 2) * n_elem
        kern = {}
        for k, (label, fl) in work.items():
            if k not in prof:
                continue
            sec = prof[k] * 1e-3
            entry = {"kernel": label, "ms": round(prof[k], 4), "unit": "TFLOP/s"}
            if k == "gnf_monotonic_fwd" and mono_fwd_split:
                entry["kernel"] = ("mono_fwd_x_k<3,2,split> (Clenshaw-Curtis quadrature; the 48 x 48 main blocks as exact 3 x bf16 splits on "
                                   "v_mfma_f32_16x16x32_bf16, units 48-49 peeled onto the VALU)")
                if k in pmc and pmc[k].get("mfma_flop") == 16384:
                    ex, entry["frac_basis"] = pmc[k]["mfma_per_image"] * 16384. * n_elem, \
                        "executed: bf16 MFMA instructions per element (%s) x 16384 flop, against the dense bf16 peak" % PMC_INPUTS
                else:
                    ex, entry["frac_basis"] = mono_split_exec, "executed: six bf16 terms per product of the padded main blocks (no counter file), against the dense bf16 peak"
                entry["bound"], entry["peak"] = "mfma (bf16, 6 terms per fp32 product)", PEAK_BF16_TFLOPS
                entry["achieved"] = round(ex / sec / 1e12, 2)
                entry["fp32_equivalent_TFLOPs_algorithmic"] = round(fl / sec / 1e12, 2)
                entry["frac"] = round(entry["achieved"] / PEAK_BF16_TFLOPS, 4)
                kern[k] = entry
                continue
            if k in executed_fallback:
                if k in pmc:
                    ex, entry["frac_basis"] = pmc[k]["mfma_per_image"] * 2048. * n_elem, \
                        "executed: MFMA instructions per image (%s) x 2048 flop" % PMC_INPUTS
                else:
                    ex, entry["frac_basis"] = executed_fallback[k], "executed: Winograd-domain flop of the contractions (no counter file)"
                entry["achieved"] = round(ex / sec / 1e12, 2)
                entry["achieved_direct_conv_equivalent"] = round(fl / sec / 1e12, 2)
                entry["frac_direct_conv_equivalent"] = round(fl / sec / 1e12 / PEAK_F32_TFLOPS, 4)
            else:
                entry["frac_basis"] = "algorithmic: SURVEY.md 8(d) flop (padding, peeled units, recompute not counted)"
                entry["achieved"] = round(fl / sec / 1e12, 2)
            entry["frac"] = round(entry["achieved"] / PEAK_F32_TFLOPS, 4)
            kern[k] = entry
        dom = max((k for k in kern), key=lambda k: prof[k])     # dominant hand-written kernel by time
        if dom != DOMINANT_OP:
            if len(set(rank_devices)) < world:                  # ranks SHARING a device (the gloo test of the N > 1 path on a
                dom = DOMINANT_OP                               # one-GPU box): per-kernel times are interleaving noise there
            else:
                raise SystemExit("the kernel timed inside the region (%s) is not the dominant one (%s)" % (DOMINANT_OP, dom))
        for k, entry in kern.items():
            if not 0. < entry["frac"] <= 1.:
                raise SystemExit("roofline fraction of %s outside (0, 1]: %r -- the flop basis is wrong" % (k, entry["frac"]))
        achieved = kern[dom]["achieved"]
        # which device every rank ran on (review item 6): under the nccl backend two ranks on one device are an error
        out = {
            "metric": "samples/sec (fwd+log|detJ|+bwd) MNIST d=784 Monotonic-DAG",
            "value": b_rank * world * args.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # round 6: the fc1 forward / data-gradient products run on the bf16 matrix pipe with fp32 accuracy (exact 3 x bf16 operand
            # splits, six cross terms, fp32 accumulate: gnf_gemm_split.hip, more accurate against fp64 than the fp32-MFMA kernels
            # they replace, profiles/r06_split_bf16_error.txt); everything else on v_mfma_f32_*.  GNF_TRUE_F32=1 turns it off.
            "mfma_operands": "3xbf16 split (fc1, Monotonic main blocks), f32 elsewhere" if abi.load().gnf_gemm_split_enabled() else "f32",
            "parity_note": "UMNN 1.0 parity unpinned: the Clenshaw-Curtis integral of the Monotonic normalizer is "
                           "checked against this repo's own restatement + mathematics (CPU: tests/test_oracle_math.py; the HIP "
                           "kernels themselves against fp64 adaptive quadrature: tests/test_gpu_integral_pin.py), not against "
                           "the absent package; everything else on the path is pinned by reference-generated fixtures",
            "replicas_identical": replicas_identical, "dist_backend": backend if collective else None,
            "rccl_world_size": (dist.get_world_size() if collective and backend == "nccl" else None),
            "rank_devices": rank_devices,
            "collective_forced_at_world_1": bool(force and world == 1),
            "allreduce_ms_per_step": round(allreduce_ms, 4) if allreduce_ms else None,
            "ms_per_step_per_rank": [round(v, 4) for v in per_rank_ms],
            "config": {"workload": "cfg4: MNIST d=784, MonotonicNormalizer[50,50,50] cond 30 S=20 + DAGConditioner("
                                   "MNISTCNN->30, prior_A_kernel=2, hot_encoding=False, Gumbel gate T=1), "
                                   "b_size=%d per GPU%s; step = fwd+logdet+NLL+bwd+allreduce+Adam"
                                   % (b_rank, " (global batch fixed at %d: strong scaling)" % args.global_batch if args.global_batch else ""),
                       "global_batch": b_rank * world, "parallelism": "dp%d" % world},
            "roofline": dict({"bound": "mfma", "ms_per_launch": kern[dom]["ms"], "peak": PEAK_F32_TFLOPS, "traffic": None},
                             **{f: v for f, v in kern[dom].items() if f != "ms"}),
            "roofline_other": [v for k, v in kern.items() if k != dom],
            "ops_ms": {k: round(v, 4) for k, v in prof.items()},
            "ops_ms_source": {DOMINANT_OP: "HIP events on the launch stream in every one of the %d timed steps" % args.steps,
                              "others": "HIP events on the launch stream in %d untimed steps right behind the timed region "
                                        "(an event pair idles the GPU for ~11 us per entry point and step)" % ops_steps,
                              DOMINANT_OP + "_in_the_untimed_pass": round(dom_ms_untimed_pass, 4) if dom_ms_untimed_pass else None},
        }
        if pmc_source:
            out["roofline"]["pmc_source"] = pmc_source
        for v in out["roofline_other"]:
            v["measured"] = "%d untimed steps behind the timed region" % ops_steps
        alg_bytes = {"gnf_mnistcnn_conv_bwd": n_elem * (784 * 4 + 2304 * 5 + 32 * 4),     # e, g_pooled + argmax, the compact de
                     "gnf_mnistcnn_conv_fwd": n_elem * (784 * 4 + 2304 * 5)}

        def issued(k, entry):
            """what the kernel ISSUES, next to `frac`.  mfma_issue_frac: MFMA instructions per image x 2048 flop / time / peak
            = the share of the f32-MFMA issue slots in use at the nominal 2.4 GHz behind the 157.3 TFLOP/s (for the conv pair
            this IS `frac`); frac_of_peak_at_clock: the same against the peak at the clock the kernel ran at in this run
            (cycles per launch from the counter pass / live launch time)."""
            p = pmc.get(k)
            if not p:
                return
            if entry.get("peak") == PEAK_BF16_TFLOPS:             # split-bf16 kernel: bf16 issue slots; counters of the fp32 kernel do not apply
                if p.get("mfma_flop") != 16384:
                    return
                mi = p["mfma_per_image"] * 16384. * n_elem / (prof[k] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS
                entry["mfma_issue_frac"] = round(mi, 4)
                entry["valu_per_mfma"] = round(p["valu_per_mfma"], 3)
                # next to bf16 MFMAs about two VALU instructions per MFMA issue for free and the rest cost ~3 cycles each
                # (tools/mfma_k16_rate.hip, profiles/r06_mfma_k16_rate.txt): 16.4 / (16.4 + 3 max(0, V - 2))
                entry["issue_frac_ceiling_shared_alu"] = round(16.4 / (16.4 + 3. * max(0., p["valu_per_mfma"] - 2.)), 3)
                if p.get("cycles_per_launch"):
                    ghz = p["cycles_per_launch"] / (prof[k] * 1e6)
                    entry["effective_clock_GHz"] = round(ghz, 3)
                    entry["effective_clock_GHz_in_pmc_pass"] = round(p["effective_clock_GHz_in_pmc_pass"], 3)
                    entry["frac_of_peak_at_clock"] = round(mi * NOMINAL_GHZ / ghz, 4)
                if p.get("mfma_pipe_busy_frac_of_simd_cycles") is not None:
                    entry["mfma_pipe_busy_frac_of_simd_cycles"] = p["mfma_pipe_busy_frac_of_simd_cycles"]
                return
            if p.get("mfma_flop") == "mixed":                     # the peeled backward with its chain on bf16 MFMAs, weight gradients fp32
                if lib.gnf_monotonic_bwd_kernel().decode() != "mono_bwd_pair_x_k<split>":
                    return
                nb, nf = p["mfma_bf16_per_image"], p["mfma_per_image"] - p["mfma_bf16_per_image"]
                # share of the matrix pipe's time at the nominal clock: every instruction at its own peak rate
                mi = (nf * 2048. / PEAK_F32_TFLOPS + nb * 16384. / PEAK_BF16_TFLOPS) * n_elem / (prof[k] * 1e-3) / 1e12
                entry["mfma_issue_frac"] = round(mi, 4)
                entry["mfma_issue_frac_basis"] = "fp32 MFMAs (weight gradients) at 157.3 + bf16 MFMAs (recompute, data gradient) at 2500 TFLOP/s"
                entry["valu_per_mfma"] = round(p["valu_per_mfma"], 3)
                cyc = (nf * 32.5 + nb * 16.4) / (nf + nb)         # mean cycles per MFMA; ~2 VALU instructions per bf16 MFMA are free
                entry["issue_frac_ceiling_shared_alu"] = round(cyc / (cyc + 3. * max(0., p["valu_per_mfma"] - 2. * nb / (nf + nb))), 3)
                if p.get("cycles_per_launch"):
                    ghz = p["cycles_per_launch"] / (prof[k] * 1e6)
                    entry["effective_clock_GHz"] = round(ghz, 3)
                    entry["effective_clock_GHz_in_pmc_pass"] = round(p["effective_clock_GHz_in_pmc_pass"], 3)
                    entry["frac_of_peak_at_clock"] = round(mi * NOMINAL_GHZ / ghz, 4)
                if p.get("mfma_pipe_busy_frac_of_simd_cycles") is not None:
                    entry["mfma_pipe_busy_frac_of_simd_cycles"] = p["mfma_pipe_busy_frac_of_simd_cycles"]
                if p.get("lds_bank_conflict_frac_of_lds_cycles") is not None:
                    entry["lds_bank_conflict_frac_of_lds_cycles"] = round(p["lds_bank_conflict_frac_of_lds_cycles"], 4)
                return
            if p.get("mfma_flop", 2048) != 2048:                  # counters of the split kernel, fp32 kernel running (GNF_TRUE_F32=1)
                return
            mi = p["mfma_per_image"] * 2048. * n_elem / (prof[k] * 1e-3) / 1e12 / PEAK_F32_TFLOPS
            entry["mfma_issue_frac"] = round(mi, 4)
            # f32 MFMA and the other VALU instructions share one ALU per SIMD on gfx950 (tools/mfma_pipe.hip, valu_cost.hip:
            # 32.5 cycles per v_mfma_f32_16x16x4_f32 + ~3 per VALU instruction, additive at 2, 3 and 4 wavefronts per SIMD),
            # so with V other VALU instructions per MFMA the issue fraction cannot exceed 32.5 / (32.5 + 3 V)
            entry["valu_per_mfma"] = round(p["valu_per_mfma"], 3)
            entry["issue_frac_ceiling_shared_alu"] = round(32.5 / (32.5 + 3. * p["valu_per_mfma"]), 3)
            # The clock of THIS run = cycles of a launch (GRBM_GUI_ACTIVE / 8 XCDs, counter pass) / the live launch time.  The
            # clock of the counter pass itself is not the clock of this run: under rocprofv3 --pmc the HBM-streaming kernels
            # take ~10 % longer at a ~10 % lower clock (same cycle count); rocm-smi reads 2.39 GHz during this benchmark.
            if p.get("cycles_per_launch"):
                ghz = p["cycles_per_launch"] / (prof[k] * 1e6)
                entry["effective_clock_GHz"] = round(ghz, 3)
                entry["effective_clock_GHz_in_pmc_pass"] = round(p["effective_clock_GHz_in_pmc_pass"], 3)
                entry["frac_of_peak_at_clock"] = round(mi * NOMINAL_GHZ / ghz, 4)
            if p.get("mfma_pipe_busy_frac_of_simd_cycles") is not None:
                entry["mfma_pipe_busy_frac_of_simd_cycles"] = p["mfma_pipe_busy_frac_of_simd_cycles"]
            if p.get("lds_bank_conflict_frac_of_lds_cycles") is not None:
                entry["lds_bank_conflict_frac_of_lds_cycles"] = round(p["lds_bank_conflict_frac_of_lds_cycles"], 4)
            if k in alg_bytes:
                entry["traffic"] = p["hbm_bytes_per_launch"]
                entry["traffic_algorithmic"] = float(alg_bytes[k])
        issued(dom, out["roofline"])
        for k, entry in kern.items():
            if k != dom:
                issued(k, entry)
        out["secondary"] = {"fwd_bwd_only_samples_per_s": round(b_rank * world / t_fb, 1) if t_fb else None,
                            "full_step_S_mix_20_29_samples_per_s": round(b_rank * world / t_mix, 1) if t_mix else None,
                            "full_step_frozen_deterministic_gate_samples_per_s":
                                round(b_rank * world / t_det, 1) if t_det else None,
                            "eval_forward_S150_samples_per_s": round(b_rank * world / t_eval, 1) if t_eval else None,
                            "eval_forward_S150_frozen_gate_samples_per_s":
                                round(b_rank * world / t_evaldet, 1) if t_evaldet else None,
                            "frozen_gate_error": det_error, "frozen_gate_eval_error": evaldet_error,
                            "note": "10 steps each, wall clock between barriers, max over ranks"}
        out["measured_peaks"] = measured_peaks(dev)
        out["roofline"]["frac_of_measured_peak"] = round(achieved / out["measured_peaks"]["mfma_f32_TFLOPs"], 4)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if collective:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST line of stdout: RCCL announces itself through C stdio ("Librccl path : ..."), whose
        # buffer would otherwise be flushed at process exit, behind Python's
        _flush_c_stdio()
        print(json.dumps(out), flush=True)


def _device_id(index):
    """'uuid=... pci=dddd:bb:dd.0 name=...' of a visible device (torch.cuda.get_device_properties)"""
    pr = torch.cuda.get_device_properties(index)
    parts = []
    if getattr(pr, "uuid", None) is not None:
        parts.append("uuid=%s" % pr.uuid)
    if hasattr(pr, "pci_bus_id"):
        parts.append("pci=%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0)))
    parts.append("name=%s" % pr.name)
    return " ".join(parts)


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


if __name__ == "__main__":
    main()
