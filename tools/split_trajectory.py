"""End-to-end check of the split-bf16 fc1 kernels: the same cfg4 training run (same seeds, same Philox gate noise, Adam) with the
default library and with GNF_TRUE_F32=1 (the fp32-MFMA kernels), each in its own process; the loss of every step and the final
parameters are compared.     python tools/split_trajectory.py [steps] [B]          (prints the two trajectories side by side)
    python tools/split_trajectory.py --child <steps> <B>   (one run: prints one loss per line, then a parameter checksum)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']


def child(steps, B):
    import torch
    from gnf_hip import abi, dp, configs
    torch.manual_seed(0)
    flow = configs.build_cfg4_flow().to("cuda:0")
    dp.seed_gates(flow, 0)
    state = dp.FlatState(flow)
    x = configs.pseudo_mnist(torch.Generator().manual_seed(1234), B, 784).to("cuda:0")
    for nrm in flow.getNormalizers():
        nrm.nb_steps = 20
    print("split_enabled", abi.load().gnf_gemm_split_enabled())
    for _ in range(steps):
        loss = dp.train_step(flow, state, x, lr=1e-3, weight_decay=1e-5)
        print("loss %.9e" % loss.item())
    print("flat_sum %.12e flat_abs %.12e" % (state.flat.double().sum().item(), state.flat.double().abs().sum().item()))


def run(steps, B, true_f32):
    env = dict(os.environ)
    env.pop("GNF_TRUE_F32", None)
    if true_f32:
        env["GNF_TRUE_F32"] = "1"
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(steps), str(B)], env=env, capture_output=True,
                       text=True, timeout=1800)
    if r.returncode:
        raise SystemExit(r.stderr[-2000:])
    lines = r.stdout.splitlines()
    assert lines[0] == "split_enabled %d" % (0 if true_f32 else 1), lines[0]
    losses = [float(l.split()[1]) for l in lines if l.startswith("loss")]
    fs = [float(v) for v in lines[-1].split()[1::2]]
    return losses, fs


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
        B = int(sys.argv[2]) if len(sys.argv) > 2 else 100
        a, fa = run(steps, B, False)
        b, fb = run(steps, B, True)
        print("# cfg4, B = %d, %d Adam steps, same seeds: loss with the split-bf16 fc1 kernels | with GNF_TRUE_F32=1 | relative difference" % (B, steps))
        worst = 0.
        for i, (u, v) in enumerate(zip(a, b)):
            d = abs(u - v) / max(abs(v), 1e-30)
            worst = max(worst, d)
            print("step %3d   %.9e   %.9e   %.2e" % (i, u, v, d))
        print("# largest relative loss difference %.2e; parameter checksums: sum %.9e vs %.9e, sum|.| %.9e vs %.9e (relative %.2e)"
              % (worst, fa[0], fb[0], fa[1], fb[1], abs(fa[1] - fb[1]) / fb[1]))
