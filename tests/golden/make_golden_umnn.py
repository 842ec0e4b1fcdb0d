"""UMNN-pinned fixtures for the Monotonic normalizer -- READY TO RUN the day `UMNN==1.0` (requirements.txt:3 of the
reference) is importable in the build container; until then it exits with a message and writes nothing.

    python tests/golden/make_golden_umnn.py      # writes tests/golden/umnn_mono.npz

Runs the REFERENCE's own MonotonicNormalizer (models/Normalizers/MonotonicNormalizer.py:41-83, which calls
UMNN.NeuralIntegral / ParallelNeuralIntegral at :58,:61) on the `_mono_case` shapes of tests/test_gpu_parity.py and
records x, h, the integrand-net parameters, z, jac, and the gradients of sum(z*gz) + sum(log(jac)*gj) w.r.t. x, h and
every parameter, for both solvers, plus `inverse_transform` on a small case.  tests/test_gpu_umnn.py consumes the file
when it is present (and is skipped while it is absent): the moment this script has run, "UMNN parity unpinned" turns
into a pinned, reference-generated check of the quadrature kernels.  No placeholder module is used here: the real
package or nothing."""
import sys

import numpy as np
import torch

REF = "/root/reference"
CASES = [  # (B, d, c, hidden, S, seed, layout)  == tests/test_gpu_parity.py::test_monotonic_forward_backward_vs_oracle
    (9, 7, 5, [10], 20, 120, "contig"), (9, 7, 5, [16, 16], 21, 221, "made"), (9, 7, 30, [50, 50, 50], 20, 320, "contig"),
    (9, 7, 30, [50, 50, 50], 29, 329, "made"), (9, 7, 30, [100, 100, 100], 20, 320, "contig"),
    (9, 7, 30, [150, 150, 150], 20, 320, "made"), (9, 7, 30, [40, 64, 24], 15, 315, "contig"),
    (9, 7, 5, [200, 200], 22, 222, "contig"), (9, 7, 5, [48, 50, 50, 50], 20, 420, "contig"),
    (9, 7, 5, [100, 100, 100, 100], 20, 420, "made"),
]


def main():
    try:
        import UMNN  # noqa: F401
    except ImportError:
        print("UMNN is not importable here: nothing written (Monotonic z stays 'UMNN 1.0 parity unpinned')")
        return 1
    sys.path.insert(0, REF)
    from models.Normalizers import MonotonicNormalizer
    out = {"umnn_version": np.array(getattr(UMNN, "__version__", "unknown"))}
    for k, (B, d, c, hidden, S, seed, layout) in enumerate(CASES):
        for solver in ("CC", "CCParallel"):
            torch.manual_seed(seed)
            norm = MonotonicNormalizer(hidden, c, nb_steps=S, solver=solver)
            x = (torch.randn(B, d) * 1.5).requires_grad_(True)
            hraw = (torch.randn(B, c * d) if layout == "made" else torch.randn(B, d, c)).requires_grad_(True)
            h = hraw.view(B, c, d).permute(0, 2, 1) if layout == "made" else hraw
            z, jac = norm(x, h)
            gz, gj = torch.randn(B, d), torch.randn(B, d)
            ((z * gz).sum() + (torch.log(jac) * gj).sum()).backward()
            pre = "c%d.%s." % (k, solver)
            out.update({pre + "cfg": np.array([B, d, c, S, seed, int(layout == "made")] + hidden),
                        pre + "x": x.detach().numpy(), pre + "hraw": hraw.detach().numpy(), pre + "z": z.detach().numpy(),
                        pre + "jac": jac.detach().numpy(), pre + "gz": gz.numpy(), pre + "gj": gj.numpy(),
                        pre + "gx": x.grad.numpy(), pre + "gh": hraw.grad.numpy()})
            for name, p in norm.integrand_net.named_parameters():
                out[pre + "p." + name] = p.detach().numpy()
                out[pre + "g." + name] = p.grad.numpy()
    torch.manual_seed(5)
    norm = MonotonicNormalizer([50, 50, 50], 30, nb_steps=30, solver="CC")
    x, h = torch.randn(11, 5) * 1.5, torch.randn(11, 5, 30)
    with torch.no_grad():
        z, _ = norm(x, h)
        xi = norm.inverse_transform(z, h)
    out.update({"inv.x": x.numpy(), "inv.h": h.numpy(), "inv.z": z.numpy(), "inv.x_inverse": xi.numpy()})
    for name, p in norm.integrand_net.named_parameters():
        out["inv.p." + name] = p.detach().numpy()
    import os
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "umnn_mono.npz"), **out)
    print("wrote umnn_mono.npz (%d arrays)" % len(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
