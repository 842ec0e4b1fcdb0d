"""CPU experiment (no GPU): would Winograd F(4x4,3x3) in fp32 keep the conv2-sized contractions of the MNISTCNN front inside
the parity budget (atol 1e-6 + rtol 1e-5 with a margin of 3)?  conv2 (16 -> 16 channels, 3x3) of relu(conv1(e)) on the
reference-generated fixture tests/golden/mnistcnn.npz and on cfg4-like masked images, evaluated three ways in fp32 --
direct (torch), F(2x2,3x3) (what the kernels do), F(4x4,3x3) -- against an fp64 direct convolution.
    python tools/wino43_error.py"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# transform matrices (Lavin & Gray 2016)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
B2T = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
A2T = np.array([[1, 1, 1, 0], [0, 1, -1, -1]])
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
B4T = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]])
A4T = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]])


def wino(x, w, m, G, BT, AT):
    """x [N,C,H,W], w [O,C,3,3] -> valid conv, F(m x m, 3x3), everything in fp32 (weights transformed in fp64 first)."""
    N, C, H, W = x.shape
    O = w.shape[0]
    t = m + 2
    U = torch.from_numpy(np.einsum("ia,ocab,jb->ocij", G, w.double().numpy(), G)).float()        # [O,C,t,t]
    BTt, ATt = torch.from_numpy(BT).float(), torch.from_numpy(AT).float()
    oh, ow = H - 2, W - 2
    th, tw = (oh + m - 1) // m, (ow + m - 1) // m
    xp = F.pad(x, (0, tw * m + 2 - W, 0, th * m + 2 - H))
    patches = xp.unfold(2, t, m).unfold(3, t, m)                                                   # [N,C,th,tw,t,t]
    V = torch.einsum("ia,nchwab,jb->nchwij", BTt, patches, BTt)
    M = torch.einsum("ocij,nchwij->nohwij", U, V)
    Y = torch.einsum("ia,nohwab,jb->nohwij", ATt, M, ATt)                                          # [N,O,th,tw,m,m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, O, th * m, tw * m)[:, :, :oh, :ow]


def report(name, e, W1, b1, W2, b2):
    a1 = F.relu(F.conv2d(e, W1, b1))
    ref = F.conv2d(a1.double(), W2.double(), b2.double())
    outs = {"direct fp32": F.conv2d(a1, W2, b2), "F(2x2,3x3) fp32": wino(a1, W2, 2, G2, B2T, A2T) + b2.view(1, -1, 1, 1),
            "F(4x4,3x3) fp32": wino(a1, W2, 4, G4, B4T, A4T) + b2.view(1, -1, 1, 1)}
    print(name, " |conv2| max %.3g" % ref.abs().max().item())
    for k, v in outs.items():
        d = (v.double() - ref).abs()
        excess = d / (1e-6 + 1e-5 * ref.abs())
        print("   %-18s max abs err %.3e   max err / (atol 1e-6 + rtol 1e-5 |ref|) = %.2f   (budget with margin 3: 0.33)"
              % (k, d.max().item(), excess.max().item()))


g = np.load(os.path.join(ROOT, "tests", "golden", "mnistcnn.npz"))
keys = list(g.keys())
P = {k: torch.from_numpy(g[k]) for k in keys if g[k].dtype.kind == "f"}
W1 = next(v for k, v in P.items() if k.endswith("conv1.weight")); b1 = next(v for k, v in P.items() if k.endswith("conv1.bias"))
W2 = next(v for k, v in P.items() if k.endswith("conv2.weight")); b2 = next(v for k, v in P.items() if k.endswith("conv2.bias"))
x = next(v for k, v in P.items() if v.dim() == 2 and v.shape[1] == 784)
report("fixture mnistcnn.npz (%d images)" % x.shape[0], x.view(-1, 1, 28, 28), W1, b1, W2, b2)
torch.manual_seed(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "graphical-normalizing-flows_amd")]
from gnf_hip import configs
xs = configs.pseudo_mnist(torch.Generator().manual_seed(1234), 4, 784)
gate = torch.rand(4, 64, 784)                         # a soft (stochastic) gate: masked copies of logit-space pseudo-MNIST
e = (xs[:, None, :] * gate).reshape(-1, 1, 28, 28)
report("cfg4-like masked images (256), fixture weights", e, W1, b1, W2, b2)
W2r, b2r = torch.randn(16, 16, 3, 3) * (2. / 144) ** .5, torch.randn(16) * .05
report("cfg4-like masked images, He-initialised conv2", e, W1, b1, W2r, b2r)
