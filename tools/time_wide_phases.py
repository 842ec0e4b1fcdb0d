"""Debug: s_memtime stamps of one batch of mono_bwd_wide_k (workgroup 0, third group, fourth batch) per wavefront.
    bash tools/build_variant.sh wide_timing gnf_monotonic_wide.hip -DGNF_WIDE_TIMING ; python tools/time_wide_phases.py [H]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi
abi.LIB_PATH = os.path.join(ROOT, 'tools/wide_timing.bin')
from models import MonotonicNormalizer
H = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = 'cuda:0'
torch.manual_seed(0)
B, d, c, S = [int(v) for v in os.environ.get('GNF_MONO_SHAPE', '50000,63,30,20').split(',')]
norm = MonotonicNormalizer([H, H, H], c, nb_steps=S).to(dev)
x = torch.randn(B, d, device=dev, requires_grad=True); h = torch.randn(B, d, c, device=dev, requires_grad=True)
lib = abi.load()
lib.gnf_debug_wide_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
for it in range(3):
    lib.gnf_debug_wide_stamps(None, 1 if it == 2 else 0)
    z, jac = norm(x, h)
    (z.sum() + torch.log(jac).sum()).backward()
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
lib.gnf_debug_wide_stamps(buf, -1)
t = [[buf[w * 32 + i] for i in range(32)] for w in range(8)]
t0 = min(t[w][0] for w in range(8))
cn = {0: "start", 1: "F1 done", 2: "bar", 3: "pass1 done", 4: "relu+write", 5: "bar", 6: "pass2 done", 12: "dot+sred", 13: "bar",
      14: "dp2 written", 15: "bar", 16: "da2 done", 17: "gate", 18: "bar", 19: "dp1 written", 20: "bar", 21: "da1 done", 22: "gate", 23: "bar",
      30: "first layer done"}
dn = {0: "start", 1: "bar", 2: "bar", 3: "bar", 8: "bar(dp2)", 9: "dW2 done", 10: "bar", 11: "bar(dp1)", 12: "dW1 done", 13: "bar"}
for w in range(8):
    names = cn if w < 4 else dn
    print("wave", w, "  ".join("%s@%d" % (names[i], (t[w][i] - t0) * 1) for i in sorted(names) if t[w][i]))
