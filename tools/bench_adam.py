"""The fused Adam launch at the parameter count of BASELINE cfg3 (3.2 M fp32; GNF_ADAM_N for others): both entry points, for
`bash tools/kstats.sh <dir> tools/bench_adam.py` (rocprofv3 durations) or HIP events.    python tools/bench_adam.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi, ops
if os.environ.get('GNF_AB_LIB'):
    abi.LIB_PATH = os.path.join(ROOT, os.environ['GNF_AB_LIB'])
n = int(os.environ.get('GNF_ADAM_N', 4510240 - 1305600))
dev = 'cuda:0'
p, g = torch.randn(n, device=dev), torch.randn(n, device=dev)
m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
step = torch.zeros(2, dtype=torch.int32, device=dev)
for name, fn in (("gnf_adam_step", lambda t: ops.adam_step(p, g, m, v, t, lr=1e-3, weight_decay=1e-5)),
                 ("gnf_adam_step_dev", lambda t: ops.adam_step_dev(p, g, m, v, step, lr=1e-3, weight_decay=1e-5))):
    for t in range(1, 6): fn(t)
    abi.profile_enable((name,))
    for t in range(6, 46): fn(t)
    ms = abi.profile_collect()[name]
    print("%s n=%d %.1f us  %.2f TB/s (28 B per parameter)" % (name, n, ms * 1e3, 28. * n / ms / 1e9))
