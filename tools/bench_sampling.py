"""MNIST sampling latency (cfg4 flow, B = 100): level-scheduled inversion of a post-processed DAG whose parents are the
window pixels preceding a pixel in raster order (109 levels), sparse vs dense embedding front.
Usage: python tools/bench_sampling.py"""
import sys, time, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
import bench
import os
if os.environ.get('GNF_AB_LIB'):                      # A/B against another build of the library (tools/*.bin)
    from gnf_hip import abi
    abi.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ['GNF_AB_LIB'])
DEV = "cuda:0"
flow = bench.build_flow().to(DEV)
cond = flow.getConditioners()[0]
# a DAG inside the prior's windows: parents = window pixels that precede the pixel in raster order
with torch.no_grad():
    A = cond.A.detach().clone()
    idx = torch.arange(784, device=DEV)
    A = A * (idx[None, :] < idx[:, None]).float()
    cond.A.data = A
    cond.post_process(zero_threshold=.1)
for n in flow.getNormalizers(): n.nb_steps = 20
z = torch.randn(100, 784, device=DEV) * .3
lv = cond.levels(cond.deterministic_importance())
print("levels", len(lv))
only = os.environ.get("BENCH_SAMPLING_ONLY")           # "sparse" / "dense": one front only (clean kernel traces)
for sparse in ((True, False) if only is None else (only == "sparse",)):
    cond.sparse_front = sparse
    for _ in range(2): x = flow.invert(z)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): x = flow.invert(z)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    with torch.no_grad():
        zb, _ = flow(x)
    print("sparse" if sparse else "dense", "invert ms", round(dt * 1e3, 2), "roundtrip err", float((zb - z).abs().max()))
