"""Round 5: why is bench.py's `fwd_bwd_only` secondary slower than the full step?  Wall-clock per iteration of (a) dp.train_step,
(b) flow(x) -> loss -> backward with p.grad = None in front (the secondary's loop), (c) the same with the cached ones cotangent,
(d) the same without clearing the gradients, (e) = (b) on the separate gate / conv nodes (cond.fused_front = False)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import dp
from gnf_hip.configs import baseline_config
flow, x = baseline_config("cfg4")
for nrm in flow.getNormalizers():
    nrm.nb_steps = 20
state = dp.FlatState(flow)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def fb(clear=True, one=False):
    if clear:
        for p in flow.parameters():
            p.grad = None
    z, ld = flow(x)
    loss = flow.loss(z, ld)
    loss.backward(dp._one(loss) if one else None)


print("train_step            %.3f ms" % timed(lambda: dp.train_step(flow, state, x)))
print("fwd+bwd, grads cleared %.3f ms" % timed(lambda: fb()))
print("  + cached ones        %.3f ms" % timed(lambda: fb(one=True)))
print("  grads NOT cleared    %.3f ms" % timed(lambda: fb(clear=False)))
state.drop_grads()
print("  drop_grads() first   %.3f ms" % timed(lambda: (state.drop_grads(), fb(clear=False))))
for c in flow.getConditioners():
    c.fused_front = False
print("separate nodes:  train_step %.3f ms   fwd+bwd %.3f ms" % (timed(lambda: dp.train_step(flow, state, x)), timed(lambda: fb())))
