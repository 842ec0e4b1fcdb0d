"""Monotonic forward of the wide integrand nets, split-bf16 kernel (gnf_monotonic_fwd) against the fp32-MFMA kernel
(gnf_monotonic_fwd_f32), alternating on one box:  python tools/bench_mono_split.py
  cfg5: BSDS300 d = 63, B = 50 000, [150]^3, S = 20 (training) and S = 150 on 5 000 rows (evaluation)
  cfg2: POWER d = 6, B = 10 000, [100]^3, S = 20 / 150"""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
from gnf_hip import abi, ops
from gnf_hip.abi import call, ptr, stream
from _warm import warm_gpu
DEV = "cuda:0"


def params(hidden, c):
    dims = [1 + c] + hidden + [1]
    ps = []
    for i in range(len(dims) - 1):
        b = 1. / dims[i] ** .5
        ps += [((torch.rand(dims[i + 1], dims[i]) * 2 - 1) * b * 1.7).to(DEV), ((torch.rand(dims[i + 1]) * 2 - 1) * b).to(DEV)]
    return ps


def main():
    warm_gpu()
    for tag, hidden, c, B, d, S in [("cfg4 train", [50] * 3, 30, 100, 784, 20), ("cfg4 eval ", [50] * 3, 30, 100, 784, 150),
                                    ("cfg5 train", [150] * 3, 30, 50000, 63, 20), ("cfg5 eval ", [150] * 3, 30, 5000, 63, 150),
                                    ("cfg2 train", [100] * 3, 30, 10000, 6, 20), ("cfg2 eval ", [100] * 3, 30, 10000, 6, 150)]:
        ps = params(hidden, c)
        x = torch.randn(B, d, device=DEV) * 2.
        h = torch.randn(B, d, c, device=DEV)
        net = ops._mono_net(ps)
        pack = ops._mono_pack(net, x)
        w, t = ops.cc_rule(S, x.device)
        z, jac = torch.empty_like(x), torch.empty_like(x)
        res = {}
        for rep in range(3):
            for entry in ("gnf_monotonic_fwd", "gnf_monotonic_fwd_f32"):
                def run():
                    call(entry, ptr(pack), ctypes.byref(net), ptr(x), ptr(h), h.stride(0), h.stride(1), h.stride(2), ptr(w), ptr(t), S,
                         ptr(z), ptr(jac), B, d, stream())
                run(); torch.cuda.synchronize()
                n = 5 if B * d * S > 3e7 else 20
                t0 = time.perf_counter()
                for _ in range(n):
                    run()
                torch.cuda.synchronize()
                res.setdefault(entry, []).append((time.perf_counter() - t0) / n * 1e3)
        if S <= 30:                                                       # training shapes: the backward too
            gz, gjac = torch.randn_like(x), torch.randn_like(x)
            gx, gh = torch.empty_like(x), torch.empty_like(h)
            gp = [torch.empty_like(p) for p in ps]
            nl = net.nl
            gW = (ctypes.c_void_p * nl)(*[gp[2 * l].data_ptr() for l in range(nl)])
            gb = (ctypes.c_void_p * nl)(*[gp[2 * l + 1].data_ptr() for l in range(nl)])
            nbytes = abi.load().gnf_monotonic_bwd_ws_bytes(ctypes.byref(net), S, B, d)
            ws = torch.empty(max(nbytes // 4, 1), device=DEV)
            for rep in range(3):
                for entry in ("gnf_monotonic_bwd", "gnf_monotonic_bwd_f32"):
                    def runb():
                        call(entry, ptr(pack), ctypes.byref(net), ptr(x), ptr(h), h.stride(0), h.stride(1), h.stride(2), ptr(w), ptr(t), S,
                             ptr(gz), ptr(gjac), ptr(gx), ptr(gh), gh.stride(0), gh.stride(1), gh.stride(2), gW, gb,
                             ctypes.c_void_p(ws.data_ptr()), ws.numel() * 4, B, d, stream())
                    runb(); torch.cuda.synchronize()
                    n = 3 if B * d > 1e6 else 20
                    t0 = time.perf_counter()
                    for _ in range(n):
                        runb()
                    torch.cuda.synchronize()
                    res.setdefault(entry, []).append((time.perf_counter() - t0) / n * 1e3)
            sb, fb = min(res["gnf_monotonic_bwd"]), min(res["gnf_monotonic_bwd_f32"])
            print("%s backward: split chain %.3f ms (%s)  fp32-MFMA %.3f ms (%s)  x%.2f" % (
                tag, sb, " ".join("%.3f" % v for v in res["gnf_monotonic_bwd"]), fb, " ".join("%.3f" % v for v in res["gnf_monotonic_bwd_f32"]), fb / sb))
        macs = sum(a * b for a, b in zip(hidden[:-1], hidden[1:]))        # hidden->hidden MACs per evaluation
        ev = B * d * (S + 2)
        s_, f_ = min(res["gnf_monotonic_fwd"]), min(res["gnf_monotonic_fwd_f32"])
        print("%s %s B=%d d=%d S=%d: split-bf16 %.3f ms (%s)  fp32-MFMA %.3f ms (%s)  x%.2f   hidden products %.1f / %.1f TFLOP/s fp32-equivalent"
              % (tag, hidden, B, d, S, s_, " ".join("%.3f" % v for v in res["gnf_monotonic_fwd"]), f_,
                 " ".join("%.3f" % v for v in res["gnf_monotonic_fwd_f32"]), f_ / s_, 2 * macs * ev / s_ / 1e9, 2 * macs * ev / f_ / 1e9))


if __name__ == "__main__":
    main()
