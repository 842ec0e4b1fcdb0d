"""Golden trajectories of the epoch-level DAG machinery, from the REFERENCE ITSELF (build container only).

    python tests/golden/make_golden_dual.py      # writes tests/golden/dag_dual.npz

Drives the reference's DAGConditioner.step() / update_dual_param() / post_process()
(models/Conditionners/DAGConditioner.py:76-92,196-260,273-293) through every branch on small problems and records,
after each call, the buffers and flags the drivers read (lambd, c, prev_trace, dag_const, l1_weight, alpha, exponent,
no_update, gate flags, requires_grad, is_invertible) and A.  The reference calls `nx.from_numpy_matrix`, which
networkx >= 3 removed; the name is aliased to its successor `nx.from_numpy_array` here (an alias, no arithmetic).
The reference's progress prints are swallowed.

Scenarios (each: an initial A, a list of (epoch, loss_avg) calls):
  dual      d=8  dense random A: exponent back-off (trace > 50), dual updates with c *= eta, skipped updates that count
                 no_update up, the forced update after 10 skips
  success   d=8  acyclic A: trace 0 -> post_process() (auto threshold) -> dag_const = l1_weight = 0; the next update takes
                 the acyclic else-branch -> is_invertible
  exponent  d=60 acyclic A: exponent 10 -> 60 before post-processing
  failure   d=8  a faint 2-cycle the fp32 trace cannot see, post_process() pinned to a threshold that keeps it: the
                 binarised graph has a cycle -> gate re-opened, A restored, c /= eta, lambd += c h, dag_const = 1
  reopen    d=8  constraints already off but A given a cycle: the else-branch re-opens the gate
"""
import contextlib
import io

import networkx as nx
import numpy as np
import torch

from make_golden import _import_reference, npy, save

FIELDS = ("lambd", "c", "prev_trace", "dag_const", "l1_weight", "alpha")


def snapshot(c):
    vec = [float(getattr(c, f)) for f in FIELDS]
    vec += [float(c.exponent), float(c.no_update), float(c.stoch_gate), float(c.noise_gate), float(c.s_thresh),
            float(c.h_thresh), float(c.A.requires_grad), float(bool(c.is_invertible))]
    return np.array(vec, dtype=np.float64), npy(c.A)


def drive(c, calls):
    states, As = [], []
    for epoch, loss_avg in calls:
        if c.A.requires_grad and c.A.grad is None:
            c.A.grad = torch.zeros_like(c.A)             # step() prints A.grad statistics (:274-276)
        with contextlib.redirect_stdout(io.StringIO()):
            c.step(epoch, torch.tensor(loss_avg))
        s, A = snapshot(c)
        states.append(s)
        As.append(A)
    return np.stack(states), np.stack(As)


def scenarios():
    g = torch.Generator().manual_seed(2024)
    out = {}
    d = 8
    out["dual"] = dict(d=d, l1=.3, nb_epoch_update=1, A0=torch.ones(d, d) * 1.5 + torch.randn(d, d, generator=g) * .02,
                       calls=[(0, 10.), (1, 1e6), (2, 1e6), (3, 1e-9)] + [(4 + i, 1e-9) for i in range(11)] + [(20, 1e6)],
                       threshold=None)
    tri = torch.tril(torch.rand(d, d, generator=g) + .8, -1)
    out["success"] = dict(d=d, l1=.2, nb_epoch_update=2, A0=tri.clone(), calls=[(1, 5.), (2, 5.), (4, 5.), (6, 5.)],
                          threshold=None)
    d60 = 60
    tri60 = torch.tril(torch.rand(d60, d60, generator=g) + .8, -1) * (torch.rand(d60, d60, generator=g) < .1).float()
    out["exponent"] = dict(d=d60, l1=0., nb_epoch_update=1, A0=tri60, calls=[(1, 5.), (2, 5.)], threshold=None)
    faint = torch.tril(torch.rand(d, d, generator=g) + .8, -1)
    faint[:, 0] = 0.
    faint[0, 1] = 3e-4
    faint[1, 0] = 3e-4                                     # 2-cycle 0 <-> 1, invisible to the fp32 trace
    out["failure"] = dict(d=d, l1=.1, nb_epoch_update=1, A0=faint, calls=[(1, 5.), (2, 5.)], threshold=1e-9)
    cyc = torch.zeros(d, d)
    cyc[0, 1] = cyc[1, 2] = cyc[2, 0] = 1.
    out["reopen"] = dict(d=d, l1=0., nb_epoch_update=1, A0=cyc, calls=[(1, 5.), (2, 5.)], threshold=None, off=True)
    return out


def main():
    if not hasattr(nx, "from_numpy_matrix"):
        nx.from_numpy_matrix = nx.from_numpy_array          # alias for the function networkx 3 renamed
    _import_reference()
    from models.Conditionners import DAGConditioner
    arrays = {}
    for name, sc in scenarios().items():
        torch.manual_seed(0)
        c = DAGConditioner(sc["d"], [8], 2, l1=sc["l1"], nb_epoch_update=sc["nb_epoch_update"], A_prior=sc["A0"].clone())
        if sc["threshold"] is not None:
            ref_pp, th = c.post_process, sc["threshold"]
            c.post_process = lambda zero_threshold=None: ref_pp(th)
        if sc.get("off"):                                   # the state a successful post-processing leaves
            c.dag_const = torch.tensor(0.)
            c.l1_weight = torch.tensor(0.)
            c.stoch_gate, c.noise_gate, c.s_thresh, c.h_thresh = False, False, False, 0.
            c.A.requires_grad = False
        s0, A_init = snapshot(c)
        states, As = drive(c, sc["calls"])
        arrays[name + ".A0"] = npy(sc["A0"])
        arrays[name + ".A_init"] = A_init                   # after the constructor's constrainA
        arrays[name + ".cfg"] = np.array([sc["d"], sc["l1"], sc["nb_epoch_update"],
                                          -1. if sc["threshold"] is None else sc["threshold"], float(bool(sc.get("off")))])
        arrays[name + ".calls"] = np.array(sc["calls"], dtype=np.float64)
        arrays[name + ".state0"] = s0
        arrays[name + ".states"] = states
        arrays[name + ".A"] = As
        print(name)
        for call, s in zip(sc["calls"], states):
            print("   ", call, np.array2string(s, precision=4, max_line_width=200))
    arrays["fields"] = np.array(list(FIELDS) + ["exponent", "no_update", "stoch_gate", "noise_gate", "s_thresh",
                                                "h_thresh", "A.requires_grad", "is_invertible"])
    save("dag_dual", **arrays)


if __name__ == "__main__":
    main()
