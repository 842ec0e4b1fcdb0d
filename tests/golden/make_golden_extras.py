"""Golden vectors for corners of the plug-in API, from the REFERENCE ITSELF (build container only).

    python tests/golden/make_golden_extras.py      # writes tests/golden/dag_clipped_normal.npz

dag_clipped_normal: DAGConditioner.forward with `gumble = False`, i.e. the clipped-normal branch of stochastic_gate
(models/Conditionners/DAGConditioner.py:104-111), soft- and hard-thresholded importance, with the N(0,1) samples the
reference drew (torch.manual_seed before the call; one torch.randn(importance.shape) per forward).  Recorded: x, A,
embedding-net parameters, the samples, h and the gradients of sum(h * gh) w.r.t. x, A and the parameters.
"""
import numpy as np
import torch

from make_golden import _import_reference, npy, save, state_np


def main():
    _import_reference()
    from models.Conditionners import DAGConditioner
    arrays = {}
    for tag, h_thresh, hot in (("soft", 0., False), ("hard_hot", .4, True)):
        torch.manual_seed(31)
        B, d = 5, 7
        c = DAGConditioner(d, [12, 12], 3, h_thresh=0., hot_encoding=hot)
        with torch.no_grad():
            c.A.copy_(torch.rand(d, d) * 1.4 * (1 - torch.eye(d)))
        c.h_thresh = h_thresh
        c.gumble = False
        x = torch.randn(B, d, requires_grad=True)
        torch.manual_seed(77)
        n = torch.randn(B, d, d)                       # what stochastic_gate draws next
        torch.manual_seed(77)
        h = c(x)
        gh = torch.randn_like(h)
        (h * gh).sum().backward()
        arrays.update({tag + ".x": npy(x), tag + ".n": n.numpy(), tag + ".h": npy(h), tag + ".gh": npy(gh),
                       tag + ".gx": npy(x.grad), tag + ".gA": npy(c.A.grad), tag + ".cfg": np.array([h_thresh, float(hot)])})
        arrays.update(state_np(c, tag + ".p."))
        for k, p in c.named_parameters():
            if k != "A":
                arrays[tag + ".g." + k] = npy(p.grad)
    save("dag_clipped_normal", **arrays)


if __name__ == "__main__":
    main()
