"""Golden vectors for the sampled-ordering MADE (reference models/Conditionners/AutoregressiveConditioner.py:70-101:
`random=True`, `num_masks > 1`), from the REFERENCE ITSELF (build container only).

    python tests/golden/make_golden_made_random.py      # writes tests/golden/made_random.npz

Two nets (a permuted input order with three cycling masks; the natural input order with sampled hidden degrees): the masks
and the input-order map after each of four consecutive update_masks() calls, and for the first net the output and the
gradients of sum(h * gh) under its second mask set.
"""
import numpy as np
import torch

from make_golden import _import_reference, npy, save, state_np


def main():
    _import_reference()
    from models.Conditionners.AutoregressiveConditioner import MADE
    arrays = {}
    for tag, nin, hidden, nout, num_masks, natural in (("perm", 5, [7, 6], 10, 3, False), ("nat", 4, [9], 4, 2, True)):
        torch.manual_seed(11)
        net = MADE(nin, hidden, nout, num_masks=num_masks, natural_ordering=natural, random=True)
        arrays[tag + ".cfg"] = np.array([nin, nout, num_masks, int(natural)] + hidden)
        arrays.update(state_np(net, tag + ".p."))                   # weights, biases and the FIRST mask set
        for call in range(4):
            if call:
                net.update_masks()
            layers = [l for l in net.net.modules() if hasattr(l, "mask")]
            for k, l in enumerate(layers):
                arrays["%s.mask%d.%d" % (tag, call, k)] = l.mask.numpy().copy()
            arrays["%s.imap%d" % (tag, call)] = np.asarray(net.i_map).copy()
            if tag == "perm" and call == 1:
                x = torch.randn(6, nin, requires_grad=True)
                h = net(x)
                gh = torch.randn_like(h)
                (h * gh).sum().backward()
                arrays.update({tag + ".x": npy(x), tag + ".h": npy(h), tag + ".gh": npy(gh), tag + ".gx": npy(x.grad)})
                for name, p in net.named_parameters():
                    arrays[tag + ".g." + name] = npy(p.grad)
    save("made_random", **arrays)


if __name__ == "__main__":
    main()
