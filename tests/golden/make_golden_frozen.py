"""Golden vectors of the MNIST DAG flow AFTER the DAG phase, from the REFERENCE ITSELF (build container only).

    python tests/golden/make_golden_frozen.py      # writes tests/golden/flow_mnist_affine_dag_frozen.npz

Same flow, parameters and input as flow_mnist_affine_dag.npz (they are read from that file), with the conditioner in the
state the reference's DAGConditioner.post_process() leaves (DAGConditioner.py:228-244: stoch_gate / noise_gate /
s_thresh off, h_thresh 0, A binarised and frozen; the kernel-2 prior is already binary).  post_process itself is not
called: its networkx 2 calls do not exist in this container.  Recorded: z, log-det, loss and the gradients of the embedding
net (first 8 rows of the 128 x 2304 fc1 weight gradient).  This is the case the sparse masked-image kernels cover.
"""
import numpy as np
import torch

from make_golden import _import_reference, npy, save, OUT


def main():
    models = _import_reference()
    from models.NormalizingFlowFactories import buildMNISTNormalizingFlow
    from models.Normalizers import AffineNormalizer
    g = dict(np.load(OUT + "/flow_mnist_affine_dag.npz"))
    torch.manual_seed(14)
    f = buildMNISTNormalizingFlow([1], AffineNormalizer, {}, l1=0., nb_epoch_update=10, hot_encoding=False,
                                  prior_kernel=2)
    sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}
    cond = f.steps[0].conditioner
    sd["steps.0.conditioner.A"] = cond.A.detach().clone()
    f.load_state_dict(sd)
    cond.stoch_gate, cond.noise_gate, cond.s_thresh, cond.h_thresh = False, False, False, 0.
    cond.A.data = (cond.A.data != 0).float()
    cond.A.requires_grad = False
    z, ld = f(torch.from_numpy(g["x"]).clone())
    loss = f.loss(z, ld)
    loss.backward()
    arr = {}
    for k, p in f.named_parameters():
        if p.grad is None:
            continue
        arr[("g8." if p.numel() > 50000 else "g.") + k] = npy(p.grad)[:8] if p.numel() > 50000 else npy(p.grad)
    save("flow_mnist_affine_dag_frozen", z=npy(z), logdet=npy(ld), loss=npy(loss), **arr)


if __name__ == "__main__":
    main()
