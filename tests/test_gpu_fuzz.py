"""A short fixed-seed walk over random Monotonic-normalizer shapes (tests/fuzz_mono.py): forward, every gradient, the inverse
and its scattered form against the oracle, knife-edge elements arbitrated in fp64.  The parametrised tests pin the shapes the
reference uses; this one walks between them (hidden widths 1..208, 1-4 layers, c 1..40, S 1..40) -- the weight pack's unit order
and the kernels' launch conditions are functions of exactly these numbers."""
import pytest

pytestmark = pytest.mark.gpu


def test_monotonic_random_shapes():
    import fuzz_mono
    res = fuzz_mono.walk(30, 7)
    bad = [(case, what, why) for case, what, why in res if why]
    assert not bad, bad


def test_gemm_random_shapes():
    """gnf_gemm against an fp64 product over random M, N, K, operand orders, paddings and epilogues (tests/fuzz_gemm.py): the
    dispatcher chooses among six kernel families from exactly these numbers; the error bound is 2e-6 of the summed term
    magnitudes per entry"""
    import fuzz_gemm
    res = fuzz_gemm.walk(80, 2)
    bad = [(case, desc, worst) for case, desc, worst, b in res if b]
    assert not bad, bad
    assert len({desc.split()[-1] for _, desc, _, _ in res}) >= 3          # (more than one kernel family was exercised)


def test_linear_random_chains():
    """Linear + ReLU chains of random depth / widths / batch sizes / mask kinds through the gnf_linear_* entry points against
    an fp64 autograd (tests/fuzz_linear.py); rows on a ReLU knife edge get a zero cotangent"""
    import fuzz_linear
    res = fuzz_linear.walk(60, 3)
    bad = [(case, desc, why) for case, desc, errs, why in res if why]
    assert not bad, bad


def test_sparse_front_random_rows():
    """random row subsets (one crop origin, a level diagonal, anything), batch sizes and window gates through the sparse
    front -- both crop-kernel forms, the grouped GEMM pair and the one-launch fc kernel -- against the oracle's dense
    MNISTCNN on the masked copies, element-wise at atol 1e-6 + rtol 1e-5 (tests/fuzz_sparse.py)"""
    import fuzz_sparse
    res = fuzz_sparse.walk(40, 5)
    bad = [(case, desc, w) for case, desc, w, b in res if b]
    assert not bad, bad


def test_random_flows_logdet_equals_jacobian_and_invert_round_trips():
    """flows of the reference's factory at random sizes (Coupling / MADE / DAG x Affine / Monotonic, 1-3 steps): the log-det the
    flow returns equals log|det| of the Jacobian assembled from d backward passes of the same flow, invert(forward(x)) == x, the
    loss is finite and every parameter receives a finite gradient (tests/fuzz_flow.py)"""
    import fuzz_flow
    res = fuzz_flow.walk(40, 1)
    bad = [(case, desc, why) for case, desc, e1, e2, why in res if why]
    assert not bad, bad


def test_dag_gate_random_modes_and_sizes():
    """every importance form x gate form of the DAG gate with injected noise, dimensions that are no multiple of 4, one-hot
    columns, sparse and dense A, against the oracle's masked inputs and their autograd (tests/fuzz_gate.py)"""
    import fuzz_gate
    res = fuzz_gate.walk(80, 1)
    bad = [(case, desc, why) for case, desc, errs, why in res if why]
    assert not bad, bad


def test_rowwise_kernels_random_shapes():
    """the Affine normalizer (both h layouts, values beyond the clamps), the log-sum of Jacobian rows, the standard-normal
    log-density and the one-launch loss at random (B, d) incl. odd widths and 4 099 rows, against the oracle / fp64 torch
    (tests/fuzz_rowwise.py)"""
    import fuzz_rowwise
    res = fuzz_rowwise.walk(60, 1)
    bad = [(case, desc, why) for case, desc, errs, why in res if why]
    assert not bad, bad


def test_sparse_front_training_gradients_random_rows():
    """the training path of the sparse front (argmax record, grouped fc1 GEMMs, two-wavefront crop backward, closed-form
    background terms) against the dense kernels' gradients on random row subsets / batch sizes; copies with a pool window or
    a ReLU (conv1, fc1) on the knife edge get a zero cotangent when a first comparison fails (tests/fuzz_sparse_grad.py)"""
    import fuzz_sparse_grad
    res = fuzz_sparse_grad.walk(20, 1)
    bad = [(case, desc, why) for case, desc, errs, why in res if why]
    assert not bad, bad
