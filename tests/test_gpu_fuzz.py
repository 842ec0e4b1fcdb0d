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
