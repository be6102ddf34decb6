"""DISTS `auto`'s admission rule (nerf_qa_amd/DISTS_pytorch/DISTS_pt.py::admitted) on the calibration figures the three
pinned stand-in weight sets were MEASURED at on MI355X (max, rms of |mode - f32s| over the 384 calibration pairs;
profiles/r03_*) -- the policy, pinned on CPU: which rung each weight set ends on and why."""
import math

MEASURED = {  # gain: {mode: (max, rms)}
    1.0: {"f16": (6.472e-05, 1.341e-05), "f16w": (2.295e-05, 6.19e-06), "f32m4": (1.436e-05, 4.13e-06),
          "f32m": (8.37e-06, 2.47e-06), "f32m2": (4.26e-06, 1.2e-06)},
    1.3: {"f16": (4.0799e-04, 5.074e-05), "f16w": (2.0755e-04, 2.906e-05), "f32m4": (8.488e-05, 1.639e-05),
          "f32m": (6.154e-05, 9.7e-06), "f32m2": (2.576e-05, 4.09e-06)},
    1.6: {"f16": (6.7049e-04, 1.0153e-04), "f16w": (7.8653e-04, 7.672e-05), "f32m4": (3.7024e-04, 3.945e-05),
          "f32m": (1.5187e-04, 2.119e-05), "f32m2": (7.525e-05, 8.97e-06)},
}


def _choice(gain):
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt import LADDER, admitted
    assert LADDER == ("f16", "f16w", "f32m4", "f32m", "f32m2", "f32s")  # fastest first, f32s always admitted
    for mode in LADDER[:-1]:
        if admitted(*MEASURED[gain][mode]):
            return mode
    return "f32s"


def test_the_three_pinned_weight_sets_end_on_their_rungs():
    assert _choice(1.0) == "f16w"   # plain f16: 6.5e-5 with an outlier-shaped tail (4.8) -> refused
    assert _choice(1.3) == "f32m2"  # 2.6e-5: more than 3x below the bar, admitted whatever the tail (6.3)
    assert _choice(1.6) == "f32s"   # even f32m2 sits at 7.5e-5


def test_rule_edges():
    from nerf_qa_amd.DISTS_pytorch.DISTS_pt import AUTO_F16_BUDGET, AUTO_F16_RMS, AUTO_SAFE_MAX, AUTO_TAIL, admitted
    assert (AUTO_F16_BUDGET, AUTO_F16_RMS, AUTO_SAFE_MAX, AUTO_TAIL) == (6e-5, 2e-5, 3e-5, 4.2)
    assert admitted(3e-5, 3e-6)            # far below the bar: the tail (10) does not matter
    assert not admitted(3.1e-5, 3e-6)      # a little above it with that tail: refused
    assert admitted(5.9e-5, 1.5e-5)        # noise-like (3.9) and under the budget
    assert not admitted(6.1e-5, 1.9e-5)    # over the budget
    assert not admitted(2e-5, 2.1e-5)      # rms over its budget (cannot happen with max < rms, but the rule is the rule)
    assert not admitted(float("nan"), 1e-6) and not admitted(float("inf"), 1e-6) and not admitted(1e-6, float("nan"))
    assert admitted(0.0, 0.0)              # identical scores
    assert not math.isnan(AUTO_TAIL)
