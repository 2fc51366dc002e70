"""bench.py's accounting constants against SURVEY.md §8(d) (CPU-only: no kernel is launched)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_macs_match_the_survey():
    b = _bench()
    assert sum(b.GROUP_MACS["inception"]) == 1_144_688     # deterministic Inception forward, MAC per window
    assert sum(b.GROUP_MACS["linear"]) == 191_552          # Linear net (out_size = 2)
    # ELBO step = forward + dX + dW, two contractions for LRT / Flipout: 13.74 MFLOP per MC-sample x window
    assert abs(3 * 2 * 2 * sum(b.GROUP_MACS["inception"]) / 1e6 - 13.74) < 0.01


def test_workloads_are_the_baseline_configs():
    b = _bench()
    w = b.WORKLOADS
    assert w["flipout_conv_s10"]["S"] == 10 and w["flipout_conv_s10"]["net"] == "inception"
    assert w["radial_conv_s20"]["S"] == 20 and w["radial_conv_s20"]["guide"] == "radial"
    assert w["lrt_linear_s1"]["S"] == 1 and w["lrt_linear_s1"]["net"] == "linear"
    assert w["predict_conv_s100"]["S"] == 100 and w["predict_conv_s100"]["B"] == 10000
    assert b.PEAK_TFLOPS["bf16x3"] == 2500.0 and b.N_DATA == 238200
