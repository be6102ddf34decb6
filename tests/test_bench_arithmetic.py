"""The roofline arithmetic of bench.py against the figures SURVEY.md section 8(d) states (no GPU involved)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_flops_and_bytes_match_the_survey():
    b = _bench()
    ig256, c1_256 = b.conv_flops_per_image(256, 256)
    ig1080, c1_1080 = b.conv_flops_per_image(1080, 1920)
    # 40.089 GFLOP per 256x256 image and 1269.295 GFLOP per 1080p image over all 13 layers
    assert abs((ig256 + c1_256) / 1e9 - 40.089) < 0.01
    assert abs((ig1080 + c1_1080) / 1e9 - 1269.295) < 0.05
    # L2-pool row of SURVEY 8(d): 39.3 MB (fp32) per 256x256 image, 1244.3 MB per 1080p image
    assert abs(b.pool_bytes_per_image(256, 256, "f32s") / 1e6 - 39.3) < 0.1
    assert abs(b.pool_bytes_per_image(1080, 1920, "f32s") / 1e6 - 1244.3) < 0.5
    assert b.pool_bytes_per_image(1080, 1920, "f16") * 2 == b.pool_bytes_per_image(1080, 1920, "f32")
    # mixed modes: half taps up to the last two-term stage, whose pool writes 4-byte split16 records
    f16, f32 = b.pool_bytes_per_image(1080, 1920, "f16"), b.pool_bytes_per_image(1080, 1920, "f32s")
    assert f16 < b.pool_bytes_per_image(1080, 1920, "f32m") < b.pool_bytes_per_image(1080, 1920, "f32m2") < f32
    # issued f16-MFMA FLOPs: f32s three per product everywhere, f32m two in layers 1..6, f32m2 two in layers 1..3
    alg = b.conv_flops_per_image(1080, 1920)[0]
    assert b.conv_flops_per_image(1080, 1920, "f32s")[0] == 3 * alg and b.conv_flops_per_image(1080, 1920, "f16")[0] == alg
    assert 2 * alg < b.conv_flops_per_image(1080, 1920, "f32m")[0] < b.conv_flops_per_image(1080, 1920, "f32m2")[0] < 3 * alg


def test_workloads_name_the_baseline_configs():
    b = _bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "1080" in base["configs"][2] and b.WORKLOADS["1080p"]["B"] == 8 and b.WORKLOADS["1080p"]["H"] == 1080
    assert b.WORKLOADS["256"]["B"] == 32 and b.WORKLOADS["adists1080p"]["metric"] == "A-DISTS"
    assert {e[0] for e in b.COMPANIONS} <= set(b.WORKLOADS)
    # the shipped default on the two other pinned weight sets and the exact-f32 mode ride on the line
    assert ("1080p", None, "synth:1234:1.3") in b.COMPANIONS and ("1080p", None, "synth:1234:1.6") in b.COMPANIONS
    assert ("1080p", "f32") in b.COMPANIONS


def test_committed_traffic_summary_is_readable():
    b = _bench()
    t = b.load_traffic()
    entries = {k: v for k, v in t.items() if isinstance(v, dict)}
    assert t == {} or (entries and all("conv" in v and "pool" in v for v in entries.values()) and "lib_sha16" in t)
    # the figures are shown only for the library sources they were measured on, and the line always says where from
    shown = b.traffic_for(t, "1080p/f16")
    assert "_source" in shown and (("conv" in shown) == bool(t.get("_lib_matches")))
    assert b.traffic_for(t, "no/such")["_source"].startswith("none")


def test_hbm_roofline_prices_only_the_taps_that_still_go_through_pool_stats():
    b = _bench()
    from nerf_qa_amd import ops
    # pure f16 at 1080p: stage 1 and conv2_2 close their taps themselves; two-term stage 1 stays unfused; f32s all unfused
    assert ops.dists_fused_taps(8, 1080, 1920, "f16") == (1, 2)
    assert ops.dists_fused_taps(8, 1080, 1920, "f32m") == (2,)
    assert ops.dists_fused_taps(8, 1080, 1920, "f32s") == ()
    assert ops.dists_fused_taps(1, 8, 8, "f16") == ()  # (below the fused kernels' smallest tile)
    full = b.pool_bytes_per_image(1080, 1920, "f16")
    rest = b.pool_bytes_per_image(1080, 1920, "f16", (3, 4))
    assert rest < 0.21 * full  # taps 1 and 2 are four fifths of the pass's bytes
    kt = {"conv1_1": (0, 0.0), "conv_igemm": (24, 34.0), "l2pool": (4, 1.2), "stats": (4, 0.3), "adists": (0, 0.0),
          "prep": (0, 0.0), "pool_seam": (4, 0.05)}
    roof, hbm, kms = b.rooflines(kt, 1080, 1920, 8, "f16", {}, fused=(1, 2))
    assert hbm["fused_taps"] == [1, 2] and hbm["launches"] == 4 and hbm["seam_launches"] == 4
    assert abs(hbm["achieved"] - rest * 16 * 2 / 1.2e-3 / 1e9) < 1.0 and hbm["bytes_per_launch_avg"] == round(rest * 16 / 2)
    assert abs(kms["conv_igemm"] - 17.0) < 1e-9 and roof["launches"] == 24


def test_roofline_of_a_step_run_as_two_half_batches():
    """A-DISTS runs a batch as two half-batches on two streams: 24 conv launches per step.  With the step count given the
    FLOPs are those of ONE pyramid pass over the batch per step (inferring the steps from the launches doubled them)."""
    b = _bench()
    kt = {"conv1_1": (0, 0.0), "conv_igemm": (48, 96.0), "l2pool": (16, 9.0), "stats": (8, 0.3), "adists": (40, 36.0),
          "prep": (0, 0.0), "pool_seam": (0, 0.0)}
    roof, hbm, kms = b.rooflines(kt, 1080, 1920, 8, "f32s", {}, fused=(), steps=2)
    flops = b.conv_flops_per_image(1080, 1920)[0] * 16 * 2
    assert abs(roof["achieved"] - flops / 96e-3 / 1e12) < 0.01 and abs(kms["conv_igemm"] - 48.0) < 1e-9
    assert roof["flop_per_launch_avg"] == round(flops / 2 / 24)
    inferred, _, _ = b.rooflines(kt, 1080, 1920, 8, "f32s", {}, fused=())
    assert abs(inferred["achieved"] - 2 * roof["achieved"]) < 0.02  # (what the line said before the count was passed in)
