"""bench.py itself, as the driver starts it (`--steps 20 --warmup 5`), on the GPU box: ONE JSON line on stdout that carries the
contract's keys, BASELINE.json's metric, the roofline and cpu_baseline objects, and numbers that agree with each other."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_driver_form_line_holds_the_contract(gpu_device):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-seconds", "0.3"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, lines                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == "env-steps/s" and d["unit"] == "env-steps/s" and d["metric"] in base["metric"]
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 20, 5)
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "N=65536" in cfg["workload"] and "model" not in cfg and cfg["n_envs_total"] == 65536
    assert cfg["policy_image_handover"] in ("device memory (epoch word)", "graph edge")
    assert cfg["setup_vector_steps"] >= cfg["setup_settle"]["vector_steps"] >= 0
    # value = whole-job env-steps over the timed region's wall clock
    assert d["value"] == pytest.approx(65536 * 20 / (d["ms_per_step"] * 1e-3 * 20), rel=1e-9)
    assert 0.02 < d["ms_per_step"] < 1.0               # (0.09-0.10 ms on an MI355X; a CPU fallback would be 1e4 times that)
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and roof["kernel"] == "k_step"
    assert roof["achieved"] == pytest.approx(roof["alg_bytes_per_env_step"] * 65536 / (roof["kernel_ms"] * 1e-3) / 1e9, rel=1e-9)
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-12) and 0.05 < roof["frac"] < 1.0
    assert roof["kernel_ms"] < d["ms_per_step"]        # the step kernel is one launch of the step
    # HBM bytes per LAUNCH from the committed PMC passes: close to the algorithmic bytes (nothing re-read)
    assert roof["traffic"] is None or 0.9 < roof["traffic"] / (roof["alg_bytes_per_env_step"] * 65536) < 2.0
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["unit"] == "env-steps/s" and cpu["value"] > 0 and cpu["sample"]
    mf = d["roofline_mfma"]
    assert mf["bound"] == "mfma" and 0.0 < mf["frac"] < 1.0


def test_fallback_to_a_graph_edge_when_a_policy_launch_gave_up_in_setup(gpu_device):
    """The policy launch of a captured step waits for its image in device memory, with a time limit.  bench.py looks at the
    give-up word after its warm-up; if it is set, the steps are captured again with a graph edge and the line says so (the word
    is set artificially here: no launch has ever given up on this stack)."""
    env = dict(os.environ, TT_BENCH_TEST_GAVE_UP="1")
    env.pop("TT_POLICY_EDGE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                        "--repeats", "1"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "gave up waiting for the other chain of its step during setup" in r.stderr
    d = json.loads([x for x in r.stdout.splitlines() if x.strip()][-1])
    assert d["config"]["policy_image_handover"].startswith("graph edge (fallback")
    assert 0.02 < d["ms_per_step"] < 1.0 and d["steps"] == 20
