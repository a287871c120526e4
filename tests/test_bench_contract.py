"""The bench line the driver parses: the committed run of the driver's own command (profiles/r03z_bench_driver_cmd_20steps.json) must
carry every field of the contract -- BASELINE.json's metric, the whole-job value with its step time, `roofline` with the live kernel
time and the PMC traffic, `cpu_baseline` with its sample -- and be internally consistent.  (CPU test: it reads a committed file;
the numbers themselves come from the GPU run.)"""
import json
import os

from conftest import ROOT


def _line():
    text = open(os.path.join(ROOT, "profiles", "r03z_bench_driver_cmd_20steps.json")).read().strip().splitlines()
    assert len(text) == 1, "ONE JSON line on stdout"
    return json.loads(text[0])


def test_contract_fields():
    j = _line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert base["metric"].startswith(j["metric"].split(" (")[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert (j["n_gpus"], j["steps"], j["warmup"]) == (1, 20, 5)          # the driver's command
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 1e3 / j["ms_per_step"]) < 1e-9 * j["value"] + 1e-12      # one block per step per GPU
    assert j["final_proof_verified"] is True


def test_roofline_and_cpu_baseline_blocks():
    j = _line()
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] and r["kernel_ms"] > 0
    alg = (8 * 234 + 32) * (1 << 20)                                     # algorithmic bytes of the live Merkle stage
    assert abs(r["achieved"] - alg / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert 0.9 < r["traffic"] / alg < 1.1                                # every byte read once
    v = r["valu"]
    assert v["unit"] == "T lane-instr/s" and abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-9
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["sample"] and c["value"] > 0
    for name in ("msm", "lde"):
        st = j["stages"][name]
        assert st["roofline"]["traffic"] and "valu" in st["roofline"], name
        assert st["cpu_baseline"]["kind"] == "port"


def test_per_block_telemetry_is_recorded():
    b = _line()["block_i"]
    assert len(b["per_step_s"]) == 20 and len(b["per_step_telemetry"]) == 20
    assert max(b["per_step_s"]) < 1.1 * min(b["per_step_s"]), "no outlier block (the 20 s collector stall of round 2)"
    assert abs(sum(b["per_step_s"]) / 20 - b["seconds_per_block"]) < 0.05
    t = b["per_step_telemetry"][-1]
    assert t["sclk_mhz"] > 1000 and t["busy_pct"] > 90 and t["rss_mb"] < 1.05 * b["rss_mb_before"] + 512
