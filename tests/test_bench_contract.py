"""The bench line the driver parses.  Round 3's line had grown to 20 KB and the driver recorded `parsed: null`; the contract now is ONE
stdout line below 8 KB with the contract keys, `roofline` (+ `valu`), `cpu_baseline` and one short entry per stage, the rest in the
detail file the line names.  CPU tests: (1) `bench.compact_line` over the full report of a committed GPU run, (2) the committed line of
this round's GPU run of the driver's own command, (3) `python bench.py --gpus 2` WITHOUT torch.distributed.run starts its two ranks
itself (`--cpu-only`: gloo, no kernels) and rank 0 prints exactly one line."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _fracs(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            if "frac" in k and isinstance(v, (int, float)):
                yield path + "/" + k, v
            else:
                yield from _fracs(v, path + "/" + k)
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _fracs(v, "%s[%d]" % (path, i))


def _check_line(text):
    assert len(text) < 8192, "the driver did not parse a 20 KB line: keep it below 8 KB (%d)" % len(text)
    j = json.loads(text)
    for k in CONTRACT:
        assert k in j, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert base["metric"].startswith(j["metric"].split(" (")[0])
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - j["n_gpus"] * 1e3 / j["ms_per_step"]) < 1e-9 * j["value"] + 1e-12      # one block per step per GPU
    assert j["final_proof_verified"] is True
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 * r["frac"]
    assert r["traffic"] and r["kernel_ms"] > 0
    alg = (8 * 234 + 32) * (1 << 20)                                     # algorithmic bytes of the live Merkle stage
    assert abs(r["achieved"] - alg / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert 0.9 < r["traffic"] / alg < 1.1                                # every byte read once
    assert r["valu"]["unit"] == "T lane-instr/s"
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["sample"] and c["value"] > 0
    for name, f in _fracs(j):
        assert 0 <= f <= 1.0, "a fraction above 1 is not evidence: %s = %r" % (name, f)
    for stale in ("are not in the C prover",):
        assert stale not in text
    return j


def test_compact_line_of_a_full_report():
    """the full report of round 3's driver-command run (20 KB as one line) through bench.compact_line"""
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r03z_bench_driver_cmd_20steps.json")).read())
    full["roofline"]["valu"] = bench.valu_block(8024440832.0, full["roofline"]["kernel_ms"], "test")
    for st in ("msm", "lde"):
        v = full["stages"][st]["roofline"]["valu"]
        v["frac"] = min(v["frac"], 1.0)          # (round 3's lde block printed 1.046 from a stale numerator)
    full["detail"] = "bench_detail.json"
    text = json.dumps(bench.compact_line(full), separators=(",", ":"))
    j = _check_line(text)
    assert j["stages"]["msm"]["unit"] == "Melem/s" and j["stages"]["lde"]["traffic"] > 0 and "prove_ms" in j["stages"]
    assert len(j["block_i"]["per_step_s"]) == 20 and j["detail"] == "bench_detail.json"


def test_committed_line_of_this_round():
    path = os.path.join(ROOT, "profiles", "r06x_bench_driver_cmd_line.json")
    if not os.path.exists(path):
        pytest.skip("no GPU run of this round committed yet")
    text = open(path).read().strip().splitlines()
    assert len(text) == 1, "ONE JSON line on stdout"
    j = _check_line(text[0])
    b = j["block_i"]
    assert b["blocks_checked"] == b["blocks_timed"] == j["steps"], "every timed block's final proofs are checked"
    # the software pipeline of round 6 fills over the first steps and drains in a short last one: the steps between
    # must not hold an outlier block (the 20 s collector stall of round 2)
    mid = sorted(b["per_step_s"][3:-1])
    assert mid[-1] < 1.35 * mid[len(mid) // 2], "no outlier block"
    # with the read-backs settled the blocks of a run are alike and the chip is never without work
    assert mid[-1] < 1.06 * mid[0] and b["gpu"]["busy_pct"] > 98 and b["host_cores_busy"] < 3.0
    assert j["value"] > 0.21 and 0.8 < j["roofline"]["block"]["frac_of_issue_limit"] < 1
    assert j["stages"]["c5"]["validators"] == 1000 and j["roofline"]["block"]["frac_of_issue_limit"] < 1


def test_bench_starts_its_own_ranks():
    """the driver's multi-GPU command may be plain `python bench.py --gpus N`: bench.py re-executes itself under torch.distributed.run
    (127.0.0.1 rendezvous on a free port); here with --cpu-only (gloo, no kernels): the two ranks meet, exchange the 72-byte MSM
    partial frames, and exactly one line comes out"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-only", "--detail", os.devnull],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2 and "not a measurement" in j["data"]
