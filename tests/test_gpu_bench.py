"""bench.py on one GPU: the one-line JSON contract of the driver and the roofline objects it carries."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                                     # the LAST line of stdout is the JSON line
    assert sum(l.lstrip().startswith("{") for l in lines) == 1
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "patches/s" and d["dtype"] == "bf16" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] in ("TFLOP/s", "GB/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # per-kernel and per-group evidence: the dominant encoder kernels, the decoder group and one fused decoder Block
    for k in ("roofline_kernel", "roofline_kernel_hbm", "roofline_decoder", "roofline_decoder_block", "encoder_mfma_frac", "step_ms"):
        assert k in d, k
    # the line says what library produced it (VERDICT r05 "Next round" 5): the shipped one, built from these sources, no variant switch
    b = d["build"]
    assert b["variant"] == [] and b["variant_bits"] == 0 and b["default_flags"] is True and b["matches_sources"] is True
    blk = d["roofline_decoder_block"]
    assert 50 < blk["fwd_us"] < 1000 and 100 < blk["bwd_us"] < 3000 and 0 < blk["frac"] < 1


@pytest.mark.gpu
def test_bench_data_parallel_line_carries_the_self_check():
    """`bench.py --verify` (on whenever the reducer runs): the 1-rank RCCL path (`--force-ddp`: real all-reduces of every bucket)
    must report a consistent step — the bucketed, backward-overlapped reduction equals one plain all-reduce of the same step's
    local gradients — and list the buckets it issued."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--force-ddp", "--no-extras"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.lstrip().startswith("{")][-1])
    assert d["dp_consistent"] is True, d.get("dp_verify")
    v = d["dp_verify"]
    assert v["ranks"] == 1 and v["matches_single_collective"] is True and len(v["buckets"]) >= 3
    assert sum(hi - lo for lo, hi in v["buckets"]) * 4 == d["comm"]["bytes"]
