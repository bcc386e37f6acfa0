#!/usr/bin/env python3
"""DESIGN.md section 4 ("current kernels"), generated from the committed measurements so it cannot go stale:

    python scripts/kernel_table.py <tag> [--write]      e.g.  python scripts/kernel_table.py r03_f --write

For each model (base / large / huge) it joins
    profiles/<tag>_kernel_stats_<model>_single_stream.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 2` (5 steps)
    profiles/step_traffic_<model>.json                       PMC passes: (2 FETCH_SIZE + WRITE_SIZE) * 1024 per kernel and step
with the algorithmic work of each kernel at that model's shapes (SURVEY.md 8a / Appendix B closed forms, restated below) and
prints one markdown table per model: launches per step, average launch, ms per step, algorithmic GFLOP per launch and the
fraction of the dense bf16 MFMA peak it amounts to, counter bytes per launch against the kernel's COMPULSORY bytes (inputs it
must read + outputs the next kernel must see; saved activations and weight-gradient operands are a design choice and are not
compulsory), and the achieved counter bandwidth.  --write replaces the block between the markers in DESIGN.md."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_TF, STEPS = 2500.0, 5
MODELS = {  # name: (bands, D, hidden, N per GPU)
    "base": (96, 128, 344, 4096), "large": (96, 256, 684, 4096), "huge": (192, 512, 1368, 1024)}


def shapes(model):
    bands, D, h, N = MODELS[model]
    T = bands // 8
    TL, K = T * 9, {12: 27, 24: 54}[T]
    return dict(D=D, h=h, N=N, TL=TL, K=K, Me=N * K, Md=N * TL, Dd=64, hd=172)


def work(name, s):
    """(algorithmic FLOPs, compulsory bytes) per launch of kernel `name`; None where no closed form is kept here."""
    D, h, Me, Md, Dd, hd, TL, K = s["D"], s["h"], s["Me"], s["Md"], s["Dd"], s["hd"], s["TL"], s["K"]
    att_e = 2 * 2 * Me * D * 13           # encoder attention: ~13 keys per query on average over the 9+9+3 blocks (9, 3, 27)
    att_d = 2 * 2 * Md * Dd * TL
    t = {
        "blk128_fwd_kernel": (2 * Me * D * 4 * D + att_e, Me * (4 * D + 4 * D)),                       # x in, x1 out
        "enc_mlp_fwd_kernel<128": (2 * Me * 3 * D * h, Me * 8 * D),
        "enc_mlp_fwd_kernel<256": (2 * Me * 3 * D * h, Me * 8 * D),
        "enc_mlp_fwd_kernel<64": (2 * Md * 3 * Dd * hd, Md * 8 * Dd),
        "enc_mlp_bwd_kernel<128": (2 * 2 * Me * 3 * D * h / 2, Me * 12 * D),                             # dg, du2; x1, dY in, dx1 out
        "enc_mlp_bwd_kernel<256": (2 * 2 * Me * 3 * D * h / 2, Me * 12 * D),
        "enc_mlp_bwd_kernel<64": (2 * 2 * Md * 3 * Dd * hd / 2, Md * 12 * Dd),
        "wgrad_dma_kernel": (2 * Me * (4 * D * D + 3 * D * h), Me * 2 * (3 * D + 3 * D + 3 * h + D)),  # every operand once
        "attn128_bwd_kernel": (2 * Me * D * D + 2 * att_e, Me * (2 * 3 * D + 2 * D + 2 * D + 2 * 3 * D)),
        "lnbwd_dma_kernel": (2 * Me * 3 * D * D, Me * (2 * 3 * D + 4 * D + 4 * D + 4 * D)),
        "dec_attn_fwd_kernel": (2 * Md * 4 * Dd * Dd + att_d, Md * 8 * Dd),
        "dec_block_fwd_kernel": (2 * Md * (4 * Dd * Dd + 3 * Dd * hd) + att_d, Md * 8 * Dd),
        "dec_bwd_mlp_kernel": (2 * 2 * Md * 3 * Dd * hd, Md * 12 * Dd),
        "dec_bwd_attn_kernel": (2 * 2 * Md * 4 * Dd * Dd + 2 * att_d, Md * 12 * Dd),
    }
    for k, v in t.items():
        if name.startswith(k):
            return v
    return None


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)


def table(tag, model):
    st = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_{model}_single_stream.csv")
    if not os.path.exists(st):
        return None
    tr_path = os.path.join(ROOT, "profiles", f"step_traffic_{model if model != 'huge' else 'huge_fp8'}.json")
    traffic = {}
    total_traffic = None
    if os.path.exists(tr_path):
        tj = json.load(open(tr_path))
        traffic = tj.get("kernels_bytes_per_step") or tj.get("top_kernels_bytes_per_step") or {}
        total_traffic = tj["hbm_bytes_per_step"]
    s = shapes(model)
    rows = [r for r in csv.DictReader(open(st)) if not r["Name"].startswith(("void at::", "__amd"))]
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / STEPS / 1e6
    out = [f"**{model}** (N = {s['N']}, encoder rows {s['Me']:,}, decoder rows {s['Md']:,}); kernel time {tot:.2f} ms per step single-stream"
           + (f", counter traffic {total_traffic / 1e9:.1f} GB per step" if total_traffic else "") + f" — `profiles/{tag}_kernel_stats_{model}_single_stream.csv`",
           "",
           "| kernel | launches / step | µs / launch | ms / step | GFLOP / launch (algorithmic) | of bf16 MFMA peak | counter MB / launch | compulsory MB | ratio | counter TB/s |",
           "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        ms = float(r["TotalDurationNs"]) / STEPS / 1e6
        if ms < 0.04:
            continue
        n = short(r["Name"])
        calls = int(r["Calls"]) / STEPS
        us = float(r["AverageNs"]) / 1e3
        w = work(n, s)
        if n.startswith("wgrad_dma_kernel") and calls < 20:      # the decoder_pred / decoder_embed / patch_embed launches: other shapes
            w = None
        key = next((k for k in traffic if k.replace("void ", "") == n), None)
        mb = traffic[key] / calls / 1e6 if key and calls else None
        fl = f"{w[0] / 1e9:.1f}" if w else "—"
        fr = f"{w[0] / (us * 1e-6) / 1e12 / PEAK_TF:.3f}" if w else "—"
        cb = f"{w[1] / 1e6:.0f}" if w else "—"
        ratio = f"{mb / (w[1] / 1e6):.1f}×" if (w and mb) else "—"
        out.append(f"| `{n}` | {calls:g} | {us:.1f} | {ms:.2f} | {fl} | {fr} | {mb:.0f} | {cb} | {ratio} | {mb / us:.2f} |" if mb else
                   f"| `{n}` | {calls:g} | {us:.1f} | {ms:.2f} | {fl} | {fr} | — | {cb} | — | — |")
    return "\n".join(out)


def main():
    tag = sys.argv[1]
    parts = [t for t in (table(tag, m) for m in ("base", "large", "huge")) if t]
    text = "\n\n".join(parts)
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        d = open(path).read()
        a, b = "<!-- kernel-table:begin -->", "<!-- kernel-table:end -->"
        assert a in d and b in d, "markers missing in DESIGN.md"
        d = d[:d.index(a) + len(a)] + f"\n(generated by `python scripts/kernel_table.py {tag} --write`)\n\n" + text + "\n" + d[d.index(b):]
        open(path, "w").write(d)
    else:
        print(text)


if __name__ == "__main__":
    main()
