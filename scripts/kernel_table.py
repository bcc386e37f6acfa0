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
        "blk256_fwd_kernel": (2 * Me * D * 4 * D + att_e, Me * (4 * D + 4 * D)),                       # the same half at D = 256 (attn_wide.hip)
        "enc_mlp_fwd_kernel<128": (2 * Me * 3 * D * h, Me * 8 * D),
        "enc_mlp_fwd_kernel<256": (2 * Me * 3 * D * h, Me * 8 * D),
        "enc_mlp_fwd_kernel<64": (2 * Md * 3 * Dd * hd, Md * 8 * Dd),
        "enc_mlp_bwd_kernel<128": (2 * 2 * Me * 3 * D * h / 2, Me * 12 * D),                             # dg, du2; x1, dY in, dx1 out
        "enc_mlp_bwd_kernel<256": (2 * 2 * Me * 3 * D * h / 2, Me * 12 * D),
        "enc_mlp_bwd_kernel<64": (2 * 2 * Md * 3 * Dd * hd / 2, Md * 12 * Dd),
        "wgrad_dma_kernel": (2 * Me * (4 * D * D + 3 * D * h), Me * 2 * (3 * D + 3 * D + 3 * h + D)),  # every operand once
        "blk128_bwd_kernel": (2 * Me * D * 4 * D + 2 * att_e, Me * (4 * D + 4 * D + 4 * D)),          # x, dx1 in, dx out (attention half, backward)
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
    return generic_work(name, s)


def generic_work(name, s):
    """The layer-at-a-time kernels (every linear / attention that is not inside a fused kernel): the shape follows from the
    template arguments and the schedule of csrc/api.hip (block_fwd / block_bwd / decode).  gemm_kernel<AK, EPI, KC, BM, F8, NCH>:
    AK 0 = bf16 A, 1 = fp32 A, 2 = fp32 A + LayerNorm prologue; EPI 0 bf16, 1 fp32, 2 + residual, 3 + pos-embed, 4 SwiGLU gate,
    5 gate backward, 6 LayerNorm backward; KC = 128 marks the decoder-width / embedding launches, NCH > 1 the deep-K (k-outer)
    products.  An instantiation that serves two shapes gets the mean of the two.  Compulsory bytes = operands in + result out
    (saved activations — u, h1|h3 — are this design's choice and are not counted)."""
    D, h, Me, Md, Dd, hd, TL, K = s["D"], s["h"], s["Me"], s["Md"], s["Dd"], s["hd"], s["TL"], s["K"]
    hp, hpd = (h + 31) // 32 * 32, (hd + 31) // 32 * 32
    m = re.match(r"gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        ak, epi, kc, bm, f8, nch = (int(x) for x in m.groups())
        enc = kc != 128 or D == 128                   # encoder-width launch (at Base every launch is 128 wide: decided by role below)
        M, d, hh = (Me, D, h) if (kc != 128) else (Md, Dd, hd)
        lin = lambda M, N, Kd, inb, outb: (2.0 * M * N * Kd, M * (inb + outb))
        if kc == 128 and D == 128:                    # Base: only the embedding / prediction launches run layer at a time
            roles = {(2, 1): [lin(Me, Dd, D, 4 * D, 4 * Dd), lin(Md, 72, Dd, 4 * Dd, 4 * 72)],       # norm + decoder_embed, decoder_norm + decoder_pred
                     (0, 1): [lin(Md, Dd, 72, 2 * 96, 4 * Dd), lin(Me, D, Dd, 2 * Dd, 4 * D)],       # their data gradients
                     (0, 6): [lin(Md, Dd, 72, 2 * 96 + 4 * Dd, 4 * Dd)],                             # d(pred) -> decoder_norm backward
                     (0, 3): [lin(Me, D, 72, 2 * 96, 4 * D)]}                                        # patch embedding (+ pos)
            r = roles.get((ak, epi))
            return (sum(x[0] for x in r) / len(r), sum(x[1] for x in r) / len(r)) if r else None
        table = {
            (2, 0): [lin(M, 3 * d, d, 4 * d, 2 * 3 * d)],                                  # LN1 + q|k|v
            (0, 0): [lin(M, d, d, 2 * d, 2 * d)],                                          # dO = dx1 Wp
            (2, 4): [lin(M, 2 * hh, d, 4 * d, 2 * hh)],                                    # LN2 + W1|W3 + gate
            (1, 5): [lin(M, hh, d, 4 * d + 2 * 2 * hh, 2 * 2 * hh)],                       # gate backward (dg = dY W2^T)
            (2, 1): [lin(Me, Dd, D, 4 * D, 4 * Dd), lin(Md, 72, Dd, 4 * Dd, 4 * 72)],
            (0, 3): [lin(Me, D, 72, 2 * 96, 4 * D)],
        }
        if (ak, epi) == (0, 2):                                                            # + residual: w2 (deep K) or proj
            r = [lin(M, d, hh, 2 * hh + 4 * d, 4 * d)] if (nch > 1 or (kc == 128 and False)) else [lin(M, d, d, 2 * d + 4 * d, 4 * d)]
            if kc == 128:                                                                  # decoder width, 128-deep chunks serve both
                r = [lin(M, d, hh, 2 * hh + 4 * d, 4 * d), lin(M, d, d, 2 * d + 4 * d, 4 * d)]
        elif (ak, epi) == (0, 6):                                                          # LayerNorm backward epilogue: du2 (K = 2 h) and du (K = 3 d)
            r = [lin(M, d, 2 * hh, 2 * 2 * hh + 8 * d, 4 * d), lin(M, d, 3 * d, 2 * 3 * d + 8 * d, 4 * d)]
            if D == 256 and kc != 128:
                r = r[1:]                                                                  # Large: the MLP half is fused, only du + LN1 backward is left
        elif (ak, epi) == (0, 1):                                                          # fp32 result, no LayerNorm epilogue: du2 / du at the decoder width
            r = [lin(M, d, 2 * hh, 2 * 2 * hh, 4 * d), lin(M, d, 3 * d, 2 * 3 * d, 4 * d)] if kc == 128 else [lin(M, d, 3 * d, 2 * 3 * d, 4 * d)]
        else:
            r = table.get((ak, epi))
        return (sum(x[0] for x in r) / len(r), sum(x[1] for x in r) / len(r)) if r else None
    m = re.match(r"attn16_(fwd|bwd)_kernel<(\d+), (\d+), (\d+), (true|false)>", name)
    if m:
        bwd, nt, hdim = m.group(1) == "bwd", int(m.group(2)), int(m.group(3))
        if hdim == 8:                                 # the decoder's attention (8 heads of 8) over all TL tokens
            fl, M, d = 2.0 * 2 * Md * Dd * TL, Md, Dd
        else:                                         # encoder: ~13 keys per query on average over the 9 + 9 + 3 blocks
            fl, M, d = 2.0 * 2 * Me * D * 13, Me, D
        return (2 * fl, M * (2 * 3 * d + 2 * d + 2 * d + 2 * 3 * d)) if bwd else (fl, M * (2 * 3 * d + 2 * d))
    if name.startswith("ln_bwd_kernel"):
        tpr = int(re.search(r"<(\d+)>", name).group(1))
        d, M = tpr * 8, (Md if tpr * 8 == Dd else Me)
        return (0.0, M * d * 12)          # x, dres in, dx out (du is L2-hot from the GEMM in front of it)
    if name.startswith("rows_to_bf16_kernel"):
        return (0.0, Me * D * 6)
    if name.startswith("loss_sample_kernel"):
        return (0.0, s["N"] * (TL * 72 * 4 * 3 + TL * 96 * 2 + TL * 72 * 4))    # cube in, pred in, two images + dpred out
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
        fl = f"{w[0] / 1e9:.1f}" if (w and w[0]) else "—"
        fr = f"{w[0] / (us * 1e-6) / 1e12 / PEAK_TF:.3f}" if (w and w[0]) else "—"
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
