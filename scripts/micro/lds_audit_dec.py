"""LDS bank audit of the decoder backward kernels (csrc/fused_dec.hip "backward: LDS layouts"), round-3 layout vs round 4.

    python scripts/micro/lds_audit_dec.py            # table: cycles per wave-instruction (model) / conflict-free cycles
    python scripts/micro/lds_audit_dec.py --search   # the brute-force search that picked swz()

Model: scripts/micro/lds_banks.py (MI355X_MICROARCH.md, section LDS).  Every access pattern of dec_bwd_mlp_kernel and
dec_bwd_attn_kernel is listed with its per-sample count per wave (MT = 7), so the total is comparable with the counters
(SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS of profiles/r03_j_sq_counters_base.txt: 2.5 / 2.2 conflict cycles per LDS instruction;
profiles/r04_*: see DESIGN.md section 4.1).
"""
import itertools
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lds_banks import cycles, lanes  # noqa: E402


def pat(kind, f):
    addrs = [None] * 64
    for l, c16, g in lanes():
        addrs[l] = f(l, c16, g)
    return cycles(kind, addrs)


def swz(row):
    return (((row >> 1) & 3) << 1) ^ (((row >> 3) & 1) * 5)


def old_layout():
    P = 72 * 2      # bytes per image row (D + 8 elements)
    T = 24 * 2      # transposition tile row
    return {
        "row fragment 16 B (MFMA A/B operand)": ("read_b128", lambda l, c, g: c * P + g * 16),
        "weight row fragment 16 B": ("read_b128", lambda l, c, g: c * P + g * 16),
        "tile write 8 B (swapped accumulator)": ("write_b64", lambda l, c, g: c * P + g * 8),
        "transpose read, rows 8g+q4 (dW operands)": ("read_tr64", lambda l, c, g: (8 * g + (c >> 2)) * P + 8 * (c & 3)),
        "transpose read of a weight image": ("read_tr64", lambda l, c, g: (8 * g + (c >> 2)) * P + 8 * (c & 3)),
        "head row read 8 B (attention q/k/v/dO)": ("read_b64", lambda l, c, g: c * P + 16 + 8 * g),
        "head transpose read (K^T, Q^T, dO^T)": ("read_tr64", lambda l, c, g: (4 * g + (c >> 2)) * P + 16 + 8 * (c & 3)),
        "transposition tile write 8 B": ("write_b64", lambda l, c, g: c * T + 8 * g),
        "transposition tile transpose read": ("read_tr64", lambda l, c, g: (4 * g + (c >> 2)) * T + 8 * (c & 3)),
        "wide fill 16 B (LayerNorm prologue)": ("write_b128", lambda l, c, g: (l >> 3) * P + (l & 7) * 16),
    }


def new_layout(f=swz):
    R = 128         # image row: 128 B, chunk c of row r at c ^ f(r)
    W = 80 * 2      # weight image row (D + 16 elements)
    pc = lambda c: 4 * (c >> 3) + (c & 3) + 16 * ((c >> 2) & 1)
    head = 1
    return {
        "row fragment 16 B (MFMA A/B operand)": ("read_b128", lambda l, c, g: c * R + ((g ^ f(c)) << 4)),
        "weight row fragment 16 B": ("read_b128", lambda l, c, g: pc(c) * W + g * 16),
        "tile write 8 B (swapped accumulator)": ("write_b64", lambda l, c, g: c * R + (((2 + (g >> 1)) ^ f(c)) << 4) + (g & 1) * 8),
        "transpose read, rows 8g+q4 (dW operands)": ("read_tr64", lambda l, c, g: (4 * g + (c >> 2)) * R + (((2 + ((c & 3) >> 1)) ^ f(4 * g + (c >> 2))) << 4) + (c & 1) * 8),
        "transpose read of a weight image": ("read_tr64", lambda l, c, g: (4 * g + (c >> 2)) * W + 32 + 8 * (c & 3)),
        "head row read 8 B (attention q/k/v/dO)": ("read_b64", lambda l, c, g: c * R + ((((head + (g >> 1)) & 7) ^ f(c)) << 4) + (g & 1) * 8),
        "head transpose read (K^T, Q^T, dO^T)": ("read_tr64", lambda l, c, g: (4 * g + (c >> 2)) * R + ((((head + ((c & 3) >> 1)) & 7) ^ f(4 * g + (c >> 2))) << 4) + (c & 1) * 8),
        "transposition tile write 8 B": ("write_b64", lambda l, c, g: c * 32 + (((g + (c >> 2)) & 3) << 3)),
        "transposition tile transpose read": ("read_tr64", lambda l, c, g: (4 * g + (c >> 2)) * 32 + ((((c & 3) + g) & 3) << 3)),
        "wide fill 16 B (LayerNorm prologue)": ("write_b128", lambda l, c, g: (l >> 3) * R + (((l & 7) ^ f(l >> 3)) << 4)),
    }


# accesses per wave and sample at MT = 7 (source: fused_dec.hip; the ISA's static counts agree: every loop but the query-tile
# loop and the kk loops of the attention kernel is unrolled): (mlp kernel, attention kernel)
COUNTS = {
    "row fragment 16 B (MFMA A/B operand)": (3 * (2 * 6 + 2 * 2 * 2), 12 + 4 + 12),
    "weight row fragment 16 B": (3 * 2 * 8, 12),
    "tile write 8 B (swapped accumulator)": (36, 12 + 4 + 7 + 14),
    "transpose read, rows 8g+q4 (dW operands)": (3 * 59, 20 + 35),
    "transpose read of a weight image": (3 * 16, 8 + 24),
    "head row read 8 B (attention q/k/v/dO)": (0, 116),
    "head transpose read (K^T, Q^T, dO^T)": (0, 21),
    "transposition tile write 8 B": (0, 98),
    "transposition tile transpose read": (0, 98),
    "wide fill 16 B (LayerNorm prologue)": (4, 6),
}


def table():
    old, new = old_layout(), new_layout()
    print(f"{'access pattern':46s} {'round 3':>10s} {'round 4':>10s} {'free':>5s}   per wave-sample: mlp, attn")
    tot = {"old": [0, 0], "new": [0, 0], "free": [0, 0]}
    for k in old:
        co, base = pat(*old[k])
        cn, _ = pat(*new[k])
        n = COUNTS[k]
        print(f"{k:46s} {co:10d} {cn:10d} {base:5d}   {n[0]:4d} {n[1]:4d}")
        for i in (0, 1):
            tot["old"][i] += co * n[i]; tot["new"][i] += cn * n[i]; tot["free"][i] += base * n[i]
    for i, name in ((0, "dec_bwd_mlp"), (1, "dec_bwd_attn")):
        ninst = sum(v[i] for v in COUNTS.values())
        print(f"{name}: LDS-array cycles per wave and sample  round 3 {tot['old'][i]}  round 4 {tot['new'][i]}  conflict-free {tot['free'][i]};"
              f"  conflict cycles per instruction {(tot['old'][i] - tot['free'][i]) / ninst:.2f} -> {(tot['new'][i] - tot['free'][i]) / ninst:.2f}")


def search():
    """swz restricted to the even rows of a 16-row tile must be a bijection onto the 8 chunks (8-byte head row reads), take 4
    values that differ above bit 0 on rows 0,2,4,6 and on 8..14 (transpose reads of 8 consecutive rows x 2 chunks) and make the
    16-byte row fragment's lane groups {0-3,12-15 | g} u {4-11 | g ^ 1} hit 8 different chunks."""
    rows = [0, 2, 4, 6, 8, 10, 12, 14]
    sol = []
    for perm in itertools.permutations(range(8)):
        f = dict(zip(rows, perm))
        if len({f[r] >> 1 for r in (0, 2, 4, 6)}) < 4 or len({f[r] >> 1 for r in (8, 10, 12, 14)}) < 4:
            continue
        if len({f[0], f[2], f[12], f[14]} | {f[4] ^ 1, f[6] ^ 1, f[8] ^ 1, f[10] ^ 1}) < 8:
            continue
        sol.append(perm)
    lin = [p for p in sol if p[0] == 0 and all(p[a ^ b] == p[a] ^ p[b] for a in range(8) for b in range(8))]
    print(f"{len(sol)} of 40320 bijections qualify; {len(lin)} are XOR-linear in the row bits; chosen: {tuple(swz(r) for r in rows)}")
    assert tuple(swz(r) for r in rows) in lin


if __name__ == "__main__":
    search() if "--search" in sys.argv else table()
