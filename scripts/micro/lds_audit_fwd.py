"""LDS bank audit of the two FORWARD attention-half kernels: blk128_fwd_kernel (attn.hip; 3.39 bank-conflict
cycles per LDS instruction at the round-4 counters) and dec_attn_fwd_kernel (fused_dec.hip; 2.59).  For every access pattern of
each kernel: LDS-array cycles per wave-instruction under the layout of rounds 3-4 and under the swizzled unpadded layout the
backward kernels use, against the conflict-free count, weighted by how often the pattern is issued per sample and wave.
Model: scripts/micro/lds_banks.py (MI355X_MICROARCH.md, LDS).   python scripts/micro/lds_audit_fwd.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lds_banks import cycles, lanes  # noqa: E402


def pat(kind, f):
    a = [None] * 64
    for l, c16, g in lanes():
        a[l] = f(l, c16, g)
    return cycles(kind, a)


def bsw(r):          # attn.hip: 256-byte rows
    return ((r & 1) << 1) ^ ((r & 2) << 1) ^ ((r & 4) << 1) ^ (((r >> 3) & 1) * 9)


def swz(r):          # fused_dec.hip: 128-byte rows
    return (((r >> 1) & 3) << 1) ^ (((r >> 3) & 1) * 5)


def report(title, rows):
    print(title)
    print(f"  {'access pattern':62s} {'kind':11s} {'per sample+wave':>15s} {'old':>5s} {'new':>5s} {'free':>5s}")
    tot = [0, 0, 0, 0]
    for name, kind, n, fo, fn in rows:
        co, base = pat(kind, fo)
        cn = pat(kind, fn)[0]
        print(f"  {name:62s} {kind:11s} {n:15.1f} {co:5d} {cn:5d} {base:5d}")
        tot[0] += n; tot[1] += n * co; tot[2] += n * cn; tot[3] += n * base
    print(f"  {'LDS-array cycles per instruction (weighted)':62s} {'':11s} {tot[0]:15.1f} {tot[1] / tot[0]:5.2f} {tot[2] / tot[0]:5.2f} {tot[3] / tot[0]:5.2f}"
          f"   -> conflict cycles per instruction {(tot[1] - tot[3]) / tot[0]:.2f} -> {(tot[2] - tot[3]) / tot[0]:.2f}\n")


# ---------------------------------------------------------------- blk128_fwd_kernel<2, 2>: 8 waves = 8 heads, 64 image rows (2 slots x 2 m-tiles)
H, P = 3, 272          # head 3, old pitch
hc = H * 32
MTT = 4                # m-tiles of a group; counts below are per GROUP (2 samples) and wave
old128 = dict(
    wide=lambda l, c, g: ((l >> 4) + 4) * P + (l & 15) * 16,
    frag=lambda l, c, g: c * P + 64 + g * 16,
    cell=lambda l, c, g: c * P + hc + 8 * g,
    tr=lambda l, c, g: (4 * g + (c >> 2)) * P + hc + 8 * (c & 3),
)
new128 = dict(
    wide=lambda l, c, g: ((l >> 4) + 4) * 256 + (((l & 15) ^ bsw((l >> 4) + 4)) << 4),
    frag=lambda l, c, g: c * 256 + (((4 * 1 + g) ^ bsw(c)) << 4),
    cell=lambda l, c, g: c * 256 + (((2 * H + (g >> 1)) ^ bsw(c)) << 4) + (g & 1) * 8,
    tr=lambda l, c, g: (4 * g + (c >> 2)) * 256 + (((2 * H + ((c & 3) >> 1)) ^ bsw(4 * g + (c >> 2))) << 4) + (c & 1) * 8,
)
rows128 = [
    ("LayerNorm output, 16-byte row pieces (U image)", "write_b128", 2, old128["wide"], new128["wide"]),
    ("row fragment of U: q | k | v products (4 k-steps x 4 m-tiles x 3)", "read_b128", 16, old128["frag"], new128["frag"]),
    ("q | k | v accumulator tiles -> images, 8 B per lane", "write_b64", 12, old128["cell"], new128["cell"]),
    ("head rows of q and k, 8 B per lane (score MFMA operands)", "read_b64", 12, old128["cell"], new128["cell"]),
    ("V^T of a key tile (transposed read)", "read_tr64", 8, old128["tr"], new128["tr"]),
    ("O tile -> image, 8 B per lane", "write_b64", 4, old128["cell"], new128["cell"]),
    ("row stores: 16-byte pieces of O (and q | k | v when saved)", "read_b128", 2, old128["wide"], new128["wide"]),
    ("row fragment of O: projection (4 k-steps x 4 m-tiles)", "read_b128", 16, old128["frag"], new128["frag"]),
]
report("blk128_fwd_kernel<2, 2> (per group of 2 samples and wave; old = 272-byte pitch, new = 256-byte rows, chunk ^ bsw(row))", rows128)

# ---------------------------------------------------------------- dec_attn_fwd_kernel<7>: 4 waves, wave = heads 2w, 2w + 1; 112 image rows of 64 columns
W, PD = 1, 144         # wave 1, old pitch (LU = 72 elements)
oldd = dict(
    wide=lambda l, c, g: ((l >> 3) + 8) * PD + (l & 7) * 16,
    frag=lambda l, c, g: c * PD + 64 + g * 16,
    cell=lambda l, c, g: c * PD + W * 32 + 8 * g,
)
newd = dict(
    wide=lambda l, c, g: ((l >> 3) + 8) * 128 + (((l & 7) ^ swz((l >> 3) + 8)) << 4),
    frag=lambda l, c, g: c * 128 + (((4 + g) ^ swz(c)) << 4),
    cell=lambda l, c, g: c * 128 + (((2 * W + (g >> 1)) ^ swz(c)) << 4) + (g & 1) * 8,
)
rowsd = [
    ("LayerNorm output, 16-byte row pieces (U image)", "write_b128", 3.5, oldd["wide"], newd["wide"]),
    ("row fragment of U: q | k | v products (2 k-steps x 7 m-tiles)", "read_b128", 14, oldd["frag"], newd["frag"]),
    ("q^T parked in the O image, 8 B per lane", "write_b64", 7, oldd["cell"], newd["cell"]),
    ("q^T of a query tile, 8 B per lane (2 heads x 7 tiles)", "read_b64", 14, oldd["cell"], newd["cell"]),
    ("O tile -> image, 8 B per lane (2 heads x 7 tiles)", "write_b64", 14, oldd["cell"], newd["cell"]),
    ("row stores: 16-byte pieces of O", "read_b128", 3.5, oldd["wide"], newd["wide"]),
    ("row fragment of O: projection (2 k-steps x 7 m-tiles)", "read_b128", 14, oldd["frag"], newd["frag"]),
]
report("dec_attn_fwd_kernel<7> (per sample and wave; old = 144-byte pitch, new = 128-byte rows, chunk ^ swz(row) as the backward kernels)", rowsd)
