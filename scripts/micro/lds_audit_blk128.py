"""LDS bank audit of blk128_fwd_kernel / blk128_bwd_kernel (attn.hip): image pitch FS = 136 elements (272 B), transposition tiles
RS16 = 24 elements (48 B), fp32 tiles at 132 floats.   python scripts/micro/lds_audit_blk128.py [FS] [RS16]
Model: scripts/micro/lds_banks.py.  Prints LDS-array cycles per wave-instruction / conflict-free cycles for every access pattern."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lds_banks import cycles, lanes  # noqa: E402

FS = int(sys.argv[1]) if len(sys.argv) > 1 else 136
RS = int(sys.argv[2]) if len(sys.argv) > 2 else 24
P, T = FS * 2, RS * 2
hc = 3 * 16 * 2          # byte offset of head 3's columns


def pat(kind, f):
    a = [None] * 64
    for l, c16, g in lanes():
        a[l] = f(l, c16, g)
    return cycles(kind, a)


pats = {
    "row fragment 16 B (MFMA operand of the products)": ("read_b128", lambda l, c, g: c * P + 64 + g * 16),
    "head row read 8 B (q, k, v, dO of one head)": ("read_b64", lambda l, c, g: c * P + hc + 8 * g),
    "head row write 8 B (dq / dk / dv / q|k|v / dO in place)": ("write_b64", lambda l, c, g: c * P + hc + 8 * g),
    "head transpose read (K^T, Q^T, dO^T, V^T)": ("read_tr64", lambda l, c, g: (4 * g + (c >> 2)) * P + hc + 8 * (c & 3)),
    "transposition tile write 8 B (P, dS)": ("write_b64", lambda l, c, g: c * T + 8 * g),
    "transposition tile transpose read": ("read_tr64", lambda l, c, g: (4 * g + (c >> 2)) * T + 8 * (c & 3)),
    "wide fill / row store 16 B (48 pieces per row)": ("write_b128", lambda l, c, g: ((l + 64) // 48) * P + ((l + 64) % 48 % 16) * 16),
    "wide read 16 B (row stores)": ("read_b128", lambda l, c, g: ((l + 64) // 48) * P + ((l + 64) % 48 % 16) * 16),
    "fp32 tile write 16 B (du)": ("write_b128", lambda l, c, g: c * 528 + hc * 2 + 16 * g),
    "fp32 tile read 16 B (epilogue, 16 lanes per row)": ("read_b128", lambda l, c, g: (l >> 4) * 528 + (l & 15) * 32),
    "lse / delta read 4 B ([head][row])": ("read_b32", lambda l, c, g: c * 4),
}


def bsw(r):
    return ((r & 1) << 1) ^ ((r & 2) << 1) ^ ((r & 4) << 1) ^ (((r >> 3) & 1) * 9)


H = 3
new = {   # blk128_bwd_kernel since round 4: 256-byte rows, chunk c of row r at c ^ bsw(r); 32-byte transposition tile rows, rotated chunks
    "row fragment 16 B (MFMA operand of the products)": ("read_b128", lambda l, c, g: c * 256 + (((4 * 1 + g) ^ bsw(c)) << 4)),
    "head row read 8 B (q, k, v, dO of one head)": ("read_b64", lambda l, c, g: c * 256 + (((2 * H + (g >> 1)) ^ bsw(c)) << 4) + (g & 1) * 8),
    "head row write 8 B (dq / dk / dv / q|k|v / dO in place)": ("write_b64", lambda l, c, g: c * 256 + (((2 * H + (g >> 1)) ^ bsw(c)) << 4) + (g & 1) * 8),
    "head transpose read (K^T, Q^T, dO^T, V^T)": ("read_tr64", lambda l, c, g: (4 * g + (c >> 2)) * 256 + (((2 * H + ((c & 3) >> 1)) ^ bsw(4 * g + (c >> 2))) << 4) + (c & 1) * 8),
    "transposition tile write 8 B (P, dS)": ("write_b64", lambda l, c, g: c * 32 + (((g + (c >> 2)) & 3) << 3)),
    "transposition tile transpose read": ("read_tr64", lambda l, c, g: (4 * g + (c >> 2)) * 32 + ((((c & 3) + g) & 3) << 3)),
    "wide fill / row store 16 B (48 pieces per row)": ("write_b128", lambda l, c, g: ((l + 64) // 48) * 256 + ((((l + 64) % 48 % 16) ^ bsw((l + 64) // 48)) << 4)),
    "wide read 16 B (row stores)": ("read_b128", lambda l, c, g: ((l + 64) // 48) * 256 + ((((l + 64) % 48 % 16) ^ bsw((l + 64) // 48)) << 4)),
}
print(f"{'access pattern':58s} {'':11s} pitch {P} B / tile {T} B      256-B rows + bsw / 32-B rotated tile      conflict-free")
for k, (kind, f) in pats.items():
    cy, base = pat(kind, f)
    cn = pat(*new[k])[0] if k in new else cy
    print(f"{k:58s} {kind:11s} {cy:8d} {cn:38d} {base:24d}")
