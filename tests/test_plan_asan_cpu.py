"""SURVEY 5 (sanitizers): the host-side planning code of the library (csrc/plan.h: geometry checks, flat parameter layout,
packed-weight layout + descriptor table, workspace carve, weight-gradient launch shape) built with
`g++ -fsanitize=address,undefined` (python -m hsimae_amd.build --asan) and driven through the library's own C entry points
with exactly-sized buffers, for every supported configuration family and for malformed ones.  Runs in a child process with
libasan preloaded; any AddressSanitizer / UBSan report fails the test.  The results are also compared with the shipped
library's (same source, hipcc build)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, json, sys
sys.path.insert(0, '@ROOT@')
from hsimae_amd import _lib, swiglu_hidden
asan = C.CDLL('@ASAN@')
real = _lib.load()
i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
for lib in (asan,):
    lib.hsimae_param_layout.restype, lib.hsimae_param_layout.argtypes = C.c_int, [C.POINTER(_lib.Config), C.POINTER(i64), C.POINTER(i64), C.c_int]
    lib.hsimae_wpk_elems.restype, lib.hsimae_wpk_elems.argtypes = i64, [C.POINTER(_lib.Config)]
    lib.hsimae_pack_table_bytes.restype, lib.hsimae_pack_table_bytes.argtypes = i64, [C.POINTER(_lib.Config)]
    lib.hsimae_build_pack_table.restype, lib.hsimae_build_pack_table.argtypes = C.c_int, [C.POINTER(_lib.Config), vp, vp, vp]
    lib.hsimae_workspace_bytes.restype, lib.hsimae_workspace_bytes.argtypes = i64, [C.POINTER(_lib.Config), i32, i32, i32]
    lib.hsimae_wgrad_msplit.restype, lib.hsimae_wgrad_msplit.argtypes = i32, [i32, i64]

def cfg(bands, D, heads, Dd, dheads, depth=12, s_depth=9, dd=8, prec=0, hidden=None, dhidden=None):
    return _lib.Config(bands=bands, embed_dim=D, depth=depth, s_depth=s_depth, num_heads=heads, dec_dim=Dd, dec_depth=dd,
                       dec_heads=dheads, hidden=swiglu_hidden(D, 4.0) if hidden is None else hidden,
                       dec_hidden=swiglu_hidden(Dd, 4.0) if dhidden is None else dhidden, norm_pix_loss=1, precision=prec)

good = [cfg(48, 128, 8, 64, 8), cfg(96, 128, 8, 64, 8), cfg(96, 256, 16, 64, 8), cfg(192, 512, 32, 64, 8), cfg(192, 512, 32, 64, 8, prec=1),
        cfg(96, 256, 16, 64, 8, prec=1), cfg(32, 32, 2, 32, 4, depth=3, s_depth=2, dd=2), cfg(32, 64, 4, 32, 4, depth=12, s_depth=12, dd=1),
        cfg(32, 64, 4, 32, 4, depth=4, s_depth=0, dd=1), cfg(32, 144, 9, 72, 9, depth=12, s_depth=6, dd=2),
        cfg(32, 64, 4, 48, 6, depth=12, s_depth=6, dd=2)]
bad = [cfg(7, 128, 8, 64, 8), cfg(0, 128, 8, 64, 8), cfg(96, 0, 8, 64, 8), cfg(96, 128, 0, 64, 8), cfg(96, 128, 7, 64, 8),
       cfg(96, 1024, 64, 64, 8), cfg(96, 128, 8, 64, 8, hidden=0), cfg(96, 128, 4, 64, 8), cfg(96, 128, 8, 64, 8, prec=7),
       cfg(96, 132, 33, 64, 8), cfg(96, -128, 8, 64, 8), cfg(4096, 128, 8, 64, 8)]
out = {"good": [], "bad": []}
for c in good:
    n = asan.hsimae_param_layout(C.byref(c), None, None, 0)
    assert n > 0, n
    offs, sizes = (i64 * n)(), (i64 * n)()                       # exactly n entries: an overrun is an ASan report
    assert asan.hsimae_param_layout(C.byref(c), offs, sizes, n) == n
    o2, s2 = (i64 * n)(), (i64 * n)()
    assert real.hsimae_param_layout(C.byref(c), o2, s2, n) == n and list(offs) == list(o2) and list(sizes) == list(s2)
    assert all(offs[i] + sizes[i] == offs[i + 1] for i in range(n - 1)) and offs[0] == 0
    half = (i64 * (n // 2))()
    assert asan.hsimae_param_layout(C.byref(c), half, None, n // 2) == n      # truncated query writes only what fits
    we, tb = asan.hsimae_wpk_elems(C.byref(c)), asan.hsimae_pack_table_bytes(C.byref(c))
    assert we == real.hsimae_wpk_elems(C.byref(c)) and tb == real.hsimae_pack_table_bytes(C.byref(c)) and we > 0 and tb > 0
    table = (C.c_ubyte * tb)()                                   # exactly tb bytes
    fake_p, fake_w = 0x10000000, 0x40000000                     # never dereferenced on the host: pointer arithmetic only
    assert asan.hsimae_build_pack_table(C.byref(c), fake_p, fake_w, table) == 0
    t2 = (C.c_ubyte * tb)()
    assert real.hsimae_build_pack_table(C.byref(c), fake_p, fake_w, t2) == 0 and bytes(table) == bytes(t2)
    descs = (_lib.PackDesc * (tb // C.sizeof(_lib.PackDesc))).from_buffer(table)
    total = offs[n - 1] + sizes[n - 1]
    for d in descs:                                              # every descriptor reads inside the flat parameter buffer ...
        assert fake_p <= d.src and d.src + 4 * d.rows * d.cols <= fake_p + 4 * total
        assert fake_w <= d.dst < fake_w + 2 * we                 # ... and writes inside the packed buffer
    T = c.bands // 8
    for (N, lt, ll) in ((1, 2, 2), (5, T, 9), (64, 2, 7), (4096, min(3, T), 9)):
        wb = asan.hsimae_workspace_bytes(C.byref(c), N, lt, ll)
        assert wb == real.hsimae_workspace_bytes(C.byref(c), N, lt, ll) and wb > 0 and wb % 256 == 0
    assert asan.hsimae_workspace_bytes(C.byref(c), 0, 2, 2) == -1 and asan.hsimae_workspace_bytes(C.byref(c), 4, 0, 2) == -1
    out["good"].append([n, int(total), int(we), int(tb)])
for c in bad:
    r = [asan.hsimae_param_layout(C.byref(c), None, None, 0), asan.hsimae_wpk_elems(C.byref(c)), asan.hsimae_pack_table_bytes(C.byref(c)),
         asan.hsimae_workspace_bytes(C.byref(c), 4, 2, 2), asan.hsimae_build_pack_table(C.byref(c), 1, 1, 1)]
    assert r[0] < 0 and r[1] == -1 and r[2] == -1 and r[3] == -1 and r[4] < 0, r
    out["bad"].append(r[0])
assert asan.hsimae_param_layout(None, None, None, 0) < 0 and asan.hsimae_wpk_elems(None) == -1
for tiles, M in ((13, 110592), (52, 110592), (196, 55296), (1, 64), (1, 1), (600, 10), (0, 0)):
    ms = asan.hsimae_wgrad_msplit(tiles, M)
    assert 1 <= ms <= max(1, (M + 63) // 64) and ms == real.hsimae_wgrad_msplit(tiles, M), (tiles, M, ms)
print("PLAN_ASAN_OK", json.dumps(out))
'''


def test_planning_code_is_clean_under_asan_and_ubsan():
    from hsimae_amd import build as B
    if not os.path.exists(B.LIB):
        pytest.skip("libhsimae_hip.so not built")
    asan = B.build_asan()
    rt = B.asan_runtime()
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD.replace("@ROOT@", ROOT).replace("@ASAN@", asan)], capture_output=True, text=True, timeout=600, env=env)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "PLAN_ASAN_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
