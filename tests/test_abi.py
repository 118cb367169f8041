"""The C-ABI library builds for gfx950, loads, and exports every symbol include/d3hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in os.listdir(os.path.join(ROOT, "include")):
        if h.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", h)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(d3_[a-z0-9_]+)\s*\(", txt))
    return names


def test_every_declared_symbol_is_exported(built_lib):
    lib = ctypes.CDLL(built_lib)
    decl = _declared()
    assert len(decl) >= 20
    missing = [n for n in sorted(decl) if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header(built_lib):
    from d3net_amd import _lib
    assert set(_lib.SIGNATURES) == _declared()
    l = _lib.lib()
    assert l.d3_arch() == b"gfx950"
    assert l.d3_version() >= 100


def test_host_side_constants(built_lib):
    """Pure host entry points (no GPU): the padded ball query's slot size is the reference's 1000-neighbour cap
    (src/bfs_cluster/bfs_cluster.cu:38-44) and the AdamW chunk matches the block map d3net_amd.optim builds."""
    from d3net_amd import _lib
    l = _lib.lib()
    assert l.d3_ballquery_cap() == 1000
    assert l.d3_adamw_chunk() == 4096
    assert l.d3_bfs_cluster_erec_bytes(10) == 160
    assert l.d3_bfs_cluster_ws_bytes(1000) > 17 * 4 * 1000
    # the padded lists' range (round 5): n * cap slots inside start_len's int range -- the library's own bound, not 2 GiB of slots
    from d3net_amd import pointgroup_ops as P
    assert P.ballquery_padded_fits(1) and P.ballquery_padded_fits(852_000) and P.ballquery_padded_fits(2_147_483)
    assert not P.ballquery_padded_fits(2_147_484) and not P.ballquery_padded_fits(0)
    assert P.ballquery_padded_fits(536_870, max_bytes=2 << 30) and not P.ballquery_padded_fits(536_871, max_bytes=2 << 30)


def test_product_never_imports_oracle():
    """d3net_amd/ must not reference oracle/ (no CPU fallback through the checker)."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "d3net_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|libpgoracle|pg_oracle", t, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_switch_table_stays_small(built_lib):
    """the library's measurement / cross-check switches live in ONE table (csrc/tuning.hip): at most 30, every name unique"""
    from d3net_amd import _lib
    l = _lib.lib()
    n = l.d3_tuning_count()
    names = [l.d3_tuning_name(i) for i in range(n)]
    assert n <= 30 and len(set(names)) == n and all(x.startswith(b"D3_") for x in names), names
