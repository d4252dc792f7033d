"""The generated per-MFMA schedule tables of the two software-pipelined kernels (csrc/attention_pw_sched.inc, csrc/gemm_ws_sched.inc):
the committed files are what the generators produce, every micro-op is placed exactly once, no micro-op shares a gap with (or precedes)
something it depends on, deadlines hold, and the vector-memory order the GEMM kernel's counted waits rely on (stores before the barrier
gap, requests after it) is what the table says.  CPU only: the tables are data."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "glue-factory_amd", "csrc")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(CSRC, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _check(ops, gaps, ng):
    placed = [o for g in gaps for o in g]
    assert len(placed) == len(ops) and len(set(map(id, placed))) == len(ops)
    pos = {id(o): (g, i) for g, lst in enumerate(gaps) for i, o in enumerate(lst)}
    for o in ops:
        g, i = pos[id(o)]
        assert o.earliest <= g <= o.deadline < ng, (o, g, o.earliest, o.deadline)
        for d, dist in o.deps:
            gd, idd = pos[id(d)]
            assert gd <= g - dist, (o, d, dist)
            if dist == 0 and gd == g:
                assert idd < i, (o, d)  # same gap allowed only behind its producer / the reader of what it overwrites


def _table_codes(text, name):
    body = re.search(name + r"\[\d+\]\[W\] = \{(.*?)\n\};", text, re.S).group(1)
    return [[int(x, 16) for x in re.findall(r"0x[0-9a-f]+", line.split("//")[0])] for line in body.strip().splitlines() if "{" in line]


def test_attention_table_is_current_and_consistent(tmp_path):
    gen = _load("gen_attention_pw_sched")
    for cap in range(gen.CAP, 41):
        ops = gen.build_ops()
        gen.propagate_deadlines(ops)
        res = gen.schedule(ops, cap)
        if res:
            break
    gaps, cost = res
    _check(ops, gaps, gen.NG)
    out = tmp_path / "a.inc"
    gen.emit(gaps, cost, str(out))
    assert out.read_text() == open(os.path.join(CSRC, "attention_pw_sched.inc")).read(), "run glue-factory_amd/csrc/gen_attention_pw_sched.py"
    # every soft-max element of both query tiles is subtracted, exponentiated, summed and split exactly once
    kinds = {}
    for g in gaps:
        for o in g:
            kinds[o.kind] = kinds.get(o.kind, 0) + 1
    assert (kinds["SB"], kinds["EX"], kinds["AD"], kinds["CV"], kinds["ML"], kinds["MH"], kinds["VR"], kinds["KR"], kinds["DM"]) == (32, 32, 32, 16, 16, 16, 16, 8, 4)
    codes = _table_codes(out.read_text(), "P1") + _table_codes(out.read_text(), "P2")
    assert len(codes) == 48 and sum(1 for row in codes for c in row if c) == len(ops)


@pytest.mark.parametrize("rot,sc", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_ws_tables_are_consistent(rot, sc):
    gen = _load("gen_gemm_ws_sched")
    for cap in range(20, 60):
        ops = gen.build(rot, sc)
        gen.propagate(ops)
        res = gen.schedule(ops, cap)
        if res:
            break
    gaps, cost = res
    _check(ops, gaps, gen.NG)
    vm = [(g, o.kind) for g, lst in enumerate(gaps) for o in lst if o.kind in ("ST", "DM", "LE")]
    assert all(g < gen.BARRIER_GAP for g, k in vm if k == "ST") and all(g >= gen.BARRIER_GAP for g, k in vm if k != "ST")
    assert sum(k == "ST" for _, k in vm) == 4 and sum(k == "DM" for _, k in vm) == 8 and sum(k == "LE" for _, k in vm) == (4 if rot else 0)
    text = open(os.path.join(CSRC, "gemm_ws_sched.inc")).read()
    codes = _table_codes(text, f"T{rot}{sc}")
    assert len(codes) == 48
    assert [[o.code for o in g] for g in gaps] == [[c for c in row if c] for row in codes], "run glue-factory_amd/csrc/gen_gemm_ws_sched.py"
