"""Pins the oracle's logdouble / InvertPath restatement to the REFERENCE's own code.

tests/golden/ref_logdouble.json was produced by oracle/_ref (our driver TU compiled against
/root/reference/logdouble.hpp and utility.h in place). When oracle/_ref is present (build
container, or shipped prebuilt to the GPU box) the live library is checked too."""
import json
import math
import os

import numpy as np
import pytest

import oracle_py as op

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_logdouble.json")))


def h(x):
    return float.fromhex(x) if x != "nan" else math.nan


def same(a, b):
    return (math.isnan(a) and math.isnan(b)) or a == b and math.copysign(1, a) == math.copysign(1, b)


def test_logdouble_binary_ops_bit_exact_vs_reference_golden():
    L = op.lib()
    for row in GOLD["binary"]:
        a, b = h(row["a"]), h(row["b"])
        assert same(L.orc_ld_add(a, b), h(row["add"])), row
        assert same(L.orc_ld_add(a, b), h(row["add_assign"])), row  # operator+= and operator+ agree in the reference
        assert same(L.orc_ld_mul(a, b), h(row["mul"])), row
        if row["div"] != "nan":
            assert same(L.orc_ld_div(a, b), h(row["div"])), row
        assert int(a < b) == row["lt"] and int(a > b) == row["gt"]


def test_logdouble_ctor_and_pow_bit_exact_vs_reference_golden():
    L = op.lib()
    assert h(GOLD["default"]) == -math.inf
    for row in GOLD["ctor"]:
        assert same(L.orc_ld_from_linear(h(row["x"])), h(row["log"])), row
    for row in GOLD["pow"]:
        assert same(L.orc_ld_pow(h(row["a"]), h(row["e"])), h(row["pow"])), row


def test_invert_path_vs_reference_golden():
    L = op.lib()
    for row in GOLD["paths"]:
        w = row["walk"]
        a = np.array(w if w else [0], np.int32)
        out = np.zeros(max(1, len(w)), np.int32)
        L.orc_invert_walk(a, len(w), out)
        assert [int(x) for x in out[:len(w)]] == row["invert"]
        assert row["invert"] == row["reverse"]  # ReversePath == InvertPath (utility.h:28-47)


def test_live_reference_library_if_present():
    R = op.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built here (reference tree absent)")
    L = op.lib()
    rng = np.random.default_rng(0)
    vals = list(rng.uniform(-800, 5, 400)) + [-math.inf, 0.0]
    for a in vals[:60]:
        for b in vals[-60:]:
            assert same(L.orc_ld_add(a, b), R.ref_ld_add(a, b))
            assert same(L.orc_ld_mul(a, b), R.ref_ld_mul(a, b))
    for x in rng.uniform(0, 2, 100):
        assert same(L.orc_ld_from_linear(x), R.ref_ld_from_linear(x))
