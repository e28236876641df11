"""LayeredBxDF (CoatedDiffuse / CoatedConductor: bxdf.rs:883-1620) against an INDEPENDENT restatement: tests/golden/gen_golden_layered.py re-evaluates
`f`, `sample_f` and `pdf` — the three random walks, the interface BxDFs with their TransportMode / BxDFReflTransFlags arguments, Henyey-Greenstein,
`tr`, the exponential "sample" — in float64 Python written from the Rust text, driven by this repository's defined random stream (the reference seeds
the walks from OS entropy). The shared header shm/bxdf.h (what the oracle AND the GPU kernels compile) must reproduce every vector within the stated
tolerance. Data only: tests/golden/golden_layered.json holds inputs and expected outputs."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

import oracle_py
from oracle_py import fa

GOLD = json.loads((Path(__file__).parent / "golden" / "golden_layered.json").read_text())
REL, ABS = GOLD["tolerance"]["rel"], GOLD["tolerance"]["abs"]


@pytest.fixture(scope="module")
def orc():
    return oracle_py.load()


def close(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.all(np.abs(got - want) <= ABS + REL * np.abs(want))


def test_vector_set_is_what_the_review_asked_for():
    """>= 60 vectors; both materials; smooth and rough interfaces; albedo 0 and > 0; both hemispheres of wo and of wi."""
    fp, sf = GOLD["f_pdf"], GOLD["sample_f"]
    assert len(fp) + len(sf) >= 60 and len(fp) >= 40 and len(sf) >= 30
    for kind in (4, 5):
        rows = [r for r in fp if r["kind"] == kind]
        assert any(r["p"][13] == 0.0 for r in rows) and any(r["p"][13] > 0.0 for r in rows)        # smooth / rough top interface
        assert any(sum(r["p"][8:12]) == 0.0 for r in rows) and any(sum(r["p"][8:12]) > 0.0 for r in rows)  # without / with a medium
        assert any(r["wo"][2] > 0 for r in rows) and any(r["wo"][2] < 0 for r in rows)
        assert any(r["wo"][2] * r["wi"][2] > 0 for r in rows) and any(r["wo"][2] * r["wi"][2] < 0 for r in rows)
    assert sum(1 for r in fp if any(r["f"])) >= 20           # walks that return light
    assert sum(1 for r in sf if r["sample"] is not None) >= 20 and any(r["sample"] is None for r in sf)
    # sample_f: entrance reflections (returned at once) and walks that leave through the top after scattering inside
    assert any(r["sample"] and r["sample"]["pdf_is_proportional"] for r in sf)


@pytest.mark.parametrize("i", range(len(GOLD["f_pdf"])))
def test_layered_f_and_pdf(orc, i):
    r = GOLD["f_pdf"][i]
    out = (C.c_float * 6)()
    orc.orc_fn_layered_f_pdf(r["kind"], fa(*r["p"]), (C.c_int * 2)(r["max_depth"], r["n_samples"]), fa(*r["wo"]), fa(*r["wi"]), out)
    assert close(out[:4], r["f"]), (r["config"], list(out[:4]), r["f"])
    assert close(out[4], r["pdf"]), (r["config"], out[4], r["pdf"])
    assert int(out[5]) == r["flags"]


@pytest.mark.parametrize("i", range(len(GOLD["sample_f"])))
def test_layered_sample_f(orc, i):
    r = GOLD["sample_f"][i]
    out = (C.c_float * 10)()
    ok = orc.orc_fn_layered_sample_f(r["kind"], fa(*r["p"]), (C.c_int * 2)(r["max_depth"], r["n_samples"]), fa(*r["wo"]), float(r["uc"]), fa(*r["u"]), out)
    if r["sample"] is None:
        assert not ok, r["config"]
        return
    assert ok, r["config"]
    s = r["sample"]
    assert close(out[0:4], s["f"]), (r["config"], list(out[0:4]), s["f"])
    assert close(out[4:7], s["wi"]), (r["config"], list(out[4:7]), s["wi"])
    assert close(out[7], s["pdf"]), (r["config"], out[7], s["pdf"])
    assert int(out[8]) == s["flags"] and bool(out[9]) == s["pdf_is_proportional"]
