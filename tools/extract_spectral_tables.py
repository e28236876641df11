#!/usr/bin/env python3
"""Extract the tabulated physical DATA the path needs (CIE 1931 observer, D65, glass / metal optical
constants) from the reference's Rust array literals into shimmer_amd/data/spectral_tables.npz.

These are published measurement tables (CIE 018:2019; refractiveindex.info via pbrt-v4), i.e. inputs,
not code; the Rust host would hand the same numbers through ShmSceneDesc. Run in the build container
only (needs /root/reference); the .npz is committed so nothing reads /root/reference at run time.
"""
import re
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/src/spectra")
MIPMAP = Path("/root/reference/src/mipmap.rs")  # MIP_FILTER_LUT: the 128 tabulated EWA filter weights (a data table like the others)
COLOR = Path("/root/reference/src/color.rs")    # SRGB_TO_LINEAR_LUT: the 256 tabulated sRGB decode values (IEC 61966-2-1 EOTF of i / 255)
OUT = Path(__file__).resolve().parents[1] / "shimmer_amd" / "data" / "spectral_tables.npz"
WANT = {
    "cie.rs": ["CIE_LAMBDA", "CIE_X", "CIE_Y", "CIE_Z"],
    "named_spectrum.rs": ["CIE_ILLUM_D6500", "ACES_ILLUM_D60", "GLASS_BK7_ETA_SAMPLES", "GLASS_BAF10_ETA_SAMPLES", "GLASS_F11_ETA_SAMPLES",
                          "CU_ETA_SAMPLES", "CU_K_SAMPLES", "AU_ETA_SAMPLES", "AU_K_SAMPLES", "AG_ETA_SAMPLES",
                          "AG_K_SAMPLES", "AL_ETA_SAMPLES", "AL_K_SAMPLES", "CIE_S_LAMBDA", "CIE_S0", "CIE_S1", "CIE_S2"],
}


def main():
    tables = {}
    for fname, names in WANT.items():
        text = (REF / fname).read_text()
        for name in names:
            m = re.search(r"const\s+" + name + r"\s*:\s*\[Float;\s*([A-Z_0-9a-z]+)\]\s*=\s*\[(.*?)\];", text, re.S)
            if not m:
                sys.exit(f"table {name} not found in {fname}")
            vals = [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", m.group(2))]
            tables[name] = np.asarray(vals, dtype=np.float32)
    text = MIPMAP.read_text()
    m = re.search(r"const\s+MIP_FILTER_LUT\s*:\s*\[Float;\s*MIP_FILTER_LUT_SIZE\]\s*=\s*\[(.*?)\];", text, re.S)
    tables["MIP_FILTER_LUT"] = np.asarray([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()], dtype=np.float32)
    assert tables["MIP_FILTER_LUT"].size == 128
    m = re.search(r"const\s+SRGB_TO_LINEAR_LUT\s*:\s*\[Float;\s*256\]\s*=\s*\[(.*?)\];", COLOR.read_text(), re.S)
    tables["SRGB_TO_LINEAR_LUT"] = np.asarray([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()], dtype=np.float32)
    assert tables["SRGB_TO_LINEAR_LUT"].size == 256 and tables["SRGB_TO_LINEAR_LUT"][255] == 1.0
    assert tables["CIE_X"].size == 471 and tables["CIE_LAMBDA"][0] == 360.0 and tables["CIE_LAMBDA"][-1] == 830.0
    tables["CIE_Y_INTEGRAL"] = np.float32(106.856895)  # spectra/cie.rs:11
    OUT.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT, **tables)
    print("wrote", OUT, {k: v.shape for k, v in tables.items()})


if __name__ == "__main__":
    main()
