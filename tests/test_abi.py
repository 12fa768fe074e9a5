"""The C-ABI library loads and exports every symbol include/vpk.h declares (no compute calls:
this runs without a GPU)."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "vpk.h")).read()
    return sorted(set(re.findall(r"\b(vpk_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from vanishing_points_2017_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        pytest.skip("libvpk.so not built (run python -m vanishing_points_2017_amd.build)")
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 18
    for sym in declared:
        assert hasattr(lib, sym), "libvpk.so does not export %s" % sym
    assert set(_lib.EXPORTS) == set(declared)
    assert lib.vpk_version() == 110


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vanishing_points_2017_amd import _lib, evaluation, synth, vp_localisation
    import numpy as np
    sc = synth.make_scene(1, 20, 3)
    with pytest.raises(_lib.VpkError):
        vp_localisation.expectation_maximisation(sc["l"], sc["lp"], sc["cnn_response"],
                                                 sphere_image=np.zeros((500, 500), np.uint8))
    with pytest.raises(_lib.VpkError):                   # nor is a raster made on the host
        evaluation.get_sphere_image(sc["l"], size=500)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vanishing_points_2017_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), "%s mentions the oracle" % f
                assert "hostsim" not in src, "%s mentions the host simulator" % f
