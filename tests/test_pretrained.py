"""Reference goldens that need the pretrained 0.3.0 weights (tests/test_model.py:60-119).
Skipped unless a checkpoint directory is provided; the blobs are not available offline."""

from __future__ import annotations

import os

import numpy as np
import pytest

from golden import pretrained_targets as T

CKPT = os.environ.get("CHGNET_CHECKPOINT_DIR", "")
HAVE = os.path.exists(os.path.join(CKPT, "0.3.0", "chgnet_0.3.0_e29f68s314m37.pth.tar")) if CKPT else False


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE, reason="pretrained 0.3.0 checkpoint not available (set CHGNET_CHECKPOINT_DIR)")
def test_limno2_pretrained_goldens():
    from chgnet_amd import Structure
    from chgnet_amd.graph.structure import Lattice
    from chgnet_amd.model import CHGNet

    lat = Lattice.from_parameters(2.868779, 4.634475, 5.832507, 90, 90, 90)
    frac = [[0.5, 0.5, 0.3797505], [0, 0, 0.6202495], [0.5, 0.5, 0.8632525], [0, 0, 0.1367475],
            [0.5, 0, 0.3608245], [0, 0.5, 0.0985135], [0.5, 0, 0.9014865], [0, 0.5, 0.6391755]]
    s = Structure(lat, ["Li", "Li", "Mn", "Mn", "O", "O", "O", "O"], frac)
    model = CHGNet.load(checkpoint_dir=CKPT)
    out = model.predict_structure(s, return_site_energies=True, return_atom_feas=True, return_crystal_feas=True)
    assert out["e"] == pytest.approx(T.LIMNO2_E, rel=1e-4, abs=1e-4)
    assert out["f"] == pytest.approx(np.array(T.LIMNO2_FORCES), rel=1e-3, abs=1e-4)
    assert out["s"] == pytest.approx(np.array(T.LIMNO2_STRESS), rel=5e-3, abs=1e-4)
    assert out["m"] == pytest.approx(T.LIMNO2_MAGMOM, rel=1e-3, abs=1e-4)
    assert out["site_energies"] == pytest.approx(T.LIMNO2_SITE_ENERGIES, rel=1e-4, abs=1e-4)
    assert out["crystal_fea"].mean() == pytest.approx(T.LIMNO2_CRYSTAL_FEA_MEAN, rel=1e-4, abs=1e-4)
    assert out["atom_fea"].mean() == pytest.approx(T.LIMNO2_ATOM_FEA_MEAN, rel=1e-4, abs=1e-4)
