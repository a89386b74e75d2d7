"""Documented targets that need the pretrained 0.3.0 checkpoint (absent offline).

Numbers quoted from the reference's own assertions, tests/test_model.py:68-119 (LiMnO2,
examples/mp-18767-LiMnO2.cif, ``CHGNet.load().predict_structure``) with the tolerances used there.
They cannot be executed in a container without the ``.pth.tar`` blob (.MISSING_LARGE_BLOBS);
``tests/test_pretrained.py`` checks them on the GPU when ``$CHGNET_CHECKPOINT_DIR`` points at a
reference ``chgnet/pretrained`` directory and is skipped otherwise.
"""

LIMNO2_E = -7.36769               # eV/atom, rel=1e-4 abs=1e-4
LIMNO2_FORCES = [                  # eV/A, rel=1e-3 abs=1e-4
    [1.34110451e-07, -2.92202458e-08, 2.38135569e-02],
    [5.96046448e-08, 4.63332981e-08, -2.38130391e-02],
    [8.94069672e-08, -2.06753612e-07, 9.25870836e-02],
    [-1.49011612e-07, -1.06170774e-07, -9.25877392e-02],
    [5.96046448e-08, 2.00234354e-08, -2.43449211e-03],
    [-1.19209290e-06, -4.74974513e-08, -1.30698681e-02],
    [1.40070915e-06, 1.64378434e-07, 1.30702555e-02],
    [-5.96046448e-08, 1.66241080e-07, 2.43446976e-03],
]
LIMNO2_STRESS = [                  # GPa, rel=5e-3 abs=1e-4
    [-3.0366361e-01, -3.7709856e-07, 2.2964025e-06],
    [-1.2128221e-06, 2.2305478e-01, -3.2104114e-07],
    [1.3322200e-06, -8.3219516e-07, -1.0736181e-01],
]
LIMNO2_MAGMOM = [3.0495524e-03, 3.0494630e-03, 3.8694179e00, 3.8694181e00,      # mu_B, rel=1e-3 abs=1e-4
                 4.4136152e-02, 3.8622141e-02, 3.8622111e-02, 4.4136211e-02]
LIMNO2_SITE_ENERGIES = [-3.6264274, -3.6264274, -9.634681, -9.634682,            # eV, rel=1e-4 abs=1e-4
                        -8.024935, -8.184724, -8.184724, -8.024935]
LIMNO2_CRYSTAL_FEA_MEAN = 0.26999   # rel=1e-4 abs=1e-4
LIMNO2_ATOM_FEA_MEAN = -0.09668     # rel=1e-4 abs=1e-4
