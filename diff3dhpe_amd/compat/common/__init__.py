"""Import shim: lets the UNCHANGED reference runner pick up the MI355X engine.

Put this directory's parent (`diff3dhpe_amd/compat`) on PYTHONPATH.  The reference's own `common/` has no
`__init__.py` (it is a namespace package), so this regular package wins the import of `common`; it then appends every
other `common` directory found on sys.path (the reference's) to its `__path__`, so `common.loss`, `common.arguments`,
`common.camera`, ... still resolve to the reference files, while
    common.nets.load_net.HPE_model
    common.conditional_diffusion_ddim_normal_directPredict_variableLoss_both_crossFrames.GaussianDiffusion
    common.conditional_diffusion_s2f_ddim_normal_directPredict_variableLoss_both_crossFrames.GaussianDiffusion
resolve to the engine's classes (run_conditionalDiffusionDDIM3dhpeNormalDirectPredictVariableLoss.py:115-120, 25).
"""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
for _p in list(sys.path):
    _cand = os.path.join(_p or os.getcwd(), "common")
    if os.path.isdir(_cand) and os.path.abspath(_cand) != _here and _cand not in __path__:
        __path__.append(_cand)
