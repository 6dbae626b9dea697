"""diff3dhpe_amd -- MI355X (gfx950) engine for Diff3DHPE's DDIM sampling hot path.

Public surface mirrors the reference (csiro-icvg/Diff3DHPE):
    HPE_model(name)(**ctor_kwargs)            common/nets/load_net.py:5-10
    GaussianDiffusion(model=..., **kwargs)    common/conditional_diffusion_*_crossFrames.py:99-112
All tensor math runs in libd3d_hip.so (include/d3d.h); importing this package never imports the CPU oracle.
"""
from .spec import DenoiserConfig, S2S_NAME, S2F_NAME, denoiser_param_spec, param_count  # noqa: F401
from .nets import (HPE_model, ConditionalDiffusionMixSTES2SGRANDLinLift,  # noqa: F401
                   ConditionalDiffusionMixSTES2FGRANDLinLift)
from .diffusion import GaussianDiffusion  # noqa: F401
from ._lib import D3DError, ddim_times, LIB_PATH  # noqa: F401

__all__ = ["HPE_model", "GaussianDiffusion", "ConditionalDiffusionMixSTES2SGRANDLinLift",
           "ConditionalDiffusionMixSTES2FGRANDLinLift", "DenoiserConfig", "D3DError", "ddim_times"]
