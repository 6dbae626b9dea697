"""Drop-in for the reference module of the same name (seq2frame diffusion wrapper, DIFF-S2F): the engine's class
detects seq2frame from the model it is given."""
from diff3dhpe_amd.diffusion import GaussianDiffusion  # noqa: F401
