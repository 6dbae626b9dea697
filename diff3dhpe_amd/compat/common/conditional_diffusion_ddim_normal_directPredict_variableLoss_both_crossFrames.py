"""Drop-in for the reference module of the same name (seq2seq diffusion wrapper, DIFF:99-449)."""
from diff3dhpe_amd.diffusion import GaussianDiffusion  # noqa: F401
