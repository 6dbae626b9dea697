"""Drop-in for the reference's common/nets/load_net.py:5-10."""
from diff3dhpe_amd.nets import HPE_model  # noqa: F401
