"""hoomd.PSEv1: the plugin's Python surface (PSEv1/__init__.py of the reference), served by pse_amd."""
from pse_amd import integrate, shear_function, variant   # noqa: F401
