"""hoomd.context: initialize() is all the example script calls."""
from pse_amd.context import msg, current_timestep   # noqa: F401


def initialize(args=None):
    """The reference passes '' (no command-line options).  Nothing to do: device selection is torch's."""
    return None
