"""hoomd.group.all()."""
from pse_amd import context as _ctx


def all():   # noqa: A001  (HOOMD's name)
    if _ctx.current is None:
        raise RuntimeError("hoomd.group.all before hoomd.init.create_lattice")
    return _ctx.current.all()
