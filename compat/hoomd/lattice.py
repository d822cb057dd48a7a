"""hoomd.lattice: the simple-cubic unit cell the example script builds its system from."""


class unitcell:
    def __init__(self, kind, a):
        self.kind, self.a = kind, float(a)


def sc(a, type_name="A"):
    return unitcell("sc", a)
