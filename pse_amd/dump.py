"""Trajectory output: frames of (timestep, box, positions, images) collected every `period` steps and
written as one NumPy .npz archive -- the role GSD files play for HOOMD, without the dependency."""
import numpy as np


class Trajectory:
    def __init__(self, system, filename, period=1):
        self.system, self.filename, self.period = system, filename, max(1, int(period))
        self.timestep, self.box, self.position, self.image = [], [], [], []
        system.analyzers.append(self)

    def analyze(self, timestep):
        if timestep % self.period:
            return
        s = self.system
        self.timestep.append(int(timestep))
        self.box.append(np.array(s.box, dtype=np.float64))
        self.position.append(s.pos[:, :3].cpu().numpy().copy())
        self.image.append(s.image.cpu().numpy().copy())

    def write(self):
        np.savez(self.filename, timestep=np.array(self.timestep, dtype=np.int64), box=np.array(self.box),
                 position=np.array(self.position), image=np.array(self.image))
        return self.filename


def load(filename):
    """Frames back as a dict of arrays; unwrapped positions = position + image * box lengths (+ xy tilt for y images)."""
    with np.load(filename) as z:
        return {k: z[k] for k in z.files}
