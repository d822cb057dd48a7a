"""Minimal simulation context: what the reference's Python layer takes from `hoomd.context` (current system, current
time step, messenger).  It exists so the mirrored UI below has the same call shapes without HOOMD."""
import sys


class _Msg:
    @staticmethod
    def error(text):
        sys.stderr.write("**ERROR**: " + text)

    @staticmethod
    def notice(level, text):
        if level <= 2:
            sys.stdout.write(text)


msg = _Msg()
current = None   # the active System (set by system.System)


def current_timestep():
    return 0 if current is None else current.timestep
