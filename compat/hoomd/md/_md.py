"""Placeholder for HOOMD's md C++ module: imported by the example script, never used."""
