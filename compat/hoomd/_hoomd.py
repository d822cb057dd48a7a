"""Placeholder for HOOMD's C++ core module: the example script imports it and uses nothing from it."""
