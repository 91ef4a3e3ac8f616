"""Counterparts of the reference's example drivers (examples/*/fedm-*.py) on the device path."""
