"""Host-side mirror of the reference's ``mp_slam/`` package (mapper only)."""
