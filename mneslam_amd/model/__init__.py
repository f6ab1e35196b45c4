"""Host-side mirror of the reference's ``model/`` package for the mapping hot path."""
