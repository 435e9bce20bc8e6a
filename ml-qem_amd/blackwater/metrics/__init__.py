"""Reporting metrics of the path (host-side scalars)."""
