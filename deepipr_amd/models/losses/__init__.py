"""Hinge sign loss on the passport scale (gamma)."""
