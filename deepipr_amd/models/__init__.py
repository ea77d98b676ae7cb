"""Model families of the passport hot path (host mirror of the reference's `models` package)."""
