"""Test suite: oracle pinning, host logic (CPU), C-ABI symbols, gloo world-size-2, GPU parity."""
