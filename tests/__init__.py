"""Parity and host-logic tests (CPU: `-m "not gpu"`; MI355X: `-m gpu`)."""
