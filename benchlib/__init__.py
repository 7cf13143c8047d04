"""Helper modules of bench.py (repo root): engine, the contract layout, alternative layouts, diagnostics, the papers100M-shaped section."""
