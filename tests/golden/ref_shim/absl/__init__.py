"""Import-only stand-in for absl (absent in this image), used ONLY by
tests/golden/generate_golden.py to import the reference's NumPy arithmetic."""
