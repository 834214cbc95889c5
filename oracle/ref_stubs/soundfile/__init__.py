"""Import-time stub (oracle/make_golden.py only); wavs are read with the stdlib instead."""
