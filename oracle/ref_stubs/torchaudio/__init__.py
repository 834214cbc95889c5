"""Import-time stub (oracle/make_golden.py only); the hot path never calls torchaudio."""
__version__ = "2.7.1"
