"""Import-time stub so /root/reference's package imports without hyperpyyaml (oracle/make_golden.py only)."""


def resolve_references(*a, **k):
    raise NotImplementedError("stub")


def load_hyperpyyaml(*a, **k):
    raise NotImplementedError("stub")
