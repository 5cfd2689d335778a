"""Placeholder for lvmhelpers/psequence.pyx: the sequence factors are imported by the reference's
sparsemap.py:7 but used by no experiment; they are outside this hot path."""


class PFactorSequenceBinary:
    def __init__(self, *a, **k):
        raise NotImplementedError("sequence factors are not part of the MI355X hot path")
