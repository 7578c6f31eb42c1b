"""Stand-in for the absent `pysam` module: the hot path never touches it, the reference only imports it."""


class AlignmentFile(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("pysam shim")
