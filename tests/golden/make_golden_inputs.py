"""Helpers shared by the golden generators (inputs only)."""


def read_fasta(path):
    acc, seqs, out = None, [], {}
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            if acc is not None:
                out[acc] = "".join(seqs)
            acc, seqs = line[1:].replace(" ", "_"), []
        elif line:
            seqs.append(line)
    if acc is not None:
        out[acc] = "".join(seqs)
    return out
