"""FASTA / FASTQ input: the records the hot path starts from."""
