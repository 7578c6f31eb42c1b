"""Output writers of the pipeline: mirror of /root/reference/modules/input_output/write_output.py:9-16 (`logger`),
:18-43 (`print_candidates`) and :61-64 (`print_reads`).  Same file formats; the reference's progress prints are omitted."""
from __future__ import annotations

import datetime


def logger(message, logfile, timestamp=True):
    """write_output.py:9-16."""
    if logfile is None:
        return
    if timestamp:
        logfile.write(str(datetime.datetime.now()) + "\t" + message + "\n")
    else:
        logfile.write(message + "\n")


def print_candidates(out_file_name, C, significance_test_values, partition_of_X, X, params, final=False, reads_to_consensus_tsv=""):
    """write_output.py:18-43: candidates longest first (ties in dict order); the final file carries support, p-value,
    partition size and variants in the accession and comes with the read -> candidate table."""
    if final:
        with open(reads_to_consensus_tsv, "w") as tsv:
            for c_acc in partition_of_X:
                for x_acc in partition_of_X[c_acc]:
                    tsv.write("{0}\t{1}\t{2}\t{3}\n".format(x_acc, c_acc, len(X[x_acc]), len(C[c_acc])))
    with open(out_file_name, "w") as out_file:
        for c_acc, seq in sorted(C.items(), key=lambda x: len(x[1]), reverse=True):
            c_acc, t_acc, p_value, correction_factor, support, N_t, delta_size = significance_test_values[c_acc]
            if final:
                out_file.write(">{0}\n{1}\n".format(c_acc + "_" + str(support) + "_" + str(p_value) + "_" + str(N_t) + "_" + str(delta_size), seq))
            else:
                out_file.write(">{0}\n{1}\n".format(c_acc, seq))


def print_reads(remaining_to_align_read_file, remaining_to_align):
    """write_output.py:61-64."""
    with open(remaining_to_align_read_file, "w") as fh:
        for x_acc, seq in remaining_to_align.items():
            fh.write(">{0}\n{1}\n".format(x_acc, seq))
