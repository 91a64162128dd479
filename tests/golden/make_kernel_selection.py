"""Writes tests/golden/kernel_selection.tsv: what the kernel-selection table (csrc/dmf_select.hip, through the C-ABI's
dmf_select_describe -- no GPU) answers for a grid of shapes.  tests/test_host.py::test_kernel_selection_table holds the
library to it; after a deliberate change of a rule:

    python tests/golden/make_kernel_selection.py > tests/golden/kernel_selection.tsv   (and read the diff)
"""
import ctypes as C
import itertools
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from demethify_amd import _lib as L  # noqa: E402

S_LIST = (2, 10, 64, 127, 256, 384, 512, 1024, 3000)
TYPES = ((0, 1), (0, 2), (0, 4), (0, 5), (0, 8), (0, 12), (0, 17), (0, 25), (5, 1), (6, 2), (12, 4), (12, 6), (16, 4),
         (17, 3), (28, 4), (40, 24))
F32 = L.DMF_SELECT_COUNTS_F32_EXACT


def grid():
    """(N, S, n_c, n_u, nd, level, n_iter2, flags) rows: levels x count encodings at the CLI's 20 inner steps, then the
    inner-step counts, the flags and the row counts at level 0 with one count digit."""
    shapes = list(itertools.product(S_LIST, TYPES))
    for level, nd in itertools.product(range(5), (0, 1, 2)):
        for S, (n_c, n_u) in shapes:
            yield (1000000, S, n_c, n_u, nd, level, 20, F32)
    for n_iter2 in (0, 1, 50, 51, 500, 1100):
        for S, (n_c, n_u) in shapes:
            yield (1000000, S, n_c, n_u, 1, 0, n_iter2, F32)
    for flags in (0, F32 | L.DMF_SELECT_ALPHA_OUTSIDE_UNIT, F32 | L.DMF_SELECT_PURITY, F32 | L.DMF_SELECT_V_UNALIGNED):
        for S, (n_c, n_u) in shapes:
            yield (1000000, S, n_c, n_u, 1, 0, 20, flags)
    for N in (10, 350):
        for S, (n_c, n_u) in shapes:
            yield (N, S, n_c, n_u, 1, 0, 20, F32)


def describe(lib, row):
    buf = C.create_string_buffer(512)
    st = lib.dmf_select_describe(*row, buf, len(buf))
    return buf.value.decode() if st == L.DMF_OK else f"status {st}"


def main():
    lib = L.load()
    print("# N\tS\tn_c\tn_u\tnd\tlevel\tn_iter2\tflags\tkernels (dmf_select_describe)")
    for row in grid():
        print("\t".join(str(x) for x in row) + "\t" + describe(lib, row))


if __name__ == "__main__":
    main()
