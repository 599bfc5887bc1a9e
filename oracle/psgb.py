"""Reader for the "PSGB1" container written by oracle/ref_dump.c: the product's own reader (the format is the table file's,
integration/psgpu_table_file.h).

TEST INFRASTRUCTURE: used only by the golden-fixture generator and tests.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pocketsphinx_amd.tablefile import read_psgb  # noqa: E402,F401
