"""Reader of the table file ("PSGB1": named arrays) that the reference-side tools write out of a live decoder --
integration/psgpu_export_tables.c (a task's search tables + language model, through the binding's own flattener
integration/psgpu_search_tables.c) and, in the test infrastructure, the golden dumps.  Format: integration/psgpu_table_file.h."""
import struct

import numpy as np

_DT = {ord('f'): np.float32, ord('i'): np.int32, ord('h'): np.int16,
       ord('B'): np.uint8, ord('H'): np.uint16, ord('q'): np.int64,
       ord('d'): np.float64}


def read_psgb(path):
    """dict name -> numpy array"""
    out = {}
    with open(path, 'rb') as fh:
        buf = fh.read()
    if buf[:6] != b'PSGB1\n':
        raise ValueError("%s is not a PSGB1 table file" % path)
    o = 6
    while o < len(buf):
        (nl,) = struct.unpack_from('<I', buf, o); o += 4
        name = buf[o:o + nl].decode(); o += nl
        dt, nd = struct.unpack_from('<II', buf, o); o += 8
        dims = struct.unpack_from('<%dq' % nd, buf, o); o += 8 * nd
        dtype = np.dtype(_DT[dt])
        n = int(np.prod(dims)) if nd else 1
        arr = np.frombuffer(buf, dtype=dtype, count=n, offset=o).reshape(dims).copy()
        o += n * dtype.itemsize
        out[name] = arr
    return out
