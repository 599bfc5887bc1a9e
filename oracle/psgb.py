"""Reader for the "PSGB1" container written by oracle/ref_dump.c.

TEST INFRASTRUCTURE: used only by the golden-fixture generator and tests.
"""
import struct
import numpy as np

_DT = {ord('f'): np.float32, ord('i'): np.int32, ord('h'): np.int16,
       ord('B'): np.uint8, ord('H'): np.uint16, ord('q'): np.int64,
       ord('d'): np.float64}


def read_psgb(path):
    out = {}
    with open(path, 'rb') as fh:
        buf = fh.read()
    assert buf[:6] == b'PSGB1\n', 'not a PSGB1 file'
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
