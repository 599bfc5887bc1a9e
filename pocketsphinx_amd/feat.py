"""Host-side mirror of the reference's whole-utterance dynamic feature
computation (feat_s2mfc2feat_live(begin, end), reference src/feat/feat.c:1310)
for the "1s_c_d_dd" type with batch CMN; arithmetic in csrc/psgpu_feat.hip."""
import ctypes as C

import numpy as np

from . import capi


def dynfeat_1s_c_d_dd(cep, utt_lens):
    """cep [T][cepsize] fp32 (utterances back to back) -> features [T][3*cepsize]."""
    cep = np.ascontiguousarray(cep, np.float32)
    off = np.zeros(len(utt_lens) + 1, np.int32)
    off[1:] = np.cumsum(np.asarray(utt_lens, np.int64))
    if cep.shape[0] != int(off[-1]):
        raise ValueError("cep has %d frames, utt_lens sum to %d" % (cep.shape[0], int(off[-1])))
    out = np.empty((cep.shape[0], 3 * cep.shape[1]), np.float32)
    capi.check(capi.lib().psgpu_feat_1s_c_d_dd(cep.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p),
                                               len(utt_lens), int(cep.shape[1]),
                                               out.ctypes.data_as(C.c_void_p)), "psgpu_feat_1s_c_d_dd")
    return out
