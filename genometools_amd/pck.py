"""Host-side mirror of the reference's packed-index construction over the C ABI.

`gt packedindex trsuftab INDEX` (src/tools/gt_packedindex_trsuftab.c:44-79)
builds INDEX.bdx from the tables of a suffix-array project through
gt_trSuftab2BWTSeq (src/match/eis-bwtseq-construct.c:64-92).  `PackedIndex`
does the same from tables resident on the device (include/gtamd_pck.h); the
option names are the tool's (-bsize, -blbuck, -locfreq, -locbitmap).

Everything here goes through genometools_amd/libgtamd_esa.so (HIP); there is
no CPU implementation in this package.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import PckInfo, PckParams, check

LOCATE_BITMAP, LOCATE_COUNT, REVERSIBLY_SORTED = 1, 2, 4


def default_toggles(bsize=8, blbuck=8, locfreq=16, locbitmap=None, sprank=False):
    """the feature toggles gt_computePackedIndexDefaults derives
    (src/match/eis-bwtseq-param.c:89-103); locbitmap None = option not given"""
    return _lib.load().gtamd_pck_default_toggles(
        bsize, blbuck, locfreq, -1 if locbitmap is None else int(bool(locbitmap))) | \
        (REVERSIBLY_SORTED if sprank else 0)


class PackedIndex:
    """builder of INDEX.bdx images on one device"""

    def __init__(self, device=0):
        self._lib = _lib.load()
        self._p = self._lib.gtamd_pck_create(device)
        if not self._p:
            raise _lib.EsaError(self._lib.gtamd_esa_last_error().decode())

    def close(self):
        if self._p:
            self._lib.gtamd_pck_destroy(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _params(bsize, blbuck, locfreq, locbitmap, mkindex, sprank):
        return PckParams(bsize, blbuck, locfreq,
                         default_toggles(bsize, blbuck, locfreq, locbitmap, sprank),
                         int(bool(mkindex)))

    def build_from_esa(self, engine, bsize=8, blbuck=8, locfreq=16, locbitmap=None,
                       mkindex=False, sprank=False):
        """from an EsaEngine whose last run produced .suf and .bwt; mkindex: the
        file of `gt packedindex mkindex` (with sequence statistics) instead of
        the one of `gt packedindex trsuftab`"""
        pp = self._params(bsize, blbuck, locfreq, locbitmap, mkindex, sprank)
        check(self._lib.gtamd_pck_build_from_esa(self._p, engine._ctx, ctypes.byref(pp)))

    def build(self, bwt_ptr, suf_ptr, total_len, numofchars, longest, bsize=8, blbuck=8,
              locfreq=16, locbitmap=None, mkindex=False, sprank=False):
        """from raw device pointers of the .bwt and .suf tables"""
        pp = self._params(bsize, blbuck, locfreq, locbitmap, mkindex, sprank)
        check(self._lib.gtamd_pck_build(self._p, bwt_ptr, suf_ptr, total_len, numofchars,
                                        longest, ctypes.byref(pp)))

    def context_map_from_esa(self, engine, ilog=-1):
        """INDEX.<ilog>cxm (-ctxilog; -1: the automatic interval) from the engine's
        suffix array: (interval log used, bytes of the file)"""
        used = ctypes.c_int()
        check(self._lib.gtamd_pck_ctxmap_build_from_esa(self._p, engine._ctx, ilog,
                                                        ctypes.byref(used)))
        n = self._lib.gtamd_pck_ctxmap_bytes(self._p)
        out = np.empty(n, dtype=np.uint8)
        check(self._lib.gtamd_pck_ctxmap_copy(self._p, out.ctypes.data_as(ctypes.c_void_p), 0, n))
        return used.value, out

    def info(self):
        inf = PckInfo()
        check(self._lib.gtamd_pck_get_info(self._p, ctypes.byref(inf)))
        return {name: getattr(inf, name) for name, _ in inf._fields_}

    def device_pointer(self):
        return self._lib.gtamd_pck_image_device(self._p)

    def image(self, offset=0, count=None):
        """bytes [offset, offset + count) of INDEX.bdx"""
        total = self.info()["file_bytes"]
        count = total - offset if count is None else count
        out = np.empty(count, dtype=np.uint8)
        if count:
            check(self._lib.gtamd_pck_image_copy(self._p, out.ctypes.data_as(ctypes.c_void_p),
                                                 offset, count))
        return out
