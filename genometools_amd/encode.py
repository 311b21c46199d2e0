"""Host-side mirror of the reference's encoder interface for the device encoder.

The reference: ``GtEncseqEncoder`` (src/core/encseq_api.h:223-330,
``gt_encseq_encoder_encode(ee, seqfiles, indexname, err)``) reads the input
files byte by byte on one core.  `DeviceEncoder` hands whole files to the HIP
kernels of csrc/esa_encode.hip (C ABI include/gtamd_encode.h): the symbols stay
in HBM -- `EsaEngine.set_sequence_device` takes them from there -- and the
numbers the reference derives while reading come back as a summary.

FASTQ in its four-line form is encoded on the device too; other FASTQ the device
reader declines (`DeviceDeclined`), and the C host layer's reader takes it
(include/gtamd_host.h).  No CPU implementation here.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import EncodeSummary, EsaError


class DeviceDeclined(EsaError):
    """FASTQ input the device reader does not take (include/gtamd_encode.h): the
    host reader is to read it"""


class DeviceEncoder:
    """encodes (multi-)FASTA files, or four-line FASTQ files, on the device; protein
    selects the alphabet"""

    def __init__(self, protein=False, device=0):
        self._lib = _lib.load()
        self._enc = self._lib.gtamd_encoder_create(device, int(bool(protein)))
        if not self._enc:
            raise EsaError(self._lib.gtamd_esa_last_error().decode())
        self.protein = bool(protein)
        self._buffers = []

    def close(self):
        if self._enc:
            self._lib.gtamd_encoder_destroy(self._enc)
            self._enc = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EsaError(self._lib.gtamd_esa_last_error().decode())

    def encode(self, paths=None, buffers=None):
        """encode files (paths) or in-memory inputs ((name, bytes) pairs)"""
        inputs = []
        for p in paths or []:
            if str(p).endswith((".gz", ".bz2")):      # by name, src/core/file.c:42-53
                import bz2
                import gzip
                opener = gzip.open if str(p).endswith(".gz") else bz2.open
                with opener(p, "rb") as f:
                    inputs.append((p, np.frombuffer(f.read(), dtype=np.uint8)))
            else:
                inputs.append((p, np.fromfile(p, dtype=np.uint8)))
        for name, raw in buffers or []:
            inputs.append((name, np.frombuffer(raw, dtype=np.uint8)
                           if not isinstance(raw, np.ndarray) else raw))
        self._buffers = inputs      # must outlive finish
        for name, raw in inputs:
            self._check(self._lib.gtamd_encoder_add_file(
                self._enc, str(name).encode(), raw.ctypes.data if raw.size else None,
                raw.size))
        if self._lib.gtamd_encoder_finish(self._enc) != 0:
            msg = self._lib.gtamd_esa_last_error().decode()
            raise (DeviceDeclined if self._lib.gtamd_encoder_declined(self._enc) else EsaError)(msg)
        return self

    def fastq_records(self):
        """(file, sequence length, description length) per FASTQ record, as arrays;
        empty for FASTA input.  The file length table of FASTQ input is booked
        from these by the host layer (gtamd_device_encode_files)."""
        k = self._lib.gtamd_encoder_num_fastq_records(self._enc)
        out = [np.empty(k, dtype=np.uint32) for _ in range(3)]
        self._check(self._lib.gtamd_encoder_get_fastq_records(
            self._enc, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, k))
        return tuple(out)

    @property
    def length(self):
        return self._lib.gtamd_encoder_length(self._enc)

    @property
    def device_pointer(self):
        return self._lib.gtamd_encoder_device_symbols(self._enc)

    def symbols(self, first=0, count=None):
        count = self.length - first if count is None else count
        out = np.empty(count, dtype=np.uint8)
        self._check(self._lib.gtamd_encoder_copy_symbols(self._enc, out.ctypes.data,
                                                         first, count))
        return out

    def summary(self):
        s = EncodeSummary()
        self._check(self._lib.gtamd_encoder_get_summary(self._enc, ctypes.byref(s)))
        out = {}
        for name, ctype in s._fields_:
            v = getattr(s, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out

    def file_lengths(self):
        """GtFilelengthvalues per FASTA file"""
        if self._lib.gtamd_encoder_num_fastq_records(self._enc) > 0:
            raise EsaError("the file length table of FASTQ input is booked per buffer fill: "
                           "gtamd_device_encode_files (include/gtamd_host.h) makes it")
        out = []
        for i in range(len(self._buffers)):
            a, b = ctypes.c_uint64(), ctypes.c_uint64()
            self._check(self._lib.gtamd_encoder_file_lengths(self._enc, i, ctypes.byref(a),
                                                             ctypes.byref(b)))
            out.append((a.value, b.value))
        return out

    def descriptions(self):
        """one bytes object per sequence (carriage returns dropped, as the
        reference's reader does, src/core/sequence_buffer_fasta.c:119-120)"""
        k = self._lib.gtamd_encoder_num_descriptions(self._enc)
        file = np.empty(k, dtype=np.uint32)
        start = np.empty(k, dtype=np.uint64)
        end = np.empty(k, dtype=np.uint64)
        self._check(self._lib.gtamd_encoder_get_descriptions(
            self._enc, file.ctypes.data, start.ctypes.data, end.ctypes.data, k))
        return [self._buffers[f][1][int(a):int(b)].tobytes().replace(b"\r", b"")
                for f, a, b in zip(file, start, end)]

    def timing(self):
        t = [ctypes.c_float() for _ in range(3)]
        nbytes = ctypes.c_uint64()
        self._check(self._lib.gtamd_encoder_get_timing(
            self._enc, ctypes.byref(t[0]), ctypes.byref(t[1]), ctypes.byref(t[2]),
            ctypes.byref(nbytes)))
        return {"total_ms": t[0].value, "parse_ms": t[1].value, "stats_ms": t[2].value,
                "input_bytes": nbytes.value}
