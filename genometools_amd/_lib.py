"""Build and load the HIP engine (libgtamd_esa.so) through its C ABI.

The library is built in-tree (next to this file) so that it travels to the GPU
box with the repository snapshot.  There is no fallback: if it cannot be
loaded, everything that needs it raises.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libgtamd_esa.so")
SOURCES = [os.path.join(HERE, "csrc", f) for f in
           ("esa_prims.hip", "esa_engine.hip", "esa_synth.hip", "esa_encode.hip",
            "esa_pck.hip", "esa_comm.hip")]
HEADERS = [os.path.join(HERE, "csrc", f) for f in ("esa_common.h", "esa_prims.h", "esa_devutil.h", "esa_msd.h",
                                                     "esa_pck_replay.h")] + \
          [os.path.join(ROOT, "include", h) for h in ("gtamd_esa.h", "gtamd_encode.h", "gtamd_pck.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> genometools_amd/libgtamd_esa.so"""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
           "-shared", "-Wall", "-Wno-unused-function", "-o", LIB_PATH] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


class EsaStats(ctypes.Structure):
    _fields_ = [("totallength", ctypes.c_uint64),
                ("numberofallsortedsuffixes", ctypes.c_uint64),
                ("longest", ctypes.c_uint64),
                ("largelcpvalues", ctypes.c_uint64),
                ("maxbranchdepth", ctypes.c_uint64),
                ("lcptabsum", ctypes.c_uint64),
                ("prefixlength", ctypes.c_uint32),
                ("refine_rounds", ctypes.c_uint32),
                ("tied_suffixes", ctypes.c_uint64),
                ("pair_suffixes", ctypes.c_uint64),
                ("device_bytes", ctypes.c_uint64),
                ("msd_big_entries", ctypes.c_uint64),
                ("msd_crowded_entries", ctypes.c_uint64),
                ("rank_entries_built", ctypes.c_uint64)]


class EsaTiming(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_float),
                ("keygen_ms", ctypes.c_float),
                ("sort_ms", ctypes.c_float),
                ("finalize_ms", ctypes.c_float),
                ("refine_ms", ctypes.c_float),
                ("tie_fix_ms", ctypes.c_float),
                ("scatter_ms", ctypes.c_float),
                ("scatter_launches", ctypes.c_uint32),
                ("scatter_items", ctypes.c_uint64),
                ("comm_ms", ctypes.c_float),
                ("comm_calls", ctypes.c_uint32),
                ("comm_bytes", ctypes.c_uint64),
                ("alloc_ms", ctypes.c_float),
                ("dominant_kernel", ctypes.c_uint32),
                ("scatter_read_items", ctypes.c_uint64),
                ("scatter_written_items", ctypes.c_uint64)]


class EncodeSummary(ctypes.Structure):     # gtamd_encode_summary, include/gtamd_encode.h
    _fields_ = [(name, ctypes.c_uint64) for name in
                ("totallength", "numofsequences", "specialcharacters",
                 "realspecialranges")] + \
               [("specialrangestab", ctypes.c_uint64 * 3)] + \
               [(name, ctypes.c_uint64) for name in
                ("lengthofspecialprefix", "lengthofspecialsuffix", "wildcards",
                 "realwildcardranges")] + \
               [("wildcardrangestab", ctypes.c_uint64 * 3)] + \
               [(name, ctypes.c_uint64) for name in
                ("lengthofwildcardprefix", "lengthofwildcardsuffix",
                 "lengthoflongestnonspecial", "minseqlen", "maxseqlen",
                 "equallength")] + \
               [("characterdistribution", ctypes.c_uint64 * 32),
                ("originaldistribution", ctypes.c_uint64 * 256)]


class PckParams(ctypes.Structure):       # gtamd_pck_params, include/gtamd_pck.h
    _fields_ = [("block_size", ctypes.c_uint32), ("bucket_blocks", ctypes.c_uint32),
                ("locate_interval", ctypes.c_uint32), ("feature_toggles", ctypes.c_int32),
                ("with_statistics", ctypes.c_int32)]


class PckInfo(ctypes.Structure):         # gtamd_pck_info
    _fields_ = [(name, ctypes.c_uint64) for name in
                ("file_bytes", "cw_data_pos", "var_data_pos", "range_enc_pos",
                 "num_buckets", "num_regions", "var_bits")] + \
               [("cw_bits", ctypes.c_uint32), ("build_ms", ctypes.c_float)]


# every symbol include/gtamd_esa.h, gtamd_encode.h and gtamd_pck.h declare:
# (restype, argtypes)
_P = ctypes.c_void_p
_U64 = ctypes.c_uint64
_U32 = ctypes.c_uint32
_INT = ctypes.c_int
ABI = {
    "gtamd_device_count": (_INT, []),
    "gtamd_esa_last_error": (ctypes.c_char_p, []),
    "gtamd_recommended_prefixlength": (_U32, [_U32, _U64]),
    "gtamd_abi_selftest": (_INT, [_U64]),
    "gtamd_esa_create": (_P, [_INT, _U64, _U32]),
    "gtamd_esa_destroy": (None, [_P]),
    "gtamd_esa_set_part": (_INT, [_P, _U32, _U32]),
    "gtamd_esa_set_prefixlength": (_INT, [_P, _U32]),
    "gtamd_esa_set_readmode": (_INT, [_P, _INT]),
    "gtamd_esa_set_comm": (_INT, [_P, _P, _P, _P]),
    "gtamd_comm_threads_create": (_P, [_U32]),
    "gtamd_comm_rccl_unique_id": (_INT, [_P]),
    "gtamd_comm_rccl_create": (_P, [_P, _U32, _U32, _INT]),
    "gtamd_comm_attach": (_INT, [_P, _U32, _P, _INT]),
    "gtamd_comm_abort": (None, [_P]),
    "gtamd_esa_set_comm_abort": (_INT, [_P, _P, _P]),
    "gtamd_comm_destroy": (None, [_P]),
    "gtamd_esa_set_sequence_bytes": (_INT, [_P, _P, _U64, _INT]),
    "gtamd_esa_set_sequence_packed": (_INT, [_P, _P, _P, _U64]),
    "gtamd_esa_run": (_INT, [_P, _U32]),
    "gtamd_esa_table_entries": (_U64, [_P, _INT]),
    "gtamd_esa_table_offset": (_U64, [_P]),
    "gtamd_esa_bck_layout": (_INT, [_P, ctypes.POINTER(_U64), ctypes.POINTER(_U64),
                                    ctypes.POINTER(_U64)]),
    "gtamd_esa_table_device": (_P, [_P, _INT]),
    "gtamd_esa_table_copy": (_INT, [_P, _INT, _P, _U64, _U64]),
    "gtamd_esa_get_stats": (_INT, [_P, ctypes.POINTER(EsaStats)]),
    "gtamd_esa_get_timing": (_INT, [_P, ctypes.POINTER(EsaTiming)]),
    "gtamd_esa_build": (_INT, [_P, _U64, _U32, _U32, _P, _P, _P, _P, _U64,
                               ctypes.POINTER(_U64), ctypes.POINTER(EsaStats)]),
    "gtamd_synth_bytes": (_INT, [_INT, _INT, _U64, _U64, _P]),
    # include/gtamd_encode.h
    "gtamd_encoder_create": (_P, [_INT, _INT]),
    "gtamd_encoder_create_map": (_P, [_INT, _P, _U32, ctypes.c_uint]),
    "gtamd_encoder_destroy": (None, [_P]),
    "gtamd_encoder_add_file": (_INT, [_P, ctypes.c_char_p, _P, _U64]),
    "gtamd_encoder_finish": (_INT, [_P]),
    "gtamd_encoder_declined": (_INT, [_P]),
    "gtamd_encoder_num_fastq_records": (_U64, [_P]),
    "gtamd_encoder_get_fastq_records": (_INT, [_P, _P, _P, _P, _U64]),
    "gtamd_encoder_set_symbols": (_INT, [_P, _P, _U64]),
    "gtamd_encoder_length": (_U64, [_P]),
    "gtamd_encoder_device_symbols": (_P, [_P]),
    "gtamd_encoder_copy_symbols": (_INT, [_P, _P, _U64, _U64]),
    "gtamd_encoder_get_summary": (_INT, [_P, ctypes.POINTER(EncodeSummary)]),
    "gtamd_encoder_file_lengths": (_INT, [_P, ctypes.c_size_t, ctypes.POINTER(_U64),
                                          ctypes.POINTER(_U64)]),
    "gtamd_encoder_num_descriptions": (_U64, [_P]),
    "gtamd_encoder_get_descriptions": (_INT, [_P, _P, _P, _P, _U64]),
    "gtamd_encoder_pack_twobit": (_INT, [_P, _INT, ctypes.c_uint, _P, _U64]),
    "gtamd_encoder_pack_specialbits": (_INT, [_P, _P, _U64]),
    "gtamd_encoder_pack_bytecompress": (_INT, [_P, _P, _U64]),
    "gtamd_encoder_get_wildcard_runs": (_INT, [_P, _P, _P, _U64]),
    "gtamd_encoder_get_separators": (_INT, [_P, _P, _U64]),
    "gtamd_encoder_get_timing": (_INT, [_P, ctypes.POINTER(ctypes.c_float),
                                        ctypes.POINTER(ctypes.c_float),
                                        ctypes.POINTER(ctypes.c_float),
                                        ctypes.POINTER(_U64)]),
    # include/gtamd_pck.h
    "gtamd_pck_default_toggles": (_INT, [_U32, _U32, _U32, _INT]),
    "gtamd_pck_create": (_P, [_INT]),
    "gtamd_pck_destroy": (None, [_P]),
    "gtamd_pck_build": (_INT, [_P, _P, _P, _U64, _U32, _U64, ctypes.POINTER(PckParams)]),
    "gtamd_pck_build_host": (_INT, [_P, _P, _P, _U64, _U32, _U64, ctypes.POINTER(PckParams)]),
    "gtamd_pck_build_from_esa": (_INT, [_P, _P, ctypes.POINTER(PckParams)]),
    "gtamd_pck_get_info": (_INT, [_P, ctypes.POINTER(PckInfo)]),
    "gtamd_pck_image_device": (_P, [_P]),
    "gtamd_pck_image_copy": (_INT, [_P, _P, _U64, _U64]),
    "gtamd_pck_ctxmap_build": (_INT, [_P, _P, _U64, _INT, ctypes.POINTER(_INT)]),
    "gtamd_pck_ctxmap_build_from_esa": (_INT, [_P, _P, _INT, ctypes.POINTER(_INT)]),
    "gtamd_pck_ctxmap_build_host": (_INT, [_P, _P, _U64, _INT, ctypes.POINTER(_INT)]),
    "gtamd_pck_ctxmap_bytes": (_U64, [_P]),
    "gtamd_pck_ctxmap_copy": (_INT, [_P, _P, _U64, _U64]),
}

_lib = None


def load():
    """ctypes handle with typed entry points; raises if the library is absent"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (needs hipcc); there is no CPU fallback" % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP/HSA runtime next to the system one
        # this library links.  Both work in one process only when torch's
        # copy is mapped first (measured on the MI355X box: the other order
        # leaves torch with "No HIP GPUs are available"), so a Python process
        # that may use torch for device buffers or torch.distributed gets
        # torch imported here, before the library.  C callers are unaffected.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class EsaError(RuntimeError):
    """mirrors a GtError message: 'gt suffixerator: error: <message>'"""


def check(rc):
    if rc != 0:
        raise EsaError(load().gtamd_esa_last_error().decode())
