"""ctypes binding of libvsx.so — the C ABI declared in include/vsx.h.

The library is the product: if it is missing, importing this module raises (no PyTorch or CPU
fallback exists on the hot path).  Build it with ``python -m videoswap_amd.build``.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSX_LIB_VARIANT=next: the development build of videoswap_amd/build.py's VARIANTS (kernel candidates that are A/B-ed
# against the measured library before they replace it); unset = the product's libvsx.so
VARIANT = os.environ.get('VSX_LIB_VARIANT') or None
LIB_PATH = os.path.join(_HERE, 'lib', 'libvsx.so' if not VARIANT else f'libvsx_{VARIANT}.so')

VSX_ABI_VERSION = 10


class VsxError(RuntimeError):
    pass


class GemmDesc(Structure):
    """Mirror of ``struct vsx_gemm_desc`` (every field is 8 bytes).  pad_lo / pad_hi default to -1 (symmetric ks/2)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if 'pad_lo' not in kwargs:
            self.pad_lo = -1
        if 'pad_hi' not in kwargs:
            self.pad_hi = -1

    _fields_ = [
        ('M', c_int64), ('N', c_int64), ('K', c_int64),
        ('batch0', c_int64), ('batch1', c_int64),
        ('A', c_void_p), ('A2', c_void_p),
        ('lda', c_int64), ('a_bs0', c_int64), ('a_bs1', c_int64),
        ('a_mode', c_int64), ('H', c_int64), ('W', c_int64), ('C1', c_int64), ('C2', c_int64),
        ('ks', c_int64), ('stride', c_int64), ('upsample', c_int64),
        ('B', c_void_p), ('ldb', c_int64), ('b_bs0', c_int64), ('b_bs1', c_int64),
        ('C', c_void_p), ('ldc', c_int64), ('c_bs0', c_int64), ('c_bs1', c_int64),
        ('c_mode', c_int64), ('c_rows_per_img', c_int64), ('c_img_stride', c_int64),
        ('bias', c_void_p), ('rowvec', c_void_p), ('rows_per_vec', c_int64),
        ('residual', c_void_p), ('ldr', c_int64), ('r_bs0', c_int64), ('r_bs1', c_int64),
        ('geglu', c_int64), ('alpha', c_double),
        ('workspace', c_void_p), ('workspace_bytes', c_int64),
        ('pad_lo', c_int64), ('pad_hi', c_int64),
        ('rowscale', c_void_p), ('colvec', c_void_p),
        ('rowstats', c_void_p), ('rowstats_parts', c_int64),
    ]


# name -> (restype, argtypes); every symbol of include/vsx.h
PROTOTYPES = {
    'vsx_abi_version': (c_int, []),
    'vsx_last_error': (c_char_p, []),
    'vsx_source_digest': (c_char_p, []),
    'vsx_set_option': (c_int, [c_char_p, c_int64]),
    'vsx_gemm_f16': (c_int, [POINTER(GemmDesc), c_void_p]),
    'vsx_gemm_workspace': (c_int64, [POINTER(GemmDesc)]),
    'vsx_gemm_rowstats_parts': (c_int64, [POINTER(GemmDesc)]),
    'vsx_groupnorm_chunks': (c_int64, [c_int64, c_int64]),
    'vsx_groupnorm_stats': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                    c_void_p]),
    'vsx_groupnorm_apply': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                    c_int64, c_int64, c_void_p, c_void_p, c_float, c_int64, c_void_p, c_void_p, c_void_p]),
    'vsx_row_stats': (c_int, [c_void_p, c_int64, c_int64, c_float, c_void_p, c_void_p]),
    'vsx_row_stats_combine': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p, c_void_p]),
    'vsx_layernorm': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int64,
                              c_int64, c_void_p, c_void_p]),
    'vsx_attention_f16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int64] * 14 + [c_float, c_void_p]),
    'vsx_attention_lse_f16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int64] * 15 + [c_float, c_void_p]),
    'vsx_attention_bwd_supported': (c_int64, [c_int64]),
    'vsx_attention_bwd_f16': (c_int, [c_void_p] * 13 + [c_int64] * 9 + [c_float, c_void_p]),
    'vsx_softmax_rows': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    'vsx_softmax_rows_causal': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    'vsx_temporal_attention_f16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int64] * 9
                                   + [c_float, c_void_p]),
    'vsx_silu': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'vsx_quick_gelu': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'vsx_axpy': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    'vsx_pack_latents': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    'vsx_unpack_latents': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    'vsx_cfg_ddim_step': (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_int64,
                                  c_void_p]),
    'vsx_masked_blend': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'vsx_adapter_scatter': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                    c_int64, c_float, c_float, c_void_p]),
    'vsx_prof_enable': (c_int, [c_int64, c_int64]),
    'vsx_comm_unique_id': (c_int, [c_void_p]),
    'vsx_comm_init': (c_int, [c_int64, c_int64, c_void_p]),
    'vsx_comm_init_recording': (c_int, [c_int64, c_int64]),
    'vsx_comm_recorded': (c_int64, [POINTER(c_int64), c_int64]),
    'vsx_comm_size': (c_int64, []),
    'vsx_comm_rank': (c_int64, []),
    'vsx_comm_destroy': (c_int, []),
    'vsx_allgather_kv': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'vsx_allgather_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'vsx_allreduce_gnstats': (c_int, [c_void_p, c_int64, c_void_p]),
    'vsx_prof_pause': (c_int, [c_int64]),
    'vsx_prof_collect': (c_int, [POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
    'vsx_prof_collect_roofline': (c_int, [c_double, c_double, POINTER(c_int64), POINTER(c_double), POINTER(c_double),
                                          POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    # gradient path of the adapter training step (csrc/train.hip)
    'vsx_geglu_fwd': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'vsx_geglu_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'vsx_silu_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'vsx_groupnorm_bwd_workspace': (c_int64, [c_int64, c_int64, c_int64]),
    'vsx_groupnorm_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                  c_void_p, c_float, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'vsx_layernorm_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int64, c_void_p]),
    'vsx_softmax_bwd': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p]),
    'vsx_sum_pool2x2': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    'vsx_adapter_gather': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                   c_float, c_float, c_void_p]),
    'vsx_alltoall_f16': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, POINTER(c_int64), POINTER(c_int64),
                                 c_void_p]),
}

# entry points that only a development variant exports (typed when present); none at the moment
OPTIONAL_PROTOTYPES = {}

_lib = None


def load():
    """Load libvsx.so (once) and type every entry point.  Raises VsxError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VsxError(
            f'{LIB_PATH} not found: the HIP extension is the only implementation of the denoising path; '
            f'build it with `python -m videoswap_amd.build` (hipcc --offload-arch=gfx950).')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    for name, (restype, argtypes) in OPTIONAL_PROTOTYPES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = restype
            fn.argtypes = argtypes
    ver = lib.vsx_abi_version()
    if ver != VSX_ABI_VERSION:
        raise VsxError(f'libvsx ABI version {ver} != expected {VSX_ABI_VERSION}; rebuild the extension')
    # a library built from other sources than the ones next to it (a pull that changed csrc/ but left the old
    # git-ignored .so in place) would silently run stale kernels: refuse it
    csrc = os.path.join(_HERE, 'csrc')
    if os.path.isdir(csrc) and not os.environ.get('VSX_SKIP_DIGEST_CHECK'):
        from .build import source_digest
        have, want = lib.vsx_source_digest().decode(), source_digest(VARIANT)
        if have != want:
            raise VsxError(f'libvsx.so was built from different sources (digest {have[:12]} != {want[:12]}); '
                           f'rebuild it with `python -m videoswap_amd.build` (VSX_SKIP_DIGEST_CHECK=1 overrides)')
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        lib = load()
        msg = lib.vsx_last_error().decode(errors='replace')
        raise VsxError(f'{what} failed with code {rc}: {msg}')
