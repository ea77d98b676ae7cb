"""ctypes binding of the C-ABI kernel library (include/deepipr_hip.h).

The library is the product: there is no CPU fallback.  If csrc/libdeepipr_hip.so has not been built
(`make -C deepipr_amd/csrc`, or `python -c "import __graft_entry__ as g; g.build()"`), every passport
op raises HipLibraryMissing.
"""
import ctypes
import os

# torch must be imported (and with it torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) BEFORE our
# library is dlopen'ed: the dynamic linker then binds our NEEDED libamdhip64.so.7 to the runtime torch
# already uses.  The other order loads /opt/rocm's copy as a second HIP runtime in the process, which
# cannot see torch's device context ("no ROCm-capable device is detected").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEEPIPR_LIB: measurement builds only (csrc/libdeepipr_hip_trace.so, `make -C deepipr_amd/csrc trace`)
LIB_PATH = os.environ.get('DEEPIPR_LIB') or os.path.join(_HERE, 'csrc', 'libdeepipr_hip.so')

_c = ctypes
_f32p, _f64p, _i8p, _vp = _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p   # raw device addresses
_int, _flt, _sz = _c.c_int, _c.c_float, _c.c_size_t

# name -> (restype, argtypes); mirrors include/deepipr_hip.h one to one
SIGNATURES = {
    'deepipr_abi_version': (_int, []),
    'deepipr_last_error': (_c.c_char_p, []),
    'deepipr_event_create': (_int, [_c.POINTER(_c.c_void_p)]),
    'deepipr_event_destroy': (_int, [_vp]),
    'deepipr_event_record': (_int, [_vp, _vp]),
    'deepipr_stream_wait_event': (_int, [_vp, _vp]),
    'deepipr_event_synchronize': (_int, [_vp]),
    'deepipr_profile_enable': (_int, [_int]),
    'deepipr_profile_read': (_int, [_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_longlong)]),
    'deepipr_pooled_patch_mean': (_int, [_f32p, _int, _int, _int, _int, _int, _int, _int, _int, _int, _f64p, _vp]),
    'deepipr_gamma_beta_fwd': (_int, [_f32p, _f64p, _int, _int, _f32p, _f32p, _vp]),
    'deepipr_gamma_beta_bwd': (_int, [_f32p, _f32p, _f64p, _int, _int, _f32p, _vp]),
    'deepipr_gamma_beta_bwd_acc': (_int, [_f32p, _f32p, _f64p, _int, _int, _f32p, _vp]),
    'deepipr_gamma_beta_fwd_multi': (_int, [_vp, _int, _vp]),
    'deepipr_gamma_beta_bwd_multi': (_int, [_vp, _int, _int, _vp]),
    'deepipr_gamma_beta_dkey_workspace_bytes': (_sz, [_int, _int, _int]),
    'deepipr_gamma_beta_dkey': (_int, [_f32p, _f32p, _f32p, _int, _int, _int, _int, _int, _int, _int, _int, _int,
                                       _f32p, _vp, _vp]),
    'deepipr_affine_relu_fwd': (_int, [_f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _vp]),
    'deepipr_affine_relu_bwd_workspace_bytes': (_sz, [_int, _int, _int]),
    'deepipr_affine_relu_bwd': (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int,
                                       _vp, _vp]),
    'deepipr_sign_loss_fwd': (_int, [_f32p, _f32p, _flt, _flt, _flt, _int, _f32p, _f32p, _i8p, _vp]),
    'deepipr_sign_loss_bwd': (_int, [_f32p, _f32p, _f32p, _flt, _flt, _flt, _int, _f32p, _vp]),
    'deepipr_passport_fwd': (_int, [_f32p, _f32p, _f64p, _f32p, _flt, _flt, _flt, _int, _int, _int, _int, _int,
                                    _f32p, _f32p, _f32p, _f32p, _f32p, _i8p, _vp]),
    'deepipr_passport_bwd_workspace_bytes': (_sz, [_int, _int, _int]),
    'deepipr_passport_bwd': (_int, [_f32p, _f32p, _f32p, _f32p, _f64p, _f32p, _flt, _flt, _flt, _f32p, _f32p,
                                    _f32p, _int, _int, _int, _int, _int, _f32p, _f32p, _f32p, _f32p, _vp, _vp]),
    'deepipr_ce_top1_supported': (_int, [_int, _int]),
    'deepipr_ce_top1_workspace_bytes': (_sz, [_int]),
    'deepipr_ce_top1_fwd': (_int, [_f32p, _vp, _int, _int, _f32p, _f32p, _f32p, _vp, _vp]),
    'deepipr_ce_bwd': (_int, [_f32p, _f32p, _vp, _f32p, _int, _int, _f32p, _vp]),
    'deepipr_pooled_linear_supported': (_int, [_int, _int, _int, _int]),
    'deepipr_pooled_linear_fwd': (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _vp]),
    'deepipr_pooled_linear_bwd': (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _int, _int, _int, _int, _vp]),
    'deepipr_scalar_sums': (_int, [_vp, _int, _int, _f32p, _vp]),
    'deepipr_add_relu_fwd': (_int, [_f32p, _f32p, _f32p, _sz, _vp]),
    'deepipr_relu_bwd': (_int, [_f32p, _f32p, _f32p, _sz, _vp]),
    'deepipr_relu_bwd2': (_int, [_f32p, _f32p, _f32p, _f32p, _sz, _vp]),
    'deepipr_subsample2': (_int, [_f32p, _f32p, _sz, _int, _int, _vp]),
    'deepipr_upsample2_zero': (_int, [_f32p, _f32p, _sz, _int, _int, _vp]),
    'deepipr_maxpool3x3s2_fwd': (_int, [_f32p, _f32p, _vp, _sz, _int, _int, _vp]),
    'deepipr_maxpool3x3s2_bwd': (_int, [_f32p, _vp, _f32p, _sz, _int, _int, _vp]),
    'deepipr_maxpool2x2s2_fwd': (_int, [_f32p, _f32p, _vp, _sz, _int, _int, _vp]),
    'deepipr_maxpool2x2s2_bwd': (_int, [_f32p, _vp, _f32p, _sz, _int, _int, _vp]),
    'deepipr_sgd_momentum_step': (_int, [_f32p, _f32p, _f32p, _sz, _flt, _flt, _flt, _flt, _vp]),
    'deepipr_sgd_momentum_step_dev': (_int, [_f32p, _f32p, _f32p, _sz, _f32p, _vp]),
    'deepipr_sgd_momentum_chunk': (_int, []),
    'deepipr_sgd_momentum_step_multi': (_int, [_f32p, _f32p, _vp, _int, _sz, _f32p, _vp]),
    'deepipr_passport_bn_workspace_bytes': (_sz, [_int, _int, _int]),
    'deepipr_passport_bn_fwd': (_int, [_f32p, _f32p, _f64p, _f32p, _f32p, _f32p, _flt, _flt, _flt, _f32p, _f32p, _vp,
                                       _flt, _flt, _int, _int, _int, _int, _int, _int, _f32p, _f32p, _f32p, _f32p,
                                       _f32p, _f32p, _i8p, _f32p, _vp, _vp, _vp]),
    'deepipr_passport_bn_bwd': (_int, [_f32p, _f32p, _f32p, _f64p, _f32p, _flt, _flt, _flt, _f32p, _f32p, _f32p,
                                       _int, _int, _int, _int, _int, _int, _f32p, _f32p, _f32p, _f32p, _f32p, _vp,
                                       _vp, _f32p, _f32p, _f32p, _vp]),
    'deepipr_passport_bn_resident': (_int, [_int, _int, _int, _int]),
    'deepipr_passport_bn_slices': (_int, [_int, _int, _int]),
    'deepipr_passport_bn_passes': (_int, [_int, _int, _int, _int]),
    'deepipr_bn_dual_tail_supported': (_int, [_int, _int, _int, _int]),
    'deepipr_bn_dual_tail_fwd': (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _vp, _f32p, _f32p, _vp,
                                        _flt, _flt, _flt, _flt, _int, _int, _int, _int, _int, _f32p, _f32p, _f32p, _vp,
                                        _vp]),
    'deepipr_bn_dual_tail_bwd': (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                        _f32p, _f32p, _int, _int, _int, _int, _int, _vp, _vp]),
    'deepipr_passport_gn_supported': (_int, [_int, _int, _int, _int]),
    'deepipr_passport_gn_workspace_bytes': (_sz, [_int, _int, _int]),
    'deepipr_passport_gn_fwd': (_int, [_f32p, _f32p, _f64p, _f32p, _f32p, _f32p, _flt, _flt, _flt, _int, _flt, _int,
                                       _int, _int, _int, _int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _i8p, _vp]),
    'deepipr_passport_gn_bwd': (_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f64p, _f32p, _flt, _flt, _flt, _f32p, _f32p,
                                       _f32p, _int, _int, _int, _int, _int, _int, _f32p, _f32p, _f32p, _f32p, _vp,
                                       _vp]),
    'deepipr_set_resident': (_int, [_int]),
    'deepipr_profile_read_bytes': (_int, [_int, _c.POINTER(_c.c_double)]),
    'deepipr_profile_scope': (_int, [_int]),
    'deepipr_profile_read_scope': (_int, [_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_longlong),
                                          _c.POINTER(_c.c_double)]),
    'deepipr_conv_supported': (_int, [_int] * 9),
    'deepipr_conv_fwd': (_int, [_f32p, _f32p, _f32p] + [_int] * 8 + [_vp]),
    'deepipr_conv_dgrad': (_int, [_f32p, _f32p, _f32p] + [_int] * 8 + [_vp]),
    'deepipr_conv_workspace_bytes': (_sz, [_int] * 9),
    'deepipr_conv_fwd_ws': (_int, [_f32p, _f32p, _f32p] + [_int] * 8 + [_vp, _sz, _vp]),
    'deepipr_conv_dgrad_ws': (_int, [_f32p, _f32p, _f32p] + [_int] * 8 + [_vp, _sz, _vp]),
    'deepipr_conv_wino_image_bytes': (_sz, [_int, _int]),
    'deepipr_conv_wino_max_layers': (_int, []),
    'deepipr_conv_wino_transform_multi': (_int, [_vp, _int, _vp]),
    'deepipr_conv_fwd_pre': (_int, [_f32p, _f32p, _f32p] + [_int] * 5 + [_vp, _sz, _vp]),
    'deepipr_conv_dgrad_pre': (_int, [_f32p, _f32p, _f32p] + [_int] * 5 + [_vp, _sz, _vp]),
    'deepipr_conv_set_algo': (_int, [_int]),
    'deepipr_conv_get_algo': (_int, []),
    'deepipr_conv_algo_of': (_int, [_int] * 9),
    'deepipr_conv_set_arith': (_int, [_int]),
    'deepipr_conv_get_arith': (_int, []),
    'deepipr_conv_wgrad_workspace_bytes': (_sz, [_int] * 9),
    'deepipr_conv_wgrad': (_int, [_f32p, _f32p, _f32p] + [_int] * 9 + [_f32p, _f32p, _f64p, _vp, _sz, _vp]),
}
# exported by the measurement / test build only (libdeepipr_hip_trace.so, DEEPIPR_LIB=...): bound when present
TEST_HOOK_SIGNATURES = {
    'deepipr_debug_tune': (_int, [_c.c_char_p, _int]),
    'deepipr_debug_trace': (_int, [_vp]),
    'deepipr_debug_wino_trace': (_int, [_vp]),
}
ABI_VERSION = 12
SYNC_WORDS = 2 * (256 * 30 * 4 + 4096 + 63 * 2048) + 16     # DEEPIPR_SYNC_WORDS
SYNC_TIMEOUT_WORD = 2 * (256 * 30 * 4 + 4096 + 63 * 2048)   # DEEPIPR_SYNC_TIMEOUT_WORD


GEMV_MAX_LAYERS = 16                    # DEEPIPR_GEMV_MAX_LAYERS


class GemvLayer(_c.Structure):          # DeepiprGemvLayer
    _fields_ = [('W', _vp), ('m', _vp), ('gamma', _vp), ('beta', _vp), ('Co', _int), ('K', _int)]


class Rank2Layer(_c.Structure):         # DeepiprRank2Layer
    _fields_ = [('dgamma', _vp), ('dbeta', _vp), ('m', _vp), ('dW', _vp), ('Co', _int), ('K', _int)]


WINO_MAX_LAYERS = 24                    # DEEPIPR_WINO_MAX_LAYERS


class WinoLayer(_c.Structure):          # DeepiprWinoLayer
    _fields_ = [('W', _vp), ('Uf', _vp), ('Ud', _vp), ('Co', _int), ('Ci', _int)]


class HipLibraryMissing(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library (cached).  Raises HipLibraryMissing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            'deepipr_amd: %s not found. The passport layer has no CPU or PyTorch fallback; build the HIP '
            'library first: make -C %s' % (LIB_PATH, os.path.dirname(LIB_PATH)))
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryMissing('deepipr_amd: cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise HipLibraryMissing('deepipr_amd: %s does not export %s (stale build?)' % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in TEST_HOOK_SIGNATURES.items():
        fn = getattr(handle, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if handle.deepipr_abi_version() != ABI_VERSION:
        raise HipLibraryMissing('deepipr_amd: %s has ABI version %d, expected %d' %
                                (LIB_PATH, handle.deepipr_abi_version(), ABI_VERSION))
    _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().deepipr_last_error()
        raise RuntimeError('deepipr_hip.%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


PROFILE_KERNELS = ['pooled_patch_mean', 'gamma_beta_fwd', 'gamma_beta_bwd', 'affine_fwd', 'affine_bwd',
                   'reduce_partials', 'passport_bwd_finish', 'sign_loss_fwd', 'sign_loss_bwd', 'dkey', 'reserved',
                   'bn_stats', 'bn_affine_fwd', 'bn_bwd_reduce', 'bn_affine_bwd', 'sgd', 'add_relu', 'bn_res_fwd',
                   'bn_res_bwd', 'gn_fwd', 'gn_bwd', 'conv_wgrad', 'conv_wgrad_reduce', 'conv_fwd', 'conv_dgrad', 'conv_wgrad_b3', 'conv_split_sum',
                   'conv_wino_fwd', 'conv_wino_dgrad', 'conv_wino_wgrad', 'conv_wino_weights', 'conv1x1_wgrad', 'maxpool', 'resample2', 'conv1x1_fwd', 'conv1x1_dgrad', 'head']


class ExternalEvent:
    """An event a captured step records from INSIDE its hipGraph (deepipr_event_record on the capturing stream adds an
    external event-record node) and another stream waits for after the launch: how experiments/staged.py starts a
    gradient bucket's all-reduce while the replayed backward is still running.  Has torch.cuda.Event's `wait`."""

    def __init__(self):
        h = _c.c_void_p()
        check(lib().deepipr_event_create(_c.byref(h)), 'event_create')
        self.handle = h.value

    def record(self, stream):
        """stream: a torch.cuda.Stream (capturing or not)."""
        check(lib().deepipr_event_record(self.handle, stream.cuda_stream), 'event_record')

    def wait(self, stream):
        check(lib().deepipr_stream_wait_event(stream.cuda_stream, self.handle), 'stream_wait_event')

    def synchronize(self):
        """Block the host until the latest record of the event has completed."""
        check(lib().deepipr_event_synchronize(self.handle), 'event_synchronize')

    def __del__(self):
        try:
            if self.handle:
                lib().deepipr_event_destroy(self.handle)
        except Exception:
            pass


def has_test_hooks():
    return hasattr(lib(), 'deepipr_debug_tune')


def debug_tune(key, value):
    """Planning knobs / time-out test hooks of the single-pass kernels -- measurement / test build only
    (DEEPIPR_LIB=.../libdeepipr_hip_trace.so; include/deepipr_hip.h: deepipr_debug_tune)."""
    if not has_test_hooks():
        raise RuntimeError('deepipr_debug_tune is not part of the production library: load the test build '
                           '(make -C deepipr_amd/csrc trace; DEEPIPR_LIB=.../libdeepipr_hip_trace.so)')
    check(lib().deepipr_debug_tune(key.encode(), int(value)), 'debug_tune')


def set_resident(on):
    """Enable (default) / disable the register-resident single-pass BatchNorm kernels process-wide."""
    check(lib().deepipr_set_resident(int(bool(on))), 'set_resident')


def profile_enable(on):
    """1 / True = reset and enable, 2 = resume without resetting the counters, 0 / False = pause."""
    check(lib().deepipr_profile_enable(int(on)), 'profile_enable')


def profile_read():
    """{kernel name: (total_ms, launches)} since the last profile_enable(True)."""
    out = {}
    for i, name in enumerate(PROFILE_KERNELS):
        ms, n = ctypes.c_double(), ctypes.c_longlong()
        check(lib().deepipr_profile_read(i, ctypes.byref(ms), ctypes.byref(n)), 'profile_read')
        out[name] = (ms.value, n.value)
    return out


def profile_scope(on):
    """Open / close the scope whose launches are also accounted apart (profile_read_scope)."""
    check(lib().deepipr_profile_scope(int(bool(on))), 'profile_scope')


def profile_read_scope():
    """{kernel name: (total_ms, launches, algorithmic bytes)} of the launches issued inside a profile scope."""
    out = {}
    for i, name in enumerate(PROFILE_KERNELS):
        ms, n, b = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
        check(lib().deepipr_profile_read_scope(i, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(b)), 'profile_read_scope')
        out[name] = (ms.value, n.value, b.value)
    return out


def profile_read_bytes():
    """{kernel name: algorithmic HBM bytes of the launches timed so far} (streaming kernels; 0 for the rest)."""
    out = {}
    for i, name in enumerate(PROFILE_KERNELS):
        v = ctypes.c_double()
        check(lib().deepipr_profile_read_bytes(i, ctypes.byref(v)), 'profile_read_bytes')
        out[name] = v.value
    return out
