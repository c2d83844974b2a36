"""ctypes binding of libraftcorr.so (C ABI declared in include/raftcorr.h).

The library is the only compute path: if it is missing or fails to load, importing
:mod:`raft_amd.corr` symbols still works (so CPU-only hosts can inspect the API), but every
operator raises ``RuntimeError`` -- there is no CPU or PyTorch fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libraftcorr.so")

MAX_LEVELS = 8
PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 2, "fp16x2": 3}          # RAFTCORR_PREC_*
VOLUMES = {"fp32": 0, "fp16": 1, "bf16": 2}                              # RAFTCORR_VOL_*
MODE_FORK = 0
MODE_CANONICAL = 1

_c_fp = ctypes.c_void_p          # device pointers travel as plain addresses
_c_int = ctypes.c_int

# name -> argtypes; mirrors include/raftcorr.h one to one
_SIGNATURES = {
    "raftcorr_version": ([], _c_int),
    "raftcorr_last_error": ([], ctypes.c_char_p),
    "raftcorr_level_layout": ([_c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_int32),
                               ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)], _c_int),
    "raftcorr_level_positions": ([_c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_int32)], _c_int),
    "raftcorr_build_workspace_bytes": ([_c_int] * 4, ctypes.c_int64),
    "raftcorr_build_bwd_workspace_bytes": ([_c_int] * 4, ctypes.c_int64),
    "raftcorr_build_pyramid": ([_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int,
                                ctypes.POINTER(ctypes.c_void_p), _c_fp, ctypes.c_int64,
                                ctypes.c_void_p], _c_int),
    "raftcorr_build_pyramid_prec": ([_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int,
                                     ctypes.POINTER(ctypes.c_void_p), _c_fp, ctypes.c_int64, _c_int,
                                     ctypes.c_void_p], _c_int),
    "raftcorr_build_pyramid_noalloc": ([_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int,
                                        ctypes.POINTER(ctypes.c_void_p), _c_int, ctypes.c_void_p], _c_int),
    "raftcorr_conv1x1_packed_bytes": ([_c_int, _c_int], ctypes.c_int64),
    "raftcorr_conv1x1_pack_weight": ([_c_fp, _c_int, _c_int, _c_fp, ctypes.c_int64, ctypes.c_void_p], _c_int),
    "raftcorr_lookup_conv1x1_fwd": ([ctypes.POINTER(ctypes.c_void_p), _c_int, _c_int, _c_int, _c_int, _c_int,
                                     _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_fp, ctypes.c_void_p], _c_int),
    "raftcorr_lookup_fwd": ([ctypes.POINTER(ctypes.c_void_p), _c_int, _c_int, _c_int, _c_int, _c_int,
                             _c_fp, _c_int, _c_fp, ctypes.c_void_p], _c_int),
    "raftcorr_build_pyramid_vol": ([_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_void_p),
                                    _c_fp, ctypes.c_int64, _c_int, _c_int, ctypes.c_void_p], _c_int),
    "raftcorr_lookup_fwd_vol": ([ctypes.POINTER(ctypes.c_void_p), _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                 _c_fp, _c_int, _c_fp, ctypes.c_void_p], _c_int),
    "raftcorr_lookup_bwd": ([ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                             _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_int, _c_fp, _c_fp,
                             ctypes.c_void_p], _c_int),
    "raftcorr_build_bwd": ([_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int,
                            ctypes.POINTER(ctypes.c_void_p), _c_fp, _c_fp, _c_fp, ctypes.c_int64,
                            ctypes.c_void_p], _c_int),
    "raftcorr_build_bwd_deferred_workspace_bytes": ([_c_int] * 4, ctypes.c_int64),
    "raftcorr_build_bwd_deferred_supported": ([_c_int] * 4, _c_int),
    "raftcorr_lookup_bwd_record": ([_c_fp] + [_c_int] * 4 + [_c_fp, ctypes.c_void_p], _c_int),
    "raftcorr_lookup_bwd_record_many": ([ctypes.POINTER(ctypes.c_void_p)] + [_c_int] * 5 +
                                        [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p], _c_int),
    "raftcorr_build_bwd_deferred": ([_c_fp, _c_fp] + [_c_int] * 8 + [ctypes.POINTER(ctypes.c_void_p)] * 3 +
                                    [_c_fp, _c_fp, _c_fp, ctypes.c_int64, ctypes.c_void_p], _c_int),
    "raftcorr_alt_forward": ([_c_fp, _c_fp, _c_fp, _c_fp] + [_c_int] * 8 + [ctypes.c_void_p], _c_int),
    "raftcorr_alt_backward": ([_c_fp] * 7 + [_c_int] * 8 + [ctypes.c_void_p], _c_int),
    "raftcorr_alt_packed_bytes": ([_c_int] * 4, ctypes.c_int64),
    "raftcorr_alt_scales": ([_c_fp, _c_fp] + [_c_int] * 4 + [_c_fp, ctypes.c_void_p], _c_int),
    "raftcorr_alt_pack": ([_c_fp] + [_c_int] * 4 + [_c_fp, _c_fp, ctypes.c_int64, ctypes.c_void_p], _c_int),
    "raftcorr_alt_lookup_fwd": ([_c_fp, ctypes.POINTER(ctypes.c_void_p), _c_fp] + [_c_int] * 6 +
                                [_c_fp, _c_fp, ctypes.c_void_p], _c_int),
    "raftcorr_alt_lookup_fwd_sorted_workspace_bytes": ([_c_int] * 4, ctypes.c_int64),
    "raftcorr_alt_lookup_fwd_sorted": ([_c_fp, ctypes.POINTER(ctypes.c_void_p), _c_fp] + [_c_int] * 6 +
                                       [_c_fp, _c_fp, _c_fp, ctypes.c_int64, ctypes.c_void_p], _c_int),
    "raftcorr_alt_tiles_bytes": ([_c_int] * 4, ctypes.c_int64),
    "raftcorr_alt_pack_tiles": ([_c_fp] + [_c_int] * 4 + [_c_fp, _c_fp, ctypes.c_int64, ctypes.c_void_p], _c_int),
    "raftcorr_alt_lookup_bwd_workspace_bytes": ([_c_int] * 6, ctypes.c_int64),
    "raftcorr_alt_lookup_bwd": ([_c_fp, ctypes.POINTER(ctypes.c_void_p), _c_fp] + [_c_int] * 6 +
                                [_c_fp, _c_fp, _c_fp, ctypes.POINTER(ctypes.c_void_p), _c_int, _c_fp, ctypes.c_int64,
                                 ctypes.c_void_p], _c_int),
    "raftcorr_alt_lookup_bwd_sorted_workspace_bytes": ([_c_int] * 6, ctypes.c_int64),
    "raftcorr_alt_lookup_bwd_sorted": ([_c_fp, ctypes.POINTER(ctypes.c_void_p), _c_fp] + [_c_int] * 6 +
                                       [_c_fp, _c_fp, _c_fp, ctypes.POINTER(ctypes.c_void_p), _c_int, _c_fp, _c_fp, ctypes.c_int64,
                                        ctypes.c_void_p], _c_int),
    "raftcorr_alt_lookup_bwd_many_workspace_bytes": ([_c_int] * 8, ctypes.c_int64),
    "raftcorr_alt_lookup_bwd_many": ([_c_fp, ctypes.POINTER(ctypes.c_void_p), _c_fp] + [_c_int] * 7 +
                                     [ctypes.POINTER(ctypes.c_void_p)] * 3 + [_c_int, _c_fp, ctypes.POINTER(ctypes.c_void_p), _c_int,
                                      _c_fp, ctypes.c_int64, ctypes.c_void_p], _c_int),
    "raftcorr_alt_prepare_workspace_bytes": ([_c_int] * 5, ctypes.c_int64),
    "raftcorr_alt_prepare": ([_c_fp, _c_fp] + [_c_int] * 5 + [_c_fp, _c_fp, ctypes.POINTER(ctypes.c_void_p), _c_fp,
                                                              ctypes.POINTER(ctypes.c_void_p), _c_fp, ctypes.c_int64,
                                                              ctypes.c_void_p], _c_int),
    "raftcorr_alt_features_bwd": ([_c_fp, ctypes.POINTER(ctypes.c_void_p)] + [_c_int] * 5 + [_c_fp, _c_fp, ctypes.c_void_p],
                                  _c_int),
}

EXPORTS = tuple(_SIGNATURES)

_lib = None
_load_error = None


def load():
    """Return the loaded library or raise RuntimeError (never falls back)."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is None:
        try:
            lib = ctypes.CDLL(LIB_PATH)
            for name, (argtypes, restype) in _SIGNATURES.items():
                fn = getattr(lib, name)
                fn.argtypes = argtypes
                fn.restype = restype
            _lib = lib
            return _lib
        except (OSError, AttributeError) as e:      # missing file / missing symbol
            _load_error = e
    raise RuntimeError(
        f"raft_amd: the HIP extension {LIB_PATH} is not available ({_load_error}); build it with "
        "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C raft_amd/csrc`. "
        "There is no CPU fallback.")


_fast = {}


def fast(name):
    """A second handle of an entry point WITHOUT argtypes, for calls on the per-iteration path: the
    caller passes prebuilt ctypes objects (c_int, c_void_p, pointer arrays), which skips ctypes'
    per-argument conversion -- about 1 us of host time per call, as much as a lookup's launch
    overhead.  Pointers must be wrapped in c_void_p (a bare Python int would be passed as a C int)."""
    fn = _fast.get(name)
    if fn is None:
        load()
        fn = getattr(ctypes.CDLL(LIB_PATH), name)       # same dlopen handle, separate function object
        fn.restype = ctypes.c_int
        _fast[name] = fn
    return fn


_addr = {}


def fn_addr(name):
    """Address of an entry point (for raft_amd._pyfast, which calls the C ABI directly)."""
    a = _addr.get(name)
    if a is None:
        a = _addr[name] = ctypes.cast(getattr(load(), name), ctypes.c_void_p).value
    return a


C_MODE = {MODE_FORK: ctypes.c_int(MODE_FORK), MODE_CANONICAL: ctypes.c_int(MODE_CANONICAL)}


def check(rc, what):
    if rc != 0:
        msg = load().raftcorr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"raftcorr {what} failed (code {rc}): {msg}")


ALT_MAX_LOOKUPS = 16                                   # RAFTCORR_ALT_MAX_LOOKUPS (include/raftcorr.h)


def ptr_array(tensors):
    """Host array of device pointers (None -> NULL)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None or t.numel() == 0 else t.data_ptr()
    return arr


def level_dims(H, W, num_levels):
    """Logical pooled sizes, H_l = floor(H_{l-1} / 2) (pure host arithmetic, no library call)."""
    dims = []
    for _ in range(num_levels):
        dims.append((H, W))
        H //= 2
        W //= 2
    return dims


def level_layout(H, W, num_levels):
    """[(H_l, W_l, slab_l)] from the library (the layout's single source of truth)."""
    hl = (ctypes.c_int32 * num_levels)()
    wl = (ctypes.c_int32 * num_levels)()
    sl = (ctypes.c_int64 * num_levels)()
    check(load().raftcorr_level_layout(H, W, num_levels, hl, wl, sl), "level_layout")
    return [(hl[i], wl[i], sl[i]) for i in range(num_levels)]


_pos_cache = {}


def level_positions(H, W, level):
    """numpy int64 array (H_l*W_l,) of slab offsets of the logical cells of a level (cached)."""
    import numpy as np
    key = (H, W, level)
    if key not in _pos_cache:
        h, w = H >> level, W >> level
        buf = (ctypes.c_int32 * max(h * w, 1))()
        check(load().raftcorr_level_positions(H, W, level, buf), "level_positions")
        _pos_cache[key] = np.frombuffer(buf, dtype=np.int32, count=h * w).astype(np.int64)
    return _pos_cache[key]
