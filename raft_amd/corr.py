"""Host-side mirror of the reference's correlation interface (core/corr.py), backed by
libraftcorr.so (hand-written gfx950 kernels, C ABI in include/raftcorr.h).

Same names, arguments and defaults as the reference:

* ``CorrBlock(fmap1, fmap2, num_levels=2, radius=4)(coords)``       core/corr.py:12-61
* ``CorrBlock.corr(fmap1, fmap2)``                                   core/corr.py:53-61
* ``AlternateCorrBlock(fmap1, fmap2, num_levels=4, radius=4)(coords)`` core/corr.py:64-92

(``utils.bilinear_sampler`` has no counterpart here: the window lookup is fused into raftcorr_lookup_fwd, and general
sampling grids are outside the hot path -- callers keep the reference's own helper.)

``CorrBlock.__call__`` is bimodal: 3-D ``(B, N, 2)`` coords select the
fork's semantics exactly as shipped (no per-level scaling, ``(B, N, C)`` output); 4-D
``(B, 2, H, W)`` coords select the canonical RAFT semantics that core/raft.py:125 and
core/update.py:82-91 consume (coords / 2**l per level, ``(B, C, H, W)`` output).

Autograd: the reference differentiates through matmul / avg_pool2d / grid_sample and
materialises a dense, zero-filled gradient of a whole pyramid level for EVERY lookup.  Here
a lookup's backward only records (coords, d(out)); when autograd reaches the build node one
kernel makes every query's folded level-0 gradient row in LDS and two contractions turn it into
the feature gradients (``backward="deferred"``, the default where the shape allows it).  The
older route -- all lookups accumulate into ONE dense gradient pyramid that is folded once --
remains as ``backward="accumulate"`` and for gradients that arrive on the pyramid itself.

PyTorch is used for device memory, streams and autograd bookkeeping only; there is no
fallback path: without the HIP extension every operator raises RuntimeError.
"""
import contextlib
import ctypes

import weakref

import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _lib

try:                                    # CPython glue for the per-iteration call (csrc/pyfast.c); same C ABI
    from . import _pyfast
except ImportError:                     # not built: the ctypes handle below does the same call
    _pyfast = None

__all__ = ["CorrBlock", "AlternateCorrBlock"]

_ALIGN = 64   # level offsets inside the packed pyramid buffer, in elements (256 B)


_NULL = contextlib.nullcontext()


try:                                    # raw hipStream_t of the current stream, without wrapper objects
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                  # pragma: no cover
    _raw_stream = None


def _stream(t):
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


_current_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _guard(t):
    """Device guard only when the tensor does not live on the current device (the context
    manager costs more host time than a lookup kernel runs)."""
    return _NULL if t.device.index == _current_device() else torch.cuda.device(t.device)


def _require_cuda(**tensors):
    """CHECK_CUDA of correlation.cpp:19, plus: all on ONE device (the kernels take raw pointers, a
    tensor of another GPU would be a memory fault rather than an error)."""
    dev = None
    for name, t in tensors.items():
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"{name} is on {t.device}, expected {dev}")


def _f32c(t):
    return t.float().contiguous()


class _Pyramid:
    """Packed pyramid storage (patch-tile order, include/raftcorr.h) + the shared gradient
    accumulator of one CorrBlock."""

    def __init__(self, B, H, W, num_levels, device, volume="fp32"):
        self.B, self.H, self.W, self.L = B, H, W, num_levels
        self.volume = volume                                    # element type of the stored volume
        self.vol = _lib.VOLUMES[volume]
        self.vdtype = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[volume]
        self.N = H * W
        self.layout = _lib.level_layout(H, W, num_levels)       # [(H_l, W_l, slab_l)]
        self.offsets = []
        off = 0
        for (_, _, slab) in self.layout:
            self.offsets.append(off)
            off += (B * self.N * slab + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = max(off, 1)
        self.device = device
        self.c_dims = tuple(ctypes.c_int(v) for v in (B, H, W, num_levels))     # prebuilt call arguments
        self._c_radius = {}
        self._shapes = {}
        self.flat = None          # detached storage of the volume levels
        self.ptrs = None          # ctypes array of level pointers
        self.dflat = None         # gradient accumulator (allocated on first lookup backward)
        self.dptrs = None
        self.dtask = None         # autograd graph task that owns the accumulator's contents
        self._index = {}
        # deferred backward (raftcorr_build_bwd_deferred): the lookups of a backward pass only record (coords, d(out)),
        # the build node turns all of them into the folded level-0 gradient in one kernel.  Shapes it does not take
        # (more than 4 levels, very large maps) keep the accumulate-then-fold route.
        self.deferred = False
        self.records = []         # [(mode, radius, coords, record (B, N, C) | None, pending d(out) | None)] of the running backward pass
        self.rtask = None

    def out_shape(self, radius, mode):
        shape = self._shapes.get((radius, mode))
        if shape is None:
            C = self.L * (2 * radius + 1) ** 2
            shape = (self.B, self.N, C) if mode == _lib.MODE_FORK else (self.B, C, self.H, self.W)
            self._shapes[(radius, mode)] = shape
        return shape

    def c_radius(self, radius):
        c = self._c_radius.get(radius)
        if c is None:
            c = self._c_radius[radius] = ctypes.c_int(radius)
        return c

    def slabs(self, flat):
        """Per level: (B*N, slab_l) view of the packed buffer (device layout)."""
        return [flat[off:off + self.B * self.N * slab].view(self.B * self.N, slab)
                for (_, _, slab), off in zip(self.layout, self.offsets)]

    def logical(self, flat):
        """Per level: the reference's (B*N, 1, H_l, W_l) tensor (core/corr.py:22-27), gathered out
        of the patch-tile order with differentiable stock indexing (off the hot path)."""
        out = []
        for l, ((h, w, _), s) in enumerate(zip(self.layout, self.slabs(flat))):
            if l not in self._index:
                self._index[l] = torch.from_numpy(_lib.level_positions(self.H, self.W, l)).to(self.device)
            out.append(s.index_select(1, self._index[l]).view(self.B * self.N, 1, h, w).float())
        return out

    def attach(self, flat):
        self.flat = flat
        self.ptrs = _lib.ptr_array(self.slabs(flat))
        self.ptrs_addr = ctypes.addressof(self.ptrs)

    def grad_buffer(self):
        """The gradient accumulator of the backward pass that is running now.  What an EARLIER backward
        pass left behind without ever reaching the build node (torch.autograd.grad(out, coords) only, or
        an exception half way) is dropped, not added to: the buffer carries the id of the graph task that
        wrote it."""
        task = _graph_task_id()
        if self.dflat is None:
            self.dflat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            self.dptrs = _lib.ptr_array(self.slabs(self.dflat))
        elif self.dtask != task:
            self.dflat.zero_()
        self.dtask = task
        return self.dflat

    def pending_grad(self):
        """True when lookups of the running backward pass have written into the accumulator."""
        return self.dflat is not None and self.dtask == _graph_task_id()

    def release_grad(self):
        self.dflat = None
        self.dptrs = None

    def set_deferred(self, radius):
        self.deferred = bool(_lib.load().raftcorr_build_bwd_deferred_supported(self.H, self.W, self.L, radius))

    def add_record(self, mode, radius, coords, rec, pending=None):
        """A lookup's (coords, d(out) as (B, N, C)) for the backward pass that is running now; what an earlier pass
        left behind without reaching the build node is dropped (cf. grad_buffer).  pending: a canonical d(out)
        (B, C, H, W) that is still to be transposed (rec is None): the build node does that for all lookups at once."""
        task = _graph_task_id()
        if self.rtask != task:
            self.records = []
        self.rtask = task
        self.records.append((mode, radius, coords, rec, pending))

    def records_take(self, mode):
        """One deferred launch serves one coordinate convention: a block that is looked up in both conventions inside
        one backward pass sends the second convention's lookups through the dense accumulator."""
        return self.rtask != _graph_task_id() or not self.records or self.records[0][0] == mode

    def take_records(self):
        recs = self.records if self.rtask == _graph_task_id() else []
        self.records, self.rtask = [], None
        return recs


_graph_task_id = getattr(torch._C, "_current_graph_task_id", lambda: -1)


def _build(fmap1, fmap2, pyr, precision="fp32"):
    lib = _lib.load()

    B, D, H, W = fmap1.shape
    flat = torch.empty(pyr.total, dtype=pyr.vdtype, device=fmap1.device)
    pyr.attach(flat)
    wbytes = lib.raftcorr_build_workspace_bytes(B, D, H, W)
    work = torch.empty(wbytes, dtype=torch.uint8, device=fmap1.device)
    with _guard(fmap1):
        _lib.check(lib.raftcorr_build_pyramid_vol(fmap1.data_ptr(), fmap2.data_ptr(), B, D, H, W, pyr.L,
                                                  pyr.ptrs, work.data_ptr(), wbytes,
                                                  _lib.PRECISIONS[precision], pyr.vol, _stream(fmap1)), "build_pyramid")
    return flat


def _channels_last_rows(dout):
    """(B, N, C) view of a canonical d(out) (B, C, H, W) that sits in memory channels-last (what a model run in
    torch.channels_last hands back): it IS the record, no transposition pass.  None for any other layout."""
    if dout.dim() != 4 or dout.dtype is not torch.float32 or dout.is_contiguous():
        return None
    if not dout.is_contiguous(memory_format=torch.channels_last):
        return None
    B, C, H, W = dout.shape
    return dout.permute(0, 2, 3, 1).reshape(B, H * W, C)        # a view: the permuted tensor is contiguous


def _record(pyr, mode, radius, coords, dout, rows=None):
    """Note a lookup's gradient for the deferred backward: fork-mode d(out) (B, N, C) is the record itself, so is a
    channels-last canonical d(out) (`rows`); a canonical d(out) (B, C, H, W) is transposed to (B, N, C) by
    raftcorr_lookup_bwd_record."""
    if rows is not None:
        rec = rows
    elif mode == _lib.MODE_FORK:
        rec = dout
    else:
        # kept as it is: the build node transposes the d(out) of all lookups with one raftcorr_lookup_bwd_record_many
        # (a launch per lookup is one round of workgroups that first all read, then all write; the rounds of one launch
        # for all lookups overlap: 12 x 18 us -> ~ 150 us at Things B=6)
        pyr.add_record(mode, radius, coords, None, dout)
        return
    pyr.add_record(mode, radius, coords, rec)


def _transpose_pending(pyr, recs, like):
    """records with a pending canonical d(out) -> (B, N, C) rows, all of them in one library call."""
    todo = [i for i, r in enumerate(recs) if r[3] is None]
    if not todo:
        return recs
    B, C = recs[todo[0]][4].shape[:2]
    rows = torch.empty(len(todo), B, pyr.N, C, dtype=torch.float32, device=like.device)
    dptr = _lib.ptr_array([recs[i][4] for i in todo])
    rptr = _lib.ptr_array([rows[j] for j in range(len(todo))])
    _lib.check(_lib.load().raftcorr_lookup_bwd_record_many(dptr, len(todo), B, pyr.H, pyr.W, C, rptr, _stream(like)),
               "lookup_bwd_record_many")
    recs = list(recs)
    for j, i in enumerate(todo):
        recs[i] = recs[i][:3] + (rows[j], None)
    return recs


class _BuildPyramid(torch.autograd.Function):
    """fmap1, fmap2 -> (packed pyramid, anchor).  The anchor is a 1-element tensor that every
    lookup takes as an input: it carries the graph edge lookups -> build without a dense
    gradient per lookup (the lookups write into pyr.dflat and return no anchor gradient at all: the
    engine runs this node once all of them have run, with danchor = None)."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, pyr, precision):
        flat = _build(fmap1, fmap2, pyr, precision)
        ctx.save_for_backward(fmap1, fmap2)
        ctx.pyr = pyr
        ctx.set_materialize_grads(False)
        anchor = torch.empty(1, dtype=torch.float32, device=fmap1.device)     # (a graph edge; nobody reads the value)
        return flat, anchor

    @staticmethod
    @once_differentiable
    def backward(ctx, dflat, danchor):
        fmap1, fmap2 = ctx.saved_tensors
        pyr = ctx.pyr
        if dflat is not None:                       # someone differentiated corr_pyramid directly
            acc = pyr.grad_buffer()
            acc += dflat
        recs = pyr.take_records()
        if not pyr.pending_grad() and not recs:
            pyr.release_grad()
            return torch.zeros_like(fmap1), torch.zeros_like(fmap2), None, None
        lib = _lib.load()
        B, D, H, W = fmap1.shape
        d1 = torch.empty_like(fmap1)
        d2 = torch.empty_like(fmap2)
        if recs:
            # deferred lookups (+ the dense accumulator when something wrote into it): one kernel builds the folded
            # level-0 gradient from all records.  Lookups of one block share mode and radius.
            mode, radius = recs[0][0], recs[0][1]
            dense = pyr.pending_grad()
            wbytes = lib.raftcorr_build_bwd_workspace_bytes(B, D, H, W) if dense else \
                lib.raftcorr_build_bwd_deferred_workspace_bytes(B, D, H, W)
            work = torch.empty(wbytes, dtype=torch.uint8, device=fmap1.device)
            with _guard(fmap1):
                recs = _transpose_pending(pyr, recs, fmap1)
                cptrs = _lib.ptr_array([r[2] for r in recs])
                rptrs = _lib.ptr_array([r[3] for r in recs])
                _lib.check(lib.raftcorr_build_bwd_deferred(fmap1.data_ptr(), fmap2.data_ptr(), B, D, H, W, pyr.L, radius, mode,
                                                           len(recs), cptrs, rptrs, pyr.dptrs if dense else None,
                                                           d1.data_ptr(), d2.data_ptr(), work.data_ptr(), wbytes,
                                                           _stream(fmap1)), "build_bwd_deferred")
            pyr.release_grad()
            return d1, d2, None, None
        wbytes = lib.raftcorr_build_bwd_workspace_bytes(B, D, H, W)
        work = torch.empty(wbytes, dtype=torch.uint8, device=fmap1.device)
        with _guard(fmap1):
            _lib.check(lib.raftcorr_build_bwd(fmap1.data_ptr(), fmap2.data_ptr(), B, D, H, W, pyr.L,
                                              pyr.dptrs, d1.data_ptr(), d2.data_ptr(), work.data_ptr(),
                                              wbytes, _stream(fmap1)), "build_bwd")
        pyr.release_grad()                          # level 0 was clobbered by the fold
        return d1, d2, None, None


def _lookup(pyr, coords, radius, mode):
    # the per-iteration call: raftcorr_lookup_fwd through the CPython glue when it is built, else through
    # the argtypes-free ctypes handle with prebuilt arguments (_lib.fast) -- host cost only, same kernels
    out = coords.new_empty(pyr.out_shape(radius, mode))      # coords is float32 on the volume's device (checked by the caller)
    with _guard(coords):
        if pyr.vol:                                              # 16-bit volume: same call with the element type
            if _pyfast is not None:
                rc = _pyfast.lookup_fwd_vol(_lib.fn_addr("raftcorr_lookup_fwd_vol"), pyr.ptrs_addr, pyr.vol, pyr.B, pyr.H,
                                            pyr.W, pyr.L, radius, coords.data_ptr(), mode, out.data_ptr(), _stream(coords))
            else:
                rc = _lib.load().raftcorr_lookup_fwd_vol(pyr.ptrs, pyr.vol, pyr.B, pyr.H, pyr.W, pyr.L, radius,
                                                         coords.data_ptr(), mode, out.data_ptr(), _stream(coords))
        elif _pyfast is not None:
            rc = _pyfast.lookup_fwd(_lib.fn_addr("raftcorr_lookup_fwd"), pyr.ptrs_addr, pyr.B, pyr.H, pyr.W,
                                    pyr.L, radius, coords.data_ptr(), mode, out.data_ptr(), _stream(coords))
        else:
            cB, cH, cW, cL = pyr.c_dims
            rc = _lib.fast("raftcorr_lookup_fwd")(
                pyr.ptrs, cB, cH, cW, cL, pyr.c_radius(radius), ctypes.c_void_p(coords.data_ptr()),
                _lib.C_MODE[mode], ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(_stream(coords)))
    if rc:
        _lib.check(rc, "lookup_fwd")
    return out


class _Lookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, coords, pyr, radius, mode):
        ctx.pyr, ctx.radius, ctx.mode = pyr, radius, mode
        ctx.save_for_backward(coords)
        return _lookup(pyr, coords, radius, mode)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (coords,) = ctx.saved_tensors
        pyr = ctx.pyr
        lib = _lib.load()
        need_vol = ctx.needs_input_grad[0]
        need_co = ctx.needs_input_grad[1]
        if need_vol and not need_co and ctx.mode != _lib.MODE_FORK and pyr.deferred:
            rows = _channels_last_rows(dout)
            if rows is not None and pyr.records_take(ctx.mode):
                _record(pyr, ctx.mode, ctx.radius, coords, None, rows)
                return None, None, None, None, None
        dout = _f32c(dout)
        if need_co and pyr.vol:
            raise RuntimeError("CorrBlock: the gradient w.r.t. coords needs a float32 volume (volume_dtype='fp32'); "
                               "detach coords as core/raft.py:124 does")
        dcoords = torch.empty_like(coords) if need_co else None
        with _guard(coords):
            dptrs = None
            if need_vol and pyr.deferred and pyr.records_take(ctx.mode):
                _record(pyr, ctx.mode, ctx.radius, coords, dout)
            elif need_vol:
                pyr.grad_buffer()
                dptrs = pyr.dptrs
            if dptrs is not None or need_co:
                _lib.check(lib.raftcorr_lookup_bwd(pyr.ptrs, dptrs, pyr.B, pyr.H, pyr.W, pyr.L, ctx.radius,
                                                   coords.data_ptr(), ctx.mode, dout.data_ptr(),
                                                   dcoords.data_ptr() if need_co else None,
                                                   _stream(coords)), "lookup_bwd")
        # no gradient tensor for the anchor: the build node runs once every lookup of the pass has, whatever they return
        return None, dcoords, None, None, None


class _LookupConv(torch.autograd.Function):
    """Fused lookup + 1x1 convolution (+ bias, + ReLU), differentiable.  Forward: one launch
    (raftcorr_lookup_conv1x1_fwd), the (B, L*rd*rd, H, W) tensor is never stored.  Backward re-creates it with
    raftcorr_lookup_fwd (a 7 us kernel instead of a 9 MB tensor kept alive per iteration), takes d(weight) and
    d(bias) from it and sends W^T d(pre) through raftcorr_lookup_bwd into the block's shared gradient pyramid,
    exactly where an unfused lookup's gradient goes (core/update.py:83,90 on the training path)."""

    @staticmethod
    def forward(ctx, anchor, coords, weight, bias, blk, relu):
        out = blk._lookup_conv1x1_nograd(coords, weight, bias, relu)
        ctx.blk, ctx.relu = blk, relu
        ctx.save_for_backward(coords, weight, out if relu else None)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        coords, weight, out = ctx.saved_tensors
        blk, pyr = ctx.blk, ctx.blk._pyr
        lib = _lib.load()
        dpre = _f32c(dout)
        if ctx.relu:
            dpre = dpre * (out > 0)
        w2 = weight.reshape(weight.shape[0], -1).float()
        B, Cout = dpre.shape[:2]
        dweight = dbias = None
        if ctx.needs_input_grad[2]:
            x = _lookup(pyr, coords, blk.radius, _lib.MODE_CANONICAL)            # recomputed, (B, Cin, H, W)
            dweight = torch.einsum("bon,bkn->ok", dpre.reshape(B, Cout, -1), x.reshape(B, x.shape[1], -1)).reshape(weight.shape)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            dbias = dpre.sum(dim=(0, 2, 3))
        if ctx.needs_input_grad[0] and pyr.deferred:
            # W^T d(pre) straight in the record's layout (B, N, Cin): no transposition pass
            rec = torch.einsum("ok,bon->bnk", w2, dpre.reshape(B, Cout, -1)).contiguous()
            pyr.add_record(_lib.MODE_CANONICAL, blk.radius, coords, rec)
        elif ctx.needs_input_grad[0]:
            dx = torch.einsum("ok,bon->bkn", w2, dpre.reshape(B, Cout, -1)).reshape(B, -1, pyr.H, pyr.W).contiguous()
            with _guard(coords):
                pyr.grad_buffer()
                _lib.check(lib.raftcorr_lookup_bwd(pyr.ptrs, pyr.dptrs, pyr.B, pyr.H, pyr.W, pyr.L, blk.radius,
                                                   coords.data_ptr(), _lib.MODE_CANONICAL, dx.data_ptr(), None,
                                                   _stream(coords)), "lookup_bwd")
        return None, None, dweight, dbias, None, None


class CorrBlock:
    """All-pairs correlation volume + pooled pyramid + radius-r bilinear lookup.
    Drop-in for core/corr.py:12-61 (same constructor defaults: num_levels=2, radius=4).

    precision (keyword, not in the reference) selects the arithmetic of the forward contraction; the
    pyramid and the lookup are fp32 in every case, and the backward contractions run fp32-grade
    (range-scaled operands split into two fp16 pieces, three products, fp32 accumulation; exact fp32
    MFMA when the operands are not 16-byte aligned) with a fixed summation order -- gradients are reproducible
    bit for bit:
      "fp32"    (default) exact fp32 FMA chain on the fp32 matrix cores;
      "bf16x3"  operands split into three bfloat16 pieces, six bf16 MFMA products: fp32-grade
                results (inside the 1e-5 parity bar by orders of magnitude), faster build;
      "fp16x2"  maps scaled by a power of two, operands split into two fp16 pieces, three fp16
                MFMA products: fp32-grade too, the fastest of the accurate builds;
      "bf16"    operands rounded to bfloat16: differs by the input rounding, <= 1e-2.

    volume_dtype (keyword, not in the reference): element type the pyramid is STORED in.  "fp32" (default) is
    the reference's; "bf16" / "fp16" store every level in 16 bits (rounded once from the fp32 accumulators):
    half the memory (1080p: 2.8 instead of 5.7 GB), half the build's store traffic and half the line traffic
    of every lookup, at the price of the element rounding (bf16 2^-9, fp16 2^-12 relative; fp16 overflows
    beyond +-65504) -- the <= 1e-2 class of the north star's bf16 option.  Lookups return float32 either way;
    gradients w.r.t. the features work as usual, the gradient w.r.t. coords needs "fp32".

    backward (keyword, not in the reference): how the lookups' gradients reach the features.  "deferred": every
    lookup's backward pass only records (coords, d(out)); when autograd reaches the build, one kernel makes each
    query's folded level-0 gradient row in LDS from all records (raftcorr_build_bwd_deferred) -- nothing dense is
    zero-filled or read-modified-written per lookup.  "accumulate": the lookups add into one dense gradient pyramid
    (raftcorr_lookup_bwd) that is folded once (raftcorr_build_bwd); needed for more than 4 levels or maps beyond
    ~190 x 190.  "auto" (default): deferred where the shape allows.  Both are bit-reproducible."""

    def __init__(self, fmap1, fmap2, num_levels=2, radius=4, precision="fp32", volume_dtype="fp32", backward="auto"):
        self.num_levels = num_levels
        self.radius = radius
        if backward not in ("auto", "deferred", "accumulate"):
            raise RuntimeError("CorrBlock: backward must be 'auto', 'deferred' or 'accumulate'")
        if precision not in _lib.PRECISIONS:
            raise RuntimeError(f"CorrBlock: precision must be one of {sorted(_lib.PRECISIONS)}; got {precision!r}")
        if volume_dtype not in _lib.VOLUMES:
            raise RuntimeError(f"CorrBlock: volume_dtype must be one of {sorted(_lib.VOLUMES)}; got {volume_dtype!r}")
        if volume_dtype != "fp32" and num_levels > 4:
            raise RuntimeError("CorrBlock: a 16-bit volume has at most 4 levels")
        self.precision = precision
        self.volume_dtype = volume_dtype
        _require_cuda(fmap1=fmap1, fmap2=fmap2)
        if fmap1.dim() != 4 or fmap1.shape != fmap2.shape:
            raise RuntimeError(f"CorrBlock: fmap1/fmap2 must both be (B, D, H, W); got "
                               f"{tuple(fmap1.shape)} and {tuple(fmap2.shape)}")
        if not 1 <= num_levels <= _lib.MAX_LEVELS:
            raise RuntimeError(f"CorrBlock: num_levels must be in [1, {_lib.MAX_LEVELS}]")
        if radius < 0:
            raise RuntimeError("CorrBlock: radius must be >= 0")
        fmap1, fmap2 = _f32c(fmap1), _f32c(fmap2)
        B, D, H, W = fmap1.shape
        if (H >> (num_levels - 1)) < 1 or (W >> (num_levels - 1)) < 1:
            # the reference fails in F.avg_pool2d ("Output size is too small"), core/corr.py:26
            raise RuntimeError(f"CorrBlock: a {H} x {W} map has no level {num_levels - 1}: output size is too small")
        self._pyr = _Pyramid(B, H, W, num_levels, fmap1.device, volume_dtype)
        if backward != "accumulate" and num_levels <= 4 and radius <= 7:
            self._pyr.set_deferred(radius)
        if backward == "deferred" and not self._pyr.deferred:
            raise RuntimeError(f"CorrBlock: the deferred backward does not take a {H} x {W} map with {num_levels} levels, "
                               f"radius {radius}")
        if torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad):
            flat, self._anchor = _BuildPyramid.apply(fmap1, fmap2, self._pyr, precision)
            self._pyr.flat = flat.detach()
        else:
            flat = _build(fmap1, fmap2, self._pyr, precision)
            self._anchor = None
        self._flat = flat            # packed levels (autograd-tracked when the features are)
        self._logical = None
        self._packed = None          # lookup_conv1x1: (key, packed weight, weight)

    @property
    def corr_pyramid(self):
        """List of (B*H*W, 1, H_l, W_l) tensors, as core/corr.py:16-27 exposes it.  The kernels
        keep the levels in patch-tile order; this view is gathered on first access only."""
        if self._logical is None:
            self._logical = self._pyr.logical(self._flat)
        return self._logical

    def __call__(self, coords):
        p = self._pyr
        if not coords.is_cuda:
            raise RuntimeError("coords must be a CUDA tensor")
        if coords.device != p.device:
            raise RuntimeError(f"coords is on {coords.device}, the correlation volume on {p.device}")
        nd = coords.dim()
        if nd == 3:                                             # fork semantics, core/corr.py:29-51
            mode = _lib.MODE_FORK
            if coords.shape != (p.B, p.N, 2):
                raise RuntimeError(f"CorrBlock: coords must be ({p.B}, {p.N}, 2); got {tuple(coords.shape)}")
        elif nd == 4:                                           # canonical RAFT, core/raft.py:125
            mode = _lib.MODE_CANONICAL
            if coords.shape != (p.B, 2, p.H, p.W):
                raise RuntimeError(f"CorrBlock: coords must be ({p.B}, 2, {p.H}, {p.W}); got {tuple(coords.shape)}")
        else:
            raise RuntimeError("CorrBlock: coords must be (B, N, 2) or (B, 2, H, W)")
        if coords.dtype is not torch.float32 or not coords.is_contiguous():
            coords = _f32c(coords)
        if not (torch.is_grad_enabled() and (self._anchor is not None or coords.requires_grad)):
            return _lookup(p, coords, self.radius, mode)
        anchor = self._anchor if self._anchor is not None else \
            torch.zeros(1, dtype=torch.float32, device=coords.device)
        return _Lookup.apply(anchor, coords, p, self.radius, mode)

    def lookup_conv1x1(self, coords, weight, bias=None, relu=True):
        """Canonical lookup fused with a 1x1 convolution (+ bias, + ReLU) over its channels: what
        RAFT's motion encoder does first with the result, `F.relu(self.convc1(corr))`
        (core/update.py:83,90).  coords (B, 2, H, W); weight (Cout, L*rd*rd[, 1, 1]); returns
        (B, Cout, H, W) without materialising the (B, L*rd*rd, H, W) tensor.  Differentiable w.r.t. the
        features (through the block's shared gradient pyramid), the weight and the bias; coords are treated as
        detached (core/raft.py:124).  The packed weight is cached on this CorrBlock per weight tensor version."""
        if torch.is_grad_enabled() and (self._anchor is not None or weight.requires_grad or
                                        (bias is not None and bias.requires_grad)):
            anchor = self._anchor if self._anchor is not None else \
                torch.zeros(1, dtype=torch.float32, device=coords.device)
            return _LookupConv.apply(anchor, _f32c(coords.detach()), weight, bias, self, relu)
        return self._lookup_conv1x1_nograd(coords, weight, bias, relu)

    def _lookup_conv1x1_nograd(self, coords, weight, bias=None, relu=True):
        p = self._pyr
        _require_cuda(coords=coords, weight=weight, **({} if bias is None else {"bias": bias}))
        if coords.device != p.device:
            raise RuntimeError(f"coords is on {coords.device}, the correlation volume on {p.device}")
        if tuple(coords.shape) != (p.B, 2, p.H, p.W):
            raise RuntimeError(f"CorrBlock: coords must be ({p.B}, 2, {p.H}, {p.W}); got {tuple(coords.shape)}")
        cin = self.num_levels * (2 * self.radius + 1) ** 2
        if weight.dim() == 4 and weight.shape[2:] == (1, 1):
            weight = weight[:, :, 0, 0]
        if weight.dim() != 2 or weight.shape[1] != cin:
            raise RuntimeError(f"CorrBlock: weight must be (Cout, {cin}[, 1, 1]); got {tuple(weight.shape)}")
        if p.vol:
            raise RuntimeError("CorrBlock.lookup_conv1x1 needs a float32 volume (volume_dtype='fp32')")
        if self.num_levels > 4 or self.radius not in (3, 4):
            raise RuntimeError("CorrBlock.lookup_conv1x1 supports num_levels <= 4 and radius 3 or 4")
        cout = weight.shape[0]
        if bias is not None and tuple(bias.shape) != (cout,):
            raise RuntimeError(f"CorrBlock: bias must be ({cout},)")
        lib = _lib.load()
        # packed once per CorrBlock (i.e. per frame pair: 3 us) and weight version; the cache holds the
        # weight tensor itself, so its address cannot be reused by another tensor meanwhile
        key = (weight.data_ptr(), weight._version)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, _pack_conv_weight(lib, weight.detach(), cout, cin), weight)
        packed = self._packed[1]
        coords = _f32c(coords.detach())
        b32 = _f32c(bias.detach()) if bias is not None else None
        out = torch.empty(p.B, cout, p.H, p.W, dtype=torch.float32, device=coords.device)
        with _guard(coords):
            _lib.check(lib.raftcorr_lookup_conv1x1_fwd(p.ptrs, p.B, p.H, p.W, p.L, self.radius, coords.data_ptr(),
                                                       packed.data_ptr(), b32.data_ptr() if b32 is not None else None,
                                                       cout, 1 if relu else 0, out.data_ptr(), _stream(coords)),
                       "lookup_conv1x1_fwd")
        return out

    @staticmethod
    def corr(fmap1, fmap2):
        """core/corr.py:53-61: (B, D, H, W) x2 -> (B, H, W, 1, H, W), already divided by sqrt(D)."""
        B, D, H, W = fmap1.shape
        blk = CorrBlock(fmap1, fmap2, num_levels=1, radius=0)
        return blk.corr_pyramid[0].view(B, H, W, 1, H, W)


def _pack_conv_weight(lib, weight, cout, cin):
    w = _f32c(weight)
    nbytes = lib.raftcorr_conv1x1_packed_bytes(cout, cin)
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=weight.device)
    with _guard(w):
        _lib.check(lib.raftcorr_conv1x1_pack_weight(w.data_ptr(), cout, cin, packed.data_ptr(), nbytes, _stream(w)),
                   "conv1x1_pack_weight")
    return packed


def _alt_forward(blk, coords, keep_groups=False):
    """raftcorr_alt_lookup_fwd on the block's packed features (all levels, one launch).  Window mode builds the
    lookup's query groups in front; keep_groups: into a buffer of its own, returned for the backward pass (else into
    the block's reusable one)."""
    lib = _lib.load()
    B, C, H, W = blk._shape
    L = blk.num_levels
    out = torch.empty(B, L * (2 * blk.radius + 1) ** 2, H, W, dtype=torch.float32, device=coords.device)
    groups = None
    with _guard(coords):
        if blk.tiles == "window":
            if keep_groups or blk._sort_work is None:
                nbytes = lib.raftcorr_alt_lookup_fwd_sorted_workspace_bytes(B, H, W, L)
                groups = torch.empty(nbytes, dtype=torch.uint8, device=coords.device)
                if not keep_groups:
                    blk._sort_work = groups
            if not keep_groups:
                groups = blk._sort_work
            _lib.check(lib.raftcorr_alt_lookup_fwd_sorted(blk._p1.data_ptr(), blk._p2ptrs, blk._scales.data_ptr(), B, H, W, C, L,
                                                          blk.radius, coords.data_ptr(), out.data_ptr(),
                                                          groups.data_ptr(), groups.numel(), _stream(coords)),
                       "alt_lookup_fwd_sorted")
        else:
            _lib.check(lib.raftcorr_alt_lookup_fwd(blk._p1.data_ptr(), blk._p2ptrs, blk._scales.data_ptr(), B, H, W, C, L,
                                                   blk.radius, coords.data_ptr(), out.data_ptr(), _stream(coords)),
                       "alt_lookup_fwd")
    return (out, groups) if keep_groups else out


class _AltGradState:
    """Gradient sums of one AlternateCorrBlock (NHWC: d fmap1, d fmap2 level 0, ..).  A separate object so that the
    anchor's autograd node does not hold the block (which holds the anchor)."""

    def __init__(self, shape, num_levels, device):
        self.shape, self.L, self.device = shape, num_levels, device
        self.acc = self.ptrs = self.task = None
        self.records, self.rtask = [], None                     # deferred backward: (coords, d(out), groups) per lookup

    def record(self, coords, dout, groups):
        """Deferred backward pass: keep what a lookup's backward call was given; _AltFeatures.backward, which runs after
        all of them, does the work in one library call.  Records of an earlier pass that never reached the features are
        dropped."""
        task = _graph_task_id()
        if self.rtask != task:
            self.records, self.rtask = [], task
        self.records.append((coords, dout, groups))

    def take_records(self):
        rec = self.records if self.rtask == _graph_task_id() else []
        self.records, self.rtask = [], None
        return rec

    def buffers(self):
        """(sums, first) for the backward pass that is running now; `first`: nothing of this pass is in them yet (the
        kernel then overwrites instead of adding -- no zero fill).  What an earlier pass left behind without reaching
        the features is dropped (cf. _Pyramid.grad_buffer)."""
        task = _graph_task_id()
        first = self.acc is None or self.task != task
        if self.acc is None:
            B, C, H, W = self.shape
            self.acc = [torch.empty(B, H, W, C, dtype=torch.float32, device=self.device)] + \
                       [torch.empty(B, H >> l, W >> l, C, dtype=torch.float32, device=self.device) for l in range(self.L)]
            self.ptrs = _lib.ptr_array(self.acc[1:])
        self.task = task
        return self.acc, first

    def take(self):
        acc = self.acc if self.acc is not None and self.task == _graph_task_id() else None
        self.acc = self.ptrs = self.task = None                 # handed to autograd: the next pass gets new buffers
        return acc


def _alt_backward_many(st, rec):
    """raftcorr_alt_lookup_bwd_many over the recorded lookups (in the order their backward calls arrived), at most
    ALT_MAX_LOOKUPS per call; sums into the state's buffers like the per-lookup calls do."""
    lib = _lib.load()
    blk = st.blk()
    if blk is None:
        raise RuntimeError("AlternateCorrBlock: the block was released before its backward pass ran")
    B, C, H, W = st.shape
    L, window = st.L, blk.tiles == "window"
    f1 = blk._p1 if window else blk._t1
    # lookups per call: as many as the library takes, fewer for very large maps (32-bit offsets inside the kernels'
    # workspace: items x levels x group slots x window cells x 64 < 2^31; and a workspace of at most 8 GiB)
    rd2 = (2 * blk.radius + 1) ** 2
    slots = ((H * W + 63) // 64 + ((W + 7) // 8 + 2) * ((H + 7) // 8 + 2)) if window else ((W + 7) // 8) * ((H + 7) // 8)
    per_call = max(1, min(_lib.ALT_MAX_LOOKUPS, (2 ** 31 - 1) // max(B * L * slots * rd2 * 64, B * slots * ((C + 31) // 32 * 32) * 16)))
    while per_call > 1 and lib.raftcorr_alt_lookup_bwd_many_workspace_bytes(B, H, W, C, L, blk.radius, per_call,
                                                                             1 if window else 0) > (8 << 30):
        per_call //= 2
    for i in range(0, len(rec), per_call):
        part = rec[i:i + per_call]
        acc, first = st.buffers()
        n = len(part)
        nbytes = lib.raftcorr_alt_lookup_bwd_many_workspace_bytes(B, H, W, C, L, blk.radius, n, 1 if window else 0)
        work = torch.empty(nbytes, dtype=torch.uint8, device=st.device)
        coords, douts = [p[0] for p in part], [p[1] for p in part]
        with _guard(coords[0]):
            _lib.check(lib.raftcorr_alt_lookup_bwd_many(f1.data_ptr(), blk._t2ptrs, blk._scales.data_ptr(), B, H, W, C, L,
                                                        blk.radius, n, _lib.ptr_array(coords), _lib.ptr_array(douts),
                                                        _lib.ptr_array([p[2] for p in part]) if window else None,
                                                        1 if window else 0, acc[0].data_ptr(), st.ptrs, 0 if first else 1,
                                                        work.data_ptr(), nbytes, _stream(coords[0])), "alt_lookup_bwd_many")


class _AltFeatures(torch.autograd.Function):
    """fmap1, fmap2 -> anchor.  Like _BuildPyramid's anchor: a 1-element tensor that every lookup of the block takes
    as an input.  The lookups' backward passes add their feature gradients into the block's sums in place
    (raftcorr_alt_lookup_bwd with accumulate = 1: no dense gradient per lookup handed to autograd, no separate
    additions) and this node, which runs after all of them, turns the sums into the gradients of the two maps
    (raftcorr_alt_features_bwd: NHWC -> NCHW, pooled levels folded through the avg_pool2d chain of corr.py:69-73)."""

    @staticmethod
    def forward(ctx, state, fmap1, fmap2):
        ctx.state = state
        ctx.set_materialize_grads(False)
        return torch.empty(1, dtype=torch.float32, device=fmap1.device)    # (a graph edge; nobody reads the value)

    @staticmethod
    @once_differentiable
    def backward(ctx, danchor):
        st = ctx.state
        rec = st.take_records()
        if rec:                                                 # deferred: all lookups of the pass in one library call
            _alt_backward_many(st, rec)
        acc = st.take()
        if acc is None:                                         # no lookup of this backward pass reached the features
            return None, None, None
        B, C, H, W = st.shape
        d1 = torch.empty(B, C, H, W, dtype=torch.float32, device=st.device)
        d2 = torch.empty(B, C, H, W, dtype=torch.float32, device=st.device)
        with _guard(d1):
            _lib.check(_lib.load().raftcorr_alt_features_bwd(acc[0].data_ptr(), _lib.ptr_array(acc[1:]), B, C, H, W, st.L,
                                                             d1.data_ptr(), d2.data_ptr(), _stream(d1)), "alt_features_bwd")
        return None, d1, d2


class _AltLookup(torch.autograd.Function):
    """All levels of the on-the-fly correlation in one launch (forward) and its gradient w.r.t. the features
    (raftcorr_alt_lookup_bwd: tile-major packed features, no atomics, bit-reproducible).  The kernels read the
    block's packed copies; the graph edge to the features is the block's anchor (_AltFeatures)."""

    @staticmethod
    def forward(ctx, anchor, coords, blk):
        ctx.blk = blk
        ctx.save_for_backward(coords)
        # window mode: the lookup's query groups (a few hundred KB) stay alive for its backward pass
        out, ctx.groups = _alt_forward(blk, coords, keep_groups=True)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        lib = _lib.load()
        (coords,) = ctx.saved_tensors
        blk = ctx.blk
        B, C, H, W = blk._shape
        L = blk.num_levels
        dout = _f32c(dout)
        if blk.backward == "deferred":                          # the work is done once, by the features' node
            blk._gstate.record(coords, dout, ctx.groups)
            ctx.groups = None
            return None, None, None
        acc, first = blk._gstate.buffers()
        work = blk._bwd_workspace()
        with _guard(coords):
            if blk.tiles == "window":                            # query groups by window position (deterministic lists)
                _lib.check(lib.raftcorr_alt_lookup_bwd_sorted(blk._p1.data_ptr(), blk._t2ptrs, blk._scales.data_ptr(), B, H, W,
                                                              C, L, blk.radius, coords.data_ptr(), dout.data_ptr(),
                                                              acc[0].data_ptr(), blk._gstate.ptrs, 0 if first else 1,
                                                              ctx.groups.data_ptr() if ctx.groups is not None else None,
                                                              work.data_ptr(), work.numel(), _stream(coords)),
                           "alt_lookup_bwd_sorted")
                ctx.groups = None
            else:
                _lib.check(lib.raftcorr_alt_lookup_bwd(blk._t1.data_ptr(), blk._t2ptrs, blk._scales.data_ptr(), B, H, W, C, L,
                                                       blk.radius, coords.data_ptr(), dout.data_ptr(),
                                                       acc[0].data_ptr(), blk._gstate.ptrs, 0 if first else 1,
                                                       work.data_ptr(), work.numel(), _stream(coords)), "alt_lookup_bwd")
        # coords gradient: the reference kernel never writes it (correlation_kernel.cu:307)
        # (no gradient tensor for the anchor: _AltFeatures.backward runs once every lookup of the pass has)
        return None, None, None


class AlternateCorrBlock:
    """Memory-lean on-the-fly correlation.  Drop-in for core/corr.py:64-92 (defaults
    num_levels=4, radius=4); unlike the reference it is differentiable w.r.t. the features.

    The constructor packs the features once per frame pair for the 16-bit matrix cores (two fp16
    pieces per value, fp32-grade products: include/raftcorr.h); every call is then ONE launch for all
    levels.  float32 NHWC copies (what the backward kernels read) are made only when a gradient can
    flow to the features.

    tiles: how the forward kernel groups 64 queries into one matrix-core tile.  "pixel" (default): 8x8 pixel tiles, the
    right choice for flow fields (neighbours look at neighbouring windows).  "window": queries are dealt into tiles by
    the position of their windows in a small extra launch per call (raftcorr_alt_lookup_fwd_sorted) -- the cost no
    longer depends on how coherent the flow is (per-pixel random offsets of a few pixels cost pixel tiles 2-4x);
    about 10 us per call more when the flow is smooth.  Same results up to the last bit of a sum; the groups are a
    function of the coordinates alone, so both modes are bit-reproducible from run to run.

    backward: "deferred" (default): a lookup's backward call only keeps its (coords, d(out), groups); the node of the
    features, which autograd runs after all lookups of the pass, computes the feature gradients of all of them in one
    library call (raftcorr_alt_lookup_bwd_many: the kernels run over lookups x batch items at once, which fills the
    chip where one lookup of one frame pair does not).  "per_lookup": every lookup's backward call runs the kernels
    itself and adds into the block's sums in place.  Same bits either way (the same additions in the same order)."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, tiles="pixel", backward="deferred"):
        self.num_levels = num_levels
        self.radius = radius
        if tiles not in ("pixel", "window"):
            raise RuntimeError("AlternateCorrBlock: tiles must be 'pixel' or 'window'")
        if backward not in ("deferred", "per_lookup"):
            raise RuntimeError("AlternateCorrBlock: backward must be 'deferred' or 'per_lookup'")
        self.tiles = tiles
        self.backward = backward
        self._sort_work = None
        _require_cuda(fmap1=fmap1, fmap2=fmap2)
        if fmap1.dim() != 4 or fmap1.shape != fmap2.shape:
            raise RuntimeError(f"AlternateCorrBlock: fmap1/fmap2 must both be (B, D, H, W); got "
                               f"{tuple(fmap1.shape)} and {tuple(fmap2.shape)}")
        if not 1 <= num_levels <= _lib.MAX_LEVELS:
            raise RuntimeError(f"AlternateCorrBlock: num_levels must be in [1, {_lib.MAX_LEVELS}]")
        if radius < 0:
            raise RuntimeError("AlternateCorrBlock: radius must be >= 0")
        if radius > 7:
            raise RuntimeError("AlternateCorrBlock: radius > 7 is not built")
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        B, C, H, W = fmap1.shape
        if (H >> (num_levels - 1)) < 1 or (W >> (num_levels - 1)) < 1:
            raise RuntimeError(f"AlternateCorrBlock: a {H} x {W} map has no level {num_levels - 1}: output size is too small")
        self._fmaps = (fmap1, fmap2)
        self._pyramid = None                                    # core/corr.py:69-73, built when somebody asks for it
        self._shape = (B, C, H, W)
        lib = _lib.load()
        f1, f2 = fmap1.detach().contiguous(), fmap2.detach().contiguous()
        dev = f1.device
        self._grad = torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad)
        # one buffer for the ranges, every packed copy (fmap1 and the L pyramid levels of fmap2; tile-major copies for the
        # backward pass only when a gradient can flow to the features) and the set-up's workspace; one library call
        dims = [(H, W)] + [(H >> l, W >> l) for l in range(num_levels)]
        sizes = [1024] + [lib.raftcorr_alt_packed_bytes(B, C, h, w) for h, w in dims]
        if self._grad:
            sizes += [lib.raftcorr_alt_tiles_bytes(B, C, h, w) if i or tiles == "pixel" else 0 for i, (h, w) in enumerate(dims)]
        sizes.append(lib.raftcorr_alt_prepare_workspace_bytes(B, C, H, W, num_levels))
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + (n + 255) // 256 * 256)
        self._store = torch.empty(offs[-1], dtype=torch.uint8, device=dev)
        parts = [self._store[o:o + n] for o, n in zip(offs, sizes)]
        n = 1 + num_levels
        self._scales = parts[0].view(torch.float32)             # partial maxima of |fmap1|, |fmap2|
        self._p1, self._p2 = parts[1], parts[2:1 + n]
        self._p2ptrs = _lib.ptr_array(self._p2)
        self._t1 = self._t2 = self._t2ptrs = self._anchor = None    # tile-major packed copies, for the backward pass only
        if self._grad:
            self._t1, self._t2 = parts[1 + n], parts[2 + n:1 + 2 * n]
            self._t2ptrs = _lib.ptr_array(self._t2)
            if tiles != "pixel":                                # the window-group backward gathers fmap1 from _p1 per call
                self._t1 = None
        work = parts[-1]
        with _guard(f1):
            _lib.check(lib.raftcorr_alt_prepare(f1.data_ptr(), f2.data_ptr(), B, C, H, W, num_levels, self._scales.data_ptr(),
                                                self._p1.data_ptr(), self._p2ptrs,
                                                self._t1.data_ptr() if self._t1 is not None else None, self._t2ptrs,
                                                work.data_ptr(), work.numel(), _stream(f1)), "alt_prepare")
        if self._grad:
            self._gstate = _AltGradState(self._shape, num_levels, dev)
            self._gstate.blk = weakref.ref(self)
            self._anchor = _AltFeatures.apply(self._gstate, fmap1, fmap2)
        self._work = None                                       # workspace of the backward kernels, shared by the lookups

    @property
    def pyramid(self):
        """core/corr.py:69-73: [(fmap1, fmap2)] + num_levels pooled pairs (differentiable stock ops).  The kernels do not
        read it -- raftcorr_alt_prepare pools fmap2 itself, with the same arithmetic -- so it is built on demand."""
        if self._pyramid is None:
            fmap1, fmap2 = self._fmaps
            self._pyramid = [(fmap1, fmap2)]
            for _ in range(self.num_levels):
                fmap1 = F.avg_pool2d(fmap1, 2, stride=2) if min(fmap1.shape[-2:]) >= 2 else fmap1[..., :0, :0]
                fmap2 = F.avg_pool2d(fmap2, 2, stride=2) if min(fmap2.shape[-2:]) >= 2 else fmap2[..., :0, :0]
                self._pyramid.append((fmap1, fmap2))
        return self._pyramid

    def _bwd_workspace(self):
        if self._work is None:
            B, C, H, W = self._shape
            fn = _lib.load().raftcorr_alt_lookup_bwd_sorted_workspace_bytes if self.tiles == "window" else \
                _lib.load().raftcorr_alt_lookup_bwd_workspace_bytes
            nbytes = fn(B, H, W, C, self.num_levels, self.radius)
            self._work = torch.empty(nbytes, dtype=torch.uint8, device=self._p1.device)
        return self._work

    def __call__(self, coords):
        _require_cuda(coords=coords)
        if coords.device != self._p1.device:
            raise RuntimeError(f"coords is on {coords.device}, the feature maps on {self._p1.device}")
        B, C, H, W = self._shape
        if coords.shape != (B, 2, H, W):
            raise RuntimeError(f"AlternateCorrBlock: coords must be ({B}, 2, {H}, {W}); got {tuple(coords.shape)}")
        coords = _f32c(coords).detach()
        if self._grad and torch.is_grad_enabled():
            return _AltLookup.apply(self._anchor, coords, self)
        return _alt_forward(self, coords)
