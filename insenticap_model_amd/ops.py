"""Thin torch-tensor -> C-ABI adapters. PyTorch is plumbing here (device memory + streams);
all arithmetic happens in libinsenticap_hip.so."""
import ctypes as C
import gc
import threading
import time

import torch

from . import _lib
from ._lib import LinearProblem, LstmProblem, RolloutStep, ScanProblem, check


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the stream the kernels run on
    (torch's current stream). bench.py arms it for one decode step per timed roll-out; when
    `armed` is False the wrappers below add no work."""

    def __init__(self):
        self.armed = False
        self.arm_step = None   # decode step whose kernels get timed (-1 = prologue, None = off)
        self.phase = 'step'    # 'prologue' kernels run once per roll-out, 'step' kernels T times
        self.records = []   # (name, start_event, end_event, flops, bytes)

    def begin(self):
        if not self.armed:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._h3_before = _lib.load().isc_h3_launches()
        return e

    def end(self, e0, name, flops=0.0, nbytes=0.0):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        h3 = _lib.load().isc_h3_launches() != self._h3_before     # the op went out on the split-f16 GEMM path
        self.records.append((name, e0, e1, flops, nbytes, self.phase, h3))

    def summary(self):
        """name -> dict(n, avg_ms, flops, bytes); call after a device synchronize."""
        out = {}
        for name, e0, e1, fl, nb, ph, h3 in self.records:
            d = out.setdefault(name, dict(n=0, total_ms=0.0, flops=fl, bytes=nb, phase=ph, h3=h3))
            d['n'] += 1
            d['total_ms'] += e0.elapsed_time(e1)
        for d in out.values():
            d['avg_ms'] = d['total_ms'] / d['n']
        return out


TIMER = KernelTimer()


def planes_ptrs(planes):
    """(hi, lo) pointers of a split-f16 plane buffer: a [2, rows, K] f16 tensor used as ONE buffer in which the two
    planes are interleaved per 32-wide k-block (include/insenticap_hip.h, isc_seg.A_hi): lo = hi + 32 halfs."""
    return planes.data_ptr(), planes.data_ptr() + 64


def _seg_k(segs):
    return sum(sg[0].shape[1] for sg in segs)


_SPLITK_WS = {}
_WS_SIZES = {}


def splitk_ws_floats():
    """Size of a per-(device, stream) split-K workspace, asked of the library (isc_splitk_workspace_bytes): the
    deepest K split of a [4096 x 512] output - the widest few-tile launch of the step at the headline batch - =
    128 MB; also holds the f16 weight planes of a split-f16 launch issued outside a weights scope."""
    if 'ws' not in _WS_SIZES:
        _WS_SIZES['ws'] = int(_lib.load().isc_splitk_workspace_bytes(4096, 512)) // 4
    return _WS_SIZES['ws']


def h3w_bytes():
    """Size of a weights-scope plane buffer (isc_h3_weights_workspace_bytes): hi + lo planes of up to 24 Mi weight
    values (the reference architecture at V = 10k has 22 063 379, of which ~16 M are GEMM operands) in the forward
    layout AND transposed for the backward = 192 MB."""
    if 'h3w' not in _WS_SIZES:
        _WS_SIZES['h3w'] = int(_lib.load().isc_h3_weights_workspace_bytes(24 * 1024 * 1024, 1))
    return _WS_SIZES['h3w']


def graphs_allowed_here():
    """Graph serving that is ON BY DEFAULT (beam search, small greedy roll-outs) applies on the main host thread only:
    a capture begins with a device-wide synchronisation and an allocator sweep, and two threads capturing at once is
    not something this package has exercised.  Worker threads run the same calls eagerly (same results)."""
    return threading.current_thread() is threading.main_thread()


def graph_capture(graph, **kw):
    """torch.cuda.graph(graph, **kw) for this package's captures, safe next to a torch.distributed process group
    (`settle=False`: skip the wait described below - for the second and later captures of one owner in a row).
    The backend's watchdog thread polls the completion event of every collective still on its list (hipEventQuery,
    every ~100 ms) and retires finished work only at such a pass.  On ROCm a poll of a collective issued shortly before a
    capture opened, landing INSIDE the capture window, fails with hipErrorCapturedEvent - in the watchdog thread, which
    aborts the process (RCCL, one rank: by chance at the full model size where a capture takes ~100 ms, for certain
    with a 0.4 s window - tests/_rccl_child.py keeps that case).  So, with a group alive: everything queued is finished
    and one watchdog pass is let go by before the capture opens (its list is then empty for the whole window - nothing
    this package captures issues a collective), and the capture runs in 'thread_local' error mode (calls of other
    host threads are not policed; autograd's backward thread launches into the capture in any mode).  Captures are
    once per geometry: the 0.25 s do not recur."""
    try:
        import torch.distributed as dist
        grouped = dist.is_available() and dist.is_initialized()
    except Exception:                   # a torch build without distributed
        grouped = False
    # 'thread_local' always: another host thread serving its own captioner on its own stream (include/insenticap_hip.h
    # allows that) may allocate or synchronise while this one captures; 'global' would fail ITS calls
    kw.setdefault('capture_error_mode', 'thread_local')
    settle = kw.pop('settle', True)     # False: a capture right behind another one of the same owner (nothing was issued between)
    if grouped and settle:
        # ... and no collective may still be on the watchdog's list when the capture opens: it retires finished work
        # at its next pass (every 100 ms), so finish everything and let one pass go by.  Captures are once per geometry.
        torch.cuda.synchronize()
        time.sleep(0.25)
    return _Capture(torch.cuda.graph(graph, **kw))


class _Capture:
    """torch.cuda.graph(...) with Python's automatic garbage collection held off while the capture is open.  torch
    collects once when a capture begins; a generational collection that fires DURING it can still run the destructor of
    cyclic garbage from earlier work - an older captured graph, a stream's buffers - and destroying a HIP graph (or
    freeing through the runtime) on a thread that is capturing aborts the process (seen as 'Fatal Python error: Aborted
    ... Garbage-collecting' inside a roll-out capture).  The cycle collector is switched back on after capture_end."""

    def __init__(self, inner):
        self.inner, self.was = inner, False

    def __enter__(self):
        out = self.inner.__enter__()
        self.was = gc.isenabled()
        gc.disable()
        return out

    def __exit__(self, *exc):
        try:
            return self.inner.__exit__(*exc)
        finally:
            if self.was:
                gc.enable()


def graph_node_counts(graph):
    """{'kernel', 'memcpy', 'memset', 'other'} node counts of a torch.cuda.CUDAGraph created with keep_graph=True
    (measurement / tests: the launches one replay issues).  HIP runtime calls only - nothing of this package's library."""
    hip = C.CDLL('libamdhip64.so')          # the runtime torch has already loaded
    raw = C.c_void_p(graph.raw_cuda_graph())
    n = C.c_size_t(0)
    if hip.hipGraphGetNodes(raw, None, C.byref(n)) != 0:
        raise RuntimeError('hipGraphGetNodes failed')
    nodes = (C.c_void_p * max(n.value, 1))()
    if hip.hipGraphGetNodes(raw, nodes, C.byref(n)) != 0:
        raise RuntimeError('hipGraphGetNodes failed')
    out = {'kernel': 0, 'memcpy': 0, 'memset': 0, 'other': 0}
    names = {0: 'kernel', 1: 'memcpy', 2: 'memset'}          # hipGraphNodeTypeKernel / Memcpy / Memset
    for i in range(n.value):
        ty = C.c_int(-1)
        if hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(ty)) != 0:
            raise RuntimeError('hipGraphNodeGetType failed')
        out[names.get(ty.value, 'other')] += 1
    return out


def graph_kernel_nodes(graph):
    """Launch-issuing nodes (kernels, copies, fills) of a kept graph - what a kernel trace counts per replay."""
    c = graph_node_counts(graph)
    return c['kernel'] + c['memcpy'] + c['memset']


_CAPTURE = threading.local()      # per host thread: the (workspace, weight-plane buffer) pair of a HIP-graph capture


class capture_buffers:
    """`with ops.capture_buffers(ws, wp):` - while THIS thread captures a HIP graph, split-K launches use `ws` and
    weights scopes put their planes into `wp` (the graph's owner keeps both alive for the replays) instead of the
    per-stream buffers.  Thread-local: another thread entering a scope on its own stream meanwhile is unaffected."""

    def __init__(self, ws, wp):
        self.pair = (ws, wp)

    def __enter__(self):
        self.prev = getattr(_CAPTURE, 'pair', None)
        _CAPTURE.pair = self.pair
        return self

    def __exit__(self, *exc):
        _CAPTURE.pair = self.prev
        return False


def _capture_pair():
    return getattr(_CAPTURE, 'pair', None) or (None, None)


def splitk_ws(device=None):
    """Split-K workspace of the CURRENT stream on `device` (allocated once per (device, stream)): kernels on one
    stream use it one after another; two streams never share one - their split-K launches may overlap."""
    cap_ws = _capture_pair()[0]
    if cap_ws is not None:
        return cap_ws
    device = torch.device(device if device is not None else torch.cuda.current_device())
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, torch.cuda.current_stream(index).cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = _SPLITK_WS[key] = torch.empty(splitk_ws_floats(), dtype=torch.float32, device=device)
    return ws


def _attach_ws(problem, like_ptr_device):
    if not problem.splitk_ws:
        ws = splitk_ws(like_ptr_device)
        problem.splitk_ws, problem.splitk_ws_floats = ws.data_ptr(), ws.numel()


def ptr(t):
    return None if t is None else t.data_ptr()


def upload(values, dtype, device):
    """Small host list -> device tensor through pinned memory, asynchronously.  A pageable
    `torch.tensor(values, device=...)` is a stream-ordered blocking copy: the host would stall behind every
    kernel already queued (1.2 ms per criterion call in the XE training step).  A tensor that already lives on the
    device passes through (static inputs of a captured training iteration: no host copy may sit inside a graph)."""
    if isinstance(values, torch.Tensor):
        if values.is_cuda:
            return values if values.dtype == dtype else values.to(dtype)
        return values.to(dtype).pin_memory().to(device, non_blocking=True)
    return torch.tensor(values, dtype=dtype).pin_memory().to(device, non_blocking=True)


def to_device(x, device):
    """A batch tensor on `device` without stalling the host: a tensor in pageable host memory goes through pinned memory
    and a non-blocking copy (`x.to(device)` from pageable memory is stream-ordered AND blocks the host - it waits for
    everything queued, i.e. for the previous iteration); device tensors and data.RowGather pass through / expand."""
    if not torch.is_tensor(x):
        return x.to(device)                  # (data.RowGather: expanded on the device)
    if x.is_cuda or torch.device(device).type != 'cuda':
        return x.to(device)
    if x.is_pinned():
        return x.to(device, non_blocking=True)
    if x.numel() * x.element_size() > (1 << 20):
        # whole feature batches from pageable memory (151 MB for 512 images): page-locking a fresh buffer per batch costs
        # more than the runtime's own staged copy - data.DevicePrefetcher (kept pinned buffers) or data.DeviceFeatureStore
        # are the ways to take these off the host's critical path
        return x.to(device)
    return x.pin_memory().to(device, non_blocking=True)


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.HipLibraryError(
                'insenticap_model_amd runs on a ROCm device only (got a CPU tensor); there is no CPU path')


def _fill_segs(dst, segs):
    if not 1 <= len(segs) <= _lib.ISC_MAX_SEG:
        raise ValueError('1..4 K-segments supported')
    for i, sg in enumerate(segs):
        A, W = sg[0], sg[1]
        assert A.dim() == 2 and W.dim() == 2 and A.stride(1) == 1 and W.stride(1) == 1
        assert A.shape[1] == W.shape[1], (A.shape, W.shape)
        assert A.dtype == torch.float32 and W.dtype == torch.float32
        s = dst[i]
        s.A, s.W = A.data_ptr(), W.data_ptr()
        s.lda, s.ldw, s.K = A.stride(0), W.stride(0), A.shape[1]
        planes = sg[2] if len(sg) > 2 else None      # optional [2, M, K] f16 planes of A (isc_seg.A_hi / A_lo)
        if planes is not None:
            assert planes.dtype == torch.float16 and planes.is_contiguous() and planes.shape == (2,) + tuple(A.shape)
            s.A_hi, s.A_lo = planes_ptrs(planes)
        else:
            s.A_hi = s.A_lo = None


def set_tile_override(tile):
    """Force the GEMM tile shape (0..3, see include/insenticap_hip.h) or -1 for the cost model; returns the previous value."""
    return _lib.load().isc_set_tile_override(int(tile))


def set_h3_mode(mode):
    """Split-f16 GEMM path: 0 = off, 1 = auto (default), 2 = large kernels whenever shapes allow, 3 / 4 = skinny kernel
    with 32 x 32 / 64 x 64 tiles whenever shapes allow (include/insenticap_hip.h); returns the previous mode."""
    return _lib.load().isc_set_h3_mode(int(mode))


class OutOfDomain(Exception):
    """Raised inside the product when a roll-out met non-finite values with the split-f16 engine on - at a point where
    nothing has been updated yet; the owner of the iteration redoes it on the exact-fp32 engine (Detector.forward)."""


_EXACT = {'depth': 0, 'prev': 1, 'lock': threading.Lock()}


class exact_fp32_engine:
    """`with ops.exact_fp32_engine():` - every GEMM launched meanwhile runs on the exact-fp32 MFMA tiles
    (isc_set_h3_mode(0); operand domain = fp32's), the previous mode is restored when the outermost block ends.  The mode
    is a process-wide tuning word: launches of other host threads inside the window also take the exact tiles (same
    results up to fp32 summation order).  Used by the product to SERVE inputs beyond the split-f16 domain |x| < 65504
    instead of rejecting them (Captioner / Detector: numerics_checks)."""

    def __enter__(self):
        with _EXACT['lock']:
            if _EXACT['depth'] == 0:
                _EXACT['prev'] = set_h3_mode(0)
            _EXACT['depth'] += 1
        return self

    def __exit__(self, *exc):
        with _EXACT['lock']:
            _EXACT['depth'] -= 1
            if _EXACT['depth'] == 0:
                set_h3_mode(_EXACT['prev'])
        return False


def h3_mode():
    """The current split-f16 mode (isc_set_h3_mode) without changing it."""
    return _lib.load().isc_set_h3_mode(-1)


def set_gemv_rows(rows):
    """Few-row launches (M <= rows <= 8) as fused matrix-vector kernels; 0 = off.  Returns the previous value."""
    return _lib.load().isc_set_gemv_rows(int(rows))


_H3W_BUF = {}


class h3_weights_scope:
    """`with ops.h3_weights_scope(device):` - the weights passed to the forward GEMMs do not change inside the block
    (include/insenticap_hip.h: isc_h3_weights_begin), so their f16 planes are built once per block, not per launch.
    The scope belongs to the CURRENT stream (library side: one slot per stream; here: one plane buffer and one
    nesting depth per (device, stream)), so two host threads running two captioners on two streams are independent.
    Nested scopes on the same stream are ignored (the outer one stays in force).
    `key` (hashable, optional): identifies the weight VALUES (Captioner passes the parameters' storage pointers and
    version counters).  A keyed scope is suspended, not closed, on exit; the next keyed scope on the stream with an
    equal key resumes it with its planes intact - consecutive eval-mode calls then split the weights once."""
    _depth = {}
    _suspended = {}          # (device, stream) -> key of the suspended scope
    _cold_begins = {}        # (device, stream) -> scopes begun afresh there (planes rebuilt, entries laid out anew)
    _lock = threading.Lock()

    @classmethod
    def cold_begins(cls, keys):
        """How often a scope on these (device index, stream handle) pairs started afresh (HIP graphs captured there hold
        plane addresses of the layout that was current at capture)."""
        with cls._lock:
            return sum(cls._cold_begins.get(k, 0) for k in keys)

    @classmethod
    def rekey_epoch(cls, keys, epoch_before):
        """Suspended scopes on these streams whose key carried `epoch_before`: move them to the current WEIGHT_EPOCH
        (their planes were refreshed by launches the host did not see: a replayed training graph)."""
        with cls._lock:
            for k in keys:
                wk = cls._suspended.get(k)
                if isinstance(wk, tuple) and wk and wk[-1] == epoch_before:
                    cls._suspended[k] = wk[:-1] + (WEIGHT_EPOCH,)

    @classmethod
    def forget(cls, pred):
        """Drops the suspended scopes whose key satisfies `pred` (a Captioner being freed: its planes must never be
        resumed); the next scope on such a stream begins afresh."""
        with cls._lock:
            for k in [k for k, wk in cls._suspended.items() if pred(wk)]:
                del cls._suspended[k]

    def __init__(self, device, key=None):
        self.device = torch.device(device)
        self.wkey = key

    def __enter__(self):
        cls = h3_weights_scope
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.key = key = (index, torch.cuda.current_stream(index).cuda_stream)     # one buffer per stream, as splitk_ws
        self.opened = False
        with cls._lock:
            depth = cls._depth[key] = cls._depth.get(key, 0) + 1
        try:                      # anything that raises below must give the depth back: __exit__ will not run, and a
            buf = was = None      # depth stuck at >= 1 would turn every later scope on this stream into a no-op
            cap_wp = _capture_pair()[1]
            if depth == 1 and cap_wp is not None:         # a HIP graph is being captured: the graph owns the plane
                buf = cap_wp                              # buffer, and replays must re-split (weights may change
                self.wkey = None                          # in place between replays): never keyed, never resumed
            elif depth == 1:
                with cls._lock:
                    buf = _H3W_BUF.get(key)
                    was = cls._suspended.pop(key, None)
                if buf is None:
                    buf = torch.empty(h3w_bytes(), dtype=torch.uint8, device=self.device)
                    with cls._lock:
                        _H3W_BUF[key] = buf
            if buf is not None:
                lib = _lib.load()
                if self.wkey is not None and was == self.wkey and \
                        lib.isc_h3_weights_resume(buf.data_ptr(), C.c_void_p(key[1])) == 0:
                    self.opened = True
                else:
                    rc = lib.isc_h3_weights_begin(buf.data_ptr(), buf.numel(), C.c_void_p(key[1]))
                    with cls._lock:
                        cls._cold_begins[key] = cls._cold_begins.get(key, 0) + 1
                    if rc != -4:                          # ISC_E_WORKSPACE: all scope slots taken - run without one
                        _lib.check(rc, 'isc_h3_weights_begin')
                        self.opened = True
        except BaseException:
            with cls._lock:
                cls._depth[key] -= 1
            raise
        return self

    def __exit__(self, *exc):
        cls = h3_weights_scope
        with cls._lock:
            cls._depth[self.key] -= 1
            if self.opened and self.wkey is not None and exc[0] is None:
                cls._suspended[self.key] = self.wkey
        if self.opened:
            if self.wkey is not None and exc[0] is None:
                _lib.load().isc_h3_weights_suspend(C.c_void_p(self.key[1]))
            else:
                _lib.load().isc_h3_weights_end(C.c_void_p(self.key[1]))
        return False


def linear_problem(segs, out, bias0=None, bias1=None, bias2=None, relu=False, keep_mask=None, mask_scale=1.0,
                   out_pre=None, accumulate=False):
    """out[M,N] = act(sum_s A_s W_s^T + bias0 + bias1) [* keep_mask * mask_scale]."""
    p = LinearProblem()
    _fill_segs(p.seg, segs)
    p.nseg = len(segs)
    p.M, p.N = out.shape
    assert segs[0][0].shape[0] == p.M and segs[0][1].shape[0] == p.N
    assert out.stride(1) == 1
    p.relu = int(relu)
    p.bias0, p.bias1, p.bias2 = ptr(bias0), ptr(bias1), ptr(bias2)
    p.keep_mask = ptr(keep_mask)
    p.mask_scale = mask_scale
    p.ldc = out.stride(0)
    p.C = out.data_ptr()
    p.C_pre = ptr(out_pre)
    p.accumulate = int(accumulate)
    if out_pre is not None:
        assert out_pre.stride(0) == out.stride(0)
    return p


def linear_fwd(problems):
    lib = _lib.load()
    _attach_ws(problems[0], torch.cuda.current_device())
    arr = (LinearProblem * len(problems))(*problems)
    e0 = TIMER.begin()
    check(lib.isc_linear_fwd(arr, len(problems), stream()), 'isc_linear_fwd')
    if e0 is not None:
        fl = sum(2.0 * q.M * q.N * sum(q.seg[i].K for i in range(q.nseg)) for q in problems)
        TIMER.end(e0, 'linear[' + '+'.join('%dx%dx%d' % (q.M, q.N, sum(q.seg[i].K for i in range(q.nseg)))
                                          for q in problems) + ']', fl)


def lstm_fwd(segs, b_ih, b_hh, c_prev, h_out, c_out, gates_out=None, h_keep_mask=None,
             mask_scale=1.0, hdrop_out=None, pre=None, tab=None, tab_ids=None, h_planes=None):
    """h_planes: optional [2, M, H] f16 tensor receiving the split-f16 planes of h_out (for consumers' segments)."""
    lib = _lib.load()
    p = LstmProblem()
    _fill_segs(p.seg, segs)
    p.nseg = len(segs)
    p.M, p.H = c_prev.shape
    assert c_prev.is_contiguous() and h_out.is_contiguous() and c_out.is_contiguous()
    p.b_ih, p.b_hh = ptr(b_ih), ptr(b_hh)
    if pre is not None:
        assert pre.is_contiguous() and pre.shape == (p.M, 4 * p.H)
        p.pre = pre.data_ptr()
    if tab is not None:
        assert tab.is_contiguous() and tab.shape[1] == 4 * p.H and tab_ids.dtype == torch.int64
        p.tab, p.tab_ids, p.tab_ids_stride = tab.data_ptr(), tab_ids.data_ptr(), tab_ids.stride(0)
    p.c_prev, p.h_out, p.c_out = c_prev.data_ptr(), h_out.data_ptr(), c_out.data_ptr()
    p.gates_out = ptr(gates_out)
    p.h_keep_mask = ptr(h_keep_mask)
    p.mask_scale = mask_scale
    p.hdrop_out = ptr(hdrop_out)
    if h_planes is not None:
        assert h_planes.dtype == torch.float16 and h_planes.is_contiguous() and h_planes.shape == (2, p.M, p.H)
        p.h_hi, p.h_lo = planes_ptrs(h_planes)
    _attach_ws(p, c_prev.device)
    e0 = TIMER.begin()
    check(lib.isc_lstm_fwd(C.byref(p), stream()), 'isc_lstm_fwd')
    if e0 is not None:
        k = _seg_k(segs)
        TIMER.end(e0, 'lstm[%dx%dx%d]' % (p.M, 4 * p.H, k), 2.0 * p.M * 4 * p.H * k)


def step_fwd(plan):
    """One decode step from a prepared isc_step_plan (all kernels enqueued by the library)."""
    check(_lib.load().isc_step_fwd(C.byref(plan), stream()), 'isc_step_fwd')


class stream_gate:
    """`with ops.stream_gate(ptr):` - the forward launches enqueued on the current stream inside the block return at once
    when the device int32 at address `ptr` reads 0 at run time (isc_set_stream_gate: a batched beam search's step t under
    live[t]).  ptr None / 0: no gate, the block is plain."""

    def __init__(self, ptr):
        self.ptr = int(ptr) if ptr else 0
        self.st = None

    def __enter__(self):
        if self.ptr:
            self.st = stream()
            check(_lib.load().isc_set_stream_gate(C.c_void_p(self.ptr), self.st), 'isc_set_stream_gate')
        return self

    def __exit__(self, et, ev, tb):
        if self.st is not None:
            _lib.load().isc_set_stream_gate(None, self.st)
            self.st = None
        return False


def rows_stats_tile(V):
    """Column-tile width of the few-row classifier's statistics (isc_rows_stats_tile)."""
    return _lib.load().isc_rows_stats_tile(int(V))


def set_rows_scan_max(rows):
    """Largest inference step whose gated scan runs on the one-workgroup-per-row kernel (isc_set_rows_scan_max; 0 = off,
    negative = query).  Returns the previous value."""
    return _lib.load().isc_set_rows_scan_max(int(rows))


def copy_multi(dsts, srcs):
    """dsts[i].copy_(srcs[i]) for up to 8 pairs of contiguous same-shape, same-dtype device tensors in ONE launch
    (isc_copy_multi); longer lists go out in groups of 8."""
    pairs = [(d, s) for d, s in zip(dsts, srcs) if d is not None and d.numel()]
    for d, s in pairs:
        if d.dtype != s.dtype or d.shape != s.shape or not (d.is_contiguous() and s.is_contiguous() and d.is_cuda and s.is_cuda):
            raise ValueError('copy_multi: contiguous device tensors of equal shape and dtype expected')
    lib = _lib.load()
    for i in range(0, len(pairs), 8):
        grp = pairs[i:i + 8]
        n = len(grp)
        D = (C.c_void_p * n)(*[d.data_ptr() for d, _ in grp])
        S = (C.c_void_p * n)(*[s.data_ptr() for _, s in grp])
        Bn = (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in grp])
        check(lib.isc_copy_multi(D, S, Bn, n, stream()), 'isc_copy_multi')


def stage_inputs(dsts, srcs):
    """A call's inputs -> the static buffers of a captured graph (None entries skipped): one launch when every pair
    is contiguous and of one dtype, torch's per-dtype copies otherwise."""
    pairs = [(d, s) for d, s in zip(dsts, srcs) if d is not None]
    if all(d.dtype == s.dtype and d.shape == s.shape and s.is_cuda and s.is_contiguous() and d.is_contiguous()
           for d, s in pairs):
        copy_multi([d for d, _ in pairs], [s for _, s in pairs])
        return
    by_dtype = {}
    for d, s in pairs:
        by_dtype.setdefault(d.dtype, ([], []))
        by_dtype[d.dtype][0].append(d)
        by_dtype[d.dtype][1].append(s)
    for ds, ss in by_dtype.values():
        torch._foreach_copy_(ds, ss, non_blocking=True)


def set_h3v(on):
    """Classifier launches of the skinny split-f16 path on gemm_h3v_kernel (isc_set_h3v; 0 = the ring-staged form it
    replaced).  Returns the previous value."""
    return _lib.load().isc_set_h3v(int(on))


def rows_step_supported(plan):
    return bool(_lib.load().isc_rows_step_supported(C.byref(plan)))


def rows_step_fwd(plan, ext):
    """One decode step on <= 8 rows from a prepared isc_step_plan + isc_rows_ext (csrc/rows.hip: five launches)."""
    check(_lib.load().isc_rows_step_fwd(C.byref(plan), C.byref(ext), stream()), 'isc_rows_step_fwd')


def rows_vocab_fwd(h, W, bias, part_max, part_sum, part_idx, ext, logits=None):
    lib = _lib.load()
    M, K = h.shape
    V = W.shape[0]
    assert h.stride(1) == 1 and W.stride(1) == 1
    check(lib.isc_rows_vocab_fwd(h.data_ptr(), h.stride(0), W.data_ptr(), W.stride(0), bias.data_ptr(), M, V, K,
                                 part_max.data_ptr(), part_sum.data_ptr(), part_idx.data_ptr(), ptr(logits),
                                 logits.stride(0) if logits is not None else 0, C.byref(ext), stream()),
          'isc_rows_vocab_fwd')


def beam_select(args):
    """Top-k + candidate merge of one beam step in one launch (isc_beam_select)."""
    check(_lib.load().isc_beam_select(C.byref(args), stream()), 'isc_beam_select')


def step_bwd(plan):
    check(_lib.load().isc_step_bwd(C.byref(plan), stream()), 'isc_step_bwd')


def vocab_fwd(h, W, bias, part_max, part_sum, part_idx, logits=None, h_planes=None):
    lib = _lib.load()
    hi = lo = None
    if h_planes is not None:
        assert h_planes.dtype == torch.float16 and h_planes.is_contiguous() and h_planes.shape == (2,) + tuple(h.shape)
        hi, lo = planes_ptrs(h_planes)
    M, K = h.shape
    V = W.shape[0]
    assert h.stride(1) == 1 and W.stride(1) == 1
    ld = logits.stride(0) if logits is not None else 0
    e0 = TIMER.begin()
    ws = splitk_ws(h.device)
    check(lib.isc_vocab_fwd(h.data_ptr(), h.stride(0), W.data_ptr(), W.stride(0), bias.data_ptr(), M, V, K,
                            ptr(logits), ld, part_max.data_ptr(), part_sum.data_ptr(),
                            part_idx.data_ptr(), hi, lo, ws.data_ptr(), ws.numel(), stream()), 'isc_vocab_fwd')
    TIMER.end(e0, 'vocab[%dx%dx%d]' % (M, V, K), 2.0 * M * V * K)


def logsoftmax_apply(logits, part_max, part_sum, lse_out=None):
    lib = _lib.load()
    M, V = logits.shape
    assert logits.stride(1) == 1
    check(lib.isc_logsoftmax_apply(logits.data_ptr(), logits.stride(0), M, V, part_max.data_ptr(),
                                   part_sum.data_ptr(), ptr(lse_out), stream()), 'isc_logsoftmax_apply')


def logsoftmax_apply_steps(logits_btv, part_max_tbn, part_sum_tbn, src_tbv=None, step_rows=0):
    """log-softmax over all steps of a [B,T,V] logits tensor from the per-step tile statistics [T,B,n_tile]: in place,
    or from raw logits stacked per step `src_tbv` [T,B,V] (one classifier launch over all steps).
    step_rows > 0 (merged unroll): the statistics / src are [T, :B] row slices of stacks with `step_rows` rows per step
    (views whose first element is this branch's first row)."""
    B, T, V = logits_btv.shape
    assert logits_btv.stride(2) == 1
    if step_rows:
        n_tile = part_max_tbn.shape[2]
        assert step_rows >= B and part_max_tbn.shape[:2] == (T, B) and part_sum_tbn.shape == part_max_tbn.shape
        for x, w in ((part_max_tbn, n_tile), (part_sum_tbn, n_tile)) + (((src_tbv, V),) if src_tbv is not None else ()):
            assert x.stride(2) == 1 and x.stride(1) == w and x.stride(0) == step_rows * w, (x.shape, x.stride())
    else:
        assert part_max_tbn.is_contiguous() and part_sum_tbn.is_contiguous()
        assert part_max_tbn.shape[:2] == (T, B) and part_sum_tbn.shape == part_max_tbn.shape
        assert src_tbv is None or (src_tbv.is_contiguous() and src_tbv.shape == (T, B, V))
    check(_lib.load().isc_logsoftmax_apply_steps(logits_btv.data_ptr(), logits_btv.stride(0), logits_btv.stride(1), B, T,
                                                 V, part_max_tbn.data_ptr(), part_sum_tbn.data_ptr(), ptr(src_tbv),
                                                 int(step_rows), stream()),
          'isc_logsoftmax_apply_steps')


def _planes_ptrs(planes, like):
    if planes is None:
        return None, None
    assert planes.dtype == torch.float16 and planes.is_contiguous() and planes.shape == (2,) + tuple(like.shape)
    return planes_ptrs(planes)


def scan_problem(P, V, q, w, w_bias, out, alpha_out=None, q2=None, out_planes=None, row_ids=None):
    """P [B,R,A], V [B,R,D] contiguous; q/q2 [B,A]; w [A] (or [1,A]); out [B,D];
    alpha_out: [B,R] view with unit inner stride (row stride arbitrary).
    row_ids (gather mode): int64 [B,R]; P / V are then tables [n,A] / [n,D] and region r of row b is their row
    row_ids[b,r]."""
    s = ScanProblem()
    assert P.is_contiguous() and V.is_contiguous() and q.is_contiguous() and out.is_contiguous()
    s.P, s.V, s.q, s.q2, s.w = P.data_ptr(), V.data_ptr(), q.data_ptr(), ptr(q2), w.data_ptr()
    s.w_bias = ptr(w_bias)
    if row_ids is not None:
        assert row_ids.dtype == torch.int64 and row_ids.dim() == 2 and row_ids.stride(1) == 1 and P.dim() == 2
        s.row_ids, s.row_ids_ld = row_ids.data_ptr(), row_ids.stride(0)
        s.R, s.A, s.D = row_ids.shape[1], P.shape[1], V.shape[1]
    else:
        s.R, s.A, s.D = P.shape[1], P.shape[2], V.shape[2]
    s.out = out.data_ptr()
    s.out_hi, s.out_lo = _planes_ptrs(out_planes, out)
    if alpha_out is not None:
        assert alpha_out.stride(1) == 1
        s.alpha_out, s.alpha_ld = alpha_out.data_ptr(), alpha_out.stride(0)
    return s


def attn_scan_fwd(problems, B):
    lib = _lib.load()
    arr = (ScanProblem * len(problems))(*problems)
    e0 = TIMER.begin()
    check(lib.isc_attn_scan_fwd(arr, len(problems), B, stream()), 'isc_attn_scan_fwd')
    if e0 is not None:
        # algorithmic bytes: P and V streamed once per row (B*2*R*E*4 for the content scan)
        nb = sum(4.0 * B * q.R * (q.A + q.D) for q in problems)
        TIMER.end(e0, 'attn_scan[' + '+'.join('%dx%dx%d' % (B, q.R, q.A) for q in problems) + ']', 0.0, nb)


def attn_scan_gate_fwd(scans, G, zh, b_gc, b_gs, w_gate, b_gate, f, beta_out=None, f_planes=None):
    """isc_attn_scan_gate_fwd: content + sentiment scan, gate sum and gate mix of one decode step in one launch.
    scans: [content, sentiment] isc_scan_problem (ops.scan_problem; their `out` is ignored); G: the two projected
    feature tensors / tables (ops-level contract: same row layout as the scan's V); zh [B,A]; f [B,D]."""
    a = _lib.ScanGateArgs()
    for i in range(2):
        C.memmove(C.byref(a.scan[i]), C.byref(scans[i]), C.sizeof(ScanProblem))
        a.scan[i].out = None
        a.scan[i].out_hi = a.scan[i].out_lo = None
        assert G[i].is_contiguous() and G[i].dtype == torch.float32
        a.G[i] = G[i].data_ptr()
    B = zh.shape[0]
    assert zh.is_contiguous() and f.is_contiguous()
    a.zh, a.b_gc, a.b_gs, a.w_gate, a.b_gate = zh.data_ptr(), b_gc.data_ptr(), b_gs.data_ptr(), w_gate.data_ptr(), ptr(b_gate)
    a.f = f.data_ptr()
    a.f_hi, a.f_lo = _planes_ptrs(f_planes, f)
    if beta_out is not None:
        a.beta, a.beta_ld = beta_out.data_ptr(), beta_out.stride(0)
    e0 = TIMER.begin()
    check(_lib.load().isc_attn_scan_gate_fwd(C.byref(a), B, stream()), 'isc_attn_scan_gate_fwd')
    if e0 is not None:
        nb = sum(4.0 * B * q.R * (2 * q.A + q.D) for q in scans)
        TIMER.end(e0, 'attn_scan_gate[' + '+'.join('%dx%dx%d' % (B, q.R, q.A) for q in scans) + ']', 0.0, nb)


def gate_mix_fwd(z, w, w_bias, v, s, out, beta_out=None, out_planes=None):
    lib = _lib.load()
    B, A = z.shape
    D = v.shape[1]
    assert z.is_contiguous() and v.is_contiguous() and s.is_contiguous() and out.is_contiguous()
    bl = beta_out.stride(0) if beta_out is not None else 0
    e0 = TIMER.begin()
    check(lib.isc_gate_mix_fwd(z.data_ptr(), w.data_ptr(), ptr(w_bias), v.data_ptr(), s.data_ptr(), B, A, D,
                               out.data_ptr(), ptr(beta_out), bl, *_planes_ptrs(out_planes, out), stream()),
          'isc_gate_mix_fwd')
    TIMER.end(e0, 'gate_mix[%dx%d]' % (B, A), 0.0, 4.0 * B * (A + 3 * D))


def embed_relu_fwd(emb, ids, out, add=None):
    lib = _lib.load()
    V, W = emb.shape
    assert ids.dtype == torch.int64 and ids.dim() == 1 and out.is_contiguous()
    check(lib.isc_embed_relu_fwd(emb.data_ptr(), V, W, ids.data_ptr(), ids.stride(0), ptr(add),
                                 ids.shape[0], out.data_ptr(), stream()), 'isc_embed_relu_fwd')


def embed_relu_mean_fwd(emb, ids, out):
    lib = _lib.load()
    V, W = emb.shape
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    B, Cn = ids.shape
    check(lib.isc_embed_relu_mean_fwd(emb.data_ptr(), V, W, ids.data_ptr(), Cn, B, out.data_ptr(),
                                      stream()), 'isc_embed_relu_mean_fwd')


def embed_senti_words_fwd(emb, ids, pad_id, out, keep_mask=None, mask_scale=1.0):
    lib = _lib.load()
    V, W = emb.shape
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    B, n = ids.shape
    check(lib.isc_embed_senti_words_fwd(emb.data_ptr(), V, W, ids.data_ptr(), n, pad_id, B,
                                        ptr(keep_mask), mask_scale, out.data_ptr(), stream()),
          'isc_embed_senti_words_fwd')


def rollout_finalize(step):
    lib = _lib.load()
    e0 = TIMER.begin()
    check(lib.isc_rollout_finalize(C.byref(step), stream()), 'isc_rollout_finalize')
    TIMER.end(e0, 'rollout_finalize[%d]' % step.B, 0.0, 4.0 * step.B * (3 * step.n_tile + 2 * step.W))


def sched_sample(logp, part_max, part_sum, part_idx, u_select, u_draw, ss_prob, base_ids, out_ids, raw=False):
    """out_ids[b] = u_select[b] < ss_prob ? draw from exp(logp[b]) : base_ids[b]  (base_ids may be a strided column).
    raw: `logp` holds the row's raw logits (isc_sched_sample_raw)."""
    lib = _lib.load()
    M, V = logp.shape
    assert logp.stride(1) == 1 and out_ids.is_contiguous() and base_ids.dtype == torch.int64
    fn = lib.isc_sched_sample_raw if raw else lib.isc_sched_sample
    check(fn(logp.data_ptr(), logp.stride(0), M, V, part_max.data_ptr(), part_sum.data_ptr(),
             part_idx.data_ptr(), u_select.data_ptr(), u_draw.data_ptr(), float(ss_prob),
             base_ids.data_ptr(), base_ids.stride(0), out_ids.data_ptr(), stream()), 'isc_sched_sample')


def beam_topk(logits, part_max, part_sum, last_word, beam, pad_id, sos_id, unk_id, mask_special,
              decoding_constraint, top_val, top_idx):
    lib = _lib.load()
    rows, V = logits.shape
    n_tile = part_max.shape[1]
    check(lib.isc_beam_topk(logits.data_ptr(), logits.stride(0), part_max.data_ptr(), part_sum.data_ptr(),
                            n_tile, rows, V, beam, last_word.data_ptr(), pad_id, sos_id, unk_id,
                            int(mask_special), int(decoding_constraint), top_val.data_ptr(),
                            top_idx.data_ptr(), stream()), 'isc_beam_topk')


def beam_merge(args):
    """One candidate-merge step of a batched beam search on the device (isc_beam_merge)."""
    check(_lib.load().isc_beam_merge(C.byref(args), stream()), 'isc_beam_merge')


def beam_gather(state_next, state_cur, gather, out):
    """[2,2,rows,H] recurrent states: out = [next ; current][gather] row by row (isc_beam_gather)."""
    rows, H = state_cur.shape[-2], state_cur.shape[-1]
    planes = state_cur.numel() // (rows * H)
    assert state_next.is_contiguous() and state_cur.is_contiguous() and out.is_contiguous()
    check(_lib.load().isc_beam_gather(state_next.data_ptr(), state_cur.data_ptr(), gather.data_ptr(), out.data_ptr(),
                                      planes, rows, H, stream()), 'isc_beam_gather')


def xe_loss_fwd(logp, target, lengths_i32, out2):
    lib = _lib.load()
    B, T, V = logp.shape
    assert logp.is_contiguous() and target.is_contiguous()
    check(lib.isc_xe_loss_fwd(logp.data_ptr(), target.data_ptr(), lengths_i32.data_ptr(), B, T, V,
                              out2.data_ptr(), stream()), 'isc_xe_loss_fwd')


__all__ = [n for n in dir() if n.endswith('_fwd') or n.endswith('_problem')] + [
    'RolloutStep', 'rollout_finalize', 'beam_topk', 'logsoftmax_apply', 'require_device']


# ---------------------------------------------------------------------------- backward
NN, TN = 1, 2


def gemm_problem(segs, out, layout, accumulate=False, bias0=None):
    """layout NN: out[M,N] (+)= sum_s A_s[M,K] @ W_s[K,N];  TN: out[M,N] (+)= sum_s A_s[K,M]^T @ W_s[K,N].
    All operands are 2-D views with unit inner stride."""
    p = LinearProblem()
    if not 1 <= len(segs) <= _lib.ISC_MAX_SEG:
        raise ValueError('1..4 K-segments supported')
    p.nseg = len(segs)
    p.M, p.N = out.shape
    for i, (A, W) in enumerate(segs):
        assert A.dim() == 2 and W.dim() == 2 and A.stride(1) == 1 and W.stride(1) == 1
        assert A.dtype == torch.float32 and W.dtype == torch.float32
        s = p.seg[i]
        s.A, s.W, s.lda, s.ldw = A.data_ptr(), W.data_ptr(), A.stride(0), W.stride(0)
        if layout == NN:
            assert A.shape[0] == p.M and W.shape[1] == p.N and A.shape[1] >= W.shape[0], (A.shape, W.shape)
            s.K = W.shape[0]
        else:
            assert A.shape[1] == p.M and W.shape[1] == p.N and A.shape[0] == W.shape[0], (A.shape, W.shape)
            s.K = A.shape[0]
    assert out.stride(1) == 1
    p.ldc, p.C = out.stride(0), out.data_ptr()
    p.accumulate = int(accumulate)
    p.bias0 = ptr(bias0)
    return p


def gemm_bwd(problems, layout):
    lib = _lib.load()
    _attach_ws(problems[0], torch.cuda.current_device())
    arr = (LinearProblem * len(problems))(*problems)
    e0 = TIMER.begin()
    check(lib.isc_gemm_bwd(arr, len(problems), layout, stream()), 'isc_gemm_bwd')
    if e0 is not None:
        fl = sum(2.0 * q.M * q.N * sum(q.seg[i].K for i in range(q.nseg)) for q in problems)
        TIMER.end(e0, ('gemm_nn[' if layout == NN else 'gemm_tn[') +
                  '+'.join('%dx%dx%d' % (q.M, q.N, sum(q.seg[i].K for i in range(q.nseg))) for q in problems) + ']',
                  fl)


def logsoftmax_bwd(dlogp, logp, dlogits, M, V, remap_T=0):
    """dlogp/logp: M contiguous rows of V; dlogits [M, ld_out>=V] (padding zero-filled).
    remap_T>0: input rows are (b,t) b-major, output rows (t,b) t-major."""
    lib = _lib.load()
    assert dlogp.is_contiguous() and logp.is_contiguous() and dlogits.stride(1) == 1
    check(lib.isc_logsoftmax_bwd(dlogp.data_ptr(), logp.data_ptr(), V, M, V, dlogits.data_ptr(),
                                 dlogits.stride(0), remap_T, stream()), 'isc_logsoftmax_bwd')


def logsoftmax_bwd_sparse(dense, logp, sparse, dlogits, M, V, remap_T=0, scale=None, out_step_rows=0):
    """isc_logsoftmax_bwd_sparse: d logits from an optional dense d log-prob [M,V] plus up to two (ids [M] int64,
    coef [M] fp32) pairs - one column per row each - times the optional device scalar `scale`.
    out_step_rows > 0 (with remap_T): rows per step of the time-major output; `dlogits` starts at this branch's first row."""
    lib = _lib.load()
    assert logp.is_contiguous() and dlogits.stride(1) == 1 and (dense is None or dense.is_contiguous())
    n = len(sparse)
    ids = (C.c_void_p * max(n, 1))(*[i.data_ptr() for i, _ in sparse])
    cf = (C.c_void_p * max(n, 1))(*[c.data_ptr() for _, c in sparse])
    for i, c in sparse:
        assert i.dtype == torch.int64 and c.dtype == torch.float32 and i.is_contiguous() and c.is_contiguous()
        assert i.numel() == M and c.numel() == M
    check(lib.isc_logsoftmax_bwd_sparse(ptr(dense), logp.data_ptr(), V, M, V, ids, cf, n, ptr(scale),
                                        dlogits.data_ptr(), dlogits.stride(0), remap_T, int(out_step_rows), stream()),
          'isc_logsoftmax_bwd_sparse')


def gather_logp_raw(raw, ld_b, ld_t, B, T, V, part_max, part_sum, step_rows, ids, out, live=None):
    """out[b,t] = log p(ids[b,t]) from the RAW logits (row (b,t) at raw + b*ld_b + t*ld_t floats) and the step-stacked
    tile statistics (row t*step_rows + b) - isc_gather_logp_raw; `live` [T] optionally multiplies column t."""
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.numel() == B * T and out.is_contiguous()
    assert out.numel() == B * T and out.dtype == torch.float32
    check(_lib.load().isc_gather_logp_raw(raw.data_ptr(), int(ld_b), int(ld_t), B, T, V, part_max.data_ptr(),
                                          part_sum.data_ptr(), int(step_rows), ids.data_ptr(), ptr(live), out.data_ptr(),
                                          stream()), 'isc_gather_logp_raw')


def xe_loss_tokens_fwd(tlp, lengths_i32, out2):
    B, T = tlp.shape
    assert tlp.is_contiguous() and tlp.dtype == torch.float32
    check(_lib.load().isc_xe_loss_tokens_fwd(tlp.data_ptr(), lengths_i32.data_ptr(), B, T, out2.data_ptr(), stream()),
          'isc_xe_loss_tokens_fwd')


def logsoftmax_bwd_raw(raw, ld_b, ld_t, B, T, V, part_max, part_sum, step_rows, sparse, dlogits, scale=None,
                       out_step_rows=0):
    """isc_logsoftmax_bwd_raw: d logits (time-major rows t*out_step_rows + b of `dlogits`) from (ids, coef) pairs in
    [B,T] order, the softmax term recomputed from the raw logits and their tile statistics."""
    n = len(sparse)
    assert 1 <= n <= 2 and dlogits.stride(1) == 1
    for i, c in sparse:
        assert i.dtype == torch.int64 and c.dtype == torch.float32 and i.is_contiguous() and c.is_contiguous()
        assert i.numel() == B * T and c.numel() == B * T
    ids = (C.c_void_p * n)(*[i.data_ptr() for i, _ in sparse])
    cf = (C.c_void_p * n)(*[c.data_ptr() for _, c in sparse])
    check(_lib.load().isc_logsoftmax_bwd_raw(raw.data_ptr(), int(ld_b), int(ld_t), B, T, V, part_max.data_ptr(),
                                             part_sum.data_ptr(), int(step_rows), ids, cf, n, ptr(scale), dlogits.data_ptr(),
                                             dlogits.stride(0), int(out_step_rows), stream()), 'isc_logsoftmax_bwd_raw')


def grad_scale(sources, out2):
    """isc_grad_scale: out[0:2] = {S, 1/S}, S the power of two that brings max |x| over `sources` into [2^-4, 2^-3);
    `out2` is a ZEROED float32[4] (its last two words are the reduction's state, left zeroed)."""
    assert out2.numel() >= 4
    srcs = [x for x in sources if x is not None and x.numel() > 0]
    for x in srcs:
        assert x.dtype == torch.float32 and x.is_contiguous()
    n = len(srcs)
    pp = (C.c_void_p * max(n, 1))(*[x.data_ptr() for x in srcs])
    nn_ = (C.c_int64 * max(n, 1))(*[x.numel() for x in srcs])
    check(_lib.load().isc_grad_scale(pp, nn_, n, out2.data_ptr(), stream()), 'isc_grad_scale')


def xe_loss_bwd_sparse(lengths_i32, T, gout, sum_count, coef):
    B = lengths_i32.numel()
    check(_lib.load().isc_xe_loss_bwd_sparse(lengths_i32.data_ptr(), B, T, gout.data_ptr(), sum_count.data_ptr(),
                                             coef.data_ptr(), stream()), 'isc_xe_loss_bwd_sparse')


def reward_loss_fwd(seq_logprobs, seq_masks, reward, out2):
    B, T = seq_logprobs.shape
    check(_lib.load().isc_reward_loss_fwd(seq_logprobs.data_ptr(), seq_masks.data_ptr(), reward.data_ptr(), B, T,
                                          out2.data_ptr(), stream()), 'isc_reward_loss_fwd')


def reward_loss_bwd(seq_masks, reward, gout, sum_count, dlogp):
    B, T = seq_masks.shape
    check(_lib.load().isc_reward_loss_bwd(seq_masks.data_ptr(), reward.data_ptr(), B, T, gout.data_ptr(),
                                          sum_count.data_ptr(), dlogp.data_ptr(), stream()), 'isc_reward_loss_bwd')


def lstm_bwd(dh, dh2, dc_next, gates, c_prev, c, dgates, dc_prev, dgates_sum=None):
    lib = _lib.load()
    M, H = c.shape
    for x in (dh, dh2, dc_next, gates, c_prev, c, dgates, dc_prev, dgates_sum):
        assert x is None or x.is_contiguous()
    check(lib.isc_lstm_bwd(dh.data_ptr(), ptr(dh2), ptr(dc_next), gates.data_ptr(), c_prev.data_ptr(),
                           c.data_ptr(), M, H, dgates.data_ptr(), dc_prev.data_ptr(), ptr(dgates_sum),
                           stream()), 'isc_lstm_bwd')


def scan_bwd_problem(P, V, q, w, alpha, dout, dP, dV, dq, dw_rows, accumulate, q2=None, de_out=None):
    s = _lib.ScanBwdProblem()
    for x in (P, V, q, dout, dP, dV, dq, dw_rows):
        assert x is None or x.is_contiguous()          # (dV None: attn_dv_from_alpha forms it after the sweep)
    assert alpha.stride(1) == 1
    s.P, s.V, s.q, s.q2, s.w = P.data_ptr(), V.data_ptr(), q.data_ptr(), ptr(q2), w.data_ptr()
    s.alpha, s.alpha_ld, s.dout = alpha.data_ptr(), alpha.stride(0), dout.data_ptr()
    s.R, s.A, s.D = P.shape[1], P.shape[2], V.shape[2]
    s.accumulate = int(accumulate)
    s.dP, s.dV, s.dq, s.dw_rows = ptr(dP), ptr(dV), dq.data_ptr(), dw_rows.data_ptr()
    s.de_out = ptr(de_out)
    return s


def attn_dv_from_alpha(alpha, dout_all, dV, step_rows=0):
    """dV[b,r,:] = sum_t alpha[b,t,r] * dout_all[t,b,:] in the backward sweep's order (isc_attn_dv_from_alpha).
    alpha: [B,T,R] view with unit inner stride; dout_all [T,B,D] contiguous - or, step_rows > 0, the [:, :B] rows of a
    [T,step_rows,D] stack (merged unroll); dV [B,R,D] contiguous."""
    B, T, R = alpha.shape
    D = dV.shape[2]
    assert alpha.stride(2) == 1 and dV.is_contiguous() and dout_all.shape == (T, B, D)
    if step_rows:
        assert dout_all.stride(2) == 1 and dout_all.stride(1) == D and dout_all.stride(0) == step_rows * D
    else:
        assert dout_all.is_contiguous()
    check(_lib.load().isc_attn_dv_from_alpha(alpha.data_ptr(), alpha.stride(0), alpha.stride(1), dout_all.data_ptr(),
                                             B, T, R, D, dV.data_ptr(), int(step_rows), stream()),
          'isc_attn_dv_from_alpha')


def attn_dp_from_de(P, q_all, w, de_all, dP, q2=None):
    """dP[b,r,:] = sum_t de_all[t,b,r] * w * (1 - tanh^2(P[b,r,:] + q_all[t,b,:] (+ q2[b,:]))) in the backward sweep's
    order (isc_attn_dp_from_de).  P, dP [B,R,A]; q_all [T,B,A]; de_all [T,B,R]; all contiguous."""
    B, R, A = P.shape
    T = q_all.shape[0]
    for x in (P, q_all, de_all, dP, q2):
        assert x is None or x.is_contiguous()
    assert q_all.shape == (T, B, A) and de_all.shape == (T, B, R) and dP.shape == P.shape
    check(_lib.load().isc_attn_dp_from_de(P.data_ptr(), q_all.data_ptr(), ptr(q2), w.data_ptr(), de_all.data_ptr(),
                                          B, T, R, A, dP.data_ptr(), stream()), 'isc_attn_dp_from_de')


def attn_scan_bwd(problems, B):
    lib = _lib.load()
    arr = (_lib.ScanBwdProblem * len(problems))(*problems)
    e0 = TIMER.begin()
    check(lib.isc_attn_scan_bwd(arr, len(problems), B, stream()), 'isc_attn_scan_bwd')
    if e0 is not None:
        nb = sum(4.0 * B * q.R * 3 * (q.A + q.D) for q in problems)   # read P,V; read-modify-write dP,dV
        TIMER.end(e0, 'attn_scan_bwd[' + '+'.join('%dx%dx%d' % (B, q.R, q.A) for q in problems) + ']', 0.0, nb)


def gate_mix_bwd(z, w, v, s, beta, dfeat, dv, ds, dz, dw_rows, db_rows, accumulate):
    lib = _lib.load()
    B, A = z.shape
    D = v.shape[1]
    check(lib.isc_gate_mix_bwd(z.data_ptr(), w.data_ptr(), v.data_ptr(), s.data_ptr(), beta.data_ptr(),
                               beta.stride(0), dfeat.data_ptr(), B, A, D, dv.data_ptr(), ds.data_ptr(),
                               dz.data_ptr(), dw_rows.data_ptr(), db_rows.data_ptr(), int(accumulate),
                               stream()), 'isc_gate_mix_bwd')


def embed_relu_bwd(emb, ids, dout, demb, n_rows, rows_per_grad=1, scale=1.0, ids_stride=1, pad_first=0,
                   pad_id=0, keep_mask=None, mask_scale=1.0, skip_id=-1):
    lib = _lib.load()
    V, W = emb.shape
    assert dout.is_contiguous() and demb.is_contiguous() and ids.dtype == torch.int64
    ws = splitk_ws(emb.device)          # position index of the ids (the stream's workspace: launches are ordered)
    check(lib.isc_embed_relu_bwd_ws(emb.data_ptr(), V, W, ids.data_ptr(), ids_stride, n_rows, rows_per_grad,
                                    pad_first, pad_id, dout.data_ptr(), scale, ptr(keep_mask), mask_scale,
                                    demb.data_ptr(), skip_id, ws.data_ptr(), ws.numel() * 4, stream()),
          'isc_embed_relu_bwd_ws')


def colsum(x, out, accumulate=False):
    lib = _lib.load()
    M, N = x.shape
    assert x.stride(1) == 1 and out.is_contiguous() and out.numel() >= N
    ws = splitk_ws(x.device)
    check(lib.isc_colsum(x.data_ptr(), x.stride(0), M, N, out.data_ptr(), int(accumulate), ws.data_ptr(),
                         ws.numel(), stream()), 'isc_colsum')


def colsum_multi(jobs):
    """jobs: list of (x [M,N], [out tensors, 1..3], accumulate) - every column sum in at most two launches per 24 jobs
    (isc_colsum_multi).  The caller keeps every x alive and untouched until this returns."""
    lib = _lib.load()
    if not jobs:
        return
    ws = splitk_ws(jobs[0][0].device)
    for lo in range(0, len(jobs), _lib.ISC_COLSUM_MAX_JOBS):
        part = jobs[lo:lo + _lib.ISC_COLSUM_MAX_JOBS]
        arr = (_lib.ColsumJob * len(part))()
        for q, (x, outs, acc) in zip(arr, part):
            M, N = x.shape
            assert x.stride(1) == 1 and x.dtype == torch.float32 and 1 <= len(outs) <= _lib.ISC_COLSUM_MAX_OUT
            q.x, q.ld, q.M, q.N, q.n_out, q.accumulate = x.data_ptr(), x.stride(0), M, N, len(outs), int(acc)
            for i, o in enumerate(outs):
                assert o.is_contiguous() and o.numel() >= N and o.dtype == torch.float32
                q.out[i] = o.data_ptr()
        check(lib.isc_colsum_multi(arr, len(part), ws.data_ptr(), ws.numel(), stream()), 'isc_colsum_multi')


def relu_mask_bwd(dy, y, dz, keep_mask=None, scale=1.0):
    """dz = dy * (y > 0) [* mask * scale]; y=None skips the ReLU test (pure dropout backward)."""
    lib = _lib.load()
    assert dy.is_contiguous() and dz.is_contiguous() and (y is None or y.is_contiguous())
    check(lib.isc_relu_mask_bwd(dy.data_ptr(), ptr(y), ptr(keep_mask), scale, dy.numel(), dz.data_ptr(),
                                stream()), 'isc_relu_mask_bwd')


def xe_loss_bwd(target, lengths_i32, gout, sum_count, dlogp):
    lib = _lib.load()
    B, T, V = dlogp.shape
    check(lib.isc_xe_loss_bwd(target.data_ptr(), lengths_i32.data_ptr(), B, T, V, gout.data_ptr(),
                              sum_count.data_ptr(), dlogp.data_ptr(), stream()), 'isc_xe_loss_bwd')


WEIGHT_EPOCH = 0   # bumped by every in-place parameter update done behind torch's back (version counters)


STATUS_NONFINITE_STATS, STATUS_NONFINITE_LINEAR = 1, 2


_STATUS_WORDS = {}          # device index -> pinned int32[2] the kernels flag into (isc_set_status_words)


def register_status_words(device=None):
    """Once per device: hand the library two pinned host words for its numerics flags.  Cheap to call again."""
    index = torch.device(device).index if device is not None else None
    if index is None:
        index = torch.cuda.current_device()
    w = _STATUS_WORDS.get(index)
    if w is None:
        w = torch.zeros(2, dtype=torch.int32).pin_memory()
        with torch.cuda.device(index):
            check(_lib.load().isc_set_status_words(w.data_ptr()), 'isc_set_status_words')
        _STATUS_WORDS[index] = w
    return w


def device_status(reset=True, device=None):
    """Numerics bits flagged by the work the host has ALREADY waited for (no device call, no synchronisation)."""
    w = register_status_words(device)
    bits = (STATUS_NONFINITE_STATS if int(w[0]) else 0) | (STATUS_NONFINITE_LINEAR if int(w[1]) else 0)
    if reset and bits:
        w.zero_()
    return bits


def check_numerics(where=''):
    """Raises HipLibraryError if a decode step or a split-f16 launch over caller data went non-finite since the last
    check (and clears the flags).  Reads two host words: it sees what the host has waited for - call it after a
    synchronisation or a host read of the results (the product does: beam search, the RL step)."""
    st = device_status(reset=True)
    if st:
        what = []
        if st & STATUS_NONFINITE_LINEAR:
            what.append('a linear layer over caller-supplied features produced non-finite values')
        if st & STATUS_NONFINITE_STATS:
            what.append('a decode step saw non-finite logits')
        raise _lib.HipLibraryError(
            '%s%s: an input left the split-f16 domain |x| < 65504 (or was NaN / inf). Results since the last check '
            'are invalid; scale the features, or run the exact-fp32 engine with ops.set_h3_mode(0).'
            % (where + ': ' if where else '', '; '.join(what)))


_OWNED_STREAMS = set()      # (device index, stream handle) of the streams this package's objects hold for themselves


def private_stream(device):
    """A torch.cuda.Stream that no other owner inside this package holds (and that is not the caller's current stream).
    torch hands out the 32 pool streams of a device round robin, so the 33rd Stream() IS the first again - and
    everything kept per stream here (split-K workspace, weight planes, weights-scope slot) would be shared with it.
    Owners: train_graph.XETrainGraph (two), a captioner's roll-out stream, the eager training step's side stream."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(idx).cuda_stream
    for attempt in range(2):
        for _ in range(48):
            st = torch.cuda.Stream(device=idx)
            key = (idx, st.cuda_stream)
            if key not in _OWNED_STREAMS and st.cuda_stream != cur:
                _OWNED_STREAMS.add(key)
                return st
        # every pool stream is held: owners that are garbage but sit in reference cycles (training-graph objects, captioners
        # with captured graphs) give theirs back in their finalizers - collect once and look again
        if attempt == 0 and not torch.cuda.is_current_stream_capturing():
            gc.collect()
    raise RuntimeError('no free stream left in the pool of device %d (%d held by this package)' % (idx, len(_OWNED_STREAMS)))


def release_stream_state(index, handle):
    """Forget everything kept per (device index, stream handle): split-K workspace, weight-plane buffer, the suspended
    weights scope and the library's scope slot.  For owners of private streams that go away (train_graph)."""
    key = (index, handle)
    cls = h3_weights_scope
    with cls._lock:
        if cls._depth.get(key, 0) != 0:
            return False
        cls._suspended.pop(key, None)
        cls._cold_begins.pop(key, None)
        _H3W_BUF.pop(key, None)
    _SPLITK_WS.pop(key, None)
    _OWNED_STREAMS.discard(key)
    _lib.load().isc_h3_weights_end(C.c_void_p(handle))
    return True


_REFRESH = threading.local()


class refresh_only:
    """`with ops.refresh_only(stream_handles):` - refresh_weight_planes() on this thread touches only the suspended
    scopes of these streams (a training iteration being captured into a HIP graph must not pull a stream that is not
    part of the capture into it; the other streams' scopes go stale with the epoch and rebuild on their next use)."""

    def __init__(self, handles):
        self.only = frozenset(handles)

    def __enter__(self):
        self.prev = getattr(_REFRESH, 'only', None)
        _REFRESH.only = self.only
        return self

    def __exit__(self, *exc):
        _REFRESH.only = self.prev
        return False


def refresh_weight_planes(epoch_before):
    """After an in-place weight update enqueued on the current stream (the fused optimiser; WEIGHT_EPOCH was
    `epoch_before` when it started): re-split the weights behind every suspended weights scope of this device whose key
    was valid until then, in place and in few batched launches (isc_h3_weights_refresh), and re-key it to the new epoch -
    the next forward / backward sweeps then RESUME their scope instead of rebuilding it with one split launch per weight
    operand (39 per XE iteration at B = 128).  A scope on another stream (the seq2seq unroll's side stream) is refreshed
    on ITS stream, after that stream has been made to wait for this one.  Returns the number of scopes refreshed."""
    cls = h3_weights_scope
    cur = torch.cuda.current_stream()
    index = cur.device.index
    only = getattr(_REFRESH, 'only', None)
    with cls._lock:
        items = [(k, wk) for k, wk in cls._suspended.items()
                 if k[0] == index and isinstance(wk, tuple) and wk and wk[-1] == epoch_before
                 and cls._depth.get(k, 0) == 0 and (only is None or k[1] in only)]
    lib = _lib.load()
    done = 0
    for key, wk in items:
        handle = key[1]
        other = torch.cuda.ExternalStream(handle, device=index) if handle != cur.cuda_stream else None
        if other is not None:
            other.wait_stream(cur)
        rc = lib.isc_h3_weights_refresh(C.c_void_p(handle))
        if other is not None and torch.cuda.is_current_stream_capturing():
            cur.wait_stream(other)             # a capture must end with every forked stream joined
        with cls._lock:
            if cls._suspended.get(key) == wk:
                if rc == 0:
                    cls._suspended[key] = wk[:-1] + (WEIGHT_EPOCH,)
                    done += 1
                else:
                    del cls._suspended[key]
    return done


def adam_hyper(lr, beta1, beta2, step):
    """{lr, 1 - beta1^step, sqrt(1 - beta2^step)} as isc_clamp_adam derives them (double arithmetic, then float)."""
    return [float(lr), 1.0 - beta1 ** float(step), (1.0 - beta2 ** float(step)) ** 0.5]


def clamp_adam(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, clip, step, hyper=None):
    """hyper: optional device float tensor {lr, bc1, bc2_sqrt} read by the kernel instead of lr / step (a launch that
    is captured into a HIP graph; adam_hyper() gives the values to store there before each replay)."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1
    lib = _lib.load()
    n = len(params)
    mk = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    for t in list(params) + list(grads) + list(exp_avg) + list(exp_avg_sq):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda
    numel = (C.c_int64 * n)(*[t.numel() for t in params])
    if hyper is not None:
        assert hyper.is_cuda and hyper.dtype == torch.float32 and hyper.numel() == 3 and hyper.is_contiguous()
        check(lib.isc_clamp_adam_hyper(mk(params), mk(grads), mk(exp_avg), mk(exp_avg_sq), numel, n, hyper.data_ptr(),
                                       beta1, beta2, eps, weight_decay, clip, stream()), 'isc_clamp_adam_hyper')
        return
    check(lib.isc_clamp_adam(mk(params), mk(grads), mk(exp_avg), mk(exp_avg_sq), numel, n, lr, beta1, beta2,
                             eps, weight_decay, clip, step, stream()), 'isc_clamp_adam')
