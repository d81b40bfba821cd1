"""Execution engine: NHWC-bf16 device tensors, a define-by-run tape per module region, and ONE
torch.autograd node per region.

Why a tape instead of one autograd.Function per op: the backward of a residual network is
HBM-bound, and the fan-in of gradients at every residual junction (`x += shortcut` in [timm]
Bottleneck/BasicBlock) is where an op-by-op autograd engine burns extra passes (a separate
add kernel, 3 x |S| bytes per block).  The tape knows the consumers of every tensor, so the
second gradient to arrive is accumulated by the producing kernel's epilogue (`accumulate=1`
in tok_conv_dgrad / tok_bn_bwd_apply / tok_maxpool3x3s2_bwd), and the masked incoming
gradient of a residual unit is handed to the shortcut branch without a copy.

PyTorch is used for: device memory (torch.empty), the current HIP stream, the autograd graph
between regions (backbone -> pooling -> head -> loss).  All arithmetic is in libtok_gfx950.so.
"""
import os
import threading
import weakref
from typing import List, Optional, Sequence

import torch

from .. import _C

BF16 = torch.bfloat16


# main-stream weight gradients run this many units late (Region.defer_wgrad); 0: in place.  Measured on ResNet-50 B=256: see
# DESIGN.md section 8
WGRAD_DEFER = int(os.environ.get('TOK_WGRAD_DEFER', '0'))   # measured: 0 -> 19.62, 3 -> 19.66, 5 -> 19.74, 8 -> 19.73 ms/step: off


# TOK_HOST_PROF=1: host seconds spent enqueuing each node class's backward (tools: where the launch thread goes)
HOST_PROF = os.environ.get('TOK_HOST_PROF', '0') == '1'
_host_prof = {}


def host_prof_report():
    return sorted(((k, n, t) for k, (n, t) in _host_prof.items()), key=lambda r: -r[2])


_raw_free = {}      # device index -> library HIP events ready for reuse (Region.raw_event); an event belongs to its device


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def stream_ptr():
    """The HIP stream every kernel of the calling thread is enqueued on (torch's current one).  Through the raw accessor:
    `torch.cuda.current_stream()` builds a Stream object via three Python frames of device-index resolution (9 us under
    cProfile, 800 calls per HRNet-W48 step = a tenth of that step's launch-thread time)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_stream_objs = {}


def cur_stream():
    """torch's current Stream object of the calling thread, through the raw accessor and a table of the Stream objects seen
    so far (main / side / branch / comm streams live as long as the process): `torch.cuda.current_stream()` costs the launch
    thread ~9 us, and the engine asks 950 times per HRNet-W48 step."""
    if _raw_stream is None or _raw_device is None:
        return torch.cuda.current_stream()
    dev = _raw_device()
    key = (dev, _raw_stream(dev))
    s = _stream_objs.get(key)
    if s is None:
        s = _stream_objs[key] = torch.cuda.current_stream()
    return s


class _StreamCtx:
    """`with torch.cuda.stream(s)` without its per-entry device bookkeeping (one device per process)."""
    __slots__ = ('s', 'prev')

    def __init__(self, s):
        self.s, self.prev = s, None

    def __enter__(self):
        self.prev = cur_stream()
        if self.prev is not self.s:
            torch.cuda.set_stream(self.s)
        return self.s

    def __exit__(self, *exc):
        if self.prev is not self.s:
            torch.cuda.set_stream(self.prev)
        return False


def require_device(t: torch.Tensor):
    if t.device.type != 'cuda':
        raise RuntimeError(
            f'torchok_amd executes on MI355X (HIP) devices only, got a tensor on "{t.device}". '
            f'There is no CPU path: move the task and the batch to cuda.')


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class TTensor:
    """Engine tensor: bf16, channel-last, channels padded to a multiple of 8.

    data: (N, H, W, Cp) or (N, Cp);  c: logical channel count (<= Cp)."""
    __slots__ = ('data', 'c', 'node', 'grad', 'grad_owned', 'requires_grad', 'uses', 'arrived', 'ready', 'gevents',
                 'sub_closers', 'grad_sub', 'colsum_part', 'affine', '__weakref__')

    def __init__(self, data: torch.Tensor, c: int, requires_grad: bool = False, node=None):
        self.data = data
        self.c = c
        self.node = node
        self.grad: Optional[torch.Tensor] = None
        self.grad_owned = True
        self.requires_grad = requires_grad
        self.uses = 0       # in-region consumers that will send a gradient (counted in forward)
        self.arrived = 0    # gradient contributions received so far (backward)
        self.ready = None   # (event, stream): produced on a branch stream (forward), see Region.branch
        self.gevents = None  # [(stream, unit number)]: who wrote gradient contributions (multi-stream backward)
        self.sub_closers = 0  # consumers whose data gradient can absorb a pending half-resolution contribution (forward)
        self.grad_sub = None  # pending contribution: gradient of the stride-2 pixel subsample of this tensor (backward)
        self.colsum_part = None  # (partial [rows][C] fp32, rows): per-block column sums left by the pass that produced `data`
        self.affine = None  # (scale, shift) fp32 [Cp]: `data` is a RAW conv output whose BatchNorm apply is left to the consumer (conv_bn_act defer_apply)

    @property
    def cp(self) -> int:
        return self.data.shape[-1]

    @property
    def shape(self):
        return tuple(self.data.shape)

    def rows(self) -> int:
        """M of the [M][Cp] matrix view."""
        return self.data.numel() // self.data.shape[-1]

    def torch_view(self) -> torch.Tensor:
        """Logical NCHW (or (N, C)) view of the buffer — zero copy."""
        d = self.data
        if d.dim() == 4:
            v = d.permute(0, 3, 1, 2)
            return v if self.c == self.cp else v[:, :self.c]
        return d if self.c == self.cp else d[:, :self.c]


# ---- gradient fan-in protocol ---------------------------------------------------------------

_ms = threading.local()      # multi-stream backward bookkeeping (Region.run_backward)


def _await(events):
    cur = cur_stream()
    for ev, s in events:
        if s != cur:
            cur.wait_event(ev)


def await_ready(*tensors):
    """Forward: make the current stream wait for tensors that were produced on a branch stream, and tell the caching
    allocator about the new reader: the buffers belong to the branch stream's pool, and without `record_stream` a buffer
    whose last reference dies on the launch thread (no-grad mode; operands no unit keeps for its backward, like a bias
    vector) is handed to the branch stream's NEXT allocation while the consumer's kernel is still queued.
    Arguments: TTensors, or objects with `ready` and a `tensors` tuple (Region.branch's publish)."""
    cur = None
    for t in tensors:
        if t is not None and t.ready is not None:
            _await((t.ready,))
            if cur is None:
                cur = cur_stream()
            if t.ready[1] != cur:
                for d in (t.tensors if hasattr(t, 'tensors') else (t.data,)):
                    if d is not None and d.is_cuda:
                        d.record_stream(cur)


def _sync_writers(writers, cur):
    """Multi-stream backward: make stream `cur` wait for the work that the streams in `writers` ([(stream, seq)]: a unit
    numbered `seq` of that stream's backward wrote something `cur` is about to touch) had been given.  Events are recorded
    LAZILY, at the first cross-stream consumer: a step whose units all sit on one stream records none (an event record
    costs its stream a few microseconds of bubble; one per unit was 300+ per transformer step), and an event that already
    covers the writer is reused.  The record point is later than the write, i.e. conservative."""
    for w, seq in writers:
        if w == cur:
            continue
        have = _ms.waited.get((cur, w), -1)
        if have >= seq:
            continue                              # `cur` already waited for an event of `w` that covers the write
        ev_seq, ev = _ms.events.get(w, (-1, None))
        now = _ms.seq.get(w, 0)
        if ev_seq < seq:
            ev = torch.cuda.Event()
            ev.record(w)
            ev_seq = now
            _ms.events[w] = (ev_seq, ev)
        cur.wait_event(ev)
        _ms.waited[(cur, w)] = ev_seq


def _touch(x: 'TTensor'):
    if getattr(_ms, 'active', False):
        if x.gevents:
            cur = cur_stream()
            _sync_writers(x.gevents, cur)         # an earlier contribution may still be in flight on another stream
            if x.grad is not None and any(w != cur for w, _ in x.gevents):
                x.grad.record_stream(cur)
        _ms.touched.append(x)


def written_mark():
    """Multi-stream backward: a token for "what the current unit has enqueued so far" (None on a single stream); a unit of
    another stream that reads it passes the token to `await_mark` first.  For hand-offs that do not travel through a
    TTensor's gradient (window attention -> position-bias unit)."""
    if getattr(_ms, 'active', False):
        cur = cur_stream()
        return (cur, _ms.seq.get(cur, 0))
    return None


def await_mark(mark):
    if mark is not None and getattr(_ms, 'active', False):
        _sync_writers((mark,), cur_stream())


def flush_sub(x: TTensor):
    """Expand a pending half-resolution contribution (see functional.subsample2) into the full gradient buffer: the
    stand-alone scatter, for arrivals that cannot absorb it in their own epilogue."""
    if x.grad_sub is None:
        return
    sub, x.grad_sub = x.grad_sub, None
    n, h, w, c = x.data.shape
    if x.grad is None:
        x.grad = torch.empty_like(x.data)
        x.grad_owned = True
        acc = 0
    else:
        if not x.grad_owned:
            x.grad = x.grad.clone()
            x.grad_owned = True
        acc = 1
    _C.check(_C.lib().tok_subsample2_bwd(ptr(sub), n, h, w, c, ptr(x.grad), acc, stream_ptr()), 'tok_subsample2_bwd')


def grad_target(x: TTensor, sub_ok: bool = False):
    """Buffer to write a gradient contribution of `x` into, and whether to accumulate.

    First arrival allocates (accumulate=0); later arrivals add in the producer's epilogue.
    A gradient tensor that came from outside the region (autograd grad_output) is never
    written in place: it is cloned first.
    sub_ok: the caller takes `x.grad_sub` (a pending half-resolution contribution) into its own kernel; everybody else
    gets it expanded first."""
    _touch(x)
    if x.grad_sub is not None and not sub_ok:
        flush_sub(x)
    x.arrived += 1
    if x.grad is None:
        x.grad = torch.empty_like(x.data)
        x.grad_owned = True
        return x.grad, 0
    if not x.grad_owned:
        x.grad = x.grad.clone()
        x.grad_owned = True
    return x.grad, 1


def donate_grad(x: TTensor, buf: torch.Tensor) -> bool:
    """Hand an already-computed gradient buffer to `x` without a copy (first arrival only)."""
    if x.grad is None:
        _touch(x)
        x.arrived += 1
        x.grad = buf
        x.grad_owned = True
        return True
    return False


def is_last_contribution(x: TTensor) -> bool:
    """True when the NEXT gradient contribution to `x` completes its gradient (all in-region
    consumers counted in forward have reported)."""
    return x.uses > 0 and x.arrived + 1 == x.uses


# ---- parameter gradient slots ------------------------------------------------------------------
# id(param) -> fp32 tensor with the parameter's shape/strides.  A ParamArena (engine/arena.py)
# pre-registers views of one flat buffer here so that optimizers / all-reduce see one range.
_grad_slots = {}


def register_grad_slot(p: torch.nn.Parameter, slot: torch.Tensor):
    _grad_slots[id(p)] = (weakref.ref(p), slot)


def grad_slot(p: torch.nn.Parameter) -> torch.Tensor:
    ent = _grad_slots.get(id(p))
    if ent is not None and ent[0]() is p and ent[1].shape == p.shape and ent[1].device == p.device \
            and ent[1].stride() == p.stride():
        return ent[1]
    slot = torch.empty_like(p, dtype=torch.float32)
    _grad_slots[id(p)] = (weakref.ref(p), slot)
    return slot


def param_grad_target(p: torch.nn.Parameter):
    """(slot, accumulate): where the kernel writes dParam.  Call commit_param_grad afterwards."""
    slot = grad_slot(p)
    if p.grad is None:
        return slot, 0
    if p.grad.data_ptr() == slot.data_ptr():
        return slot, 1
    # a foreign .grad tensor (set by user code): compute into the slot, add afterwards
    return slot, 2


# callbacks fired when a parameter's gradient for this backward pass has been enqueued
# (dist/ddp.py uses them to launch a bucket's all-reduce as soon as its last gradient exists)
param_grad_hooks = []


def commit_param_grad(p: torch.nn.Parameter, slot: torch.Tensor, mode: int):
    if mode == 0:
        p.grad = slot
    elif mode == 2:
        p.grad.add_(slot)
    for h in param_grad_hooks:
        h(p)


# ---- regions -----------------------------------------------------------------------------------

_anchors = {}
_tls = threading.local()
_side_streams = {}
_branch_streams = {}
BRANCH_STREAMS = os.environ.get('TOK_BRANCH_STREAMS', '1') == '1'
LAZY_EVENTS = os.environ.get('TOK_LAZY_EVENTS', '1') == '1'


# ---- streams on distinct hardware queues -------------------------------------------------------------------------------
# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES (4) hardware queues in creation order, and two streams on one
# queue run strictly one after the other.  With the gradient reducer's comm stream and RCCL's own streams created first, the
# engine's side stream landed on the MAIN stream's queue: the weight gradients of a SwinV2-T step then ran between the main
# chain's kernels instead of beside them (24.7 vs 21.7 ms/step; more hardware queues are no answer: 8 or 16 cost ResNet-50
# under the reducer 29.6 vs 18.5 ms and HRNet-W48 112 vs 77 ms).  Every engine / reducer stream is therefore PROBED when it
# is first needed: a candidate shares a queue with stream `a` iff an event recorded on the idle candidate completes only
# after a spin kernel given to `a` before it.  The pick never shares with the main stream if any of eight candidates
# avoids it, and shares with as few (and as recently picked) of the streams already handed out as possible.
PICK_STREAMS = os.environ.get('TOK_PICK_STREAMS', '1') == '1'
_picked = {}          # device -> streams handed out by pick_stream
_main_hint = {}       # device -> the stream regions are opened on (Region.input): what "beside the main stream" refers to


def note_main_stream(device=None) -> None:
    """Remember the CALLING stream as the main stream of `device` (unless one is known already).  GraphedTrainingStep calls
    this before its warm-up steps, which run on a stream of their own."""
    if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
        return
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if str(dev) not in _main_hint:
        _main_hint[str(dev)] = torch.cuda.current_stream(dev)


def _shares_queue(a: 'torch.cuda.Stream', b: 'torch.cuda.Stream') -> bool:
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.synchronize()
    b.synchronize()
    with torch.cuda.stream(a):
        e0.record(a)
        torch.cuda._sleep(600000)           # a few hundred microseconds of spinning
        ea.record(a)
    eb.record(b)
    a.synchronize()
    b.synchronize()
    return e0.elapsed_time(eb) > 0.5 * e0.elapsed_time(ea)


# TOK_STREAM_PRIO=-1: the picked (side / branch / comm) streams are created with HIGH priority (experiment, round 5)
_STREAM_PRIO = int(os.environ.get('TOK_STREAM_PRIO', '0'))


def pick_stream(device, main: Optional['torch.cuda.Stream'] = None, share_cost: Optional[int] = None) -> 'torch.cuda.Stream':
    """A new stream for work that is meant to run BESIDE the calling (main) stream and beside the streams picked before.
    `share_cost`: what it costs a LATER pick to land on this stream's hardware queue (default: 2 + how long ago it was handed
    out — HRNet's branches are handed out busiest first).  The weight-gradient side stream passes 1: in a process that ran a
    ResNet first (bench.py's secondary workloads) HRNet-W48's third branch stream otherwise shared a queue with the second one
    and left the side stream, which only its neck / head region uses, a queue of its own (67.2 vs 65.6 ms/step)."""
    if (not PICK_STREAMS or torch.cuda.is_current_stream_capturing()):
        return torch.cuda.Stream(device=device, priority=_STREAM_PRIO)
    with torch.cuda.device(device):
        if main is None:
            main = _main_hint.get(str(device)) or torch.cuda.current_stream()
        others = _picked.setdefault(str(device), [])
        best, best_cost = None, None
        for _ in range(8):
            c = torch.cuda.Stream(device=device, priority=_STREAM_PRIO)
            cost = 1000 * int(_shares_queue(main, c))
            if cost < 1000:
                # when sharing cannot be avoided (five streams on four queues: HRNet's three branch streams + the side
                # stream), share with the cheapest one
                cost += sum((w if w is not None else 2 + len(others) - i) * int(_shares_queue(o, c))
                            for i, (o, w) in enumerate(others))
            if best is None or cost < best_cost:
                best, best_cost = c, cost
            if cost == 0:
                break
        others.append((best, share_cost))
    return best


# TOK_BRANCH_MAP="0,1,2,2": branch index -> stream slot (experiment: HRNet's four branches + the side stream are five streams on
# four hardware queues; folding two of the small branches onto one stream gives every stream a queue of its own)
_BRANCH_MAP = [int(v) for v in os.environ['TOK_BRANCH_MAP'].split(',')] if os.environ.get('TOK_BRANCH_MAP') else None


def _branch_stream(device, idx: int) -> 'torch.cuda.Stream':
    if _BRANCH_MAP is not None and idx < len(_BRANCH_MAP):
        idx = _BRANCH_MAP[idx]
    key = (device, idx)
    s = _branch_streams.get(key)
    if s is None:
        s = pick_stream(device)
        if PICK_STREAMS and torch.cuda.is_current_stream_capturing():
            return s     # an unprobed stream (no probing under capture): not remembered, the first eager use picks again
        _branch_streams[key] = s
    return s


class _Branch:
    def __init__(self, region: 'Region', idx: int, fork: bool = True):
        self.region, self.idx, self.out, self.ctx, self.prev, self.fork = region, idx, [], None, 0, fork

    def publish(self, *tensors):
        self.out.extend(t for t in tensors if t is not None)
        return tensors[0] if len(tensors) == 1 else tensors

    def __enter__(self):
        r = self.region
        dev = r.device
        self.live = bool(BRANCH_STREAMS and self.idx and dev is not None and dev.type == 'cuda'
                         and not torch.cuda.is_current_stream_capturing())
        if self.live:
            b = _branch_stream(dev, self.idx)
            if self.fork or self.idx not in r._streams:
                ev = torch.cuda.Event()
                ev.record(cur_stream())
                b.wait_event(ev)
            r._streams[self.idx] = b
            self.prev, r._tag = r._tag, self.idx
            self.ctx = _StreamCtx(b)
            self.ctx.__enter__()
            self.stream = b
        return self

    def __exit__(self, *exc):
        if self.live:
            self.ctx.__exit__(*exc)
            r = self.region
            r._tag = self.prev
            done = torch.cuda.Event()
            done.record(self.stream)
            r._branch_done.append((done, self.stream))
            for t in self.out:
                t.ready = (done, self.stream)
        return False


def _side_stream(device) -> 'torch.cuda.Stream':
    s = _side_streams.get(device)
    if s is None:
        s = pick_stream(device, share_cost=1)
        if PICK_STREAMS and torch.cuda.is_current_stream_capturing():
            return s     # see _branch_stream
        _side_streams[device] = s
    return s


def _anchor(device) -> torch.Tensor:
    a = _anchors.get(device)
    if a is None:
        a = torch.zeros(1, device=device, requires_grad=True)
        _anchors[device] = a
    return a


# data_ptr -> row length of zero-padded 2-D gradient buffers produced by our own loss kernels
_padded_rows = {}


def mark_padded(buf: torch.Tensor):
    """Remember that `buf` is one of OUR zero-padded channel-last buffers (its logical view may re-enter a region without a
    copy; one buffer may enter several regions, so entries are not consumed on the forward path).  The table is bounded
    (oldest half dropped at 4096 entries: a dropped entry only costs a copy), and an address recycled for another tensor is
    additionally caught by the stride test at the point of use."""
    if len(_padded_rows) >= 4096:
        for k in list(_padded_rows)[:2048]:      # oldest first (insertion order)
            _padded_rows.pop(k, None)
    _padded_rows.pop(buf.data_ptr(), None)
    _padded_rows[buf.data_ptr()] = buf.shape[-1]


class Node:
    """One fused unit on the tape."""
    needs_backward = False
    region = None      # set by Region.add
    stream_tag = 0     # branch stream the unit's BACKWARD runs on (0 = main): the stream it was recorded on unless re-tagged
    fwd_tag = 0        # branch stream the unit was recorded on: its saved tensors come from THAT stream's allocator pool

    def backward(self):
        raise NotImplementedError

    def release(self):
        pass


class Region:
    """A module-level execution scope: torch tensors in, torch tensors out, a tape in between."""

    def __init__(self):
        self.nodes: List[Node] = []
        self.inputs = []  # (TTensor, torch.Tensor) pairs whose torch side requires grad
        self.grad_mode = torch.is_grad_enabled()
        self.device = None
        self._side = None
        self._deferred = []
        self._wq = []          # main-stream weight-gradient launches held back (defer_wgrad)
        self._raw_used = []    # library events handed out since the last join (raw_event)
        self._tag = 0            # branch stream index of the units being recorded (0 = main)
        self._streams = {}       # branch index -> stream, for the branches this region used
        self._branch_done = []   # forward: (event, stream) of every closed branch, joined in output()

    # -- entry ---------------------------------------------------------------------------------
    def input(self, x: torch.Tensor, c_pad_to: int = 8) -> TTensor:
        require_device(x)
        self.device = x.device
        if x.is_cuda and self._tag == 0 and str(x.device) not in _main_hint and not torch.cuda.is_current_stream_capturing():
            # (a capture or its warm-up runs on a stream of its own: that one must not become "the main stream" the
            # side / branch / comm streams are probed against for the rest of the process)
            _main_hint[str(x.device)] = cur_stream()
        need = self.grad_mode and x.requires_grad
        if x.dim() == 4:
            n, c, h, w = x.shape
            cp = (c + c_pad_to - 1) // c_pad_to * c_pad_to
            cp8 = pad8(c)
            if x.dtype == BF16 and cp == c and x.permute(0, 2, 3, 1).is_contiguous():
                data = x.detach().permute(0, 2, 3, 1)
            elif x.dtype == BF16 and cp8 != c and x.stride() == (h * w * cp8, 1, w * cp8, cp8) \
                    and _padded_rows.get(x.data_ptr()) == cp8:
                # the logical view of one of our own zero-padded NHWC buffers (e.g. an 18-channel HRNet branch)
                data = torch.as_strided(x.detach(), (n, h, w, cp8), (h * w * cp8, w * cp8, cp8, 1))
            else:
                if need:
                    raise RuntimeError('torchok_amd: a 4-D region input that requires grad must already be '
                                       'channels-last bf16 with channels % 8 == 0 (produced by torchok_amd modules)')
                if x.dtype not in _DT:
                    raise TypeError(f'unsupported image dtype {x.dtype}')
                src = x.detach().contiguous()
                data = torch.empty((n, h, w, cp), dtype=BF16, device=x.device)
                _C.check(_C.lib().tok_nchw_to_nhwc_bf16(ptr(src), _DT[x.dtype], n, c, h, w, ptr(data), cp,
                                                        stream_ptr()), 'tok_nchw_to_nhwc_bf16')
            t = TTensor(data, c, requires_grad=need)
        elif x.dim() == 2:
            n, c = x.shape
            cp = pad8(c)
            if x.dtype == BF16 and x.stride(1) == 1 and (
                    (cp == c and x.stride(0) == c) or
                    (x.stride(0) == cp and _padded_rows.get(x.data_ptr()) == cp)):
                data = x.detach() if cp == c else torch.as_strided(x.detach(), (n, cp), (cp, 1))
            else:
                data = torch.zeros((n, cp), dtype=BF16, device=x.device)
                data[:, :c] = x.detach()
            t = TTensor(data, c, requires_grad=need)
        else:
            raise ValueError(f'torchok_amd regions take (N,C,H,W) or (N,C) tensors, got {tuple(x.shape)}')
        if need:
            self.inputs.append((t, x))
        return t

    def add(self, node: Node):
        node.region = self
        node.stream_tag = node.fwd_tag = self._tag
        self.nodes.append(node)

    # -- branch streams ----------------------------------------------------------------------------
    def branch(self, idx: int, fork: bool = True):
        """Context manager: the units recorded inside run on branch stream `idx` (forward now, their backward later),
        concurrently with what the main stream is given meanwhile.  For sub-graphs that are independent of the main
        chain: the projection shortcut of a residual block, the parallel branches of an HRNet module, the parameter-only
        position-bias chain of a SwinV2 block.  Publish what leaves the branch with `.publish(...)` (TTensors, or any
        object with a `ready` attribute): consumers wait for the branch through `await_ready`.
        fork=False: the units read nothing the main stream produced inside this region (parameters only), so only the
        region's first entry into the stream is ordered behind the main stream — no event record on the main queue.
        No-op on the host stand-in, inside a hipGraph capture, for idx 0 or with TOK_BRANCH_STREAMS=0."""
        return _Branch(self, idx, fork)

    def _join_forward(self):
        if self._branch_done:
            cur = cur_stream()
            for ev, b in self._branch_done:
                if b != cur:
                    cur.wait_event(ev)
            self._branch_done = []

    # -- exit ----------------------------------------------------------------------------------
    def output(self, *outs: TTensor):
        self._join_forward()
        need = self.grad_mode and any(o.requires_grad for o in outs)
        if not need:
            for n in self.nodes:
                n.release()
            self.nodes = []
            res = tuple(o.torch_view() for o in outs)
            for o in outs:
                if o.data.dim() == 4 and o.c != o.cp:
                    mark_padded(o.data)
        else:
            tins = [x for _, x in self.inputs]
            res = _RegionFn.apply(self, list(outs), _anchor(self.device), *tins)
        return res[0] if len(res) == 1 else res

    def run_backward(self):
        if self._streams and not torch.cuda.is_current_stream_capturing():
            return self._run_backward_multi()
        for node in reversed(self.nodes):
            if node.needs_backward:
                out = getattr(node, 'out', None)
                if out is not None and out.grad_sub is not None:
                    flush_sub(out)         # the consumer that would have absorbed it never reported (dead branch)
                if HOST_PROF:
                    import time
                    t0 = time.perf_counter()
                    node.backward()
                    e = _host_prof.setdefault(type(node).__name__, [0, 0.0])
                    e[0] += 1
                    e[1] += time.perf_counter() - t0
                else:
                    node.backward()
            node.release()
        self.nodes = []
        self.join_side()

    def _run_backward_multi(self):
        """Backward of a region that used branch streams: every unit runs on the stream it was recorded on; a unit waits
        for the gradient contributions other streams wrote into its output (`_sync_writers`: events are recorded lazily,
        only for edges that cross streams).  Units of branch streams are released only after the join (their tensors may
        be in flight)."""
        main = cur_stream()
        ev0 = torch.cuda.Event()
        ev0.record(main)
        for b in self._streams.values():
            b.wait_event(ev0)
        held = []
        _ms.active, _ms.touched = True, []
        _ms.seq, _ms.events, _ms.waited = {}, {}, {}
        cur = main
        try:
            for node in reversed(self.nodes):
                s = self._streams.get(node.stream_tag, main) if node.stream_tag else main
                if node.needs_backward:
                    if s is not cur:
                        torch.cuda.set_stream(s)
                        cur = s
                    _ms.seq[s] = seq = _ms.seq.get(s, 0) + 1
                    out = getattr(node, 'out', None)
                    if out is not None and out.gevents:
                        _sync_writers(out.gevents, s)
                        if out.grad is not None and any(w != s for w, _ in out.gevents):
                            out.grad.record_stream(s)
                    _ms.touched = []
                    if out is not None and out.grad_sub is not None:
                        flush_sub(out)
                    node.backward()
                    # a gradient written for a producer that sits on another stream: that edge is known now, so its event
                    # is recorded here, right behind the writer (precise: HRNet's fuse layers, 78.8 -> 77.5 ms/step against
                    # the lazy record alone).  Anything else gets its event at the first cross-stream reader, if ever.
                    if _ms.touched and (not LAZY_EVENTS or any(
                            t.node is not None and getattr(t.node, 'stream_tag', 0) != node.stream_tag for t in _ms.touched)):
                        ev = torch.cuda.Event()
                        ev.record(s)
                        _ms.events[s] = (seq, ev)
                    for t in _ms.touched:
                        if t.gevents is None:
                            t.gevents = []
                        t.gevents.append((s, seq))
                # a unit's saved tensors go back to the allocator pool of the stream that RECORDED it; released here, with the
                # backward kernel that reads them still queued on another stream, that stream's next allocation may hand the
                # memory out again (round 6: HRNet's fuse paths recorded on a row's stream with their backward re-tagged to the
                # main stream — parameters differed run to run in the last bits).  Only main-recorded, main-run units go now.
                if s is main and not node.fwd_tag:
                    node.release()
                else:
                    held.append(node)
        finally:
            if cur is not main:
                torch.cuda.set_stream(main)
            _ms.active, _ms.touched = False, []
            _ms.seq, _ms.events, _ms.waited = {}, {}, {}
        for b in self._streams.values():
            main.wait_stream(b)
        self.nodes = []
        self.join_side()
        for node in held:
            node.release()

    # -- side stream (weight gradients run beside the main chain) --------------------------------
    def mark_side(self):
        """An event on the main stream NOW, for a later fork_side(..., event=...): the side kernels are then ordered after
        the main stream's work up to this point only, although the host enqueues them after further main-stream launches."""
        ev = torch.cuda.Event()
        ev.record(cur_stream())
        return ev

    def raw_event(self):
        """A HIP event from the library's pool, for tok_next_launch_event + fork_side(raw_event=...); recycled at the join."""
        dev = torch.cuda.current_device()
        free = _raw_free.get(dev)
        ev = free.pop() if free else _C.lib().tok_event_create()
        if not ev:
            raise RuntimeError('tok_event_create failed')
        self._raw_used.append((dev, ev))
        return ev

    def fork_side(self, keep_alive, event=None, raw_event=None):
        """Context manager: kernels enqueued inside run on this device's side stream, ordered after everything the
        main stream has been given so far (or up to `event`, see mark_side).  `keep_alive` (tensors the side kernels read
        or use as scratch) stay referenced until the join, so the allocator cannot hand their memory to later main-stream
        work."""
        main = cur_stream()
        side = _side_stream(main.device)
        if raw_event is not None:
            # the event is signalled by the completion of a kernel already launched on the main stream (no record packet)
            _C.check(_C.lib().tok_stream_wait_event(side.cuda_stream, raw_event), 'tok_stream_wait_event')
        else:
            ev = event
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(main)
            side.wait_event(ev)
        self._side = (main, side)
        self._deferred.extend(keep_alive)
        return _StreamCtx(side)

    def keep_until_join(self, *tensors):
        self._deferred.extend(tensors)

    # -- held-back weight gradients ------------------------------------------------------------------
    def defer_wgrad(self, fn):
        """Run `fn` (a main-stream weight-gradient launch; nothing downstream waits for its result) WGRAD_DEFER units later
        than its place in the tape.  The last ones are therefore still pending when the backward walk reaches the region
        input, and run beside the side stream's final launches (the stem's issue-bound weight gradient in a ResNet) instead
        of leaving the main stream idle until the join."""
        if WGRAD_DEFER <= 0 or torch.cuda.is_current_stream_capturing():
            fn()
            return
        self._wq.append(fn)
        if len(self._wq) > WGRAD_DEFER:
            self._wq.pop(0)()

    def flush_wgrads(self):
        while self._wq:
            self._wq.pop(0)()

    def join_side(self):
        self.flush_wgrads()
        if self._side is not None:
            cur_stream().wait_stream(self._side[1])
            self._side = None
        if self._raw_used:
            for dev, ev in self._raw_used:        # the main stream is behind every waiter now: the events can be reused
                _raw_free.setdefault(dev, []).append(ev)
            self._raw_used = []
        self._deferred.clear()


_DT = {torch.float32: _C.TOK_F32, torch.float16: _C.TOK_F16, torch.bfloat16: _C.TOK_BF16}


def _grad_to_engine(t: TTensor, g: torch.Tensor) -> torch.Tensor:
    """Bring an autograd grad_output into the engine layout of `t` (zero copy when it already is)."""
    if g.dim() == 4:
        gp = g.permute(0, 2, 3, 1)
        if g.dtype == BF16 and t.c == t.cp and gp.is_contiguous():
            return gp
        n, h, w, cp = t.data.shape
        if g.dtype == BF16 and gp.stride() == (h * w * cp, w * cp, cp, 1) \
                and _padded_rows.pop(g.data_ptr(), None) == cp:
            return torch.as_strided(g, (n, h, w, cp), (h * w * cp, w * cp, cp, 1))   # our own zero-padded rows
        buf = torch.zeros_like(t.data)
        buf[..., :t.c] = gp
        return buf
    n, c = g.shape
    if g.dtype == BF16 and g.stride(1) == 1:
        if t.c == t.cp and g.stride(0) == c:
            return g
        if g.stride(0) == t.cp and _padded_rows.pop(g.data_ptr(), None) == t.cp:
            return torch.as_strided(g, (n, t.cp), (t.cp, 1))
    buf = torch.zeros_like(t.data)
    buf[:, :t.c] = g
    return buf


class _RegionFn(torch.autograd.Function):
    """The single autograd node of a region.  Parameter gradients are written straight into
    their fp32 slots (and `.grad` is pointed at them), so only activation inputs are returned."""

    @staticmethod
    def forward(ctx, region: Region, outs: Sequence[TTensor], anchor, *tins):
        ctx.region = region
        ctx.outs = outs
        ctx.set_materialize_grads(False)
        for o in outs:
            if o.data.dim() == 4 and o.c != o.cp:
                mark_padded(o.data)      # lets the next region take the padded buffer without a copy
        return tuple(o.torch_view() for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        region: Region = ctx.region
        for o, g in zip(ctx.outs, gouts):
            if g is None:
                continue
            ge = _grad_to_engine(o, g)
            if o.grad is None:
                o.grad = ge
                o.grad_owned = False
            else:  # same tensor returned twice from the region
                tgt, _ = grad_target(o)
                tgt.add_(ge)
        region.run_backward()
        gins = []
        for t, x in region.inputs:
            flush_sub(t)
            if t.grad is None:
                gins.append(None)
            else:
                gt = TTensor(t.grad, t.c)
                gins.append(gt.torch_view())
                if t.c != t.cp:
                    mark_padded(t.grad)
        ctx.region = None
        ctx.outs = None
        return (None, None, None, *gins)
