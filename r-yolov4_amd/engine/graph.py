"""Static-graph executor for the conv stack — the MI355X-first replacement for running the
reference's nn.Module.forward + autograd op by op.

For a fixed (batch, image size, train/eval) the whole backbone + neck is planned ONCE into two flat tapes of C-ABI kernel
launches (forward, backward) over pre-allocated NHWC bf16 buffers:
  * no allocation, no host sync and no Python tensor ops inside a step -> both tapes are hipGraph-capturable;
  * torch.cat never happens: producers are planned to write straight into their channel slice of the concat buffer;
  * gradients: "first writer stores, later writers accumulate" is resolved at plan time per channel range, so there is
    no zero-fill of activation gradients and no autograd bookkeeping;
  * parameter gradients accumulate into one flat fp32 buffer (the unit of the RCCL all-reduce and of the fused SGD step).
PyTorch only owns the memory (torch.empty) and the stream.
"""
import ctypes as C
import os
import sys

import torch

from .. import hip
from . import arena
from . import structs as S

BF16 = torch.bfloat16


_HEAD_SPARSE = os.environ.get("RYOLO_HEAD_SPARSE", "1") != "0"
_HEAD_FUSED = os.environ.get("RYOLO_HEAD_FUSED", "1") != "0"
_HEAD_FOLD_A = os.environ.get("RYOLO_HEAD_FOLD_A", "1") != "0"
_DEBUG_SKIP_SIDE = os.environ.get("RYOLO_DEBUG_SKIP_SIDE") == "1"     # tools only: never set in a run that reports numbers
if _DEBUG_SKIP_SIDE:
    import warnings
    warnings.warn("RYOLO_DEBUG_SKIP_SIDE=1: every weight-gradient launch is SKIPPED — gradients of this process are garbage; "
                  "timing diagnostic only (tools/), never for training or for a reported number", RuntimeWarning, stacklevel=1)
    print("ryolov4_amd: RYOLO_DEBUG_SKIP_SIDE=1 — weight gradients are NOT computed in this process (diagnostic mode)", file=sys.stderr, flush=True)

class Buf:
    """[N*H*W, C] bf16 activation buffer (NHWC, channel stride = C) and, lazily, its gradient twin."""

    def __init__(self, N, H, W, Cc, device, graph=None):
        self.N, self.H, self.W, self.C = N, H, W, Cc
        self.graph = graph
        self.k = None
        if graph is not None:        # member of a plan: dedicated memory, a virtual address (dry pass) or a slot of the plan's arena
            self.k = graph._nbuf
            graph._nbuf += 1
        self.t = self._storage(0, device)
        self.g = None
        self.g_written = []          # channel ranges already written during the planned backward

    def _storage(self, which, device):
        g, shape = self.graph, (self.N * self.H * self.W, self.C)
        nbytes = shape[0] * shape[1] * 2
        if g is not None and g.dry:
            v = g._virt[2 * self.k + which] = arena.Virt(2 * self.k + which, nbytes)
            return v
        if g is not None and g.layout is not None:
            off = g.layout.off.get(2 * self.k + which)
            if off is None or g.layout.sizes[2 * self.k + which] != nbytes:
                raise RuntimeError("engine: the second planning pass asked for a buffer the liveness pass did not see")
            return g.arena[off:off + nbytes].view(BF16).view(shape)
        return torch.empty(shape, dtype=BF16, device=device)

    def grad_tensor(self):
        if self.g is None:
            self.g = self._storage(1, self.t.device if isinstance(self.t, torch.Tensor) else None)
        return self.g


class TRef:
    """Channel slice [c0, c0+C) of a Buf."""

    def __init__(self, buf, c0=0, Cc=None):
        self.buf, self.c0, self.C = buf, c0, buf.C - c0 if Cc is None else Cc
        assert self.c0 % 8 == 0 and self.C % 8 == 0 and self.c0 + self.C <= buf.C

    N = property(lambda s: s.buf.N)
    H = property(lambda s: s.buf.H)
    W = property(lambda s: s.buf.W)
    ld = property(lambda s: s.buf.C)
    M = property(lambda s: s.buf.N * s.buf.H * s.buf.W)

    def ptr(self):
        return self.buf.t.data_ptr() + 2 * self.c0

    @property
    def span_bytes(self):
        return (self.M * self.ld - self.c0) * 2

    def gptr(self):
        return self.buf.grad_tensor().data_ptr() + 2 * self.c0

    def slice(self, c0, Cc):
        return TRef(self.buf, self.c0 + c0, Cc)

    def grad_write_mode(self):
        """Plan-time: returns 1 (accumulate) if [c0,c0+C) of the grad buffer was already written, else 0 (store)."""
        lo, hi = self.c0, self.c0 + self.C
        covered = any(a <= lo and hi <= b for a, b in self.buf.g_written)
        if not covered:
            for a, b in self.buf.g_written:
                if not (hi <= a or b <= lo):
                    raise RuntimeError("engine: partially overlapping gradient writes (unsupported plan)")
            self.buf.g_written.append((lo, hi))
        return 1 if covered else 0

    def to_nchw(self, grad=False):
        t = self.buf.g if grad else self.buf.t
        return t.view(self.N, self.H, self.W, self.ld)[..., self.c0:self.c0 + self.C].permute(0, 3, 1, 2).float().contiguous()


def _taps_fwd(k, pad):
    return [(r - pad, s - pad, r * k + s) for r in range(k) for s in range(k)]


def _fill_class(tc, taps, oh_add=0, ow_add=0):
    tc.ntaps = len(taps)
    tc.oh_add, tc.ow_add = oh_add, ow_add
    for i, (dh, dw, wi) in enumerate(taps):
        tc.dh[i], tc.dw[i], tc.widx[i] = dh, dw, wi


class Graph:
    def __init__(self, rt, B, Hin, Win, training, frozen=False, dry=False, layout=None):
        self.rt, self.B, self.Hin, self.Win, self.training = rt, B, Hin, Win, training
        # buffer placement (engine/arena.py): dry = liveness pass on virtual addresses; layout = its result, the buffers are slots of ONE arena
        self.dry, self.layout = dry, layout
        self._nbuf, self._virt, self._raw = 0, {}, {}
        self.arena = torch.empty(layout.total, dtype=torch.uint8, device=rt.device) if layout is not None else None
        self._done = None                      # completion events of the weight-gradient launches (bounded lag, see run())
        self.frozen = frozen                   # backward tape with eval-mode (running-statistics) BatchNorm
        self.batch_stats = training and not frozen
        self.dev = rt.device
        self.adev = torch.device("meta") if dry else rt.device      # the dry pass allocates nothing: sizes and (null) addresses only
        self.fwd, self.bwd = [], []            # tapes: lists of (fn_name, args...) closures
        self.keep = []                         # tensors/structs kept alive
        self.side_idx = set()                  # backward-tape entries launched on the weight-gradient stream
        self.fwd_side = {}                     # forward-tape index -> True for the first entry of a forked branch, False for the rest
        self.fwd_join = set()                  # forward-tape indices before which the main stream joins the forked branch
        self.serial = False                    # True: everything on the main stream (per-kernel timing passes)
        self._side = None                      # (stream, event pool)
        self._side2 = None                     # (stream, join event) of forward lane 2
        self.stream = None
        self.img = torch.empty((B, 3, Hin, Win), dtype=torch.float32, device=self.adev)   # staging of the input batch
        self._img_slot, self._img_structs = None, []
        self.heads = []                        # per scale: dict(out=fp32 tensor, dout=fp32 tensor)
        self.debug = {}
        self.meta = {}                         # (tape id, index) -> (kernel class, algorithmic flops)
        self.timer = None                      # set by bench.py: per-launch HIP-event timing of the conv kernels
        self._wgrads, self._wgrad_ws_bytes = {}, {}  # per lane: split-K workspace shared by the weight-gradient launches of ONE stream (see _emit_wgrad)
        self.grad_writes = []                  # (backward tape index, [element offsets into Runtime.gflat it writes])
        self.wprep = []                        # launches that depend on the weights only (eval-mode BN folding); run before fwd
        self._fork_open = False                # a side_branch() was emitted since the last join_side()
        self.static_weights = False            # True while a captured inference graph is recorded: pack + wprep already done
        # test instrumentation (tests/test_gpu_teacher_forced.py): with Runtime.record_tape the argument objects of every launch stay
        # inspectable, and `probe` = {(id(tape), index): (before, after)} runs host callables around single launches of a replay
        self.tape_args = {} if getattr(rt, "record_tape", False) else None
        self.probe = None

    # ------------------------------------------------------------------ helpers
    def new(self, N, H, W, Cc):
        b = Buf(N, H, W, Cc, self.dev, self)
        self.keep.append(b)
        return TRef(b)

    def f32(self, *shape, zero=False):
        t = (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=self.adev)
        self.keep.append(t)
        return t

    def side_branch(self, lane=1):
        """`with g.side_branch(): <emit one branch of a block>` — the forward launches emitted inside run on the second stream, forked
        from the main stream at the point of entry (they may read anything produced before it) while the main stream goes on with the
        sibling branch; `g.join_side()` makes the main stream wait for them (call it before the consumer of the branch's output is
        emitted).  Branches write disjoint channel slices of the concat buffer and own their statistics buffers: no ordering
        between them is needed.  Backward is unaffected (its tape is serial on the main stream but for the weight gradients)."""
        g = self

        class _Ctx:
            def __enter__(self_):
                self_.start = len(g.fwd)
                g._fork_open = True

            def __exit__(self_, *exc):
                if g.rt.fwd_fork:
                    for i in range(self_.start, len(g.fwd)):
                        g.fwd_side[i] = (i == self_.start, lane)
                return False
        return _Ctx()

    def join_side(self):
        if self.fwd_side and self._fork_open:
            self.fwd_join.add(len(self.fwd))
        self._fork_open = False

    def _call(self, tape, name, *args):
        """args may contain ctypes structs (passed by reference); the stream is appended at run time."""
        conv = tuple(C.byref(a) if isinstance(a, C.Structure) else a for a in args)
        self.keep.extend(a for a in args if isinstance(a, C.Structure))
        fn = getattr(hip.lib(), name)
        tape.append((fn, conv, name))
        if self.dry:
            self._raw[(id(tape), len(tape) - 1)] = args
        if self.tape_args is not None:
            self.tape_args[(id(tape), len(tape) - 1)] = (name, args)
        self.meta[(id(tape), len(tape) - 1)] = self._describe(name, args)
        if tape is self.bwd:
            # which bytes of the flat gradient buffer this launch writes (data-parallel overlap: a bucket is reduced as soon as
            # the last launch that touches it has been enqueued) — raw pointer arguments and the dW field of the wgrad blocks
            lo = self.rt.gflat.data_ptr()
            hi = lo + self.rt.gflat.numel() * 4
            ptrs = [a for a in args if isinstance(a, int) and lo <= a < hi]
            for a in args:
                if isinstance(a, (S.WgradParams, S.StemBwdParams)):
                    for f in ("dW", "dW2", "dgamma", "dbeta"):
                        q = getattr(a, f, None)
                        if q and lo <= q < hi:
                            ptrs.append(q)
            if ptrs:
                self.grad_writes.append((len(tape) - 1, [(q - lo) // 4 for q in ptrs]))

    @staticmethod
    def _describe(name, args):
        """(kernel class, algorithmic FLOPs, algorithmic HBM bytes, pointwise regime | None, CUs held | None, layer family | None) of a
        launch — used by bench.py's live roofline accounting.  Bytes = every operand element once: gathered input + weights + output
        (twice for accumulate epilogues).  Layer family: "3x3s1" / "3x3s2" for the forward, data-gradient and weight-gradient launches of
        3x3 convolutions (BASELINE's north star prices ALL 3x3 work, whichever kernel serves it)."""
        d = Graph._describe_kernel(name, args)
        if len(d) >= 3 and d[1]:
            fam = None
            p = args[0]
            if name == "ryolo_conv_gemm" and (p.wtaps == 9 or p.s2d_cin):
                # forward: sh == 2; data gradient of a stride-2 layer: four output-parity classes (oh_mul == 2) or the space-to-depth GEMM
                fam = "3x3s2" if (p.sh == 2 or p.oh_mul == 2 or p.s2d_cin) else "3x3s1"
            elif name == "ryolo_conv_wgrad" and p.ntaps == 9:
                fam = "3x3s2" if p.sh == 2 else "3x3s1"
            d = tuple(d) + (None,) * (5 - len(d)) + (fam,)
        return d

    @staticmethod
    def _describe_kernel(name, args):
        if name == "ryolo_conv_gemm":
            p = args[0]
            fl = 0
            for c in range(p.nclasses):
                fl += 2 * p.NB * p.OH * p.OW * p.Nout * p.cls[c].ntaps * p.Cin
            if p.s2d_cin:                                  # space-to-depth dgrad: 9 of the 16 (parity, tap) blocks are live — count the useful work
                fl = 2 * p.NB * p.OH * p.OW * p.s2d_cin * 9 * p.Cin
            outb = p.NB * p.OHf * p.OWf * (p.s2d_cin or p.Nout) * (4 if p.epi == S.EPI_F32_BIAS else 2) * (2 if p.epi == S.EPI_ACCUM else 1)
            by = p.NB * p.IH * p.IW * p.Cin * 2 + p.Nout * p.wtaps * p.Cin * 2 + outb
            kern = S.I()
            hip.call("ryolo_conv_gemm_plan", p, S.I(), kern)
            fam, kv = kern.value & 0xff, kern.value
            if fam == 1:
                return (f"conv3x3_patch_kernel<256x{((kv >> 16) & 15) * 32}>", fl, by)
            if fam == 2:
                # plain pointwise launches of the persistent kernel (on by size since r04) belong to the K <= 256 / K > 256 split of bench.py
                # like the generic kernel's 1x1 instantiation; its tapped / pool-gradient instantiations do not
                if p.nclasses == 1 and p.cls[0].ntaps == 1 and not p.s2d_cin and not p.pool_idx:
                    return ("gemm1x1_ws_kernel", fl, by, "K<=256" if p.Cin <= 256 else "K>256")
                return ("gemm1x1_ws_kernel", fl, by)
            if fam == 3:
                return ("conv3x3_ws64_kernel", fl, by)
            if fam == 5:
                return ("conv3x3s2_c32_kernel", fl, by)
            if fam == 6:
                return ("conv3x3s2_c32_dgrad_kernel", fl, by)
            if fam == 4:
                return (f"gemm256_kernel<256x{((kv >> 16) & 15) * 32}>", fl, by, "K<=256" if p.Cin <= 256 else "K>256")
            # the generic kernel's instantiations as rocprofv3 lists them: tile shape, and the 1x1 form (no tap table / tile decomposition)
            kind = f"conv_gemm_kernel<{((kv >> 12) & 15) * 64}x{((kv >> 16) & 15) * 32}{',1x1' if kv & 0x100 else ''}>"
            if kv & 0x100:
                # the pointwise class holds two regimes: short-K layers are memory streams, K > 256 layers are MFMA-side;
                # the fourth field lets bench.py report a roofline for each
                return (kind, fl, by, "K<=256" if p.Cin <= 256 else "K>256")
            return (kind, fl, by)
        if name == "ryolo_conv_wgrad":
            p = args[0]
            fl = 2 * p.NB * p.OH * p.OW * p.Cout * p.ntaps * p.Cin
            by = p.NB * p.OH * p.OW * p.Cout * 2 + p.NB * p.IH * p.IW * p.Cin * 2 + p.Cout * p.ntaps * p.Cin * 4
            kern, wgs, waves = S.I(), S.I(), S.I()
            hip.call("ryolo_conv_wgrad_kernel", p, kern)
            hip.call("ryolo_conv_wgrad_grid", p, wgs, waves)
            # fifth field: CUs a launch of this kernel holds when it runs alone (8-wave workgroups are CU-exclusive and the grid covers part
            # of the chip: conv3x3_wgrad8.hip); None = the whole chip
            cus = min(256, wgs.value) if waves.value == 8 else None
            if kern.value == 1:
                return ("conv3x3_wgrad_kernel<128x9x32>", fl, by, None, cus)
            if kern.value == 3:
                return ("wgrad1x1_8w_kernel<256x256>", fl, by, None, cus)
            return (f"conv_wgrad_kernel<{64 if p.Cout <= 64 else 128}>", fl, by)
        return (name, 0, 0)

    def grad_ready_points(self, bounds):
        """For each [a, b) element range of Runtime.gflat: index of the LAST backward-tape entry that writes a gradient whose first
        element lies in it (-1 if none does).  A gradient tensor never straddles a 64-element boundary start (slices are
        256-byte aligned), and bucket bounds are cut at parameter starts by the caller."""
        import bisect
        starts = [a for a, _ in bounds]
        last = [-1] * len(bounds)
        for idx, offs in self.grad_writes:
            for o in offs:
                k = bisect.bisect_right(starts, o) - 1
                if 0 <= k < len(bounds) and o < bounds[k][1]:
                    last[k] = max(last[k], idx)
        return last

    def run(self, tape, timer=None, after=None):
        """Replay a tape.  after: {tape index: callable} invoked right after that launch was enqueued (gradient-bucket hooks).

        Backward runs on TWO streams: the weight-gradient GEMMs (MFMA-bound, 22 ms of the step, results needed only by the optimizer)
        go to a side stream behind an event, and overlap with the data-gradient / BatchNorm-backward chain of the earlier layers
        (HBM-bound) that the main stream continues with.  The main stream joins the side stream before a gradient bucket is handed
        to RCCL and at the end of the tape."""
        st = hip.stream()
        cuda = self.dev.type == "cuda" and not self.serial
        side_idx = self.side_idx if (tape is self.bwd and self.side_idx and cuda) else None
        fork = self.fwd_side if (tape is self.fwd and self.fwd_side and cuda) else None
        if side_idx or fork:
            if self._side is None:
                n = max(len(self.fwd), len(self.bwd)) + 1
                # the side STREAMS belong to the runtime (every plan of a model shares them: a second plan with streams of its own pushed
                # the process past its hardware queues and the 8-image plan ran 1.8x slower behind the 64-image one); events per plan
                self._side = (self.rt.side_stream(1), [torch.cuda.Event() for _ in range(2 * n)], torch.cuda.Event())
            side, evs, join = self._side[:3]
            n_ev = len(evs) // 2
            main = torch.cuda.current_stream(self.dev)
            sst = side.cuda_stream
        lag = self.layout.lag if (side_idx is not None and self.layout is not None) else 0
        if side_idx is not None:
            if self._done is None:
                order = sorted(self.side_idx)
                self._done = ({t: n for n, t in enumerate(order)}, [torch.cuda.Event() for _ in order])
            wpos, done = self._done
            # weight gradient n runs on lane n % lanes: with small batches one weight-gradient kernel does not fill the GPU either
            lanes = [side] + [self.rt.side_stream(2 + k) for k in range(1, max(1, self.rt.wgrad_lanes))]
            if len(lanes) > 1 and len(self._side) == 3:
                self._side = self._side + ([torch.cuda.Event() for _ in lanes],)
            used = set()
        tid = id(tape)
        dirty, dirty2, pending, lane_stream = False, False, None, None
        probe = self.probe
        for i, (fn, args, name) in enumerate(tape):
            on_side = False
            pp = probe.get((tid, i)) if probe else None
            if pp is not None and pp[0] is not None:
                pp[0]()                              # enqueued on the main stream BEFORE the fork event of a side-stream launch is recorded
            if side_idx is not None:
                on_side = i in side_idx
                if on_side:                          # a weight gradient: its operands are final once everything before it ran
                    if lag and wpos[i] >= lag:
                        # bounded lag: weight gradient n - lag has finished before anything enqueued from here on starts — the arena
                        # hands the memory of ITS operands to later launches on that promise (engine/arena.py)
                        main.wait_event(done[wpos[i] - lag])
                    evs[i].record(main)
                    lane_stream = lanes[wpos[i] % len(lanes)]
                    lane_stream.wait_event(evs[i])
                    dirty = True
                    used.add(wpos[i] % len(lanes))
                    sst = lane_stream.cuda_stream
            elif fork is not None:
                if dirty and i in self.fwd_join:     # snapshot of the side stream at the join point (later forks are not waited for)
                    pending = evs[n_ev + i]
                    pending.record(side)
                ent = fork.get(i)
                on_side = ent is not None
                if on_side:
                    first, lane = ent
                    if lane == 2:                    # long independent tails (detection heads): their own stream, joined at the end
                        if self._side2 is None:
                            self._side2 = (self.rt.side_stream(2), torch.cuda.Event())
                        lane_stream = self._side2[0]
                        dirty2 = True
                    else:
                        lane_stream = side
                        dirty = True
                    if first:                        # fork point of a branch: it may read everything enqueued so far
                        evs[i].record(main)
                        lane_stream.wait_event(evs[i])
                    sst = lane_stream.cuda_stream
                elif pending is not None:            # first main-stream launch after a join point
                    main.wait_event(pending)
                    pending = None
            kind, fl, by, *sub = self.meta.get((tid, i), (name, 0, 0)) if timer is not None else (name, 0, 0)
            if on_side and side_idx is not None and _DEBUG_SKIP_SIDE:
                rc = 0                               # diagnostic only (RYOLO_DEBUG_SKIP_SIDE=1): the step WITHOUT its weight gradients = what the main stream costs alone
            elif fl:
                e0, e1 = timer.pair()
                e0.record(lane_stream if on_side else None)
                rc = fn(*args, sst if on_side else st)
                e1.record(lane_stream if on_side else None)
                timer.note(kind, fl, e0, e1, by, *sub)
            else:
                rc = fn(*args, sst if on_side else st)
            if rc != 0:
                raise RuntimeError(f"{name} failed with code {rc}")
            if lag and on_side:
                done[wpos[i]].record(lane_stream)
            if pp is not None and pp[1] is not None:
                pp[1]()                              # (the callable synchronises the device itself when it reads a side-stream result)
            if after:
                cb = after.get(i)
                if cb is not None:
                    if dirty:
                        # the bucket may hold gradients written on the side stream: the hook's collective stream waits for this
                        # event as well (parallel._Reducer._launch); the main stream is NOT stalled
                        if side_idx is not None and len(lanes) > 1:      # all weight-gradient lanes funnel into lane 0 first
                            for k in sorted(used - {0}):
                                self._side[3][k].record(lanes[k])
                                side.wait_event(self._side[3][k])
                        join.record(side)
                        self.rt.side_event = join
                    cb()
                    self.rt.side_event = None
        if dirty:
            if side_idx is not None and len(lanes) > 1:
                for k in sorted(used - {0}):
                    self._side[3][k].record(lanes[k])
                    main.wait_event(self._side[3][k])
            join.record(side)
            main.wait_event(join)
        if dirty2:
            self._side2[1].record(self._side2[0])
            main.wait_event(self._side2[1])
        self.rt.side_event = None

    # ------------------------------------------------------------------ convolution
    def _gemm(self, tape, A, Aptr, W, Nout, wtaps, gemm_cin, OH, OW, stride, classes, epi, out_ptr, ldC, full=None,
              coeffs=None, act=0, bias=None, s2d=0, pool=None, head=None):
        """Emit one ryolo_conv_gemm launch.  For epi == EPI_STATS the partial-statistics buffer is sized by the library's plan
        (one [2][Nout] row per M tile of the kernel it will run) and returned."""
        p = S.ConvGemmParams()
        p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = Aptr, A.N, A.H, A.W, gemm_cin, A.ld
        p.W, p.Nout, p.wtaps = W.data_ptr(), Nout, wtaps
        p.OH, p.OW, p.sh, p.sw = OH, OW, stride, stride
        if full is None:
            p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, OH, OW
        else:
            p.oh_mul, p.ow_mul, p.OHf, p.OWf = full
        p.nclasses = len(classes)
        for i, (taps, oa, wa) in enumerate(classes):
            _fill_class(p.cls[i], taps, oa, wa)
        p.epi, p.out, p.ldC = epi, out_ptr, ldC
        if coeffs is not None:
            Cn = coeffs.shape[1]
            p.scale, p.shift = coeffs.data_ptr() + 2 * Cn * 4, coeffs.data_ptr() + 3 * Cn * 4
        p.act = act
        p.bias = bias
        p.zeros = self.rt.zeros.data_ptr()
        p.a_bytes, p.w_bytes = A.span_bytes, W.numel() * 2
        p.pipe = self.rt.gemm_pipe
        p.s2d_cin = s2d
        if head is not None:                            # (attrs, och, ImplicitM pointer or None, compact objectness pointer or None)
            p.head_attrs, p.head_och, p.scale, p.stats = head
        if pool is not None:
            p.pool_idx, p.pool_dz, p.pool_ldi, p.pool_ld = pool
        stats = None
        if epi == S.EPI_STATS:
            rows = S.I()
            hip.call("ryolo_conv_gemm_plan", p, rows, None)
            stats = self.f32(rows.value + 64, 2, Nout)[:rows.value]       # +64 rows: fold scratch of ryolo_bn_finalize
            p.stats = stats.data_ptr()
        self._call(tape, "ryolo_conv_gemm", p)
        return stats

    def _conv_geom(self, conv, x):
        k, s = conv.kernel_size[0], conv.stride[0]
        pad = conv.padding[0]
        OH = (x.H + 2 * pad - k) // s + 1
        OW = (x.W + 2 * pad - k) // s + 1
        return k, s, pad, OH, OW

    def _dgrad(self, conv, pk, dy, dy_ptr, dy_cin, x, pool=None):
        """x.grad (=|+=) conv_transpose(dy).  dy: TRef-like geometry (N, OH, OW, ld).  pool = {"idx", "z"} of a MaxPool2d(2, 2) of the
        same tensor x: its gradient is added in this launch's store (1x1 stride-1 layers on the generic kernel only)."""
        k, s, pad, OH, OW = self._conv_geom(conv, x)
        mode = x.grad_write_mode()
        epi = S.EPI_ACCUM if mode else S.EPI_RAW
        cin = conv.in_channels
        if s == 1:
            taps = [(pad - r, pad - c, r * k + c) for r in range(k) for c in range(k)]
            pl = None
            if pool is not None:
                assert k == 1 and pool["z"].C == x.C
                pl = (pool["idx"].data_ptr(), pool["z"].gptr(), x.C, pool["z"].ld)
            self._gemm(self.bwd, dy, dy_ptr, pk["wd"], cin, k * k, dy_cin, x.H, x.W, 1, [(taps, 0, 0)], epi, x.gptr(), x.ld, pool=pl)
        elif (self.rt.s2d_dgrad and k == 3 and pad == 1 and cin <= self.rt.s2d_dgrad_maxc and cin % 8 == 0 and x.H % 2 == 0 and x.W % 2 == 0
              and dy_cin % 32 == 0 and dy_cin == conv.out_channels):
            # narrow stride-2 layer: ONE stride-1 GEMM over the dY grid, N = 4 parities x cin, 2x2 taps, depth-to-space store
            taps = [(da, db, 2 * da + db) for da in range(2) for db in range(2)]
            self._gemm(self.bwd, dy, dy_ptr, self.rt.packed_s2d(conv), 4 * cin, 4, dy_cin, x.H // 2, x.W // 2, 1, [(taps, 0, 0)], epi,
                       x.gptr(), x.ld, full=(2, 2, x.H, x.W), s2d=cin)
        else:
            assert s == 2 and x.H % 2 == 0 and x.W % 2 == 0
            classes = []
            for ph in range(2):
                for pw in range(2):
                    taps = [((ph + pad - r) // 2, (pw + pad - c) // 2, r * k + c) for r in range(k) for c in range(k)
                            if (ph + pad - r) % 2 == 0 and (pw + pad - c) % 2 == 0]
                    classes.append((taps, ph, pw))
            self._gemm(self.bwd, dy, dy_ptr, pk["wd"], cin, k * k, dy_cin, x.H // 2, x.W // 2, 1, classes, epi, x.gptr(), x.ld,
                       full=(2, 2, x.H, x.W))

    def _wgrad(self, conv, dy, dy_ptr, cout_pad, x, x_ptr=None, dW=None):
        k, s, pad, OH, OW = self._conv_geom(conv, x)
        p = S.WgradParams()
        p.dY, p.ldY, p.Cout, p.CoutPad = dy_ptr, dy.ld, conv.out_channels, cout_pad
        p.X, p.NB, p.IH, p.IW, p.Cin, p.ldX = x_ptr or x.ptr(), x.N, x.H, x.W, conv.in_channels, x.ld
        p.OH, p.OW, p.sh, p.sw = OH, OW, s, s
        taps = _taps_fwd(k, pad)
        p.ntaps = len(taps)
        for i, (dh, dw, _) in enumerate(taps):
            p.dh[i], p.dw[i] = dh, dw
        p.dW = dW if dW is not None else self.rt.grad_ptr(conv.weight)
        self._emit_wgrad(p, side=True)

    def _emit_wgrad(self, p, side=False):
        """side: the launch may run on the weight-gradient stream (see run()): nothing on the main stream reads what it writes (the
        split-K slabs and .grad) before the end of backward, and what it reads (a forward activation, the raw gradient of its own
        output) is final by the time it is enqueued."""
        p.zeros = self.rt.zeros.data_ptr()
        sk, need = S.I(), S.Z()
        hip.call("ryolo_conv_wgrad_plan", p, sk, need)
        on_side = bool(side and self.rt.wgrad_stream)
        if on_side or not self.rt.wgrad_stream:
            # launches of ONE stream are ordered, so they can share one split-K workspace; side launch n runs on lane n % lanes (run())
            lane = len(self.side_idx) % max(1, self.rt.wgrad_lanes) if on_side else 0
            self._wgrad_ws_bytes[lane] = max(self._wgrad_ws_bytes.get(lane, 0), need.value)
            self._wgrads.setdefault(lane, []).append(p)
        else:
            # a weight gradient that stays on the main stream (the im2col stem of yolov5) would race with the side-stream launches
            # on a shared workspace: it gets its own slabs
            own = torch.empty(max(need.value, 16), dtype=torch.uint8, device=self.adev)
            self.keep.append(own)
            p.partial = own.data_ptr()
        self._call(self.bwd, "ryolo_conv_wgrad", p)
        if on_side:
            self.side_idx.add(len(self.bwd) - 1)

    def conv_raw(self, conv, x, want_stats, fused=None, pool_grad=None):
        """Emit the forward conv; returns (y TRef [M, Cout] raw bf16, stats tensor or None, backward-emitter).
        fused = (coeffs [4][Cout], act code, z TRef): eval-mode epilogue writes act(bn(conv)) straight into z (no raw y)."""
        rt = self.rt
        pk = rt.packed(conv)
        k, s, pad, OH, OW = self._conv_geom(conv, x)
        cout = conv.out_channels
        y = fused[2] if fused else self.new(x.N, OH, OW, cout)
        epi = S.EPI_AFFINE_ACT if fused else (S.EPI_STATS if want_stats else S.EPI_RAW)
        stats = self._gemm(self.fwd, x, x.ptr(), pk["wf"], cout, k * k, conv.in_channels, OH, OW, s, [(_taps_fwd(k, pad), 0, 0)], epi,
                           y.ptr(), y.ld, coeffs=fused[0] if fused else None, act=fused[1] if fused else 0)

        def backward(need_dx=True):
            self._wgrad(conv, y, y.gptr(), cout, x)
            if need_dx:
                self._dgrad(conv, pk, y, y.gptr(), cout, x, pool=pool_grad if (pool_grad and pool_grad.get("idx") is not None) else None)
        return y, stats, backward

    def stem_raw(self, conv, want_stats, fused=None):
        """First layer (Cin = 3).  3x3 stride-1 stems (yolov4 / yolov7) run DIRECTLY on the fp32 NCHW image (csrc/stem.hip); other
        shapes (yolov5's 6x6 stride 2) go through an explicit im2col + single-tap GEMM (K padded to a multiple of 32)."""
        rt = self.rt
        pk = rt.packed(conv)
        k, s, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        OH = (self.Hin + 2 * pad - k) // s + 1
        OW = (self.Win + 2 * pad - k) // s + 1
        kp = pk["CinP"]
        cout = conv.out_channels
        y = fused[2] if fused else self.new(self.B, OH, OW, cout)
        epi = S.EPI_AFFINE_ACT if fused else (S.EPI_STATS if want_stats else S.EPI_RAW)
        if (k, s, pad, conv.in_channels, cout, kp) == (3, 1, 1, 3, 32, 32) and self.Win % 16 == 0 and self.B * 3 * self.Hin * self.Win < 2 ** 31:
            rows, wsb = S.I(), S.Z()
            hip.call("ryolo_stem3x3_plan", self.B, self.Hin, self.Win, cout, rows, wsb)
            p = S.StemParams()
            p.img, p.NB, p.H, p.W = self.img.data_ptr(), self.B, self.Hin, self.Win
            p.wf, p.Cout, p.epi, p.out, p.ldC = pk["wf"].data_ptr(), cout, epi, y.ptr(), y.ld
            stats = None
            if epi == S.EPI_STATS:
                stats = self.f32(rows.value + 64, 2, cout)[:rows.value]
                p.stats = stats.data_ptr()
            if fused:
                co = fused[0]
                p.scale, p.shift, p.act = co.data_ptr() + 2 * cout * 4, co.data_ptr() + 3 * cout * 4, fused[1]
            self._call(self.fwd, "ryolo_stem3x3_fwd", p)
            self._img_structs.append(p)

            def backward(need_dx=False, fuse=None):
                """fuse = (dz TRef, coeffs [4][C], bco, act code): BatchNorm + activation backward applied inside the kernel (the raw
                gradient of this layer, the largest activation of the network, is never materialised)."""
                scratch = self.f32(cout, kp)
                ws = self.f32(wsb.value // 4)
                q = S.StemWgradParams()
                q.img, q.NB, q.H, q.W = self.img.data_ptr(), self.B, self.Hin, self.Win
                if fuse is not None:
                    dz, co4, bco, actc = fuse
                    q.dY, q.ldY, q.Cout = dz.gptr(), dz.ld, cout
                    q.y, q.ldy, q.act, q.co, q.bco = y.ptr(), y.ld, actc, co4.data_ptr(), bco.data_ptr()
                else:
                    q.dY, q.ldY, q.Cout = y.gptr(), y.ld, cout
                q.scratch, q.workspace = scratch.data_ptr(), ws.data_ptr()
                self._call(self.bwd, "ryolo_stem3x3_wgrad", q)
                self._img_structs.append(q)
                self._call(self.bwd, "ryolo_unpack_wgrad", scratch.data_ptr(), cout, 3, k * k, kp, rt.grad_ptr(conv.weight))
            backward.can_fuse_bn = True
            return y, stats, backward
        col = self.new(self.B, OH, OW, kp)
        self._call(self.fwd, "ryolo_im2col", self.img.data_ptr(), self.B, 3, self.Hin, self.Win, k, k, s, pad, OH, OW, kp, col.ptr())
        self._img_slot = len(self.fwd) - 1         # tape entry whose first argument (the image pointer) is patched per call
        stats = self._gemm(self.fwd, col, col.ptr(), pk["wf"], cout, 1, kp, OH, OW, 1, [([(0, 0, 0)], 0, 0)], epi, y.ptr(), y.ld,
                           coeffs=fused[0] if fused else None, act=fused[1] if fused else 0)

        def backward(need_dx=False):
            scratch = self.f32(cout, kp)
            self.bwd.append((lambda *_a: (scratch.zero_(), 0)[1], (), "zero_scratch"))
            p = S.WgradParams()
            p.dY, p.ldY, p.Cout, p.CoutPad = y.gptr(), y.ld, cout, cout
            p.X, p.NB, p.IH, p.IW, p.Cin, p.ldX = col.ptr(), col.N, col.H, col.W, kp, col.ld
            p.OH, p.OW, p.sh, p.sw, p.ntaps = OH, OW, 1, 1, 1
            p.dh[0], p.dw[0] = 0, 0
            p.dW = scratch.data_ptr()
            self._emit_wgrad(p)
            self._call(self.bwd, "ryolo_unpack_wgrad", scratch.data_ptr(), cout, 3, k * k, kp, rt.grad_ptr(conv.weight))
        return y, stats, backward

    def _stem_recompute(self, conv, bn, actc, out):
        """Training plan of a 3x3 stride-1 first layer without its raw output: [statistics-only conv pass -> bn_finalize ->] conv pass
        with BatchNorm + activation applied to the bf16-rounded accumulator (bit-identical to storing y and running bn_act_fwd), and
        ONE fused backward pass (ryolo_stem3x3_bwd).  Returns None when the shape is not eligible (caller falls back)."""
        rt = self.rt
        k, s, pad, cout = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.out_channels
        pk = rt.packed(conv)
        wsb = S.Z()
        if (k, s, pad, conv.in_channels, cout, pk["CinP"]) != (3, 1, 1, 3, 32, 32) or self.B * 3 * self.Hin * self.Win >= 2 ** 31:
            return None
        if hip.lib().ryolo_stem3x3_bwd_plan(self.B, self.Hin, self.Win, cout, C.byref(wsb)) != 0:
            return None
        rows = S.I()
        hip.call("ryolo_stem3x3_plan", self.B, self.Hin, self.Win, cout, rows, None)
        z = out if out is not None else self.new(self.B, self.Hin, self.Win, cout)
        co = self.f32(4, cout)

        def params(epi):
            p = S.StemParams()
            p.img, p.NB, p.H, p.W = self.img.data_ptr(), self.B, self.Hin, self.Win
            p.wf, p.Cout, p.epi = pk["wf"].data_ptr(), cout, epi
            self._img_structs.append(p)
            return p
        if self.batch_stats:
            p0 = params(S.EPI_STATS)                      # out stays null: statistics only
            stats = self.f32(rows.value + 64, 2, cout)[:rows.value]
            p0.stats = stats.data_ptr()
            self._call(self.fwd, "ryolo_stem3x3_fwd", p0)
            self._call(self.fwd, "ryolo_bn_finalize", stats.data_ptr(), stats.shape[0], cout, float(z.M), float(bn.eps), float(bn.momentum),
                       bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), co.data_ptr())
            rt.bn_counters.append(bn)
        else:
            self._call(self.fwd, "ryolo_bn_eval_coeffs", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                       bn.running_var.data_ptr(), float(bn.eps), cout, co.data_ptr())
        p1 = params(S.EPI_AFFINE_ACT_R)
        p1.out, p1.ldC = z.ptr(), z.ld
        p1.scale, p1.shift, p1.act = co.data_ptr() + 2 * cout * 4, co.data_ptr() + 3 * cout * 4, actc
        self._call(self.fwd, "ryolo_stem3x3_fwd", p1)
        self.debug[id(conv)] = (z, z, None)

        def backward():
            ws = self.f32(wsb.value // 4)
            q = S.StemBwdParams()
            q.img, q.NB, q.H, q.W = self.img.data_ptr(), self.B, self.Hin, self.Win
            q.dz, q.lddz, q.wf, q.co, q.act, q.frozen = z.gptr(), z.ld, pk["wf"].data_ptr(), co.data_ptr(), actc, 1 if self.frozen else 0
            q.workspace, q.dW = ws.data_ptr(), rt.grad_ptr(conv.weight)
            q.dgamma, q.dbeta = rt.grad_ptr(bn.weight), rt.grad_ptr(bn.bias)
            self._call(self.bwd, "ryolo_stem3x3_bwd", q)
            self._img_structs.append(q)
        self._pending_bwd.append(backward)
        return z

    # ------------------------------------------------------------------ Conv = conv -> BN -> act (+ residual)
    def conv_bn_act(self, conv, bn, act, x, out=None, residual=None, stem=False, pool_grad=None):
        """model/utils.py:6-32.  x None => stem on the staged input image.  Returns the activation TRef."""
        rt = self.rt
        train = self.training
        bstat = self.batch_stats
        actc = S.ACT[act]
        cout = conv.out_channels
        if not train and residual is None:
            # inference: BatchNorm folded to scale/shift (running statistics) + activation inside the GEMM epilogue
            co = self.f32(4, cout)
            # folded scale/shift depend on the weights only: their own tape, so a captured inference graph can leave them out
            self._call(self.wprep, "ryolo_bn_eval_coeffs", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                       bn.running_var.data_ptr(), float(bn.eps), cout, co.data_ptr())
            k_, s_, pad_ = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            H_, W_ = (self.Hin, self.Win) if stem else (x.H, x.W)
            OH_, OW_ = (H_ + 2 * pad_ - k_) // s_ + 1, (W_ + 2 * pad_ - k_) // s_ + 1
            z = out if out is not None else self.new(self.B if stem else x.N, OH_, OW_, cout)
            fused = (co, actc, z)
            (self.stem_raw(conv, False, fused) if stem else self.conv_raw(conv, x, False, fused))
            self.debug[id(conv)] = (z, z, x)
            return z
        if stem and residual is None and rt.stem_recompute:
            z = self._stem_recompute(conv, bn, actc, out)
            if z is not None:
                return z
        y, stats, conv_bwd = (self.stem_raw(conv, bstat) if stem else self.conv_raw(conv, x, bstat, pool_grad=pool_grad))
        co = self.f32(4, cout)
        if bstat:
            self._call(self.fwd, "ryolo_bn_finalize", stats.data_ptr(), stats.shape[0], cout, float(y.M), float(bn.eps), float(bn.momentum),
                       bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), co.data_ptr())
            rt.bn_counters.append(bn)
        else:
            self._call(self.fwd, "ryolo_bn_eval_coeffs", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                       bn.running_var.data_ptr(), float(bn.eps), cout, co.data_ptr())
        z = out if out is not None else self.new(y.N, y.H, y.W, cout)
        assert z.C == cout and z.M == y.M
        self.debug[id(conv)] = (y, z, x)
        p = S.BnActParams()
        p.y1, p.ld1, p.co1 = y.ptr(), y.ld, co.data_ptr()
        if residual is not None:
            p.res, p.ldr = residual.ptr(), residual.ld
        p.z, p.ldz, p.M, p.C, p.act = z.ptr(), z.ld, y.M, cout, actc
        self._call(self.fwd, "ryolo_bn_act_fwd", p)
        if train:
            def backward():
                bco = self.f32(3, cout)
                q = S.BnActParams()
                C.memmove(C.byref(q), C.byref(p), C.sizeof(p))
                nblk, rpb = S.I(), S.I()
                hip.call("ryolo_bn_act_bwd_blocks", y.M, cout, nblk, rpb)
                partial = self.f32(nblk.value + 64, 2, cout)
                q.dz, q.lddz = z.gptr(), z.ld
                # direct stem: the apply pass moves into the weight-gradient kernel (its only consumer); here statistics only
                fuse_stem = stem and residual is None and getattr(conv_bwd, "can_fuse_bn", False) and rt.fuse_stem_bn
                if not fuse_stem:
                    q.dy1, q.lddy1 = y.gptr(), y.ld
                if residual is not None:
                    q.dres, q.lddres, q.dres_accum = residual.gptr(), residual.ld, residual.grad_write_mode()
                q.partial = partial.data_ptr()
                self._call(self.bwd, "ryolo_bn_act_bwd", q, rt.grad_ptr(bn.weight), rt.grad_ptr(bn.bias), None, None, bco.data_ptr(), 1 if self.frozen else 0)
                if fuse_stem:
                    conv_bwd(need_dx=False, fuse=(z, co, bco, actc))
                else:
                    conv_bwd(need_dx=not stem)
            self._pending_bwd.append(backward)
        return z

    def conv_bn_act_group(self, mods, x, outs):
        """Two sibling `Conv` blocks (model/utils.py:6-32) that read the same input with the same kernel / stride — cv1 + cv2 of ELAN1 /
        ELAN2 / CSP / C3 / SPPCSPC (model/utils.py:49-143,264-282) — as ONE GEMM with concatenated output channels: forward and the
        weight gradient read x once instead of twice, and the data gradient is one launch with K = Ca + Cb that stores dx instead
        of a store followed by a read-modify-write.  Each member keeps its own BatchNorm (statistics of its channel slice of the
        shared partial rows), its own activation pass into its own destination (`outs[i]`: a concat slice, or None for a new
        buffer), its own fp32 master weights and .grad.  Returns the members' activation TRefs."""
        rt = self.rt
        convs = [m.conv[0] for m in mods]
        c0 = convs[0]
        ok = (rt.merge_siblings and len(mods) == 2 and all(m.has_bn for m in mods) and c0.in_channels % 32 == 0 and c0.stride[0] == 1
              and all(c.kernel_size == c0.kernel_size and c.stride == c0.stride and c.padding == c0.padding and
                      c.in_channels == c0.in_channels and c.out_channels % 32 == 0 for c in convs))
        if ok and not self.training:
            # inference: ONE folded-BatchNorm + activation epilogue -> same activation, destinations adjacent in one buffer
            ok = (all(o is not None for o in outs) and outs[0].buf is outs[1].buf and outs[1].c0 == outs[0].c0 + outs[0].C
                  and mods[0].act == mods[1].act)
        if not ok:
            return [m.emit(self, x, out=o) for m, o in zip(mods, outs)]
        bns = [m.conv[1] for m in mods]
        couts = [c.out_channels for c in convs]
        ctot = sum(couts)
        cin = c0.in_channels
        k, s_, pad, OH, OW = self._conv_geom(c0, x)
        grp = rt.packed_group(convs)
        taps = [(_taps_fwd(k, pad), 0, 0)]
        if not self.training:
            co = self.f32(4, ctot)
            off = 0
            for bn, cout in zip(bns, couts):
                self._call(self.wprep, "ryolo_bn_eval_coeffs_slice", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                           bn.running_var.data_ptr(), float(bn.eps), cout, co.data_ptr(), ctot, off)
                off += cout
            z = TRef(outs[0].buf, outs[0].c0, ctot)
            self._gemm(self.fwd, x, x.ptr(), grp["wf"], ctot, k * k, cin, OH, OW, s_, taps, S.EPI_AFFINE_ACT, z.ptr(), z.ld, coeffs=co,
                       act=S.ACT[mods[0].act])
            for conv, o in zip(convs, outs):
                self.debug[id(conv)] = (o, o, x)
            return list(outs)
        bstat = self.batch_stats
        y = self.new(x.N, OH, OW, ctot)
        stats = self._gemm(self.fwd, x, x.ptr(), grp["wf"], ctot, k * k, cin, OH, OW, s_, taps, S.EPI_STATS if bstat else S.EPI_RAW,
                           y.ptr(), y.ld)
        zs, parts, off = [], [], 0
        for m, conv, bn, cout, out in zip(mods, convs, bns, couts, outs):
            ys = y.slice(off, cout)
            co = self.f32(4, cout)
            if bstat:
                self._call(self.fwd, "ryolo_bn_finalize_slice", stats.data_ptr(), stats.shape[0], ctot, off, cout, float(y.M), float(bn.eps),
                           float(bn.momentum), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                           bn.running_var.data_ptr(), co.data_ptr())
                rt.bn_counters.append(bn)
            else:
                self._call(self.fwd, "ryolo_bn_eval_coeffs", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                           bn.running_var.data_ptr(), float(bn.eps), cout, co.data_ptr())
            z = out if out is not None else self.new(y.N, y.H, y.W, cout)
            assert z.C == cout and z.M == y.M
            self.debug[id(conv)] = (ys, z, x)
            p = S.BnActParams()
            p.y1, p.ld1, p.co1 = ys.ptr(), ys.ld, co.data_ptr()
            p.z, p.ldz, p.M, p.C, p.act = z.ptr(), z.ld, y.M, cout, S.ACT[m.act]
            self._call(self.fwd, "ryolo_bn_act_fwd", p)
            zs.append(z)
            parts.append((p, ys, z, bn, cout))
            off += cout

        def backward():
            for p, ys, z, bn, cout in parts:
                bco = self.f32(3, cout)
                q = S.BnActParams()
                C.memmove(C.byref(q), C.byref(p), C.sizeof(p))
                nblk, rpb = S.I(), S.I()
                hip.call("ryolo_bn_act_bwd_blocks", y.M, cout, nblk, rpb)
                partial = self.f32(nblk.value + 64, 2, cout)
                q.dz, q.lddz = z.gptr(), z.ld
                q.dy1, q.lddy1 = ys.gptr(), ys.ld
                q.partial = partial.data_ptr()
                self._call(self.bwd, "ryolo_bn_act_bwd", q, rt.grad_ptr(bn.weight), rt.grad_ptr(bn.bias), None, None, bco.data_ptr(),
                           1 if self.frozen else 0)
            w = S.WgradParams()
            w.dY, w.ldY, w.Cout, w.CoutPad = y.gptr(), y.ld, ctot, ctot
            w.X, w.NB, w.IH, w.IW, w.Cin, w.ldX = x.ptr(), x.N, x.H, x.W, cin, x.ld
            w.OH, w.OW, w.sh, w.sw = OH, OW, s_, s_
            tp = _taps_fwd(k, pad)
            w.ntaps = len(tp)
            for i, (dh, dw, _) in enumerate(tp):
                w.dh[i], w.dw[i] = dh, dw
            w.dW, w.dW2, w.Cout1 = rt.grad_ptr(convs[0].weight), rt.grad_ptr(convs[1].weight), couts[0]
            self._emit_wgrad(w, side=True)
            self._dgrad(c0, {"wd": grp["wd"]}, y, y.gptr(), ctot, x)
        self._pending_bwd.append(backward)
        return zs

    # RepConv: silu(bn(conv3x3(x)) + bn(conv1x1(x)) [+ bn(x)])  (model/utils.py:189-215)
    def repconv(self, rep, x):
        if rep.rbr_identity is not None:
            raise NotImplementedError("RepConv identity branch (c1 == c2) is not used by the reference's necks")
        rt, train, bstat = self.rt, self.training, self.batch_stats
        conv_a, bn_a = rep.rbr_dense[0], rep.rbr_dense[1]
        conv_b, bn_b = rep.rbr_1x1[0], rep.rbr_1x1[1]
        cout = conv_a.out_channels
        if not train and rt.fold_repconv and conv_a.stride[0] == 1 and conv_a.in_channels % 32 == 0:
            # inference: the two affine branches collapse into ONE 3x3 GEMM with the folded-BN epilogue (csrc/elementwise.hip:
            # repconv_fold_kernel); like the BN folding it depends on the weights only -> wprep tape
            cin = conv_a.in_channels
            coa, cob, co = self.f32(4, cout), self.f32(4, cout), self.f32(4, cout)
            for bn, c4 in ((bn_a, coa), (bn_b, cob)):
                self._call(self.wprep, "ryolo_bn_eval_coeffs", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                           bn.running_var.data_ptr(), float(bn.eps), cout, c4.data_ptr())
            wfold = torch.empty((cout, 9, cin), dtype=BF16, device=self.adev)
            self.keep.append(wfold)
            self._call(self.wprep, "ryolo_repconv_fold", conv_a.weight.data_ptr(), conv_b.weight.data_ptr(), coa.data_ptr(), cob.data_ptr(),
                       cout, cin, wfold.data_ptr(), co.data_ptr())
            z = self.new(x.N, x.H, x.W, cout)
            self._gemm(self.fwd, x, x.ptr(), wfold, cout, 9, cin, x.H, x.W, 1, [(_taps_fwd(3, 1), 0, 0)], S.EPI_AFFINE_ACT, z.ptr(), z.ld,
                       coeffs=co, act=S.ACT["swish"])
            return z
        ya, sa, bwd_a = self.conv_raw(conv_a, x, bstat)
        yb, sb, bwd_b = self.conv_raw(conv_b, x, bstat)
        coa, cob = self.f32(4, cout), self.f32(4, cout)
        for bn, st, co, y in ((bn_a, sa, coa, ya), (bn_b, sb, cob, yb)):
            if bstat:
                self._call(self.fwd, "ryolo_bn_finalize", st.data_ptr(), st.shape[0], cout, float(y.M), float(bn.eps), float(bn.momentum),
                           bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), co.data_ptr())
                rt.bn_counters.append(bn)
            else:
                self._call(self.fwd, "ryolo_bn_eval_coeffs", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                           bn.running_var.data_ptr(), float(bn.eps), cout, co.data_ptr())
        z = self.new(ya.N, ya.H, ya.W, cout)
        p = S.BnActParams()
        p.y1, p.ld1, p.co1 = ya.ptr(), ya.ld, coa.data_ptr()
        p.y2, p.ld2, p.co2 = yb.ptr(), yb.ld, cob.data_ptr()
        p.z, p.ldz, p.M, p.C, p.act = z.ptr(), z.ld, ya.M, cout, S.ACT["swish"]
        self._call(self.fwd, "ryolo_bn_act_fwd", p)
        if train:
            def backward():
                nblk, rpb = S.I(), S.I()
                hip.call("ryolo_bn_act_bwd_blocks", ya.M, cout, nblk, rpb)
                partial = self.f32(nblk.value + 64, 3, cout)
                bco = self.f32(3, cout)
                q = S.BnActParams()
                C.memmove(C.byref(q), C.byref(p), C.sizeof(p))
                q.dz, q.lddz = z.gptr(), z.ld
                q.dy1, q.lddy1, q.dy2, q.lddy2 = ya.gptr(), ya.ld, yb.gptr(), yb.ld
                q.partial = partial.data_ptr()
                self._call(self.bwd, "ryolo_bn_act_bwd", q, rt.grad_ptr(bn_a.weight), rt.grad_ptr(bn_a.bias), rt.grad_ptr(bn_b.weight),
                           rt.grad_ptr(bn_b.bias), bco.data_ptr(), 1 if self.frozen else 0)
                bwd_a()
                bwd_b()
            self._pending_bwd.append(backward)
        return z

    # ------------------------------------------------------------------ pooling / upsample
    def maxpool(self, x, k, stride, out=None, grad_into=None):
        """grad_into: a dict handed earlier to the sibling 1x1 conv of the same input (conv_bn_act(..., pool_grad=grad_into)): this pool's
        backward is then NOT a launch of its own — the sibling's data-gradient store adds it (ConvGemmParams.pool_idx)."""
        pad = 0 if stride == 2 else k // 2
        OH = (x.H + 2 * pad - k) // stride + 1
        OW = (x.W + 2 * pad - k) // stride + 1
        z = out if out is not None else self.new(x.N, OH, OW, x.C)
        idx = torch.empty((x.N * OH * OW, x.C), dtype=torch.uint8, device=self.adev) if self.training else None
        self.keep.append(idx)
        p = S.PoolParams()
        p.x, p.ldx, p.z, p.ldz = x.ptr(), x.ld, z.ptr(), z.ld
        p.NB, p.H, p.W, p.C, p.k, p.stride, p.pad, p.OH, p.OW = x.N, x.H, x.W, x.C, k, stride, pad, OH, OW
        p.idx = idx.data_ptr() if idx is not None else None
        if stride == 1 and k >= 5:
            # SPP-style windows: separable row / column passes (scratch: row maxima; their argmax offsets are kept for backward)
            rowmax = torch.empty((x.M, x.C), dtype=BF16, device=self.adev)
            rowidx = torch.empty((x.M, x.C), dtype=torch.uint8, device=self.adev) if self.training else None
            grow = torch.empty((x.M, x.C), dtype=torch.float32, device=self.adev) if self.training else None
            self.keep.extend([rowmax, rowidx, grow])
            p.rowmax = rowmax.data_ptr()
            p.rowidx = rowidx.data_ptr() if rowidx is not None else None
            p.growws = grow.data_ptr() if grow is not None else None
        self._call(self.fwd, "ryolo_maxpool_fwd", p)
        if self.training and grad_into is not None:
            grad_into["idx"], grad_into["z"] = idx, z
            return z
        if self.training:
            def backward():
                q = S.PoolParams()
                C.memmove(C.byref(q), C.byref(p), C.sizeof(p))
                q.dz, q.lddz, q.dx, q.lddx, q.accum = z.gptr(), z.ld, x.gptr(), x.ld, x.grad_write_mode()
                self._call(self.bwd, "ryolo_maxpool_bwd", q)
            self._pending_bwd.append(backward)
        return z

    def upsample(self, x, out=None):
        z = out if out is not None else self.new(x.N, 2 * x.H, 2 * x.W, x.C)
        p = S.UpParams()
        p.x, p.ldx, p.z, p.ldz, p.NB, p.H, p.W, p.C = x.ptr(), x.ld, z.ptr(), z.ld, x.N, x.H, x.W, x.C
        self._call(self.fwd, "ryolo_upsample2x_fwd", p)
        if self.training:
            def backward():
                q = S.UpParams()
                q.x, q.ldx, q.z, q.ldz, q.NB, q.H, q.W, q.C, q.accum = z.gptr(), z.ld, x.gptr(), x.ld, x.N, x.H, x.W, x.C, x.grad_write_mode()
                self._call(self.bwd, "ryolo_upsample2x_bwd", q)
            self._pending_bwd.append(backward)
        return z

    def copy_slice(self, x, out):
        """Materialise an already-produced tensor into a concat slice (only needed when the producer could not be planned to
        write there directly).  Plumbing-level strided copy; gradient flows back by accumulation."""
        if self.dry:                                 # liveness pass: the two uses, no torch views of virtual buffers
            self.fwd.append((None, (), "copy_slice"))
            self._raw[(id(self.fwd), len(self.fwd) - 1)] = (x.ptr(), out.ptr())
            if self.training:
                def backward():
                    x.grad_write_mode()
                    self.bwd.append((None, (), "copy_slice_bwd"))
                    self._raw[(id(self.bwd), len(self.bwd) - 1)] = (x.gptr(), out.gptr())
                self._pending_bwd.append(backward)
            return out
        xs = x.buf.t[:, x.c0:x.c0 + x.C]
        os_ = out.buf.t[:, out.c0:out.c0 + out.C]
        self.fwd.append((lambda *_a: (os_.copy_(xs), 0)[1], (), "copy_slice"))
        if self.training:
            def backward():
                mode = x.grad_write_mode()
                gx = x.buf.grad_tensor()[:, x.c0:x.c0 + x.C]
                go = out.buf.grad_tensor()[:, out.c0:out.c0 + out.C]
                self.bwd.append(((lambda *_a: (gx.add_(go), 0)[1]) if mode else (lambda *_a: (gx.copy_(go), 0)[1]), (), "copy_slice_bwd"))
            self._pending_bwd.append(backward)
        return out

    # ------------------------------------------------------------------ detection head
    def head(self, conv, x, na, attrs, implicit_a=None, implicit_m=None):
        """[ImplicitA ->] 1x1 conv + bias [-> ImplicitM] -> fp32 [B, na, gs, gs, attrs]  (model/neck.py:173-186,201,208,215;
        the view/permute of model/yololayer.py:25 is fused into the store)."""
        rt = self.rt
        pk = rt.packed(conv)
        cout, coutp = conv.out_channels, pk["CoutP"]
        assert cout == na * attrs
        och = 4 if getattr(rt.model, "mode", None) == "csl" else 5         # objectness element of a head row (lib/loss.py:216, :411)
        mptr = implicit_m.data_ptr() if implicit_m is not None else None
        sparse = self.training and _HEAD_SPARSE
        fused = _HEAD_FUSED and rt.wgrad_lanes <= 1 and self._head_fused_ok(x, pk, cout, conv, x, attrs, och)
        if fused and implicit_a is not None and implicit_m is not None and _HEAD_FOLD_A:
            # ImplicitA rides in the bias (W (x + a) + b = W x + (b + W a)): no x + a tensor, and its gradient comes out of ryolo_head_wgrad_finish
            return self._head_fused(conv, pk, x, x, na, attrs, och, implicit_a, implicit_m, sparse, fold_a=True)
        xin = x
        if implicit_a is not None:
            xin = self.new(x.N, x.H, x.W, x.C)
            self._call(self.fwd, "ryolo_chan_add", x.ptr(), x.ld, implicit_a.data_ptr(), x.M, x.C, xin.ptr(), xin.ld)
        M = x.M
        if fused:
            return self._head_fused(conv, pk, x, xin, na, attrs, och, implicit_a, implicit_m, sparse)
        if self.training and pk.get("wd_scale") is not None:
            # the packed data-gradient image of this conv already carries ImplicitM (a fused TRAINING plan made it so): the unfused backward
            # below would apply ImplicitM a second time in ryolo_head_finish_bwd.  Inference plans never read Wd and may mix freely.
            raise RuntimeError("ryolov4_amd: a detection head was planned for training both with and without the fused (ImplicitM-carrying) "
                               "data-gradient image; use one RYOLO_HEAD_FUSED setting per runtime")
        pre = self.f32(M, coutp)
        self._gemm(self.fwd, xin, xin.ptr(), pk["wf"], cout, 1, conv.in_channels, x.H, x.W, 1, [([(0, 0, 0)], 0, 0)], S.EPI_F32_BIAS,
                   pre.data_ptr(), coutp, bias=conv.bias.data_ptr())
        out = self.f32(x.N, na, x.H, x.W, attrs)
        preobj = xobj = None
        if sparse:
            # compact copy of the objectness column for the sparse head backward (set_head_grad)
            # and of the objectness logits for the fused loss (lib/loss.py head_obj_logits: 4 bytes per cell instead of a strided walk of the map)
            preobj = self.f32(x.N, na, x.H, x.W)
            xobj = self.f32(x.N, na, x.H, x.W) if mptr is not None else preobj        # (no ImplicitM: the same values)
            self._call(self.fwd, "ryolo_head_finish_fwd_obj", pre.data_ptr(), coutp, mptr, x.N, x.H, na, attrs, out.data_ptr(), och, preobj.data_ptr(),
                       xobj.data_ptr() if mptr is not None else None)
        else:
            self._call(self.fwd, "ryolo_head_finish_fwd", pre.data_ptr(), coutp, mptr, x.N, x.H, na, attrs, out.data_ptr())
        rec = dict(out=out, dout=None, preobj=preobj, och=och, xobj=xobj)
        self.heads.append(rec)
        if self.training:
            dout = self.f32(x.N, na, x.H, x.W, attrs)
            rec["dout"] = dout
            dpre = torch.zeros((M, coutp), dtype=BF16, device=self.adev)       # pad columns stay zero
            self.keep.append(dpre)
            nblk = x.N * ((x.H * x.W + 127) // 128)
            scratch = self.f32(max((nblk + 64) * 2 * cout, ((M + 255) // 256 + 64) * max(coutp, x.C)))

            class _G:                      # geometry shim so dpre can be used as a gathered operand
                N, H, W, ld = x.N, x.H, x.W, coutp
                span_bytes = M * coutp * 2

            def backward():
                # dpre, the conv bias gradient (column sums of dpre) and the ImplicitM gradient in one pass over dout
                self._call(self.bwd, "ryolo_head_finish_bwd", dout.data_ptr(), pre.data_ptr(), coutp, mptr, x.N, x.H, na, attrs,
                           dpre.data_ptr(), coutp, rt.grad_ptr(conv.bias), rt.grad_ptr(implicit_m) if implicit_m is not None else None,
                           scratch.data_ptr())
                rec["dout_slot"] = len(self.bwd) - 1   # first argument (dout pointer) is patched per call when the caller's grad is usable as is
                self._wgrad(conv, _G, dpre.data_ptr(), coutp, xin)
                # ImplicitA: d(x + a)/dx = 1, so the data gradient is written straight into x's gradient (no buffer for xin's, no copy)
                # and d/da is its column sum
                self._dgrad(conv, pk, _G, dpre.data_ptr(), coutp, x)
                if implicit_a is not None:
                    self._call(self.bwd, "ryolo_colsum_bf16", x.gptr(), x.ld, M, x.C, x.C, rt.grad_ptr(implicit_a), scratch.data_ptr())
            self._pending_bwd.append(backward)
        return out

    def _head_fused_ok(self, xin, pk, cout, conv, x, attrs, och):
        """Does the library write this head in its final layout from the GEMM epilogue (ConvGemmParams.head_attrs)?"""
        if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or attrs < 7:
            return False
        p = S.ConvGemmParams()
        p.A, p.NB, p.IH, p.IW, p.Cin, p.ldA = xin.ptr(), x.N, x.H, x.W, conv.in_channels, xin.ld
        p.W, p.Nout, p.wtaps = pk["wf"].data_ptr(), cout, 1
        p.OH, p.OW, p.sh, p.sw = x.H, x.W, 1, 1
        p.oh_mul, p.ow_mul, p.OHf, p.OWf = 1, 1, x.H, x.W
        p.nclasses = 1
        _fill_class(p.cls[0], [(0, 0, 0)], 0, 0)
        p.epi, p.pipe, p.head_attrs, p.head_och = S.EPI_F32_BIAS, self.rt.gemm_pipe, attrs, och
        try:
            hip.call("ryolo_conv_gemm_plan", p, S.I(), S.I())
        except RuntimeError:
            return False
        return True

    def _head_fused(self, conv, pk, x, xin, na, attrs, och, implicit_a, implicit_m, sparse, fold_a=False):
        """r05: the head GEMM writes [B, na, gs, gs, attrs] itself — bias, ImplicitM and the permute in its epilogue, the compact objectness
        logits for the fused loss beside it; no row-major fp32 intermediate, no ryolo_head_finish_fwd pass.  Backward without the
        pre-ImplicitM activations: the GEMMs run on the UNSCALED head gradient (dx through a data-gradient image that carries ImplicitM,
        PackEntry.wd_scale; the weight gradient into a scratch G) and ryolo_head_wgrad_finish forms dW = m G, db = m s and
        dm = rowdot(W, G) + b s  (= sum dout (x . W + b): the chain through the multiply of model/neck.py:186)."""
        rt = self.rt
        cout, coutp, M = conv.out_channels, pk["CoutP"], x.M
        mptr = implicit_m.data_ptr() if implicit_m is not None else None
        out = self.f32(x.N, na, x.H, x.W, attrs)
        xobj = self.f32(x.N, na, x.H, x.W) if sparse else None
        aptr = implicit_a.data_ptr() if fold_a else None
        bias_ptr = conv.bias.data_ptr()
        if fold_a:
            fbias = self.f32(cout)
            self._call(self.fwd, "ryolo_head_bias_fold", conv.weight.data_ptr(), conv.bias.data_ptr(), aptr, cout, conv.in_channels, fbias.data_ptr())
            bias_ptr = fbias.data_ptr()
        self._gemm(self.fwd, xin, xin.ptr(), pk["wf"], cout, 1, conv.in_channels, x.H, x.W, 1, [([(0, 0, 0)], 0, 0)], S.EPI_F32_BIAS,
                   out.data_ptr(), attrs, bias=bias_ptr, head=(attrs, och, mptr, xobj.data_ptr() if xobj is not None else None))
        rec = dict(out=out, dout=None, preobj=None, och=och, xobj=xobj)
        self.heads.append(rec)
        if self.training:
            if implicit_m is not None and pk.get("wd_scale") is None:
                pk["wd_scale"] = implicit_m
                rt._pack_table = None
            dout = self.f32(x.N, na, x.H, x.W, attrs)
            rec["dout"] = dout
            dpre = torch.zeros((M, coutp), dtype=BF16, device=self.adev)       # pad columns stay zero
            self.keep.append(dpre)
            nblk = x.N * ((x.H * x.W + 127) // 128)
            scratch = self.f32(max((nblk + 64) * 2 * cout, ((M + 255) // 256 + 64) * max(coutp, x.C)))
            G = s = None
            if implicit_m is not None:
                G = self.f32(cout, conv.in_channels, zero=True)                # (ryolo_head_wgrad_finish clears both again)
                s = self.f32(cout, zero=True)

            class _G:                      # geometry shim so dpre can be used as a gathered operand
                N, H, W, ld = x.N, x.H, x.W, coutp
                span_bytes = M * coutp * 2

            def backward():
                # dpre = bf16(dout) and its column sums (the bias gradient, or s with ImplicitM) in one pass over dout
                self._call(self.bwd, "ryolo_head_finish_bwd", dout.data_ptr(), None, coutp, None, x.N, x.H, na, attrs, dpre.data_ptr(), coutp,
                           s.data_ptr() if s is not None else rt.grad_ptr(conv.bias), None, scratch.data_ptr())
                rec["dout_slot"] = len(self.bwd) - 1
                self._wgrad(conv, _G, dpre.data_ptr(), coutp, xin, dW=G.data_ptr() if G is not None else None)
                self._dgrad(conv, pk, _G, dpre.data_ptr(), coutp, x)
                if implicit_m is not None:
                    self._call(self.bwd, "ryolo_head_wgrad_finish", G.data_ptr(), s.data_ptr(), conv.weight.data_ptr(), conv.bias.data_ptr(),
                               mptr, aptr, cout, conv.in_channels, rt.grad_ptr(conv.weight), rt.grad_ptr(conv.bias), rt.grad_ptr(implicit_m),
                               rt.grad_ptr(implicit_a) if fold_a else None)
                    if self.rt.wgrad_stream:
                        self.side_idx.add(len(self.bwd) - 1)       # behind its weight gradient on the side stream
                if implicit_a is not None and not fold_a:
                    self._call(self.bwd, "ryolo_colsum_bf16", x.gptr(), x.ld, M, x.C, x.C, rt.grad_ptr(implicit_a), scratch.data_ptr())
            self._pending_bwd.append(backward)
        return out

    def set_image(self, imgs):
        """Use the caller's [B,3,S,S] fp32 tensor in place (no staging copy) when it is contiguous."""
        if self._img_slot is not None:
            fn, args, name = self.fwd[self._img_slot]
            self.fwd[self._img_slot] = (fn, (imgs.data_ptr(),) + args[1:], name)
        for st in self._img_structs:                # direct stem: parameter blocks are passed by reference, patch them in place
            st.img = imgs.data_ptr()
        self._img_ref = imgs                        # keep alive until the next call

    def set_head_grad(self, rec, grad, compact=None):
        """Point the head backward at `grad`.  compact = (objgrad, owner pointer, objectness channel, keep-alive) from
        lib/loss.py compact_head_grad(): the sparse entry point rebuilds unmatched rows from the compact objectness gradients."""
        fn, args, name = self.bwd[rec["dout_slot"]]
        tail = args[5:] if name == "ryolo_head_finish_bwd_sparse" else args[1:]
        if compact is not None:
            objgrad, owner, och, keep = compact
            preobj = rec["preobj"].data_ptr() if rec.get("preobj") is not None and rec["och"] == och else None
            self.bwd[rec["dout_slot"]] = (hip.lib().ryolo_head_finish_bwd_sparse, (grad.data_ptr(), objgrad.data_ptr(), owner, och, preobj) + tail,
                                          "ryolo_head_finish_bwd_sparse")
            rec["dout_ref"] = (grad, objgrad, keep)
        else:
            self.bwd[rec["dout_slot"]] = (hip.lib().ryolo_head_finish_bwd, (grad.data_ptr(),) + tail, "ryolo_head_finish_bwd")
            rec["dout_ref"] = grad

    # ------------------------------------------------------------------ plan assembly
    def begin(self):
        self._pending_bwd = []

    def finish(self):
        """Backward tape = the per-op backward emitters in reverse forward order (plan-time accumulate/store resolution
        relies on this order)."""
        for emit in reversed(self._pending_bwd):
            emit()
        self._pending_bwd = None
        for lane, plist in self._wgrads.items():
            ws = torch.empty(self._wgrad_ws_bytes[lane], dtype=torch.uint8, device=self.adev)
            self.keep.append(ws)
            for p in plist:
                p.partial = ws.data_ptr()
