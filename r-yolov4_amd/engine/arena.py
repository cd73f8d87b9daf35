"""Liveness-based placement of a plan's activation and activation-gradient buffers in ONE arena.

A plan (engine/graph.py) is a fixed sequence of launches, so every buffer has a known first and last use.  The network is planned
twice: a DRY pass hands out virtual addresses (tag bit 60, never dereferenced — the C-ABI plan helpers only do arithmetic on them),
this module finds every launch that mentions a buffer by scanning the launch arguments and parameter blocks for tagged addresses,
turns the uses into lifetimes on the main stream's timeline, and packs the buffers (largest first, lowest free offset); the second
pass emits the same launches with arena addresses.  In training every forward activation lives until its backward use, so the gain is
the gradient twins: the gradients of the early (largest) layers are written at the end of backward, into the memory of the late layers'
activations and gradients, dead by then.  In inference plans the activations themselves are recycled.

Concurrency (Graph.run): a launch on a side stream may run later / earlier than its tape position, so its uses are widened to the
interval of main-stream positions it can overlap with:
  forward, lane 1 (a sibling branch)   [fork point of its branch, first main-stream launch after the next join)
  forward, lane 2 (detection tails)    [fork point, end of the forward tape]
  backward, weight gradient n          [its position, position of weight gradient n + LAG) — Graph.run makes the main stream wait for
                                       weight gradient n before it enqueues number n + LAG (a bounded lag; without it every conv input
                                       and every dY would have to stay until the end-of-tape join)
Two buffers share memory only if their lifetimes are disjoint.  Serial replays (timing passes, CPU) are a special case of this order.
"""
import ctypes as C

VTAG = 1 << 60
_IDSHIFT = 42
ALIGN = 256


class Virt:
    """Stand-in for a device tensor in the dry pass: an address that nobody dereferences."""

    def __init__(self, ident, nbytes):
        self.ident, self.nbytes = ident, nbytes

    def data_ptr(self):
        return VTAG | (self.ident << _IDSHIFT)


def _tagged(obj, out):
    """Collect the buffer ids of every tagged address inside a launch argument (int, ctypes struct / array, nested)."""
    if obj is None:
        return
    if isinstance(obj, int):
        if obj >> 60 == 1:
            out.add((obj & (VTAG - 1)) >> _IDSHIFT)
        return
    if isinstance(obj, C.Structure):
        for name, *_ in obj._fields_:
            _tagged(getattr(obj, name), out)
        return
    if isinstance(obj, C.Array):
        if issubclass(obj._type_, (C.Structure, C.Array)) or obj._type_ in (C.c_void_p, C.c_uint64, C.c_int64, C.c_size_t):
            for e in obj:
                _tagged(e, out)


class Layout:
    def __init__(self):
        self.off = {}            # buffer ident (2k: activation of Buf k, 2k + 1: its gradient) -> byte offset
        self.total = 0
        self.sum_bytes = 0
        self.lag = 0
        self.names = None        # (forward, backward) kernel-name sequences of the dry pass: the second pass must repeat them
        self.life = {}           # ident -> (first, last) on the main stream's timeline (forward then backward positions)


def lifetimes(g, lag):
    F, B = len(g.fwd), len(g.bwd)
    main_fwd = [i for i in range(F) if i not in g.fwd_side]
    joins = sorted(g.fwd_join)

    def fwd_interval(i):
        ent = g.fwd_side.get(i)
        if ent is None:
            return i, i
        f = i
        while not g.fwd_side[f][0]:
            f -= 1
        if ent[1] == 2:
            return f, F - 1
        j = next((j for j in joins if j > i), F)
        m = next((m for m in main_fwd if m >= j), F)          # where the main stream actually waits
        return f, m - 1

    side = sorted(g.side_idx)
    pos = {t: n for n, t in enumerate(side)}

    def bwd_interval(i):
        n = pos.get(i)
        if n is None:
            return F + i, F + i
        return F + i, F + (side[n + lag] - 1 if lag and n + lag < len(side) else B - 1)

    life, pinned = {}, set()
    for tape, ival in ((g.fwd, fwd_interval), (g.bwd, bwd_interval)):
        tid = id(tape)
        for i in range(len(tape)):
            args = g._raw.get((tid, i))
            if args is None:
                continue
            ids = set()
            for a in args:
                _tagged(a, ids)
            if not ids:
                continue
            lo, hi = ival(i)
            for k in ids:
                a = life.get(k)
                life[k] = (lo, hi) if a is None else (min(a[0], lo), max(a[1], hi))
    return life


def pack(sizes, life):
    """sizes: ident -> bytes; life: ident -> (first, last).  Greedy by size: each buffer takes the lowest offset free of every already
    placed buffer whose lifetime intersects its own."""
    order = sorted(sizes, key=lambda k: (-sizes[k], life[k][0], k))
    placed = []                                      # (offset, end, first, last)
    off = {}
    total = 0
    for k in order:
        n = (sizes[k] + ALIGN - 1) // ALIGN * ALIGN
        lo, hi = life[k]
        busy = sorted((o, e) for o, e, a, b in placed if not (b < lo or hi < a))
        at = 0
        for o, e in busy:
            if o - at >= n:
                break
            at = max(at, e)
        off[k] = at
        placed.append((at, at + n, lo, hi))
        total = max(total, at + n)
    return off, total


def plan(g, lag):
    """g: a finished DRY Graph.  Returns the Layout for the real pass."""
    life = lifetimes(g, lag)
    sizes = {}
    F, B = len(g.fwd), len(g.bwd)
    for ident, v in g._virt.items():
        sizes[ident] = v.nbytes
        if ident not in life:                        # allocated but never handed to a launch: keep it apart for the whole plan
            life[ident] = (0, max(F + B - 1, 0))
    lay = Layout()
    lay.off, lay.total = pack(sizes, life)
    lay.sum_bytes = sum((n + ALIGN - 1) // ALIGN * ALIGN for n in sizes.values())
    lay.lag = lag
    lay.life = life
    lay.sizes = sizes
    lay.names = ([e[2] for e in g.fwd], [e[2] for e in g.bwd])
    return lay


def check(lay):
    """No two buffers with intersecting lifetimes intersect in memory (test helper)."""
    items = [(lay.off[k], lay.off[k] + lay.sizes[k], lay.life[k]) for k in lay.off]
    items.sort()
    for i, (o, e, (a, b)) in enumerate(items):
        for o2, e2, (a2, b2) in items[i + 1:]:
            if o2 >= e:
                break
            if not (b < a2 or b2 < a):
                return False
    return True
