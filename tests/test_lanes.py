"""CPU: memory safety of the stream-lane schedule of a launch program (engine.Program fork / join / xsync + the arena's buffer reuse).
The HRFormer tower's program is built on the host (nothing is launched) and replayed symbolically: every op reads / writes byte ranges
of arena buffers on its lane; two accesses to overlapping ranges of which at least one is a write must be ordered by the happens-before
relation the fork / join / xsync ops create (vector clocks per lane).  A race here would not necessarily show up in a parity test."""
import ctypes as C

import pytest
import torch

import i2r_amd  # noqa: F401
from i2r_amd import arch, cabi, config, engine, synth


def _ranges_of(program):
    """sorted (start, end) of every arena buffer of the program, to map raw pointers back to buffers"""
    rs = sorted((t.data_ptr(), t.data_ptr() + t.numel() * 4) for t in program.keep if isinstance(t, torch.Tensor) and t.dtype == torch.float32)
    return rs


def _accesses(kind, st):
    """(reads, writes) as lists of raw pointers of one op"""
    g = lambda name: getattr(st, name)
    if kind == cabi.OP_CONV:
        return [p for p in (st.in_, st.in2, st.res1, st.res2, st.res_post) if p], [st.out]
    if kind == cabi.OP_CONV_GROUP:
        r, w = [], []
        for i in range(st.n):
            d = st.d[i].contents
            r += [p for p in (d.in_, d.in2, d.res1, d.res2, d.res_post) if p]
            w.append(d.out)
        return r, w
    if kind == cabi.OP_STEM:
        return [], [g("out")]
    if kind in (cabi.OP_HRT_ATTN, cabi.OP_HRT_MLP):
        return [g("x")], [g("out")]
    if kind == cabi.OP_LAYERNORM:
        return [g("in_")], [g("out")]
    if kind == cabi.OP_WINATTN:
        return [g("qkv")], [g("out")]
    if kind == cabi.OP_DWCONV:
        return [g("in_")], [g("out")]
    if kind == cabi.OP_UPSAMPLE:
        return [p for p in (g("low"), g("low2"), g("low3"), g("res")) if p], [g("out")]
    if kind == cabi.OP_FUSE_UP:
        return [p for p in (g("base"), g("t1"), g("t2")) if p], [g("out")]
    if kind == cabi.OP_MAXPOOL:
        return [g("in_")], [g("out")]
    if kind == cabi.OP_CONV1X1_LP:
        return [p for p in (g("x"), g("res1"), g("res2"), g("res_post")) if p], [g("out")]
    if kind == cabi.OP_CONV1X1_PAIR:
        return [p for p in (g("x"), g("res")) if p], [p for p in (g("y"), g("z")) if p]
    raise AssertionError("op kind %d not modelled" % kind)


def _field_names(st):
    return [f[0] for f in st._fields_]


def check_program(program):
    """-> number of cross-lane ordered conflicts verified; raises on a race"""
    import bisect
    rs = _ranges_of(program)
    starts = [a for a, _ in rs]

    def buf(ptr):
        i = bisect.bisect_right(starts, ptr) - 1
        return i if i >= 0 and ptr < rs[i][1] else None  # weights / tables are read-only: not arena buffers

    clock = [[0, 0, 0, 0] for _ in range(4)]   # vector clock of each lane
    slot_clock = {}                            # record slot -> clock snapshot of the recording lane (I2R_OP_RECORD / I2R_OP_WAIT)
    last_w = {}   # buffer -> (lane, clock snapshot) of the last write
    reads = {}    # buffer -> list of (lane, snapshot) since the last write
    checked = 0

    def hb(snap, lane_from, lane_to):  # the access with snapshot `snap` on lane_from happens before the current point of lane_to
        return lane_from == lane_to or clock[lane_to][lane_from] >= snap[lane_from]

    for kind, lane, st in program.ops:
        if kind == cabi.OP_FORK:   # lanes of the mask see everything of lane 0
            for l in range(1, 4):
                if lane & (1 << l):
                    clock[l] = [max(a, b) for a, b in zip(clock[l], clock[0])]
            continue
        if kind == cabi.OP_JOIN:   # lane 0 sees everything of the lanes of the mask
            for l in range(1, 4):
                if lane & (1 << l):
                    clock[0] = [max(a, b) for a, b in zip(clock[0], clock[l])]
            continue
        if kind == cabi.OP_LANE_FLAGS:
            continue
        if kind == cabi.OP_RECORD:  # point-to-point: the slot remembers what the recording lane has seen and done
            slot_clock[(lane >> 8) & 7] = list(clock[lane & 3])
            continue
        if kind == cabi.OP_WAIT:    # the waiting lane sees everything the slot's last record saw
            clock[lane & 3] = [max(a, b) for a, b in zip(clock[lane & 3], slot_clock[(lane >> 8) & 7])]
            continue
        if kind == cabi.OP_XSYNC:  # all-to-all among the lanes of the mask
            ls = [l for l in range(4) if lane & (1 << l)]
            m = [max(clock[l][i] for l in ls) for i in range(4)]
            for l in ls:
                clock[l] = list(m)
            continue
        clock[lane][lane] += 1
        snap = list(clock[lane])
        r, w = _accesses(kind, st)
        for ptr in r:
            b = buf(ptr)
            if b is None:
                continue
            if b in last_w:
                wl, ws = last_w[b]
                assert hb(ws, wl, lane), "read on lane %d races with the write on lane %d (op kind %d)" % (lane, wl, kind)
                checked += wl != lane
            reads.setdefault(b, []).append((lane, snap))
        for ptr in w:
            b = buf(ptr)
            assert b is not None, "op kind %d writes outside the arena" % kind
            if b in last_w:
                wl, ws = last_w[b]
                assert hb(ws, wl, lane), "write on lane %d races with the write on lane %d (op kind %d)" % (lane, wl, kind)
                checked += wl != lane
            for rl, rsnap in reads.get(b, []):
                assert hb(rsnap, rl, lane), "write on lane %d races with a read on lane %d (op kind %d): buffer reused too early" % (lane, rl, kind)
                checked += rl != lane
            last_w[b] = (lane, snap)
            reads[b] = []
    return checked


@pytest.mark.parametrize("cname,precision,n,h,w", [("hrt_192_p4_b4", "bf16", 3, 256, 192), ("hrt_192_p4_b4", "fp32", 2, 256, 192),
                                                   ("coco_hrt_288_p2_b4", "fp16", 2, 384, 288)])
def test_hrformer_lane_schedule_has_no_race(cname, precision, n, h, w):
    cfg = config.load_config(cname)
    sd = synth.make_state_dict(arch.param_spec(cfg))
    dev = torch.device("cpu")
    pk = engine.Packer(sd, dev, precision)
    tower = engine.HRFormerB(pk, "singleformer.")
    P = engine.Program(dev)
    P.store_dt = pk.dtype
    ys, _ = tower.emit(P, n, h, w)
    kinds = [k for k, _, _ in P.ops]
    # 7 modules in 3 stages: one round of records per module (2 + 4 x 3 + 2 x 4 lanes), every lane waits for every other lane's record
    assert kinds.count(cabi.OP_XSYNC) == 0 and kinds.count(cabi.OP_FORK) == 3 and kinds.count(cabi.OP_JOIN) == 3
    assert kinds.count(cabi.OP_RECORD) == 2 + 4 * 3 + 2 * 4 and kinds.count(cabi.OP_WAIT) == 2 * 1 + 4 * 3 * 2 + 2 * 4 * 3
    assert {lane for k, lane, _ in P.ops if k not in cabi.SYNC_OPS} == {0, 1, 2, 3}
    assert check_program(P) > 50


def test_checker_catches_a_missing_sync():
    """the same program with its point-to-point waits removed must be reported as racy (the checker is not vacuous)"""
    cfg = config.load_config("hrt_192_p4_b4")
    sd = synth.make_state_dict(arch.param_spec(cfg))
    dev = torch.device("cpu")
    pk = engine.Packer(sd, dev, "bf16")
    P = engine.Program(dev)
    P.store_dt = pk.dtype
    engine.HRFormerB(pk, "singleformer.").emit(P, 2, 256, 192)
    P.ops = [op for op in P.ops if op[0] != cabi.OP_WAIT]
    with pytest.raises(AssertionError, match="races"):
        check_program(P)


def test_records_hand_buffers_over_only_behind_all_waits():
    """Program.records / wait / all_waited (round 6): a buffer one lane released BEFORE the records may be re-used by another lane only
    behind the point where every lane has waited for every record (all_waited); what a lane releases AFTER its record stays its own until
    the next round; join flushes everything."""
    P = engine.Program(torch.device("cpu"))
    P.fork(6)
    P.lane_ctx = 1
    a = P.alloc(1, 4, 4, 16)          # lane 1's buffer ...
    P.release(a)                      # ... released ahead of the records
    slots = P.records([0, 1, 2])
    assert sorted(slots) == [0, 1, 2] and len(set(slots.values())) == 3
    P.lane_ctx = 2
    b = P.alloc(1, 4, 4, 16)
    assert b.t.data_ptr() != a.t.data_ptr(), "lane 2 must not get lane 1's buffer before the waits"
    P.lane_ctx = 1
    c = P.alloc(1, 4, 4, 16)          # (lane 1 itself does not get it back either: it went into the round's snapshot)
    assert c.t.data_ptr() != a.t.data_ptr()
    P.release(c)                      # released AFTER the record: lane 1's own again at once, nobody else's until the next round
    for ln in (0, 1, 2):
        for lj in (0, 1, 2):
            if ln != lj:
                P.wait(ln, slots[lj])
    P.all_waited()
    P.lane_ctx = 0
    d = P.alloc(1, 4, 4, 16)
    assert d.t.data_ptr() == a.t.data_ptr(), "behind all_waited the pre-record release is everybody's"
    e = P.alloc(1, 4, 4, 16)
    assert e.t.data_ptr() != c.t.data_ptr(), "lane 1's post-record release is still lane 1's"
    P.lane_ctx = 1
    f = P.alloc(1, 4, 4, 16)
    assert f.t.data_ptr() == c.t.data_ptr()
    P.join(6)
    kinds = [k for k, _, _ in P.ops]
    assert kinds.count(cabi.OP_RECORD) == 3 and kinds.count(cabi.OP_WAIT) == 6 and kinds[0] == cabi.OP_LANE_FLAGS and kinds[-1] == cabi.OP_JOIN
    # RECORD carries lane | slot << 8 | consumer lanes << 16
    recs = [lane for k, lane, _ in P.ops if k == cabi.OP_RECORD]
    assert all((r >> 16) & 15 == 7 for r in recs) and sorted(r & 3 for r in recs) == [0, 1, 2]


def test_lanes_independent_needs_probed_overlap_for_every_pair(monkeypatch):
    """engine.lanes_independent: the device-side lane synchronisation is only allowed on streams the probe saw running beside the caller's
    stream and beside every lane chosen before them, and only for the caller stream the probe used"""
    class S:
        def __init__(self, h):
            self.cuda_stream = h
    ok = {"alone_ms": 0.06, "together_ms": 0.08, "overlap": True}
    bad = {"alone_ms": 0.06, "together_ms": 0.12, "overlap": False}
    probe = [{"stream": hex(0x10), "candidate": 1, "vs_caller": ok, "role": "lane1"},
             {"stream": hex(0x20), "candidate": 2, "vs_caller": bad, "role": "spare (shares a hardware queue)"},
             {"stream": hex(0x30), "candidate": 3, "vs_caller": ok, "vs_lane1": ok, "role": "lane2"},
             {"stream": hex(0x40), "role": "lane3 (no independent queue left: shares)"},
             {"stream": hex(0x50), "candidate": 5, "probe": "off", "role": "lane3"}]
    monkeypatch.setitem(engine._LANE_PROBE, "cuda:7", probe)
    monkeypatch.setitem(engine._LANE_CALLER, "cuda:7", 0)
    assert engine.lanes_independent("cuda:7", [S(0x10), S(0x30)], 0)
    assert not engine.lanes_independent("cuda:7", [S(0x10), S(0x30)], 0x99)      # another caller stream: nothing is known about it
    assert not engine.lanes_independent("cuda:7", [S(0x10), S(0x20)], 0)         # shares a queue with the caller
    assert not engine.lanes_independent("cuda:7", [S(0x10), S(0x30), S(0x40)], 0)  # took a shared queue
    assert not engine.lanes_independent("cuda:7", [S(0x10), S(0x30), S(0x50)], 0)  # never probed
    assert not engine.lanes_independent("cuda:3", [S(0x10)], 0)                  # no probe record for the device
