"""GPU probe: which device tensors do the two part-batch programs of one forward BOTH point to, and through which descriptor fields?
(read-only weights are expected; anything a kernel writes is a bug)"""
import os, sys, ctypes as C, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import i2r_amd
from i2r_amd import synth, models, engine, cabi
from _golden import setup
cfg, sd, _, _, _, _ = setup(sys.argv[1] if len(sys.argv) > 1 else "tph_l21")
full = [6, 4, 4, 2, 2, 1, 1, 1, 2, 5]
x, m, _ = synth.make_inputs(full, 256, 192, seed=3)
net = models.interformer.get_pose_net(cfg, is_train=False); net.load_state_dict(sd, strict=True); net = net.cuda()
eng = net.engine()
net(x.cuda(), m.cuda(), full); torch.cuda.synchronize()
A, B = eng.last_programs[:2]
def tensors(obj, out, seen):
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            out[obj.data_ptr()] = obj
    elif isinstance(obj, dict):
        for v in obj.values():
            tensors(v, out, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            tensors(v, out, seen)
ta, tb = {}, {}
tensors(A.keep, ta, set()); tensors(B.keep, tb, set())
print("tensors kept: A %d, B %d, same storage in both: %d" % (len(ta), len(tb), len(set(ta) & set(tb))))
def ranges(ts):
    return sorted((p, p + t.numel() * t.element_size()) for p, t in ts.items())
ra, rb = ranges({p: t for p, t in ta.items() if p not in tb}), ranges({p: t for p, t in tb.items() if p not in ta})
ov = [(a, b) for a in ra for b in rb if a[0] < b[1] and b[0] < a[1]]
print("overlapping private ranges:", len(ov))
shared = sorted(set(ta) & set(tb))
def fields(P):
    hits = collections.Counter()
    for kind, lane, st in P.ops:
        if st is None:
            continue
        def walk(s, prefix):
            for f in s._fields_:
                name, typ = f[0], f[1]
                v = getattr(s, name)
                if isinstance(v, C.Structure):
                    walk(v, prefix + name + "."); continue
                if isinstance(v, C.Array):
                    for i, e in enumerate(v):
                        if isinstance(e, C.Structure):
                            walk(e, prefix + name + "[].")
                        elif isinstance(e, int) and e:
                            check(e, prefix + name + "[]")
                    continue
                if isinstance(v, int) and v:
                    check(v, prefix + name)
        def check(v, name):
            for p in shared:
                t = ta[p]
                if p <= v < p + max(1, t.numel() * t.element_size()):
                    hits[(type(st).__name__, name)] += 1
        walk(st, "")
    return hits
for k, n in sorted(fields(A).items()):
    print("  %-22s %-22s %d" % (k[0], k[1], n))
