import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import torch, torch.nn.functional as F
import i2r_amd
from i2r_amd import engine, synth
from _gpu_util import to_act, from_act, run
import i2r_cpu_hrformer as H
DEV='cuda:0'
def _rand(shape, key, scale=1.0): return torch.from_numpy(synth._sym(7, key, tuple(shape), scale))
c, heads, h, w = 78, 2, 64, 48
tag='x'; p='b.attn.attn'
sd = {"b.norm1.weight": _rand((c,), "n1w", 0.3) + 1.0, "b.norm1.bias": _rand((c,), "n1b", 0.2)}
for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
    sd["%s.%s.weight" % (p, n)] = _rand((c, c), n + "w", 2.0 * (3.0 / c) ** 0.5)
    sd["%s.%s.bias" % (p, n)] = _rand((c,), n + "b", 0.3)
x = _rand((2, c, h, w), "xx")
t = x.permute(0, 2, 3, 1)
n1 = F.layer_norm(t, (c,), sd["b.norm1.weight"], sd["b.norm1.bias"], 1e-6)
P = engine.Program(torch.device(DEV)); pk = engine.Packer(sd, torch.device(DEV))
xa = to_act(P, x)
n1a = P.layernorm(xa, pk.ln("b.norm1", c))
qkvw = pk.qkv(p, c)
qkv = P.conv(n1a, qkvw)
a = P.winattn(qkv, qkvw.bias, c, heads)
out = P.conv(a, pk.linear_as_conv(sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"]), res1=xa)
run(P)
print('ln err', (from_act(n1a).permute(0,2,3,1) - n1).abs().max().item())
q = F.linear(n1, sd[p+'.q_proj.weight'], sd[p+'.q_proj.bias']); k = F.linear(n1, sd[p+'.k_proj.weight'], sd[p+'.k_proj.bias']); v = F.linear(n1, sd[p+'.v_proj.weight'], sd[p+'.v_proj.bias'])
g = qkv.t.view(2,h,w,240).cpu()
print('q err', (g[...,:78]-q).abs().max().item(), 'k err', (g[...,80:158]-k).abs().max().item(), 'v err', (g[...,160:238]-v).abs().max().item(), 'pads', g[...,78:80].abs().max().item())
# attention before out_proj
import math
sdi = dict(sd); sdi[p+'.out_proj.weight']=torch.eye(c); sdi[p+'.out_proj.bias']=torch.zeros(c)
aref = H.window_attention(sdi, p, n1, heads)
ag = from_act(a).permute(0,2,3,1)
d = (ag-aref).abs()
print('attn err', d.max().item(), 'at', (d==d.max()).nonzero()[0].tolist(), 'mean', d.mean().item())
print('rows err by y', d.amax(dim=(0,2,3))[:10], d.amax(dim=(0,1,3))[:10])
ref = (t + H.window_attention(sd, p, n1, heads))
print('final err', (from_act(out).permute(0,2,3,1)-ref).abs().max().item())
