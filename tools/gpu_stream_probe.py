"""Does overlapping two batch halves on two streams help the J=3 forward / inverse (tails of the small launches)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
x = torch.randn(128, 3, 512, 512, device=dev)
xfm = pw.DWTForward(J=3, wave='db4', mode='symmetric').to(dev)
ifm = pw.DWTInverse(wave='db4', mode='symmetric').to(dev)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
xa, xb = x[:64].contiguous(), x[64:].contiguous()
def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        ya = xfm(xa); ra = ifm(ya)
    with torch.cuda.stream(s2):
        yb = xfm(xb); rb = ifm(yb)
    cur.wait_stream(s1); cur.wait_stream(s2)
def one_stream():
    y = xfm(x); r = ifm(y)
def halves_one_stream():
    ya = xfm(xa); ra = ifm(ya); yb = xfm(xb); rb = ifm(yb)
with torch.no_grad():
    for name, fn in (('one stream, full batch', one_stream), ('one stream, two halves', halves_one_stream), ('two streams, two halves', two_streams),
                     ('one stream, full batch', one_stream), ('two streams, two halves', two_streams)):
        print('%-26s fwd+inv %.4f ms' % (name, t(fn)), flush=True)
