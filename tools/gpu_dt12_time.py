"""GPU probe: in-kernel cycle counters of the fused DTCWT kernel (timing build: WL_LIB=ab/libwl_time.so)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
with torch.no_grad():
    for shape in ((64, 3, 512, 512), (256, 3, 256, 256)):
        m = pw.DTCWTForward(J=2).to(dev)
        x = torch.randn(*shape, device=dev)
        for _ in range(5):
            yl, yh = m(x)
        torch.cuda.synchronize()
        v = yl[0, 0, 0, 16:22].tolist()
        print(json.dumps({'lib': os.environ.get('WL_LIB', ''), 'shape': shape, 'kernel': pw.last_kernel(),
                          'level1_kcycles(total,barrier)': v[0:2], 'level2': v[2:4], 'stager': v[4:6]}))
