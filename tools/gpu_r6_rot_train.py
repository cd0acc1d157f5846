"""Round 6: the training step (forward + backward to the input) of ScatLayer(biort='near_sym_b_bp') - two launches of the fused ScatLayer
kernels per direction (scatternet/lowlevel.py ScatLayerj1_rot_train_f) against the chain it replaces (FWD_J1_ROT on the tile kernel, the
magnitudes in the tensor library, the backward on seven single-axis launches), same process; the plain layers beside it."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
from pytorch_wavelets_amd.scatternet import lowlevel as sl_ll
dev = 'cuda:0'; sync = torch.cuda.synchronize
short = lambda ks: ','.join(sorted(set(k.split('(')[0].strip().replace('float', 'f') for k in ks if not k.endswith(')'))))
for shape in ((64, 3, 256, 256), (256, 3, 256, 256), (64, 3, 512, 512), (16, 3, 1024, 1024), (128, 3, 224, 224)):
    x = torch.randn(*shape, device=dev)
    rec = {'shape': shape}
    for biort in ('near_sym_a', 'near_sym_b', 'near_sym_b_bp'):
        sl = pw.ScatLayer(biort=biort).to(dev)
        for fused in ((True, False) if biort == 'near_sym_b_bp' else (True,)):
            sl_ll.ROT_TRAIN_FUSED = fused
            try:
                xg = x.clone().requires_grad_(True)
                def step():
                    z = sl(xg)
                    torch.autograd.grad(z, xg, z)
                c0 = pw.launch_count(); step(); ks = pw.kernels_since(c0)
                t = min(bench.time_seq_fn(step, 20, sync) for _ in range(3))
            finally:
                sl_ll.ROT_TRAIN_FUSED = True
            tag = biort + ('' if fused else '_chain')
            rec[tag + '_fwdbwd_ms'] = round(t, 4)
            rec[tag + '_k'] = short(ks)
    print(json.dumps(rec), flush=True)

# ScatLayerj2 with the band-pass tables: its second-order block (the first-order layer on 6 C magnitude planes) through the same launches
for shape in ((64, 3, 256, 256), (32, 3, 512, 512)):
    x = torch.randn(*shape, device=dev)
    rec = {'shape': shape, 'layer': 'ScatLayerj2(near_sym_b_bp, qshift_b_bp)'}
    sl = pw.ScatLayerj2(biort='near_sym_b_bp', qshift='qshift_b_bp').to(dev)
    for fused in (True, False):
        sl_ll.ROT_TRAIN_FUSED = fused
        try:
            with torch.no_grad():
                c0 = pw.launch_count(); sl(x); ks = pw.kernels_since(c0)
                ti = min(bench.time_seq_fn(lambda: sl(x), 20, sync) for _ in range(2))
            xg = x.clone().requires_grad_(True)
            def step():
                z = sl(xg)
                torch.autograd.grad(z, xg, z)
            tt = min(bench.time_seq_fn(step, 10, sync) for _ in range(2))
        finally:
            sl_ll.ROT_TRAIN_FUSED = True
        tag = 'layer' if fused else 'chain'
        rec[tag + '_inference_ms'] = round(ti, 4); rec[tag + '_fwdbwd_ms'] = round(tt, 4); rec[tag + '_k'] = short(ks)
    print(json.dumps(rec), flush=True)
