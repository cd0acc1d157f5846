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
