#!/usr/bin/env python
"""Benchmark of the hot path: DWTForward + DWTInverse, J=3 db4 symmetric, N x 3 x 512 x 512 fp32
(BASELINE.json configs[1]), synthetic data resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one forward + one inverse transform of this rank's batch (N=128 planes-of-3 per GPU,
weak scaling: the batch dimension shards with no data-path collective; the only collective is the
one-off broadcast of the filter banks from rank 0).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable


def algorithmic_bytes_fwd(N, C, H, W, J, L, itemsize):
    """every input element read once, every output element written once."""
    n_in = H * W
    n_out = 0
    h, w = H, W
    for _ in range(J):
        h, w = (h + L - 1) // 2, (w + L - 1) // 2
        n_out += 3 * h * w
    n_out += h * w
    return N * C * (n_in + n_out) * itemsize


def cpu_baseline(args):
    """The oracle's C port (oracle/dwt_port.c, OpenMP) - or the numpy oracle if the port is not
    built - timed on this host's cores on a bounded sample of the same workload."""
    import numpy as np
    from pytorch_wavelets_amd import filters
    h0, h1 = filters.dwt_analysis_taps('db4')
    g0, g1 = filters.dwt_synthesis_taps('db4')
    rng = np.random.RandomState(0)
    try:
        from oracle import dwt_port
        ncores = os.cpu_count() or 1
        n = max(2, min(64, ncores))
        x = rng.randn(n, 3, 512, 512).astype(np.float32)
        dwt_port.fwd_inv(x, 3, h0, h1, g0, g1, 'symmetric', threads=ncores)    # warm-up
        reps, t0 = 0, time.perf_counter()
        while reps < 3 or time.perf_counter() - t0 < 10.0:
            dwt_port.fwd_inv(x, 3, h0, h1, g0, g1, 'symmetric', threads=ncores)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        return {'value': round(x.size / dt / 1e6, 2), 'unit': 'Mpixels/s', 'cores': ncores, 'kind': 'port',
                'sample': 'oracle/dwt_port.c (OpenMP fp32 port of the oracle), fwd+inv J=3 db4 symmetric on '
                          '%dx3x512x512, %d reps' % (n, reps)}
    except Exception:
        from oracle import wavelet_oracle as wo
        x = rng.randn(2, 3, 512, 512).astype(np.float32).astype(np.float64)
        t0 = time.perf_counter()
        reps = 0
        while reps < 2 or time.perf_counter() - t0 < 10.0:
            yl, yh = wo.dwt_forward(x, 3, h0, h1, h0, h1, 'symmetric')
            wo.dwt_inverse(yl, yh, g0, g1, g0, g1, 'symmetric')
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        return {'value': round(x.size / dt / 1e6, 2), 'unit': 'Mpixels/s', 'cores': 1, 'kind': 'port',
                'sample': 'numpy float64 oracle (oracle/wavelet_oracle.py), fwd+inv J=3 db4 symmetric on '
                          '2x3x512x512, %d reps' % reps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=128, help='images per GPU (BASELINE configs[1]: 128)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the DTCWT / ScatLayer / fp16 context timings')
    args = ap.parse_args()

    import __graft_entry__ as ge
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if rank == 0:
        ge.build()
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import parallel

    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)   # nccl == RCCL on ROCm
        dist.barrier()

    N, C, H, W, J, wave, mode = args.batch, 3, 512, 512, 3, 'db4', 'symmetric'
    xfm = pw.DWTForward(J=J, wave=wave, mode=mode).to(dev)
    ifm = pw.DWTInverse(wave=wave, mode=mode).to(dev)
    if world > 1:
        parallel.broadcast_filter_banks(xfm, src=0)
        parallel.broadcast_filter_banks(ifm, src=0)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(N, C, H, W, device=dev, generator=g)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def step():
        with torch.no_grad():
            yl, yh = xfm(x)
            return ifm((yl, yh))

    for _ in range(args.warmup):
        rec = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    err = float((rec - x).abs().max() / x.abs().max())

    # ---- per-kernel roofline: forward transform timed alone with HIP events on the launch stream
    with torch.no_grad():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        xfm(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            yl, yh = xfm(x)
        e1.record()
        torch.cuda.synchronize()
        fwd_ms = e0.elapsed_time(e1) / args.steps
        ifm((yl, yh))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            ifm((yl, yh))
        e1.record()
        torch.cuda.synchronize()
        inv_ms = e0.elapsed_time(e1) / args.steps
    fwd_bytes = algorithmic_bytes_fwd(N, C, H, W, J, 8, 4)
    fwd_gbs = fwd_bytes / (fwd_ms * 1e-3) / 1e9
    inv_gbs = fwd_bytes / (inv_ms * 1e-3) / 1e9
    # dominant kernel = the level-1 analysis launch, timed alone: algorithmic bytes = x in, ll_1 + yh_0 out
    with torch.no_grad():
        xfm1 = pw.DWTForward(J=1, wave=wave, mode=mode).to(dev)
        xfm1(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            xfm1(x)
        e1.record()
        torch.cuda.synchronize()
        l1_ms = e0.elapsed_time(e1) / args.steps
    l1_bytes = algorithmic_bytes_fwd(N, C, H, W, 1, 8, 4)
    l1_gbs = l1_bytes / (l1_ms * 1e-3) / 1e9
    # what a plain device copy of the same footprint achieves on this box (read + write bytes / time)
    with torch.no_grad():
        cdst = torch.empty_like(x)
        cdst.copy_(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            cdst.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2 * x.numel() * 4 / (e0.elapsed_time(e1) / args.steps * 1e-3) / 1e9
        del cdst
    # the other BASELINE configs (parity-test cases, not the metric): timed once on rank 0 at N=1 as context
    other = None
    if world == 1 and not args.no_other_configs:
        other = {}

        def timed(fn, n=10):
            fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        with torch.no_grad():
            xd = torch.randn(64, 3, 512, 512, device=dev)
            dx, di = pw.DTCWTForward(J=3).to(dev), pw.DTCWTInverse().to(dev)
            dyl, dyh = dx(xd)
            tf, ti = timed(lambda: dx(xd)), timed(lambda: di((dyl, dyh)))
            other['dtcwt_j3_near_sym_a_qshift_a_64x3x512x512_fp32'] = {
                'fwd_ms': round(tf, 4), 'inv_ms': round(ti, 4), 'fwd_inv_mpix_s': round(xd.numel() / (tf + ti) / 1e3, 1),
                'fwd_frac_of_hbm_peak_at_20B_per_px': round(20 * xd.numel() / (tf * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del xd, dyl, dyh
            xs = torch.randn(256, 3, 256, 256, device=dev)
            sl = pw.ScatLayer().to(dev)
            ts = timed(lambda: sl(xs))
            other['scatlayer_256x3x256x256_fp32_one_gpu'] = {
                'fwd_ms': round(ts, 4), 'mpix_s': round(xs.numel() / ts / 1e3, 1),
                'frac_of_hbm_peak_at_11B_per_px': round(11 * xs.numel() / (ts * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del xs
            xh = torch.randn(8, 16, 2048, 2048, device=dev).half()
            hx = pw.DWTForward(J=4, wave='db8', mode='periodization').to(dev).half()
            th = timed(lambda: hx(xh), 5)
            other['dwt_j4_db8_periodization_8x16x2048x2048_fp16'] = {
                'fwd_ms': round(th, 4), 'mpix_s': round(xh.numel() / th / 1e3, 1),
                'frac_of_hbm_peak_at_4B_per_px': round(4 * xh.numel() / (th * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                'note': 'batch reduced from 32 to 8 images of 16 channels'}
            del xh
    info = pw.engine_info(xfm, x)
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(info['fwd_kernel'], {}).get('hbm_bytes_corrected')
        except Exception:
            traffic = None

    if rank == 0:
        pixels = world * N * C * H * W
        out = {
            'metric': 'Mpixels/s fwd+inv DWT J=3 db4, Nx3x512x512 fp32; % HBM roofline',
            'value': round(pixels * args.steps / dt / 1e6, 1),
            'unit': 'Mpixels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'DWTForward+DWTInverse J=3 db4 symmetric, %dx3x512x512 fp32 per GPU '
                                   '(BASELINE configs[1])' % N,
                       'global_batch': world * N, 'parallelism': 'batch-sharded x%d, no data-path collective' % world,
                       'fwd_path': info['fwd_path'], 'inv_path': info['inv_path']},
            'roofline': {'bound': 'hbm', 'kernel': info['fwd_kernel'] + ' (level-1 launch)',
                         'achieved': round(l1_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(l1_gbs / HBM_PEAK_GBS, 4), 'traffic': traffic,
                         'algorithmic_bytes_per_launch': l1_bytes, 'avg_launch_ms': round(l1_ms, 4),
                         'device_copy_gbs': round(copy_gbs, 1), 'frac_of_device_copy': round(l1_gbs / copy_gbs, 4),
                         'forward_all_levels': {'achieved': round(fwd_gbs, 1), 'frac': round(fwd_gbs / HBM_PEAK_GBS, 4),
                                                'algorithmic_bytes': fwd_bytes, 'avg_ms': round(fwd_ms, 4),
                                                'launches': info['fwd_launches']},
                         'inverse': {'kernel': info['inv_kernel'], 'achieved': round(inv_gbs, 1),
                                     'frac': round(inv_gbs / HBM_PEAK_GBS, 4), 'avg_ms': round(inv_ms, 4),
                                     'launches_per_inverse': info['inv_launches']}},
            'fwd_mpix_s': round(N * C * H * W / (fwd_ms * 1e-3) / 1e6, 1),
            'inv_mpix_s': round(N * C * H * W / (inv_ms * 1e-3) / 1e6, 1),
            'roundtrip_rel_err': err,
        }
        if other is not None:
            out['other_configs'] = other
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args)
        elif world > 1:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
