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


def source_digest():
    """sha256 over the engine's sources: profiles/*_hbm_traffic.json records the digest of the build it was measured
    on, and the roofline only quotes it when it matches what is running."""
    import hashlib
    h = hashlib.sha256()
    for d, exts in ((os.path.join(ROOT, 'pytorch_wavelets_amd', 'csrc'), ('.h', '.inc', '.hip')),
                    (os.path.join(ROOT, 'include'), ('.h',))):
        for f in sorted(os.listdir(d)):
            if f.endswith(exts):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def cpu_baseline(args):
    """The reference's CPU path, restated: oracle/torch_cpu.py = its gather + grouped conv2d / conv_transpose2d
    formulation on PyTorch-CPU (the reference itself is Python on ATen and cannot travel to this box; the restatement is
    pinned to its golden vectors by tests/test_oracle_golden.py), all host cores, on a bounded sample of the workload.
    The OpenMP C port of the numpy oracle (oracle/dwt_port.c) is timed next to it."""
    import numpy as np
    from oracle import torch_cpu as tc
    from pytorch_wavelets_amd import filters
    h0, h1 = filters.dwt_analysis_taps('db4')
    g0, g1 = filters.dwt_synthesis_taps('db4')
    ncores = os.cpu_count() or 1
    n = 16
    x = torch.randn(n, 3, 512, 512, generator=torch.Generator().manual_seed(0))

    def once():
        with torch.no_grad():
            yl, yh = tc.dwt_forward(x, 3, h0, h1, 'symmetric')
            return tc.dwt_inverse(yl, yh, g0, g1, 'symmetric')
    # ATen's grouped convolutions do not scale to hundreds of threads (256 threads measured 20x slower than 8 on the
    # MI355X host): time a few intra-op pool sizes for ~3 s each and report the best one, with the cores it used
    tried, best = {}, None
    for nt in sorted(set(t for t in (8, 16, 32, 64, ncores) if t <= ncores)):
        torch.set_num_threads(nt)
        once()
        reps, t0 = 0, time.perf_counter()
        while reps < 2 or time.perf_counter() - t0 < 3.0:
            once()
            reps += 1
        mp = x.numel() / ((time.perf_counter() - t0) / reps) / 1e6
        tried[str(nt)] = round(mp, 2)
        if best is None or mp > best[1]:
            best = (nt, mp, reps)
    out = {'value': round(best[1], 2), 'unit': 'Mpixels/s', 'cores': best[0], 'kind': 'restated-torch',
           'host_cores': ncores, 'mpix_s_by_torch_threads': tried,
           'sample': 'oracle/torch_cpu.py (the reference\'s conv2d / conv_transpose2d formulation on PyTorch-CPU, fp32), '
                     'fwd+inv J=3 db4 symmetric on %dx3x512x512, %d reps at the best thread count; the real reference '
                     'measured 23.0 Mpixels/s on the 8 vCPU of the authoring container with this torch build (profiles/r02_reference_cpu_timing.json; BASELINE.md quotes 16.2 for an older torch)' % (n, best[2])}
    try:
        from oracle import dwt_port
        rng = np.random.RandomState(0)
        m = max(2, min(64, ncores))
        xp = rng.randn(m, 3, 512, 512).astype(np.float32)
        dwt_port.fwd_inv(xp, 3, h0, h1, g0, g1, 'symmetric', threads=ncores)
        reps, t0 = 0, time.perf_counter()
        while reps < 3 or time.perf_counter() - t0 < 5.0:
            dwt_port.fwd_inv(xp, 3, h0, h1, g0, g1, 'symmetric', threads=ncores)
            reps += 1
        out['c_port'] = {'value': round(xp.size / ((time.perf_counter() - t0) / reps) / 1e6, 2), 'unit': 'Mpixels/s',
                         'kind': 'port', 'sample': 'oracle/dwt_port.c (OpenMP, one plane per thread) on %dx3x512x512' % m}
    except Exception as e:   # the C port is optional
        out['c_port'] = {'error': str(e)[:80]}
    return out


def _kernel_name(lib):
    """Functor of the kernel this thread launched last, as the engine reports it (wl_last_kernel)."""
    raw = lib.wl_last_kernel().decode()
    return raw.split('K = ')[-1].rstrip(']') if 'K = ' in raw else raw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=128, help='images per GPU (BASELINE configs[1]: 128)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the DTCWT / ScatLayer / fp16 context timings')
    ap.add_argument('--emulate', action='store_true',
                    help='TEST ONLY: run the whole harness on CPU tensors through the host emulation of the kernels '
                         '(tests/emu) with the gloo backend - exercises the multi-rank control flow without a GPU')
    args = ap.parse_args()

    import __graft_entry__ as ge
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    emu = args.emulate
    if emu:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import emu_backend
        from pytorch_wavelets_amd import ops
        ops._TEST_BACKEND = lib = emu_backend.handle()
    else:
        if rank == 0:
            ge.build()
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import parallel
    if not emu:
        from pytorch_wavelets_amd import _lib
        assert torch.cuda.is_available(), 'bench.py needs a GPU'
        dev = torch.device('cuda', local_rank)
        torch.cuda.set_device(dev)
    else:
        dev = torch.device('cpu')
    if world > 1:
        import torch.distributed as dist
        if emu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)   # nccl == RCCL on ROCm
        dist.barrier()
    if not emu:
        lib = _lib.get()

    def sync():
        if not emu:
            torch.cuda.synchronize()

    class _Timer(object):
        """HIP events on the launch stream (wall clock in emulation mode)."""
        def __init__(self):
            self.e0 = self.e1 = None
            if not emu:
                self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def run(self, fn, n):
            for _ in range(1 if emu else 60):   # long enough for the clocks to settle on the new load pattern
                fn()
            sync()
            if emu:
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                return (time.perf_counter() - t0) * 1e3 / n
            self.e0.record()
            for _ in range(n):
                fn()
            self.e1.record()
            torch.cuda.synchronize()
            return self.e0.elapsed_time(self.e1) / n
    timer = _Timer()

    N, C, H, W, J, wave, mode = args.batch, 3, 512, 512, 3, 'db4', 'symmetric'
    if emu:
        H = W = 64
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
        sync()

    def step():
        with torch.no_grad():
            yl, yh = xfm(x)
            return ifm((yl, yh))

    # The GPU takes ~20 ms of continuous work to reach its steady clocks (tools/gpu_rampup_probe.py: the first ~100
    # launches of a cold process run 10 % slower).  A fixed, untimed spin-up precedes the W warmup steps; a job of any
    # realistic length spends its life in the steady state.
    RAMP_STEPS = 0 if emu else 100
    for _ in range(RAMP_STEPS):
        step()
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

    # ---- per-kernel roofline.  The forward transform is ONE launch of the streaming kernel (all J levels, LL_j in
    # LDS): it is the dominant kernel, timed alone with HIP events on the launch stream; its algorithmic bytes are
    # x in + yl, yh[j] out.  The kernel names are the engine's own report of what it dispatched.
    with torch.no_grad():
        yl, yh = xfm(x)
        fwd_launches = 0
        # Per-kernel durations are taken LIVE in the step loop (the same alternating forward / inverse stream as the timed
        # region, K more steps right behind it), HIP events around each transform on the launch stream: a loop of one
        # kernel alone measures the power-management transient of a changed load pattern, not the kernel (rocprofv3
        # trace of this command: the same launch takes 170 us in the step loop and 205-213 us in the first 30 launches
        # of a forward-only loop).  The launch is asynchronous and far shorter on the host than on the GPU, so the
        # stream never runs dry between the events.
        if emu:
            fwd_ms = timer.run(lambda: xfm(x), args.steps)
            fwd_kernel = _kernel_name(lib)
            inv_ms = timer.run(lambda: ifm((yl, yh)), args.steps)
            inv_kernel = _kernel_name(lib)
        else:
            for _ in range(10):
                step()
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
            for k in range(args.steps):
                ev[k][0].record()
                a_, b_ = xfm(x)
                if k == 0:
                    fwd_kernel = _kernel_name(lib)
                ev[k][1].record()
                ifm((a_, b_))
                if k == 0:
                    inv_kernel = _kernel_name(lib)
                ev[k][2].record()
            torch.cuda.synchronize()
            fwd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
            inv_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    from pytorch_wavelets_amd.dwt import lowlevel as _ll
    fused = 'WlAfbRows' in fwd_kernel
    fwd_launches = 1 if fused else J
    fwd_bytes = algorithmic_bytes_fwd(N, C, H, W, J, 8, 4)
    fwd_gbs = fwd_bytes / (fwd_ms * 1e-3) / 1e9
    inv_gbs = fwd_bytes / (inv_ms * 1e-3) / 1e9
    inv_fused = 'WlSfbRows' in inv_kernel
    # the same transforms as one tile-kernel launch per level (the round-1 path), for the record
    with torch.no_grad():
        _ll.FUSED_LEVELS = False
        tile_ms = timer.run(lambda: xfm(x), args.steps)
        tile_kernel = _kernel_name(lib)
        inv_tile_ms = timer.run(lambda: ifm((yl, yh)), args.steps)
        inv_tile_kernel = _kernel_name(lib)
        _ll.FUSED_LEVELS = True
    # what a plain device copy of the same footprint achieves on this box (read + write bytes / time)
    with torch.no_grad():
        cdst = torch.empty_like(x)
        copy_gbs = 2 * x.numel() * 4 / (timer.run(lambda: cdst.copy_(x), args.steps) * 1e-3) / 1e9
        del cdst
    # the other BASELINE configs (parity-test cases, not the metric): timed once on rank 0 at N=1 as context
    other = None
    if world == 1 and not args.no_other_configs and not emu:
        other = {}
        with torch.no_grad():
            xd = torch.randn(64, 3, 512, 512, device=dev)
            dx, di = pw.DTCWTForward(J=3).to(dev), pw.DTCWTInverse().to(dev)
            dyl, dyh = dx(xd)
            tf, ti = timer.run(lambda: dx(xd), 10), timer.run(lambda: di((dyl, dyh)), 10)
            other['dtcwt_j3_near_sym_a_qshift_a_64x3x512x512_fp32'] = {
                'fwd_ms': round(tf, 4), 'inv_ms': round(ti, 4), 'fwd_inv_mpix_s': round(xd.numel() / (tf + ti) / 1e3, 1),
                'fwd_frac_of_hbm_peak_at_20B_per_px': round(20 * xd.numel() / (tf * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                'inv_frac_of_hbm_peak_at_20B_per_px': round(20 * xd.numel() / (ti * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del xd, dyl, dyh
            xs = torch.randn(256, 3, 256, 256, device=dev)
            sl = pw.ScatLayer().to(dev)
            ts = timer.run(lambda: sl(xs), 10)
            other['scatlayer_256x3x256x256_fp32_one_gpu'] = {
                'fwd_ms': round(ts, 4), 'mpix_s': round(xs.numel() / ts / 1e3, 1),
                'frac_of_hbm_peak_at_11B_per_px': round(11 * xs.numel() / (ts * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del xs
            xh = torch.randn(32, 16, 2048, 2048, device=dev, dtype=torch.float16)   # configs[4] at its full size (4.3 GB)
            hx = pw.DWTForward(J=4, wave='db8', mode='periodization').to(dev).half()
            hi = pw.DWTInverse(wave='db8', mode='periodization').to(dev).half()
            hyl, hyh = hx(xh)
            th = timer.run(lambda: hx(xh), 3)
            hk = _kernel_name(lib)
            tih = timer.run(lambda: hi((hyl, hyh)), 3)
            other['dwt_j4_db8_periodization_32x16x2048x2048_fp16'] = {
                'fwd_ms': round(th, 4), 'inv_ms': round(tih, 4), 'fwd_mpix_s': round(xh.numel() / th / 1e3, 1),
                'fwd_frac_of_hbm_peak_at_4B_per_px': round(4 * xh.numel() / (th * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                'inv_frac_of_hbm_peak_at_4B_per_px': round(4 * xh.numel() / (tih * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                'last_fwd_kernel': hk}
            del xh, hyl, hyh
    # HBM traffic of the dominant kernel: only from a PMC summary measured on THIS build of the sources
    traffic = inv_traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'r02_hbm_traffic.json')
    if os.path.exists(tpath) and not emu:
        try:
            tj = json.load(open(tpath))
            if tj.get('source_digest') == source_digest():
                for k, v in tj.get('kernels', {}).items():   # rocprof prints defaulted template arguments too
                    if k.strip().startswith(fwd_kernel.rstrip('>')):
                        traffic = v.get('hbm_bytes_corrected')
                    if k.strip().startswith(inv_kernel.rstrip('>')):
                        inv_traffic = v.get('hbm_bytes_corrected')
        except Exception:
            traffic = inv_traffic = None

    if rank == 0:
        pixels = world * N * C * H * W
        out = {
            'metric': 'Mpixels/s fwd+inv DWT J=3 db4, Nx3x512x512 fp32; % HBM roofline',
            'value': round(pixels * args.steps / dt / 1e6, 1),
            'unit': 'Mpixels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'clock_ramp_steps_untimed': RAMP_STEPS,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'DWTForward+DWTInverse J=3 db4 symmetric, %dx3x%dx%d fp32 per GPU '
                                   '(BASELINE configs[1])' % (N, H, W),
                       'global_batch': world * N, 'parallelism': 'batch-sharded x%d, no data-path collective' % world,
                       'fwd_path': ('one launch of the streaming kernel for all %d levels (LL_j in LDS)' % J) if fused
                                   else 'one tile-kernel launch per level',
                       'inv_path': ('one launch of the streaming kernel for all %d levels (low-passes in LDS)' % J) if inv_fused
                                   else 'one polyphase tile-kernel launch per level'},
            'roofline': {'bound': 'hbm', 'kernel': fwd_kernel + (' (all %d levels, one launch)' % J if fused else ' (last level)'),
                         'how_timed': 'HIP events around each transform inside %d further steps of the same forward/inverse '
                                      'stream (launches serialised by the events; in the free-running timed region consecutive '
                                      'launches overlap their tails, so ms_per_step < forward + inverse)' % args.steps,
                         'achieved': round(fwd_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(fwd_gbs / HBM_PEAK_GBS, 4), 'traffic': traffic,
                         'algorithmic_bytes_per_launch': fwd_bytes, 'avg_launch_ms': round(fwd_ms, 4),
                         'launches_per_forward': fwd_launches,
                         'device_copy_gbs': round(copy_gbs, 1), 'frac_of_device_copy': round(fwd_gbs / copy_gbs, 4),
                         'forward_per_level_tile_kernels': {'kernel': tile_kernel, 'avg_ms': round(tile_ms, 4),
                                                            'frac': round(fwd_bytes / (tile_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                         'inverse': {'kernel': inv_kernel + (' (all %d levels, one launch)' % J if inv_fused else ' (last level)'),
                                     'achieved': round(inv_gbs, 1),
                                     'frac': round(inv_gbs / HBM_PEAK_GBS, 4), 'avg_ms': round(inv_ms, 4),
                                     'frac_of_device_copy': round(inv_gbs / copy_gbs, 4), 'traffic': inv_traffic,
                                     'launches_per_inverse': 1 if inv_fused else J,
                                     'inverse_per_level_tile_kernels': {
                                         'kernel': inv_tile_kernel, 'avg_ms': round(inv_tile_ms, 4),
                                         'frac': round(fwd_bytes / (inv_tile_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}},
            'fwd_mpix_s': round(N * C * H * W / (fwd_ms * 1e-3) / 1e6, 1),
            'inv_mpix_s': round(N * C * H * W / (inv_ms * 1e-3) / 1e6, 1),
            'roundtrip_rel_err': err,
        }
        if emu:
            out['data'] = 'synthetic (HOST EMULATION of the kernels: control-flow test, not a measurement)'
        if other is not None:
            out['other_configs'] = other
        if not args.no_cpu_baseline and world == 1 and not emu:
            out['cpu_baseline'] = cpu_baseline(args)
        elif world > 1 or emu:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
