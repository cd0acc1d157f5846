"""world_size-2 gloo test of the batch-sharded path (runs on CPU): filter banks are broadcast from
rank 0, each rank transforms only its shard (on the host emulation of the kernels), and the
gathered result equals the single-process transform of the full batch."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['OMP_NUM_THREADS'] = '2'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import emu_backend
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import parallel
    torch.manual_seed(0)
    x = torch.randn(5, 2, 24, 28)           # same full batch on every rank (odd: uneven shards)
    xfm = pw.DWTForward(J=2, wave='db2', mode='symmetric')
    if rank != 0:                           # corrupt the non-root taps: the broadcast must fix them
        for b in xfm.buffers():
            b.zero_()
    parallel.broadcast_filter_banks(xfm, src=0)
    with emu_backend.emulated():
        yl, yh = xfm(parallel.shard_batch(x))
        full_yl, full_yh = xfm(x)
    gl = parallel.gather_batch(yl, x.shape[0])
    gh = parallel.gather_batch(yh[1], x.shape[0])
    ok = torch.equal(gl, full_yl) and torch.equal(gh, full_yh[1]) and yl.shape[0] == (3 if rank == 0 else 2)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_batch_sharding_world2():
    import emu_backend
    emu_backend.handle()   # build the emulator once, before forking workers
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:      # a port nobody holds (consecutive runs from one pytest process must not meet the previous run's socket)
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, 'worker failed (exit code %r)' % p.exitcode
    res = sorted(q.get(timeout=10) for _ in procs)
    assert res == [(0, True), (1, True)], res


def test_shard_bounds():
    from pytorch_wavelets_amd import parallel
    for n in (0, 1, 7, 128, 129):
        for w in (1, 2, 3, 8):
            b = [parallel.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def _run_bench_emulated(extra):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket() as sk:      # a port nobody holds (consecutive runs from one pytest process must not meet the previous run's socket)
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '1', '--emulate'] + extra
    env = dict(os.environ, OMP_NUM_THREADS='2')
    res = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    err = res.stderr.decode()
    assert res.returncode == 0, err[max(0, err.find('Traceback')):][:3000] + '\n...\n' + err[-1500:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, lines   # rank 0 only
    return json.loads(lines[0])


def test_bench_other_configs_multi_rank_on_emulator():
    """The same world>1 flow for BASELINE configs[3] (ScatLayer: 256 images - here 5 - split over the ranks, strong
    scaling) and configs[2] (DTCWT forward + inverse, weak), on the host emulation with two gloo ranks."""
    out = _run_bench_emulated(['--config', 'scat', '--batch', '5'])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and out['config']['global_batch'] == 5
    assert 'ScatLayer' in out['metric'] and out['config']['step'] == 'forward' and 'inverse' not in out['roofline']
    assert abs(out['value'] - 5 * 3 * 32 * 32 / (out['ms_per_step'] * 1e-3) / 1e6) <= 0.06
    out = _run_bench_emulated(['--config', 'dtcwt', '--batch', '1'])
    assert out['scaling'] == 'weak' and out['config']['global_batch'] == 2 and out['roundtrip_rel_err'] < 1e-5
    assert 'inverse' in out['roofline'] and 'closure' in out['roofline']   # (wall-clock noise on the emulator: no value check)


def test_bench_multi_rank_control_flow_on_emulator():
    """bench.py's world>1 branch (process-group init, filter-bank broadcast, barriers, MAX all-reduce of the timed
    region, one JSON line from rank 0 only) executed for real with two gloo ranks on the host emulation of the
    kernels - the driver launches exactly this with nccl on 2/4/8 GPUs."""
    out = _run_bench_emulated(['--batch', '2'])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['scaling'] == 'weak' and out['cpu_baseline'] is None
    assert out['config']['global_batch'] == 4 and out['roundtrip_rel_err'] < 1e-5
    # whole-job pixels / max-over-ranks time (value is printed with one decimal)
    assert out['value'] > 0 and abs(out['value'] - 2 * 2 * 3 * 64 * 64 / (out['ms_per_step'] * 1e-3) / 1e6) <= 0.06
    assert out['cold']['ms_per_step'] > 0 and ('WlAfbSmall' in out['roofline']['kernel'] or 'WlAfbRows' in out['roofline']['kernel']) and ('WlSfbSmall' in out['roofline']['inverse']['kernel'] or 'WlSfbRows' in out['roofline']['inverse']['kernel'])   # (64 x 64 planes of the emulated run: the small-plane kernels)
